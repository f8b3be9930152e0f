// conv1x.hip -- the z-independent bulk convolution of the coupling networks by Winograd F(4x4,3x3) minimal filtering on the
// fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Computes the same sums as conv1_ft_kernel / conv1_ft_wino_kernel of conv1.hip (reference interflow/models/modules:
// flow.Conv2d 3x3 over the 320 conditioning channels of fFeatures and fAffine, flow.py:25-64,
// FlowAffineCouplingsAblation.py:219-233,248-256, plus fFeatures' ActNorm / ReLU / 1x1 / ActNorm / ReLU) with 36
// multiplications per 4x4 outputs, input channel and output channel instead of the direct form's 144 (F(2x2,3x3): 64):
//   Y = A^T [ sum_ci U[co][ci] (.) V[ci] ] A,   U = G g G^T (host, float64 -> fp32, pack.py),   V = B^T d B
// for every 4x4 output tile and its 6x6 input window d (Toom-Cook matrices for the points 0, +-3/4, +-3/2, inf: ifl_common.h).  For each of
// the 36 positions (i, jc) of the transformed domain the sum over input channels is a GEMM
//   M[co][tile] = U[co][ci] V[ci][tile]     (fp32 throughout; the result differs from the direct form by fp32 rounding only:
//                                            tests/test_wino_numerics.py, tests/test_gpu_parity.py).
//
// Two kernels:
//   wino4_input_kernel     V = B^T d B of every tile and channel, once per loaded conditioning map (ifl_load_cond): ringed
//                          channels-last features -> vg[tile group][sub-chunk of 8 channels][position][quad][tile][4], i.e. per
//                          (group of 32 tiles, sub-chunk) the 36,864-byte LDS image the GEMM kernel multiplies.  HBM-bound.
//                          Every (network, step) of every launch reuses it: 32 uses per pass.
//   conv1_ft_wino4_kernel  workgroup = one network x one step x 32 consecutive tiles (raster order over image, tile row, tile
//                          column) = 512 output pixels x 64 channels; 8 waves = 2 halves of the output channels x 4 blocks of
//                          3x3 positions, 144 accumulator registers per wave.  The 320 input channels are walked in 40
//                          sub-chunks of 8; per sub-chunk, between two barriers, every wave multiplies its 9 positions x 4
//                          k-steps on V[s & 1] (one 16-byte LDS read per position; weights streamed global -> VGPR through a
//                          9-deep ring, fragment order, contiguous per wave), stores the image of sub-chunk s + 1 (loaded one
//                          sub-chunk earlier, 4.5 x 16 bytes per thread) into V[(s + 1) & 1] and loads that of s + 2.
// LDS image [36 positions][2 quads][32 tiles][4 channels]: the 16-byte reads of a wave fall into distinct banks.
// Epilogue: the output transform A^T M A is linear, so each wave reduces its 3x3 block to partial sums and exchanges them
// with its partner block through LDS, first along i, then along jc; afterwards wave (bi, bj) holds output pixels
// (2 bi + {0,1}, 2 bj + {0,1}) of its 32 tiles and 32 channels in the usual accumulator layout.  fAffine's raw sums are stored
// as they are; for fFeatures the two channel halves trade pixel sets and finish_and_store (conv1_fin.h) does the rest.
#include "ifl_common.h"
#include <type_traits>

namespace {

#include "conv1_fin.h"

constexpr int XN = IFL_FT / IFL_XCH;          // 40 sub-chunks
constexpr int XBUF = 36 * 32 * IFL_XCH;       // floats per V image (one tile group, one sub-chunk)
constexpr int XLDS = 8 * 4 * 16 * 64;         // LDS floats: the epilogue's exchange (8 waves x 4 x f32x16 per lane) > two V images
constexpr int XRD = 9;                        // weight ring depth (positions)
constexpr int XSLOT = 16 * 64;                // floats per exchanged f32x16 (all lanes)
static_assert(3 * XBUF <= XLDS, "three V images in the kernel's LDS array");
#ifndef IFL_X_AUX_W
#define IFL_X_AUX_W 0                         // cache policy bits of the weight loads (experiments: 2 = nt, 16 = sc1)
#endif
#ifndef IFL_X_AUX_V
#define IFL_X_AUX_V 0                         // ... of the image loads
#endif

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 y) {
    return f32x4{__builtin_fmaf(s, x[0], y[0]), __builtin_fmaf(s, x[1], y[1]), __builtin_fmaf(s, x[2], y[2]), __builtin_fmaf(s, x[3], y[3])};
}

// B^T x for one column / row of six values (points 0, +-a, +-b, inf; ifl_common.h):
//   t0 = AB x0 - (A+B) x2 + x4      t1,2 = (x4 - B x2) +- a (x3 - B x1)      t3,4 = (x4 - A x2) +- b (x3 - A x1)      t5 = AB x1 - (A+B) x3 + x5
// `get(k)` yields x_k, `put(k, v)` consumes t_k; even inputs are fetched and folded first, so few values are live at a time.
template <class Get, class Put>
__device__ __forceinline__ void bt6(Get get, Put put) {
    f32x4 a, c;
    {
        const f32x4 x2 = get(2), x4 = get(4);
        a = fma4(-IFL_W4_B, x2, x4);
        c = fma4(-IFL_W4_A, x2, x4);
        put(0, fma4(IFL_W4_AB, get(0), fma4(-IFL_W4_ApB, x2, x4)));
    }
    f32x4 b, e;
    {
        const f32x4 x1 = get(1), x3 = get(3);
        b = fma4(-IFL_W4_B, x1, x3);
        e = fma4(-IFL_W4_A, x1, x3);
        put(5, fma4(IFL_W4_AB, x1, fma4(-IFL_W4_ApB, x3, get(5))));
    }
    put(1, fma4(IFL_W4_a, b, a));
    put(2, fma4(-IFL_W4_a, b, a));
    put(3, fma4(IFL_W4_b, e, c));
    put(4, fma4(-IFL_W4_b, e, c));
}

__device__ __forceinline__ void st16(float* p, const f32x16& v) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) *(f32x4*)(p + gq * 256) = f32x4{v[gq * 4], v[gq * 4 + 1], v[gq * 4 + 2], v[gq * 4 + 3]};
}
__device__ __forceinline__ void add16(const float* p, f32x16& v) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 x = *(const f32x4*)(p + gq * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[gq * 4 + e] += x[e];
    }
}
__device__ __forceinline__ void ld16(const float* p, f32x16& v) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 x = *(const f32x4*)(p + gq * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[gq * 4 + e] = x[e];
    }
}

// One direction of the output transform for a wave that holds three consecutive of the six values (blk 0: 0,1,2; blk 1:
// 3,4,5).  A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 A A B B 0; 0 a^3 -a^3 b^3 -b^3 1] (ifl_common.h).  keep[0..1] are this block's
// parts of the two outputs it finishes (outputs 2 blk, 2 blk + 1), send[0..1] its parts of the partner's two outputs.
__device__ __forceinline__ void at_part(int blk, const f32x16& m0, const f32x16& m1, const f32x16& m2,
                                        f32x16& k0, f32x16& k1, f32x16& s0, f32x16& s1)
{
    if (blk == 0) {
        const f32x16 sm = m1 + m2, df = m1 - m2;
        s0 = sm * IFL_W4_A;      // output 2
        s1 = df * IFL_W4_a3;     // output 3
        k0 = m0 + sm;            // output 0
        k1 = df * IFL_W4_a;      // output 1
    } else {
        s0 = m0 + m1;            // output 0
        const f32x16 df = m0 - m1;
        s1 = df * IFL_W4_b;      // output 1
        k0 = s0 * IFL_W4_B;      // output 2
        k1 = df * IFL_W4_b3 + m2;      // output 3
    }
}

// Build with -DIFL_EXP_TIME (tools/build_exp.sh) to record per-wave timestamps (tools/exp_x4_timeline.py); never defined in
// the product build.
#ifdef IFL_EXP_TIME
#ifndef IFL_EXP_TIME_LVL
#define IFL_EXP_TIME_LVL 1
#endif
__device__ unsigned long long ifl_dbgx[8192 * 8 * 8];
#define XT_STAMP(k) do { if (LVL == IFL_EXP_TIME_LVL && lane == 0 && blockIdx.x < 8192) ifl_dbgx[(blockIdx.x * 8 + wave) * 8 + (k)] = wall_clock64(); } while (0)
#define XT_PUT(k, v) do { if (LVL == IFL_EXP_TIME_LVL && lane == 0 && blockIdx.x < 8192) ifl_dbgx[(blockIdx.x * 8 + wave) * 8 + (k)] = (v); } while (0)
#define XT_CLK() clock64()
#else
#define XT_STAMP(k) do {} while (0)
#define XT_PUT(k, v) do {} while (0)
#define XT_CLK() 0ll
#endif

// ---- input transform: ringed channels-last features -> Winograd-domain images -------------------------------------------
// Thread = (tile of the group, quad of channels); block = 32 tiles x 8 quads (4 sub-chunks); grid (groups, 10).  Loads: 8 lanes
// cover a pixel's 128 contiguous bytes; stores: the 32 tiles of a (position, quad) are 512 contiguous bytes.  Window pixels
// beyond the ringed plane (tiles overhanging a plane whose size is not a multiple of 4) and tiles past the last one are zeros.
__global__ __launch_bounds__(256)
void wino4_input_kernel(const float* __restrict__ ftP, float* __restrict__ vg, IflXGeom g)
{
    const int grp = blockIdx.x, sb = blockIdx.y;
    const int tid = threadIdx.x, t = tid >> 3, q8 = tid & 7;
    const int tpi = g.TH * g.TW, W2 = g.W + 2;
    int gt = grp * 32 + t;
    const bool live = gt < g.ntiles;
    if (!live) gt = g.ntiles - 1;
    const int b = gt / tpi, r2 = gt - b * tpi;
    const int tr = r2 / g.TW, tc = r2 - tr * g.TW;
    const float* src = ftP + ((int64_t)(b * (g.H + 2) + 4 * tr) * W2 + 4 * tc) * IFL_FT + sb * 32 + q8 * 4;
    f32x4 T[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const bool cin = live && 4 * tc + c <= g.W + 1;
        bt6([&](int r) { return (cin && 4 * tr + r <= g.H + 1) ? *(const f32x4*)(src + (int64_t)(r * W2 + c) * IFL_FT) : f32x4{0.f, 0.f, 0.f, 0.f}; },
            [&](int i, f32x4 v) { T[i][c] = v; });
    }
    float* dst = vg + ((int64_t)(grp * XN + sb * 4 + (q8 >> 1)) * 36) * 256 + (q8 & 1) * 128 + t * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        bt6([&](int c) { return T[i][c]; }, [&](int jc, f32x4 v) { *(f32x4*)(dst + (i * 6 + jc) * 256) = v; });
}

template <int LVL>
__global__ __launch_bounds__(512, 1)
void conv1_ft_wino4_kernel(const float* __restrict__ vg, float* __restrict__ h2all, int64_t h2_step_stride,
                           const float* __restrict__ params, IflStepList steps, IflStepOff off, IflXGeom g, int ngroups, int ncombo,
                           int raw_both)
{
    __shared__ __attribute__((aligned(16))) float lds[XLDS];              // V[0], V[1]; the epilogue's exchange slots
    // Workgroup -> (network, step, tile group).  A workgroup streams its network's whole transformed filter bank (2.95 MB per
    // step) and every workgroup of a (network, step) pair streams the same one; consecutive workgroup ids go to consecutive
    // XCDs (id % 8).  With a multiple of 8 pairs and an even number of tile groups, an XCD takes BOTH networks of one step for
    // half of the tile groups: the two workgroups of a tile group are neighbours on the XCD and share the group's images
    // through its L2 (one fetch over the fabric instead of two: FETCH_SIZE per level-1 launch 3.3 GB against 5.0 GB with one
    // pair per XCD, profiles/r02_notes.md), while the two filter banks (5.9 MB), walked in step by the XCD's workgroups, still
    // stream through its 4 MB L2 once per round.  The XCDs walk the tile groups in step, so a group's images come from HBM
    // once and the Infinity Cache serves the other XCDs.  Otherwise: one pair per XCD, or plain raster.
    int grp, combo;
    {
        const int id = blockIdx.x;
        if ((ncombo & 7) == 0 && (ngroups & 1) == 0) {
            // launches of more than 8 pairs (8 steps: passes started early): the sets of 8 pairs are the fastest index, so an XCD
            // runs steps s and s + 4 of a tile group side by side and the group's images pass through its L2 once for both
            // (dominant launch 3.22 -> 3.08 ms in situ)
            // (two sets at a time: four -- ifl_cond_prepare's 16-step launches -- put eight filter banks through an XCD's L2 at once)
            const int x = id & 7, t = id >> 3, nset = ncombo >> 3, il = (nset & 1) ? 1 : 2;
            const int rest = t / il, tl = rest % ngroups, set = (rest / ngroups) * il + t % il;
            grp = (tl >> 1) * 2 + (x >> 2);
            combo = set * 8 + (x & 3) * 2 + (tl & 1);
        }
        else if ((ncombo & 7) == 0) { const int t = id >> 3; grp = t % ngroups; combo = (t / ngroups) * 8 + (id & 7); }
        else { grp = id % ngroups; combo = id / ngroups; }
    }
    const int step = combo >> 1, net = combo & 1;
    const float* prm = params + steps.off[step];
    float* h2P = h2all + (int64_t)step * h2_step_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int mh = wave >> 2, bi = (wave >> 1) & 1, bj = wave & 1;        // waves w and w + 4 (same block, other half) share a SIMD
    const int tpi = g.TH * g.TW;
    const int vrd = ((3 * bi) * 6 + 3 * bj) * 256 + h * 128 + j * 4;      // V[(3 bi + il) * 6 + 3 bj + jl][quad h][tile j]: + (il * 6 + jl) * 256

    f32x16 acc[9];

    // Loads are buffer loads: a descriptor in scalar registers + a constant per-lane offset + a scalar offset per load, so the
    // loop holds no 64-bit address arithmetic on the vector ALU and no address registers; reads past a descriptor's end
    // (the ring's last refills, IFL_X_AUX experiments) return zeros.
    // weights: [half][block][sub-chunk][position][lane][4], contiguous per wave
    const float* wp = prm + off.w1x[net] + (int64_t)(mh * 4 + bi * 2 + bj) * (XN * 9 * 256);       // wave-uniform
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, XN * 9 * 1024, 0x00020000);
    const int wl = lane * 16;                 // bytes
    f32x4 wq[XRD];
    auto load_w = [&](int idx) {              // fragment idx = sub-chunk * 9 + position of this wave's stream
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wl, idx * 1024, IFL_X_AUX_W));
    };

    // image staging: 36,864 bytes per sub-chunk, the same split for every wave (wave-dependent branches around loads make the
    // compiler's wait counts pessimistic): 16-byte pieces tid + 512 k for k = 0..3, then 8 bytes at 32768 + 8 tid.
    // TWO register sets: an image is loaded two sub-chunks before it is stored to LDS (a load under full traffic takes about
    // as long as one sub-chunk's 4,608 matrix-pipe cycles: with one set the later of a SIMD's two waves waited for its image
    // in every sub-chunk and the loop ran at that latency, profiles/r03_notes.md).
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(vg + (int64_t)grp * (XN * XBUF)), 0, XN * XBUF * 4, 0x00020000);
    const int vl = tid * 16, vl8 = 4 * 8192 + tid * 8;             // bytes
    struct Img { f32x4 v[4]; f32x2 t; };
    Img svA, svB;
    auto load_v = [&](Img& d, int s) {
#ifdef IFL_EXP_NOWIN
        return;
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k)
            d.v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vl, s * (XBUF * 4) + k * 8192, IFL_X_AUX_V));
        d.t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vrs, vl8, s * (XBUF * 4), IFL_X_AUX_V));
    };
    auto store_v = [&](float* V, const Img& d) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)((char*)V + k * 8192 + vl) = d.v[k];
        *(f32x2*)((char*)V + vl8) = d.t;
    };
#ifdef IFL_EXP_NOWIN
#pragma unroll
    for (int k = 0; k < 4; ++k) svA.v[k] = svB.v[k] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(tid + k);
    svA.t = svB.t = f32x2{1.f, 2.f} * (float)tid;
#endif

    // Three LDS images in rotation: sub-chunk s multiplies image s (buffer s % 3), stores image s + 2 (register set s & 1,
    // loaded two sub-chunks earlier) and loads image s + 4 into that set.  Image s + 1 has been complete since the previous
    // barrier, so the first two positions' operands of sub-chunk s + 1 are read BEFORE its barrier and its first MFMAs need no
    // LDS round trip after it; the barrier itself (all reads of image s - 1 over -> its buffer may be overwritten; image
    // s + 1 visible) sits behind the first two position pairs, where the matrix pipe has work queued.
    XT_STAMP(0);
    long long xt_bar = 0, xt_pass = 0;
    f32x4 xb[2];
    // Prologue: every load at once -- images 0 and 1 into registers the accumulators will use (they are cleared afterwards),
    // then, in the order of the steady state at the head of a sub-chunk, image 2; ring positions 0, 1; image 3; ring 2..8 --
    // one exposed round trip.  The order matters beyond that: the compiler's wait counts at the loop head are the stricter of
    // the entry's and the back edge's (with the ring filled first, every second sub-chunk waited for ALL outstanding loads);
    // pinned with sched_barrier, the scheduler otherwise reorders the loads.
    {
        Img t0, t1;
        load_v(t0, 0);
        load_v(t1, 1);
        load_v(svA, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k) { wq[k] = load_w(k); __builtin_amdgcn_sched_barrier(0); }
        load_v(svB, 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 2; k < XRD; ++k) { wq[k] = load_w(k); __builtin_amdgcn_sched_barrier(0); }
        store_v(lds, t0);
        store_v(lds + XBUF, t1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    __syncthreads();
    xb[0] = *(const f32x4*)(lds + vrd);
    xb[1] = *(const f32x4*)(lds + vrd + 256);

    // STORE / LOAD: whether images s + 2 / s + 4 exist, NEXT: whether sub-chunk s + 1 does -- compile-time, so that the loop
    // body has no branch around a load (the last four sub-chunks are peeled)
    auto sub_chunk = [&](int s, auto bi_, Img& sv, auto st, auto ld, auto nx) {
        constexpr int BI = decltype(bi_)::value;
        constexpr bool STORE = decltype(st)::value, LOAD = decltype(ld)::value, NEXT = decltype(nx)::value;
        const float* vr = lds + BI * XBUF + vrd;
        const float* vn = lds + ((BI + 1) % 3) * XBUF + vrd;
        float* vw = lds + ((BI + 2) % 3) * XBUF;
#pragma unroll
        for (int pp = 0; pp < 5; ++pp) {      // position pairs (0,1) (2,3) (4,5) (6,7) and 8 alone
            const int p0 = 2 * pp, np = pp < 4 ? 2 : 1;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < np; ++u)
#ifdef IFL_EXP_NOMFMA
                    acc[p0 + u][e] = __builtin_fmaf(wq[p0 + u][e], xb[u][e], acc[p0 + u][e]);
#else
                    acc[p0 + u] = mfma32(wq[p0 + u][e], xb[u][e], acc[p0 + u]);
#endif
#ifndef IFL_EXP_NOW
#pragma unroll
            for (int u = 0; u < np; ++u) wq[p0 + u] = load_w((s + 1) * 9 + p0 + u);
#endif
            if (pp < 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = p0 + 2 + u;
                    if (p < 9) xb[u] = *(const f32x4*)(vr + ((p / 3) * 6 + p % 3) * 256);
                }
            } else if (NEXT) {
                xb[0] = *(const f32x4*)(vn);
                xb[1] = *(const f32x4*)(vn + 256);
            }
            if (pp == 1) {
                { long long c0 = XT_CLK();
                __syncthreads();
                xt_bar += XT_CLK() - c0; }
                long long c0 = XT_CLK();
                if (STORE) store_v(vw, sv);
                if (LOAD) load_v(sv, s + 4);
                xt_pass += XT_CLK() - c0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    XT_STAMP(1);
    const long long xt_c1 = XT_CLK();
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    using Yes = std::true_type; using No = std::false_type;
    static_assert(XN == 40, "the tail below is written for 40 sub-chunks");
    for (int s = 0; s < 36; s += 6) {
        sub_chunk(s, T0{}, svA, Yes{}, Yes{}, Yes{});
        sub_chunk(s + 1, T1{}, svB, Yes{}, Yes{}, Yes{});
        sub_chunk(s + 2, T2{}, svA, Yes{}, Yes{}, Yes{});
        sub_chunk(s + 3, T0{}, svB, Yes{}, Yes{}, Yes{});
        sub_chunk(s + 4, T1{}, svA, Yes{}, Yes{}, Yes{});
        sub_chunk(s + 5, T2{}, svB, Yes{}, Yes{}, Yes{});
    }
    sub_chunk(36, T0{}, svA, Yes{}, No{}, Yes{});
    sub_chunk(37, T1{}, svB, Yes{}, No{}, Yes{});
    sub_chunk(38, T2{}, svA, No{}, No{}, Yes{});
    sub_chunk(39, T0{}, svB, No{}, No{}, No{});
    XT_PUT(5, (unsigned long long)xt_bar); XT_PUT(6, (unsigned long long)(XT_CLK() - xt_c1)); XT_PUT(7, (unsigned long long)xt_pass);
    XT_STAMP(2);
    (void)xt_bar; (void)xt_pass; (void)xt_c1;

    // ---- output transform ----------------------------------------------------------------------------------------------------
    // stage 1 along i (partner: other bi = wave ^ 2), in two rounds (jl = 0,1 then jl = 2: 32 + 16 slots of 4 KB)
    f32x16 P[2][3];
    {
        float* mine = lds + wave * (4 * XSLOT) + lane * 4;
        const float* theirs = lds + (wave ^ 2) * (4 * XSLOT) + lane * 4;
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);                      // every wave's GEMM reads of V are over
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
            f32x16 s0, s1;
            at_part(bi, acc[jl], acc[3 + jl], acc[6 + jl], P[0][jl], P[1][jl], s0, s1);
            st16(mine + jl * XSLOT, s0);
            st16(mine + (2 + jl) * XSLOT, s1);
            __builtin_amdgcn_sched_barrier(0);      // keep the partial sums of one column at a time in registers
        }
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) add16(theirs + (dy * 2 + jl) * XSLOT, P[dy][jl]);
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);
        {
            f32x16 s0, s1;
            at_part(bi, acc[2], acc[5], acc[8], P[0][2], P[1][2], s0, s1);
            st16(mine, s0);
            st16(mine + XSLOT, s1);
        }
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) add16(theirs + dy * XSLOT, P[dy][2]);
    }
    // stage 2 along jc (partner: other bj = wave ^ 1)
    f32x16 Y[2][2];
    {
        float* mine = lds + wave * (4 * XSLOT) + lane * 4;
        const float* theirs = lds + (wave ^ 1) * (4 * XSLOT) + lane * 4;
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            f32x16 s0, s1;
            at_part(bj, P[dy][0], P[dy][1], P[dy][2], Y[dy][0], Y[dy][1], s0, s1);
            st16(mine + (dy * 2) * XSLOT, s0);
            st16(mine + (dy * 2 + 1) * XSLOT, s1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) add16(theirs + (dy * 2 + dx) * XSLOT, Y[dy][dx]);
    }

    XT_STAMP(3);
    // ---- this lane's tile and output pixels ------------------------------------------------------------------------------------
    int64_t pix00 = -1;                       // ring pixel index of the tile's pixel (0, 0), or -1
    int ty = 0, tx = 0;
    {
        int gt = grp * 32 + j;
        if (gt < g.ntiles) {
            const int b = gt / tpi, r2 = gt - b * tpi;
            const int tr = r2 / g.TW, tc = r2 - tr * g.TW;
            ty = 4 * tr; tx = 4 * tc;
            pix00 = ((int64_t)b * (g.H + 2) + ty + 1) * (g.W + 2) + tx + 1;
        }
    }
    auto out_pix = [&](int dy, int dx) -> int64_t {
        return (pix00 >= 0 && ty + dy < g.H && tx + dx < g.W) ? pix00 + dy * (g.W + 2) + dx : -1;
    };

    if (net == 0 || raw_both) {
        // fAffine: raw sums of this wave's 32 channels, finished by conv1_finish_kernel once z1 of the step exists
        // (raw_both: the training path takes fFeatures' raw sums too, in channels [64, 128) of the slab)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t op = out_pix(2 * bi + dy, 2 * bj + dx);
                if (op < 0) continue;
                float* dst = h2P + op * IFL_H2 + net * IFL_HID + mh * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(f32x4*)(dst + gq * 8) = f32x4{Y[dy][dx][gq * 4], Y[dy][dx][gq * 4 + 1], Y[dy][dx][gq * 4 + 2], Y[dy][dx][gq * 4 + 3]};
            }
    } else {
        // fFeatures: half mh keeps pixel row dy = mh of the block and takes the other half's channels of it (partner wave ^ 4)
        float* mine = lds + wave * (2 * XSLOT) + lane * 4;
        const float* theirs = lds + (wave ^ 4) * (2 * XSLOT) + lane * 4;
        // (compile-time indices only: a register array indexed by the runtime `mh` would be placed in scratch memory)
        auto finish_half = [&](auto mhc) {
            constexpr int MH = decltype(mhc)::value;
            __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) st16(mine + dx * XSLOT, Y[MH ^ 1][dx]);
            __syncthreads(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {  // one pixel set at a time keeps the 1x1 conv's operands and results in registers
                f32x16 a1[2][1];
                ld16(theirs + dx * XSLOT, a1[MH ^ 1][0]);
                a1[MH][0] = Y[MH][dx];
                int64_t opix[1] = {out_pix(2 * bi + MH, 2 * bj + dx)};
                finish_and_store<1, IFL_H2>(a1, prm + off.an1[1], prm + off.w2[1], prm + off.an2[1], h2P + IFL_HID, opix, lane, h);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (mh == 0) finish_half(std::integral_constant<int, 0>{});
        else finish_half(std::integral_constant<int, 1>{});
    }
    XT_STAMP(4);
}

}  // namespace

int ifl_k_wino4_input(const float* ft_ring, float* vg, int B, int H, int W, hipStream_t st)
{
    const IflXGeom g = ifl_xgeom(B, H, W);
    dim3 grid((g.ntiles + 31) / 32, XN / 4), block(256);
    hipLaunchKernelGGL(wino4_input_kernel, grid, block, 0, st, ft_ring, vg, g);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_conv1_ft_wino4(const float* vg, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                         int nsteps, const IflStepOff& off, int B, int H, int W, int level, hipStream_t st, int raw_both)
{
    if (nsteps <= 0) return 0;
    const IflXGeom g = ifl_xgeom(B, H, W);
    const int ngroups = (g.ntiles + 31) / 32, ncombo = 2 * nsteps;
    if ((int64_t)ngroups * ncombo >= (int64_t)1 << 31) {
        ifl_set_error("conv1_ft (Winograd 4x4): %d tile groups x %d exceed the grid", ngroups, ncombo);
        return IFL_ENOTSUP;
    }
    dim3 grid(ngroups * ncombo), block(512);
    if (level == 1) hipLaunchKernelGGL(conv1_ft_wino4_kernel<1>, grid, block, 0, st, vg, h2all, h2_step_stride, params, steps, off, g, ngroups, ncombo, raw_both);
    else hipLaunchKernelGGL(conv1_ft_wino4_kernel<2>, grid, block, 0, st, vg, h2all, h2_step_stride, params, steps, off, g, ngroups, ncombo, raw_both);
    IFL_LAUNCH_CHECK();
    return 0;
}

#ifdef IFL_EXP_TIME
extern "C" int ifl_dbgx_read(void* host, size_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ifl_dbgx), bytes); }
#endif
