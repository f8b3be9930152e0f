// conv1.hip -- first two layers of the coupling networks as one implicit-GEMM kernel on the fp32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, for one conditional FlowStep (reference interflow/models/modules):
//   flow.Conv2d 3x3 (C_in -> 64, no bias) + ActNorm + ReLU          flow.py:25-64, FlowAffineCouplingsAblation.py:249
//   flow.Conv2d 1x1 (64 -> 64)            + ActNorm + ReLU          FlowAffineCouplingsAblation.py:251-253
// for both coupling networks: fFeatures(ft) and fAffine(cat[z1, ft]) (FlowAffineCouplingsAblation.py:147-157,
// 219-233).  The output is the 64-channel input of each network's final Conv2dZeros, written to the
// ringed channels-last hidden buffer h2[..][0:64] (fAffine) / h2[..][64:128] (fFeatures).
//
// GEMM orientation: D[out channel][pixel] += W[out channel][k] * X[k][pixel], k = (tap, in channel).
//   A operand = weights, pre-packed on the host in MFMA fragment order, streamed global -> VGPR (each wave
//               owns one network's 64 output channels, so weights are not shared between waves: no LDS).
//   B operand = activations: the tile's (rows+2) x (bw+2) pixel patch of a 32-channel chunk is staged once in
//               LDS and reused by the 9 taps and by every wave of the workgroup.
// The 1x1 conv consumes the accumulators directly as its B operand (the 32x32 C/D layout has the pixel on the
// lane and the channel in the register index; k may be summed in any order, so the 1x1 weights are packed to
// follow the accumulator's channel order) -- no LDS transpose between the two GEMMs.
#include "ifl_common.h"

// Two kernels (DESIGN.md "conv1"):
//   conv1_ft_kernel      the z-independent bulk: the 9 x 320-channel conditioning part of BOTH networks' first conv for
//                        EVERY conditional step of a level in one launch (grid.z = step).  fFeatures is finished here
//                        (ActNorm, ReLU, 1x1, ActNorm, ReLU -> h2[..][64:128]); fAffine's 64 raw pre-activation sums go
//                        to h2[..][0:64].  ~98 % of the path's FLOPs, no dependence on the running z, so one launch per
//                        level fills the chip without a partly filled last round per step.
//   conv1_finish_kernel  (BF16 = true: the same GEMMs with operands rounded to bf16 on v_mfma_f32_32x32x8_bf16, fp32 accumulate,
//                        ActNorm / ReLU in fp32: BASELINE.json configs[4])
//                        per step, in the sequential chain: fAffine's remaining K = 9 x C/2 taps over z1 are added to
//                        the stored sums, then its ActNorm, ReLU, 1x1, ActNorm, ReLU -> ha (the slab keeps the raw sums,
//                        so a prepared conditioning serves any number of passes: ifl_cond_prepare).

namespace {

constexpr int NCC = IFL_FT / IFL_CK;          // 10 chunks of 32 conditioning channels
constexpr int LD_SLOTS = (IFL_PP_MAX * 8 + 255) / 256;   // float4 patch loads per thread per chunk

#include "conv1_fin.h"

// this lane's pixels of the tile: LDS byte address of the top-left tap (+16*h) and ring pixel index (or -1)
template <int NT>
__device__ __forceinline__ void lane_pixels(const IflTile& t, const IflGeom& g, int b, int first, int j, int h, int cs,
                                            int (&a0)[NT], int64_t (&opix)[NT])
{
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int idx = t.s + first + n * 32 + j;
        bool valid = idx < t.npx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        a0[n] = (((ty - t.y_first - 1) * t.PW + tx) * cs + 4 * h) * 4;
        opix[n] = valid ? ((int64_t)(b * (g.H + 2) + ty + 1) * (g.W + 2) + t.x0 + tx + 1) : -1;
    }
}

// ---- phase 1: conditioning part of both networks, all steps of a level --------------------------------------------
// 4 waves = 2 networks x 2 pixel halves of the 128-pixel tile; each wave owns 64 output channels x 64 pixels.
// LVL (1 or 2) only names the instantiation, so that a profile lists the two flow levels separately.
template <int LVL>
__global__ __launch_bounds__(256, 2)
void conv1_ft_kernel(const float* __restrict__ ftP, float* __restrict__ h2all, int64_t h2_step_stride,
                     const float* __restrict__ params, IflStepList steps, IflStepOff off, IflGeom g)
{
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) float lds[LD_SLOTS * 32 * IFL_CS];   // >= IFL_PP_MAX pixels: staging stores need no bound test

    // Workgroup -> (tile, image, step) = (x, y, z), step slowest: at any moment the chip works on one or two steps, whose
    // weights (1.5 MB per step, streamed by every workgroup) stay in the XCDs' L2s.  Grouping the steps of one tile on
    // one XCD instead (so that they share the conditioning patch in L2) was measured 10 % slower: four steps' weights
    // no longer fit beside the patches (profiles/r01_ab_schedule.txt).
    const int tile = blockIdx.x, step = blockIdx.z;
    const IflTile t = ifl_tile(g, tile);
    if (!t.active) return;                    // uniform for the whole workgroup
    const int b = blockIdx.y;
    const float* prm = params + steps.off[step];
    float* h2P = h2all + (int64_t)step * h2_step_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int net = wave & 1, part = wave >> 1;
    const int W2 = g.W + 2;

    int a0[NT];
    int64_t opix[NT];
    lane_pixels<NT>(t, g, b, part * 64, j, h, IFL_CS, a0, opix);

    // ---- patch staging: per-thread source offsets are the same for every chunk -----------------
    // Slots past the patch read the plane's first float4 (valid memory) and land in LDS rows nobody reads, so the
    // loads and stores below carry no branches.
    const float* ft_b = ftP + (int64_t)b * (g.H + 2) * W2 * IFL_FT;
    int srcoff[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) {
        int idx = tid + i * 256;
        int p = idx >> 3, q = idx & 7;
        int r = p / t.PW, c = p - r * t.PW;
        srcoff[i] = (p < t.PP) ? (((t.y_first + 1 + r) * W2 + t.x0 + c) * IFL_FT + q * 4) : 0;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // Weights: wave-uniform base + per-lane offset.  They stream through a register ring WD k-steps deep, so each global
    // load has WD * 16 MFMAs of cover (the blocks of consecutive k-steps are contiguous in memory, also across chunk
    // boundaries; the last prefetches read a little past the field, inside the step's block).
    constexpr int WD = 3;
    constexpr int ITERS = 36;                 // 9 taps x 4 k-steps of 8 channels per chunk
    static_assert(ITERS % WD == 0, "ring slots must be compile-time");
    const float* wnet = prm + off.w1[net];
    const unsigned lane4 = lane * 4;
    f32x4 wq[WD][2];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) wq[d][m] = *(const f32x4*)(wnet + (d * 2 + m) * 256 + lane4);

    f32x4 stage[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)(ft_b + srcoff[i]);

    for (int cc = 0; cc < NCC; ++cc) {
        __syncthreads();                      // everyone finished reading the previous chunk
#pragma unroll
        for (int i = 0; i < LD_SLOTS; ++i) {
            int idx = tid + i * 256;
            *(f32x4*)&lds[(idx >> 3) * IFL_CS + (idx & 7) * 4] = stage[i];
        }
        __syncthreads();
        if (cc + 1 < NCC) {                   // next chunk's patch travels while this one is multiplied
#pragma unroll
            for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)(ft_b + srcoff[i] + (cc + 1) * IFL_CK);
        }
        const float* wc = wnet + (int64_t)cc * (ITERS * 2 * 256);
        // Software pipeline, pinned with sched_barrier so hipcc keeps it: while k-step `it` multiplies, the LDS
        // reads of k-step it+1 and the weight loads of k-step it+WD are in flight.
        f32x4 xb[2][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) xb[0][n] = *(const f32x4*)((const char*)lds + a0[n]);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (it + 1 < ITERS) {
                const int tap = (it + 1) >> 2, kq = (it + 1) & 3;
                const int tapoff = ((tap / 3) * t.PW + (tap % 3)) * (IFL_CS * 4);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    xb[(it + 1) & 1][n] = *(const f32x4*)((const char*)lds + a0[n] + tapoff + kq * 32);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = mfma32(wq[it % WD][m][e], xb[it & 1][n][e], acc[m][n]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
                wq[it % WD][m] = *(const f32x4*)(wc + ((it + WD) * 2 + m) * 256 + lane4);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (net == 0) {
        // fAffine: raw sums, finished by conv1_finish_kernel once z1 of this step exists
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (opix[n] < 0) continue;
            float* dst = h2P + opix[n] * IFL_H2 + 4 * h;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(f32x4*)(dst + m * 32 + gq * 8) = f32x4{acc[m][n][gq * 4], acc[m][n][gq * 4 + 1],
                                                             acc[m][n][gq * 4 + 2], acc[m][n][gq * 4 + 3]};
        }
    } else {
        finish_and_store<NT, IFL_H2>(acc, prm + off.an1[1], prm + off.w2[1], prm + off.an2[1], h2P + IFL_HID, opix, lane, h);
    }
}


// ---- phase 1, Winograd F(2x2, 3x3) form -----------------------------------------------------------------------------------
// The same sums as conv1_ft_kernel with 2.25x fewer multiplications (Lavin & Gray's minimal filtering, fp32 throughout):
//   Y = A^T [ sum_ci U[co][ci] (.) V[ci] ] A,   U = G g G^T (host, float64 -> fp32, pack.py),   V = B^T d B
// for every 2x2 output tile and its 4x4 input window d.  For each of the 16 positions f = (i, jc) of the transformed
// domain the sum over input channels is a GEMM  M_f[co][tile] = U_f[co][ci] V_f[ci][tile]  on the fp32 matrix cores.
//
// Workgroup = 32 consecutive tiles (raster order over image, tile row, tile column) = 128 output pixels, 8 waves =
// 2 networks x 4 columns jc of the transformed domain; a wave accumulates its 4 positions (i = 0..3, jc) x 64 output
// channels x 32 tiles = 128 accumulator registers.  Per 32-channel chunk:
//   * every thread loads the 3 x 4 window pixels it needs for (tile, channel quad, upper/lower half of i) straight from
//     global memory (the neighbours' windows overlap; the overlap is served by the L1), transforms them in registers and
//     writes V[f][tile][ch] to LDS (double-buffered: the loads and the transform of chunk c+1 overlap the GEMM of chunk c,
//     one barrier per chunk);
//   * the GEMM reads V as the B operand (one 16-byte LDS read per 8 MFMAs); weights stream global -> VGPR through a ring.
// Epilogue: A^T along i is local to a wave; A along jc goes through LDS between the four waves of a network, after which
// wave jc holds the pixels (dy, dx) = (jc >> 1, jc & 1) of the 32 tiles in the usual accumulator layout, so the ActNorm /
// ReLU / 1x1 / store code of the direct kernel is shared.
// Build with -DIFL_EXP_TIME to record per-workgroup timestamps (tools/exp_wg_timeline.py); never defined in the product build.
#ifdef IFL_EXP_TIME
__device__ unsigned long long ifl_dbg[8 * 65536];
#define DBG_T(k) do { if (LVL == 1 && lane == 0 && (wave == 0 || wave == 4)) { \
    unsigned long long t_ = wall_clock64(); \
    int wgid_ = blockIdx.z * gridDim.x + blockIdx.x; \
    if (wgid_ < 16384) ifl_dbg[((wgid_ * 2 + (wave >> 2)) * 8) + (k)] = t_; } } while (0)
#else
#define DBG_T(k) do {} while (0)
#endif
constexpr int WV_TS = 36;                     // words per (position, tile) row of V: 32 channels + 4 (conflict-free b128 reads)
constexpr int WV_BUF = 16 * 32 * WV_TS;       // floats per V buffer (73,728 B)

template <int LVL>
__global__ __launch_bounds__(512, 1)
void conv1_ft_wino_kernel(const float* __restrict__ ftP, float* __restrict__ h2all, int64_t h2_step_stride,
                          const float* __restrict__ params, IflStepList steps, IflStepOff off, IflWGeom g)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * WV_BUF];
    const int step = blockIdx.z;
    const float* prm = params + steps.off[step];
    float* h2P = h2all + (int64_t)step * h2_step_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    // waves w and w + 4 share a SIMD: one wave of each network per SIMD, so that fFeatures' extra epilogue (ActNorm, 1x1
    // conv) is spread over all four matrix cores
    const int net = wave >> 2, jc = wave & 3;
    const int W2 = g.W + 2;
    const int tpi = g.TH * g.TW;              // tiles per image
    // Workgroups go to the 8 XCDs round-robin (linear id % 8).  Give each XCD a contiguous eighth of the tile groups, so
    // that the window rows a group shares with the groups above and below it are found in that XCD's L2.
    int grp = blockIdx.x;
    if ((gridDim.x & 7) == 0) grp = (grp & 7) * (gridDim.x >> 3) + (grp >> 3);
    DBG_T(0);
#ifdef IFL_EXP_TIME
    if (LVL == 1 && lane == 0 && wave == 0) {   // HW_REG_XCC_ID, HW_REG_HW_ID: which CU this workgroup runs on
        int wgid_ = blockIdx.z * gridDim.x + blockIdx.x;
        if (wgid_ < 16384) ifl_dbg[(wgid_ * 2) * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif

    // ---- transform role: (tile, channel quad, half) ---------------------------------------------------------------------
    // B^T d along the window's rows, for the rows i = 2*half, 2*half + 1 of the transformed domain:
    //   half 0:  R0 = d0 - d2,  R1 = d1 + d2          half 1:  R2 = d2 - d1,  R3 = d1 - d3
    // i.e. first = A - B, second = (half ? B - C : C + B) with window rows (A, B, C) = (0, 2, 1) / (2, 1, 3): rows A and B are
    // loaded first, row C later into A's registers (32 staging registers instead of 48).
    const int tt = (tid & 255) >> 3, tq = tid & 7, half = tid >> 8;
    int offA, offB, offC;                     // float offsets of the three window rows' first pixel, channel quad tq
    int cstep3;                               // step to the window's last column, clamped at the ring's edge (odd plane width)
    {
        int gt = grp * 32 + tt;
        if (gt >= g.ntiles) gt = g.ntiles - 1;
        int b = gt / tpi, rem = gt - b * tpi;
        int tr = rem / g.TW, tc = rem - tr * g.TW;
        int x0 = 2 * tc;                      // ring coordinates
        int yA = 2 * tr + (half ? 2 : 0), yB = 2 * tr + (half ? 1 : 2), yC = 2 * tr + (half ? 3 : 1);
        if (yA > g.H + 1) yA = g.H + 1;       // odd plane height: the clamped rows only reach outputs that are not stored
        if (yB > g.H + 1) yB = g.H + 1;
        if (yC > g.H + 1) yC = g.H + 1;
        cstep3 = (x0 + 3 > g.W + 1 ? 2 : 3) * IFL_FT;
        const int img = b * (g.H + 2);
        offA = ((img + yA) * W2 + x0) * IFL_FT + tq * 4;
        offB = ((img + yB) * W2 + x0) * IFL_FT + tq * 4;
        offC = ((img + yC) * W2 + x0) * IFL_FT + tq * 4;
    }
    // ---- output role: wave jc owns pixel (dy, dx) of tile j ---------------------------------------------------------------
    int64_t opix[1];
    {
        int gt = grp * 32 + j;
        bool valid = gt < g.ntiles;
        if (!valid) gt = g.ntiles - 1;
        int b = gt / tpi, rem = gt - b * tpi;
        int tr = rem / g.TW, tc = rem - tr * g.TW;
        int y = 2 * tr + (jc >> 1), x = 2 * tc + (jc & 1);
        valid = valid && y < g.H && x < g.W;
        opix[0] = valid ? ((int64_t)(b * (g.H + 2) + y + 1) * W2 + x + 1) : -1;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;

    constexpr int WD = 4;
    constexpr int ITERS = 16;                 // 4 k-steps of 8 channels x 4 positions i per chunk
    static_assert(ITERS % WD == 0, "ring slots must be compile-time");
    const float* wnet = prm + off.w1w[net] + (int64_t)jc * (NCC * ITERS * 2 * 256) + lane * 4;
    f32x4 wq[WD][2];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) wq[d][m] = *(const f32x4*)(wnet + (d * 2 + m) * 256);

    const int vwr = ((half * 8) * 32 + tt) * WV_TS + tq * 4;          // V[4*(2*half) + 0][tt][tq*4]
    const int vrd = (jc * 32 + j) * WV_TS + h * 4;                    // V[jc][j][4h]

    // one row R of B^T d -> the four columns of R B, written to V[4i + 0..3][tile][quad]
    auto put_row = [&](float* dst, const f32x4 (&R)[4]) {
        *(f32x4*)(dst) = R[0] - R[2];
        *(f32x4*)(dst + 32 * WV_TS) = R[1] + R[2];
        *(f32x4*)(dst + 2 * 32 * WV_TS) = R[2] - R[1];
        *(f32x4*)(dst + 3 * 32 * WV_TS) = R[1] - R[3];
    };

    f32x4 wa[4], wb[4];                       // window rows A (later C) and B of the chunk being transformed
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        wa[c] = *(const f32x4*)(ftP + offA + (c < 3 ? c * IFL_FT : cstep3));
        wb[c] = *(const f32x4*)(ftP + offB + (c < 3 ? c * IFL_FT : cstep3));
    }
    {                                         // chunk 0 is transformed up front; later chunks inside the GEMM loop
        f32x4 R[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) R[c] = wa[c] - wb[c];
        put_row(lds + vwr, R);
#pragma unroll
        for (int c = 0; c < 4; ++c) wa[c] = *(const f32x4*)(ftP + offC + (c < 3 ? c * IFL_FT : cstep3));
#pragma unroll
        for (int c = 0; c < 4; ++c) R[c] = wb[c] + (half ? -wa[c] : wa[c]);
        put_row(lds + vwr + 4 * 32 * WV_TS, R);
    }

    DBG_T(1);
    for (int cc = 0; cc < NCC; ++cc) {
        const float* vr = lds + (cc & 1) * WV_BUF + vrd;
        float* vnext = lds + ((cc + 1) & 1) * WV_BUF + vwr;
        const bool more = cc + 1 < NCC;
        const int coff = (cc + 1) * IFL_CK;
        __syncthreads();                      // V[cc & 1] complete; everyone is past the GEMM of chunk cc - 1, so
                                              // V[(cc + 1) & 1] may be overwritten while this chunk is multiplied
        const float* wc = wnet + (int64_t)cc * (ITERS * 2 * 256);
        f32x4 xb[2];
        xb[0] = *(const f32x4*)(vr);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {  // it = kq * 4 + i: consecutive k-steps go to different accumulators
            if (it + 1 < ITERS) {
                const int kq = (it + 1) >> 2, i = (it + 1) & 3;
                xb[(it + 1) & 1] = *(const f32x4*)(vr + (i * 4) * 32 * WV_TS + kq * 8);
            }
            // the next chunk's window: rows A, B requested at it 0, used at it 7, where row C is requested, used at it 14
            if (it == 0 && more) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    wa[c] = *(const f32x4*)(ftP + offA + (c < 3 ? c * IFL_FT : cstep3) + coff);
                    wb[c] = *(const f32x4*)(ftP + offB + (c < 3 ? c * IFL_FT : cstep3) + coff);
                }
            }
            if (it == 7 && more) {
                f32x4 R[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) R[c] = wa[c] - wb[c];
                put_row(vnext, R);
#pragma unroll
                for (int c = 0; c < 4; ++c) wa[c] = *(const f32x4*)(ftP + offC + (c < 3 ? c * IFL_FT : cstep3) + coff);
            }
            if (it == 14 && more) {
                f32x4 R[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) R[c] = wb[c] + (half ? -wa[c] : wa[c]);
                put_row(vnext + 4 * 32 * WV_TS, R);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[it & 3][m] = mfma32(wq[it % WD][m][e], xb[it & 1][e], acc[it & 3][m]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
                wq[it % WD][m] = *(const f32x4*)(wc + ((it + WD) * 2 + m) * 256);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- A^T along i (local), A along jc through LDS: X[net][jc][dy][m][r / 4][lane][r % 4], 128 KB ----------------------
    f32x16 y[2][1];
    DBG_T(2);
    __syncthreads();                          // every wave's GEMM reads of V are over
    DBG_T(3);
    {
        float* xw = lds + (net * 4 + jc) * (2 * 2 * 4 * 256) + lane * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 t0, t1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = gq * 4 + e;
                    t0[e] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
                    t1[e] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
                }
                *(f32x4*)(xw + ((0 * 2 + m) * 4 + gq) * 256) = t0;
                *(f32x4*)(xw + ((1 * 2 + m) * 4 + gq) * 256) = t1;
            }
    }
    __syncthreads();
    {
        // pixel (dy, dx) of the tile: dx = 0: (x0 + x1) + x2;  dx = 1: (x1 - x2) - x3  ==  (p + s q) + s r, s = +-1 (exact)
        const int dy = jc >> 1, dx = jc & 1;
        const float sg = dx ? -1.f : 1.f;
        const float* xr = lds + (net * 4 + dx) * (2 * 2 * 4 * 256) + dy * (2 * 4 * 256) + lane * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float* xp = xr + (m * 4 + gq) * 256;
                f32x4 p = *(const f32x4*)(xp), q = *(const f32x4*)(xp + 2 * 2 * 4 * 256), r = *(const f32x4*)(xp + 2 * 2 * 2 * 4 * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[m][0][gq * 4 + e] = (p[e] + sg * q[e]) + sg * r[e];
            }
    }

    DBG_T(4);
    if (net == 0) {
        if (opix[0] >= 0) {
            float* dst = h2P + opix[0] * IFL_H2 + 4 * h;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(f32x4*)(dst + m * 32 + gq * 8) = f32x4{y[m][0][gq * 4], y[m][0][gq * 4 + 1],
                                                             y[m][0][gq * 4 + 2], y[m][0][gq * 4 + 3]};
        }
    } else {
        finish_and_store<1, IFL_H2>(y, prm + off.an1[1], prm + off.w2[1], prm + off.an2[1], h2P + IFL_HID, opix, lane, h);
    }
    DBG_T(5);
}

// ---- bf16 variant of phase 1 (BASELINE.json configs[4]: bf16 coupling-network convs, fp32 ActNorm / log-det) ------------
// Same tiling and data flow as conv1_ft_kernel; the two GEMMs run on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//   * the conditioning patch is converted fp32 -> bf16 (round to nearest even, v_cvt_pk_bf16_f32) while it is staged
//     into LDS: 32 channels = 64 B per pixel, pixel stride 80 B (16-byte reads of consecutive pixels hit disjoint banks);
//   * weights are bf16 copies in fragment order (lane (i, h) holds W[out i][k = 8h .. 8h+7] of a 16-channel k-step);
//   * accumulation, ActNorm, ReLU and the stores stay fp32; fFeatures' 1x1 conv takes the fp32 accumulators, rounded to
//     bf16, as its B operand in registers (the k order inside a step is that of the accumulator rows, and the 1x1 weights
//     are packed to match).
// The result differs from the fp32 path by bf16 rounding of the operands (tolerance stated in tests/test_gpu_parity.py);
// encode and decode still invert each other to fp32 accuracy, because both directions compute the same network outputs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int CSB = 80;                       // LDS pixel stride in bytes of a 32-channel bf16 chunk

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fFeatures' epilogue of the bf16 kernels: ActNorm + ReLU (fp32), 1x1 conv on the bf16 matrix cores straight from the
// accumulators (rounded to bf16; the 1x1 weights are packed in the accumulator rows' k order), ActNorm + ReLU, store
template <int NT>
__device__ __forceinline__ void bf16_finish_and_store(f32x16 (&acc)[2][NT], const float* __restrict__ an1, const bf16x8* w2,
                                                      const float* __restrict__ an2, float* __restrict__ h2P,
                                                      const int64_t (&opix)[NT], int h)
{
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x4 bb[4], ss[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bb[q] = *(const f32x4*)(an1 + (m * 2 + h) * 16 + q * 4);
            ss[q] = *(const f32x4*)(an1 + 64 + (m * 2 + h) * 16 + q * 4);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[m][n][r] + bb[r >> 2][r & 3]) * ss[r >> 2][r & 3];
                acc[m][n][r] = v > 0.f ? v : 0.f;
            }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x16 y[2];
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[m2][r] = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 xk;
#pragma unroll
                for (int e = 0; e < 8; ++e) xk[e] = (__bf16)acc[m][n][8 * s2 + e];
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) y[m2] = mfma_bf16(w2[((m2 * 2 + m) * 2 + s2) * 64], xk, y[m2]);
            }
        if (opix[n] >= 0) {
            float* dst = h2P + IFL_HID + opix[n] * IFL_H2 + 4 * h;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 bb = *(const f32x4*)(an2 + (m2 * 2 + h) * 16 + gq * 4);
                    f32x4 ss = *(const f32x4*)(an2 + 64 + (m2 * 2 + h) * 16 + gq * 4);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = (y[m2][gq * 4 + e] + bb[e]) * ss[e];
                        o[e] = v > 0.f ? v : 0.f;
                    }
                    *(f32x4*)(dst + m2 * 32 + gq * 8) = o;
                }
        }
    }
}

template <int LVL>
__global__ __launch_bounds__(256, 2)
void conv1_ft_bf16_kernel(const float* __restrict__ ftP, float* __restrict__ h2all, int64_t h2_step_stride,
                          const float* __restrict__ params, IflStepList steps, IflStepOff off, IflGeom g)
{
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LD_SLOTS * 32 * CSB];
    const int tile = blockIdx.x, step = blockIdx.z;
    const IflTile t = ifl_tile(g, tile);
    if (!t.active) return;
    const int b = blockIdx.y;
    const float* prm = params + steps.off[step];
    float* h2P = h2all + (int64_t)step * h2_step_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int net = wave & 1, part = wave >> 1;
    const int W2 = g.W + 2;

    int a0[NT];
    int64_t opix[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int idx = t.s + part * 64 + n * 32 + j;
        bool valid = idx < t.npx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        a0[n] = ((ty - t.y_first - 1) * t.PW + tx) * CSB + 16 * h;
        opix[n] = valid ? ((int64_t)(b * (g.H + 2) + ty + 1) * (g.W + 2) + t.x0 + tx + 1) : -1;
    }

    const float* ft_b = ftP + (int64_t)b * (g.H + 2) * W2 * IFL_FT;
    int srcoff[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) {
        int idx = tid + i * 256;
        int p = idx >> 3, q = idx & 7;
        int r = p / t.PW, c = p - r * t.PW;
        srcoff[i] = (p < t.PP) ? (((t.y_first + 1 + r) * W2 + t.x0 + c) * IFL_FT + q * 4) : 0;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int WD = 6;
    constexpr int ITERS = 18;                 // 9 taps x 2 k-steps of 16 channels per chunk
    static_assert(ITERS % WD == 0, "ring slots must be compile-time");
    const bf16x8* wnet = (const bf16x8*)(prm + off.w1b[net]) + lane;          // block bi at wnet[bi * 64]
    bf16x8 wq[WD][2];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) wq[d][m] = wnet[(d * 2 + m) * 64];

    f32x4 stage[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)(ft_b + srcoff[i]);

    for (int cc = 0; cc < NCC; ++cc) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LD_SLOTS; ++i) {
            int idx = tid + i * 256;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)stage[i][e];
            *(bf16x4*)&lds[(idx >> 3) * CSB + (idx & 7) * 8] = v;
        }
        __syncthreads();
        if (cc + 1 < NCC) {
#pragma unroll
            for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)(ft_b + srcoff[i] + (cc + 1) * IFL_CK);
        }
        const bf16x8* wc = wnet + (int64_t)cc * (ITERS * 2 * 64);
        bf16x8 xb[2][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) xb[0][n] = *(const bf16x8*)(lds + a0[n]);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (it + 1 < ITERS) {
                const int tap = (it + 1) >> 1, ks = (it + 1) & 1;
                const int tapoff = ((tap / 3) * t.PW + (tap % 3)) * CSB;
#pragma unroll
                for (int n = 0; n < NT; ++n) xb[(it + 1) & 1][n] = *(const bf16x8*)(lds + a0[n] + tapoff + ks * 32);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mfma_bf16(wq[it % WD][m], xb[it & 1][n], acc[m][n]);
#pragma unroll
            for (int m = 0; m < 2; ++m) wq[it % WD][m] = wc[((it + WD) * 2 + m) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (net == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (opix[n] < 0) continue;
            float* dst = h2P + opix[n] * IFL_H2 + 4 * h;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(f32x4*)(dst + m * 32 + gq * 8) = f32x4{acc[m][n][gq * 4], acc[m][n][gq * 4 + 1],
                                                             acc[m][n][gq * 4 + 2], acc[m][n][gq * 4 + 3]};
        }
        return;
    }
    bf16_finish_and_store<NT>(acc, prm + off.an1[1], (const bf16x8*)(prm + off.w2b) + lane, prm + off.an2[1], h2P, opix, h);
}

// ---- bf16 phase 1, region form (default for the bf16 variant) --------------------------------------------------------------
// conv1_ft_bf16_kernel streams 2 KB of weight fragments per wave for 4 MFMAs (64 B per clock and CU at the matrix cores'
// rate: twice what a CU ingests), which holds it at 0.30 of the bf16 peak.  Here a wave keeps 64 outputs x 128 pixels
// (8 accumulators of 32x32), so a weight fragment serves 4 MFMAs and the stream halves; to make 128 pixels per wave
// affordable the workgroup's pixels are a 16 x 16 REGION (patch 18 x 18 = 324 pixels, about the band form's 304 for half
// as many pixels): 4 waves = 2 networks x 2 halves (8 rows) of the region, MFMA pixel tile n of a wave = 2 rows x 16
// (an 8 x 32 region with one-row tiles -- conflict-free LDS reads by construction -- measures the same at 512 x 512 and wastes
// more lanes on the 80- and 40-pixel planes of 160 x 160 patches).
// The patch is loaded, rounded to bf16 and stored to LDS between the two barriers of a chunk (no register prefetch: the
// accumulators take half the register file); the CU's second workgroup multiplies meanwhile.
constexpr int RBW = 16, RBH = 16;             // region width, height
constexpr int RBP = RBW + 2;                  // patch width (height RBH + 2)
constexpr int RB_PX = RBP * (RBH + 2);        // patch pixels (324)
constexpr int RB_SLOTS = (RB_PX * 8 + 255) / 256;          // float4 loads per thread per chunk (11)
constexpr int RB_SG = 4;                      // steps whose workgroups of one region run side by side on an XCD

template <int LVL>
__global__ __launch_bounds__(256, 2)
void conv1_ft_bf16r_kernel(const float* __restrict__ ftP, float* __restrict__ h2all, int64_t h2_step_stride,
                           const float* __restrict__ params, IflStepList steps, IflStepOff off, int H, int W, int nimg, int nsteps)
{
    constexpr int NT = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RB_SLOTS * 32 * CSB];
    // Workgroup -> (step, image, region).  Consecutive workgroup ids go to consecutive XCDs; within an XCD the steps of one
    // launch are the fastest index, so the workgroups that stage the SAME patch (one per step) run side by side on one XCD and
    // all but the first find it in that XCD's L2 (otherwise every step re-reads the conditioning features from HBM: at 512 x 512
    // 2.2 TB/s of them); the launch's weights (4 steps x 2 networks x 368 KB) stay in the 4 MB L2 as well.
    const int gx = (W + RBW - 1) / RBW;
    const int nreg = gx * ((H + RBH - 1) / RBH);
    // (groups of RB_SG steps, one group after the other, when a launch holds more)
    const int per_group = (nreg * nimg + 7) / 8 * 8 * RB_SG;
    const int grp_ = blockIdx.x / per_group, loc_ = blockIdx.x - grp_ * per_group, t_ = loc_ >> 3;
    const int step = grp_ * RB_SG + t_ % RB_SG, rb = (t_ / RB_SG) * 8 + (loc_ & 7);
    if (rb >= nreg * nimg || step >= nsteps) return;
    const int b = rb / nreg, reg = rb - b * nreg;
    const int ry0 = (reg / gx) * RBH, rx0 = (reg % gx) * RBW;                      // the region's pixel (0, 0), plane coordinates
    const float* prm = params + steps.off[step];
    float* h2P = h2all + (int64_t)step * h2_step_stride;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int net = wave & 1, part = wave >> 1;
    const int W2 = W + 2;

    // tile n of this wave: pixels (part * 8 + 2 n + (j >> 4), j & 15) of the region; patch pixel (r, c) is pixel (r, c)'s top-left tap
    const int a00 = ((part * 8 + (j >> 4)) * RBP + (j & 15)) * CSB + 16 * h;      // tile n: + n * 2 * RBP * CSB

    // patch pixel (r, c) = ring pixel (ry0 + r, rx0 + c); outside the ringed plane (partial regions): pixel (0, 0) of the ring, a zero
    // (buffer loads: descriptor + 32-bit lane offset + scalar offset, no 64-bit address arithmetic or address registers)
    const float* ft_b = ftP + (int64_t)b * (H + 2) * W2 * IFL_FT;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)ft_b, 0, (H + 2) * W2 * IFL_FT * 4, 0x00020000);
    int srcoff[RB_SLOTS];                     // bytes
#pragma unroll
    for (int i = 0; i < RB_SLOTS; ++i) {
        const int idx = tid + i * 256;
        const int p = idx >> 3, q = idx & 7;
        const int r = p / RBP, c = p - r * RBP;
        const bool in = p < RB_PX && ry0 + r <= H + 1 && rx0 + c <= W + 1;
        srcoff[i] = (in ? ((ry0 + r) * W2 + rx0 + c) * IFL_FT + q * 4 : q * 4) * 4;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int WD = 3;
    constexpr int ITERS = 18;                 // 9 taps x 2 k-steps of 16 channels per chunk
    static_assert(ITERS % WD == 0, "ring slots must be compile-time");
    // weight block bi (one MFMA A fragment for every lane) at byte bi * 1024 of the network's bf16 copy; reads past its end
    // (the ring's last refills) return zeros
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(prm + off.w1b[net]), 0, NCC * ITERS * 2 * 1024, 0x00020000);
    auto load_w = [&](int bi) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, bi * 1024, 0)); };
    bf16x8 wq[WD][2];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) wq[d][m] = load_w(d * 2 + m);

    auto load_patch = [&](f32x4 (&stage)[RB_SLOTS], int cc) {
#pragma unroll
        for (int i = 0; i < RB_SLOTS; ++i) {
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, srcoff[i], cc * (IFL_CK * 4), 0));
        }
    };
    f32x4 stage[RB_SLOTS];
    load_patch(stage, 0);
    for (int cc = 0; cc < NCC; ++cc) {
        __syncthreads();                      // the previous chunk's LDS reads are over
#pragma unroll
        for (int i = 0; i < RB_SLOTS; ++i) {
            const int idx = tid + i * 256;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)stage[i][e];
            if (idx < RB_PX * 8) *(bf16x4*)&lds[(idx >> 3) * CSB + (idx & 7) * 8] = v;
        }
        __syncthreads();
        if (cc + 1 < NCC) load_patch(stage, cc + 1);        // in flight beside this chunk's MFMAs
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int tap = it >> 1, ks = it & 1;
            const int tapoff = ((tap / 3) * RBP + (tap % 3)) * CSB;
            bf16x8 xb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) xb[n] = *(const bf16x8*)(lds + a00 + n * (2 * RBP * CSB) + tapoff + ks * 32);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m][n] = mfma_bf16(wq[it % WD][m], xb[n], acc[m][n]);
#pragma unroll
            for (int m = 0; m < 2; ++m) wq[it % WD][m] = load_w((cc * ITERS + it + WD) * 2 + m);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    int64_t opix[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int r = part * 8 + n * 2 + (j >> 4), c = j & 15;
        opix[n] = (ry0 + r < H && rx0 + c < W) ? ((int64_t)(b * (H + 2) + ry0 + r + 1) * W2 + rx0 + c + 1) : -1;
    }
    if (net == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (opix[n] < 0) continue;
            float* dst = h2P + opix[n] * IFL_H2 + 4 * h;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *(f32x4*)(dst + m * 32 + gq * 8) = f32x4{acc[m][n][gq * 4], acc[m][n][gq * 4 + 1],
                                                             acc[m][n][gq * 4 + 2], acc[m][n][gq * 4 + 3]};
        }
        return;
    }
    bf16_finish_and_store<NT>(acc, prm + off.an1[1], (const bf16x8*)(prm + off.w2b) + lane, prm + off.an2[1], h2P, opix, h);
}

// ---- phase 2: fAffine's z1 taps + its epilogue --------------------------------------------------------------------
// 4 waves x 32 pixels; each wave owns all 64 channels of its pixels.  K = 9 taps x KZ channels (C/2 padded to a multiple
// of 8 with zero weights).
// PRE = 1 (encode chain): the step's ActNorm2d + InvertibleConv1x1 (pointwise.hip: anic, y = W ((x + b) .* exp(logs)),
// FlowActNorms.py:87-92, Permutations.py:114-118) is applied here, on the way into LDS, instead of by a kernel of its own
// before this one: every patch pixel's z is read from `zP` (the running latent BEFORE the map), mapped, its first KZ channels
// staged for the taps (pixels outside the plane stay zero: the convolution pads the MAPPED z1), and the tile's own pixels are
// written, all C channels, to `zoutP` -- a second buffer, because the neighbouring workgroups still read the unmapped halo.
// PRE = 2 (decode chain, C = 48): the PREVIOUS step's inverse pair x = (Winv y) .* exp(-logs) - b followed by this step's
// z = (z + shiftFt) * scaleFt from `sfP` -- what the previous step's last kernel (anic48 with sf_next) would have done.
// One launch and one z round trip less per conditional step of the sequential chain.  C = 12: a thread per patch pixel, W as
// LDS broadcasts; C = 48: v_mfma_f32_16x16x4_f32 with anic48_mfma_kernel's operand layout (a wave takes 64 patch pixels).
template <int KZ, bool BF16, int PRE = 0>
__global__ __launch_bounds__(256, 2)
void conv1_finish_kernel(const float* __restrict__ zP, const float* __restrict__ h2P, float* __restrict__ haP,
                         const float* __restrict__ prm, IflStepOff off, IflGeom g, int C,
                         float* __restrict__ zoutP = nullptr, const float* __restrict__ anicP = nullptr,
                         const float* __restrict__ sfP = nullptr)
{
    constexpr int NT = 1;
    __shared__ __attribute__((aligned(16))) float lds[IFL_PP_MAX * (KZ + 4)];     // KZ = 8: 14.6 KB, KZ = 24: 34 KB
    const IflTile t = ifl_tile(g, blockIdx.x);
    if (!t.active) return;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int W2 = g.W + 2;
    constexpr int CSZ = KZ + 4;               // 12 or 28 words: CSZ/4 odd => conflict-free b128 reads
    constexpr int nq = KZ / 4;

    int a0[NT];
    int64_t opix[NT];
    lane_pixels<NT>(t, g, b, wave * 32, j, h, CSZ, a0, opix);

    const float* z_b = zP + (int64_t)b * (g.H + 2) * W2 * C;
    if constexpr (PRE == 0) {
        for (int idx = tid; idx < t.PP * nq; idx += 256) {
            int p = idx / nq, q = idx - p * nq;
            int r = p / t.PW, c = p - r * t.PW;
            f32x4 v = *(const f32x4*)(z_b + (int64_t)((t.y_first + 1 + r) * W2 + t.x0 + c) * C + q * 4);
            *(f32x4*)&lds[p * CSZ + q * 4] = v;
        }
    } else {
        float* zo_b = zoutP + (int64_t)b * (g.H + 2) * W2 * C;
        const int e_end = t.s + g.tpx < t.npx ? t.s + g.tpx : t.npx;       // the tile's pixels: band indices [t.s, e_end)
        // patch pixel p = (r, c): ring row t.y_first + 1 + r, ring column t.x0 + c; inside the plane when both are in 1..H / 1..W;
        // one of this tile's own pixels when its band index (row * bwb + c - 1) falls into [t.s, e_end)
        static_assert(PRE == 1 || KZ == 24, "the inverse map is built for C = 48 only");
        if constexpr (KZ == 8) {                 // C == 12
            constexpr int CC = 12;
            const float* sw = anicP;              // wave-uniform addresses: the map's 168 parameters come through the scalar cache
            for (int p = tid; p < t.PP; p += 256) {
                const int r = p / t.PW, c = p - r * t.PW;
                const int ry = t.y_first + 1 + r, rx = t.x0 + c;
                f32x4 o[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                if (ry >= 1 && ry <= g.H && rx >= 1 && rx <= g.W) {
                    const int64_t po = ((int64_t)ry * W2 + rx) * CC;
                    float v[CC];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const f32x4 x = *(const f32x4*)(z_b + po + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float u = x[e] + sw[q * 4 + e]; v[q * 4 + e] = u * sw[CC + q * 4 + e]; }
                    }
#pragma unroll
                    for (int co = 0; co < CC; ++co) {
                        float acc = 0.f;
#pragma unroll
                        for (int ci = 0; ci < CC; ci += 4) {
                            const f32x4 w = *(const f32x4*)(sw + 2 * CC + co * CC + ci);
                            acc = fmaf(w[0], v[ci], acc); acc = fmaf(w[1], v[ci + 1], acc);
                            acc = fmaf(w[2], v[ci + 2], acc); acc = fmaf(w[3], v[ci + 3], acc);
                        }
                        o[co >> 2][co & 3] = acc;
                    }
                    const int bi = (ry - 1) * t.bwb + c - 1;
                    if (c >= 1 && c <= t.bwb && bi >= t.s && bi < e_end) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) *(f32x4*)(zo_b + po + q * 4) = o[q];
                    }
                }
                *(f32x4*)&lds[p * CSZ] = o[0];
                *(f32x4*)&lds[p * CSZ + 4] = o[1];
            }
        } else {                                 // C == 48
            constexpr int CC = 48;
            const int n = lane & 15, gk = lane >> 4;
            const float* wm = anicP + 2 * CC;
            f32x4 a[3][3], bb[3], sc[3];
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                bb[tt] = *(const f32x4*)(anicP + 16 * tt + 4 * gk);
                sc[tt] = *(const f32x4*)(anicP + CC + 16 * tt + 4 * gk);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) a[tt][jj] = *(const f32x4*)(wm + (16 * tt + n) * CC + 16 * jj + 4 * gk);
            }
            for (int base = wave * 64; base < t.PP; base += 256) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    const int p = base + pt * 16 + n;
                    const bool live = p < t.PP;
                    const int pc = live ? p : 0;
                    const int r = pc / t.PW, c = pc - r * t.PW;
                    const int ry = t.y_first + 1 + r, rx = t.x0 + c;
                    const bool inplane = live && ry >= 1 && ry <= g.H && rx >= 1 && rx <= g.W;
                    const int64_t po = ((int64_t)ry * W2 + rx) * CC + 4 * gk;
                    f32x4 v[3];
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        v[jj] = inplane ? *(const f32x4*)(z_b + po + 16 * jj) : f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (PRE == 1) { v[jj] = v[jj] + bb[jj]; v[jj] = v[jj] * sc[jj]; }
                    }
                    f32x4 acc[3];
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) {
                        acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][jj][e], v[jj][e], acc[tt], 0, 0, 0);
                    }
                    if constexpr (PRE == 2) {    // the inverse pair (Permutations.py:119-123, FlowActNorms.py:93-96), then this step's
                                                 // z = (z + shiftFt) * scaleFt (FlowAffineCouplingsAblation.py:170-175), as anic48_mfma_kernel
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt) { acc[tt] = acc[tt] * sc[tt]; acc[tt] = acc[tt] - bb[tt]; }
                        if (inplane) {
                            const float* sf = sfP + ((int64_t)(b * (g.H + 2) + ry) * W2 + rx) * (2 * CC) + 8 * gk;
#pragma unroll
                            for (int tt = 0; tt < 3; ++tt) {
                                const f32x4 f0 = *(const f32x4*)(sf + 32 * tt), f1 = *(const f32x4*)(sf + 32 * tt + 4);
                                acc[tt][0] = acc[tt][0] + f0[0]; acc[tt][0] = acc[tt][0] * f0[1];
                                acc[tt][1] = acc[tt][1] + f0[2]; acc[tt][1] = acc[tt][1] * f0[3];
                                acc[tt][2] = acc[tt][2] + f1[0]; acc[tt][2] = acc[tt][2] * f1[1];
                                acc[tt][3] = acc[tt][3] + f1[2]; acc[tt][3] = acc[tt][3] * f1[3];
                            }
                        }
                    }
                    if (!inplane) { acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; }
                    const int bi = (ry - 1) * t.bwb + c - 1;
                    if (inplane && c >= 1 && c <= t.bwb && bi >= t.s && bi < e_end) {
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt) *(f32x4*)(zo_b + po + 16 * tt) = acc[tt];
                    }
                    if (live) {                  // channels 0..23 = z1: accumulator 0 (4 gk ..) and, for gk < 2, accumulator 1 (16 + 4 gk ..)
                        *(f32x4*)&lds[p * CSZ + 4 * gk] = acc[0];
                        if (gk < 2) *(f32x4*)&lds[p * CSZ + 16 + 4 * gk] = acc[1];
                    }
                }
            }
        }
    }
    // accumulators start from the stored conditioning-part sums
    f32x16 acc[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float* src = h2P + (opix[n] >= 0 ? opix[n] : 0) * IFL_H2 + 4 * h;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = *(const f32x4*)(src + m * 32 + gq * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][n][gq * 4 + e] = v[e];
            }
    }
    __syncthreads();
    if constexpr (PRE != 0) __builtin_amdgcn_sched_barrier(0);      // the tap loop's hoisted weight loads stay behind the map (registers)
    const float* wz = prm + off.w1[0] + (int64_t)NCC * (9 * 4 * 2 * 256) + lane * 4;
    constexpr int nkq = KZ / 8;               // compile-time, so the whole tap loop unrolls and its weight loads are hoisted
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int tapoff = ((tap / 3) * t.PW + (tap % 3)) * (CSZ * 4);
#pragma unroll
        for (int kq = 0; kq < nkq; ++kq) {
            f32x4 wa[2], xb[NT];
#pragma unroll
            for (int m = 0; m < 2; ++m) wa[m] = *(const f32x4*)(wz + ((tap * nkq + kq) * 2 + m) * 256);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                xb[n] = *(const f32x4*)((const char*)lds + a0[n] + tapoff + kq * 32);
            if (BF16) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma32x4<true>(wa[m], xb[n], acc[m][n]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[m][n] = mfma32(wa[m][e], xb[n][e], acc[m][n]);
            }
        }
    }
    finish_and_store<NT, IFL_HID, BF16>(acc, prm + off.an1[0], prm + off.w2[0], prm + off.an2[0], haP, opix, lane, h);
}

}  // namespace

int ifl_k_conv1_ft(const float* ft, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                   int nsteps, const IflStepOff& off, const IflGeom& g, int B, int level, int bf16, hipStream_t st)
{
    if (nsteps <= 0) return 0;
    dim3 grid(g.ntiles, B, nsteps), block(256);
    if (bf16) {
        static const int region_form = [] { const char* e = getenv("IFL_BF16_REGION"); return e ? atoi(e) : 1; }();
        if (region_form) {
            const int64_t nrb = (int64_t)((g.H + RBH - 1) / RBH) * ((g.W + RBW - 1) / RBW) * B;
            const int64_t nwg = (nrb + 7) / 8 * 8 * RB_SG * ((nsteps + RB_SG - 1) / RB_SG);
            if (nwg >= (int64_t)1 << 31) { ifl_set_error("conv1_ft (bf16): %lld workgroups exceed the grid", (long long)nwg); return IFL_ENOTSUP; }
            dim3 rgrid((unsigned)nwg);
            if (level == 1) hipLaunchKernelGGL(conv1_ft_bf16r_kernel<1>, rgrid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g.H, g.W, B, nsteps);
            else hipLaunchKernelGGL(conv1_ft_bf16r_kernel<2>, rgrid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g.H, g.W, B, nsteps);
        }
        else if (level == 1) hipLaunchKernelGGL(conv1_ft_bf16_kernel<1>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
        else hipLaunchKernelGGL(conv1_ft_bf16_kernel<2>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
        IFL_LAUNCH_CHECK();
        return 0;
    }
    if (level == 1) hipLaunchKernelGGL(conv1_ft_kernel<1>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
    else hipLaunchKernelGGL(conv1_ft_kernel<2>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_conv1_ft_wino(const float* ft, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                        int nsteps, const IflStepOff& off, int B, int H, int W, int level, hipStream_t st)
{
    if (nsteps <= 0) return 0;
    const IflWGeom g = ifl_wgeom(B, H, W);
    if ((int64_t)B * (H + 2) * (W + 2) * IFL_FT >= (int64_t)1 << 31) {
        ifl_set_error("conv1_ft (Winograd): conditioning slab of %d x %d x %d exceeds 2^31 floats", B, H, W);
        return IFL_ENOTSUP;
    }
    dim3 grid((g.ntiles + 31) / 32, 1, nsteps), block(512);
    if (level == 1) hipLaunchKernelGGL(conv1_ft_wino_kernel<1>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
    else hipLaunchKernelGGL(conv1_ft_wino_kernel<2>, grid, block, 0, st, ft, h2all, h2_step_stride, params, steps, off, g);
    IFL_LAUNCH_CHECK();
    return 0;
}

#ifdef IFL_EXP_TIME
extern "C" int ifl_dbg_read(void* host, size_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ifl_dbg), bytes); }
#endif

// conv1_finish with the step's ActNorm + 1x1 conv applied on the way in (encode chain, fp32): reads z_in, writes the mapped z
// of every pixel to z_out (another buffer with a zero ring) and fAffine's finished hidden activations to ha
int ifl_k_conv1_finish_pre(const float* z_in, float* z_out, const float* anic_fwd, const float* h2, float* ha, const float* prm,
                           const IflStepOff& off, const IflGeom& g, int B, int C, hipStream_t st, const float* sf_inverse)
{
    dim3 grid(g.ntiles, B), block(256);
    if (z_in == z_out) { ifl_set_error("conv1_finish (fused ActNorm + 1x1): the mapped z needs a buffer of its own"); return IFL_EINVAL; }
    if (sf_inverse) {                         // decode: `anic_fwd` is the previous step's inverse block, sf_inverse this step's sf slab
        if (C != 48) { ifl_set_error("conv1_finish (fused inverse ActNorm + 1x1): C=%d not supported (48)", C); return IFL_ENOTSUP; }
        hipLaunchKernelGGL((conv1_finish_kernel<24, false, 2>), grid, block, 0, st, z_in, h2, ha, prm, off, g, C, z_out, anic_fwd, sf_inverse);
    }
    else if (C == 12) hipLaunchKernelGGL((conv1_finish_kernel<8, false, 1>), grid, block, 0, st, z_in, h2, ha, prm, off, g, C, z_out, anic_fwd);
    else if (C == 48) hipLaunchKernelGGL((conv1_finish_kernel<24, false, 1>), grid, block, 0, st, z_in, h2, ha, prm, off, g, C, z_out, anic_fwd);
    else { ifl_set_error("conv1_finish (fused ActNorm + 1x1): C=%d not supported (12 or 48)", C); return IFL_ENOTSUP; }
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_conv1_finish(const float* z, const float* h2, float* ha, const float* prm, const IflStepOff& off,
                       const IflGeom& g, int B, int C, int bf16, hipStream_t st)
{
    dim3 grid(g.ntiles, B), block(256);
    if (ifl_kz(C) == 8 && !bf16) hipLaunchKernelGGL((conv1_finish_kernel<8, false>), grid, block, 0, st, z, h2, ha, prm, off, g, C);
    else if (ifl_kz(C) == 24 && !bf16) hipLaunchKernelGGL((conv1_finish_kernel<24, false>), grid, block, 0, st, z, h2, ha, prm, off, g, C);
    else if (ifl_kz(C) == 8) hipLaunchKernelGGL((conv1_finish_kernel<8, true>), grid, block, 0, st, z, h2, ha, prm, off, g, C);
    else if (ifl_kz(C) == 24) hipLaunchKernelGGL((conv1_finish_kernel<24, true>), grid, block, 0, st, z, h2, ha, prm, off, g, C);
    else { ifl_set_error("conv1_finish: C=%d not supported (12 or 48)", C); return IFL_ENOTSUP; }
    IFL_LAUNCH_CHECK();
    return 0;
}
