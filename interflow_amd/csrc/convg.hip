// convg.hip -- generic 3x3 / 1x1 'same' convolution on ringed channels-last planes as an implicit GEMM on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32), with a fused epilogue: bias, (leaky) ReLU, up to two scaled residual adds.
//
// Used by the conditioning encoder (rrdb.hip): every nn.Conv2d of RRDBNet (RRDBNet_arch.py:13-17,60-75) including the
// dense concatenations of ResidualDenseBlock_5C (torch.cat at :25-28 becomes "read the first Cin channels of the
// block's buffer, write 32 more behind them") and its residuals (x5*0.2 + x, :29; out*0.2 + x, :45).
//
// GEMM orientation and fragments are those of conv1.hip: D[out ch][pixel] += W[out ch][k] * X[k][pixel], k = (tap, in
// channel); weights pre-packed in fragment order and streamed global -> VGPR, the (rows+2) x (bw+2) pixel patch of a
// 32-channel chunk staged in LDS and reused by the taps and the four waves.  One workgroup = 128 pixels x (32*MT) output
// channels; wave w owns pixels [32w, 32w+32).
#include "ifl_common.h"

namespace {

constexpr int LD_SLOTS = (IFL_PP_MAX * 8 + 255) / 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// HALF: a tile of 64 pixels instead of 128 -- waves (w & 1) own 32 pixels each, waves (w >> 1) split the K loop (k-steps
// {0,1} / {2,3} of every chunk and tap) and add their halves through LDS.  Twice the workgroups of half the length for
// launches that would leave most CUs idle (the encoder's 40x40 planes at batch 8: 104 workgroups).  The full form keeps the
// same two partial sums in two accumulator sets and adds them last, so both forms give the same bits and the choice may
// depend on the batch size without breaking batch invariance.
template <int MT, int TAPS, bool HALF>
__global__ __launch_bounds__(256, 2)
void convg_kernel(IflConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[LD_SLOTS * 32 * IFL_CS];
    const IflGeom& g = a.g;
    const IflTile t = ifl_tile(g, blockIdx.x);
    if (!t.active) return;
    int b = blockIdx.y;
    const int grp = blockIdx.z;
    const float *src0 = a.src[0], *src1 = a.src[1], *wbase = a.w, *biasp = a.bias, *res1p = a.res1;
    float* outp = a.out;
    if (a.nz > 1) {                                               // batched: problem z of nz
        const int z = b / a.Bz;
        b -= z * a.Bz;
        src0 += z * a.z_src[0];
        if (src1) src1 += z * a.z_src[1];
        wbase += z * a.z_w;
        if (biasp) biasp += z * a.z_bias;
        if (res1p) res1p += z * a.z_res1;
        outp += z * a.z_out;
    }
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int pg = HALF ? (wave & 1) : wave;                      // 32-pixel group of the tile
    const int kh = HALF ? (wave >> 1) : 0;                        // K half of this wave
    const int W2 = g.W + 2;
    const int64_t plane = (int64_t)(g.H + 2) * W2;

    // this lane's pixel
    int a0;
    int64_t opix;
    {
        int idx = t.s + pg * 32 + j;
        bool valid = idx < t.npx && pg * 32 + j < g.tpx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        constexpr int HALO = TAPS == 9 ? 1 : 0;                   // 1x1: the patch still has the halo, read its centre
        a0 = (((ty - t.y_first - 1 + (1 - HALO)) * t.PW + tx + (1 - HALO)) * IFL_CS + 4 * h) * 4 + kh * 64;
        opix = valid ? ((int64_t)b * plane + (int64_t)(ty + 1) * W2 + t.x0 + tx + 1) : -1;
    }

    int srcpix[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) {
        int p = (tid + i * 256) >> 3;
        int r = p / t.PW, c = p - r * t.PW;
        srcpix[i] = (p < t.PP) ? ((t.y_first + 1 + r) * W2 + t.x0 + c) : 0;
    }
    const int q4 = (tid & 7) * 4;

    constexpr int NH = HALF ? 1 : 2;                              // partial sums a wave holds: k-steps {0,1} and {2,3}
    f32x16 acc[NH][MT];
#pragma unroll
    for (int n = 0; n < NH; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    const int nchunks = a.nchunks;
    constexpr int KQ = 4;                                         // k-steps of 8 channels per 32-channel chunk
    constexpr int KQW = HALF ? 2 : 4;                             // ... of which this wave multiplies KQW
    // weights: one record per 32 output channels, [chunk][tap][kq][64 lanes x 4]; this workgroup reads records grp*MT .. +MT
    const int64_t wrec = (int64_t)nchunks * (TAPS * KQ * 256);
    const float* wl = wbase + (int64_t)grp * MT * wrec + lane * 4 + kh * (2 * 256);

    auto chunk_src = [&](int cc, int i) -> const float* {         // source pointer of staging slot i for chunk cc
        int s = 0, c = cc;
        if (a.nsrc > 1 && cc >= a.src_chunks[0]) { s = 1; c = cc - a.src_chunks[0]; }
        return (s ? src1 : src0) + ((int64_t)b * plane + srcpix[i]) * a.src_stride[s] + a.src_off[s] + c * IFL_CK + q4;
    };

    f32x4 stage[LD_SLOTS];
#pragma unroll
    for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)chunk_src(0, i);

    // The weights come from global memory (L2) straight into registers, PD k-steps ahead of their use and across the chunk
    // boundary: with one or two waves per SIMD nothing else hides the L2 latency, and at a distance of one k-step (four
    // matrix instructions, 0.12 us) every k-step waited for its load -- 27 us for a launch whose arithmetic takes 5.
    constexpr int ITERS = TAPS * KQW;
    constexpr int PD = ITERS < 4 ? ITERS : 4;
    auto wptr = [&](int cc, int it, int m) -> const f32x4* {
        const int tap = it / KQW, kq = it % KQW;
        return (const f32x4*)(wl + (int64_t)cc * (TAPS * KQ * 256) + m * wrec + (tap * KQ + kq) * 256);
    };
    f32x4 wnext[PD][MT];
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int m = 0; m < MT; ++m) wnext[p][m] = *wptr(0, p, m);

    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LD_SLOTS; ++i) {
            int idx = tid + i * 256;
            *(f32x4*)&lds[(idx >> 3) * IFL_CS + (idx & 7) * 4] = stage[i];
        }
        __syncthreads();
        const bool more = cc + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int i = 0; i < LD_SLOTS; ++i) stage[i] = *(const f32x4*)chunk_src(cc + 1, i);
        }
        f32x4 wq[ITERS][MT], xb[2];
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int m = 0; m < MT; ++m) wq[p][m] = wnext[p][m];
        xb[0] = *(const f32x4*)((const char*)lds + a0);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (it + PD < ITERS) {
#pragma unroll
                for (int m = 0; m < MT; ++m) wq[it + PD][m] = *wptr(cc, it + PD, m);
            } else if (more) {
#pragma unroll
                for (int m = 0; m < MT; ++m) wnext[it + PD - ITERS][m] = *wptr(cc + 1, it + PD - ITERS, m);
            }
            if (it + 1 < ITERS) {
                const int tap = (it + 1) / KQW, kq = (it + 1) % KQW;
                const int tapoff = TAPS == 9 ? ((tap / 3) * t.PW + (tap % 3)) * (IFL_CS * 4) : 0;
                xb[(it + 1) & 1] = *(const f32x4*)((const char*)lds + a0 + tapoff + kq * 32);
            }
            const int half = HALF ? 0 : ((it % KQW) >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[half][m] = mfma32(wq[it][m][e], xb[it & 1][e], acc[half][m]);
        }
    }

    // the two halves of the K loop
    f32x16 tot[MT];
    if constexpr (!HALF) {
#pragma unroll
        for (int m = 0; m < MT; ++m) tot[m] = acc[0][m] + acc[NH - 1][m];
    } else {
        float* ex = lds + (pg * MT) * (16 * 64) + lane;
        __syncthreads();                                          // every wave's reads of the last patch are over
        if (kh == 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) ex[(m * 16 + r) * 64] = acc[0][m][r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[m][r] = acc[0][m][r] + ex[(m * 16 + r) * 64];
    }

    if (opix < 0) return;
    // epilogue: accumulator element (m, r) of lane (j, h) is output channel m*32 + (r&3) + 8*(r>>2) + 4*h of pixel j
    const int cbase = grp * 32 * MT + 4 * h;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = cbase + m * 32 + gq * 8;
            f32x4 v = {tot[m][gq * 4], tot[m][gq * 4 + 1], tot[m][gq * 4 + 2], tot[m][gq * 4 + 3]};
            if (biasp) {
                f32x4 bb = *(const f32x4*)(biasp + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + bb[e];
            }
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.2f;
            } else if (a.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (res1p) {
                f32x4 r = *(const f32x4*)(res1p + opix * a.res1_stride + a.res1_off + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v[e] * a.alpha; v[e] = v[e] + r[e]; }
            }
            if (a.res2) {
                f32x4 r = *(const f32x4*)(a.res2 + opix * a.res2_stride + a.res2_off + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v[e] * a.beta; v[e] = v[e] + r[e]; }
            }
            if (a.mask && c >= a.mask_c0) {
                f32x4 m = *(const f32x4*)(a.mask + opix * a.mask_stride + a.mask_off + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * a.mask_slope;
            }
            *(f32x4*)(outp + opix * a.out_stride + a.out_off + c) = v;
        }
}

}  // namespace

int ifl_k_convg(const IflConvArgs& a, int B, hipStream_t st)
{
    IFL_CHECK_ARG(a.cout % 32 == 0 && a.cout > 0, IFL_EINVAL, "convg: Cout=%d is not a multiple of 32", a.cout);
    IFL_CHECK_ARG(a.taps == 9 || a.taps == 1, IFL_EINVAL, "convg: taps=%d", a.taps);
    IFL_CHECK_ARG(a.nsrc >= 1 && a.nsrc <= 2 && a.nchunks >= 1, IFL_EINVAL, "convg: sources");
    IFL_CHECK_ARG(a.g.tpx == IFL_MT, IFL_EINVAL, "convg: tiling of %d pixels per tile", a.g.tpx);
    const int nz = a.nz > 1 ? a.nz : 1;
    IFL_CHECK_ARG((int64_t)B * nz <= 65535 && (nz == 1 || a.Bz == B), IFL_EINVAL, "convg: %d images x %d problems exceed the grid", B, nz);
    const int By = B * nz;                                        // grid.y: (problem, image)
    // fewer workgroups than 1.5 per CU even at 32 output channels each: 64-pixel tiles with the K loop split inside the
    // workgroup (same bits, see the kernel)
    if ((int64_t)a.g.ntiles * By * (a.cout / 32) < 384) {
        IflConvArgs h = a;
        h.g = ifl_choose_geom(a.g.H, a.g.W, IFL_MT / 2);
        dim3 grid(h.g.ntiles, By, a.cout / 32), block(256);
        if (a.taps == 9) hipLaunchKernelGGL((convg_kernel<1, 9, true>), grid, block, 0, st, h);
        else hipLaunchKernelGGL((convg_kernel<1, 1, true>), grid, block, 0, st, h);
        IFL_LAUNCH_CHECK();
        return 0;
    }
    // 64 output channels per workgroup when that still fills the chip, else 32 (twice the workgroups, half the latency each)
    const int mt = (a.cout % 64 == 0 && (int64_t)a.g.ntiles * By * (a.cout / 64) >= 256) ? 2 : 1;
    dim3 grid(a.g.ntiles, By, a.cout / (32 * mt)), block(256);
    if (mt == 2 && a.taps == 9) hipLaunchKernelGGL((convg_kernel<2, 9, false>), grid, block, 0, st, a);
    else if (mt == 1 && a.taps == 9) hipLaunchKernelGGL((convg_kernel<1, 9, false>), grid, block, 0, st, a);
    else if (mt == 2) hipLaunchKernelGGL((convg_kernel<2, 1, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((convg_kernel<1, 1, false>), grid, block, 0, st, a);
    IFL_LAUNCH_CHECK();
    return 0;
}

// ======================================================================================================================
// Weight gradient of the generic convolution:  dW[co][ci][tap] = sum over images and pixels of dY[px][co] * X[px + tap][ci]
// (torch.nn.grad.conv2d_weight for a stride-1 'same' conv).  GEMM with K = pixels: D[32 co][32 ci] += A[co][px] * B[px][ci]
// on v_mfma_f32_32x32x2_f32, two pixels per instruction.
//   workgroup  = one 32-channel input chunk (blockIdx.y) x a strided set of (image, pixel tile) pairs (blockIdx.x);
//   LDS        = the tile's dY rows [128 px][cout], the chunk's X patch [(rows+2) x (bw+2) px][32 ch], the patch position
//                of every tile pixel;
//   wave w     = the (co tile, tap) pairs p = w, w+4, ...: one 32x32 accumulator each, kept in registers over all the
//                workgroup's tiles and added to dW with float atomics once at the end.
template <int CT, int TAPS>        // CT = cout / 32 (1..3)
__global__ __launch_bounds__(256, 2)   // two workgroups per CU (74 KB of LDS each): without the bound CT = 2 takes 264 registers, one wave per SIMD
void wgrad_kernel(const float* __restrict__ X, int xs, int xoff, const float* __restrict__ DY, int dys, int dyoff,
                  float* __restrict__ dW, int cin_total, int ci0, IflGeom g, int B, int64_t xz, int64_t dyz, int64_t dwz)
{
    // blockIdx.z: one of several independent problems of the same shape (the same convolution of different steps), their
    // operands xz / dyz / dwz floats apart
    X += (int64_t)blockIdx.z * xz;
    DY += (int64_t)blockIdx.z * dyz;
    dW += (int64_t)blockIdx.z * dwz;
    constexpr int NP = CT * TAPS, PPW = (NP + 3) / 4;            // pairs, pairs per wave
    constexpr int DS = CT * 32;                                   // dY row stride in LDS: the 32 lanes of a half-wave read consecutive words
    constexpr int XS = 32;
    constexpr int NDY = IFL_MT * CT * 8 / 256;                    // float4 loads per thread: dY tile (4 CT)
    constexpr int NXP = (IFL_PP_MAX * 8 + 255) / 256;             // ... and X patch (10)
    constexpr int NDYF = IFL_MT * DS, NXPF = NXP * 32 * XS, NTF = 32 * 32 * TAPS;
    __shared__ __attribute__((aligned(16))) float smem[NDYF + NXPF > NTF ? NDYF + NXPF : NTF];
    __shared__ int pos[IFL_MT];
    float* dy = smem;                                             // [128 px][cout]
    float* xp = smem + NDYF;                                      // [patch px][32 ch]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.y;
    const int W2 = g.W + 2;
    const int64_t plane = (int64_t)(g.H + 2) * W2;

    f32x16 acc[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // staging through registers, one work item ahead: the loads of tile k + 1 travel while tile k is multiplied
    f32x4 rdy[NDY], rxp[NXP];
    int rpos[NDY];
    IflTile t{};
    auto fetch = [&](int wk) {
        const int b = wk / g.ntiles;
        t = ifl_tile(g, wk - b * g.ntiles);
        const float* dyb = DY + (int64_t)b * plane * dys + dyoff;
        const float* xb = X + (int64_t)b * plane * xs + xoff + chunk * 32;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int idx = tid + k * 256, p = idx / (CT * 8), c4 = idx - p * (CT * 8);
            const int pi = t.s + p;
            rdy[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            rpos[k] = 0;
            if (t.active && pi < t.npx) {
                const int ty = pi / t.bwb, tx = pi - ty * t.bwb;
                rdy[k] = *(const f32x4*)(dyb + ((int64_t)(ty + 1) * W2 + t.x0 + tx + 1) * dys + c4 * 4);
                rpos[k] = (ty - t.y_first - (TAPS == 9 ? 1 : 0)) * t.PW + tx + (TAPS == 9 ? 0 : 1);
            }
        }
#pragma unroll
        for (int k = 0; k < NXP; ++k) {
            const int idx = tid + k * 256, p = idx >> 3, c4 = idx & 7;
            rxp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t.active && p < t.PP) {
                const int r = p / t.PW, cc = p - r * t.PW;
                rxp[k] = *(const f32x4*)(xb + ((int64_t)(t.y_first + 1 + r) * W2 + t.x0 + cc) * xs + c4 * 4);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int idx = tid + k * 256, p = idx / (CT * 8), c4 = idx - p * (CT * 8);
            *(f32x4*)&dy[p * DS + c4 * 4] = rdy[k];
            if (c4 == 0) pos[p] = rpos[k];
        }
#pragma unroll
        for (int k = 0; k < NXP; ++k) *(f32x4*)&xp[(tid + k * 256) * 4] = rxp[k];
    };

    const int nwork = g.ntiles * B;
    int wk = blockIdx.x;
    if (wk < nwork) fetch(wk);
    for (; wk < nwork; wk += gridDim.x) {
        const int PW = t.PW;
        const bool active = t.active;
        __syncthreads();                                          // previous tile fully consumed
        commit();
        __syncthreads();
        if (wk + (int)gridDim.x < nwork) fetch(wk + gridDim.x);
        if (!active) continue;
#pragma unroll 4
        for (int kk = 0; kk < IFL_MT / 2; ++kk) {
            const int p = 2 * kk + h;
            const int pp = pos[p];
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                const int pr = wave + 4 * q;
                if (pr < NP) {                                   // wave-uniform
                    const int ct = pr / TAPS, tap = pr - ct * TAPS;
                    const int tapoff = TAPS == 9 ? (tap / 3) * PW + (tap % 3) : 0;
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(dy[p * DS + ct * 32 + j], xp[(pp + tapoff) * XS + j], acc[q], 0, 0, 0);
                }
            }
        }
    }
    // flush: accumulator element r of lane (j, h) is dW[co = ct*32 + (r&3) + 8*(r>>2) + 4*h][ci = chunk*32 + j][tap].  For a
    // fixed co the chunk's (ci, tap) are 32 * TAPS contiguous floats of dW, so each 32-row block of the accumulators goes through
    // LDS as T[co][ci][tap] and is added with coalesced float atomics (a lane-per-ci flush scatters every wave instruction over
    // 64 cache lines, which is an order of magnitude slower at the memory side).
#ifdef IFL_EXP_NOFLUSH
    if (acc[0][0] != 12345.678f) return;
#endif
    for (int ct = 0; ct < CT; ++ct) {
        __syncthreads();                                          // the main loop's / the previous block's LDS reads are over
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int pr = wave + 4 * q;
            if (pr >= NP || pr / TAPS != ct) continue;
            const int tap = pr - ct * TAPS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                smem[(co * 32 + j) * TAPS + tap] = acc[q][r];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < NTF; idx += 256) {
            const int co = idx / (32 * TAPS), rest = idx - co * (32 * TAPS);
            atomicAdd(dW + ((int64_t)(ct * 32 + co) * cin_total + (ci0 + chunk) * 32) * TAPS + rest, smem[idx]);
        }
    }
}

int ifl_k_wgrad(const float* X, int xs, int xoff, int cin, const float* DY, int dys, int dyoff, int cout, float* dW,
                int cin_total, int ci0, int taps, const IflGeom& g, int B, hipStream_t st, int nz, int64_t xz, int64_t dyz, int64_t dwz)
{
    IFL_CHECK_ARG(nz >= 1 && nz <= 65535, IFL_EINVAL, "wgrad: %d problems", nz);
    IFL_CHECK_ARG(cin % 32 == 0 && cin > 0 && cout % 32 == 0 && cout >= 32 && cout <= 96, IFL_ENOTSUP,
                  "wgrad: cin=%d cout=%d (cin a multiple of 32, cout 32, 64 or 96)", cin, cout);
    IFL_CHECK_ARG(taps == 9 || taps == 1, IFL_EINVAL, "wgrad: taps=%d", taps);
    int nwork = g.ntiles * B;
    // workgroups per input chunk: two per CU in ONE round (512 slots; a few workgroups more would run as a second round
    // beside an idle chip).  Each keeps its accumulators over its strided set of tiles and flushes them once with atomics,
    // 9 KB per 32x32 tile, so more workgroups also mean more atomic traffic.
    const int chunks = cin / 32;
    int gx = 512 / (chunks * nz);
    if (nz > 1) { if (gx < 4) gx = 4; }                            // batched: the problems fill the chip, keep ONE round
    else if (gx < 16) gx = 16;
#ifdef IFL_EXP_GX
    if (getenv("IFL_EXP_GX")) gx = atoi(getenv("IFL_EXP_GX"));
#endif
    if (gx > nwork) gx = nwork;
    dim3 grid(gx, cin / 32, nz), block(256);
#define IFL_WG(CT, T) hipLaunchKernelGGL((wgrad_kernel<CT, T>), grid, block, 0, st, X, xs, xoff, DY, dys, dyoff, dW, cin_total, ci0, g, B, xz, dyz, dwz)
    const int ct = cout / 32;
    if (taps == 9) { if (ct == 1) IFL_WG(1, 9); else if (ct == 2) IFL_WG(2, 9); else IFL_WG(3, 9); }
    else { if (ct == 1) IFL_WG(1, 1); else if (ct == 2) IFL_WG(2, 1); else IFL_WG(3, 1); }
#undef IFL_WG
    IFL_LAUNCH_CHECK();
    return 0;
}
