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

namespace {

constexpr int NCC = IFL_FT / IFL_CK;          // 10 chunks of 32 conditioning channels
constexpr int LD_SLOTS = (IFL_PP_MAX * 8 + 255) / 256;   // float4 patch loads per thread per chunk

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// PXW: pixels per wave (32 or 64).  NPART: pixel parts per workgroup; nets per workgroup = 4 / NPART.
template <int PXW, int NPART>
__global__ __launch_bounds__(256, 2)
void conv1_kernel(const float* __restrict__ ftP, const float* __restrict__ zP, float* __restrict__ h2P,
                  const float* __restrict__ prm, IflStepOff off, IflGeom g, int C, int net0)
{
    static_assert(PXW * NPART == IFL_MT, "tile size");
    constexpr int NT = PXW / 32;              // 32-pixel MFMA column tiles per wave
    constexpr int NETS = 4 / NPART;
    __shared__ __attribute__((aligned(16))) float lds[LD_SLOTS * 32 * IFL_CS];   // >= IFL_PP_MAX pixels: staging stores need no bound test

    const IflTile t = ifl_tile(g, blockIdx.x);
    if (!t.active) return;                    // uniform for the whole workgroup
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int net = (NETS == 2) ? (wave & 1) : net0;
    const int part = (NETS == 2) ? (wave >> 1) : wave;
    const int W2 = g.W + 2;
    const int KZ = ifl_kz(C);

    // ---- this lane's pixels -------------------------------------------------------------------
    int a0[NT];            // LDS byte address of the pixel's top-left tap, + 16*h
    int64_t opix[NT];      // ring pixel index of the output pixel (or -1 when past the band's end)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int idx = t.s + part * PXW + n * 32 + j;
        bool valid = idx < t.npx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        a0[n] = (((ty - t.y_first - 1) * t.PW + tx) * IFL_CS + 4 * h) * 4;
        opix[n] = valid ? ((int64_t)(b * (g.H + 2) + ty + 1) * W2 + t.x0 + tx + 1) : -1;
    }

    // ---- patch staging: per-thread source offsets are the same for every chunk -----------------
    // Slots past the patch read the plane's first float4 (valid memory) and land in LDS rows nobody reads, so the
    // loads and stores below carry no branches.
    const float* ft_b = ftP + (int64_t)b * (g.H + 2) * W2 * IFL_FT;
    int srcoff[LD_SLOTS];  // float offset of this thread's float4 inside the batch's ringed plane (chunk 0)
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

    // Weights: wave-uniform base (SGPR) + per-lane 32-bit offset.  They stream through a register ring WD k-steps
    // deep, so each global load has WD * 16 * NT MFMAs of cover (the blocks of consecutive k-steps are contiguous in
    // memory, also across chunk boundaries; the last prefetches read a little past the field, inside the block).
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

    // ---- conditioning-feature part: K = 9 taps x 320 channels -----------------------------------
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

    // ---- z1 part (fAffine only): K = 9 taps x KZ channels (C/2 padded to a multiple of 8 with zero weights) --
    const bool wg_has_a = (NETS == 2) || (net0 == 0);
    if (wg_has_a) {
        const int CSZ = KZ + 4;               // 12 or 28 words: CSZ/4 odd
        const int nq = KZ / 4;                // float4 per pixel
        const float* z_b = zP + (int64_t)b * (g.H + 2) * W2 * C;
        __syncthreads();
        for (int idx = tid; idx < t.PP * nq; idx += 256) {
            int p = idx / nq, q = idx - p * nq;
            int r = p / t.PW, c = p - r * t.PW;
            f32x4 v = *(const f32x4*)(z_b + (int64_t)((t.y_first + 1 + r) * W2 + t.x0 + c) * C + q * 4);
            *(f32x4*)&lds[p * CSZ + q * 4] = v;
        }
        __syncthreads();
        if (net == 0) {
            const float* wz = wnet + (int64_t)NCC * (9 * 4 * 2 * 256) + lane4;
            const int nkq = KZ / 8;
            int az[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) az[n] = (a0[n] - 16 * h) / IFL_CS * CSZ + 16 * h;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tapoff = ((tap / 3) * t.PW + (tap % 3)) * (CSZ * 4);
                for (int kq = 0; kq < nkq; ++kq) {
                    f32x4 wa[2], xb[NT];
#pragma unroll
                    for (int m = 0; m < 2; ++m) wa[m] = *(const f32x4*)(wz + ((tap * nkq + kq) * 2 + m) * 256);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        xb[n] = *(const f32x4*)((const char*)lds + az[n] + tapoff + kq * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc[m][n] = mfma32(wa[m][e], xb[n][e], acc[m][n]);
                }
            }
        }
    }

    // ---- ActNorm + ReLU on the 64 x PXW accumulator block (flow.py:60-64, FlowActNorms.py:87-92) -------------
    // accumulator element (m, r) of lane (j, h) is channel m*32 + (r&3) + 8*(r>>2) + 4*h; the tables are stored
    // in that order: an[0/1 = bias/scale][m][h][r].
    {
        const float* an = prm + off.an1[net];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 bb[4], ss[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bb[q] = *(const f32x4*)(an + (m * 2 + h) * 16 + q * 4);
                ss[q] = *(const f32x4*)(an + 64 + (m * 2 + h) * 16 + q * 4);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = (acc[m][n][r] + bb[r >> 2][r & 3]) * ss[r >> 2][r & 3];
                    acc[m][n][r] = v > 0.f ? v : 0.f;
                }
        }
    }

    // ---- 1x1 conv 64 -> 64 straight from the accumulators, then ActNorm + ReLU, then store --------------------
    const float* w2 = prm + off.w2[net] + lane * 4;
    const float* an2 = prm + off.an2[net];
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
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 wv[2];
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) wv[m2] = *(const f32x4*)(w2 + ((m2 * 2 + m) * 4 + gq) * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) y[m2] = mfma32(wv[m2][e], acc[m][n][gq * 4 + e], y[m2]);
            }
        if (opix[n] >= 0) {
            float* dst = h2P + opix[n] * IFL_H2 + net * IFL_HID + 4 * h;
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

}  // namespace

int ifl_k_conv1(const float* ft, const float* z, float* h2, const float* prm, const IflStepOff& off,
                const IflGeom& g, int B, int C, int nets, hipStream_t st)
{
    dim3 grid(g.ntiles, B), block(256);
    if (nets == 2)
        hipLaunchKernelGGL((conv1_kernel<64, 2>), grid, block, 0, st, ft, z, h2, prm, off, g, C, 0);
    else
        hipLaunchKernelGGL((conv1_kernel<32, 4>), grid, block, 0, st, ft, z, h2, prm, off, g, C, nets);
    IFL_LAUNCH_CHECK();
    return 0;
}
