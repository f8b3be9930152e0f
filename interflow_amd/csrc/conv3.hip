// conv3.hip -- last layer of the coupling networks fused with the affine coupling itself.
//
// Replaces, for one conditional FlowStep (reference interflow/models/modules):
//   flow.Conv2dZeros 3x3 (64 -> 2C for fFeatures, 64 -> C for fAffine): (conv + bias) * exp(3*logs)      flow.py:67-82
//   shift = h[:,0::2], scale = sigmoid(h[:,1::2] + 2) + eps                                              FlowAffineCouplingsAblation.py:219-240, thops.py:54-55
//   encode (multReverse flips the flag, :164-165): z2 = z2/scale - shift ; z = z/scaleFt - shiftFt        :187-207
//   decode: z = (z + shiftFt)*scaleFt ; then z2 = (z2 + shift)*scale                                      :167-186
//   logdet -/+= sum_{c,h,w} log(scale)                                                                    :175,184,197,204,216-217
//
// GEMM orientation: D[out row][pixel] on v_mfma_f32_16x16x4_f32.  The output rows of each network are permuted on
// the host so that one 16-row tile holds, for 8 z channels, (shift c, shift c+1, scale c, scale c+1) in the four
// accumulator registers of lane-quad q: a lane ends up with everything it needs for channels (c, c+1) of one pixel
// and applies the coupling in registers.  fFeatures' tiles are ordered so that tile i of fAffine (z2 channels) lines
// up with tile i of fFeatures: slot sigma = 8*tile + 2*q  ->  channel (sigma + C/2) mod C.
// The per-sample sum of log(scale) is reduced lane -> wave (shuffles) -> workgroup (LDS) and written to the tile's
// slot of `ldpart`; ifl_k_ld_finalize adds the slots in a fixed order in float64 (bitwise reproducible).
#include "ifl_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float scale_of(float h, float eps) {
    // torch.sigmoid(h + 2.) + eps
    return 1.0f / (1.0f + expf(-(h + 2.0f))) + eps;
}

enum { MODE_ENC = 0, MODE_DEC_F = 1, MODE_DEC_A = 2 };

// One 32-channel chunk of one network: 9 taps x 2 k-steps of 16 channels, NTL output tiles x 2 pixel tiles.
// Pinned software pipeline (sched_barrier): while k-step `it` multiplies, the LDS reads of it+1 and the weight loads of
// it+WD are in flight; WD is deeper when few tiles give each k-step little MFMA time to hide a global load.
template <int NTL>
__device__ __forceinline__ void chunk_mma(f32x4 (&acc)[NTL][2], const float* __restrict__ w3l, int half,
                                          const float* lds, const int (&a0)[2], int PW)
{
    constexpr int ITERS = 18;
    constexpr int WD = NTL >= 6 ? 1 : (NTL >= 3 ? 2 : (NTL == 2 ? 3 : 6));
    static_assert(ITERS % WD == 0, "ring slots must be compile-time");
    f32x4 wq[WD][NTL], xb[2][2];
    auto wptr = [&](int it, int i) { return w3l + ((i * 9 + (it >> 1)) * 4 + half * 2 + (it & 1)) * 256; };
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int i = 0; i < NTL; ++i) wq[d][i] = *(const f32x4*)wptr(d, i);
#pragma unroll
    for (int n = 0; n < 2; ++n) xb[0][n] = *(const f32x4*)((const char*)lds + a0[n]);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if (it + 1 < ITERS) {
            const int tap = (it + 1) >> 1, s2 = (it + 1) & 1;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) * (IFL_CS3 * 4);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                xb[(it + 1) & 1][n] = *(const f32x4*)((const char*)lds + a0[n] + tapoff + s2 * 64);
        }
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[i][n] = mfma16(wq[it % WD][i][e], xb[it & 1][n][e], acc[i][n]);
        if (it + WD < ITERS) {
#pragma unroll
            for (int i = 0; i < NTL; ++i) wq[it % WD][i] = *(const f32x4*)wptr(it + WD, i);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NF, int NA, int MODE>
__global__ __launch_bounds__(256, 2)
void conv3_kernel(const float* __restrict__ h2P, const float* __restrict__ haP, float* __restrict__ zP, float* __restrict__ ldpart, int ld_slots,
                  int slot0, const float* __restrict__ prm, IflStepOff off, IflGeom g, int C, float eps)
{
    constexpr int LDP = (IFL_PP_MAX * 8 + 255) / 256;
    constexpr bool DO_A = (MODE == MODE_ENC || MODE == MODE_DEC_A);
    constexpr bool DO_F = (MODE == MODE_ENC || MODE == MODE_DEC_F);
    constexpr int NAA = DO_A ? NA : 1, NFF = DO_F ? NF : 1;
    __shared__ __attribute__((aligned(16))) float lds[LDP * 32 * IFL_CS3];   // >= IFL_PP_MAX pixels: branch-free staging
    __shared__ float red[4];

    const IflTile t = ifl_tile(g, blockIdx.x);
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (!t.active) {
        if (tid == 0) ldpart[(int64_t)b * ld_slots + slot0 + blockIdx.x] = 0.f;
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int W2 = g.W + 2;

    int a0[2];
    int64_t opix[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        int idx = t.s + wave * 32 + n * 16 + j;
        bool valid = idx < t.npx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        a0[n] = (((ty - t.y_first - 1) * t.PW + tx) * IFL_CS3 + 4 * q) * 4;
        opix[n] = valid ? ((int64_t)(b * (g.H + 2) + ty + 1) * W2 + t.x0 + tx + 1) : -1;
    }

    // staging sources: fAffine's activations come from `ha` (64 floats per pixel), fFeatures' from the step's slab
    // (128 floats per pixel, upper half); slots past the patch read the plane's first pixel and land in unread LDS rows
    const float* h2_b = h2P + (int64_t)b * (g.H + 2) * W2 * IFL_H2 + IFL_HID + (tid & 7) * 4;
    const float* ha_b = haP + (int64_t)b * (g.H + 2) * W2 * IFL_HID + (tid & 7) * 4;
    int srcpix[LDP];
#pragma unroll
    for (int i = 0; i < LDP; ++i) {
        int p = (tid + i * 256) >> 3;
        int r = p / t.PW, c = p - r * t.PW;
        srcpix[i] = (p < t.PP) ? ((t.y_first + 1 + r) * W2 + t.x0 + c) : 0;
    }
    auto src_of = [&](int ch, int i) -> const float* {       // ch is a compile-time constant at every call site
        return (ch < 2) ? ha_b + srcpix[i] * IFL_HID + ch * IFL_CK : h2_b + srcpix[i] * IFL_H2 + (ch - 2) * IFL_CK;
    };

    f32x4 accA[NAA][2], accF[NFF][2];
#pragma unroll
    for (int i = 0; i < NAA; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) accA[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NFF; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) accF[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // chunk = 32 hidden channels: chunks 0,1 belong to fAffine, 2,3 to fFeatures
    constexpr int CH0 = DO_A ? 0 : 2, CH1 = DO_F ? 4 : 2;
    f32x4 stage[LDP];
#pragma unroll
    for (int i = 0; i < LDP; ++i) stage[i] = *(const f32x4*)src_of(CH0, i);
#pragma unroll
    for (int ch = CH0; ch < CH1; ++ch) {
        const int net = ch >> 1, half = ch & 1;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LDP; ++i) {
            int idx = tid + i * 256;
            *(f32x4*)&lds[(idx >> 3) * IFL_CS3 + (idx & 7) * 4] = stage[i];
        }
        __syncthreads();
        if (ch + 1 < CH1) {
#pragma unroll
            for (int i = 0; i < LDP; ++i) stage[i] = *(const f32x4*)src_of(ch + 1, i);
        }
        const float* w3 = prm + off.w3[net] + lane * 4;
        if (net == 0) chunk_mma<NAA>(accA, w3, half, lds, a0, t.PW);
        else chunk_mma<NFF>(accF, w3, half, lds, a0, t.PW);
    }

    // ---- epilogue: Conv2dZeros bias/scale, sigmoid, coupling, log-det ------------------------------------------
    const float* b3a = prm + off.b3[0];
    const float* b3f = prm + off.b3[1];
    const int Ch = C / 2;
    float ld = 0.f;
    constexpr int NTILE = DO_F ? NF : NA;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        const int sigma = 8 * i + 2 * q;
        const bool vF = DO_F && sigma < C;
        const bool vA = DO_A && i < NA && sigma < Ch;
        if (!(vF || vA)) continue;
        const int c = DO_F ? (sigma + Ch) % C : Ch + sigma;
        f32x4 bA, eA, bF, eF;
        if (DO_A && i < NA) { bA = *(const f32x4*)(b3a + i * 32 + q * 4); eA = *(const f32x4*)(b3a + i * 32 + 16 + q * 4); }
        if (DO_F) { bF = *(const f32x4*)(b3f + i * 32 + q * 4); eF = *(const f32x4*)(b3f + i * 32 + 16 + q * 4); }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (opix[n] < 0) continue;
            float* zp = zP + opix[n] * C + c;
            f32x2 z = *(const f32x2*)zp;
            if (MODE == MODE_ENC) {
                if (vA) {
                    const f32x4 a = accA[i < NAA ? i : 0][n];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float sh = (a[e] + bA[e]) * eA[e];
                        float sc = scale_of((a[2 + e] + bA[2 + e]) * eA[2 + e], eps);
                        z[e] = z[e] / sc;
                        z[e] = z[e] - sh;
                        ld -= logf(sc);
                    }
                }
                const f32x4 f = accF[i < NFF ? i : 0][n];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float sh = (f[e] + bF[e]) * eF[e];
                    float sc = scale_of((f[2 + e] + bF[2 + e]) * eF[2 + e], eps);
                    z[e] = z[e] / sc;
                    z[e] = z[e] - sh;
                    ld -= logf(sc);
                }
            } else if (MODE == MODE_DEC_F) {
                const f32x4 f = accF[i < NFF ? i : 0][n];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float sh = (f[e] + bF[e]) * eF[e];
                    float sc = scale_of((f[2 + e] + bF[2 + e]) * eF[2 + e], eps);
                    z[e] = z[e] + sh;
                    z[e] = z[e] * sc;
                    ld += logf(sc);
                }
            } else {
                const f32x4 a = accA[i < NAA ? i : 0][n];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float sh = (a[e] + bA[e]) * eA[e];
                    float sc = scale_of((a[2 + e] + bA[2 + e]) * eA[2 + e], eps);
                    z[e] = z[e] + sh;
                    z[e] = z[e] * sc;
                    ld += logf(sc);
                }
            }
            *(f32x2*)zp = z;
        }
    }
    // lane -> wave -> workgroup reduction of the log-det partial
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ld += __shfl_xor(ld, o, 64);
    if (lane == 0) red[wave] = ld;
    __syncthreads();
    if (tid == 0) ldpart[(int64_t)b * ld_slots + slot0 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int NF, int NA>
int launch(const float* h2, const float* ha, float* z, float* ldpart, int ld_slots, int slot0, const float* prm, const IflStepOff& off,
           const IflGeom& g, int B, int C, int mode, float eps, hipStream_t st)
{
    dim3 grid(g.ntiles, B), block(256);
    if (mode == MODE_ENC)
        hipLaunchKernelGGL((conv3_kernel<NF, NA, MODE_ENC>), grid, block, 0, st, h2, ha, z, ldpart, ld_slots, slot0, prm, off, g, C, eps);
    else if (mode == MODE_DEC_F)
        hipLaunchKernelGGL((conv3_kernel<NF, NA, MODE_DEC_F>), grid, block, 0, st, h2, ha, z, ldpart, ld_slots, slot0, prm, off, g, C, eps);
    else
        hipLaunchKernelGGL((conv3_kernel<NF, NA, MODE_DEC_A>), grid, block, 0, st, h2, ha, z, ldpart, ld_slots, slot0, prm, off, g, C, eps);
    IFL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int ifl_k_conv3(const float* h2, const float* ha, float* z, float* ldpart, int ld_slots, int slot0, const float* prm,
                const IflStepOff& off, const IflGeom& g, int B, int C, int mode, float eps, hipStream_t st)
{
    if (C == 12) return launch<2, 1>(h2, ha, z, ldpart, ld_slots, slot0, prm, off, g, B, C, mode, eps, st);
    if (C == 48) return launch<6, 3>(h2, ha, z, ldpart, ld_slots, slot0, prm, off, g, B, C, mode, eps, st);
    ifl_set_error("conv3: C=%d not supported (12 or 48)", C);
    return IFL_ENOTSUP;
}
