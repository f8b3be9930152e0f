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
//
// fFeatures sees only the conditioning features, so its shift / scale do not depend on the running z:
//   conv3f_kernel  computes them for every conditional step of a bulk launch (grid.z = step) right after conv1_ft has
//                  produced the step's hidden activations, and stores (shift c, scale c, shift c+1, scale c+1) per pixel
//                  in the step's `sf` slab plus the tiles' sums of log(scale) in a reserved range of `ldpart`;
//   conv3_kernel   (in the per-step chain) evaluates fAffine only and applies both couplings, reading fFeatures' from `sf`;
//   fapply_kernel  is the decode direction's first half, z = (z + shiftFt) * scaleFt: pointwise.
// With a prepared conditioning (ifl_cond_prepare) the slabs serve any number of passes.
#include "ifl_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float scale_of(float h, float eps) {
    // torch.sigmoid(h + 2.) + eps
    return 1.0f / (1.0f + expf(-(h + 2.0f))) + eps;
}

enum { MODE_ENC = 0, MODE_DEC_A = 2 };

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

// shared by the two MFMA kernels: this lane's pixels (LDS address of the top-left tap, ring pixel index or -1) and the
// per-thread source pixels of the patch staging loads
struct Conv3Idx { int a0[2]; int64_t opix[2]; int srcpix[(IFL_PP_MAX * 8 + 255) / 256]; };

__device__ __forceinline__ void conv3_index(Conv3Idx& x, const IflTile& t, const IflGeom& g, int b, int tid)
{
    constexpr int LDP = (IFL_PP_MAX * 8 + 255) / 256;
    const int wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int W2 = g.W + 2;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        int idx = t.s + wave * 32 + n * 16 + j;
        bool valid = idx < t.npx;
        int idc = valid ? idx : t.npx - 1;
        int ty = idc / t.bwb, tx = idc - ty * t.bwb;
        x.a0[n] = (((ty - t.y_first - 1) * t.PW + tx) * IFL_CS3 + 4 * q) * 4;
        x.opix[n] = valid ? ((int64_t)(b * (g.H + 2) + ty + 1) * W2 + t.x0 + tx + 1) : -1;
    }
#pragma unroll
    for (int i = 0; i < LDP; ++i) {
        int p = (tid + i * 256) >> 3;
        int r = p / t.PW, c = p - r * t.PW;
        x.srcpix[i] = (p < t.PP) ? ((t.y_first + 1 + r) * W2 + t.x0 + c) : 0;     // past the patch: valid memory, unread LDS rows
    }
}

// 64 hidden channels at `src` (pixel stride `pstride` floats) -> NTL output tiles, two 32-channel chunks through LDS
template <int NTL>
__device__ __forceinline__ void conv3_mma(f32x4 (&acc)[NTL][2], const float* __restrict__ src, int pstride,
                                          const float* __restrict__ w3, float* lds, const Conv3Idx& x, int PW, int tid)
{
    constexpr int LDP = (IFL_PP_MAX * 8 + 255) / 256;
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 stage[LDP];
#pragma unroll
    for (int i = 0; i < LDP; ++i) stage[i] = *(const f32x4*)(src + x.srcpix[i] * pstride);      // 32-bit offsets inside one image's plane (check_shape)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LDP; ++i) {
            int idx = tid + i * 256;
            *(f32x4*)&lds[(idx >> 3) * IFL_CS3 + (idx & 7) * 4] = stage[i];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < LDP; ++i) stage[i] = *(const f32x4*)(src + x.srcpix[i] * pstride + IFL_CK);
        }
        chunk_mma<NTL>(acc, w3, half, lds, x.a0, PW);
    }
}

__device__ __forceinline__ void conv3_ld_reduce(float ld, float* red, float* dst, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ld += __shfl_xor(ld, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ld;
    __syncthreads();
    if (tid == 0) *dst = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- z-independent part: fFeatures' last conv for all steps of a bulk launch ------------------------------------------
template <int NF>
__global__ __launch_bounds__(256, 2)
void conv3f_kernel(const float* __restrict__ h2all, int64_t h2_step, float* __restrict__ sfall, int64_t sf_step,
                   float* __restrict__ ldpart, int ld_slots, int slot0, const float* __restrict__ params, IflStepList steps,
                   IflStepOff off, IflGeom g, int C, float eps)
{
    constexpr int LDP = (IFL_PP_MAX * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[LDP * 32 * IFL_CS3];
    __shared__ float red[4];
    const int step = blockIdx.z, b = blockIdx.y, tid = threadIdx.x;
    float* ldp = ldpart + (int64_t)b * ld_slots + slot0 + step * g.ntiles + blockIdx.x;
    const IflTile t = ifl_tile(g, blockIdx.x);
    if (!t.active) {
        if (tid == 0) *ldp = 0.f;
        return;
    }
    const float* prm = params + steps.off[step];
    const int q = (tid & 63) >> 4;
    Conv3Idx x;
    conv3_index(x, t, g, b, tid);
    f32x4 acc[NF][2];
    const float* src = h2all + (int64_t)step * h2_step + (int64_t)b * (g.H + 2) * (g.W + 2) * IFL_H2 + IFL_HID + (tid & 7) * 4;
    conv3_mma<NF>(acc, src, IFL_H2, prm + off.w3[1] + (tid & 63) * 4, lds, x, t.PW, tid);

    const float* b3f = prm + off.b3[1];
    float* sf = sfall + (int64_t)step * sf_step;
    const int Ch = C / 2;
    float ld = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int sigma = 8 * i + 2 * q;
        if (sigma >= C) continue;
        const int c = (sigma + Ch) % C;
        const f32x4 bF = *(const f32x4*)(b3f + i * 32 + q * 4), eF = *(const f32x4*)(b3f + i * 32 + 16 + q * 4);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (x.opix[n] < 0) continue;
            const f32x4 f = acc[i][n];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                o[2 * e] = (f[e] + bF[e]) * eF[e];                                        // shift
                o[2 * e + 1] = scale_of((f[2 + e] + bF[2 + e]) * eF[2 + e], eps);           // scale
                ld += logf(o[2 * e + 1]);
            }
            *(f32x4*)(sf + x.opix[n] * (2 * C) + 2 * c) = o;
        }
    }
    conv3_ld_reduce(ld, red, ldp, tid);
}

// ---- chain part: fAffine's last conv + both couplings ------------------------------------------------------------------
// MODE_ENC  (the reference's "reverse" branch, multReverse):  z2 = z2/scale - shift, then z = z/scaleFt - shiftFt
// MODE_DEC_A:                                                  z2 = (z2 + shift)*scale   (fapply_kernel ran before conv1_finish)
// The tile's slot of `ldpart` receives fAffine's -/+ sum of log(scale); fFeatures' sums were written by conv3f_kernel.
template <int NF, int NA, int MODE>
__global__ __launch_bounds__(256, 2)
void conv3_kernel(const float* __restrict__ sfP, const float* __restrict__ haP, float* __restrict__ zP, float* __restrict__ ldpart, int ld_slots,
                  int slot0, const float* __restrict__ prm, IflStepOff off, IflGeom g, int C, float eps)
{
    constexpr int LDP = (IFL_PP_MAX * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[LDP * 32 * IFL_CS3];
    __shared__ float red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    float* ldp = ldpart + (int64_t)b * ld_slots + slot0 + blockIdx.x;
    const IflTile t = ifl_tile(g, blockIdx.x);
    if (!t.active) {
        if (tid == 0) *ldp = 0.f;
        return;
    }
    const int q = (tid & 63) >> 4;
    Conv3Idx x;
    conv3_index(x, t, g, b, tid);
    f32x4 acc[NA][2];
    const float* src = haP + (int64_t)b * (g.H + 2) * (g.W + 2) * IFL_HID + (tid & 7) * 4;
    conv3_mma<NA>(acc, src, IFL_HID, prm + off.w3[0] + (tid & 63) * 4, lds, x, t.PW, tid);

    const float* b3a = prm + off.b3[0];
    const int Ch = C / 2;
    float ld = 0.f;
    constexpr int NTILE = MODE == MODE_ENC ? NF : NA;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        const int sigma = 8 * i + 2 * q;
        const bool vA = i < NA && sigma < Ch;
        if (MODE == MODE_ENC ? sigma >= C : !vA) continue;
        const int c = MODE == MODE_ENC ? (sigma + Ch) % C : Ch + sigma;      // for i < NA both give Ch + sigma
        f32x4 bA, eA;
        if (i < NA) { bA = *(const f32x4*)(b3a + i * 32 + q * 4); eA = *(const f32x4*)(b3a + i * 32 + 16 + q * 4); }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (x.opix[n] < 0) continue;
            float* zp = zP + x.opix[n] * C + c;
            f32x2 z = *(const f32x2*)zp;
            if (vA) {
                const f32x4 a = acc[i < NA ? i : 0][n];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float sh = (a[e] + bA[e]) * eA[e];
                    float sc = scale_of((a[2 + e] + bA[2 + e]) * eA[2 + e], eps);
                    if (MODE == MODE_ENC) {
                        z[e] = z[e] / sc;
                        z[e] = z[e] - sh;
                        ld -= logf(sc);
                    } else {
                        z[e] = z[e] + sh;
                        z[e] = z[e] * sc;
                        ld += logf(sc);
                    }
                }
            }
            if (MODE == MODE_ENC) {
                const f32x4 f = *(const f32x4*)(sfP + x.opix[n] * (2 * C) + 2 * c);      // shift c, scale c, shift c+1, scale c+1
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    z[e] = z[e] / f[2 * e + 1];
                    z[e] = z[e] - f[2 * e];
                }
            }
            *(f32x2*)zp = z;
        }
    }
    conv3_ld_reduce(ld, red, ldp, tid);
}

// decode, first half of the coupling: z = (z + shiftFt) * scaleFt for every channel (Ablation.py:170-175); one thread per
// (pixel, channel pair) of the ringed plane's interior
__global__ __launch_bounds__(256)
void fapply_kernel(const float* __restrict__ sfP, float* __restrict__ zP, int B, int H, int W, int C)
{
    const int half = C / 2;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H * W * half) return;
    int cp = i % half; int64_t p = i / half;
    int xx = p % W; int64_t r = p / W; int y = r % H; int b = r / H;
    int64_t px = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + xx + 1;
    const f32x4 f = *(const f32x4*)(sfP + px * (2 * C) + 4 * cp);
    f32x2 z = *(const f32x2*)(zP + px * C + 2 * cp);
    z[0] = z[0] + f[0]; z[0] = z[0] * f[1];
    z[1] = z[1] + f[2]; z[1] = z[1] * f[3];
    *(f32x2*)(zP + px * C + 2 * cp) = z;
}

template <int NF, int NA>
int launch(const float* sf, const float* ha, float* z, float* ldpart, int ld_slots, int slot0, const float* prm, const IflStepOff& off,
           const IflGeom& g, int B, int C, int mode, float eps, hipStream_t st)
{
    dim3 grid(g.ntiles, B), block(256);
    if (mode == MODE_ENC)
        hipLaunchKernelGGL((conv3_kernel<NF, NA, MODE_ENC>), grid, block, 0, st, sf, ha, z, ldpart, ld_slots, slot0, prm, off, g, C, eps);
    else
        hipLaunchKernelGGL((conv3_kernel<NF, NA, MODE_DEC_A>), grid, block, 0, st, sf, ha, z, ldpart, ld_slots, slot0, prm, off, g, C, eps);
    IFL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// mode 0: encode (fAffine + both couplings), 2: decode (fAffine's coupling)
int ifl_k_conv3(const float* sf, const float* ha, float* z, float* ldpart, int ld_slots, int slot0, const float* prm,
                const IflStepOff& off, const IflGeom& g, int B, int C, int mode, float eps, hipStream_t st)
{
    if (mode != MODE_ENC && mode != MODE_DEC_A) { ifl_set_error("conv3: mode %d", mode); return IFL_EINVAL; }
    if (C == 12) return launch<2, 1>(sf, ha, z, ldpart, ld_slots, slot0, prm, off, g, B, C, mode, eps, st);
    if (C == 48) return launch<6, 3>(sf, ha, z, ldpart, ld_slots, slot0, prm, off, g, B, C, mode, eps, st);
    ifl_set_error("conv3: C=%d not supported (12 or 48)", C);
    return IFL_ENOTSUP;
}

// fFeatures' shift / scale of `nsteps` steps: h2all/sfall point at the first step's slab; ldpart slots slot0 + step*ntiles + tile
int ifl_k_conv3f(const float* h2all, int64_t h2_step, float* sfall, int64_t sf_step, float* ldpart, int ld_slots, int slot0,
                 const float* params, const IflStepList& steps, int nsteps, const IflStepOff& off, const IflGeom& g, int B, int C,
                 float eps, hipStream_t st)
{
    if (nsteps <= 0) return 0;
    dim3 grid(g.ntiles, B, nsteps), block(256);
    if (C == 12)
        hipLaunchKernelGGL((conv3f_kernel<2>), grid, block, 0, st, h2all, h2_step, sfall, sf_step, ldpart, ld_slots, slot0, params, steps, off, g, C, eps);
    else if (C == 48)
        hipLaunchKernelGGL((conv3f_kernel<6>), grid, block, 0, st, h2all, h2_step, sfall, sf_step, ldpart, ld_slots, slot0, params, steps, off, g, C, eps);
    else { ifl_set_error("conv3f: C=%d not supported (12 or 48)", C); return IFL_ENOTSUP; }
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_fapply(const float* sf, float* z, int B, int C, int H, int W, hipStream_t st)
{
    int64_t n = (int64_t)B * H * W * (C / 2);
    hipLaunchKernelGGL(fapply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sf, z, B, H, W, C);
    IFL_LAUNCH_CHECK();
    return 0;
}
