// pointwise.hip -- bandwidth-bound per-pixel kernels of the flow.
//
//   anic:  ActNorm2d fused with InvertibleConv1x1 (and the exact inverse pair)
//          fwd  y = W * ((x + b) .* exp(logs))            FlowActNorms.py:87-92 then Permutations.py:114-118
//          inv  x = (Winv * y) .* exp(-logs) - b          Permutations.py:119-123 then FlowActNorms.py:93-96
//          W / Winv are built on the host exactly as Permutations.py:92-100 does (inverse in float64).
//          The log-det terms (sum logs + sum log_s) * H*W are data independent and added by ld_finalize.
//   ld_finalize: per-sample log-det = input + constant + sum of the couplings' per-tile partials (float64, fixed order)
//   logp:  Gaussian.logp                                   flow.py:138-161
//   lerp:  z = wa*za + wb*zb                               flow_model.py:186-200, train.py:88-92
#include "ifl_common.h"

#include <cstdint>
#include <mutex>

// Scratch of ifl_gaussian_logp_static (the entry point that allocates nothing): the per-block partial sums that
// ifl_gaussian_logp takes from its `scratch` argument.  One copy per GPU, held until the process ends.  In host code this
// name is only a handle for hipGetSymbolAddress; kernels get the device address that call returns (g_logp_scratch_ptr).
static __device__ __align__(256) float scratch[IFL_LOGP_STATIC_MAX_FLOATS];

namespace {

// SPLIT threads per pixel of the ringed channels-last plane (adjacent lanes), each producing C / SPLIT of the outputs from all
// C inputs; W staged in LDS and read as broadcasts.  C = 48 uses 4 threads per pixel: 51 200 pixels per batch of 32 would
// otherwise be less than one wave per SIMD, each with 2 304 dependent-issue FMAs.  In place: the lanes of a pixel sit in one
// wave, whose loads all complete (they feed the arithmetic) before its stores issue.
template <int C, int SPLIT>
__global__ __launch_bounds__(256)
void anic_kernel(float* __restrict__ zr, const float* __restrict__ prm, int B, int H, int W, int inverse,
                 const float* __restrict__ sfnext)
{
    constexpr int CO = C / SPLIT;
    static_assert(CO % 4 == 0 && 256 % SPLIT == 0, "outputs per thread must be whole 16-byte words");
    __shared__ __attribute__((aligned(16))) float sw[C * C + 2 * C];
    for (int i = threadIdx.x; i < C * C + 2 * C; i += 256) sw[i] = prm[i];
    __syncthreads();
    const float* sb = sw;            // bias
    const float* ss = sw + C;        // exp(logs) or exp(-logs)
    const float* wm = sw + 2 * C;    // W[co][ci] row-major
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = t0 / SPLIT;
    const int part = (int)(t0 - p * SPLIT);
    if (p >= (int64_t)B * H * W) return;
    int x = p % W; int64_t t = p / W; int y = t % H; int b = t / H;
    const int64_t pxi = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1;
    float* px = zr + pxi * C;
    float v[C], o[CO];
#pragma unroll
    for (int c = 0; c < C; c += 4) {
        f32x4 q = *(const f32x4*)(px + c);
        v[c] = q[0]; v[c + 1] = q[1]; v[c + 2] = q[2]; v[c + 3] = q[3];
    }
    if (!inverse) {
#pragma unroll
        for (int c = 0; c < C; ++c) { v[c] = v[c] + sb[c]; v[c] = v[c] * ss[c]; }
    }
    const int co0 = part * CO;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < C; ci += 4) {
            f32x4 w = *(const f32x4*)(wm + (co0 + co) * C + ci);
            acc = fmaf(w[0], v[ci], acc); acc = fmaf(w[1], v[ci + 1], acc);
            acc = fmaf(w[2], v[ci + 2], acc); acc = fmaf(w[3], v[ci + 3], acc);
        }
        o[co] = acc;
    }
    if (inverse) {
#pragma unroll
        for (int c = 0; c < CO; ++c) { o[c] = o[c] * ss[co0 + c]; o[c] = o[c] - sb[co0 + c]; }
    }
    if (sfnext) {                    // the next layer's z = (z + shiftFt) * scaleFt (FlowAffineCouplingsAblation.py:170-175), fused
        const float* sf = sfnext + pxi * (2 * C) + 2 * co0;
#pragma unroll
        for (int c = 0; c < CO; c += 2) {
            const f32x4 f = *(const f32x4*)(sf + 2 * c);
            o[c] = o[c] + f[0]; o[c] = o[c] * f[1];
            o[c + 1] = o[c + 1] + f[2]; o[c + 1] = o[c + 1] * f[3];
        }
    }
#pragma unroll
    for (int c = 0; c < CO; c += 4) *(f32x4*)(px + co0 + c) = f32x4{o[c], o[c + 1], o[c + 2], o[c + 3]};
}

// C = 48 on the matrix cores (v_mfma_f32_16x16x4_f32): D[16 out x 16 px] += W[16 out x 4 in] X[4 in x 16 px].  A wave takes 64
// pixels as four tiles of 16; lane (n = lane % 16, g = lane / 16) loads, for j = 0..2, the 16 bytes of pixel n at channels
// 16 j + 4 g .. + 3 (the four lanes of a pixel cover 64 contiguous bytes per load), which are the B operands of k-steps
// (j, e) with k = g; the matching A fragments W[16 t + n][16 j + 4 g + e] stay in 36 registers for the whole wave.  The
// accumulator of tile t holds channels 16 t + 4 g + r of pixel n: one 16-byte store, and the same channel set as load j = t,
// so one pair of ActNorm vectors per lane serves both directions.  No LDS.  In place: a pixel's four lanes sit in one wave.
__device__ __forceinline__ f32x4 mfma16p(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(256)
void anic48_mfma_kernel(float* __restrict__ zr, const float* __restrict__ prm, int B, int H, int W, int inverse,
                        const float* __restrict__ sfnext)
{
    constexpr int C = 48;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const float* wm = prm + 2 * C;
    f32x4 a[3][3], bb[3], sc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        bb[t] = *(const f32x4*)(prm + 16 * t + 4 * g);
        sc[t] = *(const f32x4*)(prm + C + 16 * t + 4 * g);
#pragma unroll
        for (int j = 0; j < 3; ++j) a[t][j] = *(const f32x4*)(wm + (16 * t + n) * C + 16 * j + 4 * g);
    }
    const int64_t total = (int64_t)B * H * W;
    const int64_t p0 = (int64_t)blockIdx.x * 256 + (threadIdx.x >> 6) * 64;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int64_t p = p0 + pt * 16 + n;
        const bool live = p < total;
        const int64_t pc = live ? p : total - 1;
        const int x = pc % W; const int64_t q = pc / W; const int y = q % H; const int b = q / H;
        const int64_t pxi = ((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1;
        float* px = zr + pxi * C + 4 * g;
        f32x4 v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = *(const f32x4*)(px + 16 * j);
        if (!inverse) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[j] = v[j] + bb[j]; v[j] = v[j] * sc[j]; }
        }
        f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = mfma16p(a[t][j][e], v[j][e], acc[t]);
        }
        if (inverse) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { acc[t] = acc[t] * sc[t]; acc[t] = acc[t] - bb[t]; }
        }
        if (sfnext) {                // the next layer's z = (z + shiftFt) * scaleFt, fused
            const float* sf = sfnext + pxi * (2 * C) + 8 * g;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const f32x4 f0 = *(const f32x4*)(sf + 32 * t), f1 = *(const f32x4*)(sf + 32 * t + 4);
                acc[t][0] = acc[t][0] + f0[0]; acc[t][0] = acc[t][0] * f0[1];
                acc[t][1] = acc[t][1] + f0[2]; acc[t][1] = acc[t][1] * f0[3];
                acc[t][2] = acc[t][2] + f1[0]; acc[t][2] = acc[t][2] * f1[1];
                acc[t][3] = acc[t][3] + f1[2]; acc[t][3] = acc[t][3] * f1[3];
            }
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < 3; ++t) *(f32x4*)(px + 16 * t) = acc[t];
        }
    }
}

__global__ __launch_bounds__(64)
void ld_finalize_kernel(const float* __restrict__ ldpart, int used, int ld_slots, const float* __restrict__ ld_in,
                        float* __restrict__ ld_out, double constant, double fsign, int fa0, int fan, int fb0, int fbn)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* row = ldpart + (int64_t)b * ld_slots;
    double s = 0.0, f = 0.0;
    for (int i = lane; i < used; i += 64) s += (double)row[i];
    for (int i = lane; i < fan; i += 64) f += (double)row[fa0 + i];
    for (int i = lane; i < fbn; i += 64) f += (double)row[fb0 + i];
    s += fsign * f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) ld_out[b] = (float)((double)(ld_in ? ld_in[b] : 0.f) + constant + s);
}

// Gaussian log-density of every pixel's C-vector, summed over the pixels of a block -> scratch[b][block].
// z is NCHW: lane = pixel, so every channel read is coalesced.  CT = compile-time C (registers) or 0 (any C <= 64,
// re-reading z through the cache).
constexpr int LOGP_PIX = 256;
template <int CT>
__global__ __launch_bounds__(256)
void logp_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ cov_inv,
                 float logdet_cov, float* __restrict__ scratch, int nblk, int Crt, int HW)
{
    __shared__ float sci[64 * 64 + 64];
    __shared__ float red[4];
    const int C = CT ? CT : Crt;
    const int b = blockIdx.y;
    if (cov_inv) {
        for (int i = threadIdx.x; i < C * C; i += 256) sci[i] = cov_inv[i];
        for (int i = threadIdx.x; i < C; i += 256) sci[64 * 64 + i] = mean[i];
    }
    __syncthreads();
    const int p = blockIdx.x * LOGP_PIX + threadIdx.x;
    float lik = 0.f;
    if (p < HW) {
        const float* zp = z + (int64_t)b * C * HW + p;
        float q = 0.f;
        if (!cov_inv) {
            for (int c = 0; c < C; ++c) { float v = zp[(int64_t)c * HW]; q = fmaf(v, v, q); }
            logdet_cov = 0.f;
        } else if (CT) {
            float x[CT ? CT : 1];
#pragma unroll
            for (int c = 0; c < CT; ++c) x[c] = zp[(int64_t)c * HW] - sci[64 * 64 + c];
            // sum_c ( sum_k x[k] * cov_inv[k][c] ) * x[c]      (flow.py:156: (x @ cov_inv) * x)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < CT; ++k) t = fmaf(x[k], sci[k * CT + c], t);
                q = fmaf(t, x[c], q);
            }
        } else {
            for (int c = 0; c < C; ++c) {
                float t = 0.f;
                for (int k = 0; k < C; ++k) t = fmaf(zp[(int64_t)k * HW] - sci[64 * 64 + k], sci[k * C + c], t);
                q = fmaf(t, zp[(int64_t)c * HW] - sci[64 * 64 + c], q);
            }
        }
        lik = -0.5f * (logdet_cov + (float)C * 1.8378770664093453f + q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lik += __shfl_xor(lik, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lik;
    __syncthreads();
    if (threadIdx.x == 0) scratch[(int64_t)b * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// C = 48 with a full covariance on the matrix cores: the per-pixel vector-matrix product (x - mean) @ cov_inv (flow.py:156) is
// 2 304 multiply-adds per 192 bytes read -- on the vector ALUs the kernel above runs at a third of the HBM rate.  Here a wave
// takes its 64 pixels as four tiles of 16 (anic48_mfma_kernel's operand layout): lane (n = lane % 16, g = lane / 16) loads
// x[16 j + 4 g + e] of pixel n (NCHW: 16 consecutive pixels of a channel per 64-byte segment), the A fragments
// cov_inv[16 j + 4 g + e][16 t + n] stay in 36 registers, accumulator t holds y[16 t + 4 g + r] of pixel n -- the same channels
// the lane loaded, so the quadratic form is a register dot product; the lanes' shares are summed with the block's pixels.
// VEC (H*W a multiple of 4): tile pt takes the pixels 4 n + pt of the wave's 64, so a lane's four tiles are ONE 16-byte load per
// channel (16 lanes = 256 contiguous bytes) instead of four 4-byte loads that each use half a cache line.
template <bool VEC>
__global__ __launch_bounds__(256)
void logp48_mfma_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ cov_inv,
                        float logdet_cov, float* __restrict__ scratch, int nblk, int HW)
{
    constexpr int C = 48;
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    f32x4 a[3][3], mu[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[j][e] = mean[16 * j + 4 * g + e];      // (element loads: the header promises no alignment for mean)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[t][j][e] = cov_inv[(16 * j + 4 * g + e) * C + 16 * t + n];
    }
    const float cst = logdet_cov + (float)C * 1.8378770664093453f;
    float lik = 0.f;
    f32x4 xq[3][4];                           // VEC: [j][e] = the channel's values at the lane's four pixels
    if (VEC) {
        const int p0 = blockIdx.x * LOGP_PIX + wave * 64 + 4 * n;
        const float* zp = z + (int64_t)b * C * HW + (p0 < HW ? p0 : 0);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[j][e] = *(const f32x4*)(zp + (int64_t)(16 * j + 4 * g + e) * HW);
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int p = blockIdx.x * LOGP_PIX + wave * 64 + (VEC ? 4 * n + pt : pt * 16 + n);
        const bool live = p < HW;
        const float* zp = z + (int64_t)b * C * HW + (live ? p : 0);
        f32x4 x[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = (VEC ? xq[j][e][pt] : zp[(int64_t)(16 * j + 4 * g + e) * HW]) - mu[j][e];
                x[j][e] = live ? v : 0.f;
            }
        float q = 0.f;
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma16p(a[t][j][e], x[j][e], acc[t]);      // three independent chains interleaved
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) q = fmaf(acc[t][r], x[t][r], q);
        lik += -0.5f * q;
        if (live && g == 0) lik += -0.5f * cst;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lik += __shfl_xor(lik, o, 64);
    if (lane == 0) red[wave] = lik;
    __syncthreads();
    if (threadIdx.x == 0) scratch[(int64_t)b * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out = wa[b]*za + wb[b]*zb: two rounded products and one rounded sum, as torch evaluates (1-a)*z1 + a*z2
__global__ __launch_bounds__(256)
void lerp_kernel(const float* __restrict__ za, const float* __restrict__ zb, const float* __restrict__ wa,
                 const float* __restrict__ wb, float* __restrict__ out, int64_t n, int vec)
{
    const int b = blockIdx.y;
    const float a = wa[b], c = wb[b];
    const int64_t base = (int64_t)b * n;
    if (vec) {
        const f32x4* pa = (const f32x4*)(za + base);
        const f32x4* pb = (const f32x4*)(zb + base);
        f32x4* po = (f32x4*)(out + base);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
            f32x4 u = pa[i], v = pb[i], r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __fadd_rn(__fmul_rn(a, u[e]), __fmul_rn(c, v[e]));
            po[i] = r;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
            out[base + i] = __fadd_rn(__fmul_rn(a, za[base + i]), __fmul_rn(c, zb[base + i]));
    }
}

}  // namespace

int ifl_k_anic(float* zr, const float* prm, int B, int C, int H, int W, int inverse, const float* sf_next, hipStream_t st)
{
    int64_t n = (int64_t)B * H * W;
    dim3 block(256);
    if (C == 12) hipLaunchKernelGGL((anic_kernel<12, 1>), dim3((unsigned)((n + 255) / 256)), block, 0, st, zr, prm, B, H, W, inverse, sf_next);
    else if (C == 48) hipLaunchKernelGGL(anic48_mfma_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, st, zr, prm, B, H, W, inverse, sf_next);
    else { ifl_set_error("actnorm+invconv: C=%d not supported (12 or 48)", C); return IFL_ENOTSUP; }
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_ld_finalize(const float* ldpart, int slots_used, int ld_slots, const float* ld_in, float* ld_out,
                      double constant, double fsign, int fa0, int fan, int fb0, int fbn, int B, hipStream_t st)
{
    hipLaunchKernelGGL(ld_finalize_kernel, dim3(B), dim3(64), 0, st, ldpart, slots_used, ld_slots, ld_in, ld_out, constant,
                       fsign, fa0, fan, fb0, fbn);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_logp(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* scratch,
               float* out, int B, int C, int HW, hipStream_t st)
{
    int nblk = ifl_cdiv(HW, LOGP_PIX);
    dim3 grid(nblk, B), block(256);
    if (C == 48 && cov_inv && HW % 4 == 0 && ((uintptr_t)z & 15) == 0) hipLaunchKernelGGL(logp48_mfma_kernel<true>, grid, block, 0, st, z, mean, cov_inv, logdet_cov, scratch, nblk, HW);
    else if (C == 48 && cov_inv) hipLaunchKernelGGL(logp48_mfma_kernel<false>, grid, block, 0, st, z, mean, cov_inv, logdet_cov, scratch, nblk, HW);
    else if (C == 48) hipLaunchKernelGGL((logp_kernel<48>), grid, block, 0, st, z, mean, cov_inv, logdet_cov, scratch, nblk, C, HW);
    else if (C == 12) hipLaunchKernelGGL((logp_kernel<12>), grid, block, 0, st, z, mean, cov_inv, logdet_cov, scratch, nblk, C, HW);
    else hipLaunchKernelGGL((logp_kernel<0>), grid, block, 0, st, z, mean, cov_inv, logdet_cov, scratch, nblk, C, HW);
    IFL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ld_finalize_kernel, dim3(B), dim3(64), 0, st, scratch, nblk, nblk, (const float*)nullptr, out, 0.0, 0.0, 0, 0, 0, 0);
    IFL_LAUNCH_CHECK();
    return 0;
}

int64_t ifl_logp_scratch_floats(int HW) { return ifl_cdiv(HW, LOGP_PIX); }

// Device address of this GPU's copy of `scratch`, looked up once per device with that device current.
namespace {
std::mutex g_logp_scratch_mu;
float* g_logp_scratch_ptr[IFL_MAX_DEVICES] = {};
}  // namespace

int ifl_k_logp_static(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* out, int B, int C,
                      int HW, hipStream_t st)
{
    const int64_t need = (int64_t)B * ifl_logp_scratch_floats(HW);
    IFL_CHECK_ARG(need <= IFL_LOGP_STATIC_MAX_FLOATS, IFL_ENOTSUP,
                  "ifl_gaussian_logp_static: B * ceil(H*W/%d) = %lld partial sums exceed the static scratch of %d floats; "
                  "use ifl_gaussian_logp with a caller-owned scratch", LOGP_PIX, (long long)need, IFL_LOGP_STATIC_MAX_FLOATS);
    int dev = 0;
    IFL_HIP(hipGetDevice(&dev));
    IFL_CHECK_ARG(dev >= 0 && dev < IFL_MAX_DEVICES, IFL_ENOTSUP, "device index %d", dev);
    float* ptr = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_logp_scratch_mu);
        if (!g_logp_scratch_ptr[dev]) {
            void* p = nullptr;
            IFL_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            IFL_CHECK_ARG(p != nullptr, IFL_ENOTSUP, "hipGetSymbolAddress returned a null address for the static scratch");
            g_logp_scratch_ptr[dev] = (float*)p;
        }
        ptr = g_logp_scratch_ptr[dev];
    }
    // every slot the finalize kernel reads is written by this call's logp kernel first: nothing has to start at zero
    return ifl_k_logp(z, mean, cov_inv, logdet_cov, ptr, out, B, C, HW, st);
}

int ifl_k_lerp(const float* za, const float* zb, const float* wa, const float* wb, float* out, int B, int64_t n,
               hipStream_t st)
{
    if (n == 0 || B == 0) return 0;
    int vec = (n % 4 == 0) ? 1 : 0;     // per-sample rows stay 16-byte aligned only then
    int64_t work = vec ? n / 4 : n;
    int gx = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
    hipLaunchKernelGGL(lerp_kernel, dim3(gx, B), dim3(256), 0, st, za, zb, wa, wb, out, n, vec);
    IFL_LAUNCH_CHECK();
    return 0;
}
