// layout.hip -- index permutations (bit-exact): squeeze / unsqueeze and the NCHW <-> ringed channels-last
// conversions at the boundary of the fused flow.
//
//   squeeze2d / unsqueeze2d          reference interflow/models/modules/flow.py:163-193
//       out[b, c*f*f + i*f + j, h, w] = in[b, c, h*f + i, w*f + j]
//   "ringed NHWC": [B][H+2][W+2][C] float32 with a one-pixel ring of zeros -- the 3x3 convs of the coupling
//   networks read their 'same' zero padding (flow.py:27) from the ring instead of testing bounds.
#include "ifl_common.h"

namespace {

// ---- NCHW [B,C,P=H*W] -> ringed NHWC, 32x32 tile transpose through LDS ------------------------------------------
__global__ __launch_bounds__(256)
void nchw_to_ring_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W)
{
    __shared__ float tile[32][33];
    const int P = H * W;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
    const int i = threadIdx.x & 31, jj = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = c0 + jj + 8 * k, p = p0 + i;
        tile[jj + 8 * k][i] = (c < C && p < P) ? in[((int64_t)b * C + c) * P + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = p0 + jj + 8 * k, c = c0 + i;
        if (p < P && c < C) {
            int y = p / W, x = p - y * W;
            out[(((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c] = tile[i][jj + 8 * k];
        }
    }
}

__global__ __launch_bounds__(256)
void ring_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W)
{
    __shared__ float tile[32][33];
    const int P = H * W;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
    const int i = threadIdx.x & 31, jj = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = p0 + jj + 8 * k, c = c0 + i;
        float v = 0.f;
        if (p < P && c < C) {
            int y = p / W, x = p - y * W;
            v = in[(((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c];
        }
        tile[jj + 8 * k][i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = c0 + jj + 8 * k, p = p0 + i;
        if (c < C && p < P) out[((int64_t)b * C + c) * P + p] = tile[i][jj + 8 * k];
    }
}

// ---- x NCHW [B,C,H,W] -> squeezed ringed NHWC [B][H/2+2][W/2+2][4C]; one thread per output pixel ------------------
// out[b,h,w,4c+2i+j] = x[b,c,2h+i,2w+j]
__global__ __launch_bounds__(256)
void squeeze_in_kernel(const float* __restrict__ x, float* __restrict__ zr, int B, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)B * Ho * Wo) return;
    int w = p % Wo; int64_t t = p / Wo; int h = t % Ho; int b = t / Ho;
    float* dst = zr + (((int64_t)b * (Ho + 2) + h + 1) * (Wo + 2) + w + 1) * (4 * C);
    for (int c = 0; c < C; ++c) {
        const float* src = x + (((int64_t)b * C + c) * H + 2 * h) * W + 2 * w;
        f32x2 r0 = *(const f32x2*)src, r1 = *(const f32x2*)(src + W);
        dst[4 * c + 0] = r0[0]; dst[4 * c + 1] = r0[1]; dst[4 * c + 2] = r1[0]; dst[4 * c + 3] = r1[1];
    }
}

__global__ __launch_bounds__(256)
void unsqueeze_out_kernel(const float* __restrict__ zr, float* __restrict__ x, int B, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;       // zr plane; x is [B,C,H,W]
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)B * Ho * Wo) return;
    int w = p % Wo; int64_t t = p / Wo; int h = t % Ho; int b = t / Ho;
    const float* src = zr + (((int64_t)b * (Ho + 2) + h + 1) * (Wo + 2) + w + 1) * (4 * C);
    for (int c = 0; c < C; ++c) {
        float* dst = x + (((int64_t)b * C + c) * H + 2 * h) * W + 2 * w;
        *(f32x2*)dst = f32x2{src[4 * c + 0], src[4 * c + 1]};
        *(f32x2*)(dst + W) = f32x2{src[4 * c + 2], src[4 * c + 3]};
    }
}

// ---- ringed [H,W,C] <-> ringed [H/2,W/2,4C]; one thread per (output pixel, input channel) ------------------------
__global__ __launch_bounds__(256)
void squeeze_ring_kernel(const float* __restrict__ a, float* __restrict__ o, int B, int C, int H, int W, int reverse)
{
    const int Ho = H / 2, Wo = W / 2;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * Ho * Wo * C) return;
    int c = e % C; int64_t t = e / C; int w = t % Wo; t /= Wo; int h = t % Ho; int b = t / Ho;
    int64_t big = (((int64_t)b * (H + 2) + 2 * h + 1) * (W + 2) + 2 * w + 1) * C + c;       // pixel (2h, 2w) of the large plane
    int64_t sml = ((((int64_t)b * (Ho + 2) + h + 1) * (Wo + 2) + w + 1) * C + c) * 4;       // channel 4c of the small plane
    const int64_t row = (int64_t)(W + 2) * C;
    if (!reverse) {
        f32x4 v = {a[big], a[big + C], a[big + row], a[big + row + C]};
        *(f32x4*)(o + sml) = v;
    } else {      // a = small plane (input), o = large plane (output)
        f32x4 v = *(const f32x4*)(a + sml);
        o[big] = v[0]; o[big + C] = v[1]; o[big + row] = v[2]; o[big + row + C] = v[3];
    }
}

// ---- stand-alone NCHW squeeze2d / unsqueeze2d (any factor); one thread per element of the squeezed tensor --------
__global__ __launch_bounds__(256)
void squeeze_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W, int f, int reverse)
{
    // squeezed tensor S [B, C*f*f, H/f, W/f]; large tensor L [B, C, H, W]
    const int Ho = H / f, Wo = W / f;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * C * H * W) return;
    int w = e % Wo; int64_t t = e / Wo; int h = t % Ho; t /= Ho; int cs = t % (C * f * f); int b = t / (C * f * f);
    int c = cs / (f * f), ij = cs - c * f * f, i = ij / f, j = ij - i * f;
    int64_t l = (((int64_t)b * C + c) * H + h * f + i) * W + w * f + j;
    if (!reverse) out[e] = in[l]; else out[l] = in[e];
}

// ---- image quantisation for the generation loop: (clip(x, 0, 1) * 255) -> uint8, NCHW -> NHWC ---------------------------
// generate_images.py:89-91: `(np.clip(output, 0., 1.0)*255).astype(np.uint8)` then `.transpose(0,2,3,1)`.
// One thread per 4 consecutive pixels: C float4 loads (coalesced per channel plane), 4*C contiguous output bytes.
template <int C>
__global__ __launch_bounds__(256)
void quantize_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t P, int64_t nquads)
{
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * C * P;
    uint8_t* ob = out + (int64_t)b * C * P;
    for (int64_t qd = (int64_t)blockIdx.x * 256 + threadIdx.x; qd < nquads; qd += (int64_t)gridDim.x * 256) {
        const int64_t p0 = qd * 4;
        uint8_t v[4 * C];
        if (p0 + 4 <= P && (P & 3) == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                f32x4 f = *(const f32x4*)(xb + c * P + p0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e * C + c] = (uint8_t)(fminf(fmaxf(f[e], 0.f), 1.f) * 255.f);
            }
#pragma unroll
            for (int i = 0; i < 4 * C; ++i) ob[p0 * C + i] = v[i];
        } else {
            for (int64_t p = p0; p < P && p < p0 + 4; ++p)
                for (int c = 0; c < C; ++c) ob[p * C + c] = (uint8_t)(fminf(fmaxf(xb[c * P + p], 0.f), 1.f) * 255.f);
        }
    }
}

// factor-2 fast path (the only factor the model uses), W % 8 == 0: one thread per 4 consecutive squeezed pixels of one
// input channel -- two rows of 8 input floats (float4 x 4) <-> four channel planes x float4, every access 16 bytes.
__global__ __launch_bounds__(256)
void squeeze2_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t nthreads, int C, int H, int W, int reverse)
{
    const int Ho = H / 2, Wo = W / 2, Wq = Wo / 4;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nthreads) return;
    int wq = e % Wq; int64_t t = e / Wq; int h = t % Ho; t /= Ho; int c = t % C; int b = t / C;
    const int64_t l0 = (((int64_t)b * C + c) * H + 2 * h) * W + 8 * wq;                   // large tensor, row 2h
    const int64_t s0 = (((int64_t)b * C * 4 + 4 * c) * Ho + h) * Wo + 4 * wq;             // squeezed tensor, channel 4c
    const int64_t plane = (int64_t)Ho * Wo;
    if (!reverse) {
        f32x4 r0a = *(const f32x4*)(in + l0), r0b = *(const f32x4*)(in + l0 + 4);
        f32x4 r1a = *(const f32x4*)(in + l0 + W), r1b = *(const f32x4*)(in + l0 + W + 4);
        *(f32x4*)(out + s0) = f32x4{r0a[0], r0a[2], r0b[0], r0b[2]};                      // (i, j) = (0, 0)
        *(f32x4*)(out + s0 + plane) = f32x4{r0a[1], r0a[3], r0b[1], r0b[3]};              // (0, 1)
        *(f32x4*)(out + s0 + 2 * plane) = f32x4{r1a[0], r1a[2], r1b[0], r1b[2]};          // (1, 0)
        *(f32x4*)(out + s0 + 3 * plane) = f32x4{r1a[1], r1a[3], r1b[1], r1b[3]};          // (1, 1)
    } else {      // in = squeezed, out = large
        f32x4 a = *(const f32x4*)(in + s0), bq = *(const f32x4*)(in + s0 + plane);
        f32x4 cq = *(const f32x4*)(in + s0 + 2 * plane), d = *(const f32x4*)(in + s0 + 3 * plane);
        *(f32x4*)(out + l0) = f32x4{a[0], bq[0], a[1], bq[1]};
        *(f32x4*)(out + l0 + 4) = f32x4{a[2], bq[2], a[3], bq[3]};
        *(f32x4*)(out + l0 + W) = f32x4{cq[0], d[0], cq[1], d[1]};
        *(f32x4*)(out + l0 + W + 4) = f32x4{cq[2], d[2], cq[3], d[3]};
    }
}

}  // namespace

int ifl_k_quantize_u8(const float* x, uint8_t* out, int B, int C, int64_t P, hipStream_t st)
{
    if (B == 0 || P == 0) return 0;
    int64_t nquads = (P + 3) / 4;
    int gx = (int)((nquads + 255) / 256 < 4096 ? (nquads + 255) / 256 : 4096);
    dim3 grid(gx, B), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(quantize_u8_kernel<1>, grid, block, 0, st, x, out, P, nquads); break;
        case 3: hipLaunchKernelGGL(quantize_u8_kernel<3>, grid, block, 0, st, x, out, P, nquads); break;
        case 4: hipLaunchKernelGGL(quantize_u8_kernel<4>, grid, block, 0, st, x, out, P, nquads); break;
        default: ifl_set_error("quantize_u8: C=%d not supported (1, 3 or 4)", C); return IFL_ENOTSUP;
    }
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_nchw_to_ring(const float* in, float* out, int B, int C, int H, int W, hipStream_t st)
{
    dim3 grid(ifl_cdiv(H * W, 32), ifl_cdiv(C, 32), B);
    hipLaunchKernelGGL(nchw_to_ring_kernel, grid, dim3(256), 0, st, in, out, C, H, W);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_ring_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t st)
{
    dim3 grid(ifl_cdiv(H * W, 32), ifl_cdiv(C, 32), B);
    hipLaunchKernelGGL(ring_to_nchw_kernel, grid, dim3(256), 0, st, in, out, C, H, W);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_squeeze_in(const float* x, float* zr, int B, int C, int H, int W, hipStream_t st)
{
    int64_t n = (int64_t)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(squeeze_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, zr, B, C, H, W);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_unsqueeze_out(const float* zr, float* x, int B, int C, int H, int W, hipStream_t st)
{
    int64_t n = (int64_t)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(unsqueeze_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, zr, x, B, C, H, W);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_squeeze_ring(const float* a, float* b, int B, int C, int H, int W, hipStream_t st)
{
    int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(squeeze_ring_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, B, C, H, W, 0);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_unsqueeze_ring(const float* b, float* a, int B, int C, int H, int W, hipStream_t st)
{
    int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(squeeze_ring_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, a, B, C, H, W, 1);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_squeeze_nchw(const float* in, float* out, int B, int C, int H, int W, int f, int reverse, hipStream_t st)
{
    int64_t n = (int64_t)B * C * H * W;
    if (n == 0) return 0;
    if (f == 2 && W % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        int64_t nt = n / 16;
        hipLaunchKernelGGL(squeeze2_vec_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, in, out, nt, C, H, W, reverse);
        IFL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(squeeze_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, B, C, H, W, f, reverse);
    IFL_LAUNCH_CHECK();
    return 0;
}

// ---- separable gather-resize (ResizeRight's antialiased cubic of the LR-consistency loss, utils/resize_right.py:32-123) ---------
// out[n][oy][ox] = sum_j wH[oy][j] * sum_i wW[ox][i] * x[n][idxH[oy][j]][idxW[ox][i]], dimension -2 first as the reference
// (resize_right.py:97-116 walks the dims in order): per output index a field of view of T taps (left-aligned, 16 for x1/4) and
// its normalised weights, built on the host with the reference's formulas; taps on the zero padding carry weight 0 and a
// clamped index.  The same kernel evaluates the adjoint (the backward pass) from the transposed tables.  Block = one plane x
// RS_ROWS output rows: the vertical pass writes RS_ROWS x Win values to LDS, the horizontal pass reads them.
namespace {
constexpr int RS_ROWS = 4;
__global__ __launch_bounds__(256)
void resize_sep_kernel(const float* __restrict__ x, float* __restrict__ out, int Hin, int Win, int Hout, int Wout,
                       const int* __restrict__ idxH, const float* __restrict__ wH, int TH,
                       const int* __restrict__ idxW, const float* __restrict__ wW, int TW)
{
    extern __shared__ float tmp[];            // [RS_ROWS][Win]
    const int n = blockIdx.y, oy0 = blockIdx.x * RS_ROWS;
    const float* xp = x + (int64_t)n * Hin * Win;
    for (int k = threadIdx.x; k < RS_ROWS * Win; k += 256) {
        const int r = k / Win, xx = k - r * Win, oy = oy0 + r;
        float acc = 0.f;
        if (oy < Hout)
            for (int j = 0; j < TH; ++j) acc += wH[oy * TH + j] * xp[(int64_t)idxH[oy * TH + j] * Win + xx];
        tmp[k] = acc;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < RS_ROWS * Wout; k += 256) {
        const int r = k / Wout, ox = k - r * Wout, oy = oy0 + r;
        if (oy >= Hout) continue;
        float acc = 0.f;
        for (int i = 0; i < TW; ++i) acc += wW[ox * TW + i] * tmp[r * Win + idxW[ox * TW + i]];
        out[((int64_t)n * Hout + oy) * Wout + ox] = acc;
    }
}
}  // namespace

int ifl_k_resize_sep(const float* x, float* out, int N, int Hin, int Win, int Hout, int Wout, const int* idxH, const float* wH, int TH,
                     const int* idxW, const float* wW, int TW, hipStream_t st)
{
    if (N == 0 || Hout == 0 || Wout == 0) return 0;
    dim3 grid((Hout + RS_ROWS - 1) / RS_ROWS, N), block(256);
    hipLaunchKernelGGL(resize_sep_kernel, grid, block, RS_ROWS * Win * sizeof(float), st, x, out, Hin, Win, Hout, Wout, idxH, wH, TH, idxW, wW, TW);
    IFL_LAUNCH_CHECK();
    return 0;
}
