// train_glue.hip -- the per-pixel operators of the coupling networks for the TRAINING path, forward and backward, as fused
// bandwidth-bound kernels on channels-last tensors (the convolutions themselves are convg.hip's, forward / data gradient /
// weight gradient).  They replace chains of ATen element-wise launches and their autograd graphs:
//
//   an_relu      y = relu((x + b) * exp(logs))                                flow.Conv2d's ActNorm (flow.py:60-64,
//                                                                              FlowActNorms.py:87-92) followed by the ReLU of
//                                                                              FlowAffineCouplingsAblation.py:250,252
//   coupling     hs = h * exp(3 logs3); shift = hs[0::2], scale = sigmoid(hs[1::2] + 2) + eps      flow.py:80-82, Ablation.py:219-240
//                branch 0 (reference "forward"):  z' = (z + shift) * scale,  logdet += sum log scale               :170-185
//                branch 1 (reference "reverse"):  z' = z / scale - shift,    logdet -= sum log scale               :192-204
//                applied to the channel range [c0, c0 + n) of z (the other channels are copied)
//
// Backward kernels recompute what they need from the forward inputs, write the input gradients, and reduce the parameter
// gradients (per channel, over all pixels) and the per-sample log-det sums to per-block partial sums that the caller adds
// (fixed order: bit-reproducible).
#include "ifl_common.h"
#include "../../include/interflow_hip.h"

namespace {

constexpr int GPX = 64;                       // pixels per block of the an_relu kernels (256 threads = 16 channel quads x 16 pixel lanes)

__device__ __forceinline__ int64_t ring_px(int64_t p, int h, int w) {
    const int x = p % w; const int64_t t = p / w; const int y = t % h; const int64_t b = t / h;
    return (b * (h + 2) + y + 1) * (w + 2) + x + 1;
}

// x, y ringed [B][h+2][w+2][C], C = 64 (x may be a 64-channel range of a wider buffer: pixel stride xs): thread = (pixel lane, channel quad)
__global__ __launch_bounds__(256)
void an_relu_fwd_kernel(const float* __restrict__ x, int xs, float* __restrict__ y, const float* __restrict__ bias,
                        const float* __restrict__ logs, int64_t npx, int h, int w, int C, int64_t xz, int64_t yz, int64_t pz)
{
    x += blockIdx.y * xz; y += blockIdx.y * yz; bias += blockIdx.y * pz; logs += blockIdx.y * pz;     // batched: problem blockIdx.y
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const f32x4 b = *(const f32x4*)(bias + cq * 4);
    f32x4 s = *(const f32x4*)(logs + cq * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = expf(s[e]);
    for (int64_t p = (int64_t)blockIdx.x * GPX + pl; p < npx && p < (int64_t)(blockIdx.x + 1) * GPX; p += 16) {
        const int64_t rp = ring_px(p, h, w), o = rp * C + cq * 4;
        f32x4 v = *(const f32x4*)(x + rp * xs + cq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float u = (v[e] + b[e]) * s[e]; v[e] = u > 0.f ? u : 0.f; }
        *(f32x4*)(y + o) = v;
    }
}

// dx = dy * [u > 0] * s;  part[block][0:C] = sum dx (= d bias),  part[block][C:2C] = sum dy [u > 0] u (= d logs)
__global__ __launch_bounds__(256)
void an_relu_bwd_kernel(const float* __restrict__ x, int xs, const float* __restrict__ dy, float* __restrict__ dx,
                        const float* __restrict__ bias, const float* __restrict__ logs, float* __restrict__ part,
                        int64_t npx, int h, int w, int C, int64_t xz, int64_t gz, int64_t dxz, int64_t pz, int64_t partz)
{
    x += blockIdx.y * xz; dy += blockIdx.y * gz; dx += blockIdx.y * dxz; bias += blockIdx.y * pz; logs += blockIdx.y * pz;
    part += blockIdx.y * partz;
    __shared__ float red[16][2][64];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const f32x4 b = *(const f32x4*)(bias + cq * 4);
    f32x4 s = *(const f32x4*)(logs + cq * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = expf(s[e]);
    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sl = sb;
    for (int64_t p = (int64_t)blockIdx.x * GPX + pl; p < npx && p < (int64_t)(blockIdx.x + 1) * GPX; p += 16) {
        const int64_t rp = ring_px(p, h, w), o = rp * C + cq * 4;
        const f32x4 v = *(const f32x4*)(x + rp * xs + cq * 4), g = *(const f32x4*)(dy + o);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = (v[e] + b[e]) * s[e];
            const float du = u > 0.f ? g[e] : 0.f;
            d[e] = du * s[e];
            sb[e] += d[e];
            sl[e] += du * u;
        }
        *(f32x4*)(dx + o) = d;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[pl][0][cq * 4 + e] = sb[e]; red[pl][1][cq * 4 + e] = sl[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][k][c];
        part[(int64_t)blockIdx.x * 128 + k * 64 + c] = t;
    }
}

__device__ __forceinline__ float sigmoid2(float p) { return 1.0f / (1.0f + expf(-(p + 2.0f))); }

constexpr int CPX = 32;                       // pixels per block of the coupling kernels

// Thread = (pixel row r, channel c): the R = 256 / C rows of a block walk its CPX pixels, lanes run along the channels
// (consecutive addresses in z; (shift, scale) pairs of 8 bytes in h), and a thread's channel -- hence its exp(3 logs3) -- is
// fixed.  (One thread per pixel with a loop over the channels, the first form of these kernels, ran a group of four 40x40
// samples on 28 workgroups: 20-30 us each, four of them on every flow step's critical path.)
//
// z, zout [B][h][w][C] contiguous; hr ringed [B][h+2][w+2][hs] with rows (2 i, 2 i + 1) = (shift, scale) of channel c0 + i;
// ldpart[b][blk] = this block's signed sum of log scale (blocks never straddle samples: grid (ceil(h w / CPX), B))
__global__ __launch_bounds__(256)
void coupling_fwd_kernel(const float* __restrict__ z, float* __restrict__ zout, const float* __restrict__ hr, int hs,
                         const float* __restrict__ logs3, float eps, int branch, int c0, int n, float* __restrict__ ldpart,
                         int h, int w, int C)
{
    __shared__ float red[4];
    const int b = blockIdx.y, hw = h * w;
    const int R = 256 / C;
    const int r = threadIdx.x / C, c = threadIdx.x - r * C;
    const bool coupled = r < R && c >= c0 && c < c0 + n;
    const int i = c - c0;
    float e0 = 0.f, e1 = 0.f;
    if (coupled) { e0 = expf(logs3[2 * i] * 3.0f); e1 = expf(logs3[2 * i + 1] * 3.0f); }
    float ld = 0.f;
    if (r < R) {
        const int pend = min((int)(blockIdx.x + 1) * CPX, hw);
        for (int p = blockIdx.x * CPX + r; p < pend; p += R) {
            const int64_t o = ((int64_t)b * hw + p) * C + c;
            float v = z[o];
            if (coupled) {
                const int y = p / w, x = p - y * w;
                const float2 hh = *(const float2*)(hr + (((int64_t)b * (h + 2) + y + 1) * (w + 2) + x + 1) * hs + 2 * i);
                const float sh = hh.x * e0;
                const float sc = sigmoid2(hh.y * e1) + eps;
                if (branch == 0) { v = v + sh; v = v * sc; ld += logf(sc); }
                else { v = v / sc; v = v - sh; ld -= logf(sc); }
            }
            zout[o] = v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ld += __shfl_xor(ld, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ld;
    __syncthreads();
    if (threadIdx.x == 0) ldpart[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dz (all channels), dh (interior pixels, rows [0, 2n) of the ringed gradient buffer, the rest must be zero already),
// lpart[b * nblk + blk][2n] = per-block sums of d(hs) * hs * 3 (= d logs3); bpart (optional), same shape = per-block sums of
// d h (= the gradient of Conv2dZeros' bias)
__global__ __launch_bounds__(256)
void coupling_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dzo, const float* __restrict__ dld,
                         const float* __restrict__ hr, int hs, const float* __restrict__ logs3, float eps, int branch, int c0,
                         int n, float* __restrict__ dz, float* __restrict__ dh, float* __restrict__ lpart,
                         float* __restrict__ bpart, int h, int w, int C)
{
    __shared__ float sm[256 * 4];             // [row r][channel c][l0, l1, b0, b1]
    const int b = blockIdx.y, hw = h * w;
    const int R = 256 / C;
    const int r = threadIdx.x / C, c = threadIdx.x - r * C;
    const bool coupled = r < R && c >= c0 && c < c0 + n;
    const int i = c - c0;
    const float gl = dld ? dld[b] : 0.f;
    float e0 = 0.f, e1 = 0.f;
    if (coupled) { e0 = expf(logs3[2 * i] * 3.0f); e1 = expf(logs3[2 * i + 1] * 3.0f); }
    float l0 = 0.f, l1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (r < R) {
        const int pend = min((int)(blockIdx.x + 1) * CPX, hw);
        for (int p = blockIdx.x * CPX + r; p < pend; p += R) {
            const int64_t o = ((int64_t)b * hw + p) * C + c;
            const float g = dzo[o];
            float dzi = g;
            if (coupled) {
                const int y = p / w, x = p - y * w;
                const int64_t ro = (((int64_t)b * (h + 2) + y + 1) * (w + 2) + x + 1) * hs + 2 * i;
                const float2 hh = *(const float2*)(hr + ro);
                const float v = z[o];
                const float hs0 = hh.x * e0, hs1 = hh.y * e1;
                const float sg = sigmoid2(hs1), sc = sg + eps;
                float dsh, dsc;
                if (branch == 0) { dzi = g * sc; dsh = dzi; dsc = g * (v + hs0) + gl / sc; }
                else { dzi = g / sc; dsh = -g; dsc = -g * v / (sc * sc) - gl / sc; }
                const float dp = dsc * sg * (1.0f - sg);
                const float2 d = {dsh * e0, dp * e1};
                *(float2*)(dh + ro) = d;
                b0 += d.x;
                b1 += d.y;
                l0 += dsh * hs0 * 3.0f;
                l1 += dp * hs1 * 3.0f;
            }
            dz[o] = dzi;
        }
    }
    if (r < R) {
        float* q = sm + (r * C + c) * 4;
        q[0] = l0; q[1] = l1; q[2] = b0; q[3] = b1;
    }
    __syncthreads();
    if (threadIdx.x < 2 * n) {                // one thread per row of h: channel i = k / 2, (shift | scale) = k & 1
        const int k = threadIdx.x, cc = c0 + (k >> 1), e = k & 1;
        float tl = 0.f, tb = 0.f;
        for (int rr = 0; rr < R; ++rr) { tl += sm[(rr * C + cc) * 4 + e]; tb += sm[(rr * C + cc) * 4 + 2 + e]; }
        const int64_t slot = ((int64_t)b * gridDim.x + blockIdx.x) * 2 * n + k;
        lpart[slot] = tl;
        if (bpart) bpart[slot] = tb;
    }
}

}  // namespace

extern "C" int64_t ifl_an_relu_blocks(int B, int h, int w) { return ((int64_t)B * h * w + GPX - 1) / GPX; }

int ifl_k_an_relu_fwd(const float* x, int xs, float* y, const float* bias, const float* logs, int B, int h, int w, int C, hipStream_t stream,
                      int nz, int64_t xz, int64_t yz, int64_t pz)
{
    IFL_CHECK_ARG(nz >= 1 && nz <= 65535, IFL_EINVAL, "an_relu: %d problems", nz);
    IFL_CHECK_ARG(x && y && bias && logs && xs >= C && xs % 4 == 0, IFL_EINVAL, "null pointer / stride");
    IFL_CHECK_ARG(C == 64, IFL_ENOTSUP, "an_relu: C=%d (the coupling networks' hidden width is 64)", C);
    IFL_CHECK_ARG(B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "bad shape");
    const int64_t npx = (int64_t)B * h * w;
    hipLaunchKernelGGL(an_relu_fwd_kernel, dim3((unsigned)ifl_an_relu_blocks(B, h, w), nz), dim3(256), 0, stream, x, xs, y, bias, logs, npx, h, w, C, xz, yz, pz);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_an_relu_fwd(const float* x, float* y, const float* bias, const float* logs, int B, int h, int w, int C, void* stream)
{
    return ifl_k_an_relu_fwd(x, C, y, bias, logs, B, h, w, C, (hipStream_t)stream, 1, 0, 0, 0);
}

int ifl_k_an_relu_bwd(const float* x, int xs, const float* dy, float* dx, const float* bias, const float* logs, float* partials,
                      int B, int h, int w, int C, hipStream_t stream, int nz, int64_t xz, int64_t gz, int64_t dxz, int64_t pz, int64_t partz)
{
    IFL_CHECK_ARG(nz >= 1 && nz <= 65535, IFL_EINVAL, "an_relu: %d problems", nz);
    IFL_CHECK_ARG(x && dy && dx && bias && logs && partials && xs >= C && xs % 4 == 0, IFL_EINVAL, "null pointer / stride");
    IFL_CHECK_ARG(C == 64, IFL_ENOTSUP, "an_relu: C=%d (the coupling networks' hidden width is 64)", C);
    IFL_CHECK_ARG(B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "bad shape");
    const int64_t npx = (int64_t)B * h * w;
    hipLaunchKernelGGL(an_relu_bwd_kernel, dim3((unsigned)ifl_an_relu_blocks(B, h, w), nz), dim3(256), 0, stream, x, xs, dy, dx, bias, logs,
                       partials, npx, h, w, C, xz, gz, dxz, pz, partz);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_an_relu_bwd(const float* x, const float* dy, float* dx, const float* bias, const float* logs, float* partials,
                               int B, int h, int w, int C, void* stream)
{
    return ifl_k_an_relu_bwd(x, C, dy, dx, bias, logs, partials, B, h, w, C, (hipStream_t)stream, 1, 0, 0, 0, 0, 0);
}

extern "C" int ifl_coupling_blocks(int h, int w) { return (h * w + CPX - 1) / CPX; }

extern "C" int ifl_coupling_fwd(const float* z, float* zout, const float* h_ring, int hstride, const float* logs3, float eps, int branch,
                                int c0, int n, float* ldpart, int B, int h, int w, int C, void* stream)
{
    IFL_CHECK_ARG(z && zout && h_ring && logs3 && ldpart, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && C >= 1 && C <= 256, IFL_EINVAL, "bad shape");
    IFL_CHECK_ARG(hstride % 2 == 0, IFL_EINVAL, "odd row stride %d", hstride);
    IFL_CHECK_ARG(c0 >= 0 && n >= 1 && c0 + n <= C && 2 * n <= hstride, IFL_EINVAL, "channel range [%d, %d) of %d, %d rows", c0, c0 + n, C, hstride);
    IFL_CHECK_ARG(branch == 0 || branch == 1, IFL_EINVAL, "branch %d", branch);
    hipLaunchKernelGGL(coupling_fwd_kernel, dim3(ifl_coupling_blocks(h, w), B), dim3(256), 0, (hipStream_t)stream, z, zout, h_ring, hstride,
                       logs3, eps, branch, c0, n, ldpart, h, w, C);
    IFL_LAUNCH_CHECK();
    return 0;
}

int ifl_k_coupling_bwd(const float* z, const float* dzout, const float* dld, const float* h_ring, int hstride, const float* logs3,
                       float eps, int branch, int c0, int n, float* dz, float* dh_ring, float* dlogs_part, float* dbias_part, int B,
                       int h, int w, int C, hipStream_t stream)
{
    IFL_CHECK_ARG(z && dzout && h_ring && logs3 && dz && dh_ring && dlogs_part, IFL_EINVAL, "null pointer");
    IFL_CHECK_ARG(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && C >= 1 && C <= 256, IFL_EINVAL, "bad shape");
    IFL_CHECK_ARG(hstride % 2 == 0, IFL_EINVAL, "odd row stride %d", hstride);
    IFL_CHECK_ARG(c0 >= 0 && n >= 1 && c0 + n <= C && 2 * n <= hstride && n <= 64, IFL_EINVAL, "channel range [%d, %d) of %d, %d rows", c0, c0 + n, C, hstride);
    IFL_CHECK_ARG(branch == 0 || branch == 1, IFL_EINVAL, "branch %d", branch);
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(ifl_coupling_blocks(h, w), B), dim3(256), 0, stream, z, dzout, dld,
                       h_ring, hstride, logs3, eps, branch, c0, n, dz, dh_ring, dlogs_part, dbias_part, h, w, C);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_coupling_bwd(const float* z, const float* dzout, const float* dld, const float* h_ring, int hstride, const float* logs3,
                                float eps, int branch, int c0, int n, float* dz, float* dh_ring, float* dlogs_part, int B, int h, int w,
                                int C, void* stream)
{
    return ifl_k_coupling_bwd(z, dzout, dld, h_ring, hstride, logs3, eps, branch, c0, n, dz, dh_ring, dlogs_part, nullptr, B, h, w, C,
                              (hipStream_t)stream);
}


// ---- Adam over a device table of (parameter, gradient, moments) chunks: one launch for every parameter tensor --------------
// torch.optim.Adam (train.py:53) in its fused multi-tensor form takes 36 tensors per launch, 22 launches of ~43 us for the
// flow's 770 parameter tensors although the update moves only 230 MB.  Arithmetic as torch's fused kernel:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t = *step_in + 1 read from the device; *skip != 0 (the reference's loss guards, decided on the device) leaves
// everything untouched.  Block (0) also writes *step_out = t (or *step_in when skipped): the caller alternates two cells.
namespace {

__global__ __launch_bounds__(256)
void adam_table_kernel(const ifl_adam_row* __restrict__ table, float lr, float b1, float b2, float eps,
                       const float* __restrict__ step_in, float* __restrict__ step_out, const float* __restrict__ skip)
{
    const bool skipped = skip && *skip != 0.f;
    const float t = *step_in + 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *step_out = skipped ? *step_in : t;
    if (skipped) return;
    const ifl_adam_row r = table[blockIdx.x];
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    auto upd = [&](float& p, float g, float& m, float& v) {
        m = m + (g - m) * (1.0f - b1);
        v = b2 * v + (1.0f - b2) * g * g;
        const float denom = sqrtf(v) / bc2s + eps;
        p = p - step_size * (m / denom);
    };
    const bool vec = ((((uintptr_t)r.p | (uintptr_t)r.g | (uintptr_t)r.m | (uintptr_t)r.v) & 15) == 0);
    if (vec) {
        const int n4 = r.n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            f32x4 p = ((f32x4*)r.p)[i], m = ((f32x4*)r.m)[i], v = ((f32x4*)r.v)[i];
            const f32x4 g = ((const f32x4*)r.g)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                upd(pe, g[e], me, ve);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            ((f32x4*)r.p)[i] = p; ((f32x4*)r.m)[i] = m; ((f32x4*)r.v)[i] = v;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < r.n; i += 256) upd(r.p[i], r.g[i], r.m[i], r.v[i]);
    } else {
        for (int i = threadIdx.x; i < r.n; i += 256) upd(r.p[i], r.g[i], r.m[i], r.v[i]);
    }
}

}  // namespace

extern "C" int ifl_adam_table(const ifl_adam_row* table_dev, int n_rows, float lr, float beta1, float beta2, float eps,
                              const float* step_in, float* step_out, const float* skip, void* stream)
{
    IFL_CHECK_ARG(table_dev && step_in && step_out && step_in != step_out, IFL_EINVAL, "adam: null pointer / step cells");
    IFL_CHECK_ARG(n_rows >= 1 && n_rows <= 0x7fffffff, IFL_EINVAL, "adam: %d rows", n_rows);
    IFL_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, IFL_EINVAL, "adam: hyper-parameters");
    hipLaunchKernelGGL(adam_table_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, table_dev, lr, beta1, beta2, eps, step_in,
                       step_out, skip);
    IFL_LAUNCH_CHECK();
    return 0;
}
