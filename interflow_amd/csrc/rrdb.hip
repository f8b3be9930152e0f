// rrdb.hip -- the conditioning encoder: RRDBNet.forward(get_steps=True) + RRDBGlowNet.rrdbPreprocessing, natively.
//
// Replaces (reference interflow/models/modules):
//   RRDBNet.forward                 RRDBNet_arch.py:85-148   space2depth(4) -> conv_first -> 23 x RRDB -> trunk_conv (+fea)
//                                                             -> nearest x2 + upconv1 + LeakyReLU -> nearest x2 + upconv2 +
//                                                             LeakyReLU -> HRconv + LeakyReLU -> conv_last; fea_up0 = bilinear x1/2
//   ResidualDenseBlock_5C / RRDB    RRDBNet_arch.py:9-45     dense concatenation = channel ranges of one buffer (no copy)
//   rrdbPreprocessing               RRDBGlowNet_arch.py:217-246  blocks 1/8/15/22 stacked (256 ch), resized to every key's
//                                                             resolution with nearest neighbour and concatenated behind
//                                                             the key's own 64 channels -> 320-channel NCHW tensors
// All convolutions run through convg.hip.  Buffers are ringed channels-last planes in a caller-owned workspace.
#include "ifl_common.h"
#include <mutex>

#define RRDB_NF 64
#define RRDB_GC 32
#define RRDB_DENSE (RRDB_NF + 4 * RRDB_GC)      // 192 channels of a dense block's buffer

namespace {

// ---- x NCHW [B,3,H,W] -> space-to-depth(4) ringed NHWC [B][H/4+2][W/4+2][64], channel = c*16 + i*4 + j, 48..63 zero
// (F.unfold(x, 4, stride=4) order, RRDBNet_arch.py:79-83).  One thread per (output pixel, input channel c, row i).
__global__ __launch_bounds__(256)
void s2d_ring_kernel(const float* __restrict__ x, float* __restrict__ o, int B, int Cin, int H, int W)
{
    const int h4 = H / 4, w4 = W / 4;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * h4 * w4 * Cin * 4) return;
    int i = e & 3; int64_t t = e >> 2; int c = t % Cin; t /= Cin; int w = t % w4; t /= w4; int h = t % h4; int b = t / h4;
    f32x4 v = *(const f32x4*)(x + (((int64_t)b * Cin + c) * H + 4 * h + i) * W + 4 * w);
    *(f32x4*)(o + (((int64_t)b * (h4 + 2) + h + 1) * (w4 + 2) + w + 1) * RRDB_NF + c * 16 + i * 4) = v;
}

// ---- nearest-neighbour x2 upsampling, ringed 64-channel plane -> ringed plane of twice the size (F.interpolate
// scale_factor=2 mode='nearest', RRDBNet_arch.py:105,108).  One thread per (output pixel, float4 of channels).
__global__ __launch_bounds__(256)
void up2_ring_kernel(const float* __restrict__ in, float* __restrict__ o, int B, int h, int w)
{
    const int H = 2 * h, W = 2 * w;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * H * W * 16) return;
    int q = e & 15; int64_t t = e >> 4; int x = t % W; t /= W; int y = t % H; int b = t / H;
    f32x4 v = *(const f32x4*)(in + (((int64_t)b * (h + 2) + y / 2 + 1) * (w + 2) + x / 2 + 1) * RRDB_NF + q * 4);
    *(f32x4*)(o + (((int64_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * RRDB_NF + q * 4) = v;
}

// ---- bilinear x1/2 (F.interpolate scale_factor=1/2 mode='bilinear', align_corners=False: source coordinate 2d + 0.5,
// i.e. equal weights on pixels 2d and 2d+1 in each direction), RRDBNet_arch.py:138.  Written straight to NCHW.
__global__ __launch_bounds__(256)
void half_bilinear_nchw_kernel(const float* __restrict__ in, float* __restrict__ o, int B, int h, int w, int Ctot)
{
    const int ho = h / 2, wo = w / 2;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * RRDB_NF * ho * wo) return;
    int x = e % wo; int64_t t = e / wo; int y = t % ho; t /= ho; int c = t % RRDB_NF; int b = t / RRDB_NF;
    const float* p = in + (((int64_t)b * (h + 2) + 2 * y + 1) * (w + 2) + 2 * x + 1) * RRDB_NF + c;
    const int64_t row = (int64_t)(w + 2) * RRDB_NF;
    float top = 0.5f * p[0] + 0.5f * p[RRDB_NF], bot = 0.5f * p[row] + 0.5f * p[row + RRDB_NF];
    o[(((int64_t)b * Ctot + c) * ho + y) * wo + x] = 0.5f * top + 0.5f * bot;
}

// ---- copy channels [0, 64) of a ringed plane with channel stride `stride` into a ringed 64-channel plane ------------
__global__ __launch_bounds__(256)
void copy64_kernel(const float* __restrict__ in, int stride, float* __restrict__ o, int64_t npix_ring)
{
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npix_ring * 16) return;
    int q = e & 15; int64_t p = e >> 4;
    *(f32x4*)(o + p * RRDB_NF + q * 4) = *(const f32x4*)(in + p * stride + q * 4);
}

// ---- ringed channels-last [B][hs+2][ws+2][cs] (first `C` channels) -> NCHW [B, Ctot, ho, wo] at channel offset coff,
// resized with nearest neighbour: src = min(floor(dst * hs/ho), hs-1)  (F.interpolate(concat, (h, w)) default mode,
// RRDBGlowNet_arch.py:240; identity when the sizes agree).  32x32 transpose through LDS: coalesced on both sides.
__global__ __launch_bounds__(256)
void ring_to_nchw_resize_kernel(const float* __restrict__ in, int cs, int C, int hs, int ws, float* __restrict__ o, int Ctot,
                                int coff, int ho, int wo, float sy, float sx)
{
    __shared__ float tile[32][33];
    const int P = ho * wo;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
    const int i = threadIdx.x & 31, jj = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = p0 + jj + 8 * k, c = c0 + i;
        float v = 0.f;
        if (p < P && c < C) {
            int y = p / wo, x = p - y * wo;
            int ysrc = (ho == hs) ? y : min((int)floorf(y * sy), hs - 1);
            int xsrc = (wo == ws) ? x : min((int)floorf(x * sx), ws - 1);
            v = in[(((int64_t)b * (hs + 2) + ysrc + 1) * (ws + 2) + xsrc + 1) * cs + c];
        }
        tile[jj + 8 * k][i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = c0 + jj + 8 * k, p = p0 + i;
        if (c < C && p < P) o[((int64_t)b * Ctot + coff + c) * P + p] = tile[i][jj + 8 * k];
    }
}

struct Plane { int h, w; IflGeom g; int64_t ring; };      // ring = pixels of one image's ringed plane
Plane make_plane(int h, int w) { return Plane{h, w, ifl_choose_geom(h, w), (int64_t)(h + 2) * (w + 2)}; }

struct EncWs {
    float *s2d, *dense[3], *blocks, *last, *up2in, *f2, *up4in, *f4, *hr, *outp;
    int64_t bytes;
};

EncWs enc_layout(void* base, int B, int H, int W, int n_blocks)
{
    EncWs e{};
    int64_t p = 0;
    auto take = [&](int64_t nfloats) { int64_t r = p; p += (nfloats * 4 + 255) / 256 * 256; return (float*)((char*)base + r); };
    const int64_t r4 = (int64_t)B * (H / 4 + 2) * (W / 4 + 2), r2 = (int64_t)B * (H / 2 + 2) * (W / 2 + 2),
                  r1 = (int64_t)B * (H + 2) * (W + 2);
    e.s2d = take(r4 * RRDB_NF);
    for (int i = 0; i < 3; ++i) e.dense[i] = take(r4 * RRDB_DENSE);
    e.blocks = take(r4 * RRDB_NF * (n_blocks > 0 ? n_blocks : 1));
    e.last = take(r4 * RRDB_NF);
    e.up2in = take(r2 * RRDB_NF);
    e.f2 = take(r2 * RRDB_NF);
    e.up4in = take(r1 * RRDB_NF);
    e.f4 = take(r1 * RRDB_NF);
    e.hr = take(r1 * RRDB_NF);
    e.outp = take(r1 * 32);
    e.bytes = p;
    return e;
}

int check_enc_shape(int B, int H, int W, int nb, int n_blocks)
{
    IFL_CHECK_ARG(B >= 1 && B <= 65535 && H >= 4 && W >= 4, IFL_EINVAL, "bad shape B=%d H=%d W=%d", B, H, W);
    IFL_CHECK_ARG(H % 4 == 0 && W % 4 == 0, IFL_EINVAL, "H=%d, W=%d must be multiples of 4 (space2depth, RRDBNet_arch.py:79-83)", H, W);
    IFL_CHECK_ARG(nb >= 1 && nb <= 64 && n_blocks >= 0 && n_blocks <= 8, IFL_ENOTSUP, "nb=%d, n_blocks=%d", nb, n_blocks);
    IFL_CHECK_ARG((int64_t)(H + 2) * (W + 2) * RRDB_DENSE < (int64_t)1 << 31, IFL_ENOTSUP, "image plane too large: tile the image");
    return 0;
}

// the convolutions of the encoder in execution order: (input channels padded to 32, output channels padded to 32)
struct ConvDesc { int cin, cout; int64_t w_off, b_off; };
int conv_table(int nb, ConvDesc* d, int64_t* total)
{
    int n = 0;
    int64_t p = 0;
    auto add = [&](int cin, int cout) {
        if (d) d[n] = ConvDesc{cin, cout, p, p + (int64_t)cout * cin * 9};
        p += (int64_t)cout * cin * 9 + cout;
        p = (p + 63) / 64 * 64;
        ++n;
    };
    add(64, RRDB_NF);                                              // conv_first (48 real input channels)
    for (int k = 0; k < nb * 3; ++k) {
        for (int c = 0; c < 4; ++c) add(RRDB_NF + c * RRDB_GC, RRDB_GC);
        add(RRDB_DENSE, RRDB_NF);
    }
    add(RRDB_NF, RRDB_NF);                                         // trunk_conv
    add(RRDB_NF, RRDB_NF);                                         // upconv1
    add(RRDB_NF, RRDB_NF);                                         // upconv2
    add(RRDB_NF, RRDB_NF);                                         // HRconv
    add(RRDB_NF, 32);                                              // conv_last (3 real output channels)
    if (total) *total = p;
    return n;
}

IflConvArgs conv_args(const float* params, const ConvDesc& d, const float* src, int sstride, float* out, int ostride,
                      int ooff, int act, const Plane& pl)
{
    IflConvArgs a{};
    a.src[0] = src; a.src_stride[0] = sstride; a.src_off[0] = 0; a.src_chunks[0] = d.cin / 32;
    a.nsrc = 1; a.nchunks = d.cin / 32;
    a.w = params + d.w_off; a.bias = params + d.b_off;
    a.out = out; a.out_stride = ostride; a.out_off = ooff; a.cout = d.cout; a.taps = 9; a.act = act;
    a.alpha = a.beta = 1.f;
    a.g = pl.g;
    return a;
}

}  // namespace

extern "C" int ifl_rrdb_conv_count(int nb) { return conv_table(nb, nullptr, nullptr); }

extern "C" int64_t ifl_rrdb_param_floats(int nb)
{
    int64_t total = 0;
    conv_table(nb, nullptr, &total);
    return total;
}

extern "C" int ifl_rrdb_conv_desc(int nb, int idx, int* cin, int* cout, int64_t* w_off, int64_t* b_off)
{
    static thread_local ConvDesc d[64 * 15 + 8];
    IFL_CHECK_ARG(nb >= 1 && nb <= 64, IFL_ENOTSUP, "nb=%d", nb);
    int n = conv_table(nb, d, nullptr);
    IFL_CHECK_ARG(idx >= 0 && idx < n, IFL_EINVAL, "conv index %d outside 0..%d", idx, n - 1);
    *cin = d[idx].cin; *cout = d[idx].cout; *w_off = d[idx].w_off; *b_off = d[idx].b_off;
    return 0;
}

extern "C" int64_t ifl_rrdb_ws_bytes(int B, int H, int W, int n_blocks)
{
    int rc = check_enc_shape(B, H, W, 1, n_blocks);
    if (rc) return rc;
    return enc_layout(nullptr, B, H, W, n_blocks).bytes;
}

extern "C" int ifl_rrdb_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, int n_blocks, void* stream)
{
    int rc = check_enc_shape(B, H, W, 1, n_blocks);
    if (rc) return rc;
    IFL_CHECK_ARG(ws != nullptr, IFL_EINVAL, "null workspace");
    EncWs e = enc_layout(ws, B, H, W, n_blocks);
    IFL_CHECK_ARG(ws_bytes >= e.bytes, IFL_ESMALLWS, "workspace %lld B < required %lld B", (long long)ws_bytes, (long long)e.bytes);
    IFL_HIP(hipMemsetAsync(ws, 0, (size_t)ws_bytes, (hipStream_t)stream));      // the zero rings = the convs' padding
    return 0;
}

extern "C" int ifl_rrdb_encode(const float* params, int nb, const int* block_idxs, int n_blocks, void* ws, const float* x,
                               float* fea_up1, float* fea_up2, float* fea_up4, float* fea_up0, float* out, float* blocks_out,
                               int B, int H, int W, void* stream)
{
    int rc = check_enc_shape(B, H, W, nb, n_blocks);
    if (rc) return rc;
    IFL_CHECK_ARG(params && ws && x, IFL_EINVAL, "null pointer");
    for (int i = 0; i < n_blocks; ++i)
        IFL_CHECK_ARG(block_idxs && block_idxs[i] >= 0 && block_idxs[i] < nb && (i == 0 || block_idxs[i] > block_idxs[i - 1]),
                      IFL_EINVAL, "block indices must be increasing and below nb");
    hipStream_t st = (hipStream_t)stream;
    static thread_local ConvDesc d[64 * 15 + 8];
    conv_table(nb, d, nullptr);
    EncWs e = enc_layout(ws, B, H, W, n_blocks);
    const Plane p4 = make_plane(H / 4, W / 4), p2 = make_plane(H / 2, W / 2), p1 = make_plane(H, W);
    const int64_t r4 = (int64_t)B * p4.ring;
    auto blocks1d = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };

    // space2depth(4) + conv_first -> dense[0][0:64]
    hipLaunchKernelGGL(s2d_ring_kernel, blocks1d((int64_t)B * p4.h * p4.w * 3 * 4), dim3(256), 0, st, x, e.s2d, B, 3, H, W);
    IFL_LAUNCH_CHECK();
    int ci = 0;
    if ((rc = ifl_k_convg(conv_args(params, d[ci++], e.s2d, RRDB_NF, e.dense[0], RRDB_DENSE, 0, 0, p4), B, st))) return rc;

    // trunk: the RRDB's input lives in dense[cur][0:64]; RDB1: cur -> cur+1, RDB2: cur+1 -> cur+2, RDB3: cur+2 -> cur+1
    int cur = 0, saved = 0;
    for (int k = 0; k < nb; ++k) {
        float* bufs[3] = {e.dense[cur], e.dense[(cur + 1) % 3], e.dense[(cur + 2) % 3]};
        for (int r = 0; r < 3; ++r) {
            float* X = bufs[r];                                   // this RDB's dense buffer
            for (int c = 0; c < 4; ++c)                           // x_{c+1} = lrelu(conv(cat(x, x1..xc)))   :24-27
                if ((rc = ifl_k_convg(conv_args(params, d[ci++], X, RRDB_DENSE, X, RRDB_DENSE, RRDB_NF + c * RRDB_GC, 1, p4), B, st)))
                    return rc;
            float* Y = r == 2 ? bufs[1] : bufs[r + 1];            // x5*0.2 + x  (:29)  [RDB3: then *0.2 + RRDB input, :45]
            IflConvArgs a = conv_args(params, d[ci++], X, RRDB_DENSE, Y, RRDB_DENSE, 0, 0, p4);
            a.res1 = X; a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 0.2f;
            if (r == 2) { a.res2 = bufs[0]; a.res2_stride = RRDB_DENSE; a.res2_off = 0; a.beta = 0.2f; }
            if ((rc = ifl_k_convg(a, B, st))) return rc;
        }
        cur = (cur + 1) % 3;
        if (saved < n_blocks && block_idxs[saved] == k) {         // block_results["block_k"] = fea   (:97-99)
            hipLaunchKernelGGL(copy64_kernel, blocks1d(r4 * 16), dim3(256), 0, st, e.dense[cur], RRDB_DENSE,
                               e.blocks + (int64_t)saved * r4 * RRDB_NF, r4);
            IFL_LAUNCH_CHECK();
            ++saved;
        }
    }
    // last_lr_fea = fea + trunk_conv(fea)   (:101-103)
    {
        IflConvArgs a = conv_args(params, d[ci++], e.dense[cur], RRDB_DENSE, e.last, RRDB_NF, 0, 0, p4);
        a.res1 = e.dense[cur]; a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 1.f;
        if ((rc = ifl_k_convg(a, B, st))) return rc;
    }
    const bool need2 = fea_up2 || fea_up4 || out, need4 = fea_up4 || out;
    if (need2) {   // fea_up2 = lrelu(upconv1(nearest x2))  -- the reference's in-place LeakyReLU (:77,106) rewrites the tensor it returns
        hipLaunchKernelGGL(up2_ring_kernel, blocks1d((int64_t)B * p2.h * p2.w * 16), dim3(256), 0, st, e.last, e.up2in, B, p4.h, p4.w);
        IFL_LAUNCH_CHECK();
        if ((rc = ifl_k_convg(conv_args(params, d[ci], e.up2in, RRDB_NF, e.f2, RRDB_NF, 0, 1, p2), B, st))) return rc;
    }
    ++ci;
    if (need4) {
        hipLaunchKernelGGL(up2_ring_kernel, blocks1d((int64_t)B * p1.h * p1.w * 16), dim3(256), 0, st, e.f2, e.up4in, B, p2.h, p2.w);
        IFL_LAUNCH_CHECK();
        if ((rc = ifl_k_convg(conv_args(params, d[ci], e.up4in, RRDB_NF, e.f4, RRDB_NF, 0, 1, p1), B, st))) return rc;
    }
    ++ci;
    if (out) {     // out = conv_last(lrelu(HRconv(fea)))   (:125)
        if ((rc = ifl_k_convg(conv_args(params, d[ci], e.f4, RRDB_NF, e.hr, RRDB_NF, 0, 1, p1), B, st))) return rc;
        if ((rc = ifl_k_convg(conv_args(params, d[ci + 1], e.hr, RRDB_NF, e.outp, 32, 0, 0, p1), B, st))) return rc;
    }
    ci += 2;

    // outputs (NCHW).  320-channel tensors: the key's own 64 channels, then the saved blocks resized (nearest) to its size.
    auto emit = [&](const float* src, int cs, int C, const Plane& ps, float* dst, int Ctot, int coff, int ho, int wo) -> int {
        dim3 grid(ifl_cdiv(ho * wo, 32), ifl_cdiv(C, 32), B);
        hipLaunchKernelGGL(ring_to_nchw_resize_kernel, grid, dim3(256), 0, st, src, cs, C, ps.h, ps.w, dst, Ctot, coff, ho, wo,
                           (float)ps.h / (float)ho, (float)ps.w / (float)wo);
        IFL_LAUNCH_CHECK();
        return 0;
    };
    const int Ctot = RRDB_NF * (1 + n_blocks);
    auto emit_blocks = [&](float* dst, int ho, int wo) -> int {
        for (int i = 0; i < n_blocks; ++i)
            if (int r = emit(e.blocks + (int64_t)i * r4 * RRDB_NF, RRDB_NF, RRDB_NF, p4, dst, Ctot, RRDB_NF * (1 + i), ho, wo)) return r;
        return 0;
    };
    if (fea_up1) { if ((rc = emit(e.last, RRDB_NF, RRDB_NF, p4, fea_up1, Ctot, 0, p4.h, p4.w)) || (rc = emit_blocks(fea_up1, p4.h, p4.w))) return rc; }
    if (fea_up2) { if ((rc = emit(e.f2, RRDB_NF, RRDB_NF, p2, fea_up2, Ctot, 0, p2.h, p2.w)) || (rc = emit_blocks(fea_up2, p2.h, p2.w))) return rc; }
    if (fea_up4) { if ((rc = emit(e.f4, RRDB_NF, RRDB_NF, p1, fea_up4, Ctot, 0, p1.h, p1.w)) || (rc = emit_blocks(fea_up4, p1.h, p1.w))) return rc; }
    if (fea_up0) {
        const int ho = p4.h / 2, wo = p4.w / 2;
        if (ho > 0 && wo > 0) {
            hipLaunchKernelGGL(half_bilinear_nchw_kernel, blocks1d((int64_t)B * RRDB_NF * ho * wo), dim3(256), 0, st, e.last, fea_up0,
                               B, p4.h, p4.w, Ctot);
            IFL_LAUNCH_CHECK();
            if ((rc = emit_blocks(fea_up0, ho, wo))) return rc;
        }
    }
    if (out) { if ((rc = emit(e.outp, 32, 3, p1, out, 3, 0, p1.h, p1.w))) return rc; }
    if (blocks_out)
        for (int i = 0; i < n_blocks; ++i)
            if ((rc = emit(e.blocks + (int64_t)i * r4 * RRDB_NF, RRDB_NF, RRDB_NF, p4, blocks_out + (int64_t)i * B * RRDB_NF * p4.h * p4.w,
                           RRDB_NF, 0, p4.h, p4.w))) return rc;
    return 0;
}

// ======================================================================================================================
// The encoder with gradients (train.py:72,129 with network_G.train_RRDB, RRDBGlowNet_arch.py:26-33): forward that keeps
// every dense block's buffer, and a backward that walks the blocks in reverse -- per convolution one data-gradient launch
// (convg with the transposed, flipped weights, accumulating into the block's gradient buffer; LeakyReLU's derivative of the
// newest 32 channels in its epilogue) -- followed by the weight gradients of the same convolution of all 3 nb blocks in one
// launch each.  Only what feeds 'fea_up1' / 'fea_up2' (the maps the flow reads) is computed; upconv2, HRconv and conv_last
// receive no gradient from the flow's loss.
//   D[q], q = 0..3nb : dense buffer of block q ([0,64) = its input = the output of block q-1, [64,192) = x1..x4); D[3nb][0,64) = fea
//   G[q]             : gradients of the same; E[q] = gradient wrt x5 of block q (64 channels)
namespace {

struct TrainWs {
    float *s2d, *D, *G, *E, *last, *dlast, *up2in, *f2, *df2, *dup2in;
    int64_t dstride, estride, gbytes, bytes;
};

TrainWs train_layout(void* base, int B, int H, int W, int nb)
{
    TrainWs e{};
    int64_t p = 0;
    auto take = [&](int64_t nfloats) { int64_t r = p; p += (nfloats * 4 + 255) / 256 * 256; return (float*)((char*)base + r); };
    const int64_t r4 = (int64_t)B * (H / 4 + 2) * (W / 4 + 2), r2 = (int64_t)B * (H / 2 + 2) * (W / 2 + 2);
    const int nq = 3 * nb;
    e.dstride = (r4 * RRDB_DENSE + 63) / 64 * 64;
    e.estride = (r4 * RRDB_NF + 63) / 64 * 64;
    e.s2d = take(r4 * RRDB_NF);
    e.D = take(e.dstride * (nq + 1));
    e.G = take(e.dstride * (nq + 1));
    e.gbytes = e.dstride * (nq + 1) * 4;
    e.E = take(e.estride * nq);
    e.last = take(r4 * RRDB_NF);
    e.dlast = take(r4 * RRDB_NF);
    e.up2in = take(r2 * RRDB_NF);
    e.f2 = take(r2 * RRDB_NF);
    e.df2 = take(r2 * RRDB_NF);
    e.dup2in = take(r2 * RRDB_NF);
    e.bytes = p;
    return e;
}

// NCHW gradient [B][Ctot][f hs][f ws], channels coff + [0, 64) -> ringed plane [B][hs+2][ws+2][stride] at channel off: the
// adjoint of ring_to_nchw_resize_kernel for equal sizes (f = 1) and for the nearest x2 resize (f = 2: the four children of a
// pixel are added).  mask (optional, ringed, 64 channels): LeakyReLU(0.2)'s derivative at that activation.  accumulate: +=.
__global__ __launch_bounds__(256)
void nchw_to_ring_kernel(const float* __restrict__ src, int Ctot, int coff, int f, float* __restrict__ dst, int stride, int off,
                         int B, int hs, int ws, const float* __restrict__ mask, int accumulate)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * hs * ws * RRDB_NF) return;
    const int c = e & 63; int64_t t = e >> 6; const int x = t % ws; t /= ws; const int y = t % hs; const int b = t / hs;
    const int ho = f * hs, wo = f * ws;
    const float* s = src + (((int64_t)b * Ctot + coff + c) * ho + f * y) * wo + f * x;
    float v = s[0];
    if (f == 2) v = (v + s[1]) + (s[wo] + s[wo + 1]);
    const int64_t rp = ((int64_t)b * (hs + 2) + y + 1) * (ws + 2) + x + 1;
    if (mask) v = mask[rp * RRDB_NF + c] > 0.f ? v : v * 0.2f;
    float* d = dst + rp * stride + off + c;
    *d = accumulate ? *d + v : v;
}

// adjoint of up2_ring_kernel: dst[b][y][x] += the four children of src (ringed 64-channel planes)
__global__ __launch_bounds__(256)
void down2_add_ring_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int h, int w)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * h * w * 16) return;
    const int q = e & 15; int64_t t = e >> 4; const int x = t % w; t /= w; const int y = t % h; const int b = t / h;
    const int W2 = 2 * w + 2;
    const float* s = src + (((int64_t)b * (2 * h + 2) + 2 * y + 1) * W2 + 2 * x + 1) * RRDB_NF + q * 4;
    const f32x4 a = *(const f32x4*)s, bb = *(const f32x4*)(s + RRDB_NF), c = *(const f32x4*)(s + (int64_t)W2 * RRDB_NF),
                d = *(const f32x4*)(s + (int64_t)(W2 + 1) * RRDB_NF);
    float* o = dst + (((int64_t)b * (h + 2) + y + 1) * (w + 2) + x + 1) * RRDB_NF + q * 4;
    f32x4 v = *(f32x4*)o;
    v = v + ((a + bb) + (c + d));
    *(f32x4*)o = v;
}

// A dense block's output y = x5 * 0.2 + x (RRDBNet_arch.py:29), for the third block of an RRDB followed by out * 0.2 + the RRDB's
// input (:45).  g = the gradient wrt what block q handed on (Gnext[0,64)):  gy = g (or 0.2 g);  E = 0.2 gy;  G[0,64) += gy;
// third block: Gskip[0,64) += g.  Interior pixels only.
__global__ __launch_bounds__(256)
void rdb_out_bwd_kernel(const float* __restrict__ Gnext, float* __restrict__ E, float* __restrict__ G, float* __restrict__ Gskip,
                        int B, int h, int w)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * h * w * 16) return;
    const int q = e & 15; int64_t t = e >> 4; const int x = t % w; t /= w; const int y = t % h; const int b = t / h;
    const int64_t rp = ((int64_t)b * (h + 2) + y + 1) * (w + 2) + x + 1;
    const f32x4 g = *(const f32x4*)(Gnext + rp * RRDB_DENSE + q * 4);
    const f32x4 gy = Gskip ? g * 0.2f : g;
    *(f32x4*)(E + rp * RRDB_NF + q * 4) = gy * 0.2f;
    f32x4* gp = (f32x4*)(G + rp * RRDB_DENSE + q * 4);
    *gp = *gp + gy;
    if (Gskip) {
        f32x4* sp = (f32x4*)(Gskip + rp * RRDB_DENSE + q * 4);
        *sp = *sp + g;
    }
}

// dst[z][c] += sum over the ringed pixels of src[z][pixel][off + c], c < nch (rings are zero): bias gradients.  Block =
// (32 channels, problem z, pixel slice): 8 pixel lanes per channel, one float atomic per channel and block (dst is zeroed by
// the caller with the rest of the gradient block).
constexpr int BIAS_SLICES = 48;
__global__ __launch_bounds__(256)
void bias_grad_kernel(const float* __restrict__ src, int64_t zstride, int stride, int off, int64_t npix, float* __restrict__ dst,
                      int64_t dst_zstride)
{
    __shared__ float red[8][33];
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
    const float* s = src + (int64_t)blockIdx.y * zstride + off + blockIdx.x * 32 + c;
    const int64_t per = (npix + gridDim.z - 1) / gridDim.z;
    const int64_t p0 = (int64_t)blockIdx.z * per, p1 = p0 + per < npix ? p0 + per : npix;
    float a = 0.f;
    for (int64_t p = p0 + l; p < p1; p += 8) a += s[p * stride];
    red[l][c] = a;
    __syncthreads();
    if (l == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][c];
        atomicAdd(dst + (int64_t)blockIdx.y * dst_zstride + blockIdx.x * 32 + c, t);
    }
}

// The weight gradients only produce parameter gradients: they run on a library-owned second stream per device, in slices of
// blocks, beside the sequential chain of small data-gradient launches that is still working on the earlier blocks.
struct EncFork { hipStream_t aux = nullptr; hipEvent_t ev[8] = {}; hipEvent_t join = nullptr; std::mutex mu; };
EncFork g_encfork[IFL_MAX_DEVICES];

int enc_fork(EncFork** out)
{
    int dev = 0;
    IFL_HIP(hipGetDevice(&dev));
    IFL_CHECK_ARG(dev >= 0 && dev < IFL_MAX_DEVICES, IFL_ENOTSUP, "device %d", dev);
    EncFork& f = g_encfork[dev];
    if (!f.aux) {
        IFL_HIP(hipStreamCreateWithFlags(&f.aux, hipStreamNonBlocking));
        for (auto& e : f.ev) IFL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        IFL_HIP(hipEventCreateWithFlags(&f.join, hipEventDisableTiming));
    }
    *out = &f;
    return 0;
}

int check_train_shape(int B, int H, int W, int nb, int n_blocks, const int* block_idxs)
{
    int rc = check_enc_shape(B, H, W, nb, n_blocks);
    if (rc) return rc;
    for (int i = 0; i < n_blocks; ++i)
        IFL_CHECK_ARG(block_idxs && block_idxs[i] >= 0 && block_idxs[i] < nb && (i == 0 || block_idxs[i] > block_idxs[i - 1]),
                      IFL_EINVAL, "block indices must be increasing and below nb");
    return 0;
}

}  // namespace

extern "C" int64_t ifl_rrdb_train_ws_bytes(int B, int H, int W, int nb)
{
    int rc = check_enc_shape(B, H, W, nb, 0);
    if (rc) return rc;
    return train_layout(nullptr, B, H, W, nb).bytes;
}

extern "C" int ifl_rrdb_train_ws_init(void* ws, int64_t ws_bytes, int B, int H, int W, int nb, void* stream)
{
    int rc = check_enc_shape(B, H, W, nb, 0);
    if (rc) return rc;
    IFL_CHECK_ARG(ws != nullptr, IFL_EINVAL, "null workspace");
    TrainWs e = train_layout(ws, B, H, W, nb);
    IFL_CHECK_ARG(ws_bytes >= e.bytes, IFL_ESMALLWS, "workspace %lld B < required %lld B", (long long)ws_bytes, (long long)e.bytes);
    IFL_HIP(hipMemsetAsync(ws, 0, (size_t)ws_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" int ifl_rrdb_train_fwd(const float* params, int nb, const int* block_idxs, int n_blocks, void* ws, const float* x,
                                  float* fea_up1, float* fea_up2, int B, int H, int W, void* stream)
{
    int rc = check_train_shape(B, H, W, nb, n_blocks, block_idxs);
    if (rc) return rc;
    IFL_CHECK_ARG(params && ws && x && fea_up1, IFL_EINVAL, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    static thread_local ConvDesc d[64 * 15 + 8];
    conv_table(nb, d, nullptr);
    TrainWs e = train_layout(ws, B, H, W, nb);
    const Plane p4 = make_plane(H / 4, W / 4), p2 = make_plane(H / 2, W / 2);
    auto blocks1d = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    auto Dq = [&](int q) { return e.D + (int64_t)q * e.dstride; };
    const int nq = 3 * nb;

    hipLaunchKernelGGL(s2d_ring_kernel, blocks1d((int64_t)B * p4.h * p4.w * 3 * 4), dim3(256), 0, st, x, e.s2d, B, 3, H, W);
    IFL_LAUNCH_CHECK();
    int ci = 0;
    if ((rc = ifl_k_convg(conv_args(params, d[ci++], e.s2d, RRDB_NF, Dq(0), RRDB_DENSE, 0, 0, p4), B, st))) return rc;
    for (int q = 0; q < nq; ++q) {
        float* X = Dq(q);
        for (int c = 0; c < 4; ++c)
            if ((rc = ifl_k_convg(conv_args(params, d[ci++], X, RRDB_DENSE, X, RRDB_DENSE, RRDB_NF + c * RRDB_GC, 1, p4), B, st))) return rc;
        IflConvArgs a = conv_args(params, d[ci++], X, RRDB_DENSE, Dq(q + 1), RRDB_DENSE, 0, 0, p4);
        a.res1 = X; a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 0.2f;
        if (q % 3 == 2) { a.res2 = Dq(q - 2); a.res2_stride = RRDB_DENSE; a.res2_off = 0; a.beta = 0.2f; }
        if ((rc = ifl_k_convg(a, B, st))) return rc;
    }
    {
        IflConvArgs a = conv_args(params, d[ci++], Dq(nq), RRDB_DENSE, e.last, RRDB_NF, 0, 0, p4);
        a.res1 = Dq(nq); a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 1.f;
        if ((rc = ifl_k_convg(a, B, st))) return rc;
    }
    if (fea_up2) {
        hipLaunchKernelGGL(up2_ring_kernel, blocks1d((int64_t)B * p2.h * p2.w * 16), dim3(256), 0, st, e.last, e.up2in, B, p4.h, p4.w);
        IFL_LAUNCH_CHECK();
        if ((rc = ifl_k_convg(conv_args(params, d[ci], e.up2in, RRDB_NF, e.f2, RRDB_NF, 0, 1, p2), B, st))) return rc;
    }
    auto emit = [&](const float* src, int cs, const Plane& ps, float* dst, int Ctot, int coff, int ho, int wo) -> int {
        dim3 grid(ifl_cdiv(ho * wo, 32), ifl_cdiv(RRDB_NF, 32), B);
        hipLaunchKernelGGL(ring_to_nchw_resize_kernel, grid, dim3(256), 0, st, src, cs, RRDB_NF, ps.h, ps.w, dst, Ctot, coff, ho, wo,
                           (float)ps.h / (float)ho, (float)ps.w / (float)wo);
        IFL_LAUNCH_CHECK();
        return 0;
    };
    const int Ctot = RRDB_NF * (1 + n_blocks);
    if ((rc = emit(e.last, RRDB_NF, p4, fea_up1, Ctot, 0, p4.h, p4.w))) return rc;
    if (fea_up2 && (rc = emit(e.f2, RRDB_NF, p2, fea_up2, Ctot, 0, p2.h, p2.w))) return rc;
    for (int i = 0; i < n_blocks; ++i) {
        const float* blk = Dq(3 * (block_idxs[i] + 1));               // the RRDB's output = the next block's input channels
        if ((rc = emit(blk, RRDB_DENSE, p4, fea_up1, Ctot, RRDB_NF * (1 + i), p4.h, p4.w))) return rc;
        if (fea_up2 && (rc = emit(blk, RRDB_DENSE, p4, fea_up2, Ctot, RRDB_NF * (1 + i), p2.h, p2.w))) return rc;
    }
    return 0;
}

// params_t: the convolutions' weights transposed and flipped (IFL_PACK_TRANSPOSED with all padded input channels as outputs) at
// the same offsets as `params`.  d_params: gradients in the parameters' own order [cout][cin_pad][9] at w_off and the bias
// gradients at b_off of ifl_rrdb_conv_desc (zero it first: the weight gradients are added with atomics).
extern "C" int ifl_rrdb_train_bwd(const float* params_t, int nb, const int* block_idxs, int n_blocks, void* ws, const float* d_up1,
                                  const float* d_up2, float* d_params, int B, int H, int W, void* stream)
{
    int rc = check_train_shape(B, H, W, nb, n_blocks, block_idxs);
    if (rc) return rc;
    IFL_CHECK_ARG(params_t && ws && d_params, IFL_EINVAL, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    static thread_local ConvDesc d[64 * 15 + 8];
    conv_table(nb, d, nullptr);
    TrainWs e = train_layout(ws, B, H, W, nb);
    const Plane p4 = make_plane(H / 4, W / 4), p2 = make_plane(H / 2, W / 2);
    const int64_t r4 = (int64_t)B * p4.ring, r2 = (int64_t)B * p2.ring;
    auto blocks1d = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    auto Dq = [&](int q) { return e.D + (int64_t)q * e.dstride; };
    auto Gq = [&](int q) { return e.G + (int64_t)q * e.dstride; };
    auto Eq = [&](int q) { return e.E + (int64_t)q * e.estride; };
    const int nq = 3 * nb, Ctot = RRDB_NF * (1 + n_blocks);
    const int i_trunk = 1 + 5 * nq, i_up1 = i_trunk + 1;
    const int64_t n4 = (int64_t)B * p4.h * p4.w * RRDB_NF, n2 = (int64_t)B * p2.h * p2.w * RRDB_NF;

    IFL_HIP(hipMemsetAsync(e.G, 0, (size_t)e.gbytes, st));
    // ---- the outputs' adjoints: 'fea_up1' = (last | blocks), 'fea_up2' = (f2 | blocks resized x2)
    if (d_up1) hipLaunchKernelGGL(nchw_to_ring_kernel, blocks1d(n4), dim3(256), 0, st, d_up1, Ctot, 0, 1, e.dlast, RRDB_NF, 0, B, p4.h, p4.w, (const float*)nullptr, 0);
    else IFL_HIP(hipMemsetAsync(e.dlast, 0, (size_t)r4 * RRDB_NF * 4, st));
    IFL_LAUNCH_CHECK();
    for (int i = 0; i < n_blocks; ++i) {
        float* g = Gq(3 * (block_idxs[i] + 1));
        if (d_up1) hipLaunchKernelGGL(nchw_to_ring_kernel, blocks1d(n4), dim3(256), 0, st, d_up1, Ctot, RRDB_NF * (1 + i), 1, g, RRDB_DENSE, 0, B, p4.h, p4.w, (const float*)nullptr, 1);
        if (d_up2) hipLaunchKernelGGL(nchw_to_ring_kernel, blocks1d(n4), dim3(256), 0, st, d_up2, Ctot, RRDB_NF * (1 + i), 2, g, RRDB_DENSE, 0, B, p4.h, p4.w, (const float*)nullptr, 1);
        IFL_LAUNCH_CHECK();
    }
    if (d_up2) {   // f2 = lrelu(upconv1(nearest x2 of last))
        hipLaunchKernelGGL(nchw_to_ring_kernel, blocks1d(n2), dim3(256), 0, st, d_up2, Ctot, 0, 1, e.df2, RRDB_NF, 0, B, p2.h, p2.w, (const float*)e.f2, 0);
        IFL_LAUNCH_CHECK();
        IflConvArgs a = conv_args(params_t, d[i_up1], e.df2, RRDB_NF, e.dup2in, RRDB_NF, 0, 0, p2);
        a.bias = nullptr;
        if ((rc = ifl_k_convg(a, B, st))) return rc;
        hipLaunchKernelGGL(down2_add_ring_kernel, blocks1d((int64_t)B * p4.h * p4.w * 16), dim3(256), 0, st, e.dup2in, e.dlast, B, p4.h, p4.w);
        IFL_LAUNCH_CHECK();
        if ((rc = ifl_k_wgrad(e.up2in, RRDB_NF, 0, RRDB_NF, e.df2, RRDB_NF, 0, RRDB_NF, d_params + d[i_up1].w_off, RRDB_NF, 0, 9, p2.g, B, st))) return rc;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(2, 1, BIAS_SLICES), dim3(256), 0, st, e.df2, (int64_t)0, RRDB_NF, 0, r2, d_params + d[i_up1].b_off, (int64_t)0);
        IFL_LAUNCH_CHECK();
    }
    // ---- last = fea + trunk_conv(fea):  G[nq][0,64) (already holding the last block's share) += dlast + trunk_conv^T dlast
    {
        IflConvArgs a = conv_args(params_t, d[i_trunk], e.dlast, RRDB_NF, Gq(nq), RRDB_DENSE, 0, 0, p4);
        a.bias = nullptr;
        a.res1 = Gq(nq); a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 1.f;
        a.res2 = e.dlast; a.res2_stride = RRDB_NF; a.res2_off = 0; a.beta = 1.f;
        if ((rc = ifl_k_convg(a, B, st))) return rc;
        if ((rc = ifl_k_wgrad(Dq(nq), RRDB_DENSE, 0, RRDB_NF, e.dlast, RRDB_NF, 0, RRDB_NF, d_params + d[i_trunk].w_off, RRDB_NF, 0, 9, p4.g, B, st))) return rc;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(2, 1, BIAS_SLICES), dim3(256), 0, st, e.dlast, (int64_t)0, RRDB_NF, 0, r4, d_params + d[i_trunk].b_off, (int64_t)0);
        IFL_LAUNCH_CHECK();
    }
    // ---- the dense blocks in reverse; after each slice of blocks, its weight and bias gradients (the same convolution of all
    // the slice's blocks in one launch) on the second stream
    EncFork* fk = nullptr;
    if ((rc = enc_fork(&fk))) return rc;
    std::lock_guard<std::mutex> lock(fk->mu);
    const int64_t rec = d[6].w_off - d[1].w_off;                         // floats between consecutive blocks' records
    auto slice_wgrads = [&](int q0, int q1, hipStream_t ws_) -> int {
        const int nz = q1 - q0;
        const float* Dz = e.D + (int64_t)q0 * e.dstride;
        const float* Gz = e.G + (int64_t)q0 * e.dstride;
        const float* Ez = e.E + (int64_t)q0 * e.estride;
        float* dp = d_params + (int64_t)q0 * rec;
        for (int c = 0; c < 4; ++c) {
            const ConvDesc& cd = d[1 + c];
            if (int r = ifl_k_wgrad(Dz, RRDB_DENSE, 0, cd.cin, Gz, RRDB_DENSE, RRDB_NF + c * RRDB_GC, RRDB_GC, dp + cd.w_off, cd.cin, 0, 9,
                                    p4.g, B, ws_, nz, e.dstride, e.dstride, rec)) return r;
            hipLaunchKernelGGL(bias_grad_kernel, dim3(1, nz, BIAS_SLICES), dim3(256), 0, ws_, Gz, e.dstride, RRDB_DENSE, RRDB_NF + c * RRDB_GC, r4,
                               dp + cd.b_off, rec);
            IFL_LAUNCH_CHECK();
        }
        const ConvDesc& cd = d[5];
        if (int r = ifl_k_wgrad(Dz, RRDB_DENSE, 0, RRDB_DENSE, Ez, RRDB_NF, 0, RRDB_NF, dp + cd.w_off, RRDB_DENSE, 0, 9, p4.g, B, ws_,
                                nz, e.dstride, e.estride, rec)) return r;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(2, nz, BIAS_SLICES), dim3(256), 0, ws_, Ez, e.estride, RRDB_NF, 0, r4, dp + cd.b_off, rec);
        IFL_LAUNCH_CHECK();
        return 0;
    };
    static const int slices_env = getenv("IFL_RRDB_SLICES") ? atoi(getenv("IFL_RRDB_SLICES")) : 4;    // (1..8)
    const int nslices = nq >= 12 ? (slices_env < 1 ? 1 : slices_env > 8 ? 8 : slices_env) : 1;
    int hi = nq, slice = 0;
    for (int q = nq - 1; q >= 0; --q) {
        const bool third = q % 3 == 2;
        hipLaunchKernelGGL(rdb_out_bwd_kernel, blocks1d((int64_t)B * p4.h * p4.w * 16), dim3(256), 0, st, Gq(q + 1), Eq(q), Gq(q),
                           third ? Gq(q - 2) : (float*)nullptr, B, p4.h, p4.w);
        IFL_LAUNCH_CHECK();
        for (int c = 4; c >= 0; --c) {                                  // conv5 (c = 4), then conv4 .. conv1
            const ConvDesc& cd = d[1 + 5 * q + c];                       // forward: cd.cin inputs -> cd.cout outputs
            ConvDesc t{cd.cout, cd.cin, cd.w_off, cd.b_off};             // its data gradient: cd.cout inputs -> cd.cin outputs
            const float* src = c == 4 ? Eq(q) : Gq(q);
            IflConvArgs a = conv_args(params_t, t, src, c == 4 ? RRDB_NF : RRDB_DENSE, Gq(q), RRDB_DENSE, 0, 0, p4);
            if (c < 4) a.src_off[0] = RRDB_NF + c * RRDB_GC;
            a.bias = nullptr;
            a.res1 = Gq(q); a.res1_stride = RRDB_DENSE; a.res1_off = 0; a.alpha = 1.f;
            if (c > 0) { a.mask = Dq(q); a.mask_stride = RRDB_DENSE; a.mask_off = 0; a.mask_c0 = cd.cin - RRDB_GC; a.mask_slope = 0.2f; }
            if ((rc = ifl_k_convg(a, B, st))) return rc;
        }
        // a slice boundary: the blocks [q, hi) are complete (the third block of an RRDB also wrote into G[q - 2], whose own
        // chain has not started: slices end on RRDB boundaries, q % 3 == 0)
        const int lo = nq * (nslices - 1 - slice) / nslices / 3 * 3;
        if (q == lo && q < hi) {
            if (nslices == 1) { if ((rc = slice_wgrads(q, hi, st))) return rc; }
            else {
                IFL_HIP(hipEventRecord(fk->ev[slice], st));
                IFL_HIP(hipStreamWaitEvent(fk->aux, fk->ev[slice], 0));
                if ((rc = slice_wgrads(q, hi, fk->aux))) return rc;
            }
            hi = q;
            ++slice;
        }
    }
    if (nslices > 1) {
        IFL_HIP(hipEventRecord(fk->join, fk->aux));
        IFL_HIP(hipStreamWaitEvent(st, fk->join, 0));
    }
    // conv_first: space2depth(x) -> D[0][0,64)
    if ((rc = ifl_k_wgrad(e.s2d, RRDB_NF, 0, RRDB_NF, Gq(0), RRDB_DENSE, 0, RRDB_NF, d_params + d[0].w_off, RRDB_NF, 0, 9, p4.g, B, st))) return rc;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(2, 1, BIAS_SLICES), dim3(256), 0, st, Gq(0), (int64_t)0, RRDB_DENSE, 0, r4, d_params + d[0].b_off, (int64_t)0);
    IFL_LAUNCH_CHECK();
    return 0;
}
