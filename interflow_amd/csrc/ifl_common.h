// Internal definitions shared by the kernels and the C ABI of libinterflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/interflow_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IFL_FT 320            // conditioning channels
#define IFL_HID 64            // coupling-network hidden width
#define IFL_H2 128            // hidden buffer channels: [fAffine 64 | fFeatures 64]
#define IFL_MT 128            // pixels per workgroup tile of the two conv kernels
#define IFL_PP_MAX 304        // max patch pixels (tile + 1-pixel halo) a workgroup stages in LDS
#define IFL_CK 32             // channels per LDS chunk
#define IFL_CS 36             // LDS pixel stride (words) of a 32-channel chunk: CS/4 odd => b128 conflict-free
#define IFL_CS3 40            // same for the 16x16x4 reads of conv3

static inline __host__ __device__ int ifl_roundup(int a, int b) { return (a + b - 1) / b * b; }
static inline __host__ __device__ int ifl_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- tiling of one level's H x W image plane into workgroup tiles ---------------------------------
// The plane is cut into vertical bands of width bw; inside a band pixels are enumerated in raster
// order and cut into chunks of IFL_MT consecutive pixels.  A tile's LDS patch is the rows it touches
// plus one halo row above/below and one halo column left/right.
struct IflGeom {
    int H, W;             // plane size of this level
    int bw;               // band width (last band may be narrower)
    int nbands;
    int tiles_per_band;   // ceil(H*bw / tpx)
    int ntiles;           // nbands * tiles_per_band   (= gridDim.x)
    int tpx;              // pixels per tile: IFL_MT, or IFL_MT / 2 for the generic convolution's half-tile form
};

struct IflTile {          // decoded per workgroup (all fields wave-uniform)
    int x0, bwb;          // band origin / width
    int s, npx;           // first pixel index of the chunk inside the band, pixels in the band
    int y_first;          // first patch row in image coordinates (may be -1)
    int nrows, PW, PP;    // patch rows, patch width (bwb+2), patch pixels
    bool active;
};

static inline __host__ __device__ IflTile ifl_tile(const IflGeom& g, int tile) {
    IflTile t;
    int band = tile / g.tiles_per_band, chunk = tile - band * g.tiles_per_band;
    t.x0 = band * g.bw;
    t.bwb = g.W - t.x0 < g.bw ? g.W - t.x0 : g.bw;
    t.npx = g.H * t.bwb;
    t.s = chunk * g.tpx;
    t.active = t.s < t.npx;
    int e = t.s + g.tpx < t.npx ? t.s + g.tpx : t.npx;
    int r0 = t.s / t.bwb, r1 = (e - 1) / t.bwb;
    t.y_first = r0 - 1;
    t.nrows = r1 - r0 + 3;
    t.PW = t.bwb + 2;
    t.PP = t.nrows * t.PW;
    return t;
}

// ---- per-step packed parameter layout (floats) ----------------------------------------------------
static inline __host__ __device__ int ifl_kz(int C) { return ifl_roundup(C / 2, 8); }       // z1 channels fed to fAffine, padded
static inline __host__ __device__ int ifl_nf(int C) { return ifl_cdiv(C, 8); }              // 16-row output tiles of fFeatures' last conv
static inline __host__ __device__ int ifl_na(int C) { return ifl_cdiv(C / 2, 8); }          // same for fAffine

struct IflStepOff {       // float offsets inside one step's block
    int64_t anic_fwd, anic_inv;
    int64_t w1[2], an1[2], w2[2], an2[2], w3[2], b3[2];     // [0] = fAffine, [1] = fFeatures
    int64_t w1b[2], w2b;      // bf16 copies for the bf16 variant of conv1_ft: first conv over ft (both nets), fFeatures' 1x1
    int64_t w1w[2];           // Winograd F(2x2,3x3) transforms U = G g G^T of the conditioning part of the first conv (both nets)
    int64_t w1x[2];           // Winograd F(4x4,3x3) transforms (6x6 domain) of the same weights
    int64_t w3w[2];           // Winograd F(2x2,3x3) transforms of the last conv (Conv2dZeros), conv3w.hip: [row tile][sub-chunk][pos][q][row][4]
    int64_t total;
};

static inline __host__ __device__ IflStepOff ifl_step_off(int C, int has_coupling) {
    IflStepOff o;
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 63) / 64 * 64; return r; };   // 256-byte aligned fields
    o.anic_fwd = take(2 * C + C * C);
    o.anic_inv = take(2 * C + C * C);
    for (int n = 0; n < 2; ++n) { o.w1[n] = o.an1[n] = o.w2[n] = o.an2[n] = o.w3[n] = o.b3[n] = 0; }
    if (has_coupling) {
        o.w1[0] = take((int64_t)IFL_HID * 9 * (IFL_FT + ifl_kz(C)));
        o.w1[1] = take((int64_t)IFL_HID * 9 * IFL_FT);
        for (int n = 0; n < 2; ++n) o.an1[n] = take(2 * IFL_HID);
        for (int n = 0; n < 2; ++n) o.w2[n] = take(IFL_HID * IFL_HID);
        for (int n = 0; n < 2; ++n) o.an2[n] = take(2 * IFL_HID);
        o.w3[0] = take((int64_t)ifl_na(C) * 16 * 9 * IFL_HID);
        o.w3[1] = take((int64_t)ifl_nf(C) * 16 * 9 * IFL_HID);
        o.b3[0] = take(ifl_na(C) * 32);
        o.b3[1] = take(ifl_nf(C) * 32);
        for (int n = 0; n < 2; ++n) o.w1b[n] = take((int64_t)IFL_HID * 9 * IFL_FT / 2);      // 2 bf16 per float slot
        o.w2b = take(IFL_HID * IFL_HID / 2);
        take(4096);                                                // pad: the weight prefetch ring reads past a field's end
        for (int n = 0; n < 2; ++n) o.w1w[n] = take((int64_t)16 * IFL_HID * IFL_FT);
        take(4096);
        for (int n = 0; n < 2; ++n) o.w1x[n] = take((int64_t)36 * IFL_HID * IFL_FT);
        take(4096);
        o.w3w[0] = take((int64_t)ifl_na(C) * 16 * 16 * IFL_HID);
        o.w3w[1] = take((int64_t)ifl_nf(C) * 16 * 16 * IFL_HID);
    } else { o.w1b[0] = o.w1b[1] = o.w2b = o.w1w[0] = o.w1w[1] = o.w1x[0] = o.w1x[1] = o.w3w[0] = o.w3w[1] = 0; }
    o.total = p;
    return o;
}

// ---- Winograd tiling of a level's plane: 2x2-pixel output tiles, enumerated in raster order over (image, tile row, tile
// column); a workgroup takes 32 consecutive tiles, so only the very last workgroup of a launch can be partly filled. ------
struct IflWGeom { int H, W, TH, TW, ntiles; };      // ntiles = B * TH * TW
static inline IflWGeom ifl_wgeom(int B, int H, int W) {
    IflWGeom g; g.H = H; g.W = W; g.TH = (H + 1) / 2; g.TW = (W + 1) / 2; g.ntiles = B * g.TH * g.TW; return g;
}

// ---- Winograd F(4x4,3x3) tiling: 4x4-pixel output tiles in raster order over (image, tile row, tile column), 32 consecutive
// tiles per workgroup.  The GEMM kernel reads the conditioning features already transformed (V = B^T d B per tile and
// channel), as one 36,864-byte LDS image per (group of 32 tiles, sub-chunk of 8 channels):
// vg[group][40 sub-chunks][36 positions][2 quads][32 tiles][4 channels] (conv1x.hip). ---------------------------------------
// Interpolation points of every F(4x4,3x3) transform in the library: 0, +-a, +-b, infinity with a = 3/4, b = 3/2 (Lavin &
// Gray's 0, +-1, +-2, inf until round 3).  The fp32 rounding error of the 320-channel convolution is half of what the
// integer points give (rms 4.9e-7 against 1.0e-6 of the output's rms, max 8e-7 against 4e-6; tools/exp_wino_points.py), and
// every constant below is exact in fp32.  With A = a^2, B = b^2 and the rows of B^T scaled to a leading coefficient of one:
//   B^T x:  t0 = AB x0 - (A+B) x2 + x4    t1,2 = (x4 - B x2) +- a (x3 - B x1)    t3,4 = (x4 - A x2) +- b (x3 - A x1)
//           t5 = AB x1 - (A+B) x3 + x5
//   G:      [1/(AB), 0, 0];  [1, +-a, A] / (2A(A-B));  [1, +-b, B] / (2B(B-A));  [0, 0, 1]           (ifl_wino4_G)
//   A^T m:  y0 = m0 + (m1+m2) + (m3+m4)   y1 = a (m1-m2) + b (m3-m4)   y2 = A (m1+m2) + B (m3+m4)
//           y3 = a^3 (m1-m2) + b^3 (m3-m4) + m5
#define IFL_W4_a 0.75f
#define IFL_W4_b 1.5f
#define IFL_W4_A 0.5625f      // a^2
#define IFL_W4_B 2.25f        // b^2
#define IFL_W4_AB 1.265625f   // a^2 b^2
#define IFL_W4_ApB 2.8125f    // a^2 + b^2
#define IFL_W4_a3 0.421875f   // a^3
#define IFL_W4_b3 3.375f      // b^3
template <class T>
static inline __host__ __device__ void ifl_wino4_G(T G[6][3]) {
    const T a = (T)0.75, b = (T)1.5, A = a * a, B = b * b;
    const T da = (T)1 / (2 * A * (A - B)), db = (T)1 / (2 * B * (B - A));
    G[0][0] = (T)1 / (A * B); G[0][1] = 0; G[0][2] = 0;
    G[1][0] = da; G[1][1] = a * da; G[1][2] = A * da;
    G[2][0] = da; G[2][1] = -a * da; G[2][2] = A * da;
    G[3][0] = db; G[3][1] = b * db; G[3][2] = B * db;
    G[4][0] = db; G[4][1] = -b * db; G[4][2] = B * db;
    G[5][0] = 0; G[5][1] = 0; G[5][2] = 1;
}
#define IFL_XCH 8             // channels per sub-chunk
struct IflXGeom { int H, W, TH, TW, ntiles; };      // ntiles = B * TH * TW
static inline __host__ __device__ IflXGeom ifl_xgeom(int B, int H, int W) {
    IflXGeom g; g.H = H; g.W = W; g.TH = (H + 3) / 4; g.TW = (W + 3) / 4; g.ntiles = B * g.TH * g.TW; return g;
}
static inline int64_t ifl_vg_floats(int B, int H, int W) {
    IflXGeom g = ifl_xgeom(B, H, W); return (int64_t)((g.ntiles + 31) / 32) * (IFL_FT / IFL_XCH) * 36 * 32 * IFL_XCH;
}

#define IFL_MAX_DEVICES 64    // per-device state tables (streams, events, static scratch addresses)
#define IFL_MAX_COND 16       // conditional steps per level (flow_model.py:42 K=16)
struct IflStepList { int64_t off[IFL_MAX_COND]; };     // float offsets of the steps' parameter blocks

// ---- workspace layout -----------------------------------------------------------------------------
struct IflLevelWs {
    int H, W;                 // plane size at this level
    int C;                    // z channels
    float* ft;                // [B][H+2][W+2][320]  zero ring
    float* vg;                // the same features in the Winograd F(4x4,3x3) domain (IflXGeom, conv1x.hip)
    float* z;                 // [B][H+2][W+2][C]    zero ring
    float* z2;                // a second latent buffer of the same shape: the encode chain's fused ActNorm + 1x1 + conv1_finish
                              //   kernel writes the mapped z here (its neighbours still read the unmapped halo), then the two swap roles
    float* h2;                // [IFL_MAX_COND][B][H+2][W+2][128]  zero ring; one slab per conditional step: [0:64] fAffine's
                              //   conditioning-only pre-activation sums, [64:128] fFeatures' finished hidden activations
    float* ha;                // [B][H+2][W+2][64]  zero ring; fAffine's finished hidden activations of the step in flight
    float* sf;                // [IFL_MAX_COND][B][H+2][W+2][2C]  fFeatures' (shift c, scale c) pairs of every conditional step
    int64_t h2_step, sf_step; // floats per step slab
    int ld_f0;                // first ldpart slot of this level's fFeatures sums: [step][tile]
    int ld_step;              // slots reserved per conditional step (the larger of the direct kernels' tiles and conv3w's regions x row tiles)
    IflGeom geom;
};

struct IflWs {
    int B, H, W;
    IflLevelWs lv[IFL_MAX_LEVELS];
    float* ldpart;            // [B][ld_slots] per-tile log-det partial sums: the chain's (one per conditional step and tile, in the
                              //   order the pass runs them), then per level fFeatures' [step][tile] (written by phase 1, kept)
    int ld_slots;
    int64_t bytes;
};

// ---- generic convolution (convg.hip) ------------------------------------------------------------------------------
// Input channels = up to two sources read one after the other (each a channel range, a multiple of 32 wide, of a ringed
// channels-last buffer); output = `cout` channels (multiple of 32) written at channel out_off of a ringed buffer.
//   v = conv + bias;  act (1 LeakyReLU 0.2, 2 ReLU);  if res1: v = v*alpha + res1;  if res2: v = v*beta + res2;  if mask: see below.
// Weights: [group of 32 outputs][chunk of 32 inputs][tap][kq][lane half h][lane i][e] =
//          W[out = group*32 + i][in = chunk*32 + kq*8 + h*4 + e][tap].
struct IflConvArgs {
    const float* src[2]; int src_stride[2], src_off[2], src_chunks[2]; int nsrc, nchunks;
    const float* w; const float* bias;
    float* out; int out_stride, out_off; int cout, taps, act;
    const float* res1; int res1_stride, res1_off; float alpha;
    const float* res2; int res2_stride, res2_off; float beta;
    IflGeom g;
    // batched form: nz independent problems of one shape in one launch (blockIdx.y = z * Bz + image), operands z_* floats apart
    int nz, Bz;
    int64_t z_src[2], z_w, z_bias, z_out, z_res1;
    // last: output channels c >= mask_c0 are multiplied by LeakyReLU's derivative at the activation mask[pixel][mask_off + c]
    // (v if it is > 0, else v * mask_slope): the data gradient of a dense block's convolution completes the gradient wrt
    // the newest 32 channels it reads, which passes through that activation before the next convolution uses it
    const float* mask; int mask_stride, mask_off, mask_c0; float mask_slope;
};
int ifl_k_convg(const IflConvArgs& a, int B, hipStream_t st);
int ifl_k_wgrad(const float* X, int xs, int xoff, int cin, const float* DY, int dys, int dyoff, int cout, float* dW,
                int cin_total, int ci0, int taps, const IflGeom& g, int B, hipStream_t st, int nz = 1, int64_t xz = 0, int64_t dyz = 0,
                int64_t dwz = 0);
IflGeom ifl_choose_geom(int H, int W, int tpx = IFL_MT);

int ifl_ws_layout(void* base, int B, int H, int W, IflWs* out);     // base may be NULL (size query)
void ifl_set_error(const char* fmt, ...);

#define IFL_CHECK_ARG(cond, code, ...) do { if (!(cond)) { ifl_set_error(__VA_ARGS__); return (code); } } while (0)
#define IFL_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ifl_set_error("%s: %s", #call, hipGetErrorString(e_)); return (int)e_; } } while (0)
#define IFL_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ifl_set_error("kernel launch: %s", hipGetErrorString(e_)); return (int)e_; } } while (0)

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------------
// layout.hip
int ifl_k_nchw_to_ring(const float* in, float* out, int B, int C, int H, int W, hipStream_t st);                 // NCHW -> ringed NHWC
int ifl_k_ring_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t st);
int ifl_k_squeeze_in(const float* x, float* zr, int B, int C, int H, int W, hipStream_t st);                     // NCHW [B,C,H,W] -> ringed NHWC [B,H/2,W/2,4C]
int ifl_k_unsqueeze_out(const float* zr, float* x, int B, int C, int H, int W, hipStream_t st);                  // inverse
int ifl_k_squeeze_ring(const float* a, float* b, int B, int C, int H, int W, hipStream_t st);                    // ringed [H,W,C] -> ringed [H/2,W/2,4C]
int ifl_k_unsqueeze_ring(const float* b, float* a, int B, int C, int H, int W, hipStream_t st);
int ifl_k_squeeze_nchw(const float* in, float* out, int B, int C, int H, int W, int f, int reverse, hipStream_t st);
int ifl_k_quantize_u8(const float* x, uint8_t* out, int B, int C, int64_t P, hipStream_t st);
int ifl_k_resize_sep(const float* x, float* out, int N, int Hin, int Win, int Hout, int Wout, const int* idxH, const float* wH, int TH,
                     const int* idxW, const float* wW, int TW, hipStream_t st);
// pointwise.hip
// in place on ringed NHWC; sf_next (decode only, may be NULL): also apply the next layer's z = (z + shiftFt) * scaleFt
int ifl_k_anic(float* zr, const float* prm, int B, int C, int H, int W, int inverse, const float* sf_next, hipStream_t st);
// ld_out = ld_in + constant + sum ldpart[0, slots_used) + fsign * (sum ldpart[fa0, fa0+fan) + sum ldpart[fb0, fb0+fbn))
int ifl_k_ld_finalize(const float* ldpart, int slots_used, int ld_slots, const float* ld_in, float* ld_out,
                      double constant, double fsign, int fa0, int fan, int fb0, int fbn, int B, hipStream_t st);
int ifl_k_logp(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* scratch,
               float* out, int B, int C, int HW, hipStream_t st);
int ifl_k_logp_static(const float* z, const float* mean, const float* cov_inv, float logdet_cov, float* out, int B, int C,
                      int HW, hipStream_t st);                  // scratch = the library's static device array
int ifl_k_lerp(const float* za, const float* zb, const float* wa, const float* wb, float* out, int B,
               int64_t n, hipStream_t st);
// conv1.hip
int ifl_k_conv1_ft(const float* ft, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                   int nsteps, const IflStepOff& off, const IflGeom& g, int B, int level, int bf16, hipStream_t st);
int ifl_k_conv1_ft_wino(const float* ft, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                        int nsteps, const IflStepOff& off, int B, int H, int W, int level, hipStream_t st);
// conv1x.hip
int ifl_k_wino4_input(const float* ft_ring, float* vg, int B, int H, int W, hipStream_t st);                    // ringed NHWC -> vg
int ifl_k_conv1_ft_wino4(const float* vg, float* h2all, int64_t h2_step_stride, const float* params, const IflStepList& steps,
                         int nsteps, const IflStepOff& off, int B, int H, int W, int level, hipStream_t st, int raw_both = 0);
int ifl_k_conv1_finish_pre(const float* z_in, float* z_out, const float* anic_fwd, const float* h2, float* ha, const float* prm,
                           const IflStepOff& off, const IflGeom& g, int B, int C, hipStream_t st, const float* sf_inverse = nullptr);
int ifl_k_conv1_finish(const float* z, const float* h2, float* ha, const float* prm, const IflStepOff& off,
                       const IflGeom& g, int B, int C, int bf16, hipStream_t st);
// conv3.hip:  mode: 0 = encode (fAffine's conv, both couplings), 2 = decode (fAffine's conv and coupling)
int ifl_k_conv3(const float* sf, const float* ha, float* z, float* ldpart, int ld_slots, int slot0, const float* prm,
                const IflStepOff& off, const IflGeom& g, int B, int C, int mode, float eps, hipStream_t st);
int ifl_k_conv3f(const float* h2all, int64_t h2_step, float* sfall, int64_t sf_step, float* ldpart, int ld_slots, int slot0,
                 const float* params, const IflStepList& steps, int nsteps, const IflStepOff& off, const IflGeom& g, int B, int C,
                 float eps, hipStream_t st);
int ifl_k_fapply(const float* sf, float* z, int B, int C, int H, int W, hipStream_t st);
// conv3w.hip: the same operators by Winograd F(2x2,3x3); mode 0 = fFeatures (nsteps steps), 1 = encode chain, 2 = decode chain
int ifl_w3_wgs(int H, int W);
int ifl_w3_slots(int C, int H, int W, int f);
int ifl_k_conv3w(int mode, const float* src, int64_t src_step, int pstride, float* sf, int64_t sf_step, float* z,
                 float* ldpart, int ld_slots, int slot0, const float* params, const IflStepList& steps,
                 int nsteps, const IflStepOff& off, int B, int H, int W, int C, float eps, int bf16, hipStream_t st);
