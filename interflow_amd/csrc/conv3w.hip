// conv3w.hip -- the coupling networks' last layer by Winograd F(2x2,3x3) minimal filtering on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32), fused with the affine coupling.  Same operator as conv3.hip (the direct form), 2.25x fewer
// multiplications, same fp32 arithmetic type; F(2x2,3x3)'s transforms only add / subtract (constants +-1, 1/2 folded into the
// weights on the host in float64), so its rounding behaviour is that of the direct form.
//
// Replaces (reference interflow/models/modules):
//   flow.Conv2dZeros 3x3 (64 -> 2C for fFeatures, 64 -> C for fAffine): (conv + bias) * exp(3*logs)      flow.py:67-82
//   shift = h[:,0::2], scale = sigmoid(h[:,1::2] + 2) + eps                                              FlowAffineCouplingsAblation.py:219-240
//   the affine coupling of both directions and the sums of log(scale)                                      :167-207, 216-217
//
//   Y = A^T [ sum_ci U[row][ci] (.) V[ci] ] A,   U = G g G^T (host, float64 -> fp32, pack.py),   V = B^T d B
// for every 2x2 output tile and its 4x4 input window d; for each of the 16 positions of the transformed domain the sum over the
// 64 hidden channels is a GEMM  M[row][tile] = U[row][ci] V[ci][tile].
//
// Workgroup = one 16 x 16 pixel region of one image x ONE 16-row tile of the output (rows are permuted on the host exactly as
// for conv3.hip: a lane's four accumulator registers are (shift c, shift c+1, scale c, scale c+1)); 4 waves = the region's four
// 4 x 16 pixel strips; lane = (tile j of the strip's 2 x 8 tiles, channel quad q).  Per 16-channel sub-chunk:
//   * the region's 18 x 18 pixel patch of the hidden activations and the row tile's transformed weights are staged in LDS
//     (prefetched into registers one sub-chunk ahead);
//   * every lane reads its tile's 4 x 4 window for its 4 channels from the patch, transforms it in registers -- the 16 results
//     ARE the B operands of its 16 positions -- and multiplies them with the A fragments read from LDS: 64 MFMAs per wave.
// All 16 positions of a tile sit in one lane's accumulators, so the output transform is register arithmetic and the lane ends
// with its tile's four pixels in the accumulator layout conv3.hip's epilogue uses.
// The other row tiles of a network are other workgroups (blockIdx.z): they repeat the staging and the input transform
// (vector ALU, beside the matrix cores) instead of holding 16 x NT accumulators per lane.
#include "ifl_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16w(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// torch.sigmoid(h + 2.) + eps and log(scale), on the hardware's exp2 / rcp / log2 with the argument reduction spelled out: the
// library forms cost ~32 instructions per element (range checks, denormal paths, the IEEE division sequence), 40 % of this
// kernel's epilogue; these take 13 and stay within ~1 ulp (exp: x log2(e) split as n + r with a two-term log2(e), 2^r by
// v_exp_f32 on |r| <= 1/2; reciprocal: v_rcp_f32 + two Newton steps; log: v_log_f32 x ln 2 on arguments in [eps, 1 + eps]).
// -DIFL_LIBM_SCALE: the library forms.
__device__ __forceinline__ float scale_of(float h, float eps) {
#ifdef IFL_LIBM_SCALE
    return 1.0f / (1.0f + expf(-(h + 2.0f))) + eps;
#else
    float t = -(h + 2.0f);
    t = __builtin_fminf(__builtin_fmaxf(t, -87.0f), 88.0f);            // exp(88) is finite; sigmoid is 0 / 1 to the last bit beyond
    const float n = __builtin_rintf(t * 1.44269502162933349609375f);
    float r = __builtin_fmaf(t, 1.44269502162933349609375f, -n);
    r = __builtin_fmaf(t, 1.925963033500011e-8f, r);
    const float e = __builtin_ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
    const float d = 1.0f + e;
    const float x0 = __builtin_amdgcn_rcpf(d);
    const float x1 = __builtin_fmaf(__builtin_fmaf(-d, x0, 1.0f), x0, x0);
    const float x2 = __builtin_fmaf(__builtin_fmaf(-d, x1, 1.0f), x1, x1);      // second step: the quotient to the last bit (as the division)
    return x2 + eps;
#endif
}
__device__ __forceinline__ float log_scale(float s) {
#ifdef IFL_LIBM_SCALE
    return logf(s);
#else
    return __builtin_amdgcn_logf(s) * 0.693147182464599609375f;       // s in [eps, 1 + eps]: no denormals, no specials
#endif
}

enum { W3_F = 0, W3_ENC = 1, W3_DEC = 2 };

constexpr int RGN = 16;                       // region edge in pixels
constexpr int PWD = RGN + 2;                  // patch edge
constexpr int PPX = PWD * PWD;                // 324 patch pixels
constexpr int PST = 20;                       // LDS words per patch pixel: 16 channels + 4
constexpr int PRS = 384;                      // LDS words per patch row (18 x 20 = 360, padded to a multiple of 32): with tiles 2 pixels
                                              // = 40 words apart and the rows 0 mod 32 words, the 16-byte window reads of a wave's
                                              // 2 x 8 tiles fall into distinct banks for every lane group of ds_read_b128
constexpr int NPF = (PPX * 4 + 255) / 256;    // float4 patch loads per thread per sub-chunk (6)
constexpr int USZ = 16 * 256;                 // floats of one row tile's transformed weights per sub-chunk: [pos][q][row][4]

// B^T x for four values, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__device__ __forceinline__ void bt4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    const f32x4 t0 = a - c, t1 = b + c, t2 = c - b, t3 = b - d;
    a = t0; b = t1; c = t2; d = t3;
}

// src: the 64 hidden channels (pixel stride `pstride` floats) of a ringed channels-last plane; step-dependent pointers are
// resolved from blockIdx.z by the caller-provided tables (conv3f) or passed directly (chain).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 to_bf16(f32x4 v) {
    bf16x4w b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (__bf16)v[e];
    return __builtin_bit_cast(s16x4, b);
}

// BF16: the products run on v_mfma_f32_16x16x16_bf16 -- a lane's four k-steps (channels 4q .. 4q+3 of the sub-chunk) are one
// instruction -- with both operands (transformed weights U, transformed activations V) rounded to bf16, fp32 accumulate; the
// transforms, bias / exp / sigmoid and the coupling stay fp32 (BASELINE.json configs[4]).
// NTW: row tiles per workgroup.  2 (fFeatures at C = 12: both of its tiles): the patch is staged and transformed once for
// both, 32 accumulators per lane, two workgroups per CU instead of three.
template <int MODE, bool BF16, int NTW = 1>
__global__ __launch_bounds__(256, (NTW == 1 && !BF16) ? 3 : 2)        // (bf16: the operand conversions need registers; at 3 workgroups per CU it spilled 230 bytes per lane)
void conv3w_kernel(const float* __restrict__ srcAll, int64_t src_step, int pstride, float* sfAll, int64_t sf_step,
                   float* __restrict__ zP, float* __restrict__ ldpart, int ld_slots, int slot0,
                   const float* __restrict__ params, IflStepList steps, int64_t w3w_off, int64_t b3_off, int NT, int NA,
                   int H, int W, int C, float eps)
{
    __shared__ __attribute__((aligned(16))) float patch[PWD * PRS];
    __shared__ __attribute__((aligned(16))) float us[NTW * USZ];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int b = blockIdx.y;
    const int step = blockIdx.z / (NT / NTW), rt0 = (blockIdx.z - step * (NT / NTW)) * NTW;      // row tiles rt0 .. rt0 + NTW - 1
    const int gx = (W + RGN - 1) / RGN;
    const int wgy = blockIdx.x / gx, wgx = blockIdx.x - wgy * gx;
    const int W2 = W + 2;
    const float* prm = params + steps.off[step];
    const float* src = srcAll + (int64_t)step * src_step + (int64_t)b * (H + 2) * W2 * pstride;
    const float* wsrc = prm + w3w_off + (int64_t)rt0 * (4 * USZ);                  // tile rt0 + tw: + tw * 4 * USZ

    // ---- staging plan: per-thread source offsets of the patch (same for every sub-chunk) ----
    int poff[NPF];                            // float offset inside the image's plane, or -1 (outside the ringed plane / past the patch)
    int pdst[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int idx = tid + k * 256, px = idx >> 2, qd = idx & 3;
        const int pr = px / PWD, pc = px - pr * PWD;
        const int gy_ = wgy * RGN + pr, gx_ = wgx * RGN + pc;          // ringed coordinates
        const bool ok = px < PPX && gy_ <= H + 1 && gx_ <= W + 1;
        poff[k] = ok ? (gy_ * W2 + gx_) * pstride + qd * 4 : -1;
        pdst[k] = px < PPX ? pr * PRS + pc * PST + qd * 4 : -1;
    }
    f32x4 pre[NPF], upre[NTW][4];
    auto prefetch = [&](int s) {
#pragma unroll
        for (int k = 0; k < NPF; ++k)
#ifdef IFL_EXP3_NOSTAGE
            pre[k] = f32x4{1.f * poff[k], 0.5f, 0.25f, 1.f * s};
#else
            pre[k] = poff[k] >= 0 ? *(const f32x4*)(src + poff[k] + s * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
#ifdef IFL_EXP3_NOSTAGE
            for (int k = 0; k < 4; ++k) upre[tw][k] = f32x4{0.001f * tid, 0.5f, 0.25f, 1.f * s};
#else
            for (int k = 0; k < 4; ++k) upre[tw][k] = *(const f32x4*)(wsrc + tw * (4 * USZ) + s * USZ + (tid + k * 256) * 4);
#endif
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (pdst[k] >= 0) *(f32x4*)&patch[pdst[k]] = pre[k];
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (BF16) *(s16x4*)((short*)us + tw * USZ + (tid + k * 256) * 4) = to_bf16(upre[tw][k]);      // rounded once here, not by each of the 4 waves that read it
                else *(f32x4*)&us[tw * USZ + (tid + k * 256) * 4] = upre[tw][k];
            }
    };

    // ---- this lane's tile ----
    const int ty = j >> 3, tx = j & 7;
    const int py0 = wgy * RGN + 4 * wave + 2 * ty, px0 = wgx * RGN + 2 * tx;              // the tile's pixel (0, 0), plane coordinates
    const bool wave_live = (wgy * RGN + 4 * wave) < H;                                    // uniform per wave
    const int wbase = (4 * wave + 2 * ty) * PRS + 2 * tx * PST + 4 * q;                   // window pixel (r, c): + r * PRS + c * PST
    const int abase = q * 64 + j * 4;                                                     // A fragment of position p: + p * 256

    f32x4 accs[NTW][16];
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
        for (int p = 0; p < 16; ++p) accs[tw][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    prefetch(0);
    for (int s = 0; s < 4; ++s) {
        if (s) __syncthreads();               // the previous sub-chunk's LDS reads are over
        commit();
        __syncthreads();
        if (s + 1 < 4) prefetch(s + 1);
        if (wave_live) {
            f32x4 v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[r][c] = *(const f32x4*)&patch[wbase + r * PRS + c * PST];
#ifndef IFL_EXP3_NOXFORM
#pragma unroll
            for (int c = 0; c < 4; ++c) bt4(v[0][c], v[1][c], v[2][c], v[3][c]);          // along the rows: v[i][c]
#pragma unroll
            for (int i = 0; i < 4; ++i) bt4(v[i][0], v[i][1], v[i][2], v[i][3]);          // along the columns: v[i][jc]
#endif
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int tw = 0; tw < NTW; ++tw) {
                    if (BF16) {
                        const s16x4 ab = *(const s16x4*)((const short*)us + tw * USZ + abase + p * 256);
                        accs[tw][p] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, to_bf16(v[p >> 2][p & 3]), accs[tw][p], 0, 0, 0);
                    } else {
                        const f32x4 a = *(const f32x4*)&us[tw * USZ + abase + p * 256];
#pragma unroll
                        for (int e = 0; e < 4; ++e) accs[tw][p] = mfma16w(a[e], v[p >> 2][p & 3][e], accs[tw][p]);
                    }
                }
        }
    }

    const int nwg = gridDim.x;
#ifdef IFL_EXP3_NOEPI
    if (MODE == W3_F) {
        float sum = 0.f;
#pragma unroll
        for (int tw = 0; tw < NTW; ++tw)
#pragma unroll
            for (int p = 0; p < 16; ++p) sum += accs[tw][p][0] + accs[tw][p][1] + accs[tw][p][2] + accs[tw][p][3];
        if (sum == 12345.678f) sfAll[tid] = sum;
        return;
    }
#endif
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
    const int rt = rt0 + tw;
    const f32x4 (&acc)[16] = accs[tw];
    // ---- output transform: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1] ----
    f32x4 y[2][2];
    {
        f32x4 t[2][4];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            t[0][jc] = acc[jc] + acc[4 + jc] + acc[8 + jc];
            t[1][jc] = acc[4 + jc] - acc[8 + jc] - acc[12 + jc];
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            y[dy][0] = t[dy][0] + t[dy][1] + t[dy][2];
            y[dy][1] = t[dy][1] - t[dy][2] - t[dy][3];
        }
    }

    // ---- epilogue: bias, exp(3 logs), sigmoid, coupling (as conv3.hip) ----
    const int Ch = C / 2;
    const float* b3 = prm + b3_off;
    const int64_t ring0 = (int64_t)b * (H + 2) * W2;
    float ld = 0.f;
    const int sigma = 8 * rt + 2 * q;
    if (MODE == W3_F) {
        if (wave_live && sigma < C) {
            const int c = (sigma + Ch) % C;
            const f32x4 bF = *(const f32x4*)(b3 + rt * 32 + q * 4), eF = *(const f32x4*)(b3 + rt * 32 + 16 + q * 4);
            float* sf = sfAll + (int64_t)step * sf_step;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    if (py0 + dy >= H || px0 + dx >= W) continue;
                    const int64_t op = ring0 + (int64_t)(py0 + dy + 1) * W2 + px0 + dx + 1;
                    const f32x4 f = y[dy][dx];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        o[2 * e] = (f[e] + bF[e]) * eF[e];                                        // shift
                        o[2 * e + 1] = scale_of((f[2 + e] + bF[2 + e]) * eF[2 + e], eps);           // scale
                        ld += log_scale(o[2 * e + 1]);
                    }
                    *(f32x4*)(sf + op * (2 * C) + 2 * c) = o;
                }
        }
    } else if (wave_live) {
        const float* sfP = sfAll + (int64_t)step * sf_step;
        // virtual tile rt: fAffine's coupling on z2 channel pair Ch + sigma (and, encode, fFeatures' on it);
        // virtual tile NA + rt (encode only): fFeatures' coupling on the pair it covers (z1 channels, or z2 ones past fAffine's tiles)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            if (part == 1 && MODE != W3_ENC) break;
            const int sg = part == 0 ? sigma : 8 * (NA + rt) + 2 * q;
            const bool vA = part == 0 && sigma < Ch;
            if (MODE == W3_ENC ? sg >= C : !vA) continue;
            const int c = MODE == W3_ENC ? (sg + Ch) % C : Ch + sg;
            f32x4 bA = f32x4{0.f, 0.f, 0.f, 0.f}, eA = bA;
            if (vA) { bA = *(const f32x4*)(b3 + rt * 32 + q * 4); eA = *(const f32x4*)(b3 + rt * 32 + 16 + q * 4); }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    if (py0 + dy >= H || px0 + dx >= W) continue;
                    const int64_t op = ring0 + (int64_t)(py0 + dy + 1) * W2 + px0 + dx + 1;
                    float* zp = zP + op * C + c;
                    f32x2 z = *(const f32x2*)zp;
                    if (vA) {
                        const f32x4 a = y[dy][dx];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float sh = (a[e] + bA[e]) * eA[e];
                            const float sc = scale_of((a[2 + e] + bA[2 + e]) * eA[2 + e], eps);
                            if (MODE == W3_ENC) {
                                z[e] = z[e] / sc;
                                z[e] = z[e] - sh;
                                ld -= log_scale(sc);
                            } else {
                                z[e] = z[e] + sh;
                                z[e] = z[e] * sc;
                                ld += log_scale(sc);
                            }
                        }
                    }
                    if (MODE == W3_ENC) {
                        const f32x4 f = *(const f32x4*)(sfP + op * (2 * C) + 2 * c);      // shift c, scale c, shift c+1, scale c+1
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            z[e] = z[e] / f[2 * e + 1];
                            z[e] = z[e] - f[2 * e];
                        }
                    }
                    *(f32x2*)zp = z;
                }
        }
    }
    // ---- the workgroup's sum of log(scale) -> its slot: [step][row tile][region] ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ld += __shfl_xor(ld, o, 64);
    if (lane == 0) red[wave] = ld;
    __syncthreads();
    if (tid == 0)
        ldpart[(int64_t)b * ld_slots + slot0 + (step * NT + rt) * nwg + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (NTW > 1) __syncthreads();             // red[] is reused by the next tile
    }
}

}  // namespace

int ifl_w3_wgs(int H, int W) { return ((H + RGN - 1) / RGN) * ((W + RGN - 1) / RGN); }
// log-det slots one step writes: fFeatures (f = 1) or the chain's fAffine (f = 0)
int ifl_w3_slots(int C, int H, int W, int f) { return (f ? ifl_nf(C) : ifl_na(C)) * ifl_w3_wgs(H, W); }

// mode 0: fFeatures for `nsteps` steps (writes sf and its log-det slots [step][row tile][region] from slot0); 1: encode chain;
// 2: decode chain (nsteps = 1, the step's parameter block in steps.off[0]; ifl_w3_slots(C, H, W, 0) slots from slot0)
int ifl_k_conv3w(int mode, const float* src, int64_t src_step, int pstride, float* sf, int64_t sf_step, float* z,
                 float* ldpart, int ld_slots, int slot0, const float* params, const IflStepList& steps,
                 int nsteps, const IflStepOff& off, int B, int H, int W, int C, float eps, int bf16, hipStream_t st)
{
    if (nsteps <= 0) return 0;
    if (C != 12 && C != 48) { ifl_set_error("conv3 (Winograd): C=%d not supported (12 or 48)", C); return IFL_ENOTSUP; }
    const int NA = ifl_na(C), NF = ifl_nf(C);
    const int NT = mode == W3_F ? NF : NA;
    const int nwg = ifl_w3_wgs(H, W);
    if ((int64_t)(H + 2) * (W + 2) * pstride >= (int64_t)1 << 31) { ifl_set_error("conv3 (Winograd): plane too large for 32-bit offsets"); return IFL_ENOTSUP; }
    // fFeatures, fp32: two row tiles per workgroup (C = 12: both; C = 48: three pairs; IFL_CONV3_F_PAIR=0: one each; +0.7 % and +0.4 %)
    static const int f_pair = [] { const char* e = getenv("IFL_CONV3_F_PAIR"); return e ? atoi(e) : 3; }();     // bit 0: C = 12, bit 1: C = 48
    const int ntw = (mode == W3_F && !bf16 && NT % 2 == 0 && (f_pair & (NT == 2 ? 1 : 2))) ? 2 : 1;      // (bf16 with two tiles spills: 4.4 against 3.1 ms)
    dim3 grid(nwg, B, nsteps * (NT / ntw)), block(256);
    const int64_t w3w = off.w3w[mode == W3_F ? 1 : 0], b3 = off.b3[mode == W3_F ? 1 : 0];
#define IFL_W3_LAUNCH(M, BF, ...) hipLaunchKernelGGL((conv3w_kernel<M, BF, ##__VA_ARGS__>), grid, block, 0, st, src, src_step, pstride, sf, sf_step, z, ldpart, \
                                                ld_slots, slot0, params, steps, w3w, b3, NT, NA, H, W, C, eps)
    if (mode == W3_F && ntw == 2) IFL_W3_LAUNCH(W3_F, false, 2);
    else if (mode == W3_F) { if (bf16) IFL_W3_LAUNCH(W3_F, true); else IFL_W3_LAUNCH(W3_F, false); }
    else if (mode == W3_ENC) { if (bf16) IFL_W3_LAUNCH(W3_ENC, true); else IFL_W3_LAUNCH(W3_ENC, false); }
    else if (mode == W3_DEC) { if (bf16) IFL_W3_LAUNCH(W3_DEC, true); else IFL_W3_LAUNCH(W3_DEC, false); }
    else { ifl_set_error("conv3 (Winograd): mode %d", mode); return IFL_EINVAL; }
#undef IFL_W3_LAUNCH
    IFL_LAUNCH_CHECK();
    return 0;
}
