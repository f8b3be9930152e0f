// train_fused.hip -- the gradient-enabled pass of one flow level (encode direction) as two library calls:
// ifl_train_level_fwd runs the level's FlowSteps (FlowStep.py:114-151) and keeps what the backward needs in caller-owned
// buffers; ifl_train_level_bwd walks the steps in reverse and writes the input gradient, dense weight gradients and
// per-block partial sums of the per-channel parameter gradients.  Between the two calls there is no host work per layer:
// the Python side (interflow_amd/train_fused.py) only builds the descriptors once per shape, adds the partial sums and
// routes the gradients to the parameters.
//
// Per conditional step (FlowAffineCouplingsAblation.py:159-256), for each of its two couplings in the order the reference
// applies them:  [z1 -> ring]  conv 3x3 (z1 | ft -> 64)  ActNorm+ReLU  conv 1x1  ActNorm+ReLU  Conv2dZeros 3x3  coupling.
// The convolutions are convg.hip's kernels (forward, data gradient with transposed weights, weight gradient), the per-pixel
// operators train_glue.hip's.  New here:
//   affine_px   z' = M z + c per pixel: ActNorm2d followed by InvertibleConv1x1 (FlowActNorms.py:87-92,
//               Permutations.py:114-118) with M = W diag(exp(logs)), c = M bias (composed by the caller, who also owns the
//               chain rule back to bias / logs / the LU factors); backward: dz = M^T dz', per-block sums of dz' (x) z and dz'
//   ring_pack / ring_unpack_add   channels [0, ch) of a contiguous z <-> a ringed 32-channel plane (the coupling network's
//               z1 input and its gradient)
//   pack table  one launch that re-packs every convolution weight of a level from the parameters' own layout
//               [Cout][Cin][k][k] into the kernels' fragment order (forward and transposed/flipped for the data gradients),
//               pads the Conv2dZeros biases, and -- after the backward -- un-pads the dense weight gradients into the
//               parameters' layout
#include "ifl_common.h"
#include "../../include/interflow_hip.h"
#include <mutex>

int ifl_k_an_relu_fwd(const float* x, int xs, float* y, const float* bias, const float* logs, int B, int h, int w, int C, hipStream_t stream,
                      int nz = 1, int64_t xz = 0, int64_t yz = 0, int64_t pz = 0);
int ifl_k_an_relu_bwd(const float* x, int xs, const float* dy, float* dx, const float* bias, const float* logs, float* partials,
                      int B, int h, int w, int C, hipStream_t stream, int nz = 1, int64_t xz = 0, int64_t gz = 0, int64_t dxz = 0,
                      int64_t pz = 0, int64_t partz = 0);
int ifl_k_coupling_bwd(const float* z, const float* dzout, const float* dld, const float* h_ring, int hstride, const float* logs3,
                       float eps, int branch, int c0, int n, float* dz, float* dh_ring, float* dlogs_part, float* dbias_part, int B,
                       int h, int w, int C, hipStream_t stream);

#define IFL_TRY(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

namespace {

constexpr int APX = 32;                       // pixels per wave (= per block, = per partial-sum slot) of the affine_px kernels

// The per-pixel affine maps are small GEMMs ([pixels x C] by [C x C]) on v_mfma_f32_16x16x4_f32: A lane (row l & 15, k-slot
// l >> 4), B lane (column l & 15, k-slot l >> 4), D lane (column l & 15, rows 4 (l >> 4) + r).  The k index a (step, slot)
// pair stands for is free as long as A and B agree: for C = 48 slot g takes k = 16 S + 4 g + e (one float4 per lane and S),
// for C = 12 k = 3 g + s.  One wave = 32 pixels (two column tiles).  (The first form, one thread per pixel with the matrix
// in LDS, ran a group of four 40x40 samples on 25-50 workgroups: 19 us forward, 46 us backward per flow step.)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int C>
struct AffK {                                 // the k indices of lane group g: NK of them, k(q) for q = 0 .. NK-1
    static constexpr int NK = C / 4;
    __device__ static int k(int q, int g) { return C == 48 ? 16 * (q >> 2) + 4 * g + (q & 3) : 3 * g + q; }
};

// v[q] = row[k(q, g)] (row: C contiguous floats; zero when !ok)
template <int C>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int g, bool ok, float (&v)[C / 4]) {
    if constexpr (C == 48) {
#pragma unroll
        for (int S = 0; S < 3; ++S) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (ok) t = *(const f32x4*)(row + 16 * S + 4 * g);
            v[4 * S] = t[0]; v[4 * S + 1] = t[1]; v[4 * S + 2] = t[2]; v[4 * S + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) v[q] = ok ? row[3 * g + q] : 0.f;
    }
}

// zo[p][i] = sum_j M[i][j] z[p][j] + c[i]
template <int C>
__global__ __launch_bounds__(64)
void affine_px_fwd_kernel(const float* __restrict__ z, float* __restrict__ zo, const float* __restrict__ M,
                          const float* __restrict__ c, int64_t npx)
{
    constexpr int IT = (C + 15) / 16, NK = C / 4;
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * APX;
    float am[IT][NK], bz[2][NK];
#pragma unroll
    for (int it = 0; it < IT; ++it) load_row<C>(M + (int64_t)(it * 16 + n) * C, g, it * 16 + n < C, am[it]);
#pragma unroll
    for (int t = 0; t < 2; ++t) load_row<C>(z + (p0 + t * 16 + n) * C, g, p0 + t * 16 + n < npx, bz[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool rows = it * 16 + 4 * g < C;               // (C = 12: rows 12..15 of the only tile do not exist)
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (rows) d = *(const f32x4*)(c + it * 16 + 4 * g);
#pragma unroll
            for (int q = 0; q < NK; ++q) d = mfma16(am[it][q], bz[t][q], d);     // (every lane takes part: no divergence here)
            if (rows && p0 + t * 16 + n < npx) *(f32x4*)(zo + (p0 + t * 16 + n) * C + it * 16 + 4 * g) = d;
        }
}

// dz = M^T dzo;  part[block][i * C + j] = sum_p dzo[p][i] z[p][j],  part[block][C * C + i] = sum_p dzo[p][i]
template <int C>
__global__ __launch_bounds__(64)
void affine_px_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dzo, const float* __restrict__ M,
                          float* __restrict__ dz, float* __restrict__ part, int64_t npx)
{
    constexpr int IT = (C + 15) / 16, NK = C / 4;
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * APX;
    // ---- dz[p][j] = sum_i M[i][j] dzo[p][i]: A = M^T (row j, k = i), B = dzo (column p, k = i)
    {
        float bg[2][NK];
#pragma unroll
        for (int t = 0; t < 2; ++t) load_row<C>(dzo + (p0 + t * 16 + n) * C, g, p0 + t * 16 + n < npx, bg[t]);
#pragma unroll
        for (int jt = 0; jt < IT; ++jt) {
            float at[NK];
#pragma unroll
            for (int q = 0; q < NK; ++q) at[q] = jt * 16 + n < C ? M[(int64_t)AffK<C>::k(q, g) * C + jt * 16 + n] : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NK; ++q) d = mfma16(at[q], bg[t][q], d);
                if (jt * 16 + 4 * g < C && p0 + t * 16 + n < npx) *(f32x4*)(dz + (p0 + t * 16 + n) * C + jt * 16 + 4 * g) = d;
            }
        }
    }
    // ---- part[i][j] = sum_p dzo[p][i] z[p][j]: k = pixel p0 + 4 s + g; A lane: dzo[p][it*16 + n], B lane: z[p][jt*16 + n]
    f32x4 acc[IT][IT];
    float sa[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        sa[it] = 0.f;
#pragma unroll
        for (int jt = 0; jt < IT; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < APX / 4; ++s) {
        const int64_t p = p0 + 4 * s + g;
        const bool in = p < npx;
        float a[IT], b[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool ok = in && it * 16 + n < C;
            a[it] = ok ? dzo[p * C + it * 16 + n] : 0.f;
            b[it] = ok ? z[p * C + it * 16 + n] : 0.f;
            sa[it] += a[it];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < IT; ++jt) acc[it][jt] = mfma16(a[it], b[jt], acc[it][jt]);
    }
    float* out = part + (int64_t)blockIdx.x * (C * C + C);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int jt = 0; jt < IT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + 4 * g + r, j = jt * 16 + n;
                if (i < C && j < C) out[i * C + j] = acc[it][jt][r];
            }
        float t = sa[it];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (g == 0 && it * 16 + n < C) out[C * C + it * 16 + n] = t;
    }
}

__device__ __forceinline__ int64_t ring_index(int64_t p, int h, int w) {
    const int x = p % w; const int64_t t = p / w; const int y = t % h; const int64_t b = t / h;
    return (b * (h + 2) + y + 1) * (w + 2) + x + 1;
}

// ring[b][y+1][x+1][k] = z[b][y][x][k], k < ch (ring: [..][32], its other entries stay as they are: zero)
__global__ __launch_bounds__(256)
void ring_pack_kernel(const float* __restrict__ z, float* __restrict__ ring, int64_t npx, int h, int w, int C, int ch)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx * ch) return;
    const int64_t p = i / ch; const int k = i - p * ch;
    ring[ring_index(p, h, w) * 32 + k] = z[p * C + k];
}

// dz[b][y][x][k] += dring[b][y+1][x+1][k], k < ch
__global__ __launch_bounds__(256)
void ring_unpack_add_kernel(const float* __restrict__ dring, float* __restrict__ dz, int64_t npx, int h, int w, int C, int ch)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx * ch) return;
    const int64_t p = i / ch; const int k = i - p * ch;
    dz[p * C + k] += dring[ring_index(p, h, w) * 32 + k];
}

// padded input channel -> the parameter's input channel (or -1): source 0 holds real0 of pad0 channels, source 1 real1 of pad1
__device__ __forceinline__ int real_channel(const ifl_pack_desc& d, int cip) {
    if (cip < d.pad0) return cip < d.real0 ? cip : -1;
    const int k = cip - d.pad0;
    return k < d.real1 ? d.real0 + k : -1;
}

__global__ __launch_bounds__(256)
void pack_table_kernel(const ifl_pack_desc* __restrict__ table)
{
    const ifl_pack_desc d = table[blockIdx.y];
    const int cin_real = d.cin_src ? d.cin_src : d.real0 + d.real1, cin_pad = d.pad0 + d.pad1;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < d.nelem; idx += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        if (d.mode == IFL_PACK_FWD || d.mode == IFL_PACK_TRANSPOSED) {
            // fragment order [group of 32 outputs][chunk of 32 inputs][tap][kq][half][i][e]
            const int e = idx & 3, i = (idx >> 2) & 31, hh = (idx >> 7) & 1, kq = (idx >> 8) & 3;
            int64_t r = idx >> 10;
            const int tap = r % d.taps; r /= d.taps;
            const int nch = (d.mode == IFL_PACK_FWD ? cin_pad : d.cout_pad) / 32;
            const int chunk = r % nch, g = r / nch;
            const int o = g * 32 + i, k = chunk * 32 + kq * 8 + hh * 4 + e;
            if (d.mode == IFL_PACK_FWD) {
                const int ci = real_channel(d, k);
                if (o < d.cout_real && ci >= 0) v = d.src[((int64_t)o * cin_real + ci) * d.taps + tap];
            } else {                          // Wt[a][co][tap] = W[co][a][taps - 1 - tap] for the padded input channels a0 + [0, acnt)
                const int ci = real_channel(d, d.a0 + o);
                if (k < d.cout_real && ci >= 0) v = d.src[((int64_t)k * cin_real + ci) * d.taps + (d.taps - 1 - tap)];
            }
        } else if (d.mode == IFL_PACK_VECTOR) {
            if (idx < d.cout_real) v = d.src[idx];
        } else if (d.mode == IFL_PACK_WINO4) {
            // [mh][bi][bj][s][il][jl][hh][o][e]: position (3 bi + il, 3 bj + jl), output mh*32 + o, input s*8 + hh*4 + e
            const int e = idx & 3, o = (idx >> 2) & 31, hh = (idx >> 7) & 1;
            int r = (int)(idx >> 8);
            const int jl = r % 3; r /= 3;
            const int il = r % 3; r /= 3;
            const int sc = r % 40; r /= 40;
            const int bj = r & 1, bi = (r >> 1) & 1, mh = r >> 2;
            const int co = mh * 32 + o, ci = d.real0 + sc * 8 + hh * 4 + e;
            const int pi = 3 * bi + il, pj = 3 * bj + jl;
            const float* g = d.src + ((int64_t)co * cin_real + ci) * 9;
            double G[6][3];
            ifl_wino4_G(G);
            double u = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) u += G[pi][a] * (double)g[a * 3 + b] * G[pj][b];
            v = (float)u;
        } else if (d.mode == IFL_PACK_WINO4_T) {
            // U' = G rot180(g) G^T for the data gradient wrt the conditioning features: dst[(p*320 + ci)*acnt + a0 + co]
            const int co = idx & 63;
            const int ci = (int)((idx >> 6) % IFL_FT);
            const int p = (int)((idx >> 6) / IFL_FT);
            const int pi = p / 6, pj = p - pi * 6;
            const float* g = d.src + ((int64_t)co * cin_real + d.real0 + ci) * 9;
            double G[6][3];
            ifl_wino4_G(G);
            double u = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) u += G[pi][a] * (double)g[(2 - a) * 3 + (2 - b)] * G[pj][b];
            d.dst[((int64_t)p * IFL_FT + ci) * d.acnt + d.a0 + co] = (float)u;
            continue;
        } else {                              // IFL_PACK_GRAD: dense [cout_pad][cin_pad][taps] -> the parameter's [cout][cin][taps]
            const int tap = idx % d.taps; int64_t r = idx / d.taps;
            const int ci = r % (d.real0 + d.real1), co = r / (d.real0 + d.real1);
            const int col = ci < d.real0 ? ci : d.pad0 + (ci - d.real0);
            v = d.src[((int64_t)co * cin_pad + col) * d.taps + tap];
        }
        d.dst[idx] = v;
    }
}


// ---- Winograd F(4x4,3x3) weight gradient: K-major transforms and the map back to 3x3 taps --------------------------------
// Thread = (tile, channel quad): block 256 = 64 consecutive tiles x 4 quads; grid (tile blocks, C / 16, planes).  Stores: for
// a fixed (position, channel) the 64 tiles of a wave are 256 contiguous bytes of out[p][row][tile].
template <int MODE>
__global__ __launch_bounds__(256)
void wino4_kmajor_kernel(const float* __restrict__ src, int C, int64_t plane_stride, float* __restrict__ out, int64_t rows_total,
                         int64_t row0, IflXGeom g, int64_t K)
{
    const int tl = threadIdx.x & 63, q = (threadIdx.x >> 6) + 4 * blockIdx.y;
    const int64_t tile = (int64_t)blockIdx.x * 64 + tl;
    if (tile >= g.ntiles) return;
    const int tpi = g.TH * g.TW, W2 = g.W + 2;
    const int b = tile / tpi, r2 = tile - b * tpi;
    const int tr = r2 / g.TW, tc = r2 - tr * g.TW;
    const float* sp = src + (int64_t)blockIdx.z * plane_stride + q * 4;
    f32x4 T[6][6];
    if (MODE == 0) {                          // V = B^T d B of the 6x6 window whose corner is ring pixel (4 tr, 4 tc)
        const float* base = sp + ((int64_t)(b * (g.H + 2) + 4 * tr) * W2 + 4 * tc) * C;
        f32x4 Tc[6][6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const bool cin = 4 * tc + c <= g.W + 1;
            f32x4 x[6];
#pragma unroll
            for (int r = 0; r < 6; ++r)
                x[r] = (cin && 4 * tr + r <= g.H + 1) ? *(const f32x4*)(base + (int64_t)(r * W2 + c) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            Tc[0][c] = IFL_W4_AB * x[0] - IFL_W4_ApB * x[2] + x[4];
            Tc[1][c] = (x[4] - IFL_W4_B * x[2]) + IFL_W4_a * (x[3] - IFL_W4_B * x[1]);
            Tc[2][c] = (x[4] - IFL_W4_B * x[2]) - IFL_W4_a * (x[3] - IFL_W4_B * x[1]);
            Tc[3][c] = (x[4] - IFL_W4_A * x[2]) + IFL_W4_b * (x[3] - IFL_W4_A * x[1]);
            Tc[4][c] = (x[4] - IFL_W4_A * x[2]) - IFL_W4_b * (x[3] - IFL_W4_A * x[1]);
            Tc[5][c] = IFL_W4_AB * x[1] - IFL_W4_ApB * x[3] + x[5];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4* x = Tc[i];
            T[i][0] = IFL_W4_AB * x[0] - IFL_W4_ApB * x[2] + x[4];
            T[i][1] = (x[4] - IFL_W4_B * x[2]) + IFL_W4_a * (x[3] - IFL_W4_B * x[1]);
            T[i][2] = (x[4] - IFL_W4_B * x[2]) - IFL_W4_a * (x[3] - IFL_W4_B * x[1]);
            T[i][3] = (x[4] - IFL_W4_A * x[2]) + IFL_W4_b * (x[3] - IFL_W4_A * x[1]);
            T[i][4] = (x[4] - IFL_W4_A * x[2]) - IFL_W4_b * (x[3] - IFL_W4_A * x[1]);
            T[i][5] = IFL_W4_AB * x[1] - IFL_W4_ApB * x[3] + x[5];
        }
    } else {                                  // dM = A dY A^T of the 4x4 block whose corner is image pixel (4 tr, 4 tc)
        const float* base = sp + ((int64_t)(b * (g.H + 2) + 4 * tr + 1) * W2 + 4 * tc + 1) * C;
        f32x4 Tc[6][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool cin = 4 * tc + c < g.W;
            f32x4 y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                y[r] = (cin && 4 * tr + r < g.H) ? *(const f32x4*)(base + (int64_t)(r * W2 + c) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            Tc[0][c] = y[0];
            Tc[1][c] = (y[0] + IFL_W4_A * y[2]) + (IFL_W4_a * y[1] + IFL_W4_a3 * y[3]);
            Tc[2][c] = (y[0] + IFL_W4_A * y[2]) - (IFL_W4_a * y[1] + IFL_W4_a3 * y[3]);
            Tc[3][c] = (y[0] + IFL_W4_B * y[2]) + (IFL_W4_b * y[1] + IFL_W4_b3 * y[3]);
            Tc[4][c] = (y[0] + IFL_W4_B * y[2]) - (IFL_W4_b * y[1] + IFL_W4_b3 * y[3]);
            Tc[5][c] = y[3];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4* y = Tc[i];
            T[i][0] = y[0];
            T[i][1] = (y[0] + IFL_W4_A * y[2]) + (IFL_W4_a * y[1] + IFL_W4_a3 * y[3]);
            T[i][2] = (y[0] + IFL_W4_A * y[2]) - (IFL_W4_a * y[1] + IFL_W4_a3 * y[3]);
            T[i][3] = (y[0] + IFL_W4_B * y[2]) + (IFL_W4_b * y[1] + IFL_W4_b3 * y[3]);
            T[i][4] = (y[0] + IFL_W4_B * y[2]) - (IFL_W4_b * y[1] + IFL_W4_b3 * y[3]);
            T[i][5] = y[3];
        }
    }
    const int64_t row = row0 + (int64_t)blockIdx.z * C + q * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jc = 0; jc < 6; ++jc)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[((int64_t)(i * 6 + jc) * rows_total + row + e) * K + tile] = T[i][jc][e];
}

// thread = (row, ci): dg = G^T dU G of the 36 values dU[p][row][ci]; pair = row / 64
__global__ __launch_bounds__(256)
void wino4_dw_kernel(const float* __restrict__ dU, int64_t rows_total, const ifl_wino4_dst* __restrict__ table)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_total * IFL_FT) return;
    const int64_t row = idx / IFL_FT; const int ci = idx - row * IFL_FT;
    const ifl_wino4_dst d = table[row >> 6];
    const int co = row & 63;
    float G[6][3];
    ifl_wino4_G(G);
    float t[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 6; ++j) t[a][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float u = dU[((int64_t)(i * 6 + j) * rows_total + row) * IFL_FT + ci];
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a][j] += G[i][a] * u;
        }
    float* dst = d.dst + ((int64_t)co * d.cin + d.zoff + ci) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) v += t[a][j] * G[j][b];
            dst[a * 3 + b] = v;
        }
}

// Y [36][C][tiles] (Winograd domain, tile index contiguous) -> out = A^T Y A as an NCHW tensor [B][C][h][w]: the output
// transform of F(4x4,3x3) for the data gradient wrt the conditioning features.  Thread = (tile, channel): the 64 lanes of a
// wave read 256 contiguous bytes per position and write four-float row pieces of neighbouring tiles.
__global__ __launch_bounds__(256)
void wino4_untile_kernel(const float* __restrict__ Y, float* __restrict__ out, int C, IflXGeom g)
{
    const int64_t tile = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (tile >= g.ntiles || c >= C) return;
    const int64_t T = g.ntiles;
    const int tpi = g.TH * g.TW;
    const int b = tile / tpi, r2 = tile - (int64_t)b * tpi;
    const int tr = r2 / g.TW, tc = r2 - tr * g.TW;
    float t[4][6];                                            // A^T m (rows), then (A^T m) A
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = Y[((int64_t)(i * 6 + j) * C + c) * T + tile];
        t[0][j] = (m[0] + (m[1] + m[2])) + (m[3] + m[4]);
        t[1][j] = IFL_W4_a * (m[1] - m[2]) + IFL_W4_b * (m[3] - m[4]);
        t[2][j] = IFL_W4_A * (m[1] + m[2]) + IFL_W4_B * (m[3] + m[4]);
        t[3][j] = (IFL_W4_a3 * (m[1] - m[2]) + IFL_W4_b3 * (m[3] - m[4])) + m[5];
    }
    float* base = out + ((int64_t)b * C + c) * g.H * g.W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = 4 * tr + r;
        if (y >= g.H) break;
        const float* m = t[r];
        f32x4 o;
        o[0] = (m[0] + (m[1] + m[2])) + (m[3] + m[4]);
        o[1] = IFL_W4_a * (m[1] - m[2]) + IFL_W4_b * (m[3] - m[4]);
        o[2] = IFL_W4_A * (m[1] + m[2]) + IFL_W4_B * (m[3] + m[4]);
        o[3] = (IFL_W4_a3 * (m[1] - m[2]) + IFL_W4_b3 * (m[3] - m[4])) + m[5];
        float* dst = base + (int64_t)y * g.W + 4 * tc;
        if (4 * tc + 3 < g.W && (g.W & 3) == 0) *(f32x4*)dst = o;
        else
            for (int e = 0; e < 4; ++e) if (4 * tc + e < g.W) dst[e] = o[e];
    }
}

template <int C>
int launch_affine_fwd(const float* z, float* zo, const float* M, const float* c, int64_t npx, hipStream_t st) {
    hipLaunchKernelGGL(affine_px_fwd_kernel<C>, dim3((unsigned)((npx + APX - 1) / APX)), dim3(64), 0, st, z, zo, M, c, npx);
    IFL_LAUNCH_CHECK();
    return 0;
}
template <int C>
int launch_affine_bwd(const float* z, const float* dzo, const float* M, float* dz, float* part, int64_t npx, hipStream_t st) {
    hipLaunchKernelGGL(affine_px_bwd_kernel<C>, dim3((unsigned)((npx + APX - 1) / APX)), dim3(64), 0, st, z, dzo, M, dz, part, npx);
    IFL_LAUNCH_CHECK();
    return 0;
}

int affine_fwd(const float* z, float* zo, const float* M, const float* c, int64_t npx, int C, hipStream_t st) {
    if (C == 12) return launch_affine_fwd<12>(z, zo, M, c, npx, st);
    if (C == 48) return launch_affine_fwd<48>(z, zo, M, c, npx, st);
    ifl_set_error("affine_px: C=%d (12 or 48)", C);
    return IFL_ENOTSUP;
}
int affine_bwd(const float* z, const float* dzo, const float* M, float* dz, float* part, int64_t npx, int C, hipStream_t st) {
    if (C == 12) return launch_affine_bwd<12>(z, dzo, M, dz, part, npx, st);
    if (C == 48) return launch_affine_bwd<48>(z, dzo, M, dz, part, npx, st);
    ifl_set_error("affine_px: C=%d (12 or 48)", C);
    return IFL_ENOTSUP;
}

int conv(const float* s0, int c0, const float* s1, int c1, const float* w, const float* bias, float* out, int cout, int taps,
         int B, int h, int w_, hipStream_t st)
{
    ifl_conv_desc d = {};
    d.src[0] = s0; d.src_stride[0] = c0; d.src_ch[0] = c0;
    d.nsrc = 1;
    if (s1) { d.src[1] = s1; d.src_stride[1] = c1; d.src_ch[1] = c1; d.nsrc = 2; }
    d.w = w; d.bias = bias; d.out = out; d.out_stride = cout; d.cout = cout; d.taps = taps;
    d.alpha = d.beta = 1.f;
    return ifl_conv2d(&d, B, h, w_, st);
}


int check_level(const ifl_train_level* L) {
    IFL_CHECK_ARG(L && L->steps && L->nsteps >= 1, IFL_EINVAL, "train level: no steps");
    IFL_CHECK_ARG(L->B >= 1 && L->h >= 1 && L->w >= 1 && (L->C == 12 || L->C == 48), IFL_ENOTSUP,
                  "train level: B=%d h=%d w=%d C=%d (C 12 or 48)", L->B, L->h, L->w, L->C);
    IFL_CHECK_ARG(L->ch >= 1 && L->ch <= 32 && L->ch < L->C, IFL_ENOTSUP, "train level: %d of %d channels feed the coupling networks", L->ch, L->C);
    IFL_CHECK_ARG(L->branch == 0 || L->branch == 1, IFL_EINVAL, "branch %d", L->branch);
    IFL_CHECK_ARG(!L->f_batched || (L->raw && L->skip_step_wgrad && L->anp && L->da_all && L->anp_step >= 256), IFL_EINVAL,
                  "train level: f_batched needs raw, skip_step_wgrad, anp and da_all");
    for (int s = 0; s < L->nsteps; ++s) {
        const ifl_train_step& S = L->steps[s];
        IFL_CHECK_ARG(S.M && S.c && S.dM_part && S.z_in && S.z_a, IFL_EINVAL, "train step %d: null pointer", s);
        if (!S.cond) continue;
        IFL_CHECK_ARG(L->ft && L->ft_ch % 32 == 0 && L->z1ring && L->dz1ring && S.z_mid && S.z_out, IFL_EINVAL, "train step %d: null pointer", s);
        for (int i = 0; i < 2; ++i) {
            const ifl_train_net& N = S.net[i];
            IFL_CHECK_ARG(N.w1 && N.w2 && N.w3 && N.w2t && N.w3t && N.p1 && N.a1 && N.p2 && N.a2 && N.hraw && N.dW1 && N.dW2 && N.dW3 &&
                          N.an1_part && N.an2_part && N.dlogs_part && N.dbias_part && N.ldpart && N.bias3 && N.logs3 &&
                          N.an1_bias && N.an1_logs && N.an2_bias && N.an2_logs && L->dh[i] && (!N.has_z1 || N.w1t),
                          IFL_EINVAL, "train step %d net %d: null pointer", s, i);
            IFL_CHECK_ARG(!L->skip_step_wgrad || (N.dp1 && N.dp2 && N.dh && (!N.has_z1 || S.z1)), IFL_EINVAL,
                          "train step %d net %d: skip_step_wgrad needs dp1, dp2, dh and z1", s, i);
            IFL_CHECK_ARG(!L->skip_ft_wgrad || N.dp1, IFL_EINVAL, "train step %d net %d: skip_ft_wgrad needs dp1", s, i);
            IFL_CHECK_ARG(N.p1_stride == 64 || (N.p1_stride == 128 && L->raw && !N.has_z1), IFL_EINVAL, "train step %d net %d: p1 stride %d", s, i, N.p1_stride);
            IFL_CHECK_ARG(N.n >= 1 && N.c0 >= 0 && N.c0 + N.n <= L->C && N.hstride % 32 == 0 && 2 * N.n <= N.hstride && N.hstride <= 96,
                          IFL_EINVAL, "train step %d net %d: channels [%d, %d) of %d, %d rows", s, i, N.c0, N.c0 + N.n, L->C, N.hstride);
        }
    }
    return 0;
}

static_assert(sizeof(ifl_pack_desc) == 72 && sizeof(ifl_train_net) == 248 && sizeof(ifl_train_step) == 568 && sizeof(ifl_train_level) == 192,
              "the Python side (train_fused.py) mirrors these structs");

}  // namespace

extern "C" int ifl_pack_table(const ifl_pack_desc* table_dev, int n, int64_t max_elems, void* stream)
{
    IFL_CHECK_ARG(table_dev && n >= 1 && n <= 65535 && max_elems >= 1, IFL_EINVAL, "pack table: %d entries", n);
    int64_t gx = (max_elems + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)gx, n), dim3(256), 0, (hipStream_t)stream, table_dev);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t ifl_affine_px_blocks(int64_t npx) { return (npx + APX - 1) / APX; }

extern "C" int64_t ifl_train_vg_floats(int B, int h, int w) { return ifl_vg_floats(B, h, w); }

extern "C" int64_t ifl_wino4_tiles(int B, int h, int w) { return ifl_xgeom(B, h, w).ntiles; }

extern "C" int ifl_wino4_kmajor(int mode, const float* src, int C, int n_planes, int64_t plane_stride, float* out, int64_t rows_total,
                                int64_t row0, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(src && out && (mode == 0 || mode == 1) && C % 16 == 0 && C >= 16 && n_planes >= 1 && n_planes <= 65535, IFL_EINVAL,
                  "wino4 transform: mode %d, %d channels, %d planes", mode, C, n_planes);
    IFL_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && row0 >= 0 && row0 + (int64_t)n_planes * C <= rows_total, IFL_EINVAL, "wino4 transform: rows");
    const IflXGeom g = ifl_xgeom(B, h, w);
    dim3 grid((g.ntiles + 63) / 64, C / 16, n_planes), block(256);
    if (mode == 0) hipLaunchKernelGGL(wino4_kmajor_kernel<0>, grid, block, 0, (hipStream_t)stream, src, C, plane_stride, out, rows_total, row0, g, (int64_t)g.ntiles);
    else hipLaunchKernelGGL(wino4_kmajor_kernel<1>, grid, block, 0, (hipStream_t)stream, src, C, plane_stride, out, rows_total, row0, g, (int64_t)g.ntiles);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_wino4_dw(const float* dU, int64_t rows_total, const ifl_wino4_dst* table_dev, int n_pairs, void* stream)
{
    IFL_CHECK_ARG(dU && table_dev && n_pairs >= 1 && rows_total == (int64_t)n_pairs * 64, IFL_EINVAL, "wino4 dw: %d pairs", n_pairs);
    const int64_t n = rows_total * IFL_FT;
    hipLaunchKernelGGL(wino4_dw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dU, rows_total, table_dev);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_wino4_untile(const float* Y, float* out_nchw, int C, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(Y && out_nchw && C >= 1 && B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "wino4 untile: null pointer / shape");
    const IflXGeom g = ifl_xgeom(B, h, w);
    dim3 grid((unsigned)((g.ntiles + 63) / 64), (unsigned)((C + 3) / 4)), block(256);
    IFL_CHECK_ARG(grid.y <= 65535, IFL_EINVAL, "wino4 untile: %d channels", C);
    hipLaunchKernelGGL(wino4_untile_kernel, grid, block, 0, (hipStream_t)stream, Y, out_nchw, C, g);
    IFL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ifl_train_bulk_conv1(const float* ft_ring, float* vg, const float* wx, int64_t wx_step, int nsteps, float* raw,
                                    int64_t raw_step, int B, int h, int w, void* stream)
{
    IFL_CHECK_ARG(ft_ring && vg && wx && raw && nsteps >= 1 && B >= 1 && h >= 1 && w >= 1, IFL_EINVAL, "train bulk conv1: null pointer / shape");
    IFL_CHECK_ARG(wx_step >= 2 * (int64_t)36 * IFL_HID * IFL_FT && raw_step >= (int64_t)B * (h + 2) * (w + 2) * IFL_H2, IFL_EINVAL, "train bulk conv1: strides");
    hipStream_t st = (hipStream_t)stream;
    IFL_TRY(ifl_k_wino4_input(ft_ring, vg, B, h, w, st));
    IflStepOff off{};
    off.w1x[0] = 0;
    off.w1x[1] = (int64_t)36 * IFL_HID * IFL_FT;
    // launches of 4 steps (8 (network, step) pairs: one per XCD) on large planes, 8 on small ones, as the inference schedule
    const int per = (int64_t)B * h * w >= 100000 ? 4 : 8;
    for (int s0 = 0; s0 < nsteps; s0 += per) {
        const int n = nsteps - s0 < per ? nsteps - s0 : per;
        IflStepList steps{};
        for (int i = 0; i < n; ++i) steps.off[i] = (int64_t)(s0 + i) * wx_step;
        IFL_TRY(ifl_k_conv1_ft_wino4(vg, raw + (int64_t)s0 * raw_step, raw_step, wx, steps, n, off, B, h, w, h >= 64 ? 1 : 2, st, 1));
    }
    return 0;
}

// ---- sample groups ---------------------------------------------------------------------------------------------------------
// Samples are independent, and at batch 8 every kernel of the chain is latency-bound (a step takes the same time at batch 2):
// with L->groups == 2 the batch is cut in two halves that run the same chain on two streams (the caller's and one
// library-owned per device), forked and joined with events.  Everything a group touches is a sample range of the
// level's buffers; per-block partial sums take the ranges [group 0's blocks | group 1's blocks].
namespace {

struct TrainFork { hipStream_t aux = nullptr; hipEvent_t fork = nullptr, join = nullptr; std::mutex mu; };
TrainFork g_tfork[IFL_MAX_DEVICES];

int train_fork(TrainFork** out)
{
    int dev = 0;
    IFL_HIP(hipGetDevice(&dev));
    IFL_CHECK_ARG(dev >= 0 && dev < IFL_MAX_DEVICES, IFL_ENOTSUP, "device %d", dev);
    TrainFork& f = g_tfork[dev];
    {
        std::lock_guard<std::mutex> once(f.mu);          // two host threads making their first call on one device
        if (!f.aux) {
            hipStream_t aux = nullptr;
            IFL_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
            IFL_HIP(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
            IFL_HIP(hipEventCreateWithFlags(&f.join, hipEventDisableTiming));
            f.aux = aux;                                  // last: marks the context as complete
        }
    }
    *out = &f;
    return 0;
}

struct Grp { int b0, Bg, idx; int64_t an_blk0, aff_blk0; };    // sample range; first partial-sum block of the group

// one step of one group; cstep = the step's index among the conditional steps (the bulk slab's).
// Encode (FlowStep.normal_flow, FlowStep.py:114-151):  z_in -affine-> z_a -fAffine-> z_mid -fFeatures-> z_out.
// Decode (L->decode; reverse_flow, :153-196, the steps listed in the order the decode runs them):
//   z_in -fFeatures-> z_a -fAffine-> z_mid -affine (the inverse map)-> z_out.
int fwd_step(const ifl_train_level* L, const Grp& G, int s, int cstep, hipStream_t st)
{
    const int B = G.Bg, h = L->h, w = L->w, C = L->C;
    const int64_t npx = (int64_t)B * h * w, plane = (int64_t)(h + 2) * (w + 2);
    const int64_t zo = (int64_t)G.b0 * h * w * C, ro = (int64_t)G.b0 * plane;     // offsets of the group in z buffers / ringed pixels
    const int nblk_c = ifl_coupling_blocks(h, w);
    const ifl_train_step& S = L->steps[s];
    auto net_fwd = [&](int i, const float* zsrc, float* zdst) -> int {
        const ifl_train_net& N = S.net[i];
        const float* p1 = N.p1 + ro * N.p1_stride;
        float *a1 = N.a1 + ro * 64, *p2 = N.p2 + ro * 64, *a2 = N.a2 + ro * 64, *hraw = N.hraw + ro * N.hstride;
        const float* ft = L->ft ? L->ft + ro * L->ft_ch : nullptr;
        if (L->f_batched && !N.has_z1)        // its network ran for all steps before the chain
            return ifl_coupling_fwd(zsrc, zdst, hraw, N.hstride, N.logs3, L->eps, L->branch, N.c0, N.n, N.ldpart + (int64_t)G.b0 * nblk_c,
                                    B, h, w, C, st);
        if (N.has_z1) {
            float* z1 = (S.z1 ? S.z1 : L->z1ring) + ro * 32;
            hipLaunchKernelGGL(ring_pack_kernel, dim3((unsigned)((npx * L->ch + 255) / 256)), dim3(256), 0, st, zsrc, z1, npx, h, w, C, L->ch);
            IFL_LAUNCH_CHECK();
            if (L->raw) {                     // z1's taps here, the conditioning part from the bulk slab
                ifl_conv_desc d = {};
                d.src[0] = z1; d.src_stride[0] = 32; d.src_ch[0] = 32; d.nsrc = 1;
                d.w = N.w1; d.out = N.p1 + ro * 64; d.out_stride = 64; d.cout = 64; d.taps = 9;
                d.res1 = L->raw + (int64_t)cstep * L->raw_step + ro * 128; d.res1_stride = 128; d.res1_off = 0; d.alpha = 1.f; d.beta = 1.f;
                IFL_TRY(ifl_conv2d(&d, B, h, w, st));
            } else {
                IFL_TRY(conv(z1, 32, ft, L->ft_ch, N.w1, nullptr, N.p1 + ro * 64, 64, 9, B, h, w, st));
            }
        } else if (!L->raw) {
            IFL_TRY(conv(ft, L->ft_ch, nullptr, 0, N.w1, nullptr, N.p1 + ro * 64, 64, 9, B, h, w, st));
        }
        IFL_TRY(ifl_k_an_relu_fwd(p1, N.p1_stride, a1, N.an1_bias, N.an1_logs, B, h, w, 64, st));
        IFL_TRY(conv(a1, 64, nullptr, 0, N.w2, nullptr, p2, 64, 1, B, h, w, st));
        IFL_TRY(ifl_an_relu_fwd(p2, a2, N.an2_bias, N.an2_logs, B, h, w, 64, st));
        IFL_TRY(conv(a2, 64, nullptr, 0, N.w3, N.bias3, hraw, N.hstride, 9, B, h, w, st));
        return ifl_coupling_fwd(zsrc, zdst, hraw, N.hstride, N.logs3, L->eps, L->branch, N.c0, N.n, N.ldpart + (int64_t)G.b0 * nblk_c,
                                B, h, w, C, st);
    };
    if (!S.cond) return affine_fwd(S.z_in + zo, S.z_a + zo, S.M, S.c, npx, C, st);
    if (!L->decode) {
        IFL_TRY(affine_fwd(S.z_in + zo, S.z_a + zo, S.M, S.c, npx, C, st));
        IFL_TRY(net_fwd(0, S.z_a + zo, S.z_mid + zo));
        return net_fwd(1, S.z_mid + zo, S.z_out + zo);
    }
    IFL_TRY(net_fwd(1, S.z_in + zo, S.z_a + zo));
    IFL_TRY(net_fwd(0, S.z_a + zo, S.z_mid + zo));
    return affine_fwd(S.z_mid + zo, S.z_out + zo, S.M, S.c, npx, C, st);
}

// one step of one group; *gp = the gradient wrt the step's output on entry, wrt its input on return
int bwd_step(const ifl_train_level* L, const Grp& G, int s, const float** gp, const float* dld, float* dz_in, hipStream_t st)
{
    const int B = G.Bg, h = L->h, w = L->w, C = L->C;
    const int64_t npx = (int64_t)B * h * w, plane = (int64_t)(h + 2) * (w + 2);
    const int64_t zo = (int64_t)G.b0 * h * w * C, ro = (int64_t)G.b0 * plane;
    const int nblk_c = ifl_coupling_blocks(h, w);
    if (dld) dld += G.b0;
    const ifl_train_step& S = L->steps[s];
    // coupling i: g = gradient wrt its output, dzn = gradient wrt its input zsrc (all channels written)
    auto net_bwd = [&](int i, const float* zsrc, const float* g, float* dzn) -> int {
        const ifl_train_net& N = S.net[i];
        float* dh = (N.dh ? N.dh : L->dh[i]) + ro * N.hstride;
        float* da = L->d64[0] + ro * 64;
        float* dp = (N.dp2 ? N.dp2 : L->d64[1]) + ro * 64;  // gradient wrt the 1x1 convolution's output
        float* dp1 = (N.dp1 ? N.dp1 : L->d64[1]) + ro * 64; // gradient wrt the first convolution's output
        const float *a1 = N.a1 + ro * 64, *p2 = N.p2 + ro * 64, *a2 = N.a2 + ro * 64, *hraw = N.hraw + ro * N.hstride;
        const float* p1 = N.p1 + ro * N.p1_stride;
        const float* ft = L->ft ? L->ft + ro * L->ft_ch : nullptr;
        const int64_t co = (int64_t)G.b0 * nblk_c * 2 * N.n;
        const bool wg = !L->skip_step_wgrad;    // weight gradients here, or batched by the caller afterwards
        IFL_TRY(ifl_k_coupling_bwd(zsrc, g, dld, hraw, N.hstride, N.logs3, L->eps, L->branch, N.c0, N.n, dzn, dh, N.dlogs_part + co,
                                   N.dbias_part + co, B, h, w, C, st));
        if (L->f_batched && !N.has_z1) return 0;            // its network's backward runs for all steps after the chain
        // Conv2dZeros 3x3: a2 -> hraw
        if (wg) IFL_TRY(ifl_conv2d_wgrad(a2, 64, 0, 64, dh, N.hstride, 0, N.hstride, N.dW3, 64, 0, 9, B, h, w, st));
        IFL_TRY(conv(dh, N.hstride, nullptr, 0, N.w3t, nullptr, da, 64, 9, B, h, w, st));
        IFL_TRY(ifl_an_relu_bwd(p2, da, dp, N.an2_bias, N.an2_logs, N.an2_part + G.an_blk0 * 128, B, h, w, 64, st));
        // 1x1: a1 -> p2
        if (wg) IFL_TRY(ifl_conv2d_wgrad(a1, 64, 0, 64, dp, 64, 0, 64, N.dW2, 64, 0, 1, B, h, w, st));
        IFL_TRY(conv(dp, 64, nullptr, 0, N.w2t, nullptr, da, 64, 1, B, h, w, st));
        IFL_TRY(ifl_k_an_relu_bwd(p1, N.p1_stride, da, dp1, N.an1_bias, N.an1_logs, N.an1_part + G.an_blk0 * 128, B, h, w, 64, st));
        // first 3x3: (z1 | ft) -> p1
        if (N.has_z1) {
            const unsigned nb = (unsigned)((npx * L->ch + 255) / 256);
            const float* z1 = S.z1 ? S.z1 + ro * 32 : nullptr;
            if (!z1) {                          // not kept by the forward: pack it again
                hipLaunchKernelGGL(ring_pack_kernel, dim3(nb), dim3(256), 0, st, zsrc, L->z1ring + ro * 32, npx, h, w, C, L->ch);
                IFL_LAUNCH_CHECK();
                z1 = L->z1ring + ro * 32;
            }
            if (wg) IFL_TRY(ifl_conv2d_wgrad(z1, 32, 0, 32, dp1, 64, 0, 64, N.dW1, 32 + L->ft_ch, 0, 9, B, h, w, st));
            if (wg && !L->skip_ft_wgrad)
                IFL_TRY(ifl_conv2d_wgrad(ft, L->ft_ch, 0, L->ft_ch, dp1, 64, 0, 64, N.dW1, 32 + L->ft_ch, 1, 9, B, h, w, st));
            IFL_TRY(conv(dp1, 64, nullptr, 0, N.w1t, nullptr, L->dz1ring + ro * 32, 32, 9, B, h, w, st));
            hipLaunchKernelGGL(ring_unpack_add_kernel, dim3(nb), dim3(256), 0, st, L->dz1ring + ro * 32, dzn, npx, h, w, C, L->ch);
            IFL_LAUNCH_CHECK();
        } else if (wg && !L->skip_ft_wgrad) {
            IFL_TRY(ifl_conv2d_wgrad(ft, L->ft_ch, 0, L->ft_ch, dp1, 64, 0, 64, N.dW1, L->ft_ch, 0, 9, B, h, w, st));
        }
        return 0;
    };
    const float* g = *gp;
    float* out = (s == 0 ? dz_in : L->dz[2 + (s & 1)]) + zo;
    float* part = S.dM_part + G.aff_blk0 * (C * C + C);
    if (!S.cond) {
        IFL_TRY(affine_bwd(S.z_in + zo, g, S.M, out, part, npx, C, st));
    } else if (!L->decode) {
        IFL_TRY(net_bwd(1, S.z_mid + zo, g, L->dz[1] + zo));
        IFL_TRY(net_bwd(0, S.z_a + zo, L->dz[1] + zo, L->dz[0] + zo));
        IFL_TRY(affine_bwd(S.z_in + zo, L->dz[0] + zo, S.M, out, part, npx, C, st));
    } else {
        IFL_TRY(affine_bwd(S.z_mid + zo, g, S.M, L->dz[0] + zo, part, npx, C, st));
        IFL_TRY(net_bwd(0, S.z_a + zo, L->dz[0] + zo, L->dz[1] + zo));
        IFL_TRY(net_bwd(1, S.z_in + zo, L->dz[1] + zo, out));
    }
    *gp = out;
    return 0;
}

// The level's one or two sample groups, interleaved step by step (so that both streams have work from the start: the host
// enqueues about as fast as the device runs these small kernels); two groups: the second on the device's auxiliary stream.
struct Groups { int n; Grp g[2]; hipStream_t st[2]; TrainFork* f; };

int groups_begin(const ifl_train_level* L, hipStream_t st, Groups* out)
{
    Groups& R = *out;
    R.n = (L->groups == 2 && L->B >= 2) ? 2 : 1;
    const int B0 = R.n == 2 ? L->B / 2 : L->B;
    R.g[0] = Grp{0, B0, 0, 0, 0};
    R.st[0] = R.st[1] = st;
    R.f = nullptr;
    if (R.n == 1) return 0;
    R.g[1] = Grp{B0, L->B - B0, 1, ifl_an_relu_blocks(B0, L->h, L->w), ifl_affine_px_blocks((int64_t)B0 * L->h * L->w)};
    IFL_TRY(train_fork(&R.f));
    R.f->mu.lock();                                           // the auxiliary stream and its events are per device
    R.st[1] = R.f->aux;
    hipError_t e = hipEventRecord(R.f->fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(R.f->aux, R.f->fork, 0);
    if (e != hipSuccess) { R.f->mu.unlock(); ifl_set_error("train level fork: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int groups_end(Groups& R, int rc)
{
    if (R.n == 1) return rc;
    hipError_t e = hipEventRecord(R.f->join, R.f->aux);       // always rejoin, whatever the steps returned
    if (e == hipSuccess) e = hipStreamWaitEvent(R.st[0], R.f->join, 0);
    R.f->mu.unlock();
    if (e != hipSuccess && !rc) { ifl_set_error("train level join: %s", hipGetErrorString(e)); rc = (int)e; }
    return rc;
}

// The z-independent network of every conditional step (f_batched): strides between consecutive conditional steps, taken from
// the first two step records and checked on the rest.
struct FBatch { const ifl_train_net* N0; int nz; int64_t a1, p2, a2, hraw, w2, w3, bias3, w2t, w3t, dp1, dp2, dh, an1p, an2p; };

int f_batch(const ifl_train_level* L, FBatch* out)
{
    FBatch F{};
    const ifl_train_net* prev = nullptr;
    for (int s = 0; s < L->nsteps; ++s) {
        if (!L->steps[s].cond) continue;
        const ifl_train_net* N = nullptr;
        for (int i = 0; i < 2; ++i) if (!L->steps[s].net[i].has_z1) N = &L->steps[s].net[i];
        IFL_CHECK_ARG(N && N->dp1 && N->dp2 && N->dh && N->p1_stride == 128, IFL_EINVAL, "f_batched: step %d has no z-independent network on the bulk slab", s);
        if (!F.N0) F.N0 = N;
        else {
            const int64_t d[14] = {N->a1 - prev->a1, N->p2 - prev->p2, N->a2 - prev->a2, N->hraw - prev->hraw, N->w2 - prev->w2, N->w3 - prev->w3,
                                   N->bias3 - prev->bias3, N->w2t - prev->w2t, N->w3t - prev->w3t, N->dp1 - prev->dp1, N->dp2 - prev->dp2,
                                   N->dh - prev->dh, N->an1_part - prev->an1_part, N->an2_part - prev->an2_part};
            int64_t* f = &F.a1;
            for (int k = 0; k < 14; ++k) {
                if (F.nz == 1) f[k] = d[k];
                IFL_CHECK_ARG(f[k] == d[k] && d[k] % 4 == 0, IFL_EINVAL, "f_batched: buffer %d of step %d is not at the common stride", k, s);
            }
            IFL_CHECK_ARG(N->hstride == F.N0->hstride && N->p1 - prev->p1 == L->raw_step, IFL_EINVAL, "f_batched: step %d differs in shape", s);
        }
        prev = N;
        ++F.nz;
    }
    IFL_CHECK_ARG(F.nz >= 1 && (int64_t)F.nz * L->B <= 65535, IFL_EINVAL, "f_batched: %d steps", F.nz);
    *out = F;
    return 0;
}

int conv_b(const float* src, int c, int64_t src_z, const float* w, int64_t w_z, const float* bias, int64_t bias_z, float* out, int cout,
           int64_t out_z, int taps, int nz, int B, int h, int w_, hipStream_t st)
{
    ifl_conv_desc d = {};
    d.src[0] = src; d.src_stride[0] = c; d.src_ch[0] = c; d.nsrc = 1;
    d.w = w; d.bias = bias; d.out = out; d.out_stride = cout; d.cout = cout; d.taps = taps;
    d.alpha = d.beta = 1.f;
    if (nz == 1) return ifl_conv2d(&d, B, h, w_, st);
    return ifl_conv2d_batched(&d, nz, src_z, 0, w_z, bias_z, out_z, 0, B, h, w_, st);
}

int f_forward(const ifl_train_level* L, hipStream_t st)
{
    FBatch F;
    IFL_TRY(f_batch(L, &F));
    const ifl_train_net& N = *F.N0;
    const int B = L->B, h = L->h, w = L->w;
    IFL_TRY(ifl_k_an_relu_fwd(N.p1, 128, N.a1, L->anp, L->anp + 64, B, h, w, 64, st, F.nz, L->raw_step, F.a1, L->anp_step));
    IFL_TRY(conv_b(N.a1, 64, F.a1, N.w2, F.w2, nullptr, 0, N.p2, 64, F.p2, 1, F.nz, B, h, w, st));
    IFL_TRY(ifl_k_an_relu_fwd(N.p2, 64, N.a2, L->anp + 128, L->anp + 192, B, h, w, 64, st, F.nz, F.p2, F.a2, L->anp_step));
    return conv_b(N.a2, 64, F.a2, N.w3, F.w3, N.bias3, F.bias3, N.hraw, N.hstride, F.hraw, 9, F.nz, B, h, w, st);
}

int f_backward(const ifl_train_level* L, hipStream_t st)
{
    FBatch F;
    IFL_TRY(f_batch(L, &F));
    const ifl_train_net& N = *F.N0;
    const int B = L->B, h = L->h, w = L->w;
    const int64_t da_z = (int64_t)B * (h + 2) * (w + 2) * 64;
    IFL_TRY(conv_b(N.dh, N.hstride, F.dh, N.w3t, F.w3t, nullptr, 0, L->da_all, 64, da_z, 9, F.nz, B, h, w, st));
    IFL_TRY(ifl_k_an_relu_bwd(N.p2, 64, L->da_all, N.dp2, L->anp + 128, L->anp + 192, N.an2_part, B, h, w, 64, st, F.nz, F.p2, da_z, F.dp2,
                              L->anp_step, F.an2p));
    IFL_TRY(conv_b(N.dp2, 64, F.dp2, N.w2t, F.w2t, nullptr, 0, L->da_all, 64, da_z, 1, F.nz, B, h, w, st));
    return ifl_k_an_relu_bwd(N.p1, 128, L->da_all, N.dp1, L->anp, L->anp + 64, N.an1_part, B, h, w, 64, st, F.nz, L->raw_step, da_z, F.dp1,
                             L->anp_step, F.an1p);
}

}  // namespace

extern "C" int ifl_train_level_fwd(const ifl_train_level* L, void* stream)
{
    IFL_TRY(check_level(L));
    if (L->f_batched == 1) IFL_TRY(f_forward(L, (hipStream_t)stream));   // 2: its outputs are in the buffers already (the caller copied them from another pass over the same features)
    Groups R;
    IFL_TRY(groups_begin(L, (hipStream_t)stream, &R));
    int rc = 0, cstep = -1;
    for (int s = 0; s < L->nsteps && !rc; ++s) {
        if (L->steps[s].cond) ++cstep;
        for (int k = 0; k < R.n && !rc; ++k) rc = fwd_step(L, R.g[k], s, cstep, R.st[k]);
    }
    return groups_end(R, rc);
}

extern "C" int ifl_train_level_bwd(const ifl_train_level* L, const float* dz_out, const float* dld, float* dz_in, void* stream)
{
    IFL_TRY(check_level(L));
    IFL_CHECK_ARG(dz_out && dz_in && L->dz[0] && L->dz[1] && L->dz[2] && L->dz[3] && L->d64[0] && L->d64[1], IFL_EINVAL, "train level backward: null pointer");
    Groups R;
    IFL_TRY(groups_begin(L, (hipStream_t)stream, &R));
    const float* gp[2];
    for (int k = 0; k < R.n; ++k) gp[k] = dz_out + (int64_t)R.g[k].b0 * L->h * L->w * L->C;
    int rc = 0;
    for (int s = L->nsteps - 1; s >= 0 && !rc; --s)
        for (int k = 0; k < R.n && !rc; ++k) rc = bwd_step(L, R.g[k], s, &gp[k], dld, dz_in, R.st[k]);
    rc = groups_end(R, rc);
    if (!rc && L->f_batched) rc = f_backward(L, (hipStream_t)stream);
    return rc;
}
