// conv1_fin.h -- pieces shared by the bulk kernels of conv1.hip and conv1x.hip (included inside their anonymous namespace).
#pragma once

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// bf16 operands, fp32 accumulate (BASELINE.json configs[4]): the four k-steps (a[e], b[e]), e = 0..3, of the fp32 form are one
// v_mfma_f32_32x32x8_bf16 -- lane half h holds k = 4h .. 4h+3 in both forms -- with the operands rounded to bf16 (round to
// nearest even, v_cvt_pk_bf16_f32) on the way in.
typedef short ifl_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ifl_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ifl_s16x4 to_bf16x4(f32x4 v) {
    ifl_bf16x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (__bf16)v[e];
    return __builtin_bit_cast(ifl_s16x4, b);
}
template <bool BF16>
__device__ __forceinline__ f32x16 mfma32x4(f32x4 a, f32x4 b, f32x16 c) {
    if (BF16) return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(to_bf16x4(a), to_bf16x4(b), c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

// ActNorm + ReLU on a 64-channel x (NT*32)-pixel accumulator block (flow.py:60-64, FlowActNorms.py:87-92), then the
// 1x1 conv 64 -> 64 straight from the accumulators (the 32x32 C/D layout has the pixel on the lane and the channel in
// the register index, and the 1x1 weights are packed in that channel order), its ActNorm + ReLU, and the store.
// Accumulator element (m, r) of lane (j, h) is channel m*32 + (r&3) + 8*(r>>2) + 4*h; the ActNorm tables are stored in
// that order: an[0/1 = bias/scale][m][h][r].
template <int NT, int STRIDE, bool BF16 = false>
__device__ __forceinline__ void finish_and_store(f32x16 (&acc)[2][NT], const float* __restrict__ an1,
                                                 const float* __restrict__ w2, const float* __restrict__ an2,
                                                 float* __restrict__ h2net, const int64_t (&opix)[NT], int lane, int h)
{
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x4 bb[4], ss[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bb[q] = *(const f32x4*)(an1 + (m * 2 + h) * 16 + q * 4);
            ss[q] = *(const f32x4*)(an1 + 64 + (m * 2 + h) * 16 + q * 4);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[m][n][r] + bb[r >> 2][r & 3]) * ss[r >> 2][r & 3];
                acc[m][n][r] = v > 0.f ? v : 0.f;
            }
    }
    const float* w2l = w2 + lane * 4;
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
                for (int m2 = 0; m2 < 2; ++m2) wv[m2] = *(const f32x4*)(w2l + ((m2 * 2 + m) * 4 + gq) * 256);
                if (BF16) {
                    const f32x4 xk = f32x4{acc[m][n][gq * 4], acc[m][n][gq * 4 + 1], acc[m][n][gq * 4 + 2], acc[m][n][gq * 4 + 3]};
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) y[m2] = mfma32x4<true>(wv[m2], xk, y[m2]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) y[m2] = mfma32(wv[m2][e], acc[m][n][gq * 4 + e], y[m2]);
                }
            }
        if (opix[n] >= 0) {
            float* dst = h2net + opix[n] * STRIDE + 4 * h;
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
