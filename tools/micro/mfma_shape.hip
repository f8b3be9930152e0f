// Bare fp32 MFMA loops on random data: which shape delivers more FLOP/s on MI355X when the chip holds its clock down under load?
// (MI355X_MICROARCH.md, DVFS give-back item 7 measured it for bf16.)  One or two waves per SIMD, every CU busy, operands in
// registers, 9 independent 32x32 accumulators per wave (the bulk kernel's shape) or 36 16x16 accumulators.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape.hip -o gpurun_exp/mfma_shape && gpurun_exp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void loop(const float* __restrict__ in, float* __restrict__ out, int iters)
{
    const int tid = threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = in[(blockIdx.x * 512 + tid) * 16 + k]; b[k] = in[(blockIdx.x * 512 + tid) * 16 + 8 + k]; }
    if (SHAPE == 32) {
        f32x16 acc[9];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 9; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(p + k) & 7], b[(p * 3 + k) & 7], acc[p], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[p][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        f32x4 acc[36];
#pragma unroll
        for (int p = 0; p < 36; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 36; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(p + k) & 7], b[(p * 3 + k) & 7], acc[p], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 36; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[p][r];
        out[blockIdx.x * 512 + tid] = s;
    }
}

int main()
{
    const int nwg = 256 * 8, iters = 4000;
    std::vector<float> h((size_t)nwg * 512 * 16);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *in, *out;
    CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, (size_t)nwg * 512 * 4));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 2; ++w) {
                if (w) CK(hipEventRecord(e0));
                const int n = w ? 5 : 2;
                for (int i = 0; i < n; ++i) {
                    if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(nwg), dim3(512), 0, 0, in, out, iters);
                    else hipLaunchKernelGGL(loop<16>, dim3(nwg), dim3(512), 0, 0, in, out, iters);
                }
                if (w) {
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    // per wave and iteration: 32: 36 MFMAs of 4096 FLOP; 16: 72 MFMAs of 2048 FLOP
                    const double fl = (double)n * nwg * 8 * iters * 36 * 4096.0;
                    printf("shape %dx%d: %.2f ms for %d launches  %.1f TFLOP/s\n", shape, shape, ms, n, fl / (ms * 1e-3) / 1e12);
                } else CK(hipDeviceSynchronize());
            }
        }
    return 0;
}
