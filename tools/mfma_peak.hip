#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NT, int DEP>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NT];
    for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (DEP) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[n], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(s & 1 ? a : b, s & 2 ? a : b, acc[n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NT, int DEP> void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    int iters = 4000, blocks = 256 * (512 / threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, DEP>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, DEP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * (threads / 64) * iters * NT * 4 * 4096.0;
    printf("%s NT=%d dep=%d threads=%d: %.3f ms %.1f TFLOP/s\n", name, NT, DEP, threads, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    run<9, 1>("mfma", 512); run<9, 0>("mfma", 512);
    run<9, 1>("mfma", 256); run<9, 0>("mfma", 256);
    run<3, 1>("mfma", 512); run<3, 0>("mfma", 512);
    run<1, 1>("mfma", 512);
    return 0;
}
