// bare v_mfma_f32_32x32x16_bf16 / 16x16x32 rate under sustained load (random operands in registers):
// what the chip can hold at its power-limited clock, the ceiling for the split conv kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline bf16x8 rnd(unsigned s) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; v[j] = (short)(0x3c00 + ((s >> 12) & 0x3ff) - ((s >> 9) & 0x8000)); }
    return v;
}
template <int SHAPE, int NT>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p) { a[p] = rnd(threadIdx.x * 7 + p); b[p] = rnd(threadIdx.x * 13 + p + 5); }
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[NT];
        for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[n], 0, 0, 0);
            }
        }
        for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    } else {
        f32x4 acc[NT * 4];
        for (int n = 0; n < NT * 4; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int n = 0; n < NT * 4; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[n], 0, 0, 0);
            }
        }
        for (int n = 0; n < NT * 4; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE> void run(int threads) {
    float* out; (void)hipMalloc(&out, 1 << 22);
    const int NT = 9, iters = 20000, blocks = 256 * (512 / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NT>), dim3(blocks), dim3(threads), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NT>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * (threads / 64) * iters * NT * 6 * 32768.0;
    printf("bf16 mfma %dx%d waves/CU=%d: %.1f ms  %.0f TFLOP/s bf16 = %.0f TFLOP/s f32-equivalent (6 passes)\n", SHAPE, SHAPE, threads / 64 * (512 / threads), ms, fl / ms / 1e9, fl / ms / 1e9 / 6);
    (void)hipFree(out);
}
int main() { run<32>(512); run<32>(256); run<16>(512); run<16>(256); return 0; }
