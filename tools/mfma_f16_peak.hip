// What the f16 matrix cores sustain on this chip with nothing else running: v_mfma_f32_16x16x32_f16 from registers
// only (no LDS, no memory), 8 waves per CU on every CU, for ~20 ms -- with operands of random bits, and with zeros.
// The nominal dense peak (2.5 PFLOP/s at 2.4 GHz) is a clock figure; under the board's power limit the sustained
// rate depends on the data.     hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_peak.hip -o /tmp/mfma_f16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float *out, int iters, unsigned seed, unsigned mask)
{
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // operands: random bit patterns confined to normal f16 magnitudes 2^-4 .. 2^3 (no inf / nan)
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    u32x4 ua[4], ub[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u;
            ua[k][j] = ((s & 0x83ff83ffu) | 0x2c002c00u | ((s >> 3) & 0x1c001c00u)) & mask;
            s = s * 1664525u + 1013904223u;
            ub[k][j] = ((s & 0x83ff83ffu) | 0x2c002c00u | ((s >> 3) & 0x1c001c00u)) & mask;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua[k]), __builtin_bit_cast(f16x8, ub[(k + n) & 3]),
                                                                acc[n], 0, 0, 0);
    }
    float t = 0.f;
    for (int n = 0; n < NACC; ++n) t += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

static void run(const char *name, unsigned mask, int iters, int threads = 512)
{
    constexpr int NACC = 16;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, out, iters, 1u, mask);   // warm-up, reaches the power limit
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, out, iters, 2u, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * (threads / 64) * iters * 4 * NACC * (2.0 * 16 * 16 * 32);
    const double tf = flop / ms / 1e9;
    // 16 cycles per MFMA and SIMD: the clock the rate corresponds to if the matrix cores never idle
    printf("%-12s %7.2f ms  %7.1f TFLOP/s  = %.3f of 2500  (>= %.2f GHz matrix-core clock)\n", name, ms, tf, tf / 2500.0,
           tf * 1e12 / (cus * 4.0 * 1024.0) / 1e9);
    hipFree(out);
}

int main()
{
    run("random bits", 0xffffffffu, 60000);
    run("zeros", 0u, 60000);
    run("random bits", 0xffffffffu, 60000);
    run("random, 1 wave/SIMD", 0xffffffffu, 120000, 256);
    return 0;
}
