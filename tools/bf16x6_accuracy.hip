// accuracy probe: C(32x32) = A(32xK) * B(Kx32) via (a) v_mfma_f32_32x32x2_f32, (b) bf16 3-way split with
// 6 (or 3) v_mfma_f32_32x32x16_bf16 passes; compared with a double reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) {   // RNE
    unsigned u = __float_as_uint(x);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int MODE>   // 0: f32 mfma; 6: bf16x6; 3: bf16x3
__global__ void k(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                float a = A[i * K + k + 8 * h + j], b = B[(k + 8 * h + j) * 32 + i];
                unsigned short x0 = f2bf(a); float r = a - bf2f(x0); unsigned short x1 = f2bf(r); r = r - bf2f(x1); unsigned short x2 = f2bf(r);
                ah[j] = x0; am[j] = x1; al[j] = x2;
                x0 = f2bf(b); r = b - bf2f(x0); x1 = f2bf(r); r = r - bf2f(x1); x2 = f2bf(r);
                bh[j] = x0; bm[j] = x1; bl[j] = x2;
            }
            if (MODE == 6) {   // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
    const int K = 1120;
    std::vector<float> A(32 * K), B(K * 32), C(1024);
    std::vector<double> R(1024), Rabs(1024);
    for (int sgn = 0; sgn < 2; ++sgn) {
        srand(1);
        for (auto& x : A) x = sgn ? (float)rand() / RAND_MAX : (float)rand() / RAND_MAX * 2 - 1;   // post-ReLU-like (>=0) or signed
        for (auto& x : B) x = ((float)rand() / RAND_MAX * 2 - 1) * 0.03f;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0, sa = 0; for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * B[k * 32 + j]; s += p; sa += fabs(p); } R[i * 32 + j] = s; Rabs[i * 32 + j] = sa; }
        float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int mode : {0, 6, 3}) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double emax = 0, esum = 0, bias = 0;
            for (int n = 0; n < 1024; ++n) { double e = (C[n] - R[n]) / Rabs[n]; emax = fmax(emax, fabs(e)); esum += fabs(e); bias += e; }
            printf("inputs %s mode %d: max|err|/sum|ab| = %.3e  mean = %.3e  bias = %.3e\n", sgn ? "A>=0" : "signed", mode, emax, esum / 1024, bias / 1024);
        }
    }
    return 0;
}
