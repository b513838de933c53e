// accuracy probe for a possible next step: f32 products as a 2-way f16 split (hi + lo, 22 mantissa bits)
// with 3 (hh, hl, lh) or 4 (+ ll) v_mfma_f32_32x32x16_f16 passes and power-of-two operand scaling, against
// the exact-f32 MFMA chain, the bf16 3-way/6-pass split in use today, and a double reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) { unsigned u = __float_as_uint(x); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
__device__ inline float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int MODE>   // 0: f32 mfma; 6: bf16x6; 23: f16x2, 3 passes; 24: f16x2, 4 passes
__global__ void k(const float* A, const float* B, float* C, int K, float sa, float sb) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int kk = 0; kk < K; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + kk + h], B[(kk + h) * 32 + i], acc, 0, 0, 0);
    } else if (MODE == 6) {
        for (int kk = 0; kk < K; kk += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                float a = A[i * K + kk + 8 * h + j], b = B[(kk + 8 * h + j) * 32 + i];
                unsigned short x0 = f2bf(a); float r = a - bf2f(x0); unsigned short x1 = f2bf(r); r = r - bf2f(x1); unsigned short x2 = f2bf(r);
                ah[j] = x0; am[j] = x1; al[j] = x2;
                x0 = f2bf(b); r = b - bf2f(x0); x1 = f2bf(r); r = r - bf2f(x1); x2 = f2bf(r);
                bh[j] = x0; bm[j] = x1; bl[j] = x2;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    } else {
        for (int kk = 0; kk < K; kk += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float a = A[i * K + kk + 8 * h + j] * sa, b = B[(kk + 8 * h + j) * 32 + i] * sb;
                ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
                bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
            }
            if (MODE == 24) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc[r] *= 1.0f / (sa * sb);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

int main() {
    struct Case { const char* name; int K; int kind; float sa, sb; };
    const Case cases[] = {
        {"fwd 280ch: relu(N(0,1)) x U(+-0.03), no scaling", 1120, 0, 1.f, 1.f},
        {"fwd 280ch: relu(N(0,1)) x U(+-0.03), a*2^4 w*2^10", 1120, 0, 16.f, 1024.f},
        {"fwd 27ch:  U(0,1) x U(+-0.1), a*2^4 w*2^8", 112, 1, 16.f, 256.f},
        {"dgrad: g ~ 1e-5*N(0,1) (half zeros) x U(+-0.03), g*2^24 w*2^10", 1120, 2, 16777216.f, 1024.f},
        {"wide dynamic range: a = N(0,1)*10^U(-3,1), a*2^2 w*2^10", 1120, 3, 4.f, 1024.f},
    };
    for (const Case& cs : cases) {
        const int K = cs.K;
        std::vector<float> A(32 * K), B(K * 32), C(1024);
        std::vector<double> R(1024), Rabs(1024);
        srand(7);
        for (auto& x : A) {
            double g = gauss();
            if (cs.kind == 0) x = (float)fmax(g, 0.0);
            else if (cs.kind == 1) x = (float)rand() / RAND_MAX;
            else if (cs.kind == 2) x = (rand() & 1) ? (float)(1e-5 * g) : 0.f;
            else x = (float)(g * pow(10.0, -3.0 + 4.0 * rand() / RAND_MAX));
        }
        const float wr = cs.kind == 1 ? 0.1f : 0.03f;
        for (auto& x : B) x = ((float)rand() / RAND_MAX * 2 - 1) * wr;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0, sa = 0; for (int kk = 0; kk < K; ++kk) { double p = (double)A[i * K + kk] * B[kk * 32 + j]; s += p; sa += fabs(p); } R[i * 32 + j] = s; Rabs[i * 32 + j] = sa; }
        float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s (K=%d)\n", cs.name, K);
        for (int mode : {0, 6, 23, 24}) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, cs.sa, cs.sb);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, cs.sa, cs.sb);
            if (mode == 23) hipLaunchKernelGGL(k<23>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, cs.sa, cs.sb);
            if (mode == 24) hipLaunchKernelGGL(k<24>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, cs.sa, cs.sb);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double emax = 0, esum = 0;
            for (int n = 0; n < 1024; ++n) { double e = (C[n] - R[n]) / Rabs[n]; emax = fmax(emax, fabs(e)); esum += fabs(e); }
            printf("   mode %2d: mean |err|/sum|ab| = %.3e   max = %.3e\n", mode, esum / 1024, emax);
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
