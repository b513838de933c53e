// Accuracy probe for DESIGN.md section 4.8: the k=2 convolution of the 280 -> 280 layers written as minimal filtering
// (Winograd class), F(2,2) along W (6 instead of 8 tap-products per output pair) and F(2x2,2x2) (9 instead of 16 per
// 2x2 output tile), evaluated in the library's default arithmetic (f32 operands scaled by a power of two, split into two
// f16, three v_mfma_f32_*_f16 passes, f32 accumulate) -- against the direct form in the same arithmetic, the exact-f32
// MFMA chain, and a double evaluation of the DIRECT convolution (the true answer for all of them).
// The transforms have entries 0 / +-1: inputs d0-d1, d1, d2-d1 (formed in f32, as a kernel would), filters g0, g0+g1, g1
// (formed in f32 by the packer); an output is the sum of 2 (1-D, per filter row) or 4 (2-D) transformed products, so for
// one output the contraction length is the direct form's 4 x 280.
//   hipcc -O3 --offload-arch=gfx950 tools/winograd_accuracy.hip -o /tmp/winograd_accuracy && /tmp/winograd_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: exact-f32 MFMA chain; 23: f16 split, 3 passes
__global__ void gemm32(const float *A, const float *B, float *C, int K, float sa, float sb)
{
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int kk = 0; kk < K; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + kk + h], B[(kk + h) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int kk = 0; kk < K; kk += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float a = A[i * K + kk + 8 * h + j] * sa, b = B[(kk + 8 * h + j) * 32 + i] * sb;
                ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
                bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc[r] *= 1.0f / (sa * sb);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

static double gauss()
{
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}
static float pow2_scale(const std::vector<float> &v)       // max |x| -> [2^14, 2^15), as the library's pow2_scale_for
{
    float m = 0.f;
    for (float x : v) m = fmaxf(m, fabsf(x));
    if (m == 0.f) return 1.f;
    return exp2f(14.f - floorf(log2f(m)));
}

int main()
{
    const int C = 280, NT = 32, NO = 32;     // channels, 2x2 output tiles (MFMA rows), output channels (MFMA columns)
    struct Case { const char *name; int kind; };
    const Case cases[] = {{"forward: relu(N(0,1)) activations x U(+-0.03) weights", 0},
                          {"data gradient: 1e-5 N(0,1), half zeros x U(+-0.03)", 1},
                          {"five decades of dynamic range: N(0,1) 10^U(-3,1) x U(+-0.03)", 2}};
    float *dA, *dB, *dC;
    hipMalloc(&dA, NT * 4 * C * 4); hipMalloc(&dB, 4 * C * NO * 4); hipMalloc(&dC, NT * NO * 4);
    for (const Case &cs : cases) {
        srand(11);
        // d[t][y][x][c]: the 3x3 input patch of tile t; g[ky][kx][c][o]
        std::vector<float> d((size_t)NT * 9 * C), g((size_t)4 * C * NO);
        for (auto &x : d) {
            const double z = gauss();
            x = cs.kind == 0 ? (float)fmax(z, 0.0) : cs.kind == 1 ? ((rand() & 1) ? (float)(1e-5 * z) : 0.f)
                                                                   : (float)(z * pow(10.0, -3.0 + 4.0 * rand() / RAND_MAX));
        }
        for (auto &x : g) x = ((float)rand() / RAND_MAX * 2 - 1) * 0.03f;
        auto D = [&](int t, int y, int x, int c) -> float { return d[((size_t)(t * 3 + y) * 3 + x) * C + c]; };
        auto G = [&](int ky, int kx, int c, int o) -> float { return g[((size_t)(ky * 2 + kx) * C + c) * NO + o]; };
        // the true outputs (a, b) of every tile in double, and sum |a*b| of the DIRECT form (the error unit)
        std::vector<double> ref((size_t)4 * NT * NO), mag((size_t)4 * NT * NO);
        for (int ab = 0; ab < 4; ++ab)
            for (int t = 0; t < NT; ++t)
                for (int o = 0; o < NO; ++o) {
                    double s = 0, m = 0;
                    for (int ky = 0; ky < 2; ++ky)
                        for (int kx = 0; kx < 2; ++kx)
                            for (int c = 0; c < C; ++c) {
                                const double p = (double)D(t, (ab >> 1) + ky, (ab & 1) + kx, c) * G(ky, kx, c, o);
                                s += p; m += fabs(p);
                            }
                    ref[((size_t)ab * NT + t) * NO + o] = s; mag[((size_t)ab * NT + t) * NO + o] = m;
                }
        // 1-D transforms (f32 arithmetic, as a kernel / the packer would form them)
        auto tin = [&](int i, float x0, float x1, float x2) -> float { return i == 0 ? x0 - x1 : i == 1 ? x1 : x2 - x1; };
        auto tfl = [&](int i, float g0, float g1) -> float { return i == 0 ? g0 : i == 1 ? g0 + g1 : g1; };
        printf("%s\n", cs.name);
        for (int form = 0; form < 3; ++form) {           // 0 direct, 1 F(2,2) along W, 2 F(2x2,2x2)
            for (int mode : {0, 23}) {
                double esum = 0, emax = 0;
                for (int ab = 0; ab < 4; ++ab) {         // output (a, b) of the tile: one GEMM of contraction length 4*C each
                    const int a = ab >> 1, b = ab & 1;
                    std::vector<float> A((size_t)NT * 4 * C), B((size_t)4 * C * NO);
                    for (int part = 0; part < 4; ++part) {
                        const int p = part >> 1, q = part & 1;
                        for (int c = 0; c < C; ++c) {
                            for (int t = 0; t < NT; ++t) {
                                float v;
                                if (form == 0) v = D(t, a + p, b + q, c);
                                else if (form == 1)      // rows direct (filter row p), columns transformed: components j = b + q
                                    v = tin(b + q, D(t, a + p, 0, c), D(t, a + p, 1, c), D(t, a + p, 2, c));
                                else {                   // components (i, j) = (a + p, b + q), rows then columns
                                    float r[3];
                                    for (int x = 0; x < 3; ++x) r[x] = tin(a + p, D(t, 0, x, c), D(t, 1, x, c), D(t, 2, x, c));
                                    v = tin(b + q, r[0], r[1], r[2]);
                                }
                                A[(size_t)t * 4 * C + part * C + c] = v;
                            }
                            for (int o = 0; o < NO; ++o) {
                                float v;
                                if (form == 0) v = G(p, q, c, o);
                                else if (form == 1) v = tfl(b + q, G(p, 0, c, o), G(p, 1, c, o));
                                else v = tfl(b + q, tfl(a + p, G(0, 0, c, o), G(1, 0, c, o)), tfl(a + p, G(0, 1, c, o), G(1, 1, c, o)));
                                B[((size_t)part * C + c) * NO + o] = v;
                            }
                        }
                    }
                    const float sa = pow2_scale(A), sb = pow2_scale(B);
                    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                    if (mode == 0) hipLaunchKernelGGL(gemm32<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 4 * C, sa, sb);
                    else hipLaunchKernelGGL(gemm32<23>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 4 * C, sa, sb);
                    std::vector<float> out((size_t)NT * NO);
                    hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost);
                    for (int n = 0; n < NT * NO; ++n) {
                        const double e = fabs(out[n] - ref[(size_t)ab * NT * NO + n]) / mag[(size_t)ab * NT * NO + n];
                        esum += e; emax = fmax(emax, e);
                    }
                }
                printf("   %-22s %-10s mean |err| / sum|a*b| = %.3e   max = %.3e\n",
                       form == 0 ? "direct (16 products)" : form == 1 ? "F(2,2) along W (12)" : "F(2x2,2x2) (9)",
                       mode == 0 ? "exact f32" : "f16x3", esum / (4 * NT * NO), emax);
            }
        }
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
    return 0;
}
