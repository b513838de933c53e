// How v_mfma_f32_16x16x32_f16 rounds: the basis of the per-product constant of the f16-split arithmetic.
// D = C + sum_k a_k b_k with C and D float32.  The f16 x f16 products are exact in float32 (11 + 11 bits); what is NOT
// specified is how the 32 products and C are added.  Each pattern below has an exactly known real result that is not a
// float32, chosen so that round-to-nearest-even, truncation, and per-product sequential rounding give different floats.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rounding.hip -o /tmp/mfma_rounding && /tmp/mfma_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// lane (r16, q4) supplies A[row r16][k = 8*q4 + j], B[k = 8*q4 + j][col r16]; D[row 4*q4 + r][col r16] = acc[r]
__global__ void probe(const float *a /*[32] row 0*/, const float *b /*[32] col 0*/, float c, float *out)
{
    const int lane = threadIdx.x, r16 = lane & 15, q4 = lane >> 4;
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = r16 == 0 ? (_Float16)a[8 * q4 + j] : (_Float16)0.f;
        bv[j] = r16 == 0 ? (_Float16)b[8 * q4 + j] : (_Float16)0.f;
    }
    f32x4 acc = {c, c, c, c};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];          // D[0][0]
}

static float run(const float *a, const float *b, float c)
{
    float *da, *db, *dout, out;
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return out;
}
static void show(const char *what, float got, double exact, const char *note)
{
    const float ulp = ldexpf(1.f, -23);                 // every pattern's result lies in [1, 2) or just under 1
    printf("%-74s got 1 %+7.3f ulp   exact 1 %+7.3f ulp   %s\n", what, (double)(got - 1.0) / ulp, (exact - 1.0) / ulp, note);
}

int main()
{
    float a[32], b[32];
    auto zero = [&]() { memset(a, 0, sizeof a); memset(b, 0, sizeof b); };
    const float u = ldexpf(1.f, -23);                   // ulp of 1.0
    // P1: C = 1, one product of +0.75 ulp
    zero(); a[0] = ldexpf(1.5f, -12); b[0] = ldexpf(1.f, -12);
    show("P1  C=1 + one product 0.75 ulp", run(a, b, 1.f), 1.0 + 0.75 * u, "nearest: +1, truncation: 0");
    // P2: C = 1, one product of -0.125 ulp (below 1 the spacing is half an ulp: -0.125 ulp = a quarter of it)
    zero(); a[0] = -ldexpf(1.f, -13); b[0] = ldexpf(1.f, -13);
    show("P2  C=1 + one product -0.125 ulp", run(a, b, 1.f), 1.0 - 0.125 * u, "nearest: 0, truncation: -0.5");
    // P3: C = 1, 32 products of 1/8 ulp each = 4 ulp in all
    zero(); for (int k = 0; k < 32; ++k) { a[k] = ldexpf(1.f, -13); b[k] = ldexpf(1.f, -13); }
    show("P3  C=1 + 32 products of 0.125 ulp (sum 4 ulp)", run(a, b, 1.f), 1.0 + 4 * u, "products summed first: +4, one by one truncated: 0");
    // P4: C = 1, 4 products of 0.375 ulp (sum 1.5 ulp), in four different lane quarters
    zero(); for (int k = 0; k < 4; ++k) { a[8 * k] = ldexpf(1.5f, -12); b[8 * k] = ldexpf(1.f, -13); }
    show("P4  C=1 + 4 products of 0.375 ulp (sum 1.5 ulp), one per lane quarter", run(a, b, 1.f), 1.0 + 1.5 * u, "exact sum then nearest-even: +2, then truncation: +1");
    // P5: C = 0, products 1.0 and 0.75 ulp
    zero(); a[0] = 1.f; b[0] = 1.f; a[1] = ldexpf(1.5f, -12); b[1] = ldexpf(1.f, -12);
    show("P5  C=0 + products 1.0 and 0.75 ulp", run(a, b, 0.f), 1.0 + 0.75 * u, "nearest: +1, truncation: 0");
    // P6: C = 1, products +2^10 and -2^10 and 0.75 ulp: a wide internal adder keeps the small term through the cancellation
    zero(); a[0] = 32.f; b[0] = 32.f; a[1] = -32.f; b[1] = 32.f; a[2] = ldexpf(1.5f, -12); b[2] = ldexpf(1.f, -12);
    show("P6  C=1 + products +1024, -1024, 0.75 ulp", run(a, b, 1.f), 1.0 + 0.75 * u, "small term survives the cancellation?");
    // P7: C = 1, 32 products of 0.046875 ulp (sum 1.5 ulp): how far below the result's ulp are product bits kept?
    zero(); for (int k = 0; k < 32; ++k) { a[k] = ldexpf(1.5f, -14); b[k] = ldexpf(1.f, -14); }
    show("P7  C=1 + 32 products of 3/64 ulp (sum 1.5 ulp)", run(a, b, 1.f), 1.0 + 1.5 * u, "bits kept below the ulp: +1 or +2; dropped: 0");
    // P8: C = 1, one product of 0.5 ulp exactly (a tie) and of 1.5 ulp (a tie the other way)
    zero(); a[0] = ldexpf(1.f, -12); b[0] = ldexpf(1.f, -12);
    show("P8  C=1 + one product 0.5 ulp (tie)", run(a, b, 1.f), 1.0 + 0.5 * u, "nearest-even: 0, nearest-away: +1");
    zero(); a[0] = ldexpf(1.5f, -11); b[0] = ldexpf(1.f, -12);
    show("P9  C=1 + one product 1.5 ulp (tie)", run(a, b, 1.f), 1.0 + 1.5 * u, "nearest-even: +2, truncation: +1");
    // P10/P11: an f16 SUBNORMAL operand (2^-20; the smallest normal is 2^-14): used or flushed to zero?
    zero(); a[0] = ldexpf(1.f, -20); b[0] = ldexpf(1.f, -3);
    show("P10 C=1 + one product 2^-20 (subnormal f16) x 2^-3 = 1 ulp", run(a, b, 1.f), 1.0 + 1.0 * u, "subnormals used: +1, flushed: 0");
    zero(); a[0] = ldexpf(1.f, -3); b[0] = ldexpf(1.5f, -21);
    show("P11 C=1 + one product 2^-3 x 1.5 2^-21 (subnormal f16 B) = 0.75 ulp", run(a, b, 1.f), 1.0 + 0.75 * u, "subnormals used: +1 (nearest) or 0 (truncation); flushed: 0");
    zero(); a[0] = ldexpf(1.f, -20); b[0] = ldexpf(1.f, 0);
    show("P12 C=0 + one product 2^-20 (subnormal) x 1: result 2^-20 or 0", 1.f + run(a, b, 0.f) * ldexpf(1.f, 20) * u, 1.0 + u, "used: +1, flushed: 0");
    return 0;
}
