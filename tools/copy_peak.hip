// What a 1:1 read/write stream reaches on this part -- the ceiling of the BatchNorm apply passes.
// A float4 copy of N bytes in several shapes (read + written bytes per second), and a read-only sum beside it.
//   hipcc -O3 --offload-arch=gfx950 tools/copy_peak.hip -o /tmp/copy_peak && /tmp/copy_peak [GB per buffer, default 5.5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride, one float4 per thread and step
template <bool NT_LD, bool NT_ST, int U>
__global__ __launch_bounds__(256) void copy_gs(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT_ST) __builtin_nontemporal_store(v[u], out + i + u * stride);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// one workgroup per contiguous row of `row` float4 (the BatchNorm row passes: 98 positions x 280 channels = 6 860 float4)
template <bool NT_LD, bool NT_ST, int U>
__global__ __launch_bounds__(256) void copy_rows(const f4 *__restrict__ in, f4 *__restrict__ out, int row, size_t nrows)
{
    for (size_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const f4 *ip = in + r * row;
        f4 *op = out + r * row;
        int i = threadIdx.x;
        for (; i + (U - 1) * 256 < row; i += U * 256) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(ip + i + u * 256) : ip[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NT_ST) __builtin_nontemporal_store(v[u], op + i + u * 256);
                else op[i + u * 256] = v[u];
            }
        }
        for (; i < row; i += 256) op[i] = ip[i];
    }
}

// the same with NT threads per row: fewer rows in flight on the chip at a time (2 048 resident threads per CU either way),
// each swept in fewer steps -- a more compact frontier in memory
template <int NT, int U>
__global__ __launch_bounds__(NT) void copy_rows_wide(const f4 *__restrict__ in, f4 *__restrict__ out, int row, size_t nrows)
{
    for (size_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const f4 *ip = in + r * row;
        f4 *op = out + r * row;
        int i = threadIdx.x;
        for (; i + (U - 1) * NT < row; i += U * NT) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(ip + i + u * NT);
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], op + i + u * NT);
        }
        for (; i < row; i += NT) __builtin_nontemporal_store(__builtin_nontemporal_load(ip + i), op + i);
    }
}

// y = max(a * x + b, 0) per channel group: the arithmetic of the apply pass on the copy skeleton (coefficients in registers)
template <int U>
__global__ __launch_bounds__(256) void affine_rows(const f4 *__restrict__ in, f4 *__restrict__ out, int row, size_t nrows,
                                                   const f4 *__restrict__ sc, const f4 *__restrict__ sh, int cvn)
{
    for (size_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const f4 *ip = in + r * row;
        f4 *op = out + r * row;
        int i = threadIdx.x;
        for (; i + (U - 1) * 256 < row; i += U * 256) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(ip + i + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = (i + u * 256) % cvn;
                const f4 a = sc[c], b = sh[c];
                f4 y;
                y.x = fmaxf(fmaf(v[u].x, a.x, b.x), 0.f); y.y = fmaxf(fmaf(v[u].y, a.y, b.y), 0.f);
                y.z = fmaxf(fmaf(v[u].z, a.z, b.z), 0.f); y.w = fmaxf(fmaf(v[u].w, a.w, b.w), 0.f);
                __builtin_nontemporal_store(y, op + i + u * 256);
            }
        }
        for (; i < row; i += 256) op[i] = ip[i];
    }
}

__global__ __launch_bounds__(256) void fill_gs(f4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, out + i);
}

template <int U>
__global__ __launch_bounds__(256) void read_gs(const f4 *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

template <typename F>
static double time_ms(F &&launch, int reps = 7)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 5.5;
    const int row = 98 * 280 / 4;                          // float4 per grid row of a 96 x 96 patch at 280 channels
    const size_t nrows = (size_t)(gb * 1e9 / (row * 16.0));
    const size_t n = nrows * row;
    const double bytes = (double)n * 16;
    f4 *in, *out, *sc, *sh; float *sink;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&sc, 70 * 16)); CK(hipMalloc(&sh, 70 * 16));
    CK(hipMemset(in, 0x3c, n * 16)); CK(hipMemset(out, 0, n * 16)); CK(hipMemset(sc, 0x3c, 70 * 16)); CK(hipMemset(sh, 0, 70 * 16));
    printf("buffer %.2f GB (%zu rows of %d float4)\n", bytes / 1e9, nrows, row);
#define RUN(name, bytes_moved, ...) do { const double ms = time_ms([&] { __VA_ARGS__; }); \
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, (bytes_moved) / ms / 1e9); fflush(stdout); } while (0)
    for (int wg : {2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy grid-stride plain U4 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((copy_gs<false, false, 4>), dim3(wg), dim3(256), 0, 0, in, out, n));
        snprintf(nm, sizeof nm, "copy grid-stride nt/nt U4 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 4>), dim3(wg), dim3(256), 0, 0, in, out, n));
    }
    RUN("copy grid-stride nt ld, plain st U4 8192", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, false, 4>), dim3(8192), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride plain ld, nt st U4 8192", 2 * bytes, hipLaunchKernelGGL((copy_gs<false, true, 4>), dim3(8192), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U1 8192", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 1>), dim3(8192), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U8 8192", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 8>), dim3(8192), dim3(256), 0, 0, in, out, n));
    // long same-direction bursts: every resident thread loads U float4 before it stores any (2 048 blocks = all resident at once:
    // 0.5 / 1 MB in flight per CU, the whole chip reading 128 / 256 MB, then writing them)
    RUN("copy grid-stride nt/nt U16 grid 2048", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 16>), dim3(2048), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U32 grid 2048", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 32>), dim3(2048), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride plain U32 grid 2048", 2 * bytes, hipLaunchKernelGGL((copy_gs<false, false, 32>), dim3(2048), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U32 grid 1024", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 32>), dim3(1024), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U16 grid 4096", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 16>), dim3(4096), dim3(256), 0, 0, in, out, n));
    RUN("write-only nt grid 8192 (fill)", bytes, hipLaunchKernelGGL(fill_gs, dim3(8192), dim3(256), 0, 0, out, n));
    for (int wg : {2048, 8192, (int)nrows}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy rows plain U4 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((copy_rows<false, false, 4>), dim3(wg), dim3(256), 0, 0, in, out, row, nrows));
        snprintf(nm, sizeof nm, "copy rows nt/nt U4 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((copy_rows<true, true, 4>), dim3(wg), dim3(256), 0, 0, in, out, row, nrows));
        snprintf(nm, sizeof nm, "copy rows nt/nt U8 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((copy_rows<true, true, 8>), dim3(wg), dim3(256), 0, 0, in, out, row, nrows));
        snprintf(nm, sizeof nm, "affine+relu rows nt/nt U4 grid %d", wg);
        RUN(nm, 2 * bytes, hipLaunchKernelGGL((affine_rows<4>), dim3(wg), dim3(256), 0, 0, in, out, row, nrows, sc, sh, 70));
    }
    RUN("copy rows nt/nt 512 thr U2, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<512, 2>), dim3(nrows), dim3(512), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 512 thr U4, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<512, 4>), dim3(nrows), dim3(512), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 1024 thr U1, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<1024, 1>), dim3(nrows), dim3(1024), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 1024 thr U2, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<1024, 2>), dim3(nrows), dim3(1024), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 1024 thr U4, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<1024, 4>), dim3(nrows), dim3(1024), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 256 thr U1, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<256, 1>), dim3(nrows), dim3(256), 0, 0, in, out, row, nrows));
    RUN("copy rows nt/nt 256 thr U4, block per row", 2 * bytes, hipLaunchKernelGGL((copy_rows_wide<256, 4>), dim3(nrows), dim3(256), 0, 0, in, out, row, nrows));
    RUN("copy grid-stride nt/nt U4 grid 16384 (again)", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 4>), dim3(16384), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U2 grid 32768", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 2>), dim3(32768), dim3(256), 0, 0, in, out, n));
    RUN("copy grid-stride nt/nt U1 grid 65536", 2 * bytes, hipLaunchKernelGGL((copy_gs<true, true, 1>), dim3(65536), dim3(256), 0, 0, in, out, n));
    RUN("read-only sum nt U4 grid 8192", bytes, hipLaunchKernelGGL((read_gs<4>), dim3(8192), dim3(256), 0, 0, in, sink, n));
    RUN("read-only sum nt U8 grid 16384", bytes, hipLaunchKernelGGL((read_gs<8>), dim3(16384), dim3(256), 0, 0, in, sink, n));
    {
        const double ms = time_ms([&] { CK(hipMemcpyAsync(out, in, n * 16, hipMemcpyDeviceToDevice, 0)); });
        printf("%-44s %8.3f ms  %6.2f TB/s\n", "hipMemcpyAsync device to device", ms, 2 * bytes / ms / 1e9);
    }
    return 0;
}
