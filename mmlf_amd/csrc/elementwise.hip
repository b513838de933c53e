// elementwise.hip -- HBM-bound kernels around the convolutions: batch-norm statistics / apply /
// backward, layout pack/unpack, UPR / DPP heads, losses, Adam, Shift and the ensemble reduce.
// gfx950 only.  Reference call sites are cited per entry point in include/mmlf_hip.h.
#include "common.h"
#include "../../include/mmlf_hip.h"

thread_local char g_mmlf_err[512] = "";

extern "C" const char *mmlf_last_error(void) { return g_mmlf_err; }
extern "C" int mmlf_abi_version(void) { return MMLF_ABI_VERSION; }
extern "C" int64_t mmlf_amax_entries(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return amax_entries(make_grid(B, H, W));
}
extern "C" int mmlf_grid_pad_w(void) { return MMLF_GRID_PAD_W; }
extern "C" int mmlf_grid_pad_h(void) { return MMLF_GRID_PAD_H; }
extern "C" int mmlf_amax_head(void) { return MMLF_AMAX_HEAD; }
extern "C" int mmlf_amax_shard_stride(void) { return MMLF_AMAX_SHARD_STRIDE; }
extern "C" int64_t mmlf_grid_alloc_positions(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return grid_alloc_positions(make_grid(B, H, W));
}

// ---------------------------------------------------------------------------------------------
// per-channel reductions over grid rows.  A block walks grid rows r = blockIdx.x, +gridDim.x, ...
// (border rows are all zero and skipped).  Threads map to (position-in-group, float4 channel
// group) so that a thread keeps a fixed channel group; double accumulators.
// ---------------------------------------------------------------------------------------------
template <int V> struct VecIO;
// Four floats at an address that is 16-byte aligned -- or only 8-byte aligned: the 70-channel stream slices of the
// 280-channel concat buffer start at 280-byte multiples, and a float4 group of such a slice is two float2 accesses
// (uniform over the launch: the same branch for every lane).  Half the threads and instructions of a float2 kernel.
template <> struct VecIO<4> {
    static __device__ __forceinline__ bool aligned(const float *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
    static __device__ __forceinline__ void load(const float *p, float *o)
    {
        if (aligned(p)) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        } else {
            const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2);
            o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
        }
    }
    static __device__ __forceinline__ void store(float *p, const float *o)
    {
        if (aligned(p)) {
            *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            *reinterpret_cast<float2 *>(p) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2 *>(p + 2) = make_float2(o[2], o[3]);
        }
    }
    // streamed once: keep it out of the way of data that is re-read
    static __device__ __forceinline__ void load_nt(const float *p, float *o)
    {
        if (aligned(p)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        } else {
            load(p, o);
        }
    }
    // the same with the alignment decided once per launch by the caller (a scalar branch, no per-lane address test)
    static __device__ __forceinline__ void load_nt(const float *p, float *o, bool is_aligned)
    {
        if (is_aligned) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        } else {
            const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2);
            o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
        }
    }
    static __device__ __forceinline__ void store_nt(float *p, const float *o)
    {
        if (aligned(p)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
        } else {
            store(p, o);
        }
    }
};

// V = channels per thread access (4; slices that are only 8-byte aligned go through VecIO<4>'s two-float2 form).
template <bool BWD, int V>
__device__ __forceinline__ void bn_reduce_body(const float *__restrict__ z, int cs_z,
                                                        const float *__restrict__ gy, int cs_gy, int c_off,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ shift,
                                                        const float *__restrict__ mean,
                                                        const float *__restrict__ invstd, int C,
                                                        double *__restrict__ partial, int B, int H, int W)
{
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int cvn = (C + V - 1) / V;            // channel groups that hold real channels
    const int ppi = 256 / cvn;                  // positions per iteration
    const int tid = threadIdx.x;
    const int pl = tid / cvn, cg = tid - pl * cvn;
    const bool active = pl < ppi;
    double s0[V], s1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s0[e] = 0; s1[e] = 0; }
    // One LDS block, two uses: the backward pass keeps its four per-channel coefficients here while it streams (re-read per
    // position, NOT held in registers: with them the kernel needs 62 VGPRs, without 48 or fewer -- what is left per SIMD beside
    // the two waves of a wide weight-gradient workgroup, so that this HBM-bound pass can run under that matrix-core-bound one,
    // engine.py OVERLAP_WGRAD); afterwards the block's cross-thread sums.
    __shared__ double red[2][256][V];
    float *cf = reinterpret_cast<float *>(&red[0][0][0]);          // [4][256 * V] floats = half of red
    if (BWD) {
        for (int c = tid; c < cvn * V; c += 256) {
            const bool in = c < C;
            cf[0 * 256 * V + c] = in ? scale[c] : 0.f;
            cf[1 * 256 * V + c] = in ? shift[c] : 0.f;
            cf[2 * 256 * V + c] = in ? mean[c] : 0.f;
            cf[3 * 256 * V + c] = in ? invstd[c] : 0.f;
        }
        __syncthreads();
    }
    const int nrows = B * H;
    // every access of the launch is 16-byte aligned, or none is promised to be (channel slices of the concat buffer)
    const bool z_aligned = V == 4 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (cs_z & 3) == 0;
    const bool g_aligned = V == 4 && BWD && (reinterpret_cast<uintptr_t>(gy + c_off) & 15) == 0 && (cs_gy & 3) == 0;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int b = row / H, y = row - b * H + 1;
        const size_t base = ((size_t)(b * R + y) * P + 1);  // first interior position of the row
        for (int x = pl; active && x < W; x += ppi) {
            float zz[V];
            VecIO<V>::load_nt(z + (base + x) * cs_z + V * cg, zz, z_aligned);
            if (!BWD) {
#pragma unroll
                for (int e = 0; e < V; ++e) { s0[e] += zz[e]; s1[e] += (double)zz[e] * zz[e]; }
            } else {
                float gg[V];
                VecIO<V>::load_nt(gy + (base + x) * cs_gy + c_off + V * cg, gg, g_aligned);
                // the coefficient reads stay inside the loop and come a few at a time (the barriers): registers, see above
#pragma unroll
                for (int h = 0; h < V; h += 2) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = h; e < h + 2 && e < V; ++e) {
                        const float sc = cf[0 * 256 * V + V * cg + e], sh = cf[1 * 256 * V + V * cg + e];
                        const float u = fmaf(zz[e], sc, sh);
                        gg[e] = u > 0.f ? gg[e] : 0.f;
                    }
                }
#pragma unroll
                for (int h = 0; h < V; h += 2) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = h; e < h + 2 && e < V; ++e) {
                        const float mu = cf[2 * 256 * V + V * cg + e], iv = cf[3 * 256 * V + V * cg + e];
                        s0[e] += gg[e];
                        s1[e] += (double)gg[e] * (double)((zz[e] - mu) * iv);
                    }
                }
            }
        }
    }
    __syncthreads();                                                // every wave is done with the coefficients
#pragma unroll
    for (int e = 0; e < V; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
    __syncthreads();
    if (tid < cvn) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            double a0 = 0, a1 = 0;
            for (int p = 0; p < ppi; ++p) { a0 += red[0][p * cvn + tid][e]; a1 += red[1][p * cvn + tid][e]; }
            const int c = V * tid + e;
            if (c < C) {
                partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a0;
                partial[((size_t)blockIdx.x * 2 + 1) * C + c] = a1;
            }
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float *__restrict__ z, int cs_z, const float *__restrict__ gy, int cs_gy,
                                                        int c_off, const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd, int C,
                                                        double *__restrict__ partial, int B, int H, int W)
{
    bn_reduce_body<false, V>(z, cs_z, gy, cs_gy, c_off, scale, shift, mean, invstd, C, partial, B, H, W);
}

// The backward sums, held to 48 registers (the attribute is not taken from a template, hence the plain kernel): that is what
// a SIMD has left beside the two waves of a wide weight-gradient workgroup (2 x 232 of 512).
__global__ __launch_bounds__(256)
void bn_reduce_bwd_kernel(const float *__restrict__ z, int cs_z, const float *__restrict__ gy, int cs_gy, int c_off,
                          const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
                          const float *__restrict__ invstd, int C, double *__restrict__ partial, int B, int H, int W)
{
    bn_reduce_body<true, 4>(z, cs_z, gy, cs_gy, c_off, scale, shift, mean, invstd, C, partial, B, H, W);
}

// one wave per channel: lanes stride over the per-block partials, then a shuffle reduction
__device__ __forceinline__ void reduce_partials(const double *__restrict__ partial, int nblocks, int C, int c,
                                                double &s0, double &s1)
{
    double a = 0, b = 0;
    for (int blk = threadIdx.x; blk < nblocks; blk += 64) {
        a += partial[((size_t)blk * 2 + 0) * C + c];
        b += partial[((size_t)blk * 2 + 1) * C + c];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    s0 = a; s1 = b;
}

__global__ __launch_bounds__(64) void bn_stats_finalize_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                               double n, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, float *running_mean,
                                                               float *running_var, double momentum, double eps,
                                                               float *save_mean, float *save_invstd, float *scale,
                                                               float *shift)
{
    const int c = blockIdx.x;
    double s, ss;
    reduce_partials(partial, nblocks, C, c, s, ss);
    if (threadIdx.x != 0) return;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + eps);
    const float meanf = (float)mean, invf = (float)invstd;
    save_mean[c] = meanf;
    save_invstd[c] = invf;
    if (running_mean) {
        const double unb = n > 1 ? var * n / (n - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    const float sc = (gamma ? gamma[c] : 1.f) * invf;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - meanf * sc;
}

__global__ void bn_coeffs_eval_kernel(const float *gamma, const float *beta, const float *rm, const float *rv,
                                      double eps, float *scale, float *shift, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invf = (float)(1.0 / sqrt((double)rv[c] + eps));
    const float sc = (gamma ? gamma[c] : 1.f) * invf;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                             double n, const float *__restrict__ gamma,
                                                             const float *__restrict__ invstd, float *dgamma,
                                                             float *dbeta, int accumulate, float *coef)
{
    const int c = blockIdx.x;
    double sg, sgx;
    reduce_partials(partial, nblocks, C, c, sg, sgx);
    if (threadIdx.x != 0) return;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
    const double k1 = (double)(gamma ? gamma[c] : 1.f) * invstd[c];
    coef[c] = (float)k1;
    coef[C + c] = (float)(k1 * sg / n);
    coef[2 * C + c] = (float)(k1 * (double)invstd[c] * sgx / n);
}

// eval-mode BatchNorm folded into the preceding convolution: w'[co] = w[co]*scale[co],
// b' = b*scale + shift  (then conv + ReLU is the whole block tail; no BN pass over the activations)
__global__ void fold_bn_kernel(const float *__restrict__ w, const float *__restrict__ b,
                               const float *__restrict__ scale, const float *__restrict__ shift,
                               float *__restrict__ w_out, float *__restrict__ b_out, int Cout, int per_co)
{
    const long long total = (long long)Cout * per_co;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i / per_co);
        w_out[i] = w[i] * scale[co];
        if (i % per_co == 0) b_out[co] = fmaf(b ? b[co] : 0.f, scale[co], shift[co]);
    }
}

// ---------------------------------------------------------------------------------------------
// row-wise elementwise kernels: one block per grid row (b, y), y in [0, R)
// MODE 0: y = relu(z*scale+shift) (interior) | MODE 1: dz = k1*g - k2 - k3*(z-mean), g = gy*(u>0)
// ---------------------------------------------------------------------------------------------
template <int MODE, int V>
__global__ __launch_bounds__(256) void bn_rows_kernel(const float *__restrict__ z, int cs_z,
                                                      const float *__restrict__ gy, int cs_gy, int c_off_gy,
                                                      const float *__restrict__ scale,
                                                      const float *__restrict__ shift,
                                                      const float *__restrict__ mean,
                                                      const float *__restrict__ coef, int C,
                                                      float *__restrict__ out, int cs_out, int c_off_out,
                                                      int C_store, int H, int W, float *__restrict__ amax, int nrows)
{
    // a block takes grid rows blockIdx.x, + gridDim.x, ... (launched with one block per row); a thread walks
    // (position, channel group) pairs with stride 256 without divisions: (x, cg) += (256 / cvn, 256 % cvn) with carry.
    // Per-channel coefficients sit in LDS.  (Quarter rows per block -- 8 % faster stand-alone at 64 patches, where whole
    // rows leave a tail of blocks on an empty chip -- cost 4 ms per 60 ms step inside it: profiles/r04_ab64_bn_row_segments.log.)
    extern __shared__ float coefs[];           // [5][Cpad]: scale, shift, (mean, k1, k2, k3 for MODE 1)
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int cvn = (C_store + V - 1) / V;
    const int Cpad = cvn * V;
    for (int c = threadIdx.x; c < Cpad; c += blockDim.x) {
        const bool ok = c < C;
        coefs[c] = ok ? scale[c] : 0.f;
        coefs[Cpad + c] = ok ? shift[c] : 0.f;
        if (MODE == 1) {
            coefs[2 * Cpad + c] = ok ? mean[c] : 0.f;
            coefs[3 * Cpad + c] = ok ? coef[c] : 0.f;
            coefs[4 * Cpad + c] = ok ? coef[C + c] : 0.f;
            coefs[5 * Cpad + c] = ok ? coef[2 * C + c] : 0.f;
        }
    }
    __syncthreads();
    const int dx = 256 / cvn, dc = 256 - dx * cvn;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int y = row % R;
    const size_t base = (size_t)row * P;
    const bool row_in = (y >= 1 && y <= H);
    int x = threadIdx.x / cvn, cg = threadIdx.x - x * cvn;
    float mx = 0.f;
    for (; x < P; x += dx, cg += dc) {
        if (cg >= cvn) { cg -= cvn; ++x; if (x >= P) break; }
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = 0.f;
        if (row_in && x >= 1 && x <= W && V * cg < C) {
            float zz[V], gg[V];
            VecIO<V>::load_nt(z + (base + x) * cs_z + V * cg, zz);
            if (MODE == 1) VecIO<V>::load_nt(gy + (base + x) * cs_gy + c_off_gy + V * cg, gg);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int c = V * cg + k;      // channels >= C have zero coefficients -> output 0
                const float u = fmaf(zz[k], coefs[c], coefs[Cpad + c]);
                if (MODE == 0) {
                    o[k] = fmaxf(u, 0.f);
                } else {
                    const float g = u > 0.f ? gg[k] : 0.f;
                    o[k] = coefs[3 * Cpad + c] * g - coefs[4 * Cpad + c] - coefs[5 * Cpad + c] * (zz[k] - coefs[2 * Cpad + c]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) mx = fmaxf(mx, fabsf(o[k]));
        float *op = out + (base + x) * cs_out + c_off_out + V * cg;
        if (V * cg + V - 1 < C_store) {
            VecIO<V>::store_nt(op, o);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (V * cg + k < C_store) op[k] = o[k];
        }
    }
    if (amax) mmlf_amax_update_row(mx, amax, row);      // this workgroup wrote grid row `row`
    }
}

// BatchNorm apply + ReLU of the four stream nets' last blocks in ONE pass over the concat buffer (torch.cat,
// reference feed_forward.py:266-267): four launches that each write a 280-byte slice of every 1120-byte position row
// run at half the rate of a dense pass (partial lines: 2.5 TB/s against 5.1); this one writes whole rows.
// Channel PAIRS across the threads (a 70-channel slice starts on an 8-byte boundary only).
struct BnApply4 { const float *z[4]; const float *scale[4]; const float *shift[4]; };
__global__ __launch_bounds__(256) void bn_apply4_kernel(BnApply4 s, int cs_z, int C, float *__restrict__ out,
                                                        int H, int W, float *__restrict__ amax, int nrows)
{
    extern __shared__ float coefs[];           // [2][4 * C]: scale, shift in concat order
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int C4 = 4 * C, half = C / 2, cvn = 2 * C;     // channel pairs per position row
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
        const int k = c / C, cl = c - k * C;
        const float *sc = k == 0 ? s.scale[0] : k == 1 ? s.scale[1] : k == 2 ? s.scale[2] : s.scale[3];
        const float *sh = k == 0 ? s.shift[0] : k == 1 ? s.shift[1] : k == 2 ? s.shift[2] : s.shift[3];
        coefs[c] = sc[cl];
        coefs[C4 + c] = sh[cl];
    }
    __syncthreads();
    const int dx = 256 / cvn, dc = 256 - dx * cvn;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int y = row % R;
        const size_t base = (size_t)row * P;
        const bool row_in = (y >= 1 && y <= H);
        int x = threadIdx.x / cvn, g = threadIdx.x - x * cvn;
        float mx = 0.f;
        for (; x < P; x += dx, g += dc) {
            if (g >= cvn) { g -= cvn; ++x; if (x >= P) break; }
            float2 o = make_float2(0.f, 0.f);
            if (row_in && x >= 1 && x <= W) {
                const int k = g / half, cl = 2 * (g - k * half);
                const float *zp = k == 0 ? s.z[0] : k == 1 ? s.z[1] : k == 2 ? s.z[2] : s.z[3];
                const float2 zz = *reinterpret_cast<const float2 *>(zp + (base + x) * cs_z + cl);
                o.x = fmaxf(fmaf(zz.x, coefs[2 * g], coefs[C4 + 2 * g]), 0.f);
                o.y = fmaxf(fmaf(zz.y, coefs[2 * g + 1], coefs[C4 + 2 * g + 1]), 0.f);
            }
            mx = fmaxf(mx, fmaxf(o.x, o.y));
            *reinterpret_cast<float2 *>(out + (base + x) * C4 + 2 * g) = o;
        }
        if (amax) mmlf_amax_update_row(mx, amax, row);
    }
}

// (Round 4 measured a second form of these two passes -- a fixed channel group per thread with its coefficients in
// registers and four positions' loads in flight, every combination of non-temporal / plain accesses, persistent row grids
// -- and persistent row grids for the form above: the same bytes per second stand-alone, 1-3 % slower inside the step;
// DESIGN.md section 4.8, profiles/r04_bn_bench_*.log, r04_ab_bench_bn*.log; the code is in the history: commit 5c543cc.)

// NCHW <-> grid
#define PACK_XT 128   // positions per transpose tile of pack_nchw_kernel (halved until the tile fits 32 KB: wide tensors;
                      // at 64 KB two workgroups per CU packed the 108-channel DPP gradient 0.5 ms slower)
__global__ __launch_bounds__(256) void pack_nchw_kernel(const float *__restrict__ src, int C,
                                                        float *__restrict__ grid, int cs, int H, int W,
                                                        float *__restrict__ amax, int xt)
{
    // a transpose through LDS: the NCHW planes are read with x across the lanes, the grid row is written with the channel
    // groups across the lanes (whole 16*c4n-byte position rows per instruction instead of one 16-byte piece per line)
    extern __shared__ float tile[];            // [cs][xt | 1]: channel-major, odd pitch (conflict-free both ways)
    float mx = 0.f;
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int row = blockIdx.x;
    const int b = row / R, y = row - b * R;
    const size_t base = (size_t)row * P;
    const int c4n = cs / 4;
    const int pitch = xt | 1;
    const bool row_in = (y >= 1 && y <= H);
    for (int x0 = 0; x0 < P; x0 += xt) {                       // the row in pieces of xt positions
        const int nx = min(xt, P - x0);
        for (int e = threadIdx.x; e < cs * nx; e += blockDim.x) {
            const int c = e / nx, x = x0 + e - c * nx;         // x fastest: coalesced NCHW reads
            float v = 0.f;
            if (row_in && x >= 1 && x <= W && c < C) v = src[(((size_t)b * C + c) * H + (y - 1)) * W + (x - 1)];
            tile[c * pitch + (x - x0)] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nx * c4n; e += blockDim.x) {
            const int xl = e / c4n, cg = e - xl * c4n;         // channel group fastest: coalesced grid writes
            const float4 o = make_float4(tile[(4 * cg) * pitch + xl], tile[(4 * cg + 1) * pitch + xl],
                                         tile[(4 * cg + 2) * pitch + xl], tile[(4 * cg + 3) * pitch + xl]);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            *reinterpret_cast<float4 *>(grid + (base + x0 + xl) * cs + 4 * cg) = o;
        }
        __syncthreads();
    }
    if (amax) mmlf_amax_update_row(mx, amax, row);      // this workgroup wrote grid row `row`
}

__global__ __launch_bounds__(256) void unpack_nchw_kernel(const float *__restrict__ grid, int cs,
                                                          float *__restrict__ dst, int C, int H, int W, int xt)
{
    // the inverse transpose, through LDS as well: the grid row is read with the channels across the lanes (whole position
    // rows), the NCHW planes are written with x across the lanes.  (One thread per (c, x) straight from the grid read
    // with a stride of cs floats between lanes: 1.7 TB/s on the 108-channel DPP scores; 2.3 -> 0.9 ms per launch.)
    extern __shared__ float tile[];            // [xt][C | 1]: position-major, odd pitch (conflict-free both ways)
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int row = blockIdx.x;  // b*H + (y-1)
    const int b = row / H, y = row - b * H + 1;
    const size_t base = ((size_t)(b * R + y)) * P;
    const int pitch = C | 1;
    for (int x0 = 0; x0 < W; x0 += xt) {
        const int nx = min(xt, W - x0);
        for (int e = threadIdx.x; e < nx * C; e += blockDim.x) {
            const int xl = e / C, c = e - xl * C;
            tile[xl * pitch + c] = grid[(base + x0 + xl + 1) * cs + c];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < C * nx; e += blockDim.x) {
            const int c = e / nx, xl = e - c * nx;
            dst[(((size_t)b * C + c) * H + (y - 1)) * W + x0 + xl] = tile[xl * pitch + c];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// heads
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float laplace_pdf(float x, float mu, float b)
{
    // 1.0 / (2.0 * b) * exp(-|x - mu| / b), reference feed_forward.py:9-12 (float32 op order)
    const float a = __fdiv_rn(1.0f, __fmul_rn(2.0f, b));
    const float t = __fdiv_rn(-fabsf(__fsub_rn(x, mu)), b);
    return __fmul_rn(a, expf(t));
}

__global__ void head_upr_kernel(const float *__restrict__ out, const float *__restrict__ grid,
                                float *__restrict__ post, int steps, int HW, long long total)
{
    // out: (B,2,H,W); post: (B,steps,H,W)
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const float mu = out[(b * 2 + 0) * HW + p];
        const float bb = expf(out[(b * 2 + 1) * HW + p]);
        for (int k = 0; k < steps; ++k) post[(b * steps + k) * HW + p] = laplace_pdf(grid[k], mu, bb);
    }
}

__global__ void head_dpp_kernel(const float *__restrict__ sc, const float *__restrict__ gt, const float *__restrict__ gn,
                                float *__restrict__ one_hot, float *__restrict__ post, float *__restrict__ mean,
                                float *__restrict__ logvar, int steps, int HW, long long total)
{
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const float *s = sc + b * steps * HW + p;
        float mx = -INFINITY, z = 0.f;
        for (int k = 0; k < steps; ++k) {
            const float v = s[(size_t)k * HW];
            mx = fmaxf(mx, v);
            z = __fadd_rn(z, expf(v));
        }
        float m = 0.f;
        for (int k = 0; k < steps; ++k) {
            const float v = s[(size_t)k * HW];
            const float oh = (v == mx) ? 1.f : 0.f;
            one_hot[(b * steps + k) * HW + p] = oh;
            m = __fadd_rn(m, __fmul_rn(gt[k], oh));
        }
        float lv = 0.f;
        for (int k = 0; k < steps; ++k) {
            const float v = s[(size_t)k * HW];
            const float pk = __fdiv_rn(expf(v), z);
            post[(b * steps + k) * HW + p] = pk;
            const float d = __fsub_rn(gn[k], m);
            lv = __fadd_rn(lv, __fmul_rn(__fmul_rn(d, d), pk));
        }
        mean[idx] = m;
        logvar[idx] = logf(lv);
    }
}

// Backward of the two heads (round 5): the reference builds `posterior` (UPR, DPP) and the DPP `logvar` as differentiable
// functions of the network output (feed_forward.py:276-302); no reference loss uses them (loss.py:70,146,264), but a
// drop-in module must not silently cut a graph the reference has.
//  UPR: post_k = exp(-|g_k - mu| / b) / (2 b), b = exp(lv):  d post_k / d mu = post_k sign(g_k - mu) / b,
//       d post_k / d lv = post_k (|g_k - mu| / b - 1).
//  DPP: p = exp(s) / sum exp(s), lv = log V, V = sum_k (g_k - m)^2 p_k with m (from the one-hot arg-max) a constant:
//       dL/dp_k = go_post_k + go_lv (g_k - m)^2 / V,  dL/ds_i = p_i (dL/dp_i - sum_j dL/dp_j p_j).
__global__ void head_upr_bwd_kernel(const float *__restrict__ out, const float *__restrict__ grid,
                                    const float *__restrict__ gpost, float *__restrict__ gout, int steps, int HW, long long total)
{
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const float mu = out[(b * 2 + 0) * HW + p];
        const float bb = expf(out[(b * 2 + 1) * HW + p]);
        float gm = 0.f, gl = 0.f;
        for (int k = 0; k < steps; ++k) {
            const float go = gpost[(b * steps + k) * HW + p];
            const float d = grid[k] - mu, pk = laplace_pdf(grid[k], mu, bb);
            gm = fmaf(go * pk, (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / bb, gm);     // torch: d|x|/dx = sign(x), 0 at 0
            gl = fmaf(go * pk, fabsf(d) / bb - 1.f, gl);
        }
        gout[(b * 2 + 0) * HW + p] = gm;
        gout[(b * 2 + 1) * HW + p] = gl;
    }
}

__global__ void head_dpp_bwd_kernel(const float *__restrict__ sc, const float *__restrict__ gn, const float *__restrict__ mean,
                                    const float *__restrict__ gpost, const float *__restrict__ glv,
                                    float *__restrict__ gsc, int steps, int HW, long long total)
{
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const float *s = sc + b * steps * HW + p;
        const float m = mean[idx];
        float z = 0.f;
        for (int k = 0; k < steps; ++k) z += expf(s[(size_t)k * HW]);
        float V = 0.f;
        for (int k = 0; k < steps; ++k) {
            const float d = gn[k] - m;
            V = fmaf(d * d, expf(s[(size_t)k * HW]) / z, V);
        }
        const float gl = glv ? glv[idx] / V : 0.f;
        float dot = 0.f;                       // sum_j dL/dp_j p_j
        for (int k = 0; k < steps; ++k) {
            const float d = gn[k] - m, pk = expf(s[(size_t)k * HW]) / z;
            const float dp = (gpost ? gpost[(b * steps + k) * HW + p] : 0.f) + gl * d * d;
            dot = fmaf(dp, pk, dot);
        }
        for (int k = 0; k < steps; ++k) {
            const float d = gn[k] - m, pk = expf(s[(size_t)k * HW]) / z;
            const float dp = (gpost ? gpost[(b * steps + k) * HW + p] : 0.f) + gl * d * d;
            gsc[(b * steps + k) * HW + p] = pk * (dp - dot);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// losses: pass 1 partial (count, sum), finalize, pass 2 gradient
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgnf(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__device__ __forceinline__ float ce_pixel(const float *s, int steps, int HW, float gtv, const float *grid,
                                          float half_step, float *z_out)
{
    float dot = 0.f, z = 0.f;
    for (int k = 0; k < steps; ++k) {
        const float v = fmaxf(s[(size_t)k * HW], 0.f);
        const float t = fabsf(__fsub_rn(grid[k], gtv)) < half_step ? 1.f : 0.f;
        dot = __fadd_rn(dot, __fmul_rn(v, t));
        z = __fadd_rn(z, expf(v));
    }
    *z_out = z;
    return -logf(__fdiv_rn(expf(dot), z));
}

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float *__restrict__ out, int oc,
                                                           const float *__restrict__ gt,
                                                           const int32_t *__restrict__ mask,
                                                           const float *__restrict__ grid, float half_step,
                                                           double *__restrict__ scratch, int HW, long long total)
{
    double cnt = 0, sum = 0;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const int mk = mask[idx];
        cnt += mk;
        float l;
        if (kind == 0) {
            l = fabsf(out[(b * oc) * HW + p] - gt[idx]);
        } else if (kind == 1) {
            const float lv = out[(b * oc + 1) * HW + p];
            l = __fadd_rn(__fmul_rn(expf(-lv), fabsf(out[(b * oc) * HW + p] - gt[idx])), lv);
        } else {
            float z;
            l = ce_pixel(out + b * oc * HW + p, oc, HW, gt[idx], grid, half_step, &z);
        }
        sum += (double)(l * (float)mk);
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = cnt; r1[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { scratch[2 + 2 * blockIdx.x] = r0[0]; scratch[3 + 2 * blockIdx.x] = r1[0]; }
}

__global__ void loss_finalize_kernel(double *scratch, int nblocks, float *loss_out, const double *den_override)
{
    // one wave, fixed order: lane l sums blocks l, l + 64, ..., then a butterfly (a single thread walking 1024 dependent
    // loads took 88 us)
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double cnt = 0, sum = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) { cnt += scratch[2 + 2 * b]; sum += scratch[3 + 2 * b]; }
    for (int off = 32; off; off >>= 1) { cnt += __shfl_xor(cnt, off); sum += __shfl_xor(sum, off); }
    if (threadIdx.x != 0) return;
    if (den_override) cnt = *den_override;
    const double den = cnt == 0 ? 1.0 : cnt;
    scratch[0] = 1.0 / den;
    scratch[1] = cnt;
    *loss_out = (float)(sum / den);
}

__global__ __launch_bounds__(256) void loss_grad_kernel(int kind, const float *__restrict__ out, int oc,
                                                        const float *__restrict__ gt,
                                                        const int32_t *__restrict__ mask,
                                                        const float *__restrict__ grid, float half_step,
                                                        const double *__restrict__ scratch,
                                                        float *__restrict__ grad, int HW, long long total)
{
    const float inv = (float)scratch[0];
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        const float mk = (float)mask[idx] * inv;
        if (kind == 0) {
            grad[(b * oc) * HW + p] = sgnf(out[(b * oc) * HW + p] - gt[idx]) * mk;
        } else if (kind == 1) {
            const float d = out[(b * oc) * HW + p] - gt[idx];
            const float e = expf(-out[(b * oc + 1) * HW + p]);
            grad[(b * oc) * HW + p] = e * sgnf(d) * mk;
            grad[(b * oc + 1) * HW + p] = (1.f - e * fabsf(d)) * mk;
        } else {
            const float *s = out + b * oc * HW + p;
            float z;
            ce_pixel(s, oc, HW, gt[idx], grid, half_step, &z);
            for (int k = 0; k < oc; ++k) {
                const float raw = s[(size_t)k * HW];
                const float t = fabsf(__fsub_rn(grid[k], gt[idx])) < half_step ? 1.f : 0.f;
                const float g = raw > 0.f ? (expf(raw) / z - t) * mk : 0.f;
                grad[(b * oc + k) * HW + p] = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// multimodal / padded training losses (reference loss.py:80-103, 336-372, 264-294 with mask_padding;
// dl.py:134-157 mpi_to_weights feeding loss.py:146-160).  target = mpi (B,P,5,H,W): [:, :, 3] alpha,
// [:, :, 4] disparity -- or gt (B,H,W) for the padded single-mode loss.  Two of them need whole-batch sums
// first (aux[0], aux[1]): kind 4 the sum of the total alpha and the number of pixels without a surface,
// kind 6 the number of in-range pixels.
// ---------------------------------------------------------------------------------------------
enum { LOSS_MULTI_L1 = 3, LOSS_MULTI_UPR = 4, LOSS_MULTI_CE = 5, LOSS_UPR_PADDED = 6 };

struct MultiLossArgs {
    const float *out;            // (B, oc, H, W)
    const float *target;         // mpi (B, P, 5, H, W) | gt (B, H, W)
    const int32_t *mask, *mask_padding;
    const float *grid;           // kind 5: torch.linspace bin centres
    double *scratch;             // [0] 1/den [1] count [2 + 2b] block partials ... [2 + 2*nblocks + {0,1}] aux sums
    float *grad;
    const double *den_override, *aux_override;
    float half_step;
    int kind, oc, P, HW, nblocks;
    long long total;             // B*H*W
};

__device__ __forceinline__ void block_sum2(double &a, double &b)
{
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = a; r1[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
        __syncthreads();
    }
    a = r0[0]; b = r1[0];
}

// pass 0: whole-batch sums the per-pixel loss depends on (kinds 4 and 6), per-block partials
__global__ __launch_bounds__(256) void loss_multi_aux_kernel(MultiLossArgs a)
{
    double s0 = 0, s1 = 0;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < a.total; idx += 256ll * gridDim.x) {
        if (a.kind == LOSS_MULTI_UPR) {
            const long long b = idx / a.HW;
            const int p = (int)(idx - b * a.HW);
            float tot = 0.f;
            for (int k = 0; k < a.P; ++k) tot = __fadd_rn(tot, a.target[((b * a.P + k) * 5 + 3) * a.HW + p]);
            s0 += tot;
            s1 += tot < 0.01f ? 1.0 : 0.0;
        } else {
            s0 += a.mask_padding[idx];
        }
    }
    block_sum2(s0, s1);
    if (threadIdx.x == 0) {
        a.scratch[2 + 2 * a.nblocks + 2 + 2 * blockIdx.x] = s0;
        a.scratch[2 + 2 * a.nblocks + 3 + 2 * blockIdx.x] = s1;
    }
}
__global__ void loss_multi_aux_finalize_kernel(MultiLossArgs a)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;        // one wave, fixed order (as loss_finalize_kernel)
    double s0 = 0, s1 = 0;
    if (a.aux_override) { s0 = a.aux_override[0]; s1 = a.aux_override[1]; }
    else {
        for (int b = threadIdx.x; b < a.nblocks; b += 64) {
            s0 += a.scratch[2 + 2 * a.nblocks + 2 + 2 * b];
            s1 += a.scratch[2 + 2 * a.nblocks + 3 + 2 * b];
        }
        for (int off = 32; off; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    }
    if (threadIdx.x != 0) return;
    a.scratch[2 + 2 * a.nblocks] = s0;
    a.scratch[2 + 2 * a.nblocks + 1] = s1;
}

// per-pixel loss and (grad != null) its derivative w.r.t. mean / logvar; CE handled apart
struct PixelLoss { float l, dm, dlv; };
__device__ __forceinline__ PixelLoss multi_pixel(const MultiLossArgs &a, long long idx, float f0, float f1)
{
    const long long b = idx / a.HW;
    const int p = (int)(idx - b * a.HW);
    const float m = a.out[(b * a.oc) * a.HW + p];
    PixelLoss r = {0.f, 0.f, 0.f};
    if (a.kind == LOSS_MULTI_L1) {
        for (int k = 0; k < a.P; ++k) {
            const float w = a.target[((b * a.P + k) * 5 + 3) * a.HW + p], t = a.target[((b * a.P + k) * 5 + 4) * a.HW + p];
            const float d = __fsub_rn(m, t);
            r.l = __fadd_rn(r.l, __fmul_rn(fabsf(d), w));
            r.dm = __fadd_rn(r.dm, __fmul_rn(sgnf(d), w));
        }
    } else if (a.kind == LOSS_MULTI_UPR) {         // f0 = mean total alpha, f1 = n / #(pixels without a surface)
        const float lv = a.out[(b * a.oc + 1) * a.HW + p], e = expf(-lv);
        float tot = 0.f, acc = 0.f, gm = 0.f, glv = 0.f;
        for (int k = 0; k < a.P; ++k) {
            const float w = a.target[((b * a.P + k) * 5 + 3) * a.HW + p], t = a.target[((b * a.P + k) * 5 + 4) * a.HW + p];
            const float d = __fsub_rn(m, t);
            acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(__fmul_rn(e, fabsf(d)), lv), w));
            gm = __fadd_rn(gm, __fmul_rn(__fmul_rn(e, sgnf(d)), w));
            glv = __fadd_rn(glv, __fmul_rn(__fsub_rn(1.f, __fmul_rn(e, fabsf(d))), w));
            tot = __fadd_rn(tot, w);
        }
        const float oor = tot < 0.01f ? 1.f : 0.f;
        const float l_oor = __fmul_rn(__fmul_rn(-lv, oor), f1);        // 0 * inf = NaN when no pixel lacks a surface, as in the reference
        r.l = __fdiv_rn(__fadd_rn(__fdiv_rn(acc, f0), l_oor), 2.f);
        r.dm = __fdiv_rn(__fdiv_rn(gm, f0), 2.f);
        r.dlv = __fdiv_rn(__fsub_rn(__fdiv_rn(glv, f0), __fmul_rn(oor, f1)), 2.f);
    } else {                                        // LOSS_UPR_PADDED: f0 = n / #in-range, f1 = n / #out-of-range (1 if none)
        const float lv = a.out[(b * a.oc + 1) * a.HW + p], e = expf(-lv);
        const float d = __fsub_rn(m, a.target[idx]);
        const float mp = (float)a.mask_padding[idx], oor = 1.f - mp;
        const float l_in = __fmul_rn(__fmul_rn(__fadd_rn(__fmul_rn(e, fabsf(d)), lv), mp), f0);
        const float l_oor = __fmul_rn(__fmul_rn(-lv, oor), f1);
        r.l = __fdiv_rn(__fadd_rn(l_in, l_oor), 2.f);
        r.dm = __fdiv_rn(__fmul_rn(__fmul_rn(__fmul_rn(e, sgnf(d)), mp), f0), 2.f);
        r.dlv = __fdiv_rn(__fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, __fmul_rn(e, fabsf(d))), mp), f0), __fmul_rn(oor, f1)), 2.f);
    }
    return r;
}
__device__ __forceinline__ void multi_factors(const MultiLossArgs &a, float &f0, float &f1)
{
    const double s0 = a.scratch[2 + 2 * a.nblocks], s1 = a.scratch[2 + 2 * a.nblocks + 1], n = (double)a.total;
    f0 = f1 = 1.f;
    if (a.kind == LOSS_MULTI_UPR) { f0 = (float)(s0 / n); f1 = (float)n / (float)s1; }
    else if (a.kind == LOSS_UPR_PADDED) {
        if (s0 > 0) f0 = (float)n / (float)s0;
        if (n - s0 > 0) f1 = (float)n / (float)(n - s0);
    }
}
// MaskedCrossEntropy on the weighted more-hot target mpi_to_weights(mpi): t_k = sum_p [|grid_k - d_p| < step/2] * alpha_p
__device__ __forceinline__ float multi_ce_target(const MultiLossArgs &a, long long b, int p, int k)
{
    float t = 0.f;
    for (int q = 0; q < a.P; ++q) {
        const float w = a.target[((b * a.P + q) * 5 + 3) * a.HW + p], d = a.target[((b * a.P + q) * 5 + 4) * a.HW + p];
        t = __fadd_rn(t, fabsf(__fsub_rn(a.grid[k], d)) < a.half_step ? w : 0.f);
    }
    return t;
}
__device__ __forceinline__ float multi_ce_pixel(const MultiLossArgs &a, long long b, int p, float *z_out)
{
    const float *s = a.out + b * a.oc * a.HW + p;
    float dot = 0.f, z = 0.f;
    for (int k = 0; k < a.oc; ++k) {
        const float v = fmaxf(s[(size_t)k * a.HW], 0.f);
        dot = __fadd_rn(dot, __fmul_rn(v, multi_ce_target(a, b, p, k)));
        z = __fadd_rn(z, expf(v));
    }
    *z_out = z;
    return -logf(__fdiv_rn(expf(dot), z));
}

__global__ __launch_bounds__(256) void loss_multi_partial_kernel(MultiLossArgs a)
{
    float f0, f1;
    multi_factors(a, f0, f1);
    double cnt = 0, sum = 0;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < a.total; idx += 256ll * gridDim.x) {
        const int mk = a.mask[idx];
        cnt += mk;
        float l;
        if (a.kind == LOSS_MULTI_CE) {
            float z;
            const long long b = idx / a.HW;
            l = multi_ce_pixel(a, b, (int)(idx - b * a.HW), &z);
        } else {
            l = multi_pixel(a, idx, f0, f1).l;
        }
        sum += (double)(l * (float)mk);
    }
    block_sum2(cnt, sum);
    if (threadIdx.x == 0) { a.scratch[2 + 2 * blockIdx.x] = cnt; a.scratch[3 + 2 * blockIdx.x] = sum; }
}

__global__ __launch_bounds__(256) void loss_multi_grad_kernel(MultiLossArgs a)
{
    float f0, f1;
    multi_factors(a, f0, f1);
    const float inv = (float)a.scratch[0];
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < a.total; idx += 256ll * gridDim.x) {
        const long long b = idx / a.HW;
        const int p = (int)(idx - b * a.HW);
        const float mk = (float)a.mask[idx] * inv;
        if (a.kind == LOSS_MULTI_CE) {
            float z;
            multi_ce_pixel(a, b, p, &z);
            const float *s = a.out + b * a.oc * a.HW + p;
            for (int k = 0; k < a.oc; ++k) {
                const float raw = s[(size_t)k * a.HW];
                a.grad[(b * a.oc + k) * a.HW + p] = raw > 0.f ? (expf(raw) / z - multi_ce_target(a, b, p, k)) * mk : 0.f;
            }
        } else {
            const PixelLoss r = multi_pixel(a, idx, f0, f1);
            a.grad[(b * a.oc) * a.HW + p] = r.dm * mk;
            if (a.kind != LOSS_MULTI_L1) a.grad[(b * a.oc + 1) * a.HW + p] = r.dlv * mk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults: no weight decay, no amsgrad)
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, long long n, float step_size, float beta1, float beta2,
                            float eps, float bc2_sqrt, float grad_scale)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + gi * gi * (1.f - beta2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// ---------------------------------------------------------------------------------------------
// Shift (hci4d.py:907-990) for S shift values at once; in: (views,3,H,W) -> out: (S,views,3,H,W)
// ---------------------------------------------------------------------------------------------
struct ShiftTab { int s0, s1; float w0, w1; };

__device__ __forceinline__ int wrapi(int a, int n) { a %= n; return a < 0 ? a + n : a; }
// source index of a roll by +s with Python-slice clamping (|s| >= n -> identity)
__device__ __forceinline__ int roll_src(int x, int s, int n) { return (s >= n || -s >= n) ? x : wrapi(x - s, n); }

__global__ __launch_bounds__(256) void shift_kernel(const float *__restrict__ h, const float *__restrict__ v,
                                                    const float *__restrict__ i_, const float *__restrict__ d,
                                                    float *__restrict__ oh, float *__restrict__ ov,
                                                    float *__restrict__ oi, float *__restrict__ od,
                                                    const int32_t *__restrict__ tab_s,
                                                    const float *__restrict__ tab_w, int views, int H, int W)
{
    // grid: (S*views*3*H) rows, threads over x
    const int row = blockIdx.x;
    const int y = row % H;
    int r = row / H;
    const int ch = r % 3; r /= 3;
    const int view = r % views;
    const int s = r / views;
    ShiftTab t;
    t.s0 = tab_s[2 * (s * views + view)]; t.s1 = tab_s[2 * (s * views + view) + 1];
    t.w0 = tab_w[2 * (s * views + view)]; t.w1 = tab_w[2 * (s * views + view) + 1];
    const float *ph = h + ((size_t)(view * 3 + ch) * H) * W;
    const float *pv = v + ((size_t)(view * 3 + ch) * H) * W;
    const float *pi = i_ + ((size_t)(view * 3 + ch) * H) * W;
    const float *pd = d + ((size_t)(view * 3 + ch) * H) * W;
    const size_t ob = ((size_t)((s * views + view) * 3 + ch) * H + y) * W;
    auto lerp = [&](float a, float b) { return __fadd_rn(__fmul_rn(a, t.w0), __fmul_rn(b, t.w1)); };
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int x0 = roll_src(x, t.s0, W), x1 = roll_src(x, t.s1, W);
        // h: along W
        oh[ob + x] = lerp(ph[(size_t)y * W + x0], ph[(size_t)y * W + x1]);
        // v: along H
        const int y0 = roll_src(y, t.s0, H), y1 = roll_src(y, t.s1, H);
        ov[ob + x] = lerp(pv[(size_t)y0 * W + x], pv[(size_t)y1 * W + x]);
        // d: W pass then H pass (+s)
        {
            const float ta = lerp(pd[(size_t)y0 * W + x0], pd[(size_t)y0 * W + x1]);
            const float tb = lerp(pd[(size_t)y1 * W + x0], pd[(size_t)y1 * W + x1]);
            od[ob + x] = lerp(ta, tb);
        }
        // i: W pass then H pass with the NEGATED shift (hci4d.py:971-975)
        {
            const int yi0 = roll_src(y, -t.s0, H), yi1 = roll_src(y, -t.s1, H);
            const float ta = lerp(pi[(size_t)yi0 * W + x0], pi[(size_t)yi0 * W + x1]);
            const float tb = lerp(pi[(size_t)yi1 * W + x0], pi[(size_t)yi1 * W + x1]);
            oi[ob + x] = lerp(ta, tb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shift + pack (round 6): the Ensamble's members (ensamble.py:61-76) go from the four view stacks straight into the padded NHWC
// grid the trunk reads -- mmlf_shift_views followed by mmlf_pack_nchw in one pass per stream, without the (S, views, 3, H, W)
// intermediate (7.9 GB written and read again per 512 x 512 scene).  The arithmetic per sample is shift_kernel's, operation for
// operation (the two launches write the same bits: tests/test_ensamble.py); the transpose through LDS is pack_nchw_kernel's.
// KIND: 0 = horizontal stack (roll along W), 1 = vertical (along H), 2 = increasing diagonal (W, then H with the NEGATED
// shift, hci4d.py:971-975), 3 = decreasing diagonal (W, then H).
// ---------------------------------------------------------------------------------------------
// the same index as roll_src for 0 <= x < n, without an integer division: |s| < n (else the identity), so x - s lies in (-n, 2n)
__device__ __forceinline__ int roll_src_fast(int x, int s, int n)
{
    if (s >= n || -s >= n) return x;
    int a = x - s;
    a += a < 0 ? n : 0;
    a -= a >= n ? n : 0;
    return a;
}
struct ShiftPackTab { int s0, s1; float w0, w1; int ya, yb, pad0, pad1; };     // + the two source rows of this grid row
template <int KIND>
__global__ __launch_bounds__(256) void shift_pack_kernel(const float *__restrict__ src, int views, float *__restrict__ grid,
                                                         int cs, const int32_t *__restrict__ tab_s,
                                                         const float *__restrict__ tab_w, int H, int W,
                                                         float *__restrict__ amax, int xt_log)
{
    // No integer division in the loops (the first form, with pack_nchw_kernel's e / nx indexing and roll_src's two remainders per
    // index, took 1.4-2.2 ms per stream of a 70-member 512 x 512 scene: as long as the two kernels it replaced): a thread keeps
    // ONE position of the piece and walks the channels; a grid row's source rows are found once per view.
    extern __shared__ float tile[];            // [cs][xt | 1] channel-major, then the member's shift table
    const int P = W + MMLF_GRID_PAD_W, R = H + MMLF_GRID_PAD_H;
    const int C = views * 3;
    const int xt = 1 << xt_log;
    const int row = blockIdx.x;                // (member s, grid row y)
    const int s = row / R, y = row - s * R;
    const size_t base = (size_t)row * P;
    const int c4n = cs / 4;
    const int pitch = xt | 1;
    const bool row_in = (y >= 1 && y <= H);
    const int yi = y - 1;
    ShiftPackTab *tab = reinterpret_cast<ShiftPackTab *>(tile + (size_t)cs * pitch);      // [views]
    for (int k = threadIdx.x; k < views; k += blockDim.x) {
        ShiftPackTab t;
        t.s0 = tab_s[2 * (s * views + k)]; t.s1 = tab_s[2 * (s * views + k) + 1];
        t.w0 = tab_w[2 * (s * views + k)]; t.w1 = tab_w[2 * (s * views + k) + 1];
        const int sg = KIND == 2 ? -1 : 1;
        t.ya = (KIND == 0 || !row_in) ? yi : roll_src_fast(yi, sg * t.s0, H);
        t.yb = (KIND == 0 || !row_in) ? yi : roll_src_fast(yi, sg * t.s1, H);
        t.pad0 = t.pad1 = 0;
        tab[k] = t;
    }
    __syncthreads();
    const int tl = threadIdx.x & (xt - 1), c_first = threadIdx.x >> xt_log, c_step = 256 >> xt_log;
    const int dxl = 256 / c4n, dcg = 256 - dxl * c4n;          // the write-out's (position, channel group) stride, once
    const size_t plane = (size_t)H * W;
    float mx = 0.f;
    for (int x0 = 0; x0 < P; x0 += xt) {
        const int nx = min(xt, P - x0);
        const int x = x0 + tl, xi = x - 1;
        const bool in_x = row_in && tl < nx && x >= 1 && x <= W;
        // c = 3 view + ch, stepped without a division (c_step = 256 / xt channels per iteration: 2 at 128-position pieces).  The
        // loop is unrolled so that several channels' source loads (L2 hits, 2-4 per sample, dependent on nothing but the index
        // arithmetic) are in flight at once: one channel at a time the kernel was bound by their latency.
        int view = c_first / 3, ch = c_first - 3 * view;        // (c_first < 256 / xt: a small constant divide, once per piece)
        const int dv = c_step / 3, dc = c_step - 3 * dv;
#pragma unroll 8
        for (int c = c_first; c < cs; c += c_step) {
            float v = 0.f;
            if (in_x && c < C) {
                const ShiftPackTab t = tab[view];
                const float *p = src + (size_t)c * plane;       // plane (view, colour) = channel c
                auto lerp = [&](float a, float b) { return __fadd_rn(__fmul_rn(a, t.w0), __fmul_rn(b, t.w1)); };
                if (KIND == 0) {
                    v = lerp(p[(size_t)yi * W + roll_src_fast(xi, t.s0, W)], p[(size_t)yi * W + roll_src_fast(xi, t.s1, W)]);
                } else if (KIND == 1) {
                    v = lerp(p[(size_t)t.ya * W + xi], p[(size_t)t.yb * W + xi]);
                } else {
                    const int xa = roll_src_fast(xi, t.s0, W), xb = roll_src_fast(xi, t.s1, W);
                    const float ta = lerp(p[(size_t)t.ya * W + xa], p[(size_t)t.ya * W + xb]);
                    const float tb = lerp(p[(size_t)t.yb * W + xa], p[(size_t)t.yb * W + xb]);
                    v = lerp(ta, tb);
                }
            }
            if (tl < nx) tile[c * pitch + tl] = v;
            view += dv; ch += dc;
            if (ch >= 3) { ch -= 3; ++view; }
        }
        __syncthreads();
        int xl = threadIdx.x / c4n, cg = threadIdx.x - xl * c4n;          // channel group fastest: coalesced grid writes
        for (; xl < nx; xl += dxl, cg += dcg) {
            if (cg >= c4n) { cg -= c4n; ++xl; if (xl >= nx) break; }
            const float4 o = make_float4(tile[(4 * cg) * pitch + xl], tile[(4 * cg + 1) * pitch + xl],
                                         tile[(4 * cg + 2) * pitch + xl], tile[(4 * cg + 3) * pitch + xl]);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            *reinterpret_cast<float4 *>(grid + (base + x0 + xl) * cs + 4 * cg) = o;
        }
        __syncthreads();
    }
    if (amax) mmlf_amax_update_row(mx, amax, row);      // this workgroup wrote grid row `row`
}

// ---------------------------------------------------------------------------------------------
// Training patch pipeline: the transform chain of reference train/cli.py:72-91 on scenes cached in HBM,
// fused into one gather per output pixel.  Per sample (host-drawn parameters, reference call order):
// DownSampling f (hci4d.py:483-510) -> Shift disp (:907-990) -> Crop at (y0,x0) (:532-575; RandomCrop +
// CenterCrop collapse to one offset) -> rot x Rotate90 (:1041-1071) -> RedistColor (:698-715) ->
// Brightness (:773-782); Contrast (:740-751) needs the mean of the finished horizontal stack and runs
// as a second pass.
// ---------------------------------------------------------------------------------------------
struct PatchSample {
    int scene, f, y0, x0, rot, flags;      // flags: 1 = shift, 2 = colour, 4 = brightness
    float fdiv, disp, bright;
};
#define PATCH_IP 8
#define PATCH_FP 4

__device__ __forceinline__ PatchSample patch_sample(const int32_t *ip, const float *fp, int b)
{
    PatchSample p;
    p.scene = ip[b * PATCH_IP + 0]; p.f = ip[b * PATCH_IP + 1]; p.y0 = ip[b * PATCH_IP + 2];
    p.x0 = ip[b * PATCH_IP + 3]; p.rot = ip[b * PATCH_IP + 4]; p.flags = ip[b * PATCH_IP + 5];
    p.fdiv = fp[b * PATCH_FP + 0]; p.disp = fp[b * PATCH_FP + 1]; p.bright = fp[b * PATCH_FP + 2];
    return p;
}

// patch coordinates before `rot` applications of out[y][x] = in[x][ps-1-y] (flip(transpose(.)), hci4d.py:1058-1061)
__device__ __forceinline__ void unrotate(int rot, int ps, int &y, int &x)
{
    for (int k = 0; k < rot; ++k) {
        const int t = y;
        y = x;
        x = ps - 1 - t;
    }
}

// RedistColor: float64 matrix entry x float32 image, rounded to float32 after every step (numpy >= 2)
__device__ __forceinline__ void redist_color(const double *m, float &c0, float &c1, float &c2)
{
    const double s0 = c0, s1 = c1, s2 = c2;
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = (float)__dmul_rn(m[3 * r], s0);
        a = (float)__dadd_rn((double)a, __dmul_rn(m[3 * r + 1], s1));
        a = (float)__dadd_rn((double)a, __dmul_rn(m[3 * r + 2], s2));
        o[r] = a;
    }
    c0 = o[0]; c1 = o[1]; c2 = o[2];
}

// grid: (B * 4 stacks * V views * ps rows); threads over x.  out: (4, B, V, 3, ps, ps)
__global__ __launch_bounds__(128) void patch_stacks_kernel(
    const float *__restrict__ stacks, const int32_t *__restrict__ ip, const float *__restrict__ fp,
    const int32_t *__restrict__ tab_s, const float *__restrict__ tab_w, const double *__restrict__ mat,
    const int32_t *__restrict__ rot_src, float *__restrict__ out, double *__restrict__ mean_sum, int B, int V,
    int Hf, int Wf, int ps)
{
    int r = blockIdx.x;
    const int y = r % ps; r /= ps;
    const int n = r % V; r /= V;
    const int so = r % 4;
    const int b = r / 4;
    const PatchSample p = patch_sample(ip, fp, b);
    const int src = rot_src[(p.rot * 4 + so) * V + n];
    const int ss = src / V, sn = src % V;                 // stack and view the pixel comes from
    const int Hd = (Hf + p.f - 1) / p.f, Wd = (Wf + p.f - 1) / p.f;
    ShiftTab t;
    t.s0 = tab_s[2 * (b * V + sn)]; t.s1 = tab_s[2 * (b * V + sn) + 1];
    t.w0 = tab_w[2 * (b * V + sn)]; t.w1 = tab_w[2 * (b * V + sn) + 1];
    const bool do_shift = p.flags & 1;
    const size_t plane = (size_t)Hf * Wf;
    const float *base = stacks + (((size_t)p.scene * 4 + ss) * V + sn) * 3 * plane;
    auto lerp = [&](float a, float c) { return __fadd_rn(__fmul_rn(a, t.w0), __fmul_rn(c, t.w1)); };
    double local = 0.0;
    for (int x = threadIdx.x; x < ps; x += blockDim.x) {
        int yy = y, xx = x;
        unrotate(p.rot, ps, yy, xx);
        const int Y = p.y0 + yy, X = p.x0 + xx;           // downsampled-frame coordinates
        float c[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float *pl = base + ch * plane;
            auto at = [&](int yd, int xd) { return pl[(size_t)(yd * p.f) * Wf + xd * p.f]; };
            if (!do_shift) { c[ch] = at(Y, X); continue; }
            const int x0 = roll_src(X, t.s0, Wd), x1 = roll_src(X, t.s1, Wd);
            if (ss == 0) { c[ch] = lerp(at(Y, x0), at(Y, x1)); continue; }
            const int sg = ss == 2 ? -1 : 1;              // the I stack's vertical pass rolls by -s (:971-975)
            const int ya = roll_src(Y, sg * t.s0, Hd), yb = roll_src(Y, sg * t.s1, Hd);
            if (ss == 1) { c[ch] = lerp(at(ya, X), at(yb, X)); continue; }
            c[ch] = lerp(lerp(at(ya, x0), at(ya, x1)), lerp(at(yb, x0), at(yb, x1)));
        }
        if (p.flags & 2) redist_color(mat + 9 * b, c[0], c[1], c[2]);
        if (p.flags & 4) { c[0] = __fmul_rn(c[0], p.bright); c[1] = __fmul_rn(c[1], p.bright); c[2] = __fmul_rn(c[2], p.bright); }
        float *o = out + ((((size_t)so * B + b) * V + n) * 3 * ps + y) * ps + x;
        o[0] = c[0]; o[(size_t)ps * ps] = c[1]; o[(size_t)2 * ps * ps] = c[2];
        local += (double)c[0] + (double)c[1] + (double)c[2];
    }
    if (so == 0 && mean_sum) {                            // Contrast's mean is over data[0] (:741)
        __shared__ double red[128];
        red[threadIdx.x] = local;
        __syncthreads();
        for (int k = 64; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(mean_sum + b, red[0]);
    }
}

// centre view, ground truth, MPI planes and mask.  grid: (B * (3 + 1 + 5P + 1) planes * ps rows)
__global__ __launch_bounds__(128) void patch_planes_kernel(
    const float *__restrict__ center, const float *__restrict__ gt, const float *__restrict__ mpi,
    const int32_t *__restrict__ mask, const int32_t *__restrict__ ip, const float *__restrict__ fp,
    const double *__restrict__ mat, float *__restrict__ o_center, float *__restrict__ o_gt,
    float *__restrict__ o_mpi, int32_t *__restrict__ o_mask, int B, int P, int Hf, int Wf, int ps)
{
    const int nplanes = 1 + 1 + 5 * P + 1;                // centre (3 channels at once), gt, mpi, mask
    int r = blockIdx.x;
    const int y = r % ps; r /= ps;
    const int k = r % nplanes;
    const int b = r / nplanes;
    const PatchSample p = patch_sample(ip, fp, b);
    const size_t plane = (size_t)Hf * Wf;
    for (int x = threadIdx.x; x < ps; x += blockDim.x) {
        int yy = y, xx = x;
        if (k != nplanes - 1) unrotate(p.rot, ps, yy, xx);            // the mask is not rotated (:1056)
        const size_t src = (size_t)((p.y0 + yy) * p.f) * Wf + (p.x0 + xx) * p.f;
        const size_t dst = (size_t)y * ps + x;
        if (k == 0) {                                                  // centre: no shift (:929-978 touch stacks only)
            const float *pc = center + (size_t)p.scene * 3 * plane;
            float c0 = pc[src], c1 = pc[plane + src], c2 = pc[2 * plane + src];
            if (p.flags & 2) redist_color(mat + 9 * b, c0, c1, c2);
            if (p.flags & 4) { c0 = __fmul_rn(c0, p.bright); c1 = __fmul_rn(c1, p.bright); c2 = __fmul_rn(c2, p.bright); }
            float *o = o_center + (size_t)b * 3 * ps * ps + dst;
            o[0] = c0; o[(size_t)ps * ps] = c1; o[(size_t)2 * ps * ps] = c2;
        } else if (k == 1) {                                           // gt / f - disp (:506-507, :982-983)
            float g = __fdiv_rn(gt[(size_t)p.scene * plane + src], p.fdiv);
            if (p.flags & 1) g = __fsub_rn(g, p.disp);
            o_gt[(size_t)b * ps * ps + dst] = g;
        } else if (k == nplanes - 1) {
            o_mask[(size_t)b * ps * ps + dst] = mask[(size_t)p.scene * plane + src];
        } else {                                                       // mpi[:, 4] carries the same corrections
            const int q = k - 2;
            float v = mpi[((size_t)p.scene * 5 * P + q) * plane + src];
            if (q % 5 == 4) {
                v = __fdiv_rn(v, p.fdiv);
                if (p.flags & 1) v = __fsub_rn(v, p.disp);
            }
            o_mpi[((size_t)b * 5 * P + q) * ps * ps + dst] = v;
        }
    }
}

// Contrast (:740-751): x*alpha + mean*(1-alpha) on the four stacks and the centre view
__global__ void patch_contrast_kernel(float *__restrict__ stacks, float *__restrict__ center,
                                      const double *__restrict__ mean_sum, const float *__restrict__ alpha,
                                      int B, int per_stack /* V*3*ps*ps */, int per_center)
{
    const long long n_st = (long long)4 * B * per_stack, total = n_st + (long long)B * per_center;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        float *ptr;
        int b;
        if (idx < n_st) { b = (int)((idx / per_stack) % B); ptr = stacks + idx; }
        else { b = (int)((idx - n_st) / per_center); ptr = center + (idx - n_st); }
        const float a = alpha[2 * b];                               // float32(alpha), float32(1 - alpha)
        const float mean = (float)(mean_sum[b] / (double)per_stack);
        const float off = __fmul_rn(mean, alpha[2 * b + 1]);
        *ptr = __fadd_rn(__fmul_rn(*ptr, a), off);
    }
}

// Ensamble reduce (ensamble.py:78-101)
__global__ void ensamble_reduce_kernel(const float *__restrict__ means, const float *__restrict__ logvars,
                                       const float *__restrict__ grid, float *__restrict__ mean,
                                       float *__restrict__ logvar, float *__restrict__ post, int S, int HW,
                                       long long total /* B*HW */)
{
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW;
        const int p = (int)(idx - b * HW);
        float best = INFINITY;
        int bi = 0;
        for (int j = 0; j < S; ++j) {
            const float lv = logvars[(size_t)j * total + idx];
            if (lv < best) { best = lv; bi = j; }
        }
        mean[idx] = means[(size_t)bi * total + idx];
        logvar[idx] = best;
        for (int k = 0; k < S; ++k) {
            const float gk = grid[k];
            float acc = 0.f;
            for (int j = 0; j < S; ++j)
                acc = __fadd_rn(acc, laplace_pdf(gk, means[(size_t)j * total + idx],
                                                 expf(logvars[(size_t)j * total + idx])));
            post[(b * S + k) * HW + p] = __fdiv_rn(acc, (float)S);
        }
    }
}

// Discretised Laplace mixture (validate/cli.py:74-118): out[b][k][p] = mean over members s of
// cdf(edge[k+1]; m, v) - cdf(edge[k]; m, v) with m = means[s][b][p], v = exp(logvars[s][b][p]) (float32 exp,
// then float64 like numpy's promotion), accumulated member by member in float64.
__device__ __forceinline__ double cdf_laplace_f64(double x, double m, double v)
{
    const double z = (x - m) / v;
    return x < m ? exp(z) / 2.0 : 1.0 - exp(-z) / 2.0;
}

__global__ void lmm_to_discrete_kernel(const float *__restrict__ means, const float *__restrict__ logvars,
                                       const double *__restrict__ edges, double *__restrict__ out, int S, int B,
                                       int n_bins, long long HW)
{
    const long long total = (long long)B * n_bins * HW;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long p = idx % HW;
        const long long r = idx / HW;
        const int k = (int)(r % n_bins);
        const long long b = r / n_bins;
        const double e0 = edges[k], e1 = edges[k + 1];
        double acc = 0.0;
        for (int s = 0; s < S; ++s) {
            const size_t o = ((size_t)s * B + b) * HW + p;
            const double m = (double)means[o];
            const double v = (double)expf(logvars[o]);
            acc += cdf_laplace_f64(e1, m, v) - cdf_laplace_f64(e0, m, v);
        }
        out[idx] = acc / (double)S;
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI wrappers
// ---------------------------------------------------------------------------------------------
static int ew_blocks(long long total, int per = 256)
{
    long long b = (total + per - 1) / per;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int mmlf_bn_stats_train(const float *z, int cs, int C, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, double momentum, double eps,
                                   float *save_mean, float *save_invstd, float *scale, float *shift,
                                   double *partial, int nblocks, int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(z && save_mean && save_invstd && scale && shift && partial, "mmlf_bn_stats_train: null pointer");
    MMLF_CHECK_ARG(cs % 4 == 0 && C > 0 && C <= cs && (C + 3) / 4 <= 256, "mmlf_bn_stats_train: C=%d cs=%d", C, cs);
    MMLF_CHECK_ARG(nblocks > 0 && nblocks <= 4096, "mmlf_bn_stats_train: nblocks=%d", nblocks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_reduce_kernel<4>), dim3(nblocks), dim3(256), 0, st, z, cs, nullptr, 0, 0, nullptr,
                       nullptr, nullptr, nullptr, C, partial, B, H, W);
    const double n = (double)B * H * W;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, st, partial, nblocks, C, n, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    return mmlf_launch_status("mmlf_bn_stats_train");
}

extern "C" int mmlf_bn_stats_finalize(const double *partial, int nblocks, int C, const float *gamma,
                                      const float *beta, float *running_mean, float *running_var, double momentum,
                                      double eps, float *save_mean, float *save_invstd, float *scale, float *shift,
                                      int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(partial && save_mean && save_invstd && scale && shift && C > 0 && nblocks > 0,
                   "mmlf_bn_stats_finalize: bad argument");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, nblocks, C,
                       (double)B * H * W, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                       save_invstd, scale, shift);
    return mmlf_launch_status("mmlf_bn_stats_finalize");
}

extern "C" int mmlf_bn_coeffs_eval(const float *gamma, const float *beta, const float *rm, const float *rv,
                                   double eps, float *scale, float *shift, int C, void *stream)
{
    MMLF_CHECK_ARG(rm && rv && scale && shift && C > 0, "mmlf_bn_coeffs_eval: bad argument");
    hipLaunchKernelGGL(bn_coeffs_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta, rm,
                       rv, eps, scale, shift, C);
    return mmlf_launch_status("mmlf_bn_coeffs_eval");
}

extern "C" int mmlf_fold_bn_eval(const float *w_oihw, const float *bias, const float *scale, const float *shift,
                                 float *w_out, float *bias_out, int Cout, int Cin, void *stream)
{
    MMLF_CHECK_ARG(w_oihw && scale && shift && w_out && bias_out && Cout > 0 && Cin > 0, "mmlf_fold_bn_eval: bad argument");
    const long long total = (long long)Cout * Cin * 4;
    hipLaunchKernelGGL(fold_bn_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, bias, scale,
                       shift, w_out, bias_out, Cout, Cin * 4);
    return mmlf_launch_status("mmlf_fold_bn_eval");
}

extern "C" int mmlf_bn_apply_relu(const float *z, int cs_z, int C, const float *scale, const float *shift, float *y,
                                  int cs_y, int c_off, int C_store, int B, int H, int W, float *amax_out,
                                  void *stream)
{
    MMLF_CHECK_ARG(z && scale && shift && y, "mmlf_bn_apply_relu: null pointer");
    MMLF_CHECK_ARG(cs_z % 4 == 0 && cs_y % 2 == 0 && c_off % 2 == 0 && C <= cs_z && C_store >= C &&
                       c_off + C_store <= cs_y,
                   "mmlf_bn_apply_relu: C=%d cs_z=%d cs_y=%d c_off=%d C_store=%d", C, cs_z, cs_y, c_off, C_store);
    const int nrows = B * (H + MMLF_GRID_PAD_H);
    hipLaunchKernelGGL((bn_rows_kernel<0, 4>), dim3(nrows), dim3(256), 6 * (C_store + 4) * sizeof(float), (hipStream_t)stream, z, cs_z,
                       nullptr, 0, 0, scale, shift, nullptr, nullptr, C, y, cs_y, c_off, C_store, H, W, amax_out, nrows);
    return mmlf_launch_status("mmlf_bn_apply_relu");
}

extern "C" int mmlf_bn_apply_relu4(const float *const z[4], int cs_z, int C, const float *const scale[4],
                                   const float *const shift[4], float *y, int cs_y, int B, int H, int W,
                                   float *amax_out, void *stream)
{
    MMLF_CHECK_ARG(z && scale && shift && y && B > 0 && H > 0 && W > 0, "mmlf_bn_apply_relu4: bad argument");
    MMLF_CHECK_ARG(C > 0 && C % 2 == 0 && C <= cs_z && cs_z % 2 == 0 && cs_y == 4 * C && 8 * C * sizeof(float) <= 48 * 1024,
                   "mmlf_bn_apply_relu4: C=%d cs_z=%d cs_y=%d (needs even C, cs_y == 4 * C)", C, cs_z, cs_y);
    BnApply4 s;
    for (int k = 0; k < 4; ++k) {
        MMLF_CHECK_ARG(z[k] && scale[k] && shift[k], "mmlf_bn_apply_relu4: null pointer in source %d", k);
        s.z[k] = z[k]; s.scale[k] = scale[k]; s.shift[k] = shift[k];
    }
    const int nrows = B * (H + MMLF_GRID_PAD_H);
    hipLaunchKernelGGL(bn_apply4_kernel, dim3(nrows), dim3(256), 8 * C * sizeof(float), (hipStream_t)stream, s, cs_z, C,
                       y, H, W, amax_out, nrows);
    return mmlf_launch_status("mmlf_bn_apply_relu4");
}

extern "C" int mmlf_bn_bwd_reduce(const float *gy, int cs_gy, int c_off, const float *z, int cs_z, int C,
                                  const float *scale, const float *shift, const float *gamma,
                                  const float *save_mean, const float *save_invstd, float *dgamma, float *dbeta,
                                  int accumulate, float *coef, double *partial, int nblocks, int B, int H, int W,
                                  void *stream)
{
    MMLF_CHECK_ARG(gy && z && scale && shift && save_mean && save_invstd && coef && partial,
                   "mmlf_bn_bwd_reduce: null pointer");
    MMLF_CHECK_ARG(cs_gy % 2 == 0 && c_off % 2 == 0 && cs_z % 4 == 0 && C <= cs_z && (C + 1) / 2 <= 256,
                   "mmlf_bn_bwd_reduce: layout");
    MMLF_CHECK_ARG(nblocks > 0 && nblocks <= 4096, "mmlf_bn_bwd_reduce: nblocks=%d", nblocks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_reduce_bwd_kernel, dim3(nblocks), dim3(256), 0, st, z, cs_z, gy, cs_gy, c_off,
                       scale, shift, save_mean, save_invstd, C, partial, B, H, W);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, partial, nblocks, C,
                       (double)B * H * W, gamma, save_invstd, dgamma, dbeta, accumulate, coef);
    return mmlf_launch_status("mmlf_bn_bwd_reduce");
}

extern "C" int mmlf_bn_bwd_apply(const float *gy, int cs_gy, int c_off, const float *z, int cs_z, int C,
                                 const float *scale, const float *shift, const float *save_mean, const float *coef,
                                 float *dz, int cs_dz, int B, int H, int W, float *amax_out, void *stream)
{
    MMLF_CHECK_ARG(gy && z && scale && shift && save_mean && coef && dz, "mmlf_bn_bwd_apply: null pointer");
    MMLF_CHECK_ARG(cs_gy % 2 == 0 && c_off % 2 == 0 && cs_z % 4 == 0 && cs_dz % 4 == 0 && C <= cs_dz,
                   "mmlf_bn_bwd_apply: layout");
    const int nrows = B * (H + MMLF_GRID_PAD_H);
    hipLaunchKernelGGL((bn_rows_kernel<1, 4>), dim3(nrows), dim3(256), 6 * (cs_dz + 4) * sizeof(float), (hipStream_t)stream, z, cs_z, gy,
                       cs_gy, c_off, scale, shift, save_mean, coef, C, dz, cs_dz, 0, cs_dz, H, W, amax_out, nrows);
    return mmlf_launch_status("mmlf_bn_bwd_apply");
}

extern "C" int mmlf_pack_nchw(const float *nchw, int C, float *grid, int cs, int B, int H, int W, float *amax_out,
                              void *stream)
{
    MMLF_CHECK_ARG(nchw && grid && C > 0 && cs % 4 == 0 && C <= cs, "mmlf_pack_nchw: C=%d cs=%d", C, cs);
    int xt = PACK_XT;            // DPP with many views packs a gradient of 4*views*3 channels (132 at 11 views)
    static const size_t tile_limit = [] { const char *e = getenv("MMLF_PACK_LDS_KB"); return (size_t)(e ? atoi(e) : 32) * 1024; }();
    while (xt > 4 && (size_t)cs * (xt | 1) * sizeof(float) > tile_limit) xt >>= 1;
    const size_t lds = (size_t)cs * (xt | 1) * sizeof(float);
    MMLF_CHECK_ARG(lds <= 64 * 1024, "mmlf_pack_nchw: cs=%d does not fit the transpose tile", cs);
    hipLaunchKernelGGL(pack_nchw_kernel, dim3(B * (H + MMLF_GRID_PAD_H)), dim3(256), lds, (hipStream_t)stream, nchw, C, grid, cs, H, W,
                       amax_out, xt);
    return mmlf_launch_status("mmlf_pack_nchw");
}

// zero the head and tail slack of a grid buffer in one launch
__global__ void zero_slack_kernel(float *__restrict__ buf, long long head, long long tail_off, long long tail,
                                  float *__restrict__ amax, long long n_amax)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < head + tail + n_amax;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < head + tail) buf[i < head ? i : tail_off + (i - head)] = 0.f;
        else amax[i - head - tail] = 0.f;
    }
}

extern "C" int mmlf_zero_slack(float *grid, int cs, int B, int H, int W, float *amax, void *stream)
{
    MMLF_CHECK_ARG(grid && cs > 0 && B > 0 && H > 0 && W > 0, "mmlf_zero_slack: bad argument");
    const Grid g = make_grid(B, H, W);
    const long long head = (long long)(g.P + 1) * cs, tail_off = g.NQ * cs;
    const long long tail = (grid_alloc_positions(g) - g.NQ) * cs;
    const long long n_amax = amax ? amax_entries(g) : 0;
    hipLaunchKernelGGL(zero_slack_kernel, dim3(ew_blocks(head + tail + n_amax)), dim3(256), 0, (hipStream_t)stream, grid,
                       head, tail_off, tail, amax, n_amax);
    return mmlf_launch_status("mmlf_zero_slack");
}

struct ZeroSlack4 { float *buf[4]; float *amax[4]; long long head[4], tail_off[4], tail[4], end[4]; long long n_amax; };
__global__ void zero_slack4_kernel(ZeroSlack4 z)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < z.end[3]; i += (long long)gridDim.x * blockDim.x) {
        const int k = i < z.end[0] ? 0 : i < z.end[1] ? 1 : i < z.end[2] ? 2 : 3;
        const long long j = i - (k ? z.end[k - 1] : 0);
        if (j < z.head[k] + z.tail[k]) z.buf[k][j < z.head[k] ? j : z.tail_off[k] + (j - z.head[k])] = 0.f;
        else z.amax[k][j - z.head[k] - z.tail[k]] = 0.f;
    }
}

extern "C" int mmlf_zero_slack4(float *const grid[4], const int cs[4], float *const amax[4], int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(grid && cs && amax && B > 0 && H > 0 && W > 0, "mmlf_zero_slack4: bad argument");
    const Grid g = make_grid(B, H, W);
    ZeroSlack4 z;
    long long run = 0;
    for (int k = 0; k < 4; ++k) {
        z.buf[k] = grid[k]; z.amax[k] = amax[k];
        z.head[k] = z.tail_off[k] = z.tail[k] = 0;
        if (grid[k]) {
            MMLF_CHECK_ARG(cs[k] > 0, "mmlf_zero_slack4: cs[%d]=%d", k, cs[k]);
            z.head[k] = (long long)(g.P + 1) * cs[k];
            z.tail_off[k] = g.NQ * cs[k];
            z.tail[k] = (grid_alloc_positions(g) - g.NQ) * cs[k];
            run += z.head[k] + z.tail[k] + (amax[k] ? amax_entries(g) : 0);
        }
        z.end[k] = run;
    }
    z.n_amax = amax_entries(g);
    if (run == 0) return 0;
    hipLaunchKernelGGL(zero_slack4_kernel, dim3(ew_blocks(run)), dim3(256), 0, (hipStream_t)stream, z);
    return mmlf_launch_status("mmlf_zero_slack4");
}

extern "C" int mmlf_unpack_nchw(const float *grid, int cs, float *nchw, int C, int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(nchw && grid && C > 0 && C <= cs, "mmlf_unpack_nchw: C=%d cs=%d", C, cs);
    int xt = 32;
    while (xt > 1 && (size_t)xt * (C | 1) * sizeof(float) > 32 * 1024) xt >>= 1;
    hipLaunchKernelGGL(unpack_nchw_kernel, dim3(B * H), dim3(256), (size_t)xt * (C | 1) * sizeof(float), (hipStream_t)stream,
                       grid, cs, nchw, C, H, W, xt);
    return mmlf_launch_status("mmlf_unpack_nchw");
}

extern "C" int mmlf_head_upr(const float *output, const float *grid108, float *posterior, int steps, int B, int H,
                             int W, void *stream)
{
    MMLF_CHECK_ARG(output && grid108 && posterior && steps > 0, "mmlf_head_upr: bad argument");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(head_upr_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, output, grid108,
                       posterior, steps, H * W, total);
    return mmlf_launch_status("mmlf_head_upr");
}

extern "C" int mmlf_head_dpp(const float *scores, const float *grid_torch, const float *grid_np, float *one_hot,
                             float *posterior, float *mean, float *logvar, int steps, int B, int H, int W,
                             void *stream)
{
    MMLF_CHECK_ARG(scores && grid_torch && grid_np && one_hot && posterior && mean && logvar && steps > 0,
                   "mmlf_head_dpp: bad argument");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(head_dpp_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, scores, grid_torch,
                       grid_np, one_hot, posterior, mean, logvar, steps, H * W, total);
    return mmlf_launch_status("mmlf_head_dpp");
}

extern "C" int mmlf_head_upr_bwd(const float *output, const float *grid108, const float *grad_posterior, float *grad_output,
                                 int steps, int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(output && grid108 && grad_posterior && grad_output && steps > 0, "mmlf_head_upr_bwd: bad argument");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(head_upr_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, output, grid108,
                       grad_posterior, grad_output, steps, H * W, total);
    return mmlf_launch_status("mmlf_head_upr_bwd");
}

extern "C" int mmlf_head_dpp_bwd(const float *scores, const float *grid_np, const float *mean, const float *grad_posterior,
                                 const float *grad_logvar, float *grad_scores, int steps, int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(scores && grid_np && mean && grad_scores && steps > 0 && (grad_posterior || grad_logvar),
                   "mmlf_head_dpp_bwd: bad argument");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(head_dpp_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, scores, grid_np, mean,
                       grad_posterior, grad_logvar, grad_scores, steps, H * W, total);
    return mmlf_launch_status("mmlf_head_dpp_bwd");
}

extern "C" int mmlf_loss_fwd_bwd(int kind, const float *output, int oc, const float *gt, const int32_t *mask,
                                 const float *grid_torch, double half_step, float *loss_out, float *grad,
                                 double *scratch, int nblocks, const double *den_override, int B, int H, int W,
                                 void *stream)
{
    MMLF_CHECK_ARG(kind >= 0 && kind <= 2, "mmlf_loss_fwd_bwd: kind=%d", kind);
    MMLF_CHECK_ARG(output && gt && mask && loss_out && scratch, "mmlf_loss_fwd_bwd: null pointer");
    MMLF_CHECK_ARG((kind == 0 && oc >= 1) || (kind == 1 && oc >= 2) || (kind == 2 && grid_torch && oc >= 1),
                   "mmlf_loss_fwd_bwd: oc=%d for kind=%d", oc, kind);
    MMLF_CHECK_ARG(nblocks > 0 && nblocks <= 4096, "mmlf_loss_fwd_bwd: nblocks=%d", nblocks);
    const long long total = (long long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nblocks), dim3(256), 0, st, kind, output, oc, gt, mask, grid_torch,
                       (float)half_step, scratch, H * W, total);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, scratch, nblocks, loss_out, den_override);
    if (grad)
        hipLaunchKernelGGL(loss_grad_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, kind, output, oc, gt, mask,
                           grid_torch, (float)half_step, scratch, grad, H * W, total);
    return mmlf_launch_status("mmlf_loss_fwd_bwd");
}

extern "C" int64_t mmlf_loss_multi_scratch_doubles(int nblocks) { return nblocks > 0 ? 4ll * nblocks + 4 : -1; }

extern "C" int mmlf_loss_multi_fwd_bwd(int kind, const float *output, int oc, const float *target, int P,
                                       const int32_t *mask, const int32_t *mask_padding, const float *grid_torch,
                                       double half_step, float *loss_out, float *grad, double *scratch, int nblocks,
                                       const double *den_override, const double *aux_override, int B, int H, int W,
                                       void *stream)
{
    MMLF_CHECK_ARG(kind >= LOSS_MULTI_L1 && kind <= LOSS_UPR_PADDED, "mmlf_loss_multi_fwd_bwd: kind=%d", kind);
    MMLF_CHECK_ARG(output && target && mask && loss_out && scratch, "mmlf_loss_multi_fwd_bwd: null pointer");
    MMLF_CHECK_ARG(kind == LOSS_UPR_PADDED ? mask_padding != nullptr : P >= 1, "mmlf_loss_multi_fwd_bwd: P=%d / mask_padding", P);
    MMLF_CHECK_ARG((kind == LOSS_MULTI_L1 && oc >= 1) || (kind == LOSS_MULTI_CE && grid_torch && oc >= 1) ||
                       ((kind == LOSS_MULTI_UPR || kind == LOSS_UPR_PADDED) && oc >= 2),
                   "mmlf_loss_multi_fwd_bwd: oc=%d for kind=%d", oc, kind);
    MMLF_CHECK_ARG(nblocks > 0 && nblocks <= 4096 && B > 0 && H > 0 && W > 0, "mmlf_loss_multi_fwd_bwd: nblocks=%d", nblocks);
    MultiLossArgs a;
    a.out = output; a.target = target; a.mask = mask; a.mask_padding = mask_padding; a.grid = grid_torch;
    a.scratch = scratch; a.grad = grad; a.den_override = den_override; a.aux_override = aux_override;
    a.half_step = (float)half_step; a.kind = kind; a.oc = oc; a.P = P; a.HW = H * W; a.nblocks = nblocks;
    a.total = (long long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (kind == LOSS_MULTI_UPR || kind == LOSS_UPR_PADDED) {
        if (!aux_override) hipLaunchKernelGGL(loss_multi_aux_kernel, dim3(nblocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(loss_multi_aux_finalize_kernel, dim3(1), dim3(64), 0, st, a);
    }
    hipLaunchKernelGGL(loss_multi_partial_kernel, dim3(nblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, scratch, nblocks, loss_out, den_override);
    if (grad) hipLaunchKernelGGL(loss_multi_grad_kernel, dim3(ew_blocks(a.total)), dim3(256), 0, st, a);
    return mmlf_launch_status("mmlf_loss_multi_fwd_bwd");
}

extern "C" int mmlf_adam_step(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1,
                              double beta2, double eps, int64_t step, double grad_scale, void *stream)
{
    MMLF_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "mmlf_adam_step: bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                       (float)(lr / bc1), (float)beta1, (float)beta2, (float)eps, (float)sqrt(bc2), (float)grad_scale);
    return mmlf_launch_status("mmlf_adam_step");
}

extern "C" int mmlf_shift_views(const float *h, const float *v, const float *i, const float *d, float *oh, float *ov,
                                float *oi, float *od, const int32_t *tab_s, const float *tab_w, int S, int views,
                                int H, int W, void *stream)
{
    MMLF_CHECK_ARG(h && v && i && d && oh && ov && oi && od && tab_s && tab_w, "mmlf_shift_views: null pointer");
    MMLF_CHECK_ARG(S > 0 && views > 0 && H > 0 && W > 0, "mmlf_shift_views: S=%d views=%d", S, views);
    hipLaunchKernelGGL(shift_kernel, dim3(S * views * 3 * H), dim3(256), 0, (hipStream_t)stream, h, v, i, d, oh, ov,
                       oi, od, tab_s, tab_w, views, H, W);
    return mmlf_launch_status("mmlf_shift_views");
}

extern "C" int mmlf_shift_pack(const float *in, int kind, float *grid, int cs, const int32_t *tab_s, const float *tab_w,
                               int S, int views, int H, int W, float *amax_out, void *stream)
{
    MMLF_CHECK_ARG(in && grid && tab_s && tab_w, "mmlf_shift_pack: null pointer");
    MMLF_CHECK_ARG(kind >= 0 && kind <= 3, "mmlf_shift_pack: kind=%d", kind);
    MMLF_CHECK_ARG(S > 0 && views > 0 && H > 0 && W > 0 && cs % 4 == 0 && views * 3 <= cs, "mmlf_shift_pack: S=%d views=%d cs=%d",
                   S, views, cs);
    int xt = PACK_XT, xt_log = 7;                 // 128 positions per piece, halved until the tile fits 32 KB
    static_assert(PACK_XT == 128, "xt_log");
    while (xt > 4 && (size_t)cs * (xt | 1) * sizeof(float) > 32 * 1024) { xt >>= 1; --xt_log; }
    const size_t lds = (size_t)cs * (xt | 1) * sizeof(float) + (size_t)views * sizeof(ShiftPackTab);
    MMLF_CHECK_ARG(lds <= 64 * 1024 && cs / 4 <= 256, "mmlf_shift_pack: cs=%d does not fit the transpose tile", cs);
    xt = xt_log;
    const dim3 g((unsigned)(S * (H + MMLF_GRID_PAD_H))), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
    case 0: hipLaunchKernelGGL(shift_pack_kernel<0>, g, b, lds, st, in, views, grid, cs, tab_s, tab_w, H, W, amax_out, xt); break;
    case 1: hipLaunchKernelGGL(shift_pack_kernel<1>, g, b, lds, st, in, views, grid, cs, tab_s, tab_w, H, W, amax_out, xt); break;
    case 2: hipLaunchKernelGGL(shift_pack_kernel<2>, g, b, lds, st, in, views, grid, cs, tab_s, tab_w, H, W, amax_out, xt); break;
    default: hipLaunchKernelGGL(shift_pack_kernel<3>, g, b, lds, st, in, views, grid, cs, tab_s, tab_w, H, W, amax_out, xt); break;
    }
    return mmlf_launch_status("mmlf_shift_pack");
}

extern "C" int mmlf_ensamble_reduce(const float *means, const float *logvars, const float *grid, float *mean,
                                    float *logvar, float *posterior, int S, int B, int H, int W, void *stream)
{
    MMLF_CHECK_ARG(means && logvars && grid && mean && logvar && posterior && S > 0, "mmlf_ensamble_reduce: bad argument");
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(ensamble_reduce_kernel, dim3(ew_blocks(total, 64)), dim3(64), 0, (hipStream_t)stream, means,
                       logvars, grid, mean, logvar, posterior, S, H * W, total);
    return mmlf_launch_status("mmlf_ensamble_reduce");
}

extern "C" int mmlf_patch_gather(const float *stacks, const float *center, const float *gt, const float *mpi,
                                 const int32_t *mask, int S, int V, int P, int Hf, int Wf, const int32_t *iparam,
                                 const float *fparam, const int32_t *tab_s, const float *tab_w, const double *mat,
                                 const int32_t *rot_src, float *o_stacks, float *o_center, float *o_gt, float *o_mpi,
                                 int32_t *o_mask, double *mean_sum, int B, int ps, void *stream)
{
    MMLF_CHECK_ARG(stacks && center && gt && mask && iparam && fparam && tab_s && tab_w && mat && rot_src,
                   "mmlf_patch_gather: null input");
    MMLF_CHECK_ARG(o_stacks && o_center && o_gt && o_mask, "mmlf_patch_gather: null output");
    MMLF_CHECK_ARG(S > 0 && V > 0 && P >= 0 && Hf > 0 && Wf > 0 && B > 0 && ps > 0,
                   "mmlf_patch_gather: bad shape S=%d V=%d P=%d frame %dx%d B=%d ps=%d", S, V, P, Hf, Wf, B, ps);
    MMLF_CHECK_ARG(P == 0 || (mpi && o_mpi), "mmlf_patch_gather: P=%d but no mpi buffers", P);
    MMLF_CHECK_ARG((long long)B * 4 * V * ps < (1ll << 31), "mmlf_patch_gather: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(patch_stacks_kernel, dim3((unsigned)(B * 4 * V * ps)), dim3(128), 0, st, stacks, iparam, fparam,
                       tab_s, tab_w, mat, rot_src, o_stacks, mean_sum, B, V, Hf, Wf, ps);
    hipLaunchKernelGGL(patch_planes_kernel, dim3((unsigned)(B * (3 + 5 * P) * ps)), dim3(128), 0, st, center, gt, mpi,
                       mask, iparam, fparam, mat, o_center, o_gt, o_mpi, o_mask, B, P, Hf, Wf, ps);
    return mmlf_launch_status("mmlf_patch_gather");
}

extern "C" int mmlf_patch_contrast(float *o_stacks, float *o_center, const double *mean_sum, const float *alpha,
                                   int B, int V, int ps, void *stream)
{
    MMLF_CHECK_ARG(o_stacks && o_center && mean_sum && alpha && B > 0 && V > 0 && ps > 0,
                   "mmlf_patch_contrast: bad argument");
    const long long total = (long long)B * (4 * V + 1) * 3 * ps * ps;
    hipLaunchKernelGGL(patch_contrast_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, o_stacks,
                       o_center, mean_sum, alpha, B, V * 3 * ps * ps, 3 * ps * ps);
    return mmlf_launch_status("mmlf_patch_contrast");
}

extern "C" int mmlf_lmm_to_discrete(const float *means, const float *logvars, const double *edges, double *out,
                                    int S, int B, int n_bins, long long HW, void *stream)
{
    MMLF_CHECK_ARG(means && logvars && edges && out && S > 0 && B > 0 && n_bins > 0 && HW > 0,
                   "mmlf_lmm_to_discrete: bad argument");
    const long long total = (long long)B * n_bins * HW;
    hipLaunchKernelGGL(lmm_to_discrete_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, means,
                       logvars, edges, out, S, B, n_bins, HW);
    return mmlf_launch_status("mmlf_lmm_to_discrete");
}
