// wgrad.hip -- weight + bias gradient of the k=2 convolutions on the padded NHWC grid (exact-f32 MFMA kernel and the
// split-arithmetic kernels wgrad4tap_x6w / x6n), the thin (1-2 output channel) weight gradient, their C ABI entry points.
// gfx950 only.  Arithmetic replaced: the weight / bias gradient of nn.Conv2d(k=2, pad 1|0), reference
// mmlf/model/feed_forward.py:123,125 under autograd (mmlf/train/cli.py:257).
#include "conv_device.h"

#ifdef MMLF_BOUNDS_DEBUG
int mmlf_oob_counts_wgrad(unsigned long long *host8, int reset)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_mmlf_oob), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_mmlf_oob), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel: dW[t][ci][co] = sum_q in[q + off_t][ci] * g[q + g_shift][co]
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float *in;
    const float *g;
    float *part;          // [nsplit][4][CIP][NP]
    long long NQpad;
    int cs_in, cin, cs_g, g_shift, P, nsplit, nslice, chunks_per_split, nchunks;
    const float *in_amax, *g_amax;   // f16 split: amax arrays of in and g (common.h)
    const float *chunk_scales;       // f16 split: [nchunks][2] power-of-two operand scales (wgrad_chunk_scales_kernel)
    long long in_bytes, g_bytes, part_floats;   // what the buffers hold by the ABI's contract: the wide kernel's staging loads are
                                                // range-checked against in_bytes / g_bytes (buffer descriptors); the rest is read by
                                                // the MMLF_BOUNDS_DEBUG build only
};

// Block -> (slice, position split).  Blocks b, b + 8, ... land on one XCD (round-robin dispatch): the slices of one split are
// placed there and share its L2 for their gradient reads.  nsplit need not be a multiple of 8: the 8 * (nsplit / 8) regular
// splits are numbered as they always were (same partial-sum order), the slices of the remaining ones are dealt behind them,
// `per` blocks to an XCD, consecutive slices together (their gradient reads come from memory once per XCD they touch: a few
// percent of the launch's reads at 2 of 42 splits).
__device__ __forceinline__ bool wgrad_block_map(const WgradArgs &a, int &slice, int &split)
{
    const int b = blockIdx.x, x = b & 7, j = b >> 3;
    const int reg = (a.nsplit >> 3) * a.nslice;           // regular blocks per XCD
    if (j < reg) {
        slice = j % a.nslice;
        split = (j / a.nslice) * 8 + x;
        return true;
    }
    const int left = (a.nsplit & 7) * a.nslice, per = (left + 7) >> 3;
    const int m = x * per + (j - reg);
    slice = m % a.nslice;
    split = (a.nsplit & ~7) + m / a.nslice;
    return m < left;
}

static inline unsigned wgrad_grid_blocks(int nslice, int nsplit)
{
    return 8u * (unsigned)((nsplit >> 3) * nslice + ((nsplit & 7) * nslice + 7) / 8);
}

#define WG_KQ 32  // positions per chunk

// 256 threads = 4 waves, wave t = tap t.  Block = (32-channel ci slice, position split).
// MFMA rows = ci (A operand), cols = co (B operand), K = positions.  The first channel past Cin
// is staged as 1.0 so that row Cin of tap 0 accumulates the bias gradient for free.
template <int NT>
__global__ __launch_bounds__(256, 2) void wgrad4tap_kernel(WgradArgs a)
{
    constexpr int NP = NT * 32;
    constexpr int A_FL = 2 * 33 * 32;      // floats
    constexpr int NA = 3;                  // 528 float4 / 256 threads
    constexpr int G_F4 = WG_KQ * NP / 4;   // float4s
    constexpr int NG = G_F4 / 256;         // == NT
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);
    float *Gs = As + A_FL;

    const int tid = threadIdx.x;
    const int lane = tid & 63, t = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    int slice, split;
    if (!wgrad_block_map(a, slice, split)) return;
    const int ci0 = slice * 32;
    int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    float4 ra[NA], rg[NG];
    auto gload = [&](int c) {
        const long long Qc = (long long)c * WG_KQ;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + 256 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < 528) {
                const int row = idx >> 3, f = idx & 7;
                const int seg = row >= 33, pix = row - 33 * seg;
                const int ch = ci0 + 4 * f;
                if (ch < a.cs_in)
                    v = *reinterpret_cast<const float4 *>(a.in + (size_t)(Qc + seg * a.P + pix) * a.cs_in + ch);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / (NP / 4), f = idx - row * (NP / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * f < a.cs_g)
                v = *reinterpret_cast<const float4 *>(a.g + (size_t)(Qc + a.g_shift + row) * a.cs_g + 4 * f);
            rg[j] = v;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + 256 * j;
            if (idx < 528) {
                float4 v = ra[j];          // ones row (bias gradient) patched at store time
                const int ch = ci0 + 4 * (idx & 7);
                if (ch == a.cin) v.x = 1.f;
                if (ch + 1 == a.cin) v.y = 1.f;
                if (ch + 2 == a.cin) v.z = 1.f;
                if (ch + 3 == a.cin) v.w = 1.f;
                reinterpret_cast<float4 *>(As)[idx] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) reinterpret_cast<float4 *>(Gs)[tid + 256 * j] = rg[j];
    };

    if (c_begin < c_end) {
        gload(c_begin);
        for (int c = c_begin; c < c_end; ++c) {
            lstore();
            __syncthreads();
            if (c + 1 < c_end) gload(c + 1);
            const float *ap = As + ((t >> 1) * 33 + (t & 1) + kh) * 32 + i;
            const float *gp = Gs + kh * NP + i;
#pragma unroll
            for (int s = 0; s < WG_KQ / 2; ++s) {
                const float av = ap[2 * s * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = gp[2 * s * NP + 32 * nt];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nt], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    const int CIP = a.nslice * 32;
    float *pp = a.part + ((size_t)(split * 4 + t) * CIP + ci0) * NP;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            pp[(size_t)row * NP + 32 * nt + i] = acc[nt][r];
        }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient kernels, split-bf16 arithmetic (same decomposition as wgrad4tap_kernel).
// K = positions: the bf16 MFMA wants 8 consecutive positions per lane for a fixed channel, i.e. the
// transposed image of the NHWC tiles.  The tiles are split into three bf16 planes while they are staged
// (registers -> LDS, row-major [position][channel] like global memory) and the fragments are fetched
// with ds_read_b64_tr_b16, gfx950's transposing LDS read.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__device__ __forceinline__ bf16x8 tr_frag(const char *lds_row0, int row_stride_bytes)
{
    // two transposed 4-row reads = k 0..3 and 4..7 of this lane's column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(lds_row0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(lds_row0 + 4 * row_stride_bytes));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void split_store4(float4 v, char *p0, int plane_stride_bytes)
{
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2 *>(p0) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(p0 + plane_stride_bytes) = make_uint2(m0, m1);
    *reinterpret_cast<uint2 *>(p0 + 2 * plane_stride_bytes) = make_uint2(l0, l1);
}

// plane-count generic pieces of the weight-gradient kernels: PL = 3 bf16 planes / six cross terms, or
// PL = 2 f16 planes of the scaled operands / three cross terms (see split2_pair_f16)
template <int PL>
__device__ __forceinline__ void split_store4_pl(float4 v, float scale, char *p0, int plane_stride_bytes)
{
    if constexpr (PL == 3) {
        split_store4(v, p0, plane_stride_bytes);
    } else {
        unsigned h0, l0, h1, l1;
        split2_pair_f16<true>(v.x, v.y, scale, h0, l0);
        split2_pair_f16<true>(v.z, v.w, scale, h1, l1);
        *reinterpret_cast<uint2 *>(p0) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(p0 + plane_stride_bytes) = make_uint2(l0, l1);
    }
}
template <int PL>
__device__ __forceinline__ f32x4 mfma16_pl(bf16x8 x, bf16x8 y, f32x4 c)
{
    if constexpr (PL == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
}
// cross terms (activation plane, gradient plane), small ones first
template <int PL> __device__ __forceinline__ constexpr int term_a(int t)
{
    return PL == 3 ? (t == 0 ? 2 : t == 1 ? 0 : t <= 3 ? 1 : 0) : (t == 0 ? 1 : 0);
}
template <int PL> __device__ __forceinline__ constexpr int term_b(int t)
{
    return PL == 3 ? (t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0) : (t == 1 ? 1 : 0);
}
// f16 split of the weight gradient.  The sum runs over all positions, so every chunk of 32 positions must carry
// the SAME product of operand scales -- but the split between the two operands is free per chunk:
//   in * (sA * 2^x)  and  g * (sG * 2^-x),   sA, sG = the tensors' global scales (max |.| -> [2^14, 2^15)),
// with x chosen per chunk by wgrad_chunk_scales_kernel from the maxima of the grid rows the chunk reads: if the
// chunk's activations sit u binades below their tensor's maximum and its gradients v binades below theirs,
// x = (u - v) / 2 gives each operand (u + v) / 2 binades of headroom loss instead of u resp. v, so a chunk keeps
// all 22 bits of both operands as long as its products are within 2^-36 of the largest products of the launch
// (smaller ones are below float32's accumulation error of the sum anyway).  The ones row that yields the bias
// gradient is staged as 1 / sA, i.e. 2^x after scaling: exact in f16 for -24 <= x <= 15, which bounds x.
struct WgradScales { float inv_sa, inv_sg; };
template <int PL> __device__ __forceinline__ WgradScales wgrad_scales(const WgradArgs &a)
{
    WgradScales s = {1.f, 1.f};
    if constexpr (PL == 2) {
        // left behind the per-chunk scales by wgrad_chunk_scales_kernel (the tensors' maxima live in 64 shards)
        s.inv_sa = a.chunk_scales[2 * (size_t)a.nchunks];
        s.inv_sg = a.chunk_scales[2 * (size_t)a.nchunks + 1];
    }
    return s;
}
// operand scales of chunk c (PL == 3: no scaling)
// (Round 5 tried issuing the LOAD of a chunk's scales a whole chunk before their first use -- read at the top of a chunk and used by
// its first staging piece, the wave waits there for a memory round trip -- and the wide kernel got SLOWER: 8.44 against 7.98 ms on
// one box, profiles/r05_kbench_wgrad_zeropad.log; the narrow one did not move.  Left as it was.)
template <int PL> __device__ __forceinline__ void wgrad_chunk_scale(const WgradArgs &a, int c, float &sa, float &sg)
{
    if constexpr (PL == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(a.chunk_scales + 2 * (size_t)c);
        sa = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v.x)));
        sg = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v.y)));
    } else {
        sa = 1.f; sg = 1.f;
    }
}
struct ChunkScaleArgs {
    const float *in_amax, *g_amax;
    float *out;                      // [nchunks][2], then 1 / sA, 1 / sG of the two tensors
    long long NQ;
    int nchunks, P, g_shift, nrows;
    Magic divP;
};
__device__ __forceinline__ int pow2_exponent(float p) { return (int)(__float_as_uint(p) >> 23) - 127; }
__global__ __launch_bounds__(256) void wgrad_chunk_scales_kernel(ChunkScaleArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const float sA = pow2_scale_for(mmlf_amax_tensor_max(a.in_amax)), sG = pow2_scale_for(mmlf_amax_tensor_max(a.g_amax));
    if (c == 0) {                                   // the tensors' scales (inverted) for the weight-gradient kernel
        a.out[2 * (size_t)a.nchunks] = 1.f / sA;
        a.out[2 * (size_t)a.nchunks + 1] = 1.f / sG;
    }
    if (c >= a.nchunks) return;
    const long long q0 = (long long)c * WG_KQ;
    int x = 0;
    if (q0 < a.NQ) {
        // activations: the chunk stages in[q0 .. q0 + 32 + P] (its positions' four taps); gradients: g[q + g_shift].
        // EVERY staged element must stay inside the f16 range, also those that only meet zero gradients (an
        // overflow would put inf * 0 into the sum), so all rows the staging touches count.
        const long long ql = q0 + WG_KQ - 1;
        const unsigned r0 = fastdiv((unsigned)q0, a.divP);
        unsigned r1 = fastdiv((unsigned)(ql + a.P + 1), a.divP);
        if (r1 >= (unsigned)a.nrows) r1 = a.nrows - 1;
        float ma = 0.f, mg = 0.f;
        for (unsigned r = r0; r <= r1; ++r) ma = fmaxf(ma, a.in_amax[MMLF_AMAX_HEAD + r]);
        const unsigned g0 = fastdiv((unsigned)(q0 + a.g_shift), a.divP);
        unsigned g1 = fastdiv((unsigned)(ql + a.g_shift), a.divP);
        if (g1 >= (unsigned)a.nrows) g1 = a.nrows - 1;
        for (unsigned r = g0; r <= g1; ++r) mg = fmaxf(mg, a.g_amax[MMLF_AMAX_HEAD + r]);
        // headroom (binades) of the chunk's operands under the global scales; an all-zero operand takes any scale
        const int u = ma > 0.f ? pow2_exponent(pow2_scale_for(ma)) - pow2_exponent(sA) : 60;
        const int v = mg > 0.f ? pow2_exponent(pow2_scale_for(mg)) - pow2_exponent(sG) : 60;
        const int uu = u < 0 ? 0 : u, vv = v < 0 ? 0 : v;        // (row maxima never exceed the tensor's)
        x = (uu - vv) >> 1;                                        // floor
        x = x > 15 ? 15 : (x < -24 ? -24 : x);
    }
    a.out[2 * (size_t)c] = sA * exp2f((float)x);
    a.out[2 * (size_t)c + 1] = sG * exp2f((float)-x);
}

// ---------------------------------------------------------------------------------------------
// weight gradient on v_mfma_f32_16x16x32_bf16 (K = the chunk's 32 positions).  A workgroup owns a slice
// of 16*MB input channels (MFMA rows, the ones row included) x 16*NB output channels and a split of the
// positions; wave t = tap t accumulates MB x NB tiles.
//  <5,5>: the 70-channel stream layers in ONE slice, 80 x 80 (three 32-row slices x 96 columns would
//         be 1.85x the useful MFMA work), gradient tile staged and split once per chunk;
//  <2,18>: 32-channel slices x 288 columns for the 280-wide layers.
// LDS rows are padded to an odd multiple of 32 B: the 8 position rows one half-wave touches in a
// transposed read then start on 8 distinct multiples of 8 banks (conflict-free).  Lane group q4 takes
// positions {4q4..4q4+3, 16+4q4..}: the same k permutation for both operands, which a dot product
// does not see.
// ---------------------------------------------------------------------------------------------
template <int MB, int NB, int PL>
__global__ __launch_bounds__(256, 2) void wgrad4tap_x6n_kernel(WgradArgs a)
{
    constexpr int ROWA = 32 * (MB | 1), ROWG = 32 * (NB | 1);   // bytes per position row
    constexpr int A_PLANE = 34 * ROWA;                          // per (seg, plane)
    constexpr int A_BYTES = 2 * PL * A_PLANE;
    constexpr int NTERM = PL == 3 ? 6 : 3;
    const WgradScales sc = wgrad_scales<PL>(a);
    constexpr int G_PLANE = WG_KQ * ROWG;
    constexpr int FA = 4 * MB, FG = 4 * NB;                     // float4 per staged row
    constexpr int NA = (66 * FA + 255) / 256, NG = (WG_KQ * FG + 255) / 256;
    constexpr bool HOLD_G = NB <= 5;    // few columns: keep all gradient fragments, stream the activations
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *As = smem;
    char *Gs = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, t = tid >> 6;
    const int r16 = lane & 15, q4 = lane >> 4;
    int slice, split;
    if (!wgrad_block_map(a, slice, split)) return;
    const int ci0 = slice * 16 * MB;
    int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;

    // MMLF_WGRADN_CLAMP (round 5): staging items and channels past the tile are CLAMPED -- surplus threads load and store the last
    // item again, padding channels are copies of the tensor's last four -- instead of predicated and zero-filled: no divergent
    // branch per piece; the padding only feeds accumulator rows / columns the reduction never reads (as in the wide kernel).
    float4 ra[NA], rg[NG];
#define WN_GLOAD(c)                                                                                         \
    do {                                                                                                    \
        const long long Qc = (long long)(c) * WG_KQ;                                                        \
        _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                    \
            const int idx = MMLF_WGRADN_CLAMP ? min(tid + 256 * j, 66 * FA - 1) : tid + 256 * j;            \
            const int row = idx / FA, f = idx - row * FA;                                                   \
            const int seg = row >= 33, pix = row - 33 * seg;                                                \
            const int ch = MMLF_WGRADN_CLAMP ? min(ci0 + 4 * f, a.cs_in - 4) : ci0 + 4 * f;                 \
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
            MMLF_OOB(OOB_WG_IN, idx < 66 * FA && ch < a.cs_in && ((Qc + seg * a.P + pix) * a.cs_in + ch + 4) * 4ll > a.in_bytes); \
            if (MMLF_WGRADN_CLAMP || (idx < 66 * FA && ch < a.cs_in))                                       \
                v = *reinterpret_cast<const float4 *>(a.in + (size_t)(Qc + seg * a.P + pix) * a.cs_in + ch); \
            ra[j] = v;                                                                                      \
        }                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NG; ++j) {                                                    \
            const int idx = MMLF_WGRADN_CLAMP ? min(tid + 256 * j, WG_KQ * FG - 1) : tid + 256 * j;         \
            const int row = idx / FG, f = idx - row * FG;                                                   \
            const int gc = MMLF_WGRADN_CLAMP ? min(4 * f, a.cs_g - 4) : 4 * f;                              \
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
            MMLF_OOB(OOB_WG_G, idx < WG_KQ * FG && gc < a.cs_g && ((Qc + a.g_shift + row) * a.cs_g + gc + 4) * 4ll > a.g_bytes); \
            if (MMLF_WGRADN_CLAMP || (idx < WG_KQ * FG && 4 * f < a.cs_g))                                  \
                v = *reinterpret_cast<const float4 *>(a.g + (size_t)(Qc + a.g_shift + row) * a.cs_g + gc);  \
            rg[j] = v;                                                                                      \
        }                                                                                                   \
    } while (0)
#define WN_LSTORE()                                                                                         \
    do {                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                    \
            const int idx = MMLF_WGRADN_CLAMP ? min(tid + 256 * j, 66 * FA - 1) : tid + 256 * j;            \
            if (MMLF_WGRADN_CLAMP || idx < 66 * FA) {                                                       \
                const int row = idx / FA, f = idx - row * FA;                                               \
                const int seg = row >= 33, pix = row - 33 * seg;                                            \
                float4 v = ra[j];         /* ones row (bias gradient): patched here, not at load time, */ \
                const int ch = ci0 + 4 * f; /* so that the global loads issue back to back               */ \
                if (ch == a.cin) v.x = sc.inv_sa;   /* = 1 after scaling */                                 \
                if (ch + 1 == a.cin) v.y = sc.inv_sa;                                                       \
                if (ch + 2 == a.cin) v.z = sc.inv_sa;                                                       \
                if (ch + 3 == a.cin) v.w = sc.inv_sa;                                                       \
                split_store4_pl<PL>(v, st_sa, As + seg * PL * A_PLANE + pix * ROWA + 8 * f, A_PLANE);       \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NG; ++j) {                                                    \
            const int idx = MMLF_WGRADN_CLAMP ? min(tid + 256 * j, WG_KQ * FG - 1) : tid + 256 * j;         \
            if (MMLF_WGRADN_CLAMP || idx < WG_KQ * FG) {                                                    \
                const int row = idx / FG, f = idx - row * FG;                                               \
                split_store4_pl<PL>(rg[j], st_sg, Gs + row * ROWG + 8 * f, G_PLANE);                        \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
    // transposed-read geometry: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const char *a_lane = As + (t >> 1) * PL * A_PLANE + ((t & 1) + 4 * q4 + tq) * ROWA + 8 * tp;
    const char *g_lane = Gs + (4 * q4 + tq) * ROWG + 8 * tp;

    float st_sa = 1.f, st_sg = 1.f;          // operand scales of the chunk being staged
    if (c_begin < c_end) {
        WN_GLOAD(c_begin);
        for (int c = c_begin; c < c_end; ++c) {
            wgrad_chunk_scale<PL>(a, c, st_sa, st_sg);
            WN_LSTORE();
            __syncthreads();
            if (c + 1 < c_end) WN_GLOAD(c + 1);
            if constexpr (HOLD_G) {
                bf16x8 gf[NB][PL];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) gf[nb][pl] = tr_frag(g_lane + pl * G_PLANE + 32 * nb, 4 * ROWG);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    bf16x8 af[PL];
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) af[pl] = tr_frag(a_lane + pl * A_PLANE + 32 * mb, 4 * ROWA);
#pragma unroll
                    for (int term = 0; term < NTERM; ++term)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = mfma16_pl<PL>(af[term_a<PL>(term)], gf[nb][term_b<PL>(term)], acc[mb][nb]);
                }
            } else {
                bf16x8 af[MB][PL];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) af[mb][pl] = tr_frag(a_lane + pl * A_PLANE + 32 * mb, 4 * ROWA);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    bf16x8 gf[PL];
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) gf[pl] = tr_frag(g_lane + pl * G_PLANE + 32 * nb, 4 * ROWG);
#pragma unroll
                    for (int term = 0; term < NTERM; ++term)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
                            acc[mb][nb] = mfma16_pl<PL>(af[mb][term_a<PL>(term)], gf[term_b<PL>(term)], acc[mb][nb]);
                }
            }
            __syncthreads();
        }
    }
#undef WN_GLOAD
#undef WN_LSTORE
    constexpr int NP = 16 * NB;
    const int CIP = a.nslice * 16 * MB;
    float *pp = a.part + ((size_t)(split * 4 + t) * CIP + ci0) * NP;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mb + 4 * q4 + r;      // the ones row (bias gradient) carries no input scale
                const float un = (ci0 + row == a.cin ? 1.f : sc.inv_sa) * sc.inv_sg;
                MMLF_OOB(OOB_WG_PART, (long long)(pp - a.part) + (long long)row * NP + 16 * nb + r16 >= a.part_floats);
                pp[(size_t)row * NP + 16 * nb + r16] = acc[mb][nb][r] * un;
            }
}

// ---------------------------------------------------------------------------------------------
// wide-layer weight gradient (Cout <= 288): 512 threads = 8 waves, wave w = (tap w&3, column half w>>2).
// A workgroup owns a slice of 16*MB input channels x all 288 columns and a split of the positions; the
// LDS image is DOUBLE-buffered: while chunk c is multiplied, the registers holding chunk c+1 (loaded
// from global memory one iteration earlier) are split and stored into the other buffer between the
// MFMAs, and chunk c+2's loads are issued.  One barrier per chunk, no phase in which the matrix
// cores wait for staging.  155.9 KB of LDS: one workgroup per CU.
// ---------------------------------------------------------------------------------------------
template <int MB, int NBH, int PL>
__global__ __launch_bounds__(512, 2) void wgrad4tap_x6w_kernel(WgradArgs a)
{
    constexpr int NB = 2 * NBH;
    const WgradScales sc = wgrad_scales<PL>(a);
    constexpr int ROWA = 32 * (MB | 1), ROWG = 32 * (NB | 1);
    constexpr int A_PLANE = 34 * ROWA;
    constexpr int A_BYTES = 2 * PL * A_PLANE;
    constexpr int G_PLANE = WG_KQ * ROWG;
    constexpr int BUF_BYTES = A_BYTES + PL * G_PLANE;
    constexpr int FA = 4 * MB, FG = 4 * NB;
    constexpr int NA = (66 * FA + 511) / 512, NG = (WG_KQ * FG + 511) / 512;
    static_assert(2 * BUF_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = w & 3, h = w >> 2;
    const int r16 = lane & 15, q4 = lane >> 4;
    int slice, split;
    if (!wgrad_block_map(a, slice, split)) return;
    const int ci0 = slice * 16 * MB;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    f32x4 acc[MB][NBH];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;

    float4 ra[NA], rg[NG];
    // Input channels past cs_in and gradient columns past cs_g (the slices' and the 288 columns' padding) are staged as COPIES of
    // the tensor's last four channels (the clamped load) instead of zeros: they only feed accumulator rows / columns that
    // wgrad_reduce_kernel never reads (ci > Cin, co >= Cout), the copies are in-range values (no f16 overflow), and the ones row
    // (ci == Cin) is patched in WW_STORE_A either way.  Four selects per staging piece less (round 5;
    // -DMMLF_WGRAD_ZEROPAD=1 builds the zero-filling form: profiles/r05_kbench_wgrad_zeropad.log).

    // staging items are clamped to the last one instead of predicated: surplus threads load and store that
    // item again (same value), which keeps the loop free of branches.  Also of SCALAR ones: round 5 let the waves whose
    // items of a piece are all surplus (3 of 8 on the second activation piece, 4 of 8 on the fifth gradient piece at <3, 9>)
    // skip the piece -- less work, no 64 lanes storing to one LDS address -- and the launch took 8.53 ms instead of 8.17
    // (profiles/r05_kbench_wgrad_surplus.log): a branch per column block ends the basic block the MFMAs and the staging
    // instructions are interleaved in.  Nor do the surplus lanes' stores to ONE address cost anything: giving every such lane
    // an LDS slot of its own measured 8.04 ms against 8.02 (profiles/r05_kbench_wgrad_surplus.log) -- equal addresses merge.
    // Per-lane parts of every staging address, formed ONCE: the byte offset of a piece's item from its chunk's first position
    // (32 bits: the load is `global_load_dwordx4 v, v_offset, s[base]`, no vector arithmetic per load) and its byte offset
    // inside an LDS buffer.  Round 5: with the addresses recomputed from 64-bit terms in the loop and the padding channels
    // zero-filled by selects, a chunk had 136 vector instructions per wave next to its 81 MFMAs; the vector-issue port is what
    // this kernel runs out of (each one removed is ~0.01 ms per launch: profiles/r05_kbench_wgrad_zeropad.log).
    unsigned ga_off[NA], gg_off[NG];
    int la_off[NA], lg_off[NG], a_ch[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = min(tid + 512 * j, 66 * FA - 1);
        const int row = idx / FA, f = idx - row * FA;
        const int seg = row >= 33, pix = row - 33 * seg;
        a_ch[j] = ci0 + 4 * f;
        ga_off[j] = (unsigned)((seg * a.P + pix) * a.cs_in + min(ci0 + 4 * f, a.cs_in - 4)) * 4u;
        la_off[j] = seg * PL * A_PLANE + pix * ROWA + 8 * f;
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int idx = min(tid + 512 * j, WG_KQ * FG - 1);
        const int row = idx / FG, f = idx - row * FG;
        gg_off[j] = (unsigned)((a.g_shift + row) * a.cs_g + min(4 * f, a.cs_g - 4)) * 4u;
        lg_off[j] = A_BYTES + row * ROWG + 8 * f;
    }
    const char *const in_b = reinterpret_cast<const char *>(a.in), *const g_b = reinterpret_cast<const char *>(a.g);
#define WW_GLOAD_A(j, c)                                                                                    \
    do {                                                                                                    \
        const long long cb_ = (long long)(c) * WG_KQ * a.cs_in * 4;      /* wave-uniform */                   \
        MMLF_OOB(OOB_WG_IN, cb_ + ga_off[j] + 16 > a.in_bytes);                                             \
        /* buffer load on a per-chunk descriptor: base and what is left of the tensor in scalar registers, the lane's */ \
        /* 32-bit offset as is -- no vector address arithmetic, and the address unit range-checks the access          */ \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                               \
            const_cast<char *>(in_b + cb_), 0, mmlf_records_left(a.in_bytes, cb_), MMLF_BUF_FLAGS);         \
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_, ga_off[j], 0, 0)); \
        ra[j] = (MMLF_WGRAD_ZEROPAD && !(a_ch[j] < a.cs_in)) ? make_float4(0.f, 0.f, 0.f, 0.f) : v;         \
    } while (0)
#define WW_GLOAD_G(j, c)                                                                                    \
    do {                                                                                                    \
        const long long cb_ = (long long)(c) * WG_KQ * a.cs_g * 4;                                          \
        MMLF_OOB(OOB_WG_G, cb_ + gg_off[j] + 16 > a.g_bytes);                                               \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                               \
            const_cast<char *>(g_b + cb_), 0, mmlf_records_left(a.g_bytes, cb_), MMLF_BUF_FLAGS);           \
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_, gg_off[j], 0, 0)); \
        rg[j] = (MMLF_WGRAD_ZEROPAD && !(lg_off[j] - A_BYTES - (lg_off[j] - A_BYTES) / ROWG * ROWG < 2 * a.cs_g)) ? make_float4(0.f, 0.f, 0.f, 0.f) : v; \
    } while (0)
#define WW_GLOAD(c)                                                                                         \
    do {                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NA; ++j) WW_GLOAD_A(j, c);                                    \
        _Pragma("unroll") for (int j = 0; j < NG; ++j) WW_GLOAD_G(j, c);                                    \
    } while (0)
#define WW_STORE_A(j, dst)                                                                                  \
    do {                                                                                                    \
        float4 v = ra[j];                 /* ones row -> bias gradient */                                   \
        const int ch = a_ch[j];                                                                             \
        v.x = ch == a.cin ? sc.inv_sa : v.x;   /* = 1 after scaling */                                      \
        v.y = ch + 1 == a.cin ? sc.inv_sa : v.y;                                                            \
        v.z = ch + 2 == a.cin ? sc.inv_sa : v.z;                                                            \
        v.w = ch + 3 == a.cin ? sc.inv_sa : v.w;                                                            \
        split_store4_pl<PL>(v, st_sa, (dst) + la_off[j], A_PLANE);                                          \
    } while (0)
#define WW_STORE_G(j, dst)                                                                                  \
    do {                                                                                                    \
        if (MMLF_ABL_WGRAD_STAGE == 1 && PL == 2) {      /* ablation: the same bytes, no split */            \
            char *p0_ = (dst) + lg_off[j];                                                                  \
            *reinterpret_cast<uint2 *>(p0_) = make_uint2(__float_as_uint(rg[j].x), __float_as_uint(rg[j].y)); \
            *reinterpret_cast<uint2 *>(p0_ + G_PLANE) = make_uint2(__float_as_uint(rg[j].z), __float_as_uint(rg[j].w)); \
        } else {                                                                                            \
            split_store4_pl<PL>(rg[j], st_sg, (dst) + lg_off[j], G_PLANE);                                  \
        }                                                                                                   \
    } while (0)

    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int a_off = (t >> 1) * PL * A_PLANE + ((t & 1) + 4 * q4 + tq) * ROWA + 8 * tp;
    const int g_off = A_BYTES + (4 * q4 + tq) * ROWG + 8 * tp + 32 * NBH * h;

    float st_sa = 1.f, st_sg = 1.f;          // operand scales of the chunk being staged
    if (c_begin < c_end) {
        WW_GLOAD(c_begin);
        wgrad_chunk_scale<PL>(a, c_begin, st_sa, st_sg);
#pragma unroll
        for (int j = 0; j < NA; ++j) WW_STORE_A(j, smem);
#pragma unroll
        for (int j = 0; j < NG; ++j) WW_STORE_G(j, smem);
        if (c_begin + 1 < c_end) WW_GLOAD(c_begin + 1);
        __syncthreads();
        int buf = 0;
        // EARLY (as in the conv kernel): the chunk's barrier stands two column blocks before its end -- every staging
        // store of the chunk is out by then (one piece per column block, NA + NG <= NBH - 2) and every fragment read
        // requested -- and behind it the wave asks for the NEXT chunk's first gradient fragments (the rotating slots run
        // on, NBH % 3 == 0) and its first two activation row blocks, which arrive under the last 2 x 3 x MB MFMAs.
        constexpr bool EARLY = MMLF_WGRAD_EARLY && PL == 2 && NBH % 3 == 0 && NA + NG <= NBH - 2 && MB >= 2;
        bf16x8 af[MB][PL], gfr[3][PL], afp[2][PL];
        if constexpr (EARLY) {
            const char *cur0 = smem;
#pragma unroll
            for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) gfr[g0][pl] = tr_frag(cur0 + g_off + pl * G_PLANE + 32 * g0, 4 * ROWG);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) af[mb][pl] = tr_frag(cur0 + a_off + pl * A_PLANE + 32 * mb, 4 * ROWA);
        }
        // One chunk: gradient fragments are read two column blocks ahead (three register sets); with STAGE the
        // registers holding chunk c+1 are split and stored into the other buffer, one piece per column
        // block, branch-free so that the stores interleave with the MFMAs -- and the register a piece leaves is
        // loaded with the same piece of chunk c+2 right away: a whole chunk between a global load and its use
        // (loaded in bulk at the end of the chunk, every load was waited for at the next chunk's first column blocks).
#define WW_TERM(gf, pa, pb)                                                                                  \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                        \
        acc[mb][nb] = mfma16_pl<PL>(af[mb][pa], gf[pb], acc[mb][nb])
#define WW_CHUNK(STAGE, RELOAD)                                                                              \
    do {                                                                                                     \
        const char *cur = smem + buf * BUF_BYTES;                                                            \
        char *nxt = smem + (buf ^ 1) * BUF_BYTES;                                                            \
        if (!EARLY) {                                                                                        \
            _Pragma("unroll") for (int g0 = 0; g0 < 2; ++g0)                                                 \
                _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                             \
                    gfr[g0][pl] = tr_frag(cur + g_off + pl * G_PLANE + 32 * g0, 4 * ROWG);                   \
        }                                                                                                    \
        _Pragma("unroll") for (int mb = EARLY ? 2 : 0; mb < MB; ++mb)                                        \
            _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                                 \
                af[mb][pl] = tr_frag(cur + a_off + pl * A_PLANE + 32 * mb, 4 * ROWA);                        \
        _Pragma("unroll") for (int nb = 0; nb < NBH; ++nb) {                                                 \
            if (EARLY && (STAGE) && nb == NBH - 2) {                                                         \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* my staging stores and fragment reads */ \
                __syncthreads();                                                                             \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                             \
                    _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                         \
                        afp[mb][pl] = tr_frag(nxt + a_off + pl * A_PLANE + 32 * mb, 4 * ROWA);               \
            }                                                                                                \
            if (nb + 2 < NBH) {                                                                              \
                _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                             \
                    gfr[(nb + 2) % 3][pl] = tr_frag(cur + g_off + pl * G_PLANE + 32 * (nb + 2), 4 * ROWG);   \
            } else if (EARLY && (STAGE)) {                                                                   \
                _Pragma("unroll") for (int pl = 0; pl < PL; ++pl)                                             \
                    gfr[(nb + 2) % 3][pl] = tr_frag(nxt + g_off + pl * G_PLANE + 32 * (nb + 2 - NBH), 4 * ROWG); \
            }                                                                                                \
            if (STAGE) {                                                                                     \
                if (nb < NA) {                                                                               \
                    if (MMLF_ABL_WGRAD_STAGE < 3) {                                                          \
                        WW_STORE_A(nb < NA ? nb : 0, nxt);                                                   \
                        if (RELOAD) WW_GLOAD_A(nb < NA ? nb : 0, c + 2);                                     \
                    }                                                                                        \
                } else if (nb - NA < NG) {                                                                   \
                    if (MMLF_ABL_WGRAD_STAGE < 2) {                                                          \
                        WW_STORE_G(nb - NA < NG ? nb - NA : 0, nxt);                                         \
                        if (RELOAD) WW_GLOAD_G(nb - NA < NG ? nb - NA : 0, c + 2);                           \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
            _Pragma("unroll") for (int term = (PL == 3 ? 0 : 3 - MMLF_ABL_TERMS); term < (PL == 3 ? 6 : 3); ++term) \
                WW_TERM(gfr[nb % 3], term_a<PL>(term), term_b<PL>(term));                                    \
        }                                                                                                    \
        if (EARLY && (STAGE)) {                                                                              \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                 \
                _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) af[mb][pl] = afp[mb][pl];                   \
        }                                                                                                    \
    } while (0)
        static_assert(NA + NG <= NBH, "one staging piece per column block");
        for (int c = c_begin; c + 1 < c_end; ++c) {
            wgrad_chunk_scale<PL>(a, c + 1, st_sa, st_sg);
            // (the scalar branch around each of the seven reloads stays: with unconditional loads -- the split's last chunk loaded
            // once more, never staged -- the chunk is one basic block and the compiler's own order of it is 1.7 % SLOWER, 8.17
            // against 8.04 ms, profiles/r05_kbench_wgrad_zeropad.log; the branches are what pins the interleaving written here)
            const bool reload = c + 2 < c_end;
            WW_CHUNK(true, reload);
            if constexpr (!EARLY) __syncthreads();
            buf ^= 1;
        }
        {
            const int c = c_end - 1;
            (void)c;
            WW_CHUNK(false, false);          // last chunk: nothing left to stage
        }
#undef WW_CHUNK
#undef WW_TERM
    }
#undef WW_GLOAD
#undef WW_GLOAD_A
#undef WW_GLOAD_G
#undef WW_STORE_A
#undef WW_STORE_G
    constexpr int NP = 16 * NB;
    const int CIP = a.nslice * 16 * MB;
    float *pp = a.part + ((size_t)(split * 4 + t) * CIP + ci0) * NP + 16 * NBH * h;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mb + 4 * q4 + r;      // the ones row (bias gradient) carries no input scale
                const float un = (ci0 + row == a.cin ? 1.f : sc.inv_sa) * sc.inv_sg;
                MMLF_OOB(OOB_WG_PART, (long long)(pp - a.part) + (long long)row * NP + 16 * nb + r16 >= a.part_floats);
                pp[(size_t)row * NP + 16 * nb + r16] = acc[mb][nb][r] * un;
            }
}

// sum the position splits and scatter to the OIHW master gradient (+ bias gradient).  One thread per output element:
// at 64 patches per GPU this launch is latency-bound and wants every CU full of waves -- a float4-per-thread form with a
// quarter of the threads ran 2.5x longer (130 vs 50 us at 280 -> 280, measured in the step).
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, float *__restrict__ gb,
                                     int Cin, int Cout, int CIP, int NP, int nsplit, int variant, int accumulate)
{
    const int total = 4 * (Cin + 1) * Cout;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = idx % Cout;
    int r = idx / Cout;
    const int ci = r % (Cin + 1);
    const int t = r / (Cin + 1);
    if (ci == Cin && (t != 0 || gb == nullptr)) return;
    const float *pp = part + ((size_t)t * CIP + ci) * NP + co;
    const size_t stride = (size_t)4 * CIP * NP;
    constexpr int U = 16;
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.0;
    int sp = 0;
    for (; sp + U <= nsplit; sp += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = pp[(size_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += v[u];
    }
    for (; sp < nsplit; ++sp) acc[0] += pp[(size_t)sp * stride];
#pragma unroll
    for (int h = U / 2; h > 0; h >>= 1)
#pragma unroll
        for (int u = 0; u < h; ++u) acc[u] += acc[u + h];
    const double s = acc[0];
    if (ci == Cin) {
        gb[co] = accumulate ? gb[co] + (float)s : (float)s;
    } else {
        const size_t o = ((size_t)co * Cin + ci) * 4 + master_tap(t, variant);
        gw[o] = accumulate ? gw[o] + (float)s : (float)s;
    }
}

// the same for MANY partials of few elements (the thin weight gradient: 4096 wave-private partials of 4 x 281 x 2 sums):
// one WAVE per output element, lanes stride over the partials, fixed-order butterfly at the end (one thread per element
// walked 4096 partials 9 KB apart by itself: 1.95 ms per step for 2 248 sums)
__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(const float *__restrict__ part, float *__restrict__ gw,
                                                                float *__restrict__ gb, int Cin, int Cout, int CIP, int NP,
                                                                int nsplit, int variant, int accumulate)
{
    const int total = 4 * (Cin + 1) * Cout;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= total) return;
    const int co = idx % Cout;
    int r = idx / Cout;
    const int ci = r % (Cin + 1);
    const int t = r / (Cin + 1);
    if (ci == Cin && (t != 0 || gb == nullptr)) return;
    const float *pp = part + ((size_t)t * CIP + ci) * NP + co;
    const size_t stride = (size_t)4 * CIP * NP;
    double acc = 0.0;
    for (int sp = lane; sp < nsplit; sp += 64) acc += pp[(size_t)sp * stride];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane) return;
    if (ci == Cin) {
        gb[co] = accumulate ? gb[co] + (float)acc : (float)acc;
    } else {
        const size_t o = ((size_t)co * Cin + ci) * 4 + master_tap(t, variant);
        gw[o] = accumulate ? gw[o] + (float)acc : (float)acc;
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static int wgrad_nsplit(int nslice)
{
    int n = 512 / nslice;       // two 256-thread blocks per CU on 256 CUs
    n = n / 8 * 8;
    if (n < 8) n = 8;
    return n;
}

// split-arithmetic kernel configuration for a layer: MB 16-row blocks of input channels (+ ones row) per
// slice, NB 16-column blocks of output channels, and the (slice x position-split) grid
struct Wgrad16Cfg { int mb, nb, nslice, nsplit; };
// nchunks: 32-position chunks of the launch, or -1 for the largest layout (workspace sizing)
static inline bool wgrad16_cfg(int Cin, int Cout, long long nchunks, Wgrad16Cfg *c)
{
    if (Cout <= 0 || Cout > 288) return false;
    c->nb = Cout <= 32 ? 2 : Cout <= 80 ? 5 : Cout <= 128 ? 8 : 18;
    if (c->nb == 18) {           // wgrad4tap_x6w_kernel: one 512-thread workgroup per CU, three even rounds
        c->mb = 3;
        c->nslice = (Cin + 1 + 47) / 48;
        // Small batches: ONE round on 256 CUs (42 splits = 252 workgroups at six slices; a third of the partial sums to
        // write and reduce: -2...-4 % per launch at 64 patches, +0.9 % per 64-patch step).  From ~150 patches on, three
        // even rounds of 256 (128 splits) are 0.8 % faster inside the step (profiles/r04_wgrad_nsplit.log).
        // (the workspace is sized by the LARGER of the two counts at 256 CUs: a part with fewer CUs only lowers one_round)
        const int cus = nchunks < 0 ? 256 : (device_cus() < 256 ? device_cus() : 256);
        const int one_round = cus / c->nslice, three_rounds = (768 / c->nslice + 7) / 8 * 8;
        c->nsplit = nchunks < 0 ? (one_round > three_rounds ? one_round : three_rounds)
                                : (nchunks >= 42 * 1024 ? three_rounds : one_round);
        static const int forced = [] { const char *e = getenv("MMLF_WGRAD_NSPLIT"); return e ? atoi(e) : 0; }();
        if (forced > 0 && nchunks >= 0)                          // A/B switch (tools/ab_env.sh), inside the sized workspace
            c->nsplit = forced <= (three_rounds > one_round ? three_rounds : one_round) ? forced : c->nsplit;
        if (c->nsplit < 8) c->nsplit = 8;
    } else {                     // wgrad4tap_x6n_kernel: two 256-thread workgroups per CU
        c->mb = (c->nb <= 5 && Cin + 1 > 32 && Cin + 1 <= 80) ? 5 : 2;
        // 128 columns (the 108-class DPP head): 48-row slices from 128 input channels on -- six slices of 281 rows instead of
        // nine, the gradient tile staged six times: 4.88 -> 4.47 ms at 280 -> 108 (96-row slices spill: 18.7 ms)
        if (c->nb == 8 && Cin + 1 >= 128) c->mb = 3;
        c->nslice = (Cin + 1 + 16 * c->mb - 1) / (16 * c->mb);
        c->nsplit = wgrad_nsplit(c->nslice);
    }
    return true;
}

// partial sums of the position splits (largest of the kernels' layouts)
static int64_t wgrad_partial_floats(int Cin, int Cout)
{
    const int nt = pick_nt(Cout);
    if (nt < 0) return -1;
    const int nslice = (Cin + 1 + 31) / 32;
    int64_t n = (int64_t)wgrad_nsplit(nslice) * 4 * (nslice * 32) * (nt * 32);          // exact-f32 kernel
    Wgrad16Cfg c;
    if (wgrad16_cfg(Cin, Cout, -1, &c)) {                                                // split kernel
        const int64_t m = (int64_t)c.nsplit * 4 * (c.nslice * 16 * c.mb) * (16 * c.nb);
        if (m > n) n = m;
    }
    return (n + 3) / 4 * 4;
}

extern "C" int64_t mmlf_wgrad_workspace_floats(int Cin, int Cout, int B, int H, int W)
{
    const int64_t n = wgrad_partial_floats(Cin, Cout);
    if (n < 0 || B <= 0 || H <= 0 || W <= 0) return -1;
    return n + 2 * (make_grid(B, H, W).NQpad / WG_KQ) + 4;  // + the f16 split's per-chunk operand scales and the two tensor scales
}

template <int NT>
static int launch_wgrad(const WgradArgs &a, hipStream_t st)
{
    constexpr size_t lds = (2 * 33 * 32 + WG_KQ * NT * 32) * sizeof(float);
    hipLaunchKernelGGL(wgrad4tap_kernel<NT>, dim3(wgrad_grid_blocks(a.nslice, a.nsplit)), dim3(256), lds, st, a);
    return mmlf_launch_status("mmlf_conv2x2_wgrad");
}

template <int MB, int NB, int PL>
static int launch_wgrad16(const WgradArgs &a, hipStream_t st)
{
    constexpr size_t lds = 2 * PL * 34 * 32 * (MB | 1) + PL * WG_KQ * 32 * (NB | 1);
    static PerDeviceOnce attr_once;
    if (attr_once.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(wgrad4tap_x6n_kernel<MB, NB, PL>), lds, "mmlf_conv2x2_wgrad_split"); }))
        return 1;
    hipLaunchKernelGGL((wgrad4tap_x6n_kernel<MB, NB, PL>), dim3(wgrad_grid_blocks(a.nslice, a.nsplit)), dim3(256), lds, st, a);
    return mmlf_launch_status("mmlf_conv2x2_wgrad_split");
}

template <int PL>
static int launch_wgrad_wide(const WgradArgs &a, hipStream_t st)
{
    constexpr size_t lds = 2 * (2 * PL * 34 * 32 * (3 | 1) + PL * WG_KQ * 32 * (18 | 1));
    static PerDeviceOnce attr_once;
    if (attr_once.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(wgrad4tap_x6w_kernel<3, 9, PL>), lds, "mmlf_conv2x2_wgrad_split(wide)"); }))
        return 1;
    hipLaunchKernelGGL((wgrad4tap_x6w_kernel<3, 9, PL>), dim3(wgrad_grid_blocks(a.nslice, a.nsplit)), dim3(512), lds, st, a);
    return mmlf_launch_status("mmlf_conv2x2_wgrad_split");
}

template <int PL>
static int launch_wgrad_split(const Wgrad16Cfg &c, const WgradArgs &a, hipStream_t st)
{
    switch (10 * c.mb + c.nb) {
    case 22: return launch_wgrad16<2, 2, PL>(a, st);
    case 52: return launch_wgrad16<5, 2, PL>(a, st);
    case 25: return launch_wgrad16<2, 5, PL>(a, st);
    case 55: return launch_wgrad16<5, 5, PL>(a, st);
    case 28: return launch_wgrad16<2, 8, PL>(a, st);
    case 38: return launch_wgrad16<3, 8, PL>(a, st);
    default: return launch_wgrad_wide<PL>(a, st);
    }
}

// planes: 0 = exact-f32 MFMA, 3 = bf16 split, 2 = f16 split (needs in_amax / g_amax)
static int wgrad_impl(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout, int g_shift,
                      float *gw, float *gb, int variant, int accumulate, float *workspace, int B, int H, int W,
                      void *stream, int planes, const float *in_amax = nullptr, const float *g_amax = nullptr);

extern "C" int mmlf_conv2x2_wgrad(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                                  int g_shift, float *gw, float *gb, int variant, int accumulate,
                                  float *workspace, int B, int H, int W, void *stream)
{
    return wgrad_impl(in, cs_in, Cin, g, cs_g, Cout, g_shift, gw, gb, variant, accumulate, workspace, B, H, W,
                      stream, 0);
}

extern "C" int mmlf_conv2x2_wgrad_split(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                                        int g_shift, float *gw, float *gb, int variant, int accumulate,
                                        float *workspace, int B, int H, int W, void *stream)
{
    return wgrad_impl(in, cs_in, Cin, g, cs_g, Cout, g_shift, gw, gb, variant, accumulate, workspace, B, H, W,
                      stream, 3);
}

extern "C" int mmlf_conv2x2_wgrad_h2(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                                     int g_shift, float *gw, float *gb, int variant, int accumulate,
                                     float *workspace, int B, int H, int W, const float *in_amax,
                                     const float *g_amax, void *stream)
{
    MMLF_CHECK_ARG(in_amax && g_amax, "mmlf_conv2x2_wgrad_h2: the f16 split needs max |in| and max |g|");
    return wgrad_impl(in, cs_in, Cin, g, cs_g, Cout, g_shift, gw, gb, variant, accumulate, workspace, B, H, W,
                      stream, 2, in_amax, g_amax);
}

static int wgrad_impl(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout, int g_shift,
                      float *gw, float *gb, int variant, int accumulate, float *workspace, int B, int H, int W,
                      void *stream, int planes, const float *in_amax, const float *g_amax)
{
    MMLF_CHECK_ARG(in && g && gw && workspace, "mmlf_conv2x2_wgrad: null pointer");
    MMLF_CHECK_ARG(cs_in % 4 == 0 && cs_g % 4 == 0, "mmlf_conv2x2_wgrad: strides must be multiples of 4");
    MMLF_CHECK_ARG(Cin > 0 && Cin <= cs_in && Cout > 0 && Cout <= cs_g, "mmlf_conv2x2_wgrad: channels vs strides");
    const int nt = pick_nt(Cout);
    MMLF_CHECK_ARG(nt > 0, "mmlf_conv2x2_wgrad: Cout=%d not supported", Cout);
    MMLF_CHECK_ARG(variant >= 0 && variant <= 2, "mmlf_conv2x2_wgrad: bad variant");
    Grid gr = make_grid(B, H, W);
    MMLF_CHECK_ARG(g_shift >= 0 && g_shift <= gr.P + 1, "mmlf_conv2x2_wgrad: g_shift=%d", g_shift);
    WgradArgs a = {};
    a.in = in; a.g = g; a.part = workspace; a.NQpad = gr.NQpad;
    a.cs_in = cs_in; a.cin = Cin; a.cs_g = cs_g; a.g_shift = g_shift; a.P = gr.P;
    a.in_amax = in_amax; a.g_amax = g_amax; a.chunk_scales = nullptr;
    a.in_bytes = grid_alloc_positions(gr) * cs_in * 4; a.g_bytes = grid_alloc_positions(gr) * cs_g * 4;
    a.part_floats = wgrad_partial_floats(Cin, Cout);
    a.nslice = (Cin + 1 + 31) / 32;   // +1: the ones row that yields the bias gradient
    a.nsplit = wgrad_nsplit(a.nslice);
    a.nchunks = (int)(gr.NQpad / WG_KQ);
    a.chunks_per_split = (a.nchunks + a.nsplit - 1) / a.nsplit;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    Wgrad16Cfg c;
    if (planes == 2) {          // per-chunk operand scales, behind the partial sums in the workspace
        MMLF_CHECK_ARG(gr.NQpad + 2 * gr.P + 64 < (1ll << 31), "mmlf_conv2x2_wgrad_h2: batch x image too large");
        ChunkScaleArgs ca;
        ca.in_amax = in_amax; ca.g_amax = g_amax; ca.out = workspace + wgrad_partial_floats(Cin, Cout);
        ca.NQ = gr.NQ; ca.nchunks = a.nchunks; ca.P = gr.P; ca.g_shift = g_shift;
        ca.nrows = (int)amax_rows(gr);
        ca.divP = make_magic((unsigned)gr.P);
        hipLaunchKernelGGL(wgrad_chunk_scales_kernel, dim3((a.nchunks + 255) / 256), dim3(256), 0, st, ca);
        a.chunk_scales = ca.out;
    }
    if (planes && wgrad16_cfg(Cin, Cout, a.nchunks, &c)) {
        a.nslice = c.nslice;
        a.nsplit = c.nsplit;
        a.chunks_per_split = (a.nchunks + a.nsplit - 1) / a.nsplit;
        rc = planes == 3 ? launch_wgrad_split<3>(c, a, st) : launch_wgrad_split<2>(c, a, st);
        if (rc) return rc;
        const int total = 4 * (Cin + 1) * Cout;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, gw, gb, Cin,
                           Cout, a.nslice * 16 * c.mb, 16 * c.nb, a.nsplit, variant, accumulate);
        return mmlf_launch_status("mmlf_conv2x2_wgrad(reduce)");
    }
    switch (nt) {
    case 1: rc = launch_wgrad<1>(a, st); break;
    case 3: rc = launch_wgrad<3>(a, st); break;
    case 4: rc = launch_wgrad<4>(a, st); break;
    default: rc = launch_wgrad<9>(a, st); break;
    }
    if (rc) return rc;
    const int total = 4 * (Cin + 1) * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, gw, gb, Cin,
                       Cout, a.nslice * 32, nt * 32, a.nsplit, variant, accumulate);
    return mmlf_launch_status("mmlf_conv2x2_wgrad(reduce)");
}

// weight + bias gradient of a thin convolution: wave-private sums over its positions, reduced by wgrad_reduce_kernel.
// Four positions per iteration with every load issued up front (the loop is latency-bound otherwise); the wave index
// is made scalar so that the gradients -- the same address for all lanes -- come in through the scalar cache.
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ThinArgs a)
{
    const int lane = threadIdx.x & 63;
    const long long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const long long nwaves = gridDim.x * 4ll;
    float acc[2][4][THIN_MAXN][4];                        // [half][tap][o][channel of the float4]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < THIN_MAXN; ++o)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[h][t][o][k] = 0.f;
    float gsum[THIN_MAXN] = {0.f, 0.f};
    const bool first = 4 * lane < a.cs_in, second = 256 + 4 * lane < a.cs_in;
    const int offs[4] = {0, 1, a.P, a.P + 1};
    constexpr int U = 4;
    // a wave owns runs of U consecutive positions: p0 = U * (wave + k * nwaves)
    for (long long p0 = U * wave; p0 < a.npos; p0 += U * nwaves) {
        float4 x0[U], x1[U];
        float gt[U][4][THIN_MAXN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + u < a.npos ? p0 + u : a.npos - 1;      // (clamped: its gradients are zeroed below)
            const float *row = a.in + (size_t)p * a.cs_in;
            x0[u] = first ? *reinterpret_cast<const float4 *>(row + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
            x1[u] = second ? *reinterpret_cast<const float4 *>(row + 256 + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // in[p] is tap t's operand of output position p - off_t, whose gradient sits at g[p - off_t + g_shift]
                const long long q = p0 + u - offs[t];
                const bool ok = p0 + u < a.npos && q >= 0 && q < a.NQ;
                // unconditional loads from a clamped position (a select per load would serialise them behind branches);
                // cs_g >= 2: both columns exist, column 1 is a zero pad channel when N == 1
                const float2 gv = *reinterpret_cast<const float2 *>(a.g + (size_t)((ok ? q : 0) + a.g_shift) * a.cs_g);
                gt[u][t][0] = ok ? gv.x : 0.f;
                gt[u][t][1] = ok ? gv.y : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float xs[2][4] = {{x0[u].x, x0[u].y, x0[u].z, x0[u].w}, {x1[u].x, x1[u].y, x1[u].z, x1[u].w}};
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int o = 0; o < THIN_MAXN; ++o)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[h][t][o][k] = fmaf(xs[h][k], gt[u][t][o], acc[h][t][o][k]);
#pragma unroll
            for (int o = 0; o < THIN_MAXN; ++o) gsum[o] += gt[u][0][o];    // tap 0: every gradient position once
        }
    }
    float *pp = a.wpart + (size_t)wave * 4 * a.CIP * THIN_MAXN;           // [tap][ci][o]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 256 * h + 4 * lane + k;
                if (c < a.C)
#pragma unroll
                    for (int o = 0; o < THIN_MAXN; ++o) pp[((size_t)t * a.CIP + c) * THIN_MAXN + o] = acc[h][t][o][k];
            }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < THIN_MAXN; ++o) pp[((size_t)t * a.CIP + a.C) * THIN_MAXN + o] = t == 0 ? gsum[o] : 0.f;   // bias row
    }
}

static int thin_wgrad_waves() { return 4 * 4 * device_cus(); }

extern "C" int64_t mmlf_conv2x2_wgrad_thin_workspace_floats(int Cin)
{
    if (Cin <= 0) return -1;
    return (int64_t)thin_wgrad_waves() * 4 * (Cin + 1) * THIN_MAXN;
}

extern "C" int mmlf_conv2x2_wgrad_thin(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout, int g_shift,
                                       float *gw_oihw, float *gb, int variant, int accumulate, float *workspace, int B,
                                       int H, int W, void *stream)
{
    MMLF_CHECK_ARG(in && g && gw_oihw && workspace, "mmlf_conv2x2_wgrad_thin: null pointer");
    MMLF_CHECK_ARG(Cout >= 1 && Cout <= THIN_MAXN && Cout <= cs_g && Cin >= 1 && Cin <= cs_in && cs_in % 4 == 0 && cs_in <= 512,
                   "mmlf_conv2x2_wgrad_thin: Cin=%d Cout=%d cs_in=%d", Cin, Cout, cs_in);
    MMLF_CHECK_ARG(B > 0 && H > 0 && W > 0 && variant >= 0 && variant <= 2, "mmlf_conv2x2_wgrad_thin: bad shape");
    const Grid gr = make_grid(B, H, W);
    MMLF_CHECK_ARG(g_shift >= 0 && g_shift <= gr.P + 1, "mmlf_conv2x2_wgrad_thin: g_shift=%d", g_shift);
    ThinArgs a = {};
    a.in = in; a.g = g; a.wpart = workspace; a.NQ = gr.NQ; a.npos = gr.NQ + gr.P + 2;
    a.cs_in = cs_in; a.C = Cin; a.N = Cout; a.cs_g = cs_g; a.g_shift = g_shift; a.P = gr.P; a.CIP = Cin + 1;
    hipStream_t st = (hipStream_t)stream;
    const int nwaves = thin_wgrad_waves();
    hipLaunchKernelGGL(thin_wgrad_kernel, dim3(nwaves / 4), dim3(256), 0, st, a);
    const int total = 4 * (Cin + 1) * Cout;
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3((total + 3) / 4), dim3(256), 0, st, workspace, gw_oihw, gb, Cin, Cout,
                       Cin + 1, THIN_MAXN, nwaves, variant, accumulate);
    return mmlf_launch_status("mmlf_conv2x2_wgrad_thin");
}

extern "C" int mmlf_audit_wgrad_h2(int cs_in, int Cin, int cs_g, int Cout, int g_shift, int B, int H, int W,
                                   int64_t *ends /* [MMLF_AUDIT_WGRAD_N] */)
{
    Wgrad16Cfg c;
    MMLF_CHECK_ARG(B > 0 && H > 0 && W > 0 && ends && Cin > 0 && Cout > 0, "mmlf_audit_wgrad_h2: bad argument");
    const Grid g = make_grid(B, H, W);
    const long long nchunks = g.NQpad / WG_KQ;
    MMLF_CHECK_ARG(wgrad16_cfg(Cin, Cout, nchunks, &c), "mmlf_audit_wgrad_h2: Cout=%d", Cout);
    const long long last_c = (nchunks - 1) * WG_KQ;
    ends[0] = (last_c + g.P + 32 + 1) * cs_in * 4;                            // in: segment 1, row 32
    ends[1] = (last_c + g_shift + 31 + 1) * cs_g * 4;                         // g
    ends[2] = (int64_t)Cout * Cin * 4 * 4;                                    // gw (OIHW)
    ends[3] = (int64_t)Cout * 4;                                              // gb
    ends[4] = ((int64_t)c.nsplit * 4 * (c.nslice * 16 * c.mb) * (16 * c.nb)   // workspace: partial sums of this launch ...
               > wgrad_partial_floats(Cin, Cout) ? -1                           // (must fit the sized region: else the scales overlap them)
               : wgrad_partial_floats(Cin, Cout) + 2 * nchunks + 2) * 4;        // ... then the per-chunk and the two tensor scales
    ends[5] = (MMLF_AMAX_HEAD + amax_rows(g)) * 4;                            // in_amax (the scale kernel clamps rows to amax_rows - 1)
    ends[6] = ends[5];                                                        // g_amax
    return 0;
}
