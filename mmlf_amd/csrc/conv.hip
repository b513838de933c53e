// conv.hip -- MFMA implicit-GEMM kernels for the k=2 convolutions of the EPI-stack CNN, forward and data gradient, on the
// padded NHWC grid (the weight + bias gradient: wgrad.hip; shared device helpers: conv_device.h).  gfx950 only.
// Arithmetic paths with the same interfaces: exact-f32 MFMA (conv4tap_kernel) and the split paths at f32 accuracy
// (conv4tap_x6s_kernel, conv4tap_rs_kernel: bf16x6 / f16x3).
//
// Arithmetic replaced: nn.Conv2d(k=2, pad 1|0) forward / backward as used by
// reference mmlf/model/feed_forward.py:123,125 (autograd via mmlf/train/cli.py:257).
//
// All three are 4-tap correlations over the flat grid position q with tap offsets
// {0, 1, P, P+1} (include/mmlf_hip.h).  v_mfma_f32_32x32x2_f32 is an exact f32 fma chain, so the
// exact-f32 path is float32-exact up to summation order.
#include "conv_device.h"


#ifdef MMLF_BOUNDS_DEBUG
// (every translation unit counts into its own device array, common.h: this one's plus the weight-gradient unit's)
extern "C" int mmlf_debug_oob_counts(unsigned long long *host8, int reset)
{
    unsigned long long other[8];
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_mmlf_oob), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_mmlf_oob), z, sizeof(z)) != hipSuccess) return 1;
    }
    if (mmlf_oob_counts_wgrad(other, reset)) return 1;
    for (int k = 0; k < 8; ++k) host8[k] += other[k];
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------
// filter packing: OIHW master -> [chunk][tap][kh][NP][4] with k = 8*chunk + 4*kh + s
// ---------------------------------------------------------------------------------------------
__global__ void pack_filter_kernel(const float *__restrict__ w, float *__restrict__ out, int Cout, int Cin,
                                   int variant, int dgrad, int nchunk, int NP)
{
    const long long total = (long long)nchunk * 4 * 2 * NP * 4;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int s = idx & 3;
        long long r = idx >> 2;
        int n = r % NP; r /= NP;
        int kh = r & 1; r >>= 1;
        int t = r & 3; r >>= 2;
        int c = (int)r;
        int k = 8 * c + 4 * kh + s;
        int ci, co, tsrc;
        if (!dgrad) { ci = k; co = n; tsrc = t; }
        else { co = k; ci = n; tsrc = 3 - t; }
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * 4 + master_tap(tsrc, variant)];
        out[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const float *in;
    const float *wp;
    const float *bias;
    float *out;
    const float *ref;
    long long NQ;
    int cs_in, nchunk, cs_out, n_store, n_true, out_shift, vh, vw, P, G, relu, cs_ref;
    int a_pieces, seg_slot, seg_delta;   // split kernel: A window geometry (see conv4tap_x6s_kernel): 32-position pieces,
    int a_per_seg, a_tail;               // ... pieces per segment, positions the last piece of a segment has to fetch
    int nw;                              // waves per workgroup the launch uses (8, or 16: 512-position tiles)
    int a_blocked;                       // round-6 proxy (MMLF_PROXY_BLOCKED_A=1, tools/kbench_blocked.py): `in` is a CHUNK-BLOCKED
                                         // copy [tile][chunk of 8 channels][position][8 channels] of the NHWC tensor: one chunk of
                                         // a window is one contiguous run of whole 128-byte lines, each fetched once
    const float *in_amax;                // f16 split: amax array of `in` (common.h): per-wave power-of-two operand scales
    const float *w_unscale;              // f16 split: 1 / (power-of-two scale of packed column n), [NP]
    float *out_amax;                     // optional: amax array of `out` (tensor and grid-row maxima, atomic max)
    Magic divP, divR;                    // q / P and row / R without integer division
    int R;
    double *bn_partial;                  // optional (f16 split): per-workgroup sums of out and out^2 per channel,
                                         // [block][2][n_true] doubles, the input of the BatchNorm finalize
    unsigned *relu_mask_out;             // optional: (out > 0) as bits, [tile][wave][8 rows][64 lanes] words, bit = column block
    const unsigned *relu_mask_in;        // optional: such a mask instead of relu_ref (same launch geometry and N)
    // bytes the caller's buffers hold behind `out` / `ref` by the ABI's contract (mmlf_grid_alloc_positions): the epilogue's
    // buffer descriptors carry what is left of them as num_records, so that a store or load past the allocation would be
    // DROPPED by the address unit instead of faulting (round 5; until then the range check was switched off).  None is
    // expected: tests/test_bounds_audit.py derives every launch's extents, and a -DMMLF_BOUNDS_DEBUG build counts them.
    long long out_bytes, ref_bytes;
    long long in_bytes, amax_n, mask_words;   // read by the MMLF_BOUNDS_DEBUG build only
};


// epilogue shared by the f32 and the split-bf16 kernels: D[row = position][col = channel]; a lane
// holds column i of every N tile and rows (r&3)+8(r>>2)+4kh of its wave's 32 positions.
// Epilogues.  All eight waves of a workgroup run theirs at the same time, so the matrix cores idle
// meanwhile: written for few instructions and few serialized memory round trips.
//  * row validity: lane l classifies row l&31 of the wave's 32 positions (ONE division per lane) and a
//    ballot makes the 32-bit row mask;
//  * addressing: buffer instructions on a wave-uniform tile descriptor + 32-bit per-lane byte offsets
//    (the column-block offset folds into the instruction's immediate);
//  * bias and (data-gradient) ReLU-reference values are loaded in batches ahead of their use.
__device__ __forceinline__ unsigned wave_row_mask(const ConvArgs &a, long long Q0, int w, int lane)
{
    const unsigned qrow = (unsigned)Q0 + 32 * w + (lane & 31);
    const unsigned row = fastdiv(qrow, a.divP);                 // global grid row
    const int x = (int)(qrow - row * a.P), y = (int)(row - fastdiv(row, a.divR) * a.R);
    return (unsigned)__ballot((long long)qrow < a.NQ && y < a.vh && x < a.vw);
}

// 32x32 tiling: lane (i, kh) holds column i of each 32-column block, rows (r&3) + 8*(r>>2) + 4*kh.
template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, const f32x16 (&acc)[NT], long long Q0, int w, int i,
                                              int kh)
{
    const unsigned m = wave_row_mask(a, Q0, w, i) >> (4 * kh);
    const long long qb = Q0 + 32 * w + a.out_shift;                                     // wave-uniform
    const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(
        a.out + (size_t)qb * a.cs_out, 0, mmlf_records_left(a.out_bytes, qb * a.cs_out * 4ll), MMLF_BUF_FLAGS);
    const bool has_ref = a.ref != nullptr;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_ref ? a.ref + (size_t)qb * a.cs_ref : a.out), 0,
        has_ref ? mmlf_records_left(a.ref_bytes, qb * a.cs_ref * 4ll) : 0, MMLF_BUF_FLAGS);
    unsigned lo = ((unsigned)(4 * kh) * a.cs_out + i) * 4u;
    unsigned lr = ((unsigned)(4 * kh) * a.cs_ref + i) * 4u;
    // the per-row offsets derived from these are tile-invariant: opaque to the optimiser so that it does
    // not hoist 32 of them out of the persistent loop and hold them in VGPRs through the main loop
    asm volatile("" : "+v"(lo), "+v"(lr));
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = 32 * nt + i;
        bv[nt] = (a.bias && ch < a.n_true) ? a.bias[ch] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (32 * nt + i >= a.n_store) continue;
        unsigned keep = m;                 // bit rc: row valid (and, for a data gradient, reference > 0)
        if (has_ref) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    rb, lr + (unsigned)((r & 3) + 8 * (r >> 2)) * a.cs_ref * 4u + 128 * nt, 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (!(rv[r] > 0.f)) keep &= ~(1u << ((r & 3) + 8 * (r >> 2)));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = (r & 3) + 8 * (r >> 2);
            float v = acc[nt][r] + bv[nt];
            if (a.relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((keep >> rc & 1) ? v : 0.f), ob,
                                                  lo + (unsigned)rc * a.cs_out * 4u + 128 * nt, 0, 0);
        }
    }
}

// f16-split filter packing: [chunk][plane(2)][tap(4)][NP][8 f16] of w * scale[n], one workgroup per packed
// column n (= output channel of the launch).  Every column carries its own power-of-two scale (its max |w|
// goes to [2^14, 2^15)), so an output channel whose weights are small as a whole keeps its 22 bits;
// unscale[n] = 1 / scale[n] is what the convolution's epilogue multiplies column n by.
__device__ __forceinline__ void pack_filter_h2_column(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                      int Cout, int Cin, int variant, int dgrad, int nchunk, int NP,
                                                      float *__restrict__ unscale, int n, float *red)
{
    const int tid = threadIdx.x;
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    auto at = [&](int k, int tap) -> size_t {      // OIHW index of (packed row k, column n, master tap)
        return dgrad ? ((size_t)k * Cin + n) * 4 + tap : ((size_t)n * Cin + k) * 4 + tap;
    };
    float m = 0.f;
    if (n < N)
        for (int e = tid; e < 4 * K; e += 256) m = fmaxf(m, fabsf(w[at(e >> 2, e & 3)]));
    m = mmlf_wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const float sc = pow2_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (tid == 0) unscale[n] = 1.f / sc;
    for (int e = tid; e < 4 * nchunk; e += 256) {
        const int c = e >> 2, t = e & 3;
        const int tap = master_tap(dgrad ? 3 - t : t, variant);
        unsigned short h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * c + j;
            const float v = (k < K && n < N) ? w[at(k, tap)] * sc : 0.f;
            const _Float16 hh = (_Float16)v, ll = (_Float16)(v - (float)hh);
            h[j] = __builtin_bit_cast(unsigned short, hh);
            l[j] = __builtin_bit_cast(unsigned short, ll);
        }
        uint4 vh, vl;
        vh.x = h[0] | (unsigned)h[1] << 16; vh.y = h[2] | (unsigned)h[3] << 16;
        vh.z = h[4] | (unsigned)h[5] << 16; vh.w = h[6] | (unsigned)h[7] << 16;
        vl.x = l[0] | (unsigned)l[1] << 16; vl.y = l[2] | (unsigned)l[3] << 16;
        vl.z = l[4] | (unsigned)l[5] << 16; vl.w = l[6] | (unsigned)l[7] << 16;
        *reinterpret_cast<uint4 *>(out + ((((size_t)c * 2 + 0) * 4 + t) * NP + n) * 8) = vh;
        *reinterpret_cast<uint4 *>(out + ((((size_t)c * 2 + 1) * 4 + t) * NP + n) * 8) = vl;
    }
}

__global__ __launch_bounds__(256) void pack_filter_h2_kernel(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                             int Cout, int Cin, int variant, int dgrad, int nchunk,
                                                             int NP, float *__restrict__ unscale)
{
    __shared__ float red[4];
    pack_filter_h2_column(w, out, Cout, Cin, variant, dgrad, nchunk, NP, unscale, blockIdx.x, red);
}

// many filters, one launch: workgroup b packs column b - col0 of the filter whose column range holds b
__global__ __launch_bounds__(256) void pack_filters_h2_kernel(const mmlf_pack_desc *__restrict__ table, int n)
{
    __shared__ float red[4];
    int lo = 0, hi = n - 1;                          // last descriptor with col0 <= blockIdx.x (wave-uniform search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].col0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mmlf_pack_desc d = table[lo];
    const int K = d.dgrad ? d.Cout : d.Cin;
    const int nchunk = (K + 7) / 8;
    unsigned short *out = reinterpret_cast<unsigned short *>(d.packed);
    float *tail = reinterpret_cast<float *>(reinterpret_cast<char *>(d.packed) + (size_t)nchunk * 8 * d.np * 16);
    pack_filter_h2_column(d.w_oihw, out, d.Cout, d.Cin, d.variant, d.dgrad, nchunk, d.np, tail, (int)blockIdx.x - d.col0, red);
}

// split-precision filter packing: [chunk][plane(3)][tap(4)][NP][8 bf16], k = 8*chunk+j
__global__ void pack_filter_split_kernel(const float *__restrict__ w, unsigned short *__restrict__ out, int Cout,
                                         int Cin, int variant, int dgrad, int nchunk, int NP)
{
    const long long total = (long long)nchunk * 2 * 2 * NP * 8;   // one thread per (chunk,u,kh,n,j)
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = idx & 7;
        long long r = idx >> 3;
        const int n = r % NP; r /= NP;
        const int kh = r & 1; r >>= 1;
        const int u = r & 1; r >>= 1;
        const int c = (int)r;
        const int k = 8 * c + j, t = 2 * u + kh;
        int ci, co, tsrc;
        if (!dgrad) { ci = k; co = n; tsrc = t; }
        else { co = k; ci = n; tsrc = 3 - t; }
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * 4 + master_tap(tsrc, variant)];
        unsigned p[3];
        split3(v, p[0], p[1], p[2]);
        for (int pl = 0; pl < 3; ++pl) {
            out[((((size_t)c * 3 + pl) * 4 + t) * NP + n) * 8 + j] = (unsigned short)p[pl];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// forward / data-gradient kernel, split-bf16 arithmetic ("bf16x6"):
// every f32 operand is split EXACTLY into three bf16 (hi+mid+lo) and each product is evaluated as the
// six leading cross terms on the bf16 matrix cores with f32 accumulation (dropped terms are
// <= 2^-26 relative).  Measured on gfx950 (tools/bf16x6_accuracy.hip): error vs a double reference
// 1.4e-8*sum|a*b| mean, 1.0e-7 max at K=1120 -- slightly BELOW the f32 MFMA fma chain (1.7e-8 /
// 1.9e-7) -- at 16/6 = 2.67x the f32 MFMA rate.
// ---------------------------------------------------------------------------------------------

// 16x16 tiling: lane (r16, q4) holds column r16 of each 16-column block and rows 16*mb + 4*q4 + r.
// unscale_a undoes this wave's activation scale, a.w_unscale[n] column n's weight scale (powers of two: exact).
// With a.out_amax the wave also raises the output's amax array: its 32 positions lie in at most two grid rows
// when P >= 32 (exact row maxima); for smaller pitches the rows behind the first get the common maximum
// (an upper bound, which is all the consumers need).
// EPI selects what the epilogue is compiled for: EPI_GENERIC reads every option from the launch arguments at run
// time; the other values are the launch kinds of a training step, each compiled WITHOUT the code of the others (the
// generic form carries the registers and branches of all options through every launch: 2-3 % of a 280-channel one).
enum { EPI_GENERIC = -1, EPI_PLAIN = 0, EPI_RELU = 1, EPI_STATS = 2, EPI_BITS_IN = 4, EPI_REF_IN = 8, EPI_MASK_OUT = 16 };
template <int G, int EPI>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs &a, const f32x4 (&acc)[2][G], long long Q0, int w,
                                                int r16, int q4, float unscale_a, float &run_max,
                                                double *stats_in /* this wave's [16*G][2] sums, or null */)
{
    constexpr bool GEN = EPI == EPI_GENERIC;
    const bool do_relu = GEN ? a.relu != 0 : (EPI & EPI_RELU) != 0;
    double *const stats = (GEN || (EPI & EPI_STATS)) ? stats_in : nullptr;
    const bool mask_out = GEN ? a.relu_mask_out != nullptr : (EPI & EPI_MASK_OUT) != 0;
    const unsigned m = wave_row_mask(a, Q0, w, r16 + 16 * q4) >> (4 * q4);
    const long long qb = Q0 + 32 * w + a.out_shift;                                     // wave-uniform
    const int ob_left = mmlf_records_left(a.out_bytes, qb * a.cs_out * 4ll);
    const __amdgpu_buffer_rsrc_t ob =
        __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)qb * a.cs_out, 0, ob_left, MMLF_BUF_FLAGS);
    const bool has_ref = GEN ? a.ref != nullptr : (EPI & EPI_REF_IN) != 0;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_ref ? a.ref + (size_t)qb * a.cs_ref : a.out), 0,
        has_ref ? mmlf_records_left(a.ref_bytes, qb * a.cs_ref * 4ll) : 0, MMLF_BUF_FLAGS);
    unsigned lo = ((unsigned)(4 * q4) * a.cs_out + r16) * 4u;
    unsigned lr = ((unsigned)(4 * q4) * a.cs_ref + r16) * 4u;
    asm volatile("" : "+v"(lo), "+v"(lr));   // see conv_epilogue
    float mk[8];                             // max |out| per position this lane holds (over the column blocks)
#pragma unroll
    for (int k = 0; k < 8; ++k) mk[k] = 0.f;
    // ReLU masks as bits: the forward pass leaves (out > 0) of every element it wrote in the accumulator layout of
    // THIS kernel -- word (tile, wave, row k, lane), bit = column block -- and the data gradient of the layer above
    // (same grid, same N, hence the same layout) reads its eight words with eight coalesced loads up front instead of
    // 8 x G scattered loads of the activations between its stores (a fifth of that launch)
    const bool use_bits = GEN ? a.relu_mask_in != nullptr : (EPI & EPI_BITS_IN) != 0;
    const size_t mbase = ((size_t)(Q0 / MMLF_TILE) * 8 + w) * 512 + (r16 + 16 * q4);
    unsigned mw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) mw[k] = use_bits ? a.relu_mask_in[mbase + 64 * k] : 0u;
    MMLF_OOB(OOB_MASK, (use_bits || mask_out) && (long long)mbase + 64 * 7 >= a.mask_words);
    // bias and weight-unscale of all of this lane's columns up front: a load between the stores makes the compiler
    // wait for the wave's vector-memory counter to reach zero there, i.e. for every store issued so far (and every DMA
    // piece in flight) -- once per column block
    float bv[G], uwv[G];
#pragma unroll
    for (int nb = 0; nb < G; ++nb) {
        const int ch = 16 * nb + r16;
        bv[nb] = (a.bias && ch < a.n_true) ? a.bias[ch] : 0.f;
        uwv[nb] = (a.w_unscale && ch < 16 * G) ? a.w_unscale[ch] : 1.f;
    }
#pragma unroll
    for (int nb = 0; nb < G; ++nb) asm volatile("" : "+v"(bv[nb]), "+v"(uwv[nb]));   // loaded HERE, not sunk to their uses
#pragma unroll
    for (int nb = 0; nb < G; ++nb) {
        const int ch = 16 * nb + r16;
        if (ch >= a.n_store) continue;
        const float bvn = bv[nb];
        const float uw = uwv[nb];
        unsigned keep = m;
        float s1 = 0.f, s2 = 0.f;
        if (use_bits) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (!(mw[k] >> nb & 1)) keep &= ~(1u << (16 * (k >> 2) + (k & 3)));
        } else if (has_ref) {
            float rv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                MMLF_OOB(OOB_REF, (qb * a.cs_ref * 4ll) + lr + (long long)(16 * (k >> 2) + (k & 3)) * a.cs_ref * 4 + 64 * nb >= a.ref_bytes);
                rv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    rb, lr + (unsigned)(16 * (k >> 2) + (k & 3)) * a.cs_ref * 4u + 64 * nb, 0, 0));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (!(rv[k] > 0.f)) keep &= ~(1u << (16 * (k >> 2) + (k & 3)));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rc = 16 * (k >> 2) + (k & 3);
            float v = fmaf(acc[k >> 2][nb][k & 3] * unscale_a, uw, bvn);   // the products are exact (powers of two): one rounding, as before
            if (do_relu) v = fmaxf(v, 0.f);
            v = (keep >> rc & 1) ? v : 0.f;
            if (mask_out) mw[k] |= (v > 0.f ? 1u : 0u) << nb;
            asm("v_max_f32 %0, %1, |%2|" : "=v"(mk[k]) : "v"(mk[k]), "v"(v));   // (fmaxf canonicalises its operand first: two instructions)
            s1 += v;
            s2 = fmaf(v, v, s2);
            // row offset in the scalar offset operand, column block in the immediate: no address arithmetic per element
            MMLF_OOB(OOB_OUT, (long long)lo + 64 * nb + (long long)rc * a.cs_out * 4 >= ob_left);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ob, lo + 64 * nb, (unsigned)rc * a.cs_out * 4u, 0);
        }
        if (stats) {            // BatchNorm statistics: this wave's 32 positions of channel 16*nb + r16
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (q4 == 0) {      // one owner lane per channel: plain read-modify-write, fixed order
                stats[2 * (16 * nb + r16)] += (double)s1;
                stats[2 * (16 * nb + r16) + 1] += (double)s2;
            }
        }
    }
    if (mask_out) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a.relu_mask_out[mbase + 64 * k] = mw[k];
    }
    if (a.out_amax) {
        const unsigned d0 = (unsigned)qb;                       // first destination position of the wave
        const unsigned rd0 = fastdiv(d0, a.divP);
        const int nfirst = (int)((rd0 + 1) * (unsigned)a.P - d0);   // wave positions [0, nfirst) lie in grid row rd0
        float m_lo = 0.f, m_hi = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pos = 16 * (k >> 2) + 4 * q4 + (k & 3);
            if (pos < nfirst) m_lo = fmaxf(m_lo, mk[k]);
            else m_hi = fmaxf(m_hi, mk[k]);
        }
        m_lo = mmlf_wave_max(m_lo);
        m_hi = mmlf_wave_max(m_hi);
        if (r16 + 16 * q4 == 0) {
            MMLF_OOB(OOB_AMAX, MMLF_AMAX_HEAD + (long long)fastdiv(d0 + 31, a.divP) >= a.amax_n);
            if (m_lo > 0.f) mmlf_amax_raise_nowait(a.out_amax + MMLF_AMAX_HEAD + rd0, m_lo);
            if (m_hi > 0.f) {
                const unsigned rdl = fastdiv(d0 + 31, a.divP);
                for (unsigned r = rd0 + 1; r <= rdl; ++r) mmlf_amax_raise_nowait(a.out_amax + MMLF_AMAX_HEAD + r, m_hi);
            }
            run_max = fmaxf(run_max, fmaxf(m_lo, m_hi));        // lane 0 carries the tensor maximum
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TRANSPOSED epilogue (round 5).  The MFMA is symmetric in its two operands: with the weights as the A operand and the
// activations as B the same instruction yields the transposed tile -- lane (r16, q4) then holds FOUR CONSECUTIVE CHANNELS
// 16 nb + 4 q4 + r of ONE position 16 mb + r16 per accumulator tile instead of one channel of four positions.  What that buys:
//  * the tile is stored with 16-byte instructions (the four lanes of a position write 64 consecutive bytes): 2 G store
//    instructions per wave and tile instead of 8 G.  The epilogue's store tail is bound by the ISSUE of its vector-memory
//    instructions, not by their bytes (round 2: all eight waves stand in it at once and the matrix cores idle meanwhile);
//  * a position's validity / ReLU bits are per lane and row block, not per value: the row mask is two bits per lane,
//    the ReLU mask words hold (nb, r) as bit 4 (nb % 8) + r of word nb / 8: 2 ceil(G / 8) words per lane instead of 8
//    (a third of the mask traffic on the 280-wide layers, a quarter on the 70-wide ones);
//  * bias and weight-unscale of a lane's four channels are two ds_read_b128 from a table the workgroup put into LDS once
//    per launch (global loads between the stores made every column block wait for all stores issued so far).
// The values are formed exactly as in conv_epilogue16 -- fma(acc * unscale_a, uw, bias), one rounding -- so the two
// orientations write the same bits (tests/test_gpu_kernels.py::test_transposed_epilogue_writes_the_same_bits).
// Launch kinds: plain, ReLU, ReLU + mask-out, mask-in (three of a training block's four convolution launches); the
// statistics kind keeps conv_epilogue16 (its sums run over positions = along a lane's registers there, across lanes here).
// Needs n_store % 4 == 0 and a 16-byte aligned `out` (the host checks; channel slices at odd offsets take the other form).
// ---------------------------------------------------------------------------------------------
template <int G> constexpr int tr_mask_words() { return (G + 7) / 8; }
template <int G, int EPI>
__device__ __forceinline__ void conv_epilogue16_tr(const ConvArgs &a, const f32x4 (&acc)[2][G], long long Q0, int w,
                                                   int r16, int q4, float unscale_a, float &run_max,
                                                   const float *coef /* LDS: [16 G] weight-unscale, [16 G] bias */)
{
    static_assert(EPI >= 0 && !(EPI & (EPI_STATS | EPI_REF_IN)), "kinds of the transposed epilogue");
    constexpr int NP = 16 * G, MW = tr_mask_words<G>();
    constexpr bool do_relu = (EPI & EPI_RELU) != 0, mask_out = (EPI & EPI_MASK_OUT) != 0, use_bits = (EPI & EPI_BITS_IN) != 0;
    const unsigned m = wave_row_mask(a, Q0, w, r16 + 16 * q4);      // bit p: position p of the wave's 32 is a valid output
    const long long qb = Q0 + 32 * w + a.out_shift;                   // wave-uniform
    const int ob_left = mmlf_records_left(a.out_bytes, qb * a.cs_out * 4ll);
    const __amdgpu_buffer_rsrc_t ob =
        __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)qb * a.cs_out, 0, ob_left, MMLF_BUF_FLAGS);
    unsigned lo = ((unsigned)r16 * a.cs_out + 4u * q4) * 4u;
    asm volatile("" : "+v"(lo));             // (tile-invariant: kept out of the persistent loop's registers, see conv_epilogue)
    const size_t mbase = ((size_t)(Q0 / MMLF_TILE) * 8 + w) * 512 + (r16 + 16 * q4);
    unsigned mw[2][MW];
    bool ok[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        ok[mb] = (m >> (16 * mb + r16) & 1u) != 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            mw[mb][j] = use_bits ? a.relu_mask_in[mbase + 64 * (MW * mb + j)] : 0u;
            if (use_bits && !ok[mb]) mw[mb][j] = 0u;        // an invalid position keeps nothing
        }
    }
    MMLF_OOB(OOB_MASK, (use_bits || mask_out) && (long long)mbase + 64 * (2 * MW - 1) >= a.mask_words);
    float mk[2] = {0.f, 0.f};                 // max |out| of this lane's two positions
    // the table values of column block nb + 1 are requested while block nb is worked on; the fences keep the scheduler from
    // pulling ALL blocks' reads to the front (8 registers per block: the wide kernel has none to spare)
    const float *cl = coef + 4 * q4;
    f32x4 uw_n = *reinterpret_cast<const f32x4 *>(cl), b_n = *reinterpret_cast<const f32x4 *>(cl + NP);
#pragma unroll
    for (int nb = 0; nb < G; ++nb) {
        const int c0 = 16 * nb + 4 * q4;
        const f32x4 uw4 = uw_n, b4 = b_n;
        if (nb + 1 < G) {
            uw_n = *reinterpret_cast<const f32x4 *>(cl + 16 * (nb + 1));
            b_n = *reinterpret_cast<const f32x4 *>(cl + NP + 16 * (nb + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            u32x4 st;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(acc[mb][nb][r] * unscale_a, uw4[r], b4[r]);   // the products are exact (powers of two): one rounding
                if (do_relu) v = fmaxf(v, 0.f);
                if (use_bits) {              // 0 or ~0 from the bit: two instructions per value
                    const int keep = ((int)(mw[mb][nb >> 3] << (31 - (4 * (nb & 7) + r)))) >> 31;
                    v = __uint_as_float(__float_as_uint(v) & (unsigned)keep);
                } else {
                    v = ok[mb] ? v : 0.f;
                }
                if (mask_out) mw[mb][nb >> 3] |= (v > 0.f ? 1u : 0u) << (4 * (nb & 7) + r);
                asm("v_max_f32 %0, %1, |%2|" : "=v"(mk[mb]) : "v"(mk[mb]), "v"(v));
                st[r] = __float_as_uint(v);
            }
            // (MMLF_ABL_RS_FUSE, timing proxy: the 80-column kernels' pad-1 launches -- a stream block's first convolution -- store
            //  nothing; the values are still formed, they feed the row maxima)
            if (c0 < a.n_store && !(MMLF_ABL_RS_FUSE && G == 5 && a.out_shift == 0)) {
                MMLF_OOB(OOB_OUT, (long long)lo + 64 * nb + (long long)(16 * mb) * a.cs_out * 4 + 12 >= ob_left);
                __builtin_amdgcn_raw_buffer_store_b128(st, ob, lo + 64 * nb, (unsigned)(16 * mb) * a.cs_out * 4u, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (mask_out) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < MW; ++j) a.relu_mask_out[mbase + 64 * (MW * mb + j)] = mw[mb][j];
    }
    if (a.out_amax) {
        const unsigned d0 = (unsigned)qb;                       // first destination position of the wave
        const unsigned rd0 = fastdiv(d0, a.divP);
        const int nfirst = (int)((rd0 + 1) * (unsigned)a.P - d0);   // wave positions [0, nfirst) lie in grid row rd0
        float m_lo = 0.f, m_hi = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (16 * mb + r16 < nfirst) m_lo = fmaxf(m_lo, mk[mb]);
            else m_hi = fmaxf(m_hi, mk[mb]);
        }
        m_lo = mmlf_wave_max(m_lo);
        m_hi = mmlf_wave_max(m_hi);
        if (r16 + 16 * q4 == 0) {
            MMLF_OOB(OOB_AMAX, MMLF_AMAX_HEAD + (long long)fastdiv(d0 + 31, a.divP) >= a.amax_n);
            if (m_lo > 0.f) mmlf_amax_raise_nowait(a.out_amax + MMLF_AMAX_HEAD + rd0, m_lo);
            if (m_hi > 0.f) {
                const unsigned rdl = fastdiv(d0 + 31, a.divP);
                for (unsigned r = rd0 + 1; r <= rdl; ++r) mmlf_amax_raise_nowait(a.out_amax + MMLF_AMAX_HEAD + r, m_hi);
            }
            run_max = fmaxf(run_max, fmaxf(m_lo, m_hi));        // lane 0 carries the tensor maximum
        }
    }
}
// the workgroup's table of per-column weight-unscale and bias values (transposed epilogue), filled once per launch
template <int NP>
__device__ __forceinline__ void conv_fill_coef(const ConvArgs &a, float *coef, int tid, int nthreads)
{
    for (int k = tid; k < NP; k += nthreads) {
        coef[k] = a.w_unscale ? a.w_unscale[k] : 1.f;
        coef[NP + k] = (a.bias && k < a.n_true) ? a.bias[k] : 0.f;
    }
}

// f16 split: the power-of-two scale wave w of tile Q0 applies to its activation operand.  The wave's valid
// outputs q read in[q + {0, 1, P, P+1}], i.e. grid rows row(q0) .. row(q0 + 31) + 1 (no row is added behind a
// patch's last row: its outputs are never valid), so the scale comes from those rows' maxima alone: a wave's
// precision does not depend on what the rest of the tensor holds (neighbouring patches, far-away rows).
// Two halves so that the row loads of the NEXT tile fly during this tile's epilogue: gather (per-lane maximum of
// the rows lane, lane + 64, ... of the range) and finish (wave maximum -> scale).
__device__ __forceinline__ float wave_operand_amax_gather(const ConvArgs &a, long long Q0, int w, int lane)
{
    const long long q0 = Q0 + 32 * w;
    float m = 0.f;
    if (q0 < a.NQ) {                                             // wave-uniform; else nothing valid in this wave
        const long long ql = q0 + 31 < a.NQ ? q0 + 31 : a.NQ - 1;
        const unsigned r0 = fastdiv((unsigned)q0, a.divP);
        unsigned r1 = fastdiv((unsigned)ql, a.divP);
        r1 += (r1 - fastdiv(r1, a.divR) * (unsigned)a.R != (unsigned)(a.R - 1)) ? 1u : 0u;
        MMLF_OOB(OOB_AMAX, lane == 0 && MMLF_AMAX_HEAD + (long long)r1 >= a.amax_n);
        for (unsigned r = r0 + lane; r <= r1; r += 64) m = fmaxf(m, a.in_amax[MMLF_AMAX_HEAD + r]);
    }
    return m;
}
__device__ __forceinline__ float wave_operand_scale(float gathered)
{
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pow2_scale_for(mmlf_wave_max(gathered)))));
}

// The launch arguments as stored in the kernel-argument segment, behind a pointer the optimiser cannot see through:
// what is read through it is loaded where it is used (scalar loads) instead of living in SGPRs across the main
// loop.  The persistent kernel's main loop needs a handful of arguments, its epilogue some twenty -- kept live,
// they push the loop's own scalars into spill lanes (0.3 ms of an 8 ms launch).
typedef const __attribute__((address_space(4))) ConvArgs *ConvKernArgs;
__device__ __forceinline__ ConvArgs late_args()
{
#if defined(__HIP_DEVICE_COMPILE__)
    ConvKernArgs p = (ConvKernArgs)__builtin_amdgcn_kernarg_segment_ptr();   // ConvArgs is the kernel's first parameter
    asm volatile("" : "+s"(p));
    return *p;
#else
    return ConvArgs();
#endif
}

// 512 threads = 8 waves; tile = 256 positions x 16*G output channels; wave w owns positions
// [32w, 32w+32) x all channels as 2 x G accumulator tiles of v_mfma_f32_16x16x32_bf16 (under the
// chip's power-limited clock this shape sustains more FLOP/s than 32x32x16: 217 vs 204 TFLOP/s
// algorithmic at 280->280).  K is walked in chunks of 8 input channels x 4 taps = one MFMA K (lane
// quarter q4 = lane>>4 carries tap q4), double-buffered in LDS and filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers; the copy of chunk c+1 is in flight while chunk c
// is multiplied).  Activations stay f32 in LDS and are split in registers; weights arrive pre-split
// from pack_filter_split_kernel as [chunk][plane(3)][tap(4)][NP][8 bf16].
// G <= 6 (narrow layers): VGPRs capped at 128 so that TWO workgroups share a CU -- with 60 short MFMAs
// per wave per chunk the DMA latency of a single double-buffered workgroup is exposed.
// G = number of 16-column output blocks (NP = 16*G packed columns): 2, 5 (the 70-channel layers: 80
// columns instead of 96), 6, 7, 8 or 18.
// PL = operand planes: 3 = bf16 3-way split, six passes ("bf16x6"); 2 = f16 2-way split of the scaled
// operands, three passes ("f16x3").
// NW = waves per workgroup = 32-position row groups per tile: 8 (256-position tiles) or, for the narrow layers on small
// pitches, 16 (512 positions, ONE workgroup per CU instead of two: the 99-position halo is 19 % of the window instead of
// 39 %, the weights are fetched once per 512 positions, and more activation bytes are in flight per CU -- these launches
// are bound by memory concurrency, not by the matrix cores).  Layouts (masks, statistics, scales) are indexed by the
// global 32-position group, so the two variants produce the same bytes.
// TR: transposed accumulator tiles and epilogue (conv_epilogue16_tr above).
template <int G, int PL, int EPI = EPI_GENERIC, int NW = 8, bool TR = false>
__global__ __launch_bounds__(64 * NW, (NW == 16 || G <= 6 ? 4 : 2)) void conv4tap_x6s_kernel(ConvArgs a, int ntiles)
{
    constexpr int NP = G * 16;
    constexpr int TILE = 32 * NW;
    // pipeline buffers: two; the sixteen-wave variant has the LDS for a ring of MMLF_RING16, with the DMA of chunk
    // c + D - 1 issued during chunk c and a counted wait that leaves the newest D - 2 chunks' pieces in flight
    constexpr int D = NW == 16 ? MMLF_RING16 : 2;
    // EARLY: the chunk's barrier stands two column blocks before its end (the weight fragments of those two blocks are
    // in registers by then): behind it the wave requests the NEXT chunk's activation and first weight fragments and runs
    // its last twelve MFMAs while they arrive -- the LDS latency of the chunk head (both waves of a SIMD stand in it at
    // once: 12 % of a 280-wide chunk by the wave's own clock) is off the critical path.  Needs G % 3 == 0 (the rotating
    // fragment slots line up across chunks) and enough column blocks behind the DMA issue for the pieces to land.
    constexpr bool EARLY = PL == 2 && G % 3 == 0 && G >= 9;   // (the three-plane build has no registers to spare for it)
    // A: [640 slots][channel half(2)] float4.  Slot s holds position Q0 + s (+ seg_delta for s >= 320).
    // When the pitch is small (P + 257 <= 640, e.g. 96x96 training patches) ONE contiguous window
    // Q0 .. Q0+256+P serves all four taps (taps 2,3 read at slot offset P): 355 positions per tile
    // instead of 2 x 257.  Otherwise two 320-slot segments (rows y and y+1) are loaded.
    // A DMA piece is 32 positions x 32 bytes: lanes 2i, 2i+1 fetch the two 16-byte halves of position i's chunk, ONE
    // 32-byte sector request for the memory pipeline (half-major pieces of 64 positions asked for every sector twice,
    // from two instructions; the 70-channel launches are bound by that request rate).
    constexpr int A_F4 = 2 * 640;
    constexpr int B_F4 = 4 * PL * NP;
    constexpr int BUF_F4 = A_F4 + B_F4;
    constexpr int N_B = B_F4 / 64;
    constexpr int PER_WAVE = (20 + N_B + NW - 1) / NW;   // upper bound (20 A pieces: two-segment mode, 512-position tiles)
    constexpr int PER_SLOT = (PER_WAVE + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q4 = lane >> 4;

    f32x4 acc[2][G];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < G; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;

    // DMA addressing: a piece = wave-uniform 64-bit base (SGPRs) + one shared per-lane byte offset
    // (chunk-blocked input, a.a_blocked: a piece is 32 positions x 32 bytes of ONE contiguous 1 KiB run -- lane l fetches bytes
    //  [16 l, 16 l + 16); the tile's chunks are TILE x 32 bytes apart and slots >= TILE live in the next tile's block)
    const bool blocked = (G == 5 && PL == 2) ? a.a_blocked != 0 : false;      // (the other widths: a compile-time false)
    const unsigned voff_a = blocked ? (unsigned)lane * 16u
                                    : ((unsigned)(lane >> 1) * (unsigned)a.cs_in + 4u * (lane & 1)) * 4u;   // A pieces
    const unsigned a_chunk_stride = blocked ? (unsigned)TILE * 32u : 32u;
    const unsigned voff_b = (unsigned)lane * 16u;                      // B pieces: linear
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;
    const char *in0 = reinterpret_cast<const char *>(a.in);
    const char *wp_base = reinterpret_cast<const char *>(a.wp);
    const size_t tile_bytes = (size_t)TILE * a.cs_in * 4;
    // Piece ownership, fixed for the launch: of the chunk's pieces j = 0 .. a_pieces + N_B - 1 (activation
    // pieces first) wave w issues j = w, w+8, ...: nA activation pieces, then nB weight pieces that are
    // 8 KiB apart in both the packed filter and LDS.  The first half of them goes out in slot 0.
    const int n_a = a.a_pieces;
    const int nA = (n_a - w + NW - 1) / NW;
    const int jb0 = w + NW * nA - n_a;
    const int nB = (N_B - jb0 + NW - 1) / NW;
    int n_mine = nA + nB;                                   // re-made opaque every chunk (see below)
    unsigned a_src[3], a_dst[3];                            // activation pieces: byte offset in the tile, LDS byte
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = w + NW * k;
        const int seg = j >= a.a_per_seg, idx = j - seg * a.a_per_seg;   // two-segment mode: 320-slot segments
        const int slot = 320 * seg + 32 * idx;
        a_src[k] = blocked ? (unsigned)(slot < TILE ? slot : a.nchunk * TILE + slot - TILE) * 32u
                               : (unsigned)(slot + seg * a.seg_delta) * (unsigned)a.cs_in * 4u;
        // bit 0 marks the last piece of a window (of a segment): only its first a_tail positions are ever read; the
        // other lanes re-fetch the last of those instead of positions nobody uses
        a_dst[k] = (unsigned)slot * 32u + (idx == a.a_per_seg - 1 ? 1u : 0u);
    }
    const unsigned tail_lim = blocked ? (unsigned)(a.a_tail - 1) * 32u + 16u
                                      : ((unsigned)(a.a_tail - 1) * (unsigned)a.cs_in + 4u) * 4u;
    const unsigned b_src0 = 1024u * jb0, b_dst0 = (unsigned)(A_F4 + 64 * jb0) * 16u;

#ifdef MMLF_BOUNDS_DEBUG      // the piece's last source byte against the buffer (the address is made scalar again for the asm)
#define MMLF_DMA_SRC_CHECK()                                                                             \
    do {                                                                                                 \
        MMLF_OOB(OOB_IN, (long long)(sb_ - in0) + vo_ + 16 > a.in_bytes);                                \
        const unsigned long long u_ = (unsigned long long)sb_;                                           \
        sb_ = reinterpret_cast<const char *>(                                                            \
            ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u_ >> 32)) << 32) |           \
            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u_));                                     \
    } while (0)
#else
#define MMLF_DMA_SRC_CHECK() do { } while (0)
#endif
    // slot 0 = this wave's pieces k < PER_SLOT, slot 1 = the rest; k is a compile-time index
#define X6_DMA_PIECE(tl, c, buf, k)                                                                      \
    do {                                                                                                 \
        if ((k) < PER_WAVE && (k) < n_mine) {                                                            \
            const char *sb_;                                                                             \
            unsigned vo_, d_;                                                                            \
            if ((k) < nA) {                                                                              \
                sb_ = in0 + (size_t)(tl) * tile_bytes + a_chunk_stride * (c) + a_src[(k) < 3 ? (k) : 0]; \
                d_ = a_dst[(k) < 3 ? (k) : 0];                                                           \
                vo_ = min(voff_a, (d_ & 1u) ? tail_lim : 0xffffffffu);                                   \
                d_ &= ~1u;                                                                               \
                MMLF_DMA_SRC_CHECK();                                                                    \
            } else {                                                                                     \
                const unsigned kb_ = (1024u * NW) * (unsigned)((k) - nA);                                \
                sb_ = wp_base + (size_t)(c) * (B_F4 * 16) + b_src0 + kb_;                                \
                vo_ = voff_b;                                                                            \
                d_ = b_dst0 + kb_;                                                                       \
            }                                                                                            \
            d_ = __builtin_amdgcn_readfirstlane(d_ + lds_base + (buf) * (BUF_F4 * 16));                  \
            unsigned keep_;                                                                              \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                       \
                         "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"                           \
                         : "=&s"(keep_) : "v"(vo_), "s"(sb_), "s"(d_) : "memory");                      \
        }                                                                                                \
    } while (0)
#define X6_DMA_SLOT(tl, c, buf, slot)                                                                    \
    do {                                                                                                 \
        _Pragma("unroll") for (int k_ = 0; k_ < PER_SLOT; ++k_)                                          \
            X6_DMA_PIECE(tl, c, buf, (slot) * PER_SLOT + k_);                                            \
    } while (0)
#define X6_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define X6_CHUNK_WAIT()                                                                                  \
    do {                                                                                                 \
        if (D > 2 && more) {      /* in-order counter: everything but this wave's newest (D - 2) x n_mine pieces has landed */ \
            static_assert(PER_WAVE <= 3 || D == 2, "counted waits are written for up to three pieces per wave");             \
            if (n_mine == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (D - 2)) : "memory");            \
            else if (n_mine == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 2)) : "memory");       \
            else if (n_mine == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (D - 2)) : "memory");       \
            else X6_DMA_WAIT();                                                                          \
        } else {                                                                                         \
            X6_DMA_WAIT();                                                                               \
        }                                                                                                \
    } while (0)

    // Persistent over tiles: the chunk pipeline runs on across tile boundaries, so the DMA of the next
    // tile's first chunk is in flight while this tile's last chunk multiplies and its epilogue stores.
    // XCD-aware tile order: blocks b and b+8 share an XCD (one L2).  Give every XCD a CONTIGUOUS run of
    // tiles per sweep so that the halo rows two neighbouring tiles both read are served by one L2
    // (speed only: any placement computes the same result).
    const int per_xcd = (int)gridDim.x >> 3;
    const int first_tile = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3)
                                                : (int)blockIdx.x;
    int tile = first_tile, c = 0;          // chunk being multiplied
    int ntile = tile, nc = 0;              // chunk being fetched (one ahead)
    if (tile >= ntiles) return;
    // optional BatchNorm statistics of the output: per-wave double sums behind the two pipeline buffers
    // behind the pipeline buffers: the transposed epilogue's per-column table (TR), or the statistics' per-wave double sums
    float *coef = reinterpret_cast<float *>(lds + D * BUF_F4);               // [2][NP]
    double *stats_all = reinterpret_cast<double *>(lds + D * BUF_F4);        // [NW waves][NP][2]
    if constexpr (TR) conv_fill_coef<NP>(late_args(), coef, tid, 64 * NW);   // ordered by the barrier below
    else if (late_args().bn_partial)
        for (int k = tid; k < NW * NP * 2; k += 64 * NW) stats_all[k] = 0.0;
    // f16 split: operand scales (powers of two) and what undoes them in the epilogue
    float scale_a = 1.f, unscale_a = 1.f, run_max = 0.f;
    if constexpr (PL == 2) {
        scale_a = wave_operand_scale(wave_operand_amax_gather(late_args(), (long long)tile * TILE, w, lane));
        unscale_a = 1.f / scale_a;
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
        if (ntile < ntiles) {
            X6_DMA_SLOT(ntile, nc, d, 0);
            X6_DMA_SLOT(ntile, nc, d, 1);
        }
        if (++nc == a.nchunk) { nc = 0; ntile += gridDim.x; }
    }
    X6_DMA_WAIT();
    __syncthreads();
    int buf = 0;
    float4 ra[2][2];                                           // activation fragments of the chunk about to be split
    bf16x8 bq[3][PL];                                          // rotating [slot][plane] weight fragments
    const int a_lane = 2 * (32 * w + r16 + (q4 & 1) + (q4 >> 1) * a.seg_slot);   // float4 index: + 32*mb + half
    const int b_lane = q4 * NP + r16;                                            // bf16x8 index: + pl*4*NP + 16*nb
    if constexpr (EARLY) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) ra[mb][hf] = lds[a_lane + 32 * mb + hf];
#pragma unroll
        for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                bq[g0][pl] = (reinterpret_cast<const bf16x8 *>(lds + A_F4) + b_lane)[pl * 4 * NP + 16 * g0];
    }

    while (tile < ntiles) {
        // opaque to the optimiser: keeps the per-piece address terms derived from it from being hoisted out
        // of the persistent loop into (spilled) SGPRs; they are recomputed on the scalar unit instead
        asm volatile("" : "+s"(n_mine));
        const bool more = ntile < ntiles;
        const int fb = buf == 0 ? D - 1 : buf - 1;             // the buffer multiplied last: free for chunk c + D - 1
        const float4 *base = lds + buf * BUF_F4;
        const float4 *nbase = lds + (buf + 1 == D ? 0 : buf + 1) * BUF_F4;   // the next chunk's buffer
        // lane (r16, q4): row r16 of a 16-position block, tap q4 -> slot offset (q4&1) + (q4>>1)*seg_slot
        const float4 *ap = base + a_lane;                                                    // + 32*mb + half
        const bf16x8 *bp = reinterpret_cast<const bf16x8 *>(base + A_F4) + b_lane;           // + pl*4*NP + 16*nb
        const bf16x8 *bpn = reinterpret_cast<const bf16x8 *>(nbase + A_F4) + b_lane;
        const bool tile_end = c + 1 == a.nchunk;

        if constexpr (!EARLY) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) ra[mb][hf] = ap[32 * mb + hf];
#pragma unroll
            for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) bq[g0][pl] = bp[pl * 4 * NP + 16 * g0];
        }
        bf16x8 asp[2][PL];                                     // [row block][plane] split activations
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            unsigned hh[4], mm[4], ll[4];
            if constexpr (PL == 3) {
                split3_pair(ra[mb][0].x, ra[mb][0].y, hh[0], mm[0], ll[0]);
                split3_pair(ra[mb][0].z, ra[mb][0].w, hh[1], mm[1], ll[1]);
                split3_pair(ra[mb][1].x, ra[mb][1].y, hh[2], mm[2], ll[2]);
                split3_pair(ra[mb][1].z, ra[mb][1].w, hh[3], mm[3], ll[3]);
                const u32x4_t vm = {mm[0], mm[1], mm[2], mm[3]};
                asp[mb][1] = __builtin_bit_cast(bf16x8, vm);
            } else {
                split2_pair_f16(ra[mb][0].x, ra[mb][0].y, scale_a, hh[0], ll[0]);
                split2_pair_f16(ra[mb][0].z, ra[mb][0].w, scale_a, hh[1], ll[1]);
                split2_pair_f16(ra[mb][1].x, ra[mb][1].y, scale_a, hh[2], ll[2]);
                split2_pair_f16(ra[mb][1].z, ra[mb][1].w, scale_a, hh[3], ll[3]);
            }
            const u32x4_t vh = {hh[0], hh[1], hh[2], hh[3]}, vl = {ll[0], ll[1], ll[2], ll[3]};
            asp[mb][0] = __builtin_bit_cast(bf16x8, vh);
            asp[mb][PL - 1] = __builtin_bit_cast(bf16x8, vl);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (EARLY && g == G - 2) {
                // every LDS read of this chunk has been requested (weights run two column blocks ahead): wait for them and
                // for the next chunk's DMA pieces, pass the barrier, then ask for the next chunk's activations
                X6_CHUNK_WAIT();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();
                if (!tile_end) {            // (at a tile's end they would live through the epilogue: requested behind it)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) ra[mb][hf] = (nbase + a_lane)[32 * mb + hf];
                }
            }
            if (g + 2 < G) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) bq[(g + 2) % 3][pl] = bp[pl * 4 * NP + 16 * (g + 2)];
            } else if (EARLY && !tile_end) {   // behind the barrier: the next chunk's first two column blocks (G % 3 == 0)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) bq[(g + 2) % 3][pl] = bpn[pl * 4 * NP + 16 * (g + 2 - G)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) {   // both slots of a wave in one go, as early as the buffer is free; the two waves of a SIMD apart
                if (w < NW / 2) {
                    if (g == 0) { X6_DMA_SLOT(ntile, nc, fb, 0); X6_DMA_SLOT(ntile, nc, fb, 1); }
                } else {
                    if (g == (PL == 2 && G >= 8 ? G / 8 : G / 4)) { X6_DMA_SLOT(ntile, nc, fb, 0); X6_DMA_SLOT(ntile, nc, fb, 1); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // cross terms, small ones first; the two row blocks alternate so that consecutive MFMAs never
            // wait on each other's accumulator
#define X6_TERM(pa, pb)                                                                                      \
    _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                         \
        acc[mb][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(asp[mb][pa], bq[g % 3][pb], acc[mb][g], 0, 0, 0)
#define H2_TERM(pa, pb)                                                                                      \
    _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                         \
        acc[mb][g] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bq[g % 3][pb]),   \
                                                                 __builtin_bit_cast(f16x8, asp[mb][pa]), acc[mb][g], 0, 0, 0) \
                        : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, asp[mb][pa]),     \
                                                                 __builtin_bit_cast(f16x8, bq[g % 3][pb]), acc[mb][g], 0, 0, 0)
            if constexpr (PL == 3) {
                X6_TERM(2, 0);
                X6_TERM(0, 2);
                X6_TERM(1, 1);
                X6_TERM(1, 0);
                X6_TERM(0, 1);
                X6_TERM(0, 0);
            } else {
                // MMLF_ABL_TERMS (ablation builds only, WRONG results): run 2 or 1 of the three cross terms with everything
                // else unchanged -- the time a launch would take with fewer matrix instructions per product
                if (MMLF_ABL_TERMS >= 3) H2_TERM(1, 0);
                if (MMLF_ABL_TERMS >= 2) H2_TERM(0, 1);
                H2_TERM(0, 0);
            }
#undef X6_TERM
#undef H2_TERM
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && ++nc == a.nchunk) { nc = 0; ntile += gridDim.x; }
        // the next chunk's DMA pieces must have landed before the barrier; at a tile end wait for them
        // BEFORE the epilogue, so that its stores (same counter) stay in flight across the barrier
        if constexpr (!EARLY) X6_CHUNK_WAIT();
        if (++c == a.nchunk) {
            const ConvArgs e = late_args(); // epilogue-only arguments: loaded here, dead again at the barrier
            float next_amax = 0.f;          // the next tile's row maxima: loads in flight during the epilogue
            if constexpr (PL == 2)
                if (tile + (int)gridDim.x < ntiles)
                    next_amax = wave_operand_amax_gather(e, (long long)(tile + gridDim.x) * TILE, w, lane);
            if constexpr (TR)
                conv_epilogue16_tr<G, EPI>(e, acc, (long long)tile * TILE, w, r16, q4, unscale_a, run_max, coef);
            else
                conv_epilogue16<G, EPI>(e, acc, (long long)tile * TILE, w, r16, q4, unscale_a, run_max,
                                        e.bn_partial ? stats_all + (size_t)w * NP * 2 : nullptr);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < G; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;
            c = 0;
            tile += gridDim.x;
            if constexpr (PL == 2) {
                if (tile < ntiles) {
                    scale_a = wave_operand_scale(next_amax);
                    unscale_a = 1.f / scale_a;
                }
            }
            if constexpr (EARLY) {          // the next tile's first fragments (its chunk 0 passed the barrier above)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) ra[mb][hf] = (nbase + a_lane)[32 * mb + hf];
#pragma unroll
                for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) bq[g0][pl] = bpn[pl * 4 * NP + 16 * g0];
            }
        }
        if constexpr (!EARLY) __syncthreads();
        buf = buf + 1 == D ? 0 : buf + 1;
    }
    if constexpr (EARLY) __syncthreads();                       // orders the last tile's wave sums
    const ConvArgs e = late_args();
    if (e.out_amax) mmlf_amax_update(run_max, e.out_amax, blockIdx.x * NW + w);      // one atomic per wave per launch
    if (e.bn_partial) {                                         // the loop's last barrier ordered the wave sums
        for (int k = tid; k < 2 * e.n_true; k += 64 * NW) {
            const int ch = k % e.n_true, which = k / e.n_true;
            double t = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) t += stats_all[((size_t)ww * NP + ch) * 2 + which];
            e.bn_partial[((size_t)blockIdx.x * 2 + which) * e.n_true + ch] = t;
        }
    }
#undef X6_DMA_PIECE
#undef MMLF_DMA_SRC_CHECK
#undef X6_DMA_SLOT
#undef X6_DMA_WAIT
#undef X6_CHUNK_WAIT
}



// ---------------------------------------------------------------------------------------------
// Narrow layers, f16 split, round 4: conv4tap_rs_kernel -- "register-streamed".  The stream layers (27 -> 70, 70 -> 70:
// reference feed_forward.py:139-157) have K = 4 x 72 at most and 80 packed columns: 30 MFMAs per wave and 8-channel
// chunk, against which the tiled kernel's per-chunk costs (barrier, chunk head, DMA wait: half of its time by the wave's
// own clock) do not amortise, and whose 355- / 611-position windows come from HBM 1.4-1.5 times.  Here
//  * the WHOLE packed filter (NCH x 10 KB, 92 KB at 72 channels) is copied into LDS once per workgroup and stays;
//  * a wave owns a 32-position group at a time, end to end: every lane loads the full channel row of its (position, tap)
//    straight into registers -- 2 x NCH 16-byte loads per row block, consecutive bytes per lane, issued in one burst so
//    that a 128-byte line is touched by eight back-to-back instructions -- multiplies, runs the common epilogue, moves on;
//  * no LDS traffic for activations, no DMA, NO barrier in the loop: the two waves of a SIMD drift apart by themselves,
//    one multiplies while the other waits for its loads and stores (counted vmcnt waits per chunk: loads return in order);
//  * groups are dealt so that the waves of one XCD work on neighbouring groups at any time: the row a group reads as
//    taps 2, 3 is the row three groups further on read as taps 0, 1 -- an L2 hit instead of a second HBM read.
// Layouts (mask words, statistics, amax, scales) are indexed by the global 32-position group exactly as in
// conv4tap_x6s_kernel (tile = group / 8, wave = group % 8): the two kernels write the same bytes.
// ---------------------------------------------------------------------------------------------
template <int G, int NCH, int EPI, bool TR = false>
__global__ __launch_bounds__(512, 2) void conv4tap_rs_kernel(ConvArgs a, int ngroups)
{
    constexpr int NP = G * 16;
    constexpr int WBYTES = NCH * 2 * 4 * NP * 16;            // [chunk][plane][tap][NP][8 f16]
    // K order.  A "full" MFMA step is ONE tap x 32 channels: lane quarter q4 carries channel octet q4 of the step, so the
    // four lanes of a position read 128 consecutive bytes of its row (one or two cache lines per position and
    // instruction; with quarter = tap, as in the tiled kernel's LDS image, every lane of a load instruction sits in
    // another line and the L1 tag rate, one line per clock, made the loads cost as much as a full HBM stream).  The
    // channels left over (NCH % 4 == 1: 64..71 of 72) form one last step in the quarter = tap layout.  The packed filter
    // needs no other layout: a lane's fragment of step (tap t, octets 4s .. 4s+3) is chunk 4s + q4 of tap t.
    constexpr int NS = NCH / 4, REM = NCH % 4;
    static_assert(REM == 0 || REM == 1, "input channels: a multiple of 32, or one octet more");
    constexpr int NSTEP = 4 * NS + REM, NLOAD = 4 * NSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q4 = lane >> 4;

    for (int o = tid * 16; o < WBYTES; o += 512 * 16)
        *reinterpret_cast<uint4 *>(smem + o) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.wp) + o);
    float *coef = reinterpret_cast<float *>(smem + WBYTES);                   // transposed epilogue: [2][NP]
    double *stats_all = reinterpret_cast<double *>(smem + WBYTES);            // statistics: [8 waves][NP][2]
    if constexpr (TR) conv_fill_coef<NP>(a, coef, tid, 512);
    else if (a.bn_partial)
        for (int k = tid; k < 8 * NP * 2; k += 512) stats_all[k] = 0.0;
    __syncthreads();

    // this wave's groups: XCD x (blocks b with b % 8 == x share it) takes the contiguous range [x * per, (x + 1) * per);
    // inside it the XCD's waves stride together
    int gi, gstep, gend;
    if ((gridDim.x & 7) == 0) {
        const int per = (ngroups + 7) >> 3, nbx = (int)gridDim.x >> 3;
        const int x = (int)blockIdx.x & 7;
        gi = x * per + ((int)blockIdx.x >> 3) * 8 + w;
        gstep = nbx * 8;
        gend = min(ngroups, (x + 1) * per);
    } else {
        gi = (int)blockIdx.x * 8 + w;
        gstep = (int)gridDim.x * 8;
        gend = ngroups;
    }

    f32x4 acc[2][G];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < G; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;
    float run_max = 0.f;
    const bf16x8 *bpF = reinterpret_cast<const bf16x8 *>(smem) + q4 * 8 * NP + r16;   // full steps: + (8 s + pl) * 4 NP + t NP + 16 nb
    const bf16x8 *bpR = reinterpret_cast<const bf16x8 *>(smem) + q4 * NP + r16;       // last step:  + (8 NS + pl) * 4 NP + 16 nb
    auto bfrag = [&](int k, int nb, int pl) -> bf16x8 {        // k, nb, pl are compile-time after unrolling
        return k < 4 * NS ? bpF[(8 * (k % NS) + pl) * 4 * NP + (k / NS) * NP + 16 * nb] : bpR[(8 * NS + pl) * 4 * NP + 16 * nb];
    };
    const size_t row_off = (size_t)r16 * a.cs_in;                                     // floats
    const size_t rem_off = (size_t)((q4 & 1) + (q4 >> 1) * a.P) * a.cs_in + 32 * NS;

    const int gi0 = gi;                                                               // (used by the MMLF_ABL_RS_FUSE proxy only)
    for (; gi < gend; gi += gstep) {
        const long long Q0 = (long long)(gi >> 3) * MMLF_TILE;
        const int wv = gi & 7;
        // (MMLF_ABL_RS_FUSE, timing proxy: a pad-0 launch -- a stream block's second convolution -- reads the activations AND row
        //  maxima of the wave's FIRST group every time: real values with real sparsity (the matrix cores' power follows the data),
        //  served by the caches instead of memory, as a fused block's intermediate would be served by LDS)
        const bool abl_reuse = MMLF_ABL_RS_FUSE && a.out_shift != 0;
        const long long Qa = abl_reuse ? (long long)(gi0 >> 3) * MMLF_TILE : Q0;
        const int wa = abl_reuse ? (gi0 & 7) : wv;
        // row maxima first (their loads are the oldest: the first counted wait below covers them)
        const float gathered = wave_operand_amax_gather(a, Qa, wa, lane);
        const float *p0 = a.in + (size_t)(Qa + 32 * wa) * a.cs_in + row_off;
        float4 raw[NSTEP][2][2];                                   // [step][row block][half]
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            const int t = k / NS, s_ = k % NS;                         // (full steps)
            const float *pk = k < 4 * NS ? p0 + (size_t)((t & 1) + (t >> 1) * a.P) * a.cs_in + 32 * s_ + 8 * q4 : p0 + rem_off;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                {
                    MMLF_OOB(OOB_IN, (long long)((pk + (size_t)(16 * mb) * a.cs_in + 4 * hf + 4) - a.in) * 4 > a.in_bytes);
                    raw[k][mb][hf] = reinterpret_cast<const float4 *>(pk + (size_t)(16 * mb) * a.cs_in)[hf];
                }
            __builtin_amdgcn_sched_barrier(0);          // step order: the counted waits below rely on it
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD > 63 ? 63 : NLOAD) : "memory");
        const float scale_a = wave_operand_scale(gathered);
        const float unscale_a = 1.f / scale_a;
        bf16x8 bq[3][2];
#pragma unroll
        for (int g0 = 0; g0 < 2; ++g0)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bq[g0][pl] = bfrag(0, g0, pl);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            // loads come back in order: step k's four are done once at most 4 (NSTEP - 1 - k) younger ones are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NSTEP - 1 - k) > 63 ? 63 : 4 * (NSTEP - 1 - k)) : "memory");
            bf16x8 asp[2][2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                unsigned hh[4], ll[4];
                float4 &r0 = raw[k][mb][0], &r1 = raw[k][mb][1];
                asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
                split2_pair_f16(r0.x, r0.y, scale_a, hh[0], ll[0]);
                split2_pair_f16(r0.z, r0.w, scale_a, hh[1], ll[1]);
                split2_pair_f16(r1.x, r1.y, scale_a, hh[2], ll[2]);
                split2_pair_f16(r1.z, r1.w, scale_a, hh[3], ll[3]);
                const u32x4_t vh = {hh[0], hh[1], hh[2], hh[3]}, vl = {ll[0], ll[1], ll[2], ll[3]};
                asp[mb][0] = __builtin_bit_cast(bf16x8, vh);
                asp[mb][1] = __builtin_bit_cast(bf16x8, vl);
            }
            // The split is inline assembly: the compiler's hazard recognizer does not see a vector write in front of the
            // matrix instruction that reads it (VALU write -> MFMA source read needs wait states), and its scheduler is
            // free to put an MFMA between the two row blocks' splits.  Fenced on both sides: without the fences the nop
            // moved up and column block 0 of whichever row block was split last came out with wrong low bits.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 4" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                constexpr int NB = NSTEP * G;
                const int idx = k * G + g;                      // block index in the (step, column block) stream
                if (idx + 2 < NB) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) bq[(idx + 2) % 3][pl] = bfrag((idx + 2) / G, (idx + 2) % G, pl);
                }
#define RS_TERM(pa, pb)                                                                                      \
    _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                         \
        acc[mb][g] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bq[idx % 3][pb]), \
                                                                 __builtin_bit_cast(f16x8, asp[mb][pa]), acc[mb][g], 0, 0, 0) \
                        : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, asp[mb][pa]),     \
                                                                 __builtin_bit_cast(f16x8, bq[idx % 3][pb]), acc[mb][g], 0, 0, 0)
                RS_TERM(1, 0);
                RS_TERM(0, 1);
                RS_TERM(0, 0);
#undef RS_TERM
            }
        }
        // (round 4 also measured this epilogue with ROW stores -- values through a wave-private LDS image, 16-byte stores of
        // consecutive addresses, 1 KB per wave instruction instead of 64-byte segments: 0.96-1.0 ms either way, removed)
        if constexpr (TR)
            conv_epilogue16_tr<G, EPI>(a, acc, Q0, wv, r16, q4, unscale_a, run_max, coef);
        else
            conv_epilogue16<G, EPI>(a, acc, Q0, wv, r16, q4, unscale_a, run_max,
                                    a.bn_partial ? stats_all + (size_t)w * NP * 2 : nullptr);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < G; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mb][nb][r] = 0.f;
    }
    if (a.out_amax) mmlf_amax_update(run_max, a.out_amax, blockIdx.x * 8 + w);      // one atomic per wave per launch
    if (a.bn_partial) {
        __syncthreads();                                        // orders the waves' sums
        for (int k = tid; k < 2 * a.n_true; k += 512) {
            const int ch = k % a.n_true, which = k / a.n_true;
            double t = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) t += stats_all[((size_t)ww * NP + ch) * 2 + which];
            a.bn_partial[((size_t)blockIdx.x * 2 + which) * a.n_true + ch] = t;
        }
    }
}

// 512 threads = 8 waves; tile = 256 positions x NT*32 output channels; wave w owns positions
// [32w, 32w+32) x all channels (NT accumulator tiles of 32x32).  K is walked in chunks of 8 input
// channels x 4 taps, double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4: no
// staging registers; the copy of chunk c+1 is in flight while chunk c is multiplied):
//   A (activations): [seg(2)][kh(2)][320] float4  -- seg 0 = positions Q0.., seg 1 = Q0+P..
//   B (weights)    : [tap(4)][kh(2)][NP]  float4
// One DMA wave-instruction ("piece") writes 64 consecutive float4 slots (1 KiB); its per-lane SOURCE
// address does the gather (A: one position per lane; B: linear).  A lane (i = lane&31, kh = lane>>5)
// reads ONE float4 = channels 4kh..4kh+3 of its position (A) or of its output channel (B): four K=2
// MFMA steps pairing channel s of half 0 with 4+s of half 1.
template <int NT>
__global__ __launch_bounds__(512) void conv4tap_kernel(ConvArgs a)
{
    constexpr int NP = NT * 32;
    constexpr int A_STRIDE = 320;               // float4 slots per (seg, kh) array: 5 DMA pieces
    constexpr int A_F4 = 4 * A_STRIDE;
    constexpr int B_F4 = 4 * 2 * NP;
    constexpr int BUF_F4 = A_F4 + B_F4;
    constexpr int N_A = 20;                     // DMA pieces for A per chunk
    constexpr int N_PIECES = N_A + B_F4 / 64;   // + B pieces
    constexpr int PER_WAVE = (N_PIECES + 7) / 8;
    static_assert(PER_WAVE <= 8, "piece schedule covers k = t and t + 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const long long Q0 = (long long)blockIdx.x * MMLF_TILE;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // per-lane DMA sources of this wave's pieces j = w, w+8, ...; chunk c adds c*step floats.
    // Destinations (float4 slot index inside a buffer) are wave-uniform.
    const float *src[PER_WAVE];
    int dst_f4[PER_WAVE];
    int step[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int j = w + 8 * k;
        if (j < N_A) {
            const int arr = j / 5, blk = j - 5 * arr;           // arr = seg*2 + kh
            int p = 64 * blk + lane;
            p = p < 256 ? p : 256;                               // lanes past the tile re-read position 256
            src[k] = a.in + (size_t)(Q0 + (arr >> 1) * a.P + p) * a.cs_in + 4 * (arr & 1);
            dst_f4[k] = arr * A_STRIDE + 64 * blk;
            step[k] = 8;
        } else {
            const int b = (j < N_PIECES ? j : N_PIECES - 1) - N_A;
            src[k] = a.wp + (size_t)(64 * b + lane) * 4;
            dst_f4[k] = A_F4 + 64 * b;
            step[k] = B_F4 * 4;
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;

    // One LDS-DMA piece.  Inline asm: with the builtin, hipcc drains vmcnt(0) before every LDS read
    // that might alias the in-flight copy; completion is awaited by hand before the barrier.
#define CONV_DMA_PIECE(c, buf, k)                                                                 \
    do {                                                                                          \
        if ((k) < PER_WAVE && w + 8 * (k) < N_PIECES) {                                           \
            const float *g_ = src[k] + (size_t)(c) * step[k];                                     \
            const unsigned d_ = lds_base + (unsigned)(((buf) * BUF_F4 + dst_f4[k]) * 16);         \
            unsigned keep_;                                                                       \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                \
                         "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                   \
                         : "=&s"(keep_) : "v"(g_), "s"(d_) : "memory");                          \
        }                                                                                         \
    } while (0)
#define CONV_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) CONV_DMA_PIECE(0, 0, k);
    CONV_DMA_WAIT();
    __syncthreads();

    for (int c = 0; c < a.nchunk; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < a.nchunk;
        const float4 *base = lds + buf * BUF_F4;
        const float4 *ap = base + kh * A_STRIDE + 32 * w + i;
        const float4 *bp = base + A_F4 + kh * NP + i;
        // Software pipeline over the 4 taps: ALL fragment reads of tap t+1 are issued, then the 4*NT
        // MFMAs of tap t run back to back while those reads land (sched_barrier pins the regions;
        // left alone, hipcc sinks each read to just before its first use).  The DMA pieces of chunk
        // c+1 are spread over the taps, and the two waves that share a SIMD (w, w+4) issue theirs at
        // different points of the MFMA stream.  MFMAs stay accumulator-major (4 dependent K=2 steps
        // per accumulator): measured faster here than the independent order.
        float4 a_cur = ap[0], b_cur[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt] = bp[32 * nt];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 a_nxt = a_cur, b_nxt[NT];
            if (t < 3) {
                a_nxt = ap[((t + 1) >> 1) * 2 * A_STRIDE + ((t + 1) & 1)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = bp[(t + 1) * 2 * NP + 32 * nt];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more && w < 4) { CONV_DMA_PIECE(c + 1, buf ^ 1, t); CONV_DMA_PIECE(c + 1, buf ^ 1, t + 4); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt == NT / 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && w >= 4) { CONV_DMA_PIECE(c + 1, buf ^ 1, t); CONV_DMA_PIECE(c + 1, buf ^ 1, t + 4); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[nt].y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur[nt].z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur[nt].w, acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t < 3) {
                a_cur = a_nxt;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_cur[nt] = b_nxt[nt];
            }
        }
        CONV_DMA_WAIT();   // this wave's pieces of chunk c+1 have landed ...
        __syncthreads();   // ... and so have everyone else's; all reads of buffer `buf` are done
    }
#undef CONV_DMA_PIECE
#undef CONV_DMA_WAIT

    conv_epilogue<NT>(a, acc, Q0, w, i, kh);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int mmlf_conv_cus(void) { return device_cus(); }

extern "C" int64_t mmlf_packed_filter_floats(int K, int N)
{
    const int nt = pick_nt(N);
    if (nt < 0 || K <= 0) return -1;
    return (int64_t)((K + 7) / 8) * 4 * 2 * (nt * 32) * 4;
}

extern "C" int mmlf_pack_filter(const float *w, float *packed, int Cout, int Cin, int variant, int dgrad,
                                void *stream)
{
    MMLF_CHECK_ARG(w && packed, "mmlf_pack_filter: null pointer");
    MMLF_CHECK_ARG(variant >= 0 && variant <= 2, "mmlf_pack_filter: bad variant %d", variant);
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int nt = pick_nt(N);
    MMLF_CHECK_ARG(nt > 0, "mmlf_pack_filter: N=%d not supported (max 288)", N);
    const int nchunk = (K + 7) / 8, NP = nt * 32;
    const long long total = (long long)nchunk * 32 * NP;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_filter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       variant, dgrad, nchunk, NP);
    return mmlf_launch_status("mmlf_pack_filter");
}

// What the ABI's contract says the caller's buffers hold (bytes behind the pointers as passed).  `out` may be a channel
// slice of a wider buffer (out = base + c_off with c_off + N_store <= cs_out): the bound below is what is left behind the
// LARGEST such offset, so it never cuts a store the contract allows and never exceeds the allocation.
static void conv_buffer_bytes(ConvArgs &a, const Grid &g)
{
    const long long alloc = grid_alloc_positions(g);
    a.out_bytes = (alloc * a.cs_out - (a.cs_out - a.n_store)) * 4;
    a.ref_bytes = a.ref ? alloc * a.cs_ref * 4 : 0;
    a.in_bytes = alloc * a.cs_in * 4;
    a.amax_n = amax_entries(g);
    a.mask_words = g.NQpad / MMLF_TILE * 4096;
}

template <int NT>
static int launch_conv(const ConvArgs &a, long long ntiles, hipStream_t st)
{
    constexpr size_t lds = 2 * (4 * 320 + 4 * 2 * NT * 32) * sizeof(float4);
    static PerDeviceOnce attr_once;
    if (attr_once.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(conv4tap_kernel<NT>), lds, "mmlf_conv2x2"); }))
        return 1;
    hipLaunchKernelGGL(conv4tap_kernel<NT>, dim3((unsigned)ntiles), dim3(512), lds, st, a);
    return mmlf_launch_status("mmlf_conv2x2");
}

extern "C" int mmlf_conv2x2(const float *in, int cs_in, int K, const float *packed, const float *bias, int N,
                            float *out, int cs_out, int N_store, int out_shift, int vh, int vw, int B, int H,
                            int W, int relu, const float *relu_ref, int cs_ref, void *stream)
{
    MMLF_CHECK_ARG(in && packed && out, "mmlf_conv2x2: null pointer");
    MMLF_CHECK_ARG(B > 0 && H > 0 && W > 0, "mmlf_conv2x2: bad shape B=%d H=%d W=%d", B, H, W);
    MMLF_CHECK_ARG(cs_in > 0 && cs_in % 8 == 0, "mmlf_conv2x2: cs_in=%d must be a multiple of 8", cs_in);
    MMLF_CHECK_ARG(K > 0 && (K + 7) / 8 * 8 == cs_in, "mmlf_conv2x2: K=%d does not match cs_in=%d", K, cs_in);
    const int nt = pick_nt(N);
    MMLF_CHECK_ARG(nt > 0, "mmlf_conv2x2: N=%d not supported (max 288)", N);
    MMLF_CHECK_ARG(N_store > 0 && N_store <= cs_out && N_store <= nt * 32,
                   "mmlf_conv2x2: N_store=%d vs cs_out=%d NP=%d", N_store, cs_out, nt * 32);
    Grid g = make_grid(B, H, W);
    MMLF_CHECK_ARG(out_shift >= 0 && out_shift <= g.P + 1, "mmlf_conv2x2: out_shift=%d", out_shift);
    MMLF_CHECK_ARG(!relu_ref || cs_ref >= N_store, "mmlf_conv2x2: cs_ref=%d < N_store", cs_ref);
    ConvArgs a = {};
    a.in = in; a.wp = packed; a.bias = bias; a.out = out; a.ref = relu_ref;
    a.NQ = g.NQ; a.cs_in = cs_in; a.nchunk = cs_in / 8; a.cs_out = cs_out; a.n_store = N_store; a.n_true = N;
    a.out_shift = out_shift; a.vh = vh; a.vw = vw; a.P = g.P; a.G = g.G; a.relu = relu; a.cs_ref = cs_ref;
    a.divP = make_magic((unsigned)g.P); a.divR = make_magic((unsigned)g.R); a.R = g.R;
    a.relu_mask_out = nullptr; a.relu_mask_in = nullptr;
    conv_buffer_bytes(a, g);
    MMLF_CHECK_ARG(g.NQpad + g.P + 64 < (1ll << 31), "mmlf_conv2x2: batch x image too large for 32-bit grid positions");
    const long long ntiles = g.NQpad / MMLF_TILE;
    hipStream_t st = (hipStream_t)stream;
    switch (nt) {
    case 1: return launch_conv<1>(a, ntiles, st);
    case 3: return launch_conv<3>(a, ntiles, st);
    case 4: return launch_conv<4>(a, ntiles, st);
    default: return launch_conv<9>(a, ntiles, st);
    }
}

// ------------------------------------------------------------------ split-bf16 ("bf16x6") entry points
// Packed column count NP of the split forward/dgrad kernels: 32, 80, then 16-column granularity up to
// 128, then 288.  Packing and launch pick by this one rule, so the layouts always agree.
static inline int x6_np(int N)
{
    if (N <= 0 || N > 288) return -1;
    if (N <= 32) return 32;
    if (N <= 80) return 80;
    if (N <= 128) return (N + 15) / 16 * 16;
    return 288;
}

extern "C" int64_t mmlf_packed_filter_split_bytes(int K, int N)
{
    const int np = x6_np(N);
    if (np < 0 || K <= 0) return -1;
    return (int64_t)((K + 7) / 8) * 12 * np * 16;
}

extern "C" int mmlf_pack_filter_split(const float *w, void *packed, int Cout, int Cin, int variant, int dgrad,
                                      void *stream)
{
    MMLF_CHECK_ARG(w && packed, "mmlf_pack_filter_split: null pointer");
    MMLF_CHECK_ARG(variant >= 0 && variant <= 2, "mmlf_pack_filter_split: bad variant %d", variant);
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int NP = x6_np(N);
    MMLF_CHECK_ARG(NP > 0, "mmlf_pack_filter_split: N=%d not supported (max 288)", N);
    const int nchunk = (K + 7) / 8;
    const long long total = (long long)nchunk * 32 * NP;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_filter_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short *)packed, Cout, Cin, variant, dgrad, nchunk, NP);
    return mmlf_launch_status("mmlf_pack_filter_split");
}

// persistent launches: one workgroup per CU (two for the narrow eight-wave variants), each walks tiles b, b+grid, ...
static long long conv_split_blocks(int G, long long ntiles, int nw = 8)
{
    const long long grid = (G <= 6 && nw == 8 ? 2ll : 1ll) * device_cus();
    return grid > ntiles ? ntiles : grid;
}
// The sixteen-wave variant (512-position tiles) serves the 80-column f16 kernels on pitches whose 513 + P position
// window fits the 640-slot activation buffer (MMLF_CONV_NW16=0 turns it off).
static bool conv_sixteen_waves(int planes, int np, const Grid &g)
{
    static const int on = [] { const char *e = getenv("MMLF_CONV_NW16"); return e ? atoi(e) : 1; }();
    return on && planes == 2 && np == 80 && g.P + 513 <= 640;
}

// The transposed epilogue (conv_epilogue16_tr) serves the launch kinds plain / ReLU / ReLU + mask-out / mask-in when the
// output rows can take 16-byte stores.  MMLF_CONV_TR=0 turns it off for the whole process (A/B; the ReLU mask words then
// have the other layout, for their producer and their consumer alike).
static int conv_tr_mode()
{
    static const int on = [] { const char *e = getenv("MMLF_CONV_TR"); return e ? atoi(e) : 1; }();
    return on;
}
static bool conv_tr_enabled() { return conv_tr_mode() != 0; }
static bool conv_tr_any_shape() { return conv_tr_mode() == 2; }     // 2: also where the geometry rule says no (A/B)
static bool conv_tr_fits(const ConvArgs &a)
{
    return a.n_store % 4 == 0 && a.cs_out % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
}

template <int G, int PL, int EPI, bool TR = false>
static int launch_conv_x6s_epi(const ConvArgs &a, long long ntiles, hipStream_t st)
{
    constexpr size_t lds_pipe = 2 * (2 * 640 + 4 * PL * G * 16) * sizeof(float4);
    constexpr size_t lds_stats = 8 * (G * 16) * 2 * sizeof(double);
    constexpr size_t lds_coef = 2 * (G * 16) * sizeof(float);
    static PerDeviceOnce attr_once;
    if (attr_once.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(conv4tap_x6s_kernel<G, PL, EPI, 8, TR>),
                                                 lds_pipe + (TR ? lds_coef : PL == 2 ? lds_stats : 0), "mmlf_conv2x2_h2"); }))
        return 1;
    if constexpr (PL == 2 && G == 5) {
        if (a.nw == 16) {
            constexpr size_t lds_stats16 = 16 * (G * 16) * 2 * sizeof(double);
            constexpr size_t lds_pipe16 = MMLF_RING16 * (2 * 640 + 4 * PL * G * 16) * sizeof(float4);
            static PerDeviceOnce attr_once16;
            if (attr_once16.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(conv4tap_x6s_kernel<G, PL, EPI, 16, TR>),
                                                           lds_pipe16 + (TR ? lds_coef : lds_stats16), "mmlf_conv2x2_h2(16 waves)"); }))
                return 1;
            const size_t lds16 = lds_pipe16 + (TR ? lds_coef : a.bn_partial ? lds_stats16 : 0);
            const long long grid16 = conv_split_blocks(G, ntiles, 16);
            hipLaunchKernelGGL((conv4tap_x6s_kernel<G, PL, EPI, 16, TR>), dim3((unsigned)grid16), dim3(1024), lds16, st, a, (int)ntiles);
            return mmlf_launch_status("mmlf_conv2x2_h2(16 waves)");
        }
    }
    const size_t lds = lds_pipe + (TR ? lds_coef : a.bn_partial ? lds_stats : 0);
    const long long grid = conv_split_blocks(G, ntiles);
    hipLaunchKernelGGL((conv4tap_x6s_kernel<G, PL, EPI, 8, TR>), dim3((unsigned)grid), dim3(512), lds, st, a, (int)ntiles);
    return mmlf_launch_status(PL == 3 ? "mmlf_conv2x2_split" : "mmlf_conv2x2_h2");
}

// the register-streamed narrow kernel (conv4tap_rs_kernel) serves 80 packed columns over 4 or 9 input chunks (27 -> 70,
// 70 -> 70) in the f16 split.
// Where it runs (measured, profiles/r04_ab_bench_rs.log): on 96x96 training patches the two kernels take the same time
// (0.94 ms per 70 -> 70 launch at bs=512 either way; whole step 1122-1127 patches/s both), so the tiled sixteen-wave
// kernel keeps those launches; on pitches where that variant's 512 + P position window does not fit (full frames: the
// 70 members of a 512x512 ESE scene) the tiled kernel falls back to eight waves and two window segments, and the
// register-streamed kernel is 4.7 % faster end to end (0.436 vs 0.457 s per scene).  MMLF_CONV_RS=1 / 0 forces it on /
// off for every eligible launch (A/B).
static int conv_rs_mode()
{
    static const int mode = [] { const char *e = getenv("MMLF_CONV_RS"); return e ? atoi(e) : -1; }();
    return mode;
}
static bool conv_rs_shape(int planes, int np, int nchunk, int nw)
{
    if (!(planes == 2 && np == 80 && (nchunk == 9 || nchunk == 4))) return false;
    const int mode = conv_rs_mode();
    return mode < 0 ? nw == 8 : mode != 0;
}
static long long conv_rs_blocks(long long ngroups)
{
    const long long grid = device_cus(), need = (ngroups + 7) / 8;
    return grid > need ? need : grid;
}
template <int G, int NCH, int EPI, bool TR = false>
static int launch_conv_rs_epi(const ConvArgs &a, long long ngroups, hipStream_t st)
{
    constexpr size_t lds = (size_t)NCH * 2 * 4 * (G * 16) * 16 + (TR ? 2 * (G * 16) * sizeof(float) : 8 * (G * 16) * 2 * sizeof(double));
    static PerDeviceOnce attr_once;
    if (attr_once.run([] { return mmlf_allow_lds(reinterpret_cast<const void *>(conv4tap_rs_kernel<G, NCH, EPI, TR>), lds, "mmlf_conv2x2_h2(register-streamed)"); }))
        return 1;
    hipLaunchKernelGGL((conv4tap_rs_kernel<G, NCH, EPI, TR>), dim3((unsigned)conv_rs_blocks(ngroups)), dim3(512), lds, st, a, (int)ngroups);
    return mmlf_launch_status("mmlf_conv2x2_h2(register-streamed)");
}
// launch kind -> epilogue build.  tr: the transposed form, for the kinds that have one
// (TR_PLAIN: whether the plain kind has one too -- on the 80-column kernels it measured 3 % slower: profiles/r05_kbench_tr.log;
//  TR_SHAPE: a condition on the launch geometry -- the 288-column tiled kernel takes the transposed form on small pitches only:
//  with two-segment activation windows (full frames, pitch > 383) its ReLU launches measured 26.7 ms against 24.3 in the other
//  orientation, profiles/r05_ese_tr_kernel_stats.log, while at training pitches they are 0.5-2 % faster.  The condition depends
//  on (B, H, W) and the kernel alone, so a mask's producer and its consumer always agree.)
#define MMLF_CONV_KIND_SWITCH(LAUNCH, TR_PLAIN, TR_SHAPE, ...)                                                                              \
    do {                                                                                                                \
        const int kind = (a.relu ? EPI_RELU : 0) | (a.bn_partial ? EPI_STATS : 0) | (a.relu_mask_in ? EPI_BITS_IN : 0) | \
                         (a.ref ? EPI_REF_IN : 0) | (a.relu_mask_out ? EPI_MASK_OUT : 0);                               \
        const bool tr = conv_tr_enabled() && (TR_SHAPE) && conv_tr_fits(a);                                             \
        if ((a.relu_mask_in || a.relu_mask_out) && conv_tr_enabled() && (TR_SHAPE) && !tr)                              \
            return mmlf_fail("mmlf_conv2x2_h2: ReLU mask words need n_store %% 4 == 0 and a 16-byte aligned output");   \
        switch (kind) {                                                                                                 \
        case EPI_PLAIN: return tr && TR_PLAIN ? LAUNCH<__VA_ARGS__, EPI_PLAIN, true>(a, n, st) : LAUNCH<__VA_ARGS__, EPI_PLAIN>(a, n, st); \
        case EPI_RELU: return tr ? LAUNCH<__VA_ARGS__, EPI_RELU, true>(a, n, st) : LAUNCH<__VA_ARGS__, EPI_RELU>(a, n, st); \
        case EPI_RELU | EPI_MASK_OUT:                                                                                   \
            return tr ? LAUNCH<__VA_ARGS__, EPI_RELU | EPI_MASK_OUT, true>(a, n, st) : LAUNCH<__VA_ARGS__, EPI_RELU | EPI_MASK_OUT>(a, n, st); \
        case EPI_BITS_IN: return tr ? LAUNCH<__VA_ARGS__, EPI_BITS_IN, true>(a, n, st) : LAUNCH<__VA_ARGS__, EPI_BITS_IN>(a, n, st); \
        case EPI_STATS: return LAUNCH<__VA_ARGS__, EPI_STATS>(a, n, st);                                                \
        default: break;                                                                                                 \
        }                                                                                                               \
    } while (0)
template <int G, int NCH>
static int launch_conv_rs(const ConvArgs &a, long long n, hipStream_t st)
{
    MMLF_CONV_KIND_SWITCH(launch_conv_rs_epi, false, true, G, NCH);
    return launch_conv_rs_epi<G, NCH, EPI_GENERIC>(a, n, st);
}

// the launch kinds of a training step get their own epilogue build on the hot shapes (f16 split, 70- and 280-wide
// layers); every other combination of options runs the generic one
template <int G, int PL>
static int launch_conv_x6s(const ConvArgs &a, long long ntiles, hipStream_t st)
{
    if constexpr (PL == 2 && G == 5) {
        if (conv_rs_shape(PL, G * 16, a.nchunk, a.nw)) {
            const long long ngroups = ntiles * a.nw;            // 32-position groups of the padded grid
            return a.nchunk == 9 ? launch_conv_rs<G, 9>(a, ngroups, st) : launch_conv_rs<G, 4>(a, ngroups, st);
        }
    }
    if constexpr (PL == 2 && (G == 5 || G == 18)) {
        const long long n = ntiles;
        if (a.ref && !a.relu && !a.bn_partial && !a.relu_mask_in && !a.relu_mask_out)
            return launch_conv_x6s_epi<G, PL, EPI_REF_IN>(a, n, st);
        MMLF_CONV_KIND_SWITCH(launch_conv_x6s_epi, G == 18, G != 18 || a.seg_delta == 0 || conv_tr_any_shape(), G, PL);
    }
    // (other widths: the generic build, whose mask words have conv_epilogue16's layout for producer and consumer alike)
    return launch_conv_x6s_epi<G, PL, EPI_GENERIC>(a, ntiles, st);
}

template <int PL>
static int launch_conv_split(int np, const ConvArgs &a, long long ntiles, hipStream_t st)
{
    switch (np) {
    case 32: return launch_conv_x6s<2, PL>(a, ntiles, st);
    case 80: return launch_conv_x6s<5, PL>(a, ntiles, st);
    case 96: return launch_conv_x6s<6, PL>(a, ntiles, st);
    case 112: return launch_conv_x6s<7, PL>(a, ntiles, st);
    case 128: return launch_conv_x6s<8, PL>(a, ntiles, st);
    default: return launch_conv_x6s<18, PL>(a, ntiles, st);
    }
}

static int conv_split_impl(const char *who, int planes, const float *in, int cs_in, int K, const void *packed,
                           const float *bias, int N, float *out, int cs_out, int N_store, int out_shift, int vh,
                           int vw, int B, int H, int W, int relu, const float *relu_ref, int cs_ref,
                           const float *in_amax, float *out_amax, double *bn_partial, void *stream,
                           unsigned *relu_mask_out = nullptr, const unsigned *relu_mask_in = nullptr)
{
    MMLF_CHECK_ARG(in && packed && out, "%s: null pointer", who);
    MMLF_CHECK_ARG(!(relu_ref && relu_mask_in), "%s: relu_ref and relu_mask_in are alternatives", who);
    MMLF_CHECK_ARG(B > 0 && H > 0 && W > 0, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    MMLF_CHECK_ARG(cs_in > 0 && cs_in % 8 == 0, "%s: cs_in=%d must be a multiple of 8", who, cs_in);
    MMLF_CHECK_ARG(K > 0 && (K + 7) / 8 * 8 == cs_in, "%s: K=%d does not match cs_in=%d", who, K, cs_in);
    const int np = x6_np(N);
    MMLF_CHECK_ARG(np > 0, "%s: N=%d not supported (max 288)", who, N);
    MMLF_CHECK_ARG(N_store > 0 && N_store <= cs_out && N_store <= np, "%s: N_store=%d vs cs_out=%d NP=%d", who,
                   N_store, cs_out, np);
    Grid g = make_grid(B, H, W);
    MMLF_CHECK_ARG(out_shift >= 0 && out_shift <= g.P + 1, "%s: out_shift=%d", who, out_shift);
    MMLF_CHECK_ARG(!relu_ref || cs_ref >= N_store, "%s: cs_ref=%d < N_store", who, cs_ref);
    MMLF_CHECK_ARG((long long)cs_in * 4 * 64 < (1ll << 31), "%s: cs_in too large", who);
    MMLF_CHECK_ARG(planes == 3 || in_amax, "%s: the f16 split needs the input's max |x|", who);
    ConvArgs a = {};
    a.in = in; a.wp = reinterpret_cast<const float *>(packed); a.bias = bias; a.out = out; a.ref = relu_ref;
    a.NQ = g.NQ; a.cs_in = cs_in; a.nchunk = cs_in / 8; a.cs_out = cs_out; a.n_store = N_store; a.n_true = N;
    a.out_shift = out_shift; a.vh = vh; a.vw = vw; a.P = g.P; a.G = g.G; a.relu = relu; a.cs_ref = cs_ref;
    a.in_amax = in_amax; a.out_amax = out_amax; a.bn_partial = bn_partial;
    a.relu_mask_out = relu_mask_out; a.relu_mask_in = relu_mask_in;
    conv_buffer_bytes(a, g);
    MMLF_CHECK_ARG(!bn_partial || (planes == 2 && N_store >= N), "%s: BatchNorm statistics need the f16 split path", who);
    // the f16-packed filter ends with its columns' unscale factors
    a.w_unscale = planes == 2 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(packed) +
                                                                  (size_t)(cs_in / 8) * 8 * np * 16)
                              : nullptr;
    a.divP = make_magic((unsigned)g.P); a.divR = make_magic((unsigned)g.R); a.R = g.R;
    MMLF_CHECK_ARG(g.NQpad + g.P + 64 < (1ll << 31), "%s: batch x image too large for 32-bit grid positions", who);
    a.nw = conv_sixteen_waves(planes, np, g) ? 16 : 8;
    const int tile = 32 * a.nw;
    if (g.P + tile + 1 <= 640) {   // one contiguous window of tile + 1 + P positions
        a.a_pieces = a.a_per_seg = (g.P + tile + 1 + 31) / 32; a.seg_slot = g.P; a.seg_delta = 0;
    } else {                  // two 320-slot segments: rows y and y+1, positions Q0 .. Q0 + 256 of each
        a.a_per_seg = 9; a.a_pieces = 18; a.seg_slot = 320; a.seg_delta = g.P - 320;
    }
    a.a_tail = (a.seg_delta ? 257 : g.P + tile + 1) - 32 * (a.a_per_seg - 1);
    {   // round-6 proxy: the caller passes a chunk-blocked copy of the input (see ConvArgs::a_blocked); tiled 80-column kernel only
        const char *e = getenv("MMLF_PROXY_BLOCKED_A");
        a.a_blocked = (e && atoi(e) && planes == 2 && np == 80 && a.seg_delta == 0 && !conv_rs_shape(planes, np, a.nchunk, a.nw)) ? 1 : 0;
        MMLF_CHECK_ARG(!(e && atoi(e)) || a.a_blocked, "%s: MMLF_PROXY_BLOCKED_A is set but this launch has no chunk-blocked form", who);
    }
    const long long ntiles = g.NQpad / tile;
    hipStream_t st = (hipStream_t)stream;
    return planes == 3 ? launch_conv_split<3>(np, a, ntiles, st) : launch_conv_split<2>(np, a, ntiles, st);
}

extern "C" int mmlf_conv2x2_split(const float *in, int cs_in, int K, const void *packed, const float *bias, int N,
                                  float *out, int cs_out, int N_store, int out_shift, int vh, int vw, int B, int H,
                                  int W, int relu, const float *relu_ref, int cs_ref, void *stream)
{
    return conv_split_impl("mmlf_conv2x2_split", 3, in, cs_in, K, packed, bias, N, out, cs_out, N_store, out_shift,
                           vh, vw, B, H, W, relu, relu_ref, cs_ref, nullptr, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------ f16 2-way split ("f16x3") entry points
extern "C" int64_t mmlf_packed_filter_h2_bytes(int K, int N)
{
    const int np = x6_np(N);
    if (np < 0 || K <= 0) return -1;
    return (int64_t)((K + 7) / 8) * 8 * np * 16 + (int64_t)np * 4;      // + 1 / scale of every packed column
}

extern "C" int mmlf_pack_filter_h2(const float *w, void *packed, int Cout, int Cin, int variant, int dgrad,
                                   void *stream)
{
    MMLF_CHECK_ARG(w && packed, "mmlf_pack_filter_h2: null pointer");
    MMLF_CHECK_ARG(variant >= 0 && variant <= 2, "mmlf_pack_filter_h2: bad variant %d", variant);
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int NP = x6_np(N);
    MMLF_CHECK_ARG(NP > 0, "mmlf_pack_filter_h2: N=%d not supported (max 288)", N);
    const int nchunk = (K + 7) / 8;
    // the per-column unscale factors sit behind the packed planes
    float *tail = reinterpret_cast<float *>(reinterpret_cast<char *>(packed) + (size_t)nchunk * 8 * NP * 16);
    hipLaunchKernelGGL(pack_filter_h2_kernel, dim3(NP), dim3(256), 0, (hipStream_t)stream, w, (unsigned short *)packed,
                       Cout, Cin, variant, dgrad, nchunk, NP, tail);
    return mmlf_launch_status("mmlf_pack_filter_h2");
}

extern "C" int mmlf_packed_filter_h2_columns(int N) { return x6_np(N); }

extern "C" int mmlf_pack_filters_h2(const mmlf_pack_desc *table, int n, int total_columns, void *stream)
{
    MMLF_CHECK_ARG(table && n > 0 && total_columns > 0, "mmlf_pack_filters_h2: empty table");
    hipLaunchKernelGGL(pack_filters_h2_kernel, dim3((unsigned)total_columns), dim3(256), 0, (hipStream_t)stream, table, n);
    return mmlf_launch_status("mmlf_pack_filters_h2");
}

extern "C" int mmlf_conv2x2_h2(const float *in, int cs_in, int K, const void *packed, const float *bias, int N,
                               float *out, int cs_out, int N_store, int out_shift, int vh, int vw, int B, int H,
                               int W, int relu, const float *relu_ref, int cs_ref, const float *in_amax,
                               float *out_amax, double *bn_partial, uint32_t *relu_mask_out,
                               const uint32_t *relu_mask_in, void *stream)
{
    return conv_split_impl("mmlf_conv2x2_h2", 2, in, cs_in, K, packed, bias, N, out, cs_out, N_store, out_shift, vh,
                           vw, B, H, W, relu, relu_ref, cs_ref, in_amax, out_amax, bn_partial, stream, relu_mask_out,
                           relu_mask_in);
}

extern "C" int64_t mmlf_relu_mask_words(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return make_grid(B, H, W).NQpad / MMLF_TILE * 4096;        // [tile][8 waves][8 rows][64 lanes]
}

extern "C" int mmlf_conv2x2_blocks(int K, int N, int B, int H, int W)
{
    const int np = x6_np(N);
    if (np < 0 || K <= 0 || B <= 0 || H <= 0 || W <= 0) return -1;
    const Grid g = make_grid(B, H, W);
    const int nw = conv_sixteen_waves(2, np, g) ? 16 : 8;
    if (conv_rs_shape(2, np, (K + 7) / 8, nw)) return (int)conv_rs_blocks(g.NQpad / 32);
    return (int)conv_split_blocks(np / 16, g.NQpad / (32 * nw), nw);
}

// ---------------------------------------------------------------------------------------------
// "thin" convolutions: N <= 2 output channels over a wide input (the first convolution of the BASE / UPR head,
// reference feed_forward.py:179-182: 280 -> 1 | 2).  On the MFMA kernels such a layer pays for 32 columns to
// use one or two and runs for 2.9 ms forward and 6.3 ms in the weight gradient at bs=512; it is a matrix-VECTOR
// product, bound by reading the input once: plain float32 FMAs (no operand split), one wave per grid position.
//   forward : part[p][t][o] = sum_c in[p][c] * W[o][c][tap t]        (thin_rowdot_kernel, reads `in` once)
//             out[q + shift][o] = valid(q) ? act(b[o] + sum_t part[q + off_t][t][o]) : 0   (thin_combine_kernel)
//   weights : dW[o][c][t] += sum_p in[p][c] * g[p - off_t + g_shift][o]                    (thin_wgrad_kernel)
// ---------------------------------------------------------------------------------------------

// A wave takes 64 consecutive positions.  Channels go through LDS in slices of 32: the slice is read from global
// memory in whole 128-byte row segments (lane = (row, float4)), and consumed with lane = position, so the dot
// products need no cross-lane reduction (eight of them per position through ds_bpermute made this kernel LDS-bound);
// the filter values of a slice are wave-uniform and come through the scalar cache.
__global__ __launch_bounds__(256) void thin_rowdot_kernel(ThinArgs a)
{
    __shared__ float tile[4][64][33];                     // [wave][position][channel of the slice] (+1: conflict-free columns)
    __shared__ float4 wt[4][32][2];                       // [wave][channel of the slice][o] = the four taps of W[o][c]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const long long nwaves = gridDim.x * 4ll;
    const int nslice = (a.C + 31) / 32;
    for (long long p0 = 64 * wave; p0 < a.npos; p0 += 64 * nwaves) {
        float acc[4][THIN_MAXN];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < THIN_MAXN; ++o) acc[t][o] = 0.f;
        for (int sl = 0; sl < nslice; ++sl) {
            // 64 rows x 8 float4: lane (r8 = lane >> 3, f = lane & 7) loads rows r8, r8 + 8, ...
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = (lane >> 3) + 8 * j, f = lane & 7;
                const long long p = p0 + r < a.npos ? p0 + r : a.npos - 1;
                const int c = 32 * sl + 4 * f;
                v[j] = *reinterpret_cast<const float4 *>(a.in + (size_t)p * a.cs_in + (c < a.cs_in ? c : 0));
                if (c >= a.cs_in) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {   // this slice's filter values: lane (c = lane & 31, o = lane >> 5), taps in packed order
                const int c = 32 * sl + (lane & 31), o = lane >> 5;
                float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < a.C && o < a.N) {
                    const float *wp = a.w + ((size_t)o * a.C + c) * 4;
                    w4 = make_float4(wp[master_tap(0, a.variant)], wp[master_tap(1, a.variant)], wp[master_tap(2, a.variant)],
                                     wp[master_tap(3, a.variant)]);
                }
                wt[wv][lane & 31][o] = w4;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = (lane >> 3) + 8 * j, f = lane & 7;
                tile[wv][r][4 * f + 0] = v[j].x; tile[wv][r][4 * f + 1] = v[j].y;
                tile[wv][r][4 * f + 2] = v[j].z; tile[wv][r][4 * f + 3] = v[j].w;
            }
            // (one wave reads what it wrote: program order + the compiler's lgkmcnt waits suffice)
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {                // channels past C carry zero weights
                const float x = tile[wv][lane][k];
                const float4 w0 = wt[wv][k][0], w1 = wt[wv][k][1];
                acc[0][0] = fmaf(x, w0.x, acc[0][0]); acc[1][0] = fmaf(x, w0.y, acc[1][0]);
                acc[2][0] = fmaf(x, w0.z, acc[2][0]); acc[3][0] = fmaf(x, w0.w, acc[3][0]);
                acc[0][1] = fmaf(x, w1.x, acc[0][1]); acc[1][1] = fmaf(x, w1.y, acc[1][1]);
                acc[2][1] = fmaf(x, w1.z, acc[2][1]); acc[3][1] = fmaf(x, w1.w, acc[3][1]);
            }
        }
        const long long p = p0 + lane;
        if (p < a.npos) {
            float4 *dst = reinterpret_cast<float4 *>(a.part + (size_t)p * 4 * THIN_MAXN);
            dst[0] = make_float4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
            dst[1] = make_float4(acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
        }
    }
}

// one workgroup per DESTINATION grid row: writes every position of the row (zeros outside the valid extent)
__global__ __launch_bounds__(128) void thin_combine_kernel(ThinArgs a)
{
    const int drow = blockIdx.x;                          // global grid row of the destination
    float mx = 0.f;
    for (int x = threadIdx.x; x < a.P; x += blockDim.x) {
        const long long d = (long long)drow * a.P + x, q = d - a.out_shift;
        float v[THIN_MAXN] = {0.f, 0.f};
        bool valid = false;
        if (q >= 0 && q < a.NQ) {
            const unsigned row = fastdiv((unsigned)q, a.divP);
            const int qx = (int)(q - (long long)row * a.P), qy = (int)(row - fastdiv(row, a.divR) * a.R);
            valid = qy < a.vh && qx < a.vw;
        }
        if (valid) {
            const int offs[4] = {0, 1, a.P, a.P + 1};
#pragma unroll
            for (int o = 0; o < THIN_MAXN; ++o) {
                float s = (a.bias && o < a.N) ? a.bias[o] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) s += a.part[((size_t)(q + offs[t]) * 4 + t) * THIN_MAXN + o];
                if (a.relu) s = fmaxf(s, 0.f);
                v[o] = o < a.N ? s : 0.f;
            }
        }
        float *dst = a.out + (size_t)d * a.cs_out;
        for (int o = 0; o < a.cs_out; ++o) dst[o] = o < THIN_MAXN ? v[o] : 0.f;
        mx = fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1])));
    }
    if (a.out_amax) mmlf_amax_update_row(mx, a.out_amax, drow);
}

extern "C" int64_t mmlf_conv2x2_thin_workspace_floats(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    const Grid g = make_grid(B, H, W);
    return (g.NQ + g.P + 2) * 4 * THIN_MAXN;
}

extern "C" int mmlf_conv2x2_thin(const float *in, int cs_in, int K, const float *w_oihw, const float *bias, int N,
                                 float *out, int cs_out, int out_shift, int vh, int vw, int B, int H, int W, int relu,
                                 int variant, float *workspace, float *out_amax, void *stream)
{
    MMLF_CHECK_ARG(in && w_oihw && out && workspace, "mmlf_conv2x2_thin: null pointer");
    MMLF_CHECK_ARG(N >= 1 && N <= THIN_MAXN && K >= 1 && K <= cs_in && cs_in % 4 == 0 && cs_in <= 512,
                   "mmlf_conv2x2_thin: N=%d K=%d cs_in=%d (N <= 2, cs_in <= 512)", N, K, cs_in);
    MMLF_CHECK_ARG(B > 0 && H > 0 && W > 0 && cs_out >= THIN_MAXN && variant >= 0 && variant <= 2, "mmlf_conv2x2_thin: bad shape");
    const Grid g = make_grid(B, H, W);
    MMLF_CHECK_ARG(out_shift >= 0 && out_shift <= g.P + 1 && g.NQpad + 2 * g.P + 64 < (1ll << 31), "mmlf_conv2x2_thin: out_shift / size");
    ThinArgs a = {};
    a.in = in; a.w = w_oihw; a.bias = bias; a.part = workspace; a.out = out; a.out_amax = out_amax;
    a.NQ = g.NQ; a.npos = g.NQ + g.P + 2; a.cs_in = cs_in; a.C = K; a.N = N; a.cs_out = cs_out; a.out_shift = out_shift;
    a.vh = vh; a.vw = vw; a.P = g.P; a.R = g.R; a.relu = relu; a.variant = variant;
    a.divP = make_magic((unsigned)g.P); a.divR = make_magic((unsigned)g.R);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(thin_rowdot_kernel, dim3(4 * device_cus()), dim3(256), 0, st, a);
    hipLaunchKernelGGL(thin_combine_kernel, dim3((unsigned)(B * g.R)), dim3(128), 0, st, a);
    return mmlf_launch_status("mmlf_conv2x2_thin");
}

// ---------------------------------------------------------------------------------------------
// what this binary is (round 5): every build switch that changes behaviour, in one string; the loader refuses a
// result-changing build unless told otherwise (mmlf_amd/_lib.py), bench.py prints the string in its line
// ---------------------------------------------------------------------------------------------
#ifndef MMLF_GIT_HASH
#define MMLF_GIT_HASH "unknown"
#endif
#ifdef MMLF_BOUNDS_DEBUG
#define MMLF_BOUNDS_DEBUG_VALUE 1
#else
#define MMLF_BOUNDS_DEBUG_VALUE 0
#endif
extern "C" int mmlf_build_is_ablation(void) { return (MMLF_ABL_TERMS != 3 || MMLF_ABL_WGRAD_STAGE != 0 || MMLF_ABL_RS_FUSE != 0) ? 1 : 0; }
extern "C" const char *mmlf_build_info(void)
{
    static char text[448];
    static std::once_flag once;
    std::call_once(once, [] {
        snprintf(text, sizeof(text),
                 "abi=%d git=%s src=%s MMLF_ABL_TERMS=%d MMLF_ABL_WGRAD_STAGE=%d MMLF_ABL_RS_FUSE=%d MMLF_GRID_PAD_W=%d MMLF_GRID_PAD_H=%d MMLF_RING16=%d "
                 "MMLF_WGRAD_EARLY=%d MMLF_WGRADN_CLAMP=%d MMLF_WGRAD_ZEROPAD=%d MMLF_BOUNDS_DEBUG=%d ablation=%d",
                 MMLF_ABI_VERSION, MMLF_GIT_HASH, MMLF_SRC_HASH, MMLF_ABL_TERMS, MMLF_ABL_WGRAD_STAGE, MMLF_ABL_RS_FUSE, MMLF_GRID_PAD_W, MMLF_GRID_PAD_H,
                 MMLF_RING16, MMLF_WGRAD_EARLY, MMLF_WGRADN_CLAMP, MMLF_WGRAD_ZEROPAD, MMLF_BOUNDS_DEBUG_VALUE, mmlf_build_is_ablation());
    });
    return text;
}

// ---------------------------------------------------------------------------------------------
// Bounds audit (round 5).  For one launch of the given shape: the END (largest byte offset + 1) of what the launch may
// read or write behind each pointer argument, derived here from the same launch geometry the entry points compute
// (window pieces, tile counts, grids).  tests/test_bounds_audit.py holds every end against what the size queries of this
// ABI tell the caller to allocate, over a sweep of shapes; the -DMMLF_BOUNDS_DEBUG build counts violations on the GPU.
// Derivations (positions are grid positions q; a buffer of channel stride cs holds alloc = NQpad + P + 72 of them):
//  conv, f16 split: the activation window of the LAST tile (first position NQpad - TILE) ends at position
//    NQpad + P (one contiguous window of TILE + 1 + P positions, or two segments whose second one ends at P + 256);
//    the last piece fetches a_tail positions only, the other lanes re-fetch the last of those.  The register-streamed
//    kernel reads positions Q0 + 32 w + 16 mb + r16 + {0, 1, P, P + 1}: the same end.  Output: position q + out_shift,
//    q < NQpad, n_store channels; the ReLU reference likewise.  Row maxima: rows up to the one behind the last position
//    read (input) / written (output).  Mask words: [tile][8][8][64].  Statistics: [workgroup][2][N] doubles.
//  weight gradient: chunk c stages in[32 c .. 32 c + 32 + P] and g[32 c + g_shift .. + 31], c < NQpad / 32.
// ---------------------------------------------------------------------------------------------
extern "C" int mmlf_audit_conv_h2(int cs_in, int K, int N, int cs_out, int N_store, int out_shift, int cs_ref, int B, int H, int W,
                                  int64_t *ends /* [MMLF_AUDIT_CONV_N] */)
{
    const int np = x6_np(N);
    MMLF_CHECK_ARG(np > 0 && K > 0 && (K + 7) / 8 * 8 == cs_in && B > 0 && H > 0 && W > 0 && ends, "mmlf_audit_conv_h2: bad argument");
    const Grid g = make_grid(B, H, W);
    const int nw = conv_sixteen_waves(2, np, g) ? 16 : 8;
    const int tile = 32 * nw, nchunk = cs_in / 8;
    // the last activation position a launch fetches
    long long last_in;
    if (conv_rs_shape(2, np, nchunk, nw)) {
        last_in = (g.NQpad - MMLF_TILE) + 32 * 7 + 16 + 15 + g.P + 1;
    } else if (g.P + tile + 1 <= 640) {
        const int pieces = (g.P + tile + 1 + 31) / 32, tail = (g.P + tile + 1) - 32 * (pieces - 1);
        last_in = (g.NQpad - tile) + 32 * (pieces - 1) + tail - 1;
    } else {
        last_in = (g.NQpad - tile) + 320 + 32 * 8 + (g.P - 320) + 0;        // segment 1, piece 8, one position
    }
    const long long last_out = g.NQpad - 1 + out_shift;
    const long long blocks = mmlf_conv2x2_blocks(K, N, B, H, W);
    ends[0] = (last_in + 1) * cs_in * 4;                                      // in
    ends[1] = (int64_t)nchunk * 8 * np * 16 + (int64_t)np * 4;                // packed (planes, then the columns' 1 / scale)
    ends[2] = (int64_t)N * 4;                                                 // bias
    ends[3] = (last_out * cs_out + N_store) * 4;                              // out
    ends[4] = cs_ref ? (last_out * cs_ref + N_store) * 4 : 0;                 // relu_ref
    ends[5] = (MMLF_AMAX_HEAD + ((g.NQ - 1) / g.P + 1) + 1) * 4;              // in_amax: rows read by the last valid wave, + 1
    ends[6] = (MMLF_AMAX_HEAD + last_out / g.P + 1) * 4;                      // out_amax
    ends[7] = blocks * 2 * N * 8;                                             // bn_partial (doubles)
    ends[8] = g.NQpad / MMLF_TILE * 4096 * 4;                                 // relu mask words (in or out)
    return 0;
}

