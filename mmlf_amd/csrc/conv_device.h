// conv_device.h -- what the convolution (conv.hip) and weight-gradient (wgrad.hip) translation units share: vector types,
// the build switches, operand splits, buffer-descriptor helpers, the compute-unit count the persistent grids are sized by.
#pragma once
#include <cstdlib>
#include <stdlib.h>
#include "common.h"
#include "../../include/mmlf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// wave-uniform descriptor of what is left of a buffer of `total` bytes behind byte offset `off` (32-bit num_records).
// The remainder is rounded UP to 16 bytes (a buffer whose byte count is not a multiple of 16 -- n_store = 70 floats in a 72-float
// row -- keeps its last elements; every grid buffer has P + 72 positions of slack behind it) and an offset past the end gives
// ZERO records, not an unsigned wrap-around to "unbounded" (round 5's form rounded down and wrapped; harmless only because of
// the slack).  Formed in 16-byte units so that shift, minimum and shift back are 32-bit scalar instructions; the sign clamp is
// mask arithmetic (a 64-bit ordered compare would land on the vector unit).
// In the product build an access past num_records is DROPPED (stores) or reads 0 (loads): a quietly wrong result instead of a
// fault.  The signals that it does not happen: -DMMLF_BOUNDS_DEBUG counts every such access on the GPU (tests/test_gpu_bounds.py),
// and MMLF_CHECK_EXTENTS=1 makes the host compare the audited end of every launch with the bytes behind each pointer it was
// given (mmlf_amd/engine.py, tests/test_gpu_bounds.py::test_host_extent_check_*).
__device__ __forceinline__ int mmlf_records_left(long long total, long long off)
{
    const long long left = total - off;
    const unsigned long long pos = (unsigned long long)(left & ~(left >> 63));         // max(left, 0)
    const unsigned left16 = (unsigned)((pos + 15) >> 4);
    return (int)((left16 < 0x7ffffffu ? left16 : 0x7ffffffu) << 4);
}

// Build switches.  Round 5 removed the timing-ablation switches of rounds 3-4 whose experiments are closed (their numbers
// stay in EXPERIMENTS.md 4.7-4.8: half weight-fragment reads, double split, pre-split operand, 32x32x16 tiles, no early
// barrier, non-temporal activation DMA, wave priorities, the narrow kernel's timeline; check out round 4's tree to rebuild
// them).  What is left changes either nothing observable (MMLF_RING16, MMLF_WGRAD_EARLY, MMLF_WGRADN_CLAMP, MMLF_WGRAD_ZEROPAD:
// tuning constants and equivalent forms) or the RESULT:
//   MMLF_ABL_TERMS < 3 -- run only 2 or 1 of the f16 split's three cross terms (a timing ablation: WRONG results);
//   MMLF_ABL_WGRAD_STAGE -- timing ablations of the wide weight gradient's staging (below: WRONG results);
//   MMLF_ABL_RS_FUSE -- timing proxy of a fused evaluation stream block (below: WRONG results).
// mmlf_build_info() reports every one of them and the Python loader refuses a library with a result-changing switch
// unless MMLF_ALLOW_ABLATION=1 is set (mmlf_amd/_lib.py).
#ifndef MMLF_ABL_TERMS
#define MMLF_ABL_TERMS 3     // cross terms of the f16 split that are evaluated (3 = the arithmetic; fewer: timing ablation)
#endif
#ifndef MMLF_ABL_WGRAD_STAGE
#define MMLF_ABL_WGRAD_STAGE 0   // wide weight gradient, timing ablations of its staging (WRONG results): 1 = the gradient tile
#endif                           // is stored unsplit (loads and LDS stores stay, no vector work on it: what a producer-side split
                                 // could save at most); 2 = it is neither loaded nor stored after the first chunk (what staging
                                 // it ONCE per chunk for all six slices could save at most); 3 = nothing is staged after the first
                                 // chunk (matrix instructions, fragment reads and the barrier alone)
#ifndef MMLF_ABL_RS_FUSE
#define MMLF_ABL_RS_FUSE 0       // round-6 timing proxy of a FUSED evaluation stream block (WRONG results): in the register-streamed
#endif                           // narrow kernel, pad-1 launches (a block's first convolution) store nothing and pad-0 launches (its
                                 // second) load no activations -- what conv(p1)+ReLU+conv(p0) of a block could cost at the very least
                                 // if the intermediate never left the CU (the exchange through LDS is NOT counted): EXPERIMENTS.md 4.10
#ifndef MMLF_RING16
#define MMLF_RING16 3   // pipeline depth of the sixteen-wave conv variant (LDS: 30 KB per buffer + 20 KB; 4 measures the same)
#endif
#define MMLF_BUF_FLAGS 0x00020000   // raw dword buffer (DATA_FORMAT_32), no swizzle
// (round 5 measured the epilogue's stores with the non-temporal policy on the 80-column kernels, whose activation lines compete
// with their own output for an XCD's L2: +12...20 % time on every launch kind, profiles/r05_kbench_nt_store.log -- the L2 is what
// merges the two 64-byte halves of an output line that two store instructions write; the switch was removed again)
#ifndef MMLF_WGRAD_EARLY
#define MMLF_WGRAD_EARLY 1     // the wide weight gradient's early barrier + next-chunk fragment prefetch (16 VGPRs)
#endif
#ifndef MMLF_WGRADN_CLAMP
#define MMLF_WGRADN_CLAMP 1    // narrow weight gradient: staging items and channels past the tile are clamped, not predicated (wgrad.hip)
#endif
#ifndef MMLF_WGRAD_ZEROPAD
#define MMLF_WGRAD_ZEROPAD 0   // wide weight gradient: 1 = padding channels staged as zeros behind selects (round 4's form; wgrad.hip)
#endif
#ifndef MMLF_SRC_HASH
#define MMLF_SRC_HASH "unknown"     // content hash of csrc/*.hip, csrc/*.h and include/mmlf_hip.h (csrc/build.py, tools/build_variant.sh)
#endif
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne_bits(float x)
{
    unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_float(unsigned h) { return __uint_as_float(h << 16); }

// x = hi + mid + lo exactly (three bf16, round-to-nearest-even residual chain)
__device__ __forceinline__ void split3(float x, unsigned &hi, unsigned &mid, unsigned &lo)
{
    hi = bf16_rne_bits(x);
    const float r1 = x - bf16_bits_to_float(hi);
    mid = bf16_rne_bits(r1);
    lo = bf16_rne_bits(r1 - bf16_bits_to_float(mid));
}

// two floats -> two bf16 (RNE) packed in one dword (a low, b high): one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (a, b) = hi + mid + lo exactly, each plane packed like cvt_pk_bf16: 5.5 VALU ops per element
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(ra, rb);
    l = cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xFFFF0000u));
}

// ---- 2-way f16 split ("f16x3"): x*s = hi + lo with 22 mantissa bits, s a power of two that brings the
// tensor's max |x| into [2^14, 2^15) (f16 tops out at 65504; the lo halves of all elements within 2^-17 of
// the maximum stay normal numbers).  Three passes hi*hi + hi*lo + lo*hi on the f16 matrix cores reach the
// error of the exact-f32 MFMA chain (tools/f16x2_accuracy.hip); scaling by a power of two is exact.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ __forceinline__ float pow2_scale_for(float amax)
{
    unsigned u;
    __builtin_memcpy(&u, &amax, 4);
    const int be = (int)((u >> 23) & 0xFF);
    if (be == 0) return 1.f;                       // all-zero (or denormal) tensor
    int e = 268 - be;                              // 2^(14 - floor(log2 amax))
    e = e < 27 ? 27 : (e > 227 ? 227 : e);         // 2^-100 .. 2^100: 1/scale and scale products stay normal
    u = (unsigned)e << 23;
    float r;
    __builtin_memcpy(&r, &u, 4);
    return r;
}
// Two instructions per element: v_fma_mix{lo,hi}_f16 computes fma(x, s, c) in f32 and rounds ONCE to f16, so
// hi = f16(x * s) and lo = f16(x * s - hi) (the product by a power of two is exact, the difference is formed
// exactly inside the fma) -- the same bits as multiply / convert / convert back / subtract / convert (five to six
// instructions per element), which is what these kernels ran before (same time, measured: the loops are limited by the
// board's power, not by vector issue slots -- but fewer instructions are fewer instructions).  SCALAR is a tag only.
template <bool SCALAR = false>
__device__ __forceinline__ void split2_pair_f16(float a, float b, float s, unsigned &h, unsigned &l)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float su = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s)));   // wave-uniform by construction
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(su));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(su));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(su), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(su), "v"(h));
#else
    h = l = 0;
#endif
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Compute units the persistent launches size their grids by.  MMLF_CONV_CUS=<n> caps it (multiples of 8 keep the
// XCD-aware tile order): with n < 256 the conv / weight-gradient grids leave 256 - n CUs without a resident workgroup,
// which is where a collective's kernels can run BESIDE them under data parallelism (the wide kernels take 151 KB of a
// CU's 160 KB LDS: nothing else fits on a CU they occupy).  bench.py reports the value in config.conv_cus.
static int device_cus()
{
    static int cus[64] = {};                 // per device ordinal (a benign race only repeats the query)
    static const int cap = [] { const char *e = getenv("MMLF_CONV_CUS"); return e ? atoi(e) : 0; }();
    const int dev = current_device();
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cap >= 8 && cap < cus[dev] ? cap : cus[dev];
}

#define THIN_MAXN 2
struct ThinArgs {
    const float *in;          // grid tensor, channel stride cs_in (multiple of 4), C real channels
    const float *w;           // OIHW master (N, C, 2, 2)
    const float *bias;
    float *part;              // [positions][4 taps][THIN_MAXN]
    float *out;
    float *out_amax;
    const float *g;           // wgrad: output gradient, channel stride cs_g
    float *wpart;             // wgrad partial sums [split][4][CIP][THIN_MAXN]
    long long NQ, npos;       // valid grid positions; positions to visit (NQ + P + 1 for the forward halo)
    int cs_in, C, N, cs_out, out_shift, vh, vw, P, R, relu, variant, cs_g, g_shift, CIP, dgrad_taps;
    Magic divP, divR;
};

#ifdef MMLF_BOUNDS_DEBUG
// the weight-gradient translation unit's share of the counters (every unit has its own device array, common.h)
int mmlf_oob_counts_wgrad(unsigned long long *host8, int reset);
#endif
