// common.h -- shared helpers of libmmlf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>

#define MMLF_TILE 256  // positions per conv tile (== MMLF_TILE_POSITIONS)
#define MMLF_TILE_MAX 512  // ... of the sixteen-wave variant of the narrow layers: grids are padded to this

extern thread_local char g_mmlf_err[512];

static inline int mmlf_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mmlf_err, sizeof(g_mmlf_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define MMLF_CHECK_ARG(cond, ...)                   \
    do {                                            \
        if (!(cond)) return mmlf_fail(__VA_ARGS__); \
    } while (0)

static inline int mmlf_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mmlf_fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// per-device one-time setup (the library is called from one thread per device under nn.DataParallel)
static inline int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    return d;
}
// One-time set-up per device ordinal (hipFuncSetAttribute is per device; nn.DataParallel drives one thread per device and
// nothing stops two of them from sharing a device).  run(f) executes f ONCE per device, every caller -- also one that
// arrives while another thread is inside f -- returns only after f has finished, and f's status (0 = ok) is what every
// later call returns too: a failed set-up fails every launch of that kernel loudly instead of faulting later.
// (Until round 5 this was a flag set BEFORE the work: a second thread could launch a 151 KB-LDS kernel ahead of the
// attribute that allows it.)
struct PerDeviceOnce {
    std::once_flag flag[64];
    int status[64] = {};
    template <class F> int run(F &&f)
    {
        const int d = current_device();
        std::call_once(flag[d], [&] { status[d] = f(); });
        return status[d];
    }
};
// the usual work: allow `bytes` of dynamic LDS for `kernel` on the current device
static inline int mmlf_allow_lds(const void *kernel, size_t bytes, const char *what)
{
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return mmlf_fail("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s", what, bytes, hipGetErrorString(e));
    return 0;
}

// -DMMLF_BOUNDS_DEBUG: the convolution / weight-gradient kernels COUNT every access that leaves what the ABI says their
// buffers hold (slot = which argument; tests/test_gpu_bounds.py builds such a library on the GPU box, runs the launch kinds of
// a training step and of tools/kbench.py through it and expects zeros).  Off in the product build: the macro is empty.
#ifdef MMLF_BOUNDS_DEBUG
static __device__ unsigned long long g_mmlf_oob[8];   // one per translation unit (no relocatable device code)
#define MMLF_OOB(slot, cond) do { if (cond) atomicAdd(&g_mmlf_oob[slot], 1ull); } while (0)
#else
#define MMLF_OOB(slot, cond) do { } while (0)
#endif
enum { OOB_OUT = 0, OOB_IN = 1, OOB_WG_IN = 2, OOB_WG_G = 3, OOB_AMAX = 4, OOB_MASK = 5, OOB_WG_PART = 6, OOB_REF = 7 };

// Padded-grid geometry of one launch (see include/mmlf_hip.h).
// Columns / rows a patch's grid has beyond its image: 2 and 2, a zero column and row of its own on every side.  The flat
// index wraps, so the zero column in front of a row could double as the one behind the previous row (pitch W + 1) and the
// zero row in front of a patch as the one behind the previous patch (H + 1 rows): every kernel takes P and R from
// make_grid, and both COMPACT forms were built and measured in round 4 (-DMMLF_GRID_PAD_W=1 [-DMMLF_GRID_PAD_H=1]):
//  * H + 1 rows: the kernel tests pass, but the LAST image row of a patch then shares its 32-position wave -- and its operand
//    scale -- with the first row of the next patch: "a small patch beside a large one keeps float32-level precision" is lost
//    for up to 31 positions per patch (test_conv_patches_of_very_different_magnitude fails);
//  * pitch W + 1 alone (9 506 instead of 9 604 positions per 96 x 96 patch): same-box A/B 458.9-459.7 ms against 458.9-461.0 per
//    bs=512 step, wide conv 7.54-7.57 against 7.54-7.59 ms per launch, wide weight gradient -0.8 %
//    (profiles/r04_ab_grid_pitch.log): the 1 % of positions does not show, and two statistical precision bars move
//    (one tensor of the gradient yardstick 3.1 -> 4.5 with median and 90th percentile improved).  Not adopted.
#ifndef MMLF_GRID_PAD_W
#define MMLF_GRID_PAD_W 2
#endif
#ifndef MMLF_GRID_PAD_H
#define MMLF_GRID_PAD_H 2
#endif
struct Grid {
    int B, H, W, P, R, G;
    long long NQ, NQpad;
};

static inline Grid make_grid(int B, int H, int W)
{
    Grid g;
    g.B = B; g.H = H; g.W = W;
    g.P = W + MMLF_GRID_PAD_W; g.R = H + MMLF_GRID_PAD_H; g.G = g.P * g.R;
    g.NQ = (long long)B * g.G;
    g.NQpad = (g.NQ + MMLF_TILE_MAX - 1) / MMLF_TILE_MAX * MMLF_TILE_MAX;
    return g;
}

// slack: taps reach P+1 past a tile; the split kernel's last DMA piece reads 64 positions more
static inline long long grid_alloc_positions(const Grid &g) { return g.NQpad + g.P + 8 + 64; }

// ---- max |x| bookkeeping of the f16-split arithmetic ("amax array" of a grid tensor) ----
// [k * MMLF_AMAX_SHARD_STRIDE], k < MMLF_AMAX_SHARDS: partial maxima of |x| over the whole tensor -- the tensor's maximum
// is the largest of them (mmlf_amax_tensor_max); [MMLF_AMAX_HEAD + r] = max |x| over grid row r = q / P (all channels).
// Entries are raised by fire-and-forget atomic max (non-negative floats order like their bit patterns);
// mmlf_zero_slack zeroes them before the first producer of the tensor runs.  Rows past B*R (tile padding, tap slack)
// stay zero.  Why shards: a kernel with one workgroup per grid row raises the tensor's maximum once per wave, 25 000 -
// 200 000 times per launch; on ONE address that is 1-4 ns each at the memory side (agent-scope atomics and loads do not
// stop at an XCD's L2): 1.1 % of a bs=512 step, 3 % of a 64-patch step (profiles/r04_ab_amax_variants.log).  64 slots,
// 256 bytes apart, and no read-before-raise (a wave never waits for the counter) removed that.
#define MMLF_AMAX_SHARDS 64
#define MMLF_AMAX_SHARD_STRIDE 64
#define MMLF_AMAX_HEAD (MMLF_AMAX_SHARDS * MMLF_AMAX_SHARD_STRIDE)
static inline long long amax_rows(const Grid &g) { return (g.NQpad + 2 * g.P + 64) / g.P + 2; }
static inline long long amax_entries(const Grid &g) { return MMLF_AMAX_HEAD + amax_rows(g); }

// n / d == (n * m) >> sh for 0 <= n < 2^31 (Granlund-Montgomery round-up magic, N = 31)
struct Magic { unsigned m; int sh; };
static inline Magic make_magic(unsigned d)
{
    int l = 0;
    while ((1ull << l) < d) ++l;
    Magic r;
    r.sh = 31 + l;
    r.m = (unsigned)(((1ull << r.sh) + d - 1) / d);
    return r;
}
__host__ __device__ static inline unsigned fastdiv(unsigned n, Magic g)
{
    return (unsigned)(((unsigned long long)n * g.m) >> g.sh);
}

// supported MFMA N-tile counts (32 output channels each)
static inline int pick_nt(int N)
{
    int n = (N + 31) / 32;
    if (n <= 1) return 1;
    if (n <= 3) return 3;
    if (n <= 4) return 4;
    if (n <= 9) return 9;
    return -1;
}

// master-filter tap for (packed tap t, variant): returns sy*2+sx into the OIHW (.,.,2,2) master
__host__ __device__ static inline int master_tap(int t, int variant)
{
    const int dy = t >> 1, dx = t & 1;
    if (variant == 0) return dy * 2 + dx;         // identity
    if (variant == 1) return dx * 2 + dy;         // transpose
    return dx * 2 + (1 - dy);                     // transpose, then flip along kernel-H
}

// Running maxima in device memory: fire-and-forget atomic max -- no read of the slot, nothing to wait for.
__device__ __forceinline__ void mmlf_amax_raise_nowait(float *slot_f, float m)
{
    (void)__hip_atomic_fetch_max(reinterpret_cast<unsigned *>(slot_f), __float_as_uint(m), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float mmlf_wave_max(float m)
{
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
// every lane of the wave calls it with its own maximum; `shard` (any integer, wave-uniform) picks the slot
__device__ __forceinline__ void mmlf_amax_update(float m, float *amax, unsigned shard)
{
    m = mmlf_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f)
        mmlf_amax_raise_nowait(amax + (shard % MMLF_AMAX_SHARDS) * MMLF_AMAX_SHARD_STRIDE, m);
}
// the same for a kernel whose workgroup writes (part of) ONE grid row: raises a tensor shard and the row's entry
__device__ __forceinline__ void mmlf_amax_update_row(float m, float *amax, int row)
{
    m = mmlf_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        mmlf_amax_raise_nowait(amax + ((unsigned)row % MMLF_AMAX_SHARDS) * MMLF_AMAX_SHARD_STRIDE, m);
        mmlf_amax_raise_nowait(amax + MMLF_AMAX_HEAD + row, m);
    }
}
// max |x| of the tensor from its shards; all 64 lanes of the wave must call it (the result is wave-uniform)
__device__ __forceinline__ float mmlf_amax_tensor_max(const float *amax)
{
    static_assert(MMLF_AMAX_SHARDS == 64, "one shard per lane");
    return mmlf_wave_max(amax[(threadIdx.x & 63) * MMLF_AMAX_SHARD_STRIDE]);
}
