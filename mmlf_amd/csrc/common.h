// common.h -- shared helpers of libmmlf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MMLF_TILE 256  // positions per conv tile (== MMLF_TILE_POSITIONS)

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

// Padded-grid geometry of one launch (see include/mmlf_hip.h).
struct Grid {
    int B, H, W, P, R, G;
    long long NQ, NQpad;
};

static inline Grid make_grid(int B, int H, int W)
{
    Grid g;
    g.B = B; g.H = H; g.W = W;
    g.P = W + 2; g.R = H + 2; g.G = g.P * g.R;
    g.NQ = (long long)B * g.G;
    g.NQpad = (g.NQ + MMLF_TILE - 1) / MMLF_TILE * MMLF_TILE;
    return g;
}

// slack: taps reach P+1 past a tile; the split kernel's last DMA piece reads 64 positions more
static inline long long grid_alloc_positions(const Grid &g) { return g.NQpad + g.P + 8 + 64; }

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

// Running max |x| of a tensor in a device scalar (non-negative floats order like their bit patterns).
// Every thread of the block calls it with its own maximum; the atomic is skipped when the scalar already
// holds a larger value (a stale read only costs an extra atomic), so a launch issues few of them.
__device__ __forceinline__ void mmlf_amax_update(float m, float *amax)
{
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned bits = __float_as_uint(m);
        unsigned *slot = reinterpret_cast<unsigned *>(amax);
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}
