// Shared helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/cetpick_hip.h"

#define MI_RETURN_IF_LAUNCH_FAILED()                  \
    do {                                              \
        hipError_t e_ = hipGetLastError();            \
        if (e_ != hipSuccess) return (int)e_;         \
    } while (0)
#define MI_HIP(call)                                  \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) return (int)e_;         \
    } while (0)

static inline size_t mi_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// BatchNorm1d (training mode) + optional ReLU in the epilogue of a small dense product (conv_cube2.hip small_gemm_kernel)
struct MiSmallGemmBN {
    float* y;                 // act(bn(C))
    const float* gamma;       // may be null
    const float* beta;
    float eps, momentum;
    float* running_mean;      // may be null (with running_var)
    float* running_var;
    long long* num_batches_tracked;     // may be null
    float* save;              // mean[N], invstd[N]
    int relu;
    double* sums_only;        // not null: ONLY the column sums of C (sum[N], sum of squares[N], doubles) are produced - the
                              // statistics pass of a SyncBN whose all-reduce and apply follow as launches of their own
};
static inline int mi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// conv_cube2.hip, final form: the epilogue operands and the per-tile arrival counters of a launch that needs no reduce
struct Cube2Final {
    unsigned* tickets;        // MI_CUBE2_TICKET_BYTES, zero between launches
    float* out;
    const float* res;
    const float* mask;
    int relu;
};
constexpr size_t MI_CUBE2_TICKET_BYTES = 16384;      // 4,096 output tiles (64 samples x 32 columns each)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 64-bit sort key: (score bits << 32) | tie field.  Scores handled here are > 0, so the raw IEEE
// bits order like the floats.
__device__ __forceinline__ unsigned long long make_key(float score, unsigned tie) {
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)tie;
}

// In-LDS bitonic sort, DESCENDING, of n_pow2 keys by `nthreads` threads (whole block calls it).
__device__ __forceinline__ void block_bitonic_sort_desc(unsigned long long* keys, int n_pow2,
                                                        int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n_pow2 >> 1); t += nthreads) {
                int i = ((t / j) * (j << 1)) + (t % j);   // lower index of the pair
                int l = i + j;
                bool desc = ((i & k) == 0);
                unsigned long long a = keys[i], b = keys[l];
                bool swap = desc ? (a < b) : (a > b);
                if (swap) { keys[i] = b; keys[l] = a; }
            }
        }
    }
    __syncthreads();
}

// Same result as block_bitonic_sort_desc with fewer workgroup barriers: wave w owns the contiguous block of
// B = n_pow2 / n_waves keys; every stage whose pairs stay inside a block (2j <= B) runs without a workgroup barrier
// (LDS operations of one wave execute in order), only the strides that cross blocks are bracketed by barriers:
// 10 instead of 78 barriers for 4096 keys and 16 waves.  The pairs of a step are taken four at a time with all eight
// LDS reads issued before the first compare (one LDS round trip per four pairs: a run-time trip count left to the
// compiler serialises them and made the 8192-key tile sort 3.5 x slower).  nthreads a multiple of 64.
template <bool GUARD>
__device__ __forceinline__ void bitonic_pairs4(unsigned long long* keys, int t0, int stride, int n_pairs, int base, int j,
                                               int lj, int k) {
    unsigned long long a[4], b[4];
    int ii[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * stride;
        ii[u] = base + ((t >> lj) << (lj + 1)) + (t & (j - 1));
        if (!GUARD || t < n_pairs) { a[u] = keys[ii[u]]; b[u] = keys[ii[u] + j]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * stride;
        if (!GUARD || t < n_pairs) {
            // branch-free compare-exchange: both keys are always written back (a data-dependent branch per pair
            // diverges the wave at every step)
            const bool desc = ((ii[u] & k) == 0);
            const bool sw = desc ? (a[u] < b[u]) : (a[u] > b[u]);
            keys[ii[u]] = sw ? b[u] : a[u];
            keys[ii[u] + j] = sw ? a[u] : b[u];
        }
    }
}

__device__ __forceinline__ void block_sort_desc_fast(unsigned long long* keys, int n_pow2, int tid, int nthreads) {
    const int n_waves = nthreads >> 6;
    const int B = n_pow2 / n_waves;                       // keys per wave block (power of two)
    const int lane = tid & 63, wv = tid >> 6;
    const int half = n_pow2 >> 1;
    __syncthreads();
    for (int k = 2; k <= n_pow2; k <<= 1) {
        int j = k >> 1;
        for (; 2 * j > B; j >>= 1) {                      // pairs cross wave blocks
            const int lj = 31 - __builtin_clz(j);
            if ((half & (4 * nthreads - 1)) == 0)
                for (int t0 = tid; t0 < half; t0 += 4 * nthreads) bitonic_pairs4<false>(keys, t0, nthreads, half, 0, j, lj, k);
            else
                for (int t0 = tid; t0 < half; t0 += 4 * nthreads) bitonic_pairs4<true>(keys, t0, nthreads, half, 0, j, lj, k);
            __syncthreads();
        }
        for (; j > 0; j >>= 1) {                          // pairs inside this wave's block
            const int lj = 31 - __builtin_clz(j);
            if (((B >> 1) & 255) == 0)
                for (int t0 = lane; t0 < (B >> 1); t0 += 256) bitonic_pairs4<false>(keys, t0, 64, B >> 1, wv * B, j, lj, k);
            else
                for (int t0 = lane; t0 < (B >> 1); t0 += 256) bitonic_pairs4<true>(keys, t0, 64, B >> 1, wv * B, j, lj, k);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        if (2 * k > B) __syncthreads();                   // the next stage starts with a cross-block stride
    }
    __syncthreads();
}
