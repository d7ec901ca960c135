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
static inline int mi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

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
