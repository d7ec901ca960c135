// Probe: vector FP32 rate of scalar v_fma_f32 / v_add_f32 against packed v_pk_fma_f32 / v_pk_add_f32 (two floats per lane and
// instruction) in a filter-shaped loop (symmetric pair add + multiply-add with a scalar weight).  hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
struct W { float w[16]; };

template <int PK>
__global__ __launch_bounds__(256) void k(float* out, W wt, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (PK) {
        f2 win[16], acc[4];
        for (int i = 0; i < 16; ++i) win[i] = f2{(float)(t + i), (float)(t - i)} * 1e-3f;
        for (int u = 0; u < 4; ++u) acc[u] = f2{0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 1; d < 6; ++d)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f2 pair = win[(6 + u - d) & 15] + win[(6 + u + d) & 15];
                    acc[u] = __builtin_elementwise_fma(f2{wt.w[d], wt.w[d]}, pair, acc[u]);
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) win[i] += acc[i & 3] * 1e-9f;
        }
        out[t] = acc[0].x + acc[1].y + acc[2].x + acc[3].y;
    } else {
        float win[16], acc[4], win2[16], acc2[4];
        for (int i = 0; i < 16; ++i) { win[i] = (float)(t + i) * 1e-3f; win2[i] = (float)(t - i) * 1e-3f; }
        for (int u = 0; u < 4; ++u) { acc[u] = 0.f; acc2[u] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 1; d < 6; ++d)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float pair = win[(6 + u - d) & 15] + win[(6 + u + d) & 15];
                    acc[u] = fmaf(wt.w[d], pair, acc[u]);
                    const float pair2 = win2[(6 + u - d) & 15] + win2[(6 + u + d) & 15];
                    acc2[u] = fmaf(wt.w[d], pair2, acc2[u]);
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) { win[i] += acc[i & 3] * 1e-9f; win2[i] += acc2[i & 3] * 1e-9f; }
        }
        out[t] = acc[0] + acc2[1] + acc[2] + acc2[3];
    }
}

int main() {
    float* out; hipMalloc(&out, 4 << 20);
    W w; for (int i = 0; i < 16; ++i) w.w[i] = 0.01f * (i + 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per iteration and thread: 2 columns x (20 adds + 20 fma + 16 fma) = 112 lane-operations
            const double ops = (double)blocks * 256 * iters * 112;
            if (rep) printf("%s: %.3f ms, %.2f T lane-ops/s\n", pk ? "packed" : "scalar", ms, ops / ms * 1e-9);
        }
    }
    return 0;
}
