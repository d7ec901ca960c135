// Issue rate of packed f32 vector arithmetic on gfx950 (v_pk_fma_f32 / v_pk_add_f32) against the single forms: the question
// behind the DoG picker's two filter kernels, which are bound by vector-instruction issue (profiles/r06_experiments.txt item 10).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ab/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float w0, float w1, int iters) {
    f2 a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = f2{(float)threadIdx.x + k, 1.f + k * w1}; b[k] = f2{0.5f * k + w0, 0.25f * w1 + threadIdx.x}; }
    const f2 w = {w0, w0};
    const float wv0 = w0 + threadIdx.x * 1e-9f, wv1 = w1 + threadIdx.x * 1e-9f;     // the weights in vector registers
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MODE == 0) {            // 2 x fma, scalar weight
                    a[k].x = fmaf(w0, b[k].x, a[k].x);
                    a[k].y = fmaf(w1, b[k].y, a[k].y);
                } else if (MODE == 1) {     // packed fma
                    a[k] = __builtin_elementwise_fma(w, b[k], a[k]);
                } else if (MODE == 2) {     // packed add + packed fma (the filter's inner pair)
                    const f2 s = a[k] + b[(k + 1) & 7];
                    b[k] = __builtin_elementwise_fma(w, s, b[k]);
                } else if (MODE == 3) {     // 2 x (add + fma)
                    const float s0 = a[k].x + b[(k + 1) & 7].x, s1 = a[k].y + b[(k + 1) & 7].y;
                    b[k].x = fmaf(w0, s0, b[k].x);
                    b[k].y = fmaf(w1, s1, b[k].y);
                } else if (MODE == 4) {     // 2 x add
                    a[k].x = a[k].x + b[k].x;
                    a[k].y = a[k].y + b[k].y;
                } else if (MODE == 5) {     // 2 x fma, vector weight
                    a[k].x = fmaf(wv0, b[k].x, a[k].x);
                    a[k].y = fmaf(wv1, b[k].y, a[k].y);
                } else if (MODE == 6) {     // 4 x add
                    a[k].x = a[k].x + b[k].x;
                    a[k].y = a[k].y + b[k].y;
                    b[k].x = b[k].x + wv0;
                    b[k].y = b[k].y + wv1;
                } else if (MODE == 7) {     // packed add
                    a[k] = a[k] + b[k];
                } else if (MODE == 8) {     // 2 x fma + 2 x fma (no adds): 4 instructions
                    a[k].x = fmaf(w0, b[k].x, a[k].x);
                    a[k].y = fmaf(w1, b[k].y, a[k].y);
                    b[k].x = fmaf(w0, wv0, b[k].x);
                    b[k].y = fmaf(w1, wv1, b[k].y);
                }
            }
    }
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + b[k];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <int MODE>
float run(float* out, int iters, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2000;
    const char* names[9] = {"2 x v_fma_f32 (sgpr w)", "v_pk_fma_f32", "v_pk_add_f32 + v_pk_fma_f32", "2 x (v_add_f32 + v_fma_f32)",
                            "2 x v_add_f32", "2 x v_fma_f32 (vgpr w)", "4 x v_add_f32", "v_pk_add_f32", "4 x v_fma_f32"};
    const int instr[9] = {2, 1, 2, 4, 2, 2, 4, 1, 4};
    for (int wgs : {256 * 8, 256 * 4, 256 * 2, 256}) {
        float ms[9] = {run<0>(out, iters, wgs), run<1>(out, iters, wgs), run<2>(out, iters, wgs), run<3>(out, iters, wgs), run<4>(out, iters, wgs),
                       run<5>(out, iters, wgs), run<6>(out, iters, wgs), run<7>(out, iters, wgs), run<8>(out, iters, wgs)};
        const double waves_per_simd = wgs * 4.0 / 1024.0;
        printf("%d workgroups of 256 (%.0f waves per SIMD)\n", wgs, waves_per_simd);
        for (int m = 0; m < 9; ++m) {
            const double n_instr = waves_per_simd * iters * 64.0 * instr[m];       // vector instructions per SIMD
            printf("  %-32s %8.3f ms  %5.2f cycles per instruction at 2.4 GHz\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / n_instr);
        }
    }
    return 0;
}
