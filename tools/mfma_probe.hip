// Calibration probe: what rate does v_mfma_f32_32x32x2_f32 sustain (a) from registers, (b) with the
// per-slice LDS fragment reads of conv_igemm, (c) plus a barrier per slice, (d) with 1 vs 2 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int VARIANT, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (128 * 36 + 32 * 64)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    for (int i = tid; i < 2 * (128 * 36 + 32 * 64); i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af[2][16], bf[16];
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 16; ++t) af[i][t] = 0.001f * (lane + t + i);
    for (int t = 0; t < 16; ++t) bf[t] = 0.002f * (lane - t);
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT >= 1) {
            const float* Ab = lds + buf * (128 * 36 + 32 * 64);
            const float* Bb = Ab + 128 * 36;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (wave >> 1) * 64 + i * 32 + l32;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 v = *reinterpret_cast<const float4*>(Ab + r * 36 + h * 16 + 4 * u);
                    af[i][4 * u] = v.x; af[i][4 * u + 1] = v.y; af[i][4 * u + 2] = v.z; af[i][4 * u + 3] = v.w;
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) bf[t] = Bb[(h * 16 + t) * 64 + (wave & 1) * 32 + l32];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[(NACC == 2) ? i : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[t], acc[(NACC == 2) ? i : 0], 0, 0, 0);
        if (VARIANT >= 2) { __syncthreads(); buf ^= 1; }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int N>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V, N><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<V, N><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-34s blocks=%4d  %7.3f ms  %6.1f TF/s\n", name, blocks, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0, 2>("regs only, 2 acc", blocks);
        run<0, 1>("regs only, 1 acc", blocks);
        run<1, 2>("+ LDS fragment reads, 2 acc", blocks);
        run<2, 2>("+ LDS reads + barrier, 2 acc", blocks);
        run<2, 1>("+ LDS reads + barrier, 1 acc", blocks);
    }
    return 0;
}
