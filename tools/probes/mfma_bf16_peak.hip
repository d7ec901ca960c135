// Calibration probe: the rate v_mfma_f32_32x32x16_bf16 sustains on the whole chip from registers - dependent chains of 1, 2,
// 3 and 6 accumulators, one or two waves per SIMD, and with a co-issued VALU load (4 per MFMA) - against the nominal
// 2.5 PFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 ua = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    float v0 = lane * 0.5f, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t % NACC], 0, 0, 0);
            if (VALU) {
                v0 = v0 * 1.0001f + v1; v1 = v1 * 0.9999f + v2; v2 = v2 * 1.0001f + v3; v3 = v3 * 0.9999f + v0;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int a2 = 0; a2 < NACC; ++a2) for (int r = 0; r < 16; ++r) s += acc[a2][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(const char* name, int blocks, int threads) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, VALU><<<blocks, threads>>>(out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC, VALU><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 12 * 32768.0;
    printf("%-44s blocks %4d x %3d thr  %7.3f ms  %8.1f TFLOP/s\n", name, blocks, threads, ms, flop / ms * 1e-9);
    hipFree(out);
}

int main() {
    run<1, 0>("1 acc chain, 1 wave/SIMD", 256, 256);
    run<2, 0>("2 acc, 1 wave/SIMD", 256, 256);
    run<3, 0>("3 acc, 1 wave/SIMD", 256, 256);
    run<6, 0>("6 acc, 1 wave/SIMD", 256, 256);
    run<1, 0>("1 acc chain, 2 waves/SIMD", 512, 256);
    run<2, 0>("2 acc, 2 waves/SIMD", 512, 256);
    run<3, 0>("3 acc, 2 waves/SIMD", 512, 256);
    run<3, 1>("3 acc + 4 VALU per MFMA, 1 wave/SIMD", 256, 256);
    run<3, 1>("3 acc + 4 VALU per MFMA, 2 waves/SIMD", 512, 256);
    run<3, 0>("3 acc, 1 wave/SIMD, 64 CUs", 64, 256);
    run<3, 0>("3 acc, 1 wave/SIMD, 8 CUs", 8, 256);
    return 0;
}
