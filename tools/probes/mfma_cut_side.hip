// Probe: a chain of v_mfma_f32_32x32x16_bf16 whose A operand is produced, half a bf16x3 cut (5 / 6 vector instructions) per MFMA
// gap, by the same wave - the structure of infer_dogm.hip's groups - on ONE wave per SIMD.  Cycles per MFMA (s_memtime):
//   MODE 0: MFMAs only (operands constant)                    MODE 1: + the half-cuts, fenced behind each MFMA (sched_barrier)
//   MODE 2: + the half-cuts, left to the scheduler            MODE 3: as 1, the cut's results NOT consumed by the MFMAs
//   MODE 4: as 1 with 3 (not 6) vector instructions per gap   MODE 5: as 1, B operands rotate over 27 register quads
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probes/mfma_cut_side.hip -o tools/probes/mfma_cut_side
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    __shared__ float big[160 * 1024 / 4 - 64];          // one workgroup per CU: one wave per SIMD
    const int tid = threadIdx.x, lane = tid & 63;
    big[tid] = tid;
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 T[27];
    for (int i = 0; i < 27; ++i) T[i] = u32x4{0x3c003c00u + i, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
    u32x4 cur[3] = {T[0], T[1], T[2]}, nxt[3] = {T[3], T[4], T[5]};
    float raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = big[(tid + i) & 255] * 0.37f + i;
    float ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {                     // four groups of six MFMAs; the next A operand is cut behind 8 of each 12
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const u32x4* B = (MODE == 5) ? &T[(3 * (4 * (it & 1) + g)) % 24] : &T[6];
                acc[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[PA[k]]), __builtin_bit_cast(bf16x8, B[PB[k]]),
                                                                  acc[g & 1], 0, 0, 0);
                if (MODE == 1 || MODE >= 3) __builtin_amdgcn_sched_barrier(0);
                if (MODE != 0) {
                    const int q = (g & 1) * 6 + k;        // half-op 0 .. 11 of a pair of groups: 8 used
                    if (q < 8) {
                        const int d = q >> 1;
                        if (!(q & 1)) {
                            const unsigned a0 = __float_as_uint(raw[2 * d]), b0 = __float_as_uint(raw[2 * d + 1]);
                            ra[d] = raw[2 * d] - __uint_as_float(a0 & 0xffff0000u);
                            rb[d] = raw[2 * d + 1] - __uint_as_float(b0 & 0xffff0000u);
                            nxt[0][d] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
                        } else if (MODE != 4) {
                            const unsigned a1 = __float_as_uint(ra[d]), b1 = __float_as_uint(rb[d]);
                            const unsigned a2 = __float_as_uint(ra[d] - __uint_as_float(a1 & 0xffff0000u));
                            const unsigned b2 = __float_as_uint(rb[d] - __uint_as_float(b1 & 0xffff0000u));
                            nxt[1][d] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
                            nxt[2][d] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
                        }
                    }
                }
                if (MODE == 1 || MODE >= 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (g & 1) {
                if (MODE != 3 && MODE != 0) { cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; }
                for (int i = 0; i < 8; ++i) raw[i] = raw[i] * 1.0001f + 0.25f;     // (8 more vector instructions per 12 MFMAs)
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    s += __uint_as_float(nxt[0][0] ^ nxt[1][1] ^ nxt[2][2]) + raw[3];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 4000;
    probe<MODE><<<256, 256>>>(out, cyc, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE><<<256, 256>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[1024];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 1024; ++i) mean += (double)h[i];
    mean /= 1024;
    printf("%-62s %6.1f shader cycles per MFMA   (%.2f ns per MFMA by the wall clock)\n", name, mean / ((double)iters * 24), ms * 1e6 / ((double)iters * 24));
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    run<0>("MFMAs only", out, cyc);
    run<1>("+ half-cuts (5 / 6 vector instr.), fenced behind each MFMA", out, cyc);
    run<2>("+ half-cuts, left to the scheduler", out, cyc);
    run<3>("+ half-cuts fenced, results not consumed by the MFMAs", out, cyc);
    run<4>("+ first halves only (5 instr. in every other gap), fenced", out, cyc);
    run<5>("+ half-cuts fenced, B operands rotating over 24 register quads", out, cyc);
    return 0;
}
