// Calibration probe: what ONE extra instruction of each kind costs when it sits between v_mfma_f32_32x32x16_bf16's
// (3 accumulators, whole chip): cycles added per instruction, one and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_coissue.hip -o tools/probes/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// (loads: the value read by slot t is consumed - xor-ed into a register - by the slot that reuses its register, PER slots later at
// 4/MFMA and four MFMAs later at 1/MFMA: every load is live, none waits for its own data)
// KIND 0: nothing; 1: v_fma_f32; 2: v_and_b32 (independent); 3: v_perm_b32; 4: ds_read_b128; 5: ds_read_b64_tr_b16;
// 6: ds_write_b128; 7: buffer_load_b128 (L2-resident); 8: s_add (scalar); 9: v_cndmask; 10: ds_read_b64
// 11: v_sub_f32; 12: v_mul_f32; 13: v_add_u32; 14: v_lshlrev_b32; 15: v_alignbit_b32; 16: v_pk_add_f32 (2 floats); 17: v_xor_b32;
// 18: v_cvt_pk_bf16_f32; 19: v_ffbh_u32; 20: v_mov_b32; 21: v_and_or_b32; 22: ds_read_b32
template <int KIND, int PER>
__global__ __launch_bounds__(256, 2) void probe(float* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 ua = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    float v[4] = {lane * 0.5f, 1.f, 2.f, 3.f};
    unsigned w[4] = {(unsigned)lane, 7u, 9u, 11u};
    u32x4 q[4] = {ua, ub, ua, ub};
    int sacc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int laddr = (tid * 16 + it * 1024) & 16383;       // (iteration-dependent: the reads cannot be hoisted)
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            acc[t % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t % 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = (t * PER + k) & 3;
                if (KIND == 1) v[j] = v[j] * 1.0001f + 0.5f;
                if (KIND == 2) w[j] = w[j] & (0xffff0000u | it);
                if (KIND == 3) w[j] = __builtin_amdgcn_perm(w[j], w[(j + 1) & 3], 0x07060302u);
                if (KIND == 4) { w[j] ^= q[j][0] ^ q[j][3]; q[j] = *reinterpret_cast<const u32x4*>(lds + ((laddr + 1024 * (t & 7)) & 16383)); }
                if (KIND == 5) { w[j] ^= q[j][0] ^ q[j][1]; bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + ((laddr + 1024 * (t & 7)) & 16383))); q[j][0] = __builtin_bit_cast(unsigned long long, r) & 0xffffffffu; q[j][1] = __builtin_bit_cast(unsigned long long, r) >> 32; }
                if (KIND == 6) *reinterpret_cast<u32x4*>(lds + 16384 + ((laddr + 1024 * (t & 7)) & 16383)) = q[j];
                if (KIND == 7) w[j] ^= q[j][0] ^ q[j][3];
                if (KIND == 7) q[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid * 16 + 4096 * (t & 7) + it * 65536) & 0xfffff, 0, 0);
                if (KIND == 11) v[j] = v[j] - v[(j + 1) & 3];
                if (KIND == 12) v[j] = v[j] * v[(j + 1) & 3];
                if (KIND == 13) w[j] = w[j] + w[(j + 1) & 3];
                if (KIND == 14) w[j] = w[(j + 1) & 3] << (it & 3);
                if (KIND == 15) w[j] = __builtin_amdgcn_alignbit(w[j], w[(j + 1) & 3], 16);
                if (KIND == 16) { typedef float f2 __attribute__((ext_vector_type(2))); f2 x = {v[j], v[(j + 1) & 3]}, y = {v[(j + 2) & 3], v[(j + 3) & 3]}; x = x - y; v[j] = x[0]; v[(j + 1) & 3] = x[1]; }
                if (KIND == 17) w[j] = w[j] ^ w[(j + 1) & 3];
                if (KIND == 18) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); b2 r = {(__bf16)v[j], (__bf16)v[(j + 1) & 3]}; w[j] ^= __builtin_bit_cast(unsigned, r); }
                if (KIND == 19) w[j] = __builtin_clz(w[(j + 1) & 3] | 1u) + w[j];
                if (KIND == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(w[j]) : "v"(w[(j + 1) & 3]));
                if (KIND == 21) w[j] = (w[j] & 0xffff0000u) | w[(j + 1) & 3];
                if (KIND == 22) { w[j] ^= q[j][0]; q[j][0] = *reinterpret_cast<const unsigned*>(lds + ((tid * 4 + it * 256 + 1024 * (t & 7)) & 16383)); }
                if (KIND == 23) asm volatile("v_fmac_f32 %0, -1.0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 3]));
                if (KIND == 24) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(v[(j + 1) & 3]));
                if (KIND == 25) asm volatile("v_fma_f32 %0, -1.0, %1, %2" : "=v"(v[j]) : "v"(v[(j + 1) & 3]), "v"(v[j]));
                if (KIND == 26) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(v[(j + 1) & 3]));
                if (KIND == 27) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(v[(j + 1) & 3]));
                if (KIND == 28) asm volatile("v_exp_f32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 1) & 3]));
                if (KIND == 8) sacc = __builtin_amdgcn_readfirstlane(sacc) + it;
                if (KIND == 9) w[j] = (lane & (1 << k)) ? w[j] : w[(j + 1) & 3];
                if (KIND == 10) { w[j] ^= q[j][0] ^ q[j][1]; unsigned long long r = *reinterpret_cast<const unsigned long long*>(lds + ((tid * 8 + it * 512 + 1024 * (t & 7)) & 16383)); q[j][0] = (unsigned)r; q[j][1] = (unsigned)(r >> 32); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v[0] + v[1] + v[2] + v[3] + (float)(w[0] + w[1] + w[2] + w[3] + sacc);
    for (int j = 0; j < 4; ++j) s += (float)(q[j][0] + q[j][1] + q[j][2] + q[j][3]);
    for (int a2 = 0; a2 < 3; ++a2) for (int r = 0; r < 16; ++r) s += acc[a2][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out; static float* g_src;
template <int KIND, int PER>
double run(int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<KIND, PER><<<blocks, 256>>>(g_out, g_src, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<KIND, PER><<<blocks, 256>>>(g_out, g_src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)iters * 12) / (blocks / 256);       // seconds per MFMA slot per wave
}
template <int KIND>
void row(const char* name, double base1, double base2) {
    const double a1 = run<KIND, 1>(256), a4 = run<KIND, 4>(256), b1 = run<KIND, 1>(512), b4 = run<KIND, 4>(512);
    // in units of the bare MFMA time (= 32 cycles)
    printf("%-22s 1 wave/SIMD: +%5.1f cyc (1/MFMA) +%5.1f cyc each (4/MFMA) | 2 waves/SIMD: +%5.1f  +%5.1f\n", name,
           (a1 / base1 - 1) * 32, (a4 / base1 - 1) * 32 / 4, (b1 / base2 - 1) * 32, (b4 / base2 - 1) * 32 / 4);
}
int main() {
    (void)hipMalloc(&g_out, 512 * 256 * sizeof(float)); (void)hipMalloc(&g_src, 1 << 20); (void)hipMemset(g_src, 0, 1 << 20);
    const double base1 = run<0, 1>(256), base2 = run<0, 1>(512);
    printf("bare MFMA: %.2f ns per MFMA (1 wave/SIMD), %.2f ns (2 waves/SIMD, per wave)\n", base1 * 1e9, base2 * 1e9);
    row<1>("v_fma_f32", base1, base2);
    row<11>("v_sub_f32", base1, base2);
    row<24>("v_sub_f32 (asm)", base1, base2);
    row<23>("v_fmac_f32 x,-1.0,hi", base1, base2);
    row<25>("v_fma_f32 -1.0,hi,x", base1, base2);
    row<26>("v_add_f32 (asm)", base1, base2);
    row<27>("v_max_f32 (asm)", base1, base2);
    row<28>("v_exp_f32 (asm)", base1, base2);
    row<12>("v_mul_f32", base1, base2);
    row<16>("v_pk_add_f32", base1, base2);
    row<13>("v_add_u32", base1, base2);
    row<14>("v_lshlrev_b32", base1, base2);
    row<15>("v_alignbit_b32", base1, base2);
    row<17>("v_xor_b32", base1, base2);
    row<18>("v_cvt_pk_bf16_f32 (+xor)", base1, base2);
    row<19>("v_ffbh + add", base1, base2);
    row<20>("v_mov_b32", base1, base2);
    row<21>("v_and_or_b32", base1, base2);
    row<2>("v_and_b32", base1, base2);
    row<3>("v_perm_b32", base1, base2);
    row<9>("v_cndmask_b32", base1, base2);
    row<22>("ds_read_b32", base1, base2);
    row<4>("ds_read_b128", base1, base2);
    row<10>("ds_read_b64", base1, base2);
    row<5>("ds_read_b64_tr_b16", base1, base2);
    row<6>("ds_write_b128", base1, base2);
    row<7>("buffer_load_b128", base1, base2);
    row<8>("s_add (scalar)", base1, base2);
    return 0;
}
