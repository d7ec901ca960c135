// DoG particle picker, two launches for the whole filter stage (round 4):
//   dogf_zx_kernel : z pass of BOTH Gaussians (one read of the tomogram, register-ring march) + x pass of both (LDS row)
//   dogf_y_kernel  : y pass of both (register-ring march) + DoG + border + `_nms_xy` (3x3) + fp64 statistics of the
//                    positive survivors + candidate compaction
// Replaces (reference, cet_pick/...): the two scipy.ndimage.gaussian_filter calls, `rec_i = gaussian(s2) - gaussian(s1)`,
// the border zeroing, `_nms_xy(.., kernel=3)` and the `mean + 0.5 std` statistics of utils/image.py:152-179.
//
// Round 3's chain was z pass | y pass | x pass + DoG + NMS: every pass round-trips both Gaussians through HBM (2.14 GB
// for a 256x512x512 tomogram, 4x the algorithmic 0.54 GB).  A separable filter commutes, so the passes are regrouped
// around what each march direction can keep on the chip:
//   * a z march owns a COLUMN per thread (41-row register ring); the 512 threads of a workgroup own one y row of the
//     volume, so each z-filtered plane row passes through LDS once and the x pass (4 consecutive outputs per lane from
//     16-byte LDS reads at lane base + immediate, scipy 'reflect' halo filled by the edge threads) runs on it before
//     anything is written: 1 read + 2 writes of the volume;
//   * the y march owns a column per LANE again (two register rings), the DoG of a row exists only in registers, its
//     x neighbours come from the adjacent lanes by DPP (a wave computes 64 columns and owns the inner 62), its y
//     neighbours from a three-row register history: 2 reads, candidates out, nothing else written.
// 3 volume reads + 2 writes (of the live box only) instead of 5 + 4.  Pass order z, x, y instead of scipy's z, y, x:
// the same products summed in another order (fp32; differences of rounding-order size, tests/test_infer_gpu.py).
// Taps are symmetric scalar operands w[|d|]; (R1 + R2 + 2) multiply-adds per voxel and pass.
// hipcc-flags: -fno-slp-vectorize
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int FZ_T = 512;            // threads of a z/x workgroup = the widest row
constexpr int FY_WAVES = 4, FY_T = 64 * FY_WAVES;
constexpr int FY_OWN = 62;           // columns a wave of the y march owns (64 computed)
constexpr int FRING = 512;
#ifndef DOGF_Y_PD
#define DOGF_Y_PD 2
#endif
#ifndef DOGF_PHASE_GUARD
#define DOGF_PHASE_GUARD
#endif

template <int R>
struct FTaps { float w[R + 1]; };    // w[t] = tap at distance t from the centre

__device__ __forceinline__ int f_reflect(int i, int n) {
    // scipy 'reflect': d c b a | a b c d | d c b a - ONE reflection (the host admits only extents where that is enough:
    // the general form's modulo became ~30 scalar instructions in front of every load of the z march)
    return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i);
}
__device__ __forceinline__ float f_from_lower(float v, float edge) {      // lane l gets lane l-1's value; lane 0 `edge`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float f_from_upper(float v, float edge) {      // lane l gets lane l+1's value; lane 63 `edge`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
// voffset: per lane (range-checked against the descriptor); soffset: wave-uniform, NOT range-checked - callers keep it inside
__device__ __forceinline__ float f_ld(const __amdgpu_buffer_rsrc_t& rs, int voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)soff, 0));
}
__device__ __forceinline__ void f_st4(const __amdgpu_buffer_rsrc_t& rs, int voff, unsigned soff, const float (&g)[4]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {__float_as_uint(g[0]), __float_as_uint(g[1]), __float_as_uint(g[2]), __float_as_uint(g[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, (int)soff, 0);
}
__device__ __forceinline__ float f_mx3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ---------------------------------------------------------------------------------------------------------------------
// z + x
// ---------------------------------------------------------------------------------------------------------------------
// Workgroup = one row y of the volume (512 threads: thread t owns column x = t), marching along z.  Per phase of four
// planes: 4 loads into the ring (issued a phase ahead), 4 x (RB adds + RA + RB + 2 multiply-adds) for the z pass, 8
// LDS stores (+ the reflected halo at the row ends), ONE barrier (the LDS rows are double-buffered), then thread t takes
// plane t / 128 and the four outputs x = 4 (t % 128) ..+3 of both Gaussians and stores them as one 16-byte store each.
template <int RA, int RB>
__global__ __launch_bounds__(FZ_T, 4) void dogf_zx_kernel(DogfParams p, FTaps<RA> wa, FTaps<RB> wb) {
    constexpr int PD = 2;
    constexpr int LEAD = 2 * RB + 4 + 4 * (PD - 1);        // rows resident ahead of output i: q in [i, i + LEAD)
    constexpr int RW = LEAD + 4;                           // ring registers
    constexpr int P = FZ_T + 2 * RB;                       // LDS row: element e <-> x = e - RB
    __shared__ __attribute__((aligned(16))) float xs[2][4][2][P];
    const int tid = threadIdx.x;
    {   // the picker's header and candidate bitmap are zeroed here (a few words per thread): no clearing pass
        const long gt = (long)blockIdx.x * FZ_T + tid, nt = (long)gridDim.x * FZ_T;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (p.clr[k])
                for (long i = gt; i < (long)p.clr_n[k]; i += nt) p.clr[k][i] = 0u;
    }
    const int y = p.ylo + blockIdx.x;
    const int W = p.W, n = p.D, z0 = p.bz, z1 = p.D - p.bz;
    const bool col_ok = tid < W;
    const unsigned zs4 = 4u * (unsigned)p.H * (unsigned)W;           // bytes per plane (the volume is < 2 GiB: host check)
    // raw buffer accesses: the plane offset is a scalar (soffset), the thread's place in the plane a 32-bit VGPR (voffset)
    const __amdgpu_buffer_rsrc_t rrs = f_rsrc(p.rec, p.vol_bytes), g1rs = f_rsrc(p.g1, p.vol_bytes), g2rs = f_rsrc(p.g2, p.vol_bytes);
    const int src_off = 4 * (y * W + (col_ok ? tid : W - 1));
    auto zload = [&](int z) { return f_ld(rrs, src_off, (unsigned)f_reflect(z, n) * zs4); };
    float win[RW];
#pragma unroll
    for (int q = 0; q < LEAD; ++q) win[q] = zload(z0 + q - RB);
    // x pass mapping (the plane of a wave is wave-uniform: scalar)
    const int xu = __builtin_amdgcn_readfirstlane(tid >> 7), xb = (tid & 127) * 4;
    const bool x_ok = xb < W;
    const int dst_off = 4 * (y * W + xb);
    int par = 0;
    for (int i0 = z0; i0 < z1; i0 += RW) {
#pragma unroll
        for (int ph = 0; ph < RW / 4; ++ph) {
            const int i = i0 + 4 * ph;                     // slot of row q is (q - i0) mod RW: static per phase
            if (i < z1) {                                  // (workgroup-uniform; no `break`: the turn must stay unrolled)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = i + LEAD + u - RB;           // never below 0 (LEAD >= RB): only the far end can reflect
                win[(4 * ph + LEAD + u) % RW] = zload(j);
            }
            float accb[4], acca[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ctr = win[(4 * ph + RB + u) % RW];
                accb[u] = wb.w[0] * ctr;
                acca[u] = wa.w[0] * ctr;
            }
#pragma unroll
            for (int t = 1; t <= RB; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float pair = win[(4 * ph + RB + u - t + RW) % RW] + win[(4 * ph + RB + u + t) % RW];
                    accb[u] = fmaf(wb.w[t], pair, accb[u]);
                    if (t <= RA) acca[u] = fmaf(wa.w[t], pair, acca[u]);
                }
            float* buf = &xs[par][0][0][0];
            if (col_ok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    buf[(2 * u + 0) * P + RB + tid] = acca[u];
                    buf[(2 * u + 1) * P + RB + tid] = accb[u];
                }
                if (tid < RB) {                            // x' = -1 - x
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        buf[(2 * u + 0) * P + RB - 1 - tid] = acca[u];
                        buf[(2 * u + 1) * P + RB - 1 - tid] = accb[u];
                    }
                }
                if (tid >= W - RB) {                       // x' = 2 W - 1 - x
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        buf[(2 * u + 0) * P + RB + 2 * W - 1 - tid] = acca[u];
                        buf[(2 * u + 1) * P + RB + 2 * W - 1 - tid] = accb[u];
                    }
                }
            }
            __syncthreads();
            if (x_ok && i + xu < z1) {                     // (xu is wave-uniform)
                const unsigned so = (unsigned)(i + xu) * zs4;
                {
                    const float4* rp = reinterpret_cast<const float4*>(buf + (2 * xu + 1) * P + xb);
                    float val[2 * RB + 4];
                    // centre first, then outwards: the tap pairs of distance d need elements RB - d .. RB + 3 + d, so the
                    // multiply-adds start when the first reads are back instead of behind the whole window
#pragma unroll
                    for (int k = 0; k < (2 * RB + 4) / 4; ++k) {
                        constexpr int NQ = (2 * RB + 4) / 4, MID = NQ / 2;
                        const int q = (k & 1) ? MID - (k + 1) / 2 : MID + k / 2;
                        const float4 t4 = rp[q < 0 ? 0 : (q >= NQ ? NQ - 1 : q)];
                        if (q >= 0 && q < NQ) { val[4 * q] = t4.x; val[4 * q + 1] = t4.y; val[4 * q + 2] = t4.z; val[4 * q + 3] = t4.w; }
                    }
                    // (distance outermost, the four outputs innermost: four independent multiply-add chains in flight - with the
                    // output outermost the compiler emitted ONE dependent add -> fma chain of 4 x 41 links per window)
                    float g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = wb.w[0] * val[e + RB];
#pragma unroll
                    for (int d = 1; d <= RB; ++d)
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = fmaf(wb.w[d], val[e + RB - d] + val[e + RB + d], g[e]);
                    f_st4(g2rs, dst_off, so, g);
                }
                __builtin_amdgcn_sched_barrier(0);         // one window at a time (register pressure)
                {
                    const float4* rp = reinterpret_cast<const float4*>(buf + (2 * xu + 0) * P + xb + (RB - RA));
                    float val[2 * RA + 4];
#pragma unroll
                    for (int k = 0; k < (2 * RA + 4) / 4; ++k) {
                        constexpr int NQ = (2 * RA + 4) / 4, MID = NQ / 2;
                        const int q = (k & 1) ? MID - (k + 1) / 2 : MID + k / 2;
                        const float4 t4 = rp[q < 0 ? 0 : (q >= NQ ? NQ - 1 : q)];
                        if (q >= 0 && q < NQ) { val[4 * q] = t4.x; val[4 * q + 1] = t4.y; val[4 * q + 2] = t4.z; val[4 * q + 3] = t4.w; }
                    }
                    float g[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = wa.w[0] * val[e + RA];
#pragma unroll
                    for (int d = 1; d <= RA; ++d)
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = fmaf(wa.w[d], val[e + RA - d] + val[e + RA + d], g[e]);
                    f_st4(g1rs, dst_off, so, g);
                }
            }
            par ^= 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// y + DoG + NMS
// ---------------------------------------------------------------------------------------------------------------------
// Wave = (plane z, chunk of rows, strip of 64 columns of which it owns the inner 62).  Two register rings (g1: RA, g2: RB),
// four rows per phase; a row's DoG lives in one register per lane.  Everything that is the same for the whole wave - its
// plane, chunk, segment, row offsets - is kept scalar (readfirstlane of the wave index), the loads are raw buffer loads
// with the row in the scalar offset and the lane's column in the vector offset: no per-load address arithmetic on the
// vector unit.  Statistics are taken when 64 survivors leave the LDS ring (one per lane), not per row.
template <int RA, int RB>
__global__ __launch_bounds__(FY_T, 4) void dogf_y_kernel(DogfParams p, FTaps<RA> wa, FTaps<RB> wb) {
    constexpr int PD = DOGF_Y_PD;                          // phases (4 rows x ~90 vector instructions) a load has to land
    constexpr int LEADB = 2 * RB + 4 + 4 * (PD - 1), RW = LEADB + 4;
    constexpr int LEADA = 2 * RA + 4 + 4 * (PD - 1);
    __shared__ uint2 ring_all[FY_WAVES][FRING];
    __shared__ double s_acc[FY_WAVES][3][64];              // per lane: count, sum, sum of squares of its flushed survivors
    __shared__ double s_st[FY_WAVES][3];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long g = (long)blockIdx.x * FY_WAVES + wv;
    const float NEG = -INFINITY;
    unsigned cnt = 0, flushed = 0;
    uint2* ring = ring_all[wv];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_acc[wv][k][lane] = 0.0;
    if (g < (long)p.n_seg) {
        const int strip = (int)(g % p.n_strips);
        const long t = g / p.n_strips;
        const int yc = (int)(t % p.n_ychunks), z = p.bz + (int)(t / p.n_ychunks);
        const int W = p.W, H = p.H;
        const int x = p.bx - 1 + FY_OWN * strip + lane;
        const bool live_col = x >= p.bx && x < W - p.bx;
        const bool owned = live_col && lane >= 1 && lane <= FY_OWN;
        const int ya = p.by + yc * p.ychunk, yb = min(ya + p.ychunk, H - p.by);
        const int ylive0 = p.by, ylive1 = H - p.by;
        const unsigned plane = (unsigned)z * (unsigned)H * (unsigned)W;
        const __amdgpu_buffer_rsrc_t r1 = f_rsrc(p.g1 + plane, p.vol_bytes - 4u * plane), r2 = f_rsrc(p.g2 + plane, p.vol_bytes - 4u * plane);
        const int xoff = 4 * min(x, W - 1);
        uint2* seg_base = p.cands + (size_t)g * p.seg_cap;
        const int out_lo = ya - 1, n_out = yb + 1;         // rows whose DoG is computed: [ya - 1, yb]; emitted: [ya, yb)
        // rows are clamped into the plane on the scalar unit (a clamped row only ever meets a zero tap or a discarded output)
        auto rowo = [&](int j) { return 4u * (unsigned)(min(max(j, 0), H - 1) * W); };
        float w1[RW], w2[RW];
#pragma unroll
        for (int q = 0; q < LEADB; ++q) w2[q] = f_ld(r2, xoff, rowo(out_lo + q - RB));
#pragma unroll
        for (int q = 0; q < LEADA; ++q) w1[q] = f_ld(r1, xoff, rowo(out_lo + q - RA));
        float xm_pp = NEG, xm_p = NEG, c_p = 0.f;
        auto flush64 = [&](unsigned n_valid) {            // entries [flushed, flushed + 64) of the ring leave; lane < n_valid hold one
            const uint2 e = ring[(flushed + lane) & (FRING - 1)];
            if ((unsigned)lane < n_valid) {
                if (flushed + lane < p.seg_cap) seg_base[flushed + lane] = e;
                const double dv = (double)__uint_as_float(e.x);
                s_acc[wv][0][lane] += 1.0; s_acc[wv][1][lane] += dv; s_acc[wv][2][lane] += dv * dv;
            }
        };
        for (int i0 = out_lo; i0 < n_out; i0 += RW) {
#pragma unroll
            for (int ph = 0; ph < RW / 4; ++ph) {
                const int i = i0 + 4 * ph;
                DOGF_PHASE_GUARD {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w2[(4 * ph + LEADB + u) % RW] = f_ld(r2, xoff, rowo(i + LEADB + u - RB));
                    w1[(4 * ph + LEADA + u) % RW] = f_ld(r1, xoff, rowo(i + LEADA + u - RA));
                }
                float a2[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a2[u] = wb.w[0] * w2[(4 * ph + RB + u) % RW];
                    a1[u] = wa.w[0] * w1[(4 * ph + RA + u) % RW];
                }
#pragma unroll
                for (int d = 1; d <= RB; ++d)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        a2[u] = fmaf(wb.w[d], w2[(4 * ph + RB + u - d + RW) % RW] + w2[(4 * ph + RB + u + d) % RW], a2[u]);
#pragma unroll
                for (int d = 1; d <= RA; ++d)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        a1[u] = fmaf(wa.w[d], w1[(4 * ph + RA + u - d + RW) % RW] + w1[(4 * ph + RA + u + d) % RW], a1[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = i + u;
                    if (r < n_out) {                       // (wave-uniform)
                        const bool row_live = r >= ylive0 && r < ylive1;
                        const float dog = (row_live && live_col) ? a2[u] - a1[u] : 0.f;     // zeroed border (inside the image)
                        const float xm = f_mx3(f_from_lower(dog, NEG), dog, f_from_upper(dog, NEG));
                        const int ro = r - 1;              // row that is complete now
                        if (ro >= ya && ro < yb) {
                            const float hm = f_mx3(xm_pp, xm_p, xm);
                            const float out = (hm == c_p) ? c_p : 0.f;
                            const unsigned oidx = plane + (unsigned)(ro * W + x);
                            if (p.nms_out && owned) p.nms_out[oidx] = out;
                            const bool is = owned && out > 0.f;
                            const unsigned long long mask = __ballot(is);
                            if (is) {
                                const unsigned pos = cnt + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                                ring[pos & (FRING - 1)] = make_uint2(__float_as_uint(out), oidx);
                            }
                            cnt += (unsigned)__popcll(mask);
                        }
                        xm_pp = xm_p; xm_p = xm; c_p = dog;
                    }
                }
                // (at most 63 + 4 x 62 entries wait here: four conditional flushes - a `while` would keep the turn from unrolling)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (cnt - flushed >= 64u) { flush64(64u); flushed += 64; }
                }
            }
        }
        if (cnt > flushed) flush64(cnt - flushed);
        if (lane == 0) {
            p.seg_count[g] = min(cnt, p.seg_cap);
            if (cnt > p.seg_cap) atomicOr(p.overflow, 1u);
        }
    }
    // statistics: one (count, sum, sum of squares) per workgroup, waves added in order (deterministic)
    const double a = wave_sum(s_acc[wv][0][lane]), s = wave_sum(s_acc[wv][1][lane]), ss = wave_sum(s_acc[wv][2][lane]);
    if (lane == 0) { s_st[wv][0] = a; s_st[wv][1] = s; s_st[wv][2] = ss; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < FY_WAVES; ++k) v += s_st[k][threadIdx.x];
        p.stats[3 * (size_t)blockIdx.x + threadIdx.x] = v;
    }
}

inline int dogf_radius(float sigma) { return (int)(4.0f * sigma + 0.5f); }

template <int R>
void fill_ftaps(float sigma, FTaps<R>& t) {
    const int rs = dogf_radius(sigma);
    double tmp[R + 1], sum = 0;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int d = 0; d <= R; ++d) {
        tmp[d] = d <= rs ? exp(c * (double)d * (double)d) : 0.0;
        sum += d == 0 ? tmp[d] : 2.0 * tmp[d];
    }
    for (int d = 0; d <= R; ++d) t.w[d] = (float)(tmp[d] / sum);
}

template <int RA, int RB>
int launch_dogf(const DogfParams& p, float sa, float sb, hipStream_t st) {
    FTaps<RA> wa;
    FTaps<RB> wb;
    fill_ftaps<RA>(sa, wa);
    fill_ftaps<RB>(sb, wb);
    hipLaunchKernelGGL((dogf_zx_kernel<RA, RB>), dim3((unsigned)(p.yhi - p.ylo)), dim3(FZ_T), 0, st, p, wa, wb);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL((dogf_y_kernel<RA, RB>), dim3(p.n_wg), dim3(FY_T), 0, st, p, wa, wb);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

// The fused chain takes: two sigmas (small, large) with radii <= 12 / <= 20, a 3x3 window, rows of 64..512 voxels in
// multiples of 4, and a zeroed border at least as wide as the larger radius (no 'reflect' in y; utils/image.py:141-143:
// 30 or 60 voxels) that leaves a live box.
bool mi_dogf_usable(const float* rec, const float* g1, const float* g2, const float* nms_out, int D, int H, int W,
                    float s1, float s2, int k, int bz, int bxy) {
    if (getenv("MI_NO_DOGF")) return false;
    auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const int r1 = dogf_radius(s1), r2 = dogf_radius(s2);
    if ((size_t)D * H * W >= ((size_t)1 << 29)) return false;          // 32-bit byte offsets inside the volume
    const int rbc = r2 <= 16 ? 16 : 20;                    // the z march reads rows [bz - RB, D - bz + RB + 11): one reflection
    if (D < rbc + 12) return false;
    return k == 3 && s1 <= s2 && (W & 3) == 0 && W >= 64 && W <= FZ_T && r1 >= 1 && r1 <= 12 && r2 >= r1 && r2 <= 20 &&
           bxy >= r2 && bxy >= 1 && 2 * bxy < H && 2 * bxy < W && bz >= 0 && 2 * bz < D && al(rec) && al(g1) && al(g2) && al(nms_out);
}

int mi_dogf_own() { return FY_OWN; }

DogfGrid mi_dogf_grid(int D, int H, int W, int bz, int bxy) {
    DogfGrid g = {};
    if (2 * bxy >= H || 2 * bxy >= W || 2 * bz >= D) return g;
    const int rows = H - 2 * bxy, planes = D - 2 * bz;
    g.n_strips = mi_cdiv(W - 2 * bxy, FY_OWN);
    // chunks of rows: a chunk warms its rings up over 2 R rows it does not own, so only as many as the chip has wave slots
    // for (4 waves per SIMD = 4096): one round of waves, every SIMD with 3 - 4 of them
    int nyc = 1;
    const char* e = getenv("MI_DOGF_NYC");
    if (e) { nyc = atoi(e); if (nyc < 1) nyc = 1; }
    else while ((long)planes * g.n_strips * nyc * 2 <= 4096 && rows / (nyc * 2) >= 48) nyc *= 2;
    g.ychunk = mi_cdiv(rows, nyc);
    g.n_ychunks = mi_cdiv(rows, g.ychunk);
    g.n_seg = (unsigned)((long)planes * g.n_ychunks * g.n_strips);
    g.n_wg = (unsigned)mi_cdiv(g.n_seg, FY_WAVES);
    g.seg_cap = (unsigned)(g.ychunk * 16 + 64);          // xy-NMS survivors: at most one per 2x2 patch without plateaus
    return g;
}

int mi_launch_dogf(DogfParams p, const DogfGrid& g, float s1, float s2, hipStream_t st) {
    p.vol_bytes = 4u * (unsigned)((size_t)p.D * p.H * p.W);
    p.ychunk = g.ychunk; p.n_ychunks = g.n_ychunks; p.n_strips = g.n_strips; p.n_seg = g.n_seg; p.n_wg = g.n_wg;
    p.seg_cap = g.seg_cap;
    const int r1 = dogf_radius(s1), r2 = dogf_radius(s2);
    p.ylo = p.by - (r2 <= 16 ? 16 : 20); p.yhi = p.H - p.by + (r2 <= 16 ? 16 : 20);
    if (p.ylo < 0) p.ylo = 0;
    if (p.yhi > p.H) p.yhi = p.H;
    if (r1 <= 8 && r2 <= 16) return launch_dogf<8, 16>(p, s1, s2, st);
    if (r1 <= 8) return launch_dogf<8, 20>(p, s1, s2, st);
    if (r2 <= 16) return launch_dogf<12, 16>(p, s1, s2, st);
    return launch_dogf<12, 20>(p, s1, s2, st);
}
