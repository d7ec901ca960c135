// Pooled 3-D NMS as a z-marching stencil, fused with the value producer (plain / sigmoid+clamp /
// DoG+border) and with candidate compaction + score histogram; top-K selection kernels.
//
// Replaces (reference, cet_pick/...): models/utils.py:167-169 `_sigmoid`, models/decode.py:11-33
// `_nms_xy/_nms_z/_nms`, utils/image.py:81-105 (same + k*k*k `_nms`), models/decode.py:82-92 `_topk`,
// models/decode.py:123-155 `tomo_decode`.
//
// HBM-bound: one read of the input plane tile (+halo from L2) and at most one write per voxel.
// Layout: a 256-thread workgroup owns a 16(y) x 64(x) column of the volume and marches over a
// z-chunk; the current plane (with xy halo) sits in LDS (double buffered, one barrier per plane),
// the z-window lives in registers.  Each thread owns 4 consecutive x (16-B loads/stores).
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int TY = 16, TX = 64, NT = 256;
constexpr int LW = TX + 8;          // LDS row stride: tile centre starts at col 4 (16-B aligned)
constexpr int CAND_LDS = 2048;      // per-block candidate staging (flushed above 1024)

template <int KXY>
struct PlaneBuf {
    static constexpr int PXY = KXY / 2;
    static constexpr int LH = TY + 2 * PXY;
    float v[2][LH][LW];
};

__device__ __forceinline__ float produce(const MarchParams& p, long idx, int z, int y, int x) {
    if (p.mode == MI_LOAD_SIGMOID) {
        float v = p.in[idx];
        float s = 1.0f / (1.0f + expf(-v));
        return s != s ? s : fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);       // torch.clamp keeps a NaN
    } else if (p.mode == MI_LOAD_DOG) {
        bool border = (z < p.bz) | (z >= p.D - p.bz) | (y < p.by) | (y >= p.H - p.by) |
                      (x < p.bx) | (x >= p.W - p.bx);
        return border ? 0.0f : (p.in2[idx] - p.in[idx]);
    }
    return p.in[idx];
}

template <int KZ, int KXY>
__global__ __launch_bounds__(NT) void nms_march_kernel(MarchParams p) {
    constexpr int PZ = KZ / 2, PXY = KXY / 2;
    constexpr int LH = TY + 2 * PXY;
    constexpr int HW_ = TX + 2 * PXY;                       // halo-inclusive tile width
    constexpr int N_TOPBOT = 2 * PXY * HW_;
    constexpr int N_HALO = N_TOPBOT + TY * 2 * PXY;
    constexpr int NCOL = 4 + 2 * PXY;

    __shared__ __attribute__((aligned(16))) float plane[2][LH][LW];
    __shared__ uint2 cbuf[CAND_LDS];
    __shared__ unsigned lhist[MI_HIST_BINS];
    __shared__ unsigned s_ccount, s_cbase;
    __shared__ double s_red[3][NT / 64];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int xt0 = blockIdx.x * TX, yt0 = blockIdx.y * TY;
    const int z0 = blockIdx.z * p.zchunk;
    const int zend = min(z0 + p.zchunk, p.D);
    const int x0 = xt0 + 4 * tx, y = yt0 + ty;
    const long HW = (long)p.H * p.W;
    const bool row_ok = y < p.H;
    const bool vec_ok = p.vec_ok && (x0 + 3 < p.W);
    const bool emit = p.cands != nullptr;

    if (tid == 0) s_ccount = 0;
    if (p.hist) for (int i = tid; i < MI_HIST_BINS; i += NT) lhist[i] = 0;

    const float NEG = -INFINITY;
    float ringM[KZ][4];       // xy-pooled planes (fiber: xy-NMS'd values)
    float ringC[PZ + 1][4];   // plane centres (newest last)
#pragma unroll
    for (int k = 0; k < KZ; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) ringM[k][i] = NEG;
#pragma unroll
    for (int k = 0; k <= PZ; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) ringC[k][i] = 0.f;

    double st_n = 0, st_s = 0, st_ss = 0;
    __syncthreads();

    // ---- software pipeline: the raw values of plane zz+1 are in flight while plane zz is pooled
    constexpr int HALO_PT = (N_HALO + NT - 1) / NT;    // halo elements per thread
    float raw_c[4], raw_c2[4];                          // centre values (in, in2)
    float raw_h[HALO_PT > 0 ? HALO_PT : 1], raw_h2[HALO_PT > 0 ? HALO_PT : 1];
    // static halo geometry of this thread
    int h_row[HALO_PT > 0 ? HALO_PT : 1], h_col[HALO_PT > 0 ? HALO_PT : 1];
    bool h_ok[HALO_PT > 0 ? HALO_PT : 1];
#pragma unroll
    for (int k = 0; k < HALO_PT; ++k) {
        const int hidx = tid + k * NT;
        int row = 0, col = 0;
        if (hidx < N_HALO) {
            if (hidx < N_TOPBOT) {
                int r = hidx / HW_;
                col = hidx - r * HW_;
                row = (r < PXY) ? r : (TY + r);
            } else {
                int hh = hidx - N_TOPBOT;
                int r = hh / (2 * PXY > 0 ? 2 * PXY : 1);
                int cc = hh - r * (2 * PXY);
                row = PXY + r;
                col = (cc < PXY) ? cc : (TX + cc);
            }
        }
        h_row[k] = row; h_col[k] = col;
        const int gy = yt0 - PXY + row, gx = xt0 - PXY + col;
        h_ok[k] = hidx < N_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    }
    auto fetch = [&](int zz) {
        if (zz < 0 || zz >= p.D) return;
        const long zb = (long)zz * HW;
        const long base = zb + (long)y * p.W + x0;
        if (row_ok) {
            if (vec_ok) {
                float4 v = *reinterpret_cast<const float4*>(p.in + base);
                raw_c[0] = v.x; raw_c[1] = v.y; raw_c[2] = v.z; raw_c[3] = v.w;
                if (p.mode == MI_LOAD_DOG) {
                    float4 g = *reinterpret_cast<const float4*>(p.in2 + base);
                    raw_c2[0] = g.x; raw_c2[1] = g.y; raw_c2[2] = g.z; raw_c2[3] = g.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (x0 + i < p.W) {
                        raw_c[i] = p.in[base + i];
                        if (p.mode == MI_LOAD_DOG) raw_c2[i] = p.in2[base + i];
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < HALO_PT; ++k)
            if (h_ok[k]) {
                const long o = zb + (long)(yt0 - PXY + h_row[k]) * p.W + (xt0 - PXY + h_col[k]);
                raw_h[k] = p.in[o];
                if (p.mode == MI_LOAD_DOG) raw_h2[k] = p.in2[o];
            }
    };
    auto transform = [&](float v, float v2, int z, int yy, int xx) -> float {
        if (p.mode == MI_LOAD_SIGMOID) {
            // v_exp_f32 + v_rcp_f32 (about 1e-6 relative on the clamped range, monotone): the exact
            // expf + IEEE divide made this HBM-bound pass VALU-bound
            float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            return sg != sg ? sg : fminf(fmaxf(sg, 1e-4f), 1.0f - 1e-4f);   // torch.clamp keeps a NaN
        } else if (p.mode == MI_LOAD_DOG) {
            bool border = (z < p.bz) | (z >= p.D - p.bz) | (yy < p.by) | (yy >= p.H - p.by) |
                          (xx < p.bx) | (xx >= p.W - p.bx);
            return border ? 0.0f : (v2 - v);
        }
        return v;
    };

    fetch(z0 - PZ);
    int it = 0;
    for (int zz = z0 - PZ; zz < zend + PZ; ++zz, ++it) {
        const int b = it & 1;
        const bool plane_in = (zz >= 0) && (zz < p.D);
        float c[4] = {NEG, NEG, NEG, NEG};
        if (plane_in) {
            // ---- commit the fetched plane (+halo) to LDS
            const long base = (long)zz * HW + (long)y * p.W + x0;
            if (row_ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (x0 + i < p.W) c[i] = transform(raw_c[i], raw_c2[i], zz, y, x0 + i);
            }
            *reinterpret_cast<float4*>(&plane[b][ty + PXY][4 + 4 * tx]) = make_float4(c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int k = 0; k < HALO_PT; ++k) {
                if (tid + k * NT < N_HALO) {
                    float v = NEG;
                    if (h_ok[k]) v = transform(raw_h[k], raw_h2[k], zz, yt0 - PXY + h_row[k], xt0 - PXY + h_col[k]);
                    plane[b][h_row[k]][h_col[k] + 4 - PXY] = v;
                }
            }
            // pre-NMS value (sigmoid heat-map) is an output of the fused decode
            if (p.val_out && zz >= z0 && zz < zend && row_ok) {
                if (vec_ok) {
                    *reinterpret_cast<float4*>(p.val_out + base) = make_float4(c[0], c[1], c[2], c[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x0 + i < p.W) p.val_out[base + i] = c[i];
                }
            }
        }
        if (zz + 1 < zend + PZ) fetch(zz + 1);      // in flight across the barrier and the pooling below
        __syncthreads();

        // ---- flush the candidate staging buffer when it could overflow during this plane
        if (emit && s_ccount >= 1024) {
            unsigned cnt = s_ccount;
            if (tid == 0) s_cbase = atomicAdd(p.cand_count, cnt);
            __syncthreads();
            unsigned cb = s_cbase;
            for (unsigned i = tid; i < cnt; i += NT)
                if (cb + i < p.cand_cap) p.cands[cb + i] = cbuf[i];
            __syncthreads();
            if (tid == 0) s_ccount = 0;
            __syncthreads();
        }

        // ---- xy pooling of this plane from LDS
        float m[4] = {NEG, NEG, NEG, NEG};
        if (plane_in) {
            float colmax[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) colmax[j] = NEG;
#pragma unroll
            for (int dy = 0; dy < KXY; ++dy) {
                const float* row = &plane[b][ty + dy][0];
                float4 cv = *reinterpret_cast<const float4*>(row + 4 + 4 * tx);
                colmax[PXY + 0] = fmaxf(colmax[PXY + 0], cv.x);
                colmax[PXY + 1] = fmaxf(colmax[PXY + 1], cv.y);
                colmax[PXY + 2] = fmaxf(colmax[PXY + 2], cv.z);
                colmax[PXY + 3] = fmaxf(colmax[PXY + 3], cv.w);
#pragma unroll
                for (int j = 0; j < PXY; ++j) {
                    colmax[j] = fmaxf(colmax[j], row[4 + 4 * tx - PXY + j]);
                    colmax[PXY + 4 + j] = fmaxf(colmax[PXY + 4 + j], row[4 + 4 * tx + 4 + j]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int dx = 0; dx < KXY; ++dx) m[i] = fmaxf(m[i], colmax[i + dx]);
            if (p.fiber) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = (m[i] == c[i]) ? c[i] : 0.f;
            }
        }
        // ---- shift the z window
#pragma unroll
        for (int k = 0; k + 1 < KZ; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) ringM[k][i] = ringM[k + 1][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) ringM[KZ - 1][i] = m[i];
#pragma unroll
        for (int k = 0; k < PZ; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) ringC[k][i] = ringC[k + 1][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) ringC[PZ][i] = c[i];

        // ---- emit plane zo = zz - PZ
        const int zo = zz - PZ;
        if (zo >= z0 && zo < zend && row_ok) {
            const long obase = (long)zo * HW + (long)y * p.W + x0;
            float out[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float hm = ringM[0][i];
#pragma unroll
                for (int k = 1; k < KZ; ++k) hm = fmaxf(hm, ringM[k][i]);
                float cc = p.fiber ? ringM[PZ][i] : ringC[0][i];
                out[i] = (hm == cc) ? cc : 0.f;
            }
            if (p.accumulate && p.nms_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (x0 + i < p.W) out[i] = fmaxf(out[i], p.nms_out[obase + i]);
            }
            if (p.nms_out) {
                if (vec_ok) {
                    *reinterpret_cast<float4*>(p.nms_out + obase) = make_float4(out[0], out[1], out[2], out[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x0 + i < p.W) p.nms_out[obase + i] = out[i];
                }
            }
            if (emit || p.stats) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (x0 + i < p.W && out[i] > 0.f) {
                        if (emit) {
                            unsigned bits = __float_as_uint(out[i]);
                            unsigned slot = atomicAdd(&s_ccount, 1u);
                            cbuf[slot] = make_uint2(bits, (unsigned)(obase + i));
                            if (p.hist) atomicAdd(&lhist[bits >> MI_HIST_SHIFT], 1u);
                        }
                        double dv = (double)out[i];
                        st_n += 1.0; st_s += dv; st_ss += dv * dv;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (emit) {
        unsigned cnt = s_ccount;
        if (cnt > 0) {
            if (tid == 0) s_cbase = atomicAdd(p.cand_count, cnt);
            __syncthreads();
            unsigned cb = s_cbase;
            for (unsigned i = tid; i < cnt; i += NT)
                if (cb + i < p.cand_cap) p.cands[cb + i] = cbuf[i];
        }
        if (p.hist)
            for (int i = tid; i < MI_HIST_BINS; i += NT) {
                unsigned hcount = lhist[i];
                if (hcount) atomicAdd(&p.hist[i], hcount);
            }
    }
    if (p.stats) {
        double a = wave_sum(st_n), s = wave_sum(st_s), ss = wave_sum(st_ss);
        if ((tid & 63) == 0) { s_red[0][tid >> 6] = a; s_red[1][tid >> 6] = s; s_red[2][tid >> 6] = ss; }
        __syncthreads();
        if (tid == 0) {
            double A = 0, S = 0, SS = 0;
            for (int w = 0; w < NT / 64; ++w) { A += s_red[0][w]; S += s_red[1][w]; SS += s_red[2][w]; }
            long bid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            p.stats[3 * bid + 0] = A; p.stats[3 * bid + 1] = S; p.stats[3 * bid + 2] = SS;
        }
    }
}

template <int KZ>
int launch_kxy(const MarchParams& p, int kxy, dim3 grid, hipStream_t s) {
    switch (kxy) {
        case 1: hipLaunchKernelGGL((nms_march_kernel<KZ, 1>), grid, dim3(NT), 0, s, p); break;
        case 3: hipLaunchKernelGGL((nms_march_kernel<KZ, 3>), grid, dim3(NT), 0, s, p); break;
        case 5: hipLaunchKernelGGL((nms_march_kernel<KZ, 5>), grid, dim3(NT), 0, s, p); break;
        case 7: hipLaunchKernelGGL((nms_march_kernel<KZ, 7>), grid, dim3(NT), 0, s, p); break;
        default: return MI_E_ARG;
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

__global__ void sigmoid_clamp_kernel(float* x, float* y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = 1.0f / (1.0f + expf(-x[i]));
        x[i] = s;
        y[i] = s != s ? s : fminf(fmaxf(s, 1e-4f), 1.0f - 1e-4f);       // torch.clamp keeps a NaN
    }
}

// ---------------------------------------------------------------------------------------------
// top-K: histogram threshold -> filter -> single-block sort + emit
// ---------------------------------------------------------------------------------------------
__device__ int hist_threshold_bin(const unsigned* hist, int K, unsigned* s_scan /*256*/, int tid) {
    // 256 threads, 8 bins each; returns the highest bin T with count(bins >= T) >= K (0 if none)
    constexpr int PER = MI_HIST_BINS / 256;
    unsigned mine = 0;
    for (int b = 0; b < PER; ++b) mine += hist[tid * PER + b];
    s_scan[tid] = mine;
    __syncthreads();
    // suffix sum (inclusive) by Hillis-Steele
    for (int off = 1; off < 256; off <<= 1) {
        unsigned v = (tid + off < 256) ? s_scan[tid + off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    __shared__ int s_T;
    if (tid == 0) s_T = 0;
    __syncthreads();
    unsigned incl = s_scan[tid];
    unsigned above = (tid + 1 < 256) ? s_scan[tid + 1] : 0u;
    if (incl >= (unsigned)K && above < (unsigned)K) {
        unsigned acc = above;
        int T = tid * PER;
        for (int b = PER - 1; b >= 0; --b) {
            acc += hist[tid * PER + b];
            if (acc >= (unsigned)K) { T = tid * PER + b; break; }
        }
        s_T = T;
    }
    __syncthreads();
    return s_T;
}

// The same threshold by ONE wave without a barrier: lane l owns bins [32 l, 32 l + 32) (eight 16-byte loads), the
// suffix sums cross the lanes by shuffles.  Every wave of the filter computes it for itself.
__device__ __forceinline__ int hist_threshold_bin_wave(const unsigned* hist, int K, int lane) {
    constexpr int PER = MI_HIST_BINS / 64;      // 32
    unsigned h[PER];
    const uint4* src = reinterpret_cast<const uint4*>(hist + lane * PER);
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        const uint4 v = src[q];
        h[4 * q] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
    }
    unsigned mine = 0;
#pragma unroll
    for (int b = 0; b < PER; ++b) mine += h[b];
    unsigned incl = mine;                        // inclusive suffix sum over lanes >= this one
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += v;
    }
    const unsigned above = incl - mine;
    int T = -1;
    if (incl >= (unsigned)K && above < (unsigned)K) {
        unsigned acc = above;
        bool done = false;
        T = lane * PER;
#pragma unroll
        for (int b = PER - 1; b >= 0; --b)
            if (!done) {
                acc += h[b];
                if (acc >= (unsigned)K) { T = lane * PER + b; done = true; }
            }
    }
    // exactly one lane (or none: fewer than K candidates) holds T >= 0
    const unsigned long long m = __ballot(T >= 0);
    if (!m) return 0;
    return __shfl(T, __ffsll((long long)m) - 1, 64);
}

// survivors of the histogram threshold -> sel; one atomic per wave (a per-survivor atomic on the single counter
// serialised ~2000 returning atomics: 15 us for this kernel)
__device__ __forceinline__ void append_selected(uint2 c, bool keep, DecodeHeader* hdr, uint2* sel, unsigned sel_cap,
                                                int lane) {
    const unsigned long long km = __ballot(keep);
    if (!km) return;
    unsigned base = 0;
    const int leader = __ffsll((long long)km) - 1;
    if (lane == leader) base = atomicAdd(&hdr->sel_count, (unsigned)__popcll(km));
    base = __shfl(base, leader, 64);
    if (keep) {
        const unsigned slot = base + (unsigned)__popcll(km & ((1ull << lane) - 1ull));
        if (slot < sel_cap) sel[slot] = c;
    }
}

__global__ __launch_bounds__(256) void topk_filter_kernel(const uint2* cands, DecodeHeader* hdr,
                                                         unsigned cand_cap, uint2* sel,
                                                         unsigned sel_cap, int K) {
    __shared__ unsigned s_scan[256];
    const int tid = threadIdx.x, lane = tid & 63;
    int T = hist_threshold_bin(hdr->hist, K, s_scan, tid);
    unsigned n = min(hdr->cand_count, cand_cap);
    for (unsigned i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {       // wave-uniform trip count
        const unsigned i = i0 + tid;
        const uint2 c = i < n ? cands[i] : make_uint2(0u, 0u);
        append_selected(c, i < n && (int)(c.x >> MI_HIST_SHIFT) >= T, hdr, sel, sel_cap, lane);
    }
}

// segmented candidates (infer_peak3.hip): a wave walks whole segments; the workgroup's survivors are staged in LDS and
// leave with ONE returning atomic per workgroup (a returning atomic on one word costs ~11 ns: one per wave and
// 64-candidate step - ~1400 of them - made this kernel 15 us)
constexpr int FSEG_STAGE = 2048;
__global__ __launch_bounds__(256) void topk_filter_seg_kernel(const uint2* cands, const unsigned* seg_count,
                                                             unsigned n_seg, unsigned seg_cap, DecodeHeader* hdr,
                                                             uint2* sel, unsigned sel_cap, int K) {
    __shared__ uint2 stage[FSEG_STAGE];
    __shared__ unsigned s_n, s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0;
    const int T = hist_threshold_bin_wave(hdr->hist, K, lane);
    __syncthreads();
    // The kernel is a chain of dependent memory round trips (count -> entries -> LDS -> one atomic -> store), not a
    // bandwidth problem: a wave fetches the counts of both its segments first and then up to 8 x 64 entries of a
    // segment back to back, before it looks at any of them.
    const unsigned n_waves = gridDim.x * 4;
    const unsigned w0 = blockIdx.x * 4 + (tid >> 6);
    constexpr int PF = 8;
    auto stage_one = [&](uint2 c, bool keep) {
        const unsigned long long km = __ballot(keep);
        if (!km) return;
        unsigned b0 = 0;
        const int leader = __ffsll((long long)km) - 1;
        if (lane == leader) b0 = atomicAdd(&s_n, (unsigned)__popcll(km));      // LDS
        b0 = __shfl(b0, leader, 64);
        if (keep) {
            const unsigned slot = b0 + (unsigned)__popcll(km & ((1ull << lane) - 1ull));
            if (slot < FSEG_STAGE) stage[slot] = c;
            else {                                                              // stage full (plateaus): go direct
                const unsigned g = atomicAdd(&hdr->sel_count, 1u);
                if (g < sel_cap) sel[g] = c;
            }
        }
    };
    for (unsigned sg = w0; sg < n_seg; sg += 2 * n_waves) {
        const unsigned sg2 = sg + n_waves;
        const unsigned cnt_a = min(seg_count[sg], seg_cap);
        const unsigned cnt_b = sg2 < n_seg ? min(seg_count[sg2], seg_cap) : 0u;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned cnt = half ? cnt_b : cnt_a;
            const uint2* base = cands + (size_t)(half ? sg2 : sg) * seg_cap;
            for (unsigned i0 = 0; i0 < cnt; i0 += 64 * PF) {
                uint2 c[PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const unsigned i = i0 + 64 * u + lane;
                    c[u] = i < cnt ? base[i] : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    if (i0 + 64 * u >= cnt) break;                              // (wave-uniform)
                    const unsigned i = i0 + 64 * u + lane;
                    stage_one(c[u], i < cnt && (int)(c[u].x >> MI_HIST_SHIFT) >= T);
                }
            }
        }
    }
    __syncthreads();
    const unsigned n = min(s_n, (unsigned)FSEG_STAGE);
    if (n == 0) return;
    if (tid == 0) s_base = atomicAdd(&hdr->sel_count, n);
    __syncthreads();
    const unsigned gb = s_base;
    for (unsigned i = tid; i < n; i += 256)
        if (gb + i < sel_cap) sel[gb + i] = stage[i];
}

__device__ __forceinline__ void emit_det(float* dets, int r, unsigned long long key, int H, int W,
                                         bool valid) {
    float* o = dets + 5 * (long)r;
    if (!valid) { o[0] = 0.25f; o[1] = 0.25f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; return; }
    float score = __uint_as_float((unsigned)(key >> 32));
    unsigned idx = ~(unsigned)(key & 0xffffffffu);
    // `_convert_1d_to_3d` (decode.py:35-41): float32 division, then integer remainder
    int hw = H * W;
    int z = (int)floorf(__fdiv_rn((float)idx, (float)hw));
    int t = (int)idx - z * hw;
    float yf = floorf(__fdiv_rn((float)t, (float)W));
    int x = t % W;
    if (x < 0) x += W;
    o[0] = (float)x + 0.25f; o[1] = yf + 0.25f; o[2] = (float)z; o[3] = score; o[4] = score;
}

// Final selection, MI_FINAL_BLOCKS workgroups of 1024 threads.
//   * n_sel <= MI_RANK_MAX (the usual case: K plus the occupants of the threshold bin): no sort at all.  Every workgroup
//     stages the selected keys in LDS, a thread owns one key and counts the keys greater than it (broadcast LDS reads,
//     two keys per ds_read_b128): that count IS its output row.  Keys are unique (the voxel index is part of the key), so
//     ranks are a permutation; the work is n_sel^2 / 2 compares spread over 16 CUs, ~4 us at n_sel = 1100 against 20 us for
//     the single-workgroup bitonic network it replaces.
//   * otherwise workgroup 0 alone: the filtered set fits LDS -> sort; or (degenerate plateaus) exact 64-bit radix select
//     over the whole candidate list, then sort the survivors.  Candidates are n_seg segments of seg_cap entries (a linear
//     list = one segment whose count is hdr->cand_count).
// The last workgroup to finish leaves the header zeroed when `self_clean` is set (see mi_sigmoid_nms_topk).
constexpr int MI_RANK_MAX = 8192;
constexpr int MI_FINAL_BLOCKS = 16;

__global__ __launch_bounds__(1024) void topk_final_kernel(const uint2* cands, const unsigned* seg_count, unsigned n_seg,
                                                          unsigned seg_cap, DecodeHeader* hdr, const uint2* sel,
                                                          int K, int H, int W, float* dets,
                                                          int* n_valid_out, int self_clean) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // MI_SEL_CAP entries
    __shared__ unsigned s_hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_remaining, s_n, s_ticket;
    const int tid = threadIdx.x;
    const unsigned n_sel = hdr->sel_count;
    auto finish = [&]() {
        if (!self_clean) return;
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(&hdr->pad0, 1u);           // every workgroup has read the header by now
        __syncthreads();
        if (s_ticket == gridDim.x - 1) {                                // the last one out cleans up for the next call
            for (int i = tid; i < MI_HIST_BINS; i += 1024) hdr->hist[i] = 0u;
            if (tid == 0) { hdr->cand_count = 0u; hdr->sel_count = 0u; hdr->pad0 = 0u; hdr->pad1 = 0u; }
        }
    };
    if (n_sel <= (unsigned)MI_RANK_MAX) {
        const int n = (int)n_sel;
        const int n4 = (n + 3) & ~3;
        for (int i = tid; i < n4; i += 1024) {
            unsigned long long k = 0ull;                                // padding: below every real key (scores are > 0)
            if (i < n) { const uint2 c = sel[i]; k = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y); }
            keys[i] = k;
        }
        __syncthreads();
        // PARTS consecutive lanes share one key and take every PARTS-th key of the list each (a lone thread per key
        // would walk all n keys: 4 VALU instructions per key at one wave per SIMD = 7 us at n = 1100)
        const int total = (int)gridDim.x * 1024;
        int lp = 0;                                                     // PARTS = 2^lp <= 16, n * PARTS <= total
        while (lp < 4 && (n << (lp + 1)) <= total) ++lp;
        const int parts = 1 << lp;
        const int gid = (int)blockIdx.x * 1024 + tid;
        const int mine_i = gid >> lp, part = gid & (parts - 1);
        const bool have = mine_i < n;
        const unsigned long long mine = have ? keys[mine_i] : ~0ull;
        int rank = 0;
        // keys are < 2^63 (positive floats in the upper word): key_j > mine  <=>  the sign of mine - key_j
        for (int j = part; j < n4; j += parts) rank += (int)((unsigned long long)(mine - keys[j]) >> 63);
        for (int o = 1; o < parts; o <<= 1) rank += __shfl_xor(rank, o, 64);
        if (have && part == 0 && rank < K) emit_det(dets, rank, mine, H, W, true);
        const int n_valid = min(n, K);
        for (int r = n_valid + (int)blockIdx.x * 1024 + tid; r < K; r += (int)gridDim.x * 1024) emit_det(dets, r, 0ull, H, W, false);
        if (blockIdx.x == 0 && tid == 0 && n_valid_out) *n_valid_out = n_valid;
        finish();
        return;
    }
    if (blockIdx.x != 0) { finish(); return; }
    int n = 0;
    if (n_sel <= MI_SEL_CAP) {
        n = (int)n_sel;
        for (int i = tid; i < n; i += 1024) {
            uint2 c = sel[i];
            keys[i] = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y);
        }
    } else {
        // visit every candidate: one segment -> all threads stride over it; many -> one wave per segment
        auto for_each = [&](auto&& f) {
            if (n_seg == 1) {
                const unsigned cnt = min(seg_count[0], seg_cap);
                for (unsigned i = tid; i < cnt; i += 1024) f(cands[i]);
            } else {
                for (unsigned sg = tid >> 6; sg < n_seg; sg += 16) {
                    const unsigned cnt = min(seg_count[sg], seg_cap);
                    const uint2* base = cands + (size_t)sg * seg_cap;
                    for (unsigned i = tid & 63; i < cnt; i += 64) f(base[i]);
                }
            }
        };
        // exact select of the K-th largest key among cands (keys are unique: idx is unique)
        if (tid == 0) { s_prefix = 0ull; s_remaining = (unsigned)K; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            unsigned long long prefix = s_prefix;
            unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
            for_each([&](uint2 c) {
                unsigned long long k = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y);
                if ((k & himask) == prefix) atomicAdd(&s_hist[(unsigned)(k >> shift) & 255u], 1u);
            });
            __syncthreads();
            if (tid == 0) {
                unsigned rem = s_remaining, acc = 0;
                int d = 255;
                for (; d > 0; --d) {
                    if (acc + s_hist[d] >= rem) break;
                    acc += s_hist[d];
                }
                s_remaining = rem - acc;
                s_prefix = prefix | ((unsigned long long)d << shift);
            }
            __syncthreads();
        }
        unsigned long long thr = s_prefix;   // K-th largest key (or smallest key if fewer than K)
        if (tid == 0) s_n = 0;
        __syncthreads();
        for_each([&](uint2 c) {
            unsigned long long k = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y);
            if (k >= thr) {
                unsigned slot = atomicAdd(&s_n, 1u);
                if (slot < MI_SEL_CAP) keys[slot] = k;
            }
        });
        __syncthreads();
        n = (int)min(s_n, (unsigned)MI_SEL_CAP);
    }
    int P = 1024;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += 1024) keys[i] = 0ull;
    block_sort_desc_fast(keys, P, tid, 1024);
    int n_valid = min(n, K);
    for (int r = tid; r < K; r += 1024) emit_det(dets, r, r < n_valid ? keys[r] : 0ull, H, W, r < n_valid);
    if (tid == 0 && n_valid_out) *n_valid_out = n_valid;
    finish();
}

// zero the header of a decode workspace (mi_decode_workspace_init)
__global__ void zero_header_kernel(DecodeHeader* hdr) {
    for (int i = threadIdx.x; i < MI_HIST_BINS; i += blockDim.x) hdr->hist[i] = 0u;
    if (threadIdx.x == 0) { hdr->cand_count = 0u; hdr->sel_count = 0u; hdr->pad0 = 0u; hdr->pad1 = 0u; }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
int mi_launch_march(MarchParams p, int kz, int kxy, hipStream_t s) {
    if (p.D <= 0 || p.H <= 0 || p.W <= 0) return MI_E_ARG;
    if (!(kz == 1 || kz == 3 || kz == 5 || kz == 7)) return MI_E_ARG;
    p.vec_ok = ((p.W & 3) == 0) && (((uintptr_t)p.in & 15) == 0) &&
               (p.in2 == nullptr || ((uintptr_t)p.in2 & 15) == 0) &&
               (p.val_out == nullptr || ((uintptr_t)p.val_out & 15) == 0) &&
               (p.nms_out == nullptr || ((uintptr_t)p.nms_out & 15) == 0);
    dim3 grid = mi_march_grid(p.D, p.H, p.W, &p.zchunk);
    switch (kz) {
        case 1: return launch_kxy<1>(p, kxy, grid, s);
        case 3: return launch_kxy<3>(p, kxy, grid, s);
        case 5: return launch_kxy<5>(p, kxy, grid, s);
        default: return launch_kxy<7>(p, kxy, grid, s);
    }
}

dim3 mi_march_grid(int D, int H, int W, int* zchunk_out) {
    int gx = mi_cdiv(W, TX), gy = mi_cdiv(H, TY);
    // z-chunks: >= ~768 workgroups (3 per CU) but chunks of >= 16 planes - every workgroup pays
    // (2*PZ halo planes + one histogram flush of global atomics), so fewer, longer marches win
    int zc = D;
    long tiles = (long)gx * gy;
    int target = 768;   // (MI_MARCH_TARGET overrides; 256..2048 measured within 5 % of each other)
    if (const char* e = getenv("MI_MARCH_TARGET")) target = atoi(e);
    while (zc > 16 && tiles * mi_cdiv(D, zc) < target) zc = (zc + 1) / 2;
    if (zc < 1) zc = 1;
    *zchunk_out = zc;
    return dim3(gx, gy, mi_cdiv(D, zc));
}

extern "C" int mi_sigmoid_clamp(float* x, float* y, size_t n, mi_stream_t stream) {
    if (!x || !y) return MI_E_ARG;
    if (n == 0) return MI_OK;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(sigmoid_clamp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_nms3d(const float* heat, float* out, int D, int H, int W, int kd, int kh,
                        mi_stream_t stream) {
    if (!heat || !out || heat == out) return MI_E_ARG;
    if (kd == 3 && kh == 3 && mi_peak3_usable(heat, nullptr, out, D, H, W)) {
        Peak3Params q = {};
        q.in = heat; q.nms_out = out; q.D = D; q.H = H; q.W = W;
        return mi_launch_peak3(q, mi_peak3_grid(D, H, W), false, (hipStream_t)stream);
    }
    MarchParams p = {};
    p.in = heat; p.nms_out = out; p.mode = MI_LOAD_PLAIN;
    p.D = D; p.H = H; p.W = W;
    return mi_launch_march(p, kd, kh, (hipStream_t)stream);
}

extern "C" size_t mi_decode_workspace_bytes(int D, int H, int W, int K) {
    (void)K;
    size_t n = (size_t)D * H * W;
    // candidate storage: the linear list of the generic march (n entries) or the per-wave segments of the register
    // march (padded to whole strips / z-chunks)
    const Peak3Grid g = mi_peak3_grid(D, H, W);
    size_t n_cand = std::max(n, (size_t)g.n_seg * g.seg_cap);
    return mi_align_up(sizeof(DecodeHeader), 256) + mi_align_up(n_cand * sizeof(uint2), 256) +
           mi_align_up((size_t)MI_SEL_CAP * sizeof(uint2), 256) + mi_align_up((size_t)g.n_seg * sizeof(unsigned), 256) +
           mi_decode1_extra_bytes(D, H, W);
}

extern "C" int mi_sigmoid_nms_topk(const float* logits, float* heat_out, int D, int H, int W,
                                   int k, int fiber, int apply_sigmoid, int K, float* dets,
                                   int32_t* n_valid_out, void* workspace, size_t workspace_bytes,
                                   mi_stream_t stream) {
    if (!logits || !dets || !workspace || K <= 0 || K > MI_SEL_CAP) return MI_E_ARG;
    if (heat_out == logits) return MI_E_ARG;
    if ((size_t)D * H * W >= (1ull << 32)) return MI_E_UNSUPPORTED;
    if (workspace_bytes < mi_decode_workspace_bytes(D, H, W, K)) return MI_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    size_t n = (size_t)D * H * W;
    const Peak3Grid g = mi_peak3_grid(D, H, W);
    size_t n_cand = std::max(n, (size_t)g.n_seg * g.seg_cap);
    char* w = (char*)workspace;
    DecodeHeader* hdr = (DecodeHeader*)w;
    w += mi_align_up(sizeof(DecodeHeader), 256);
    uint2* cands = (uint2*)w;
    w += mi_align_up(n_cand * sizeof(uint2), 256);
    uint2* sel = (uint2*)w;
    w += mi_align_up((size_t)MI_SEL_CAP * sizeof(uint2), 256);
    unsigned* seg_count = (unsigned*)w;
    w += mi_align_up((size_t)g.n_seg * sizeof(unsigned), 256);
    void* extra1 = (void*)w;                                // table + bounds of the one-launch decode
    // bit 1 of `apply_sigmoid`: the caller vouches that the workspace header is clean - zeroed once by
    // mi_decode_workspace_init and, since then, only used by calls that passed this bit (they leave it clean again)
    const int self_clean = (apply_sigmoid & 2) ? 1 : 0;
    apply_sigmoid &= 1;
    if (!self_clean) MI_HIP(hipMemsetAsync(hdr, 0, sizeof(DecodeHeader), s));
    // 128 KiB of dynamic LDS for the single-workgroup sorter (above the 64 KiB default limit)
    static bool attr_set = false;
    if (!attr_set) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_final_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(MI_SEL_CAP * sizeof(unsigned long long))));
        attr_set = true;
    }
    float* val_out = apply_sigmoid ? heat_out : nullptr;

    if (!fiber && k == 3 && mi_decode1_usable(logits, val_out, D, H, W, K))
        // ONE launch: march, per-workgroup best lists, selection by the last workgroup to finish (infer_decode1.hip)
        return mi_launch_decode1(logits, val_out, D, H, W, apply_sigmoid != 0, K, dets, (int*)n_valid_out, hdr, cands, seg_count,
                                 extra1, s);

    if (!fiber && k == 3 && mi_peak3_usable(logits, val_out, nullptr, D, H, W)) {
        // register march: per-wave candidate segments, no global candidate counter
        Peak3Params q = {};
        q.in = logits; q.val_out = val_out; q.D = D; q.H = H; q.W = W;
        q.cands = cands; q.seg_count = seg_count; q.hist = hdr->hist;
        int rc = mi_launch_peak3(q, g, apply_sigmoid != 0, s);
        if (rc) return rc;
        const int fblocks = (int)std::min<unsigned>((g.n_seg + 7) / 8, 256u);       // two segments per wave
        hipLaunchKernelGGL(topk_filter_seg_kernel, dim3(fblocks), dim3(256), 0, s, cands, seg_count, g.n_seg, g.seg_cap,
                           hdr, sel, (unsigned)MI_SEL_CAP, K);
        MI_RETURN_IF_LAUNCH_FAILED();
        hipLaunchKernelGGL(topk_final_kernel, dim3(MI_FINAL_BLOCKS), dim3(1024), MI_SEL_CAP * sizeof(unsigned long long), s,
                           cands, seg_count, g.n_seg, g.seg_cap, hdr, sel, K, H, W, dets, (int*)n_valid_out, self_clean);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }

    MarchParams p = {};
    p.in = logits; p.val_out = val_out;
    p.mode = apply_sigmoid ? MI_LOAD_SIGMOID : MI_LOAD_PLAIN;
    p.D = D; p.H = H; p.W = W; p.fiber = fiber ? 1 : 0;
    p.cands = cands; p.cand_count = &hdr->cand_count; p.cand_cap = (unsigned)n; p.hist = hdr->hist;
    int rc = mi_launch_march(p, fiber ? k : 3, k, s);
    if (rc) return rc;
    int fblocks = (int)std::min<size_t>((n / 64 + 255) / 256 + 1, 512);
    hipLaunchKernelGGL(topk_filter_kernel, dim3(fblocks), dim3(256), 0, s, cands, hdr, (unsigned)n,
                       sel, (unsigned)MI_SEL_CAP, K);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(topk_final_kernel, dim3(MI_FINAL_BLOCKS), dim3(1024), MI_SEL_CAP * sizeof(unsigned long long),
                       s, cands, &hdr->cand_count, 1u, (unsigned)n, hdr, sel, K, H, W, dets, (int*)n_valid_out, self_clean);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_decode_workspace_init(void* workspace, size_t workspace_bytes, mi_stream_t stream) {
    if (!workspace || workspace_bytes < sizeof(DecodeHeader)) return MI_E_ARG;
    hipLaunchKernelGGL(zero_header_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (DecodeHeader*)workspace);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
