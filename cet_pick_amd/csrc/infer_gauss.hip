// Separable 3-D Gaussian (scipy.ndimage.gaussian_filter semantics: mode='reflect', truncate=4).
// Replaces the scipy calls at utils/image.py:152-156 / :186-187 and utils/loader.py:102 (reference).
//
// Three passes (z, y, x - scipy's axis order), each one read + one write of the volume.  A thread
// produces 4 consecutive outputs along the filtered axis from a sliding register window, so each
// tap costs one LDS read per 4 FMAs.  The x pass keeps its LDS row 4-way interleaved
// (element e -> (e&3)*pitch + e/4) so lane-consecutive threads hit consecutive banks.
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int GT = 256;
inline int mi_gauss_radius_host(float sigma) { return (int)(4.0f * sigma + 0.5f); }

__device__ __forceinline__ int reflect_idx(int i, int n) {
    // scipy 'reflect': d c b a | a b c d | d c b a
    // The common cases - inside, or one reflection away - without the modulo: an integer division per fetched element
    // was most of the instruction stream of these kernels (the scalar unit, shared by the CU's four SIMDs, was the
    // bottleneck of the marching filter).
    if ((unsigned)i < (unsigned)n) return i;
    if (i < 0 && i >= -n) return -i - 1;
    if (i >= n && i < 2 * n) return 2 * n - 1 - i;
    int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return (i >= n) ? (period - 1 - i) : i;
}

// weights w[0..nt4) in LDS: w[t] = exp(-0.5 (t-r)^2 / sigma^2) / sum for t <= 2r, else 0
__device__ __forceinline__ void build_weights(float* w, double* scratch, int r, int nt4, float sigma,
                                              int tid) {
    const int ntaps = 2 * r + 1;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int t = tid; t < nt4; t += GT) {
        double d = (double)(t - r);
        scratch[t] = (t < ntaps) ? exp(c * d * d) : 0.0;
    }
    __syncthreads();
    __shared__ double s_sum;
    if (tid == 0) {
        double s = 0;
        for (int t = 0; t < ntaps; ++t) s += scratch[t];
        s_sum = s;
    }
    __syncthreads();
    for (int t = tid; t < nt4; t += GT) w[t] = (float)(scratch[t] / s_sum);
    __syncthreads();
}

#define MI_TAP4(ACC, W, V0, V1, V2, V3) \
    ACC[0] = fmaf(W, V0, ACC[0]); ACC[1] = fmaf(W, V1, ACC[1]); \
    ACC[2] = fmaf(W, V2, ACC[2]); ACC[3] = fmaf(W, V3, ACC[3]);

// ---- filter along a strided axis (z or y); x stays the contiguous lane axis -------------------
// tile: rows = L + nt4 (conv axis), 64 columns.  thread (tx = tid&63, tg = tid>>6) makes OUTS
// outputs at conv positions c0 + tg*OUTS + [0, OUTS).
template <int OUTS>
__global__ __launch_bounds__(GT) void gauss_strided_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, int n_conv,
                                                          long conv_stride, long other_stride,
                                                          int W, int r, int nt4, float sigma) {
    constexpr int L = 4 * OUTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* scratch = reinterpret_cast<double*>(smem);
    float* w = reinterpret_cast<float*>(smem + sizeof(double) * nt4);
    float* tile = w + nt4;                      // [(L + nt4)][64]
    const int tid = threadIdx.x, tx = tid & 63, tg = tid >> 6;
    const int x = blockIdx.x * 64 + tx;
    const int c0 = blockIdx.y * L;
    const long obase = (long)blockIdx.z * other_stride;
    const int rows = L + nt4;
    const int rows_valid = L + 2 * r;
    for (int row = tg; row < rows; row += 4) {
        float v = 0.f;
        if (row < rows_valid && x < W) {
            int ci = reflect_idx(c0 - r + row, n_conv);
            v = in[obase + (long)ci * conv_stride + x];
        }
        tile[row * 64 + tx] = v;
    }
    build_weights(w, scratch, r, nt4, sigma, tid);   // ends with a barrier
#pragma unroll
    for (int g = 0; g < OUTS / 4; ++g) {
        const int base = tg * OUTS + 4 * g;
        const float* tp = tile + base * 64 + tx;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float v0 = tp[0], v1 = tp[64], v2 = tp[128], v3;
        for (int t = 0; t < nt4; t += 4) {
            float4 wv = *reinterpret_cast<const float4*>(w + t);
            const float* q = tp + (t + 3) * 64;
            v3 = q[0];   MI_TAP4(acc, wv.x, v0, v1, v2, v3)
            v0 = q[64];  MI_TAP4(acc, wv.y, v1, v2, v3, v0)
            v1 = q[128]; MI_TAP4(acc, wv.z, v2, v3, v0, v1)
            v2 = q[192]; MI_TAP4(acc, wv.w, v3, v0, v1, v2)
        }
        if (x < W) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int c = c0 + base + i;
                if (c < n_conv) out[obase + (long)c * conv_stride + x] = acc[i];
            }
        }
    }
}

// ---- filter along x (contiguous) ---------------------------------------------------------------
// block = 4 rows x 64 threads, each thread 4 consecutive x  (LX = 256 outputs per row).
__global__ __launch_bounds__(GT) void gauss_x_kernel(const float* __restrict__ in,
                                                    float* __restrict__ out, long n_rows, int W,
                                                    int r, int nt4, int pitch, float sigma) {
    constexpr int LX = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* scratch = reinterpret_cast<double*>(smem);
    float* w = reinterpret_cast<float*>(smem + sizeof(double) * nt4);
    float* tiles = w + nt4;                     // [4][4*pitch]
    const int tid = threadIdx.x, ti = tid & 63, tr = tid >> 6;
    const long row = (long)blockIdx.y * 4 + tr;
    const int xs0 = blockIdx.x * LX;
    float* tile = tiles + tr * 4 * pitch;
    const int n_el = LX + nt4 + 4;              // elements e in [0, n_el): x = xs0 - r + e
    const int n_valid = LX + 2 * r;
    if (row < n_rows) {
        const float* src = in + row * (long)W;
        for (int e = ti; e < n_el; e += 64) {
            float v = 0.f;
            if (e < n_valid) v = src[reflect_idx(xs0 - r + e, W)];
            tile[(e & 3) * pitch + (e >> 2)] = v;
        }
    }
    build_weights(w, scratch, r, nt4, sigma, tid);
    if (row >= n_rows) return;
    const float* s0 = tile, *s1 = tile + pitch, *s2 = tile + 2 * pitch, *s3 = tile + 3 * pitch;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float v0 = s0[ti], v1 = s1[ti], v2 = s2[ti], v3;
    for (int t = 0; t < nt4; t += 4) {
        float4 wv = *reinterpret_cast<const float4*>(w + t);
        const int q = ti + (t >> 2);
        v3 = s3[q];     MI_TAP4(acc, wv.x, v0, v1, v2, v3)
        v0 = s0[q + 1]; MI_TAP4(acc, wv.y, v1, v2, v3, v0)
        v1 = s1[q + 1]; MI_TAP4(acc, wv.z, v2, v3, v0, v1)
        v2 = s2[q + 1]; MI_TAP4(acc, wv.w, v3, v0, v1, v2)
    }
    const int x = xs0 + 4 * ti;
    float* dst = out + row * (long)W + x;
    if (x + 3 < W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x + i < W) dst[i] = acc[i];
    }
}

// ---- marching filter along a strided axis (z or y) --------------------------------------------------------
// A thread owns one column (fixed position on the other two axes) and walks along the filtered axis: every input
// element is read from memory exactly once - no halo re-reads (the tiled kernel above re-reads (64 + 2r)/64).
// The window of a column lives in REGISTERS: the ring of the last 2R+4+.. rows is a statically indexed register array
// (the step loop is unrolled over one full turn of the ring), a row's global load targets its ring register directly
// and is issued PD steps before the first FMA that reads it, so the prefetch costs no extra registers and there is no
// LDS traffic at all.  An LDS-ring version of the same walk (dynamic indexing) was measured at 2 TB/s (instruction-issue
// bound).  The taps are symmetric and sit in scalar registers as w[|d|]: R + 1 of them per sigma, so ONE read of the
// input can feed TWO sigmas (13 + 21 taps for sigma 3 and 5; the 82 two-sided taps of that pair did not fit the scalar
// file and the picker used two single-sigma jobs, reading the tomogram twice).  Three launch shapes:
//   single  in -> out (sigma);            dual  in -> out_a (small sigma), out_b (large sigma), one read;
//   two     blockIdx.y picks (in0 -> out0, sigma0) or (in1 -> out1, sigma1), each with its own radius.
template <int R>
struct SymTaps1 { float w[R + 1]; };         // w[t] = tap at distance t from the centre

struct MarchGeom {
    int n_conv;            // extent of the filtered axis
    long conv_stride;      // elements between consecutive positions on it
    long other_stride;     // column c -> base = (outer0 + c / w_inner) * other_stride + x0 + c % w_inner
    int w_inner;           // columns per outer index (a box of the other two axes: the picker skips its zeroed border)
    int outer0, x0;
    long n_cols;
    int out_lo, out_hi;    // outputs [out_lo, out_hi) of the filtered axis are produced (the rest is never read)
    // up to two word ranges that the launch zeroes on the side (the picker's header and candidate bitmap: the chain then
    // needs no clearing pass of its own - two fill launches fewer per pick); spread over all threads of the launch
    unsigned* clr[2];
    unsigned clr_n[2];
};

__device__ __forceinline__ void march_clear(const MarchGeom& p, long tid_linear, long n_threads) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (p.clr[k])
            for (long i = tid_linear; i < (long)p.clr_n[k]; i += n_threads) p.clr[k][i] = 0u;
}

// RB: radius of the ring (the larger sigma); RA: radius of the second output (DUAL), RA <= RB.
template <int RB, int RA, bool DUAL>
__device__ __forceinline__ void march_column(const float* __restrict__ in, float* __restrict__ out_b,
                                             float* __restrict__ out_a, const SymTaps1<RB>& wb, const SymTaps1<RA>& wa,
                                             const MarchGeom& p, long c) {
    constexpr int PD = 2;
    constexpr int LEAD = 2 * RB + 4 + 4 * (PD - 1);        // rows resident ahead of output i: q in [i, i + LEAD)
    constexpr int RW = LEAD + 4;                           // ring registers
    const long base = (p.outer0 + c / p.w_inner) * p.other_stride + p.x0 + (c % p.w_inner);
    const float* src = in + base;
    const int n = p.n_conv, n_out = p.out_hi;
    float win[RW];
#pragma unroll
    for (int q = 0; q < LEAD; ++q) win[q] = src[(long)reflect_idx(p.out_lo + q - RB, n) * p.conv_stride];
    for (int i0 = p.out_lo; i0 < n_out; i0 += RW) {
#pragma unroll
        for (int ph = 0; ph < RW / 4; ++ph) {
            const int i = i0 + 4 * ph;                     // slot of row q is (q - i0) mod RW: static per phase
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // row i + LEAD + u - RB is never below 0 here (LEAD >= RB): only the far end can reflect
                const int j = i + LEAD + u - RB;
                win[(4 * ph + LEAD + u) % RW] = src[(long)(j < n ? j : reflect_idx(j, n)) * p.conv_stride];
            }
            float accb[4], acca[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ctr = win[(4 * ph + RB + u) % RW];
                accb[u] = wb.w[0] * ctr;
                if (DUAL) acca[u] = wa.w[0] * ctr;
            }
#pragma unroll
            for (int t = 1; t <= RB; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float pair = win[(4 * ph + RB + u - t + RW) % RW] + win[(4 * ph + RB + u + t) % RW];
                    accb[u] = fmaf(wb.w[t], pair, accb[u]);
                    if (DUAL && t <= RA) acca[u] = fmaf(wa.w[t], pair, acca[u]);
                }
            if (i < n_out) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < n_out) {
                        out_b[base + (long)(i + u) * p.conv_stride] = accb[u];
                        if (DUAL) out_a[base + (long)(i + u) * p.conv_stride] = acca[u];
                    }
            }
        }
    }
}

template <int R>
__global__ __launch_bounds__(GT) void gauss_march_single_kernel(const float* in, float* out, MarchGeom p, SymTaps1<R> w) {
    const long c = (long)blockIdx.x * GT + threadIdx.x;
    if (c >= p.n_cols) return;
    march_column<R, R, false>(in, out, nullptr, w, w, p, c);
}
template <int RB, int RA>
__global__ __launch_bounds__(GT) void gauss_march_dual_kernel(const float* in, float* out_a, float* out_b, MarchGeom p,
                                                             SymTaps1<RA> wa, SymTaps1<RB> wb) {
    const long c = (long)blockIdx.x * GT + threadIdx.x;
    march_clear(p, c, (long)gridDim.x * GT);
    if (c >= p.n_cols) return;
    march_column<RB, RA, true>(in, out_b, out_a, wb, wa, p, c);
}
template <int R0, int R1>
__global__ __launch_bounds__(GT) void gauss_march_two_kernel(const float* in0, float* out0, const float* in1, float* out1,
                                                            MarchGeom p, SymTaps1<R0> w0, SymTaps1<R1> w1) {
    const long c = (long)blockIdx.x * GT + threadIdx.x;
    if (c >= p.n_cols) return;
    if (blockIdx.y == 0) march_column<R0, R0, false>(in0, out0, nullptr, w0, w0, p, c);      // (workgroup-uniform)
    else march_column<R1, R1, false>(in1, out1, nullptr, w1, w1, p, c);
}

template <int R>
struct RegMarchWeights { float w[2][2 * R + 1]; };

// ---- filter along x with the radius a compile-time constant ------------------------------------------------------
// Same tile as gauss_x_kernel (4 rows x 256 outputs, rows 4-way interleaved in LDS), but the taps come from the host as
// kernel arguments (scalar registers) instead of being rebuilt - a double-precision exp per tap, a serial sum and
// three barriers - by every workgroup for its 1024 outputs, and the window is a fully unrolled register array: 2R+4 LDS
// reads at (lane base + immediate) and 4 (2R+1) FMAs per thread.
template <int R>
__global__ __launch_bounds__(GT) void gauss_x_fixed_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long n_rows, int W, int groups_per_block, RegMarchWeights<R> wt) {
    constexpr int LX = 256, NEL = LX + 2 * R;
    constexpr int Q = (NEL + 3) / 4;
    constexpr int PITCH = Q + ((8 - (Q & 31) + 32) & 31);      // PITCH % 32 == 8: conflict-free interleaved stores
    constexpr int NLD = (NEL + 63) / 64;                       // elements a thread stages per row group
    __shared__ float tiles[2][4][4 * PITCH];
    const int tid = threadIdx.x, ti = tid & 63, tr = tid >> 6;
    const int xs0 = blockIdx.x * LX;
    // a workgroup walks `groups_per_block` groups of 4 rows; the next group's elements are in flight (registers) while
    // the current one is filtered out of LDS - a workgroup that does one group only spends its life waiting for its loads
    const long g0 = (long)blockIdx.y * groups_per_block;
    int src_idx[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) src_idx[k] = reflect_idx(xs0 - R + ti + 64 * k, W);
    float pre[NLD];
    auto fetch = [&](long g) {
        const long row = 4 * g + tr;
        const bool ok = row < n_rows;
        const float* src = in + (ok ? row : 0) * (long)W;
#pragma unroll
        for (int k = 0; k < NLD; ++k) pre[k] = (ok && ti + 64 * k < NEL) ? src[src_idx[k]] : 0.f;
    };
    auto stage = [&](int buf) {
        float* tile = tiles[buf][tr];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = ti + 64 * k;
            if (e < NEL) tile[(e & 3) * PITCH + (e >> 2)] = pre[k];
        }
    };
    fetch(g0);
    stage(0);
    __syncthreads();
    for (int it = 0; it < groups_per_block; ++it) {
        const long g = g0 + it;
        if (4 * g >= n_rows) break;                            // (block-uniform)
        const bool more = it + 1 < groups_per_block && 4 * (g + 1) < n_rows;
        if (more) fetch(g + 1);
        const long row = 4 * g + tr;
        const float* tile = tiles[it & 1][tr];
        float val[2 * R + 4];
#pragma unroll
        for (int e = 0; e < 2 * R + 4; ++e) val[e] = tile[(e & 3) * PITCH + ti + (e >> 2)];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2 * R + 1; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fmaf(wt.w[0][t], val[t + u], acc[u]);
        if (row < n_rows) {
            const int x = xs0 + 4 * ti;
            float* dst = out + row * (long)W + x;
            if (x + 3 < W && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (x + i < W) dst[i] = acc[i];
            }
        }
        if (more) stage((it + 1) & 1);                         // the other buffer: last read in iteration it-1
        __syncthreads();
    }
}

// scipy's normalised kernel for `sigma`, centred in a (2R+1)-tap window (zeros outside its own radius)
template <int R>
void fill_weights(float sigma, float* w) {
    const int rs = mi_gauss_radius_host(sigma);
    double tmp[2 * R + 1], sum = 0;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int t = 0; t < 2 * R + 1; ++t) {
        const int d = t - R;
        tmp[t] = (d >= -rs && d <= rs) ? exp(c * (double)d * (double)d) : 0.0;
        sum += tmp[t];
    }
    for (int t = 0; t < 2 * R + 1; ++t) w[t] = (float)(tmp[t] / sum);
}

// scipy's normalised kernel for `sigma` as symmetric taps w[|d|], zeros beyond its own radius
template <int R>
void fill_sym1(float sigma, SymTaps1<R>& t) {
    const int rs = mi_gauss_radius_host(sigma);
    double tmp[R + 1], sum = 0;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int d = 0; d <= R; ++d) {
        tmp[d] = d <= rs ? exp(c * (double)d * (double)d) : 0.0;
        sum += d == 0 ? tmp[d] : 2.0 * tmp[d];
    }
    for (int d = 0; d <= R; ++d) t.w[d] = (float)(tmp[d] / sum);
}

inline int march_radius_class(int r) { return r <= 8 ? 8 : r <= 12 ? 12 : r <= 16 ? 16 : 20; }

template <int R>
int launch_march_single(const float* in, float* out, const MarchGeom& g, float sigma, hipStream_t s) {
    SymTaps1<R> w;
    fill_sym1<R>(sigma, w);
    hipLaunchKernelGGL((gauss_march_single_kernel<R>), dim3((unsigned)((g.n_cols + GT - 1) / GT)), dim3(GT), 0, s, in, out, g, w);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
template <int RB, int RA>
int launch_march_dual(const float* in, float* out_a, float* out_b, const MarchGeom& g, float sa, float sb, hipStream_t s) {
    SymTaps1<RA> wa;
    SymTaps1<RB> wb;
    fill_sym1<RA>(sa, wa);
    fill_sym1<RB>(sb, wb);
    hipLaunchKernelGGL((gauss_march_dual_kernel<RB, RA>), dim3((unsigned)((g.n_cols + GT - 1) / GT)), dim3(GT), 0, s, in,
                       out_a, out_b, g, wa, wb);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
template <int R0, int R1>
int launch_march_two(const float* in0, float* out0, float s0, const float* in1, float* out1, float s1, const MarchGeom& g,
                     hipStream_t s) {
    SymTaps1<R0> w0;
    SymTaps1<R1> w1;
    fill_sym1<R0>(s0, w0);
    fill_sym1<R1>(s1, w1);
    hipLaunchKernelGGL((gauss_march_two_kernel<R0, R1>), dim3((unsigned)((g.n_cols + GT - 1) / GT), 2), dim3(GT), 0, s, in0,
                       out0, in1, out1, g, w0, w1);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

template <int R>
int launch_x_fixed(const float* in, float* out, long n_rows, int W, float sigma, hipStream_t s) {
    RegMarchWeights<R> w0 = {};
    fill_weights<R>(sigma, w0.w[0]);
    // enough workgroups for ~8 per CU, each walking several 4-row groups
    const long groups = mi_cdiv(n_rows, 4);
    const long xb = mi_cdiv(W, 256);
    long gpb = groups * xb / 2048;
    gpb = gpb < 1 ? 1 : (gpb > 64 ? 64 : gpb);
    dim3 grid((unsigned)xb, (unsigned)mi_cdiv(groups, gpb));
    hipLaunchKernelGGL((gauss_x_fixed_kernel<R>), grid, dim3(GT), 0, s, in, out, n_rows, W, (int)gpb, w0);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

int mi_gauss_radius(float sigma) { return (int)(4.0f * sigma + 0.5f); }

// Marching variant: up to two jobs (in -> out with sigma), or one job producing two sigmas from one read
// (out0b != null: out0a <- sig0a, out0b <- sig0b, sig0a <= sig0b).
// Returns MI_E_UNSUPPORTED for radii it is not instantiated for (sigma > 5.1: callers fall back to the tiled kernels).
// box (may be null = everything): {z0, z1, y0, y1, x0, x1}, the part of the OUTPUT volume that will be read later.
int mi_launch_gauss_march(const float* in0, float* out0a, float* out0b, float sig0a, float sig0b, const float* in1,
                          float* out1, float sig1, int D, int H, int W, int axis, hipStream_t s, const int* box,
                          unsigned* clr0, unsigned clr0_n, unsigned* clr1, unsigned clr1_n) {
    if (axis != 0 && axis != 1) return MI_E_ARG;
    if ((clr0 || clr1) && !out0b) return MI_E_ARG;          // only the dual (one read, two sigmas) launch clears
    const bool dual = out0b != nullptr;
    float smax = sig0a;
    if (dual) smax = std::max(smax, sig0b);
    if (in1) smax = std::max(smax, sig1);
    if (mi_gauss_radius(smax) > 20 || getenv("MI_GAUSS_NO_REGMARCH")) return MI_E_UNSUPPORTED;
    MarchGeom g = {};
    g.clr[0] = clr0; g.clr_n[0] = clr0_n; g.clr[1] = clr1; g.clr_n[1] = clr1_n;
    const int full[6] = {0, D, 0, H, 0, W};
    const int* bx = box ? box : full;
    if (bx[0] < 0 || bx[1] > D || bx[2] < 0 || bx[3] > H || bx[4] < 0 || bx[5] > W || bx[0] >= bx[1] || bx[2] >= bx[3] ||
        bx[4] >= bx[5]) return MI_E_ARG;
    g.n_conv = axis == 0 ? D : H;
    g.conv_stride = axis == 0 ? (long)H * W : (long)W;
    g.w_inner = bx[5] - bx[4];
    g.x0 = bx[4];
    if (axis == 0) {             // columns = (y, x) of the box, outputs z in [z0, z1)
        g.other_stride = W; g.outer0 = bx[2]; g.n_cols = (long)(bx[3] - bx[2]) * g.w_inner;
        g.out_lo = bx[0]; g.out_hi = bx[1];
    } else {                     // columns = (z, x) of the box, outputs y in [y0, y1)
        g.other_stride = (long)H * W; g.outer0 = bx[0]; g.n_cols = (long)(bx[1] - bx[0]) * g.w_inner;
        g.out_lo = bx[2]; g.out_hi = bx[3];
    }
    if (g.n_conv < 1) return MI_E_ARG;
    const int ra = march_radius_class(mi_gauss_radius(sig0a));
#define MI_R4(FN, R, ...)                                                       \
    switch (R) {                                                                \
        case 8: return FN<8>(__VA_ARGS__);                                      \
        case 12: return FN<12>(__VA_ARGS__);                                    \
        case 16: return FN<16>(__VA_ARGS__);                                    \
        default: return FN<20>(__VA_ARGS__);                                    \
    }
    if (dual) {
        if (in1 || sig0a > sig0b) return MI_E_ARG;
        const int rb = march_radius_class(mi_gauss_radius(sig0b));
        // (RB, RA) pairs with RA <= RB
        if (rb == 20) { if (ra == 8) return launch_march_dual<20, 8>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        if (ra == 12) return launch_march_dual<20, 12>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        if (ra == 16) return launch_march_dual<20, 16>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        return launch_march_dual<20, 20>(in0, out0a, out0b, g, sig0a, sig0b, s); }
        if (rb == 16) { if (ra == 8) return launch_march_dual<16, 8>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        if (ra == 12) return launch_march_dual<16, 12>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        return launch_march_dual<16, 16>(in0, out0a, out0b, g, sig0a, sig0b, s); }
        if (rb == 12) { if (ra == 8) return launch_march_dual<12, 8>(in0, out0a, out0b, g, sig0a, sig0b, s);
                        return launch_march_dual<12, 12>(in0, out0a, out0b, g, sig0a, sig0b, s); }
        return launch_march_dual<8, 8>(in0, out0a, out0b, g, sig0a, sig0b, s);
    }
    if (!in1) { MI_R4(launch_march_single, ra, in0, out0a, g, sig0a, s) }
    const int r1 = march_radius_class(mi_gauss_radius(sig1));
    if (ra > r1)          // instantiated for R0 <= R1 only: the two jobs are independent, swap them
        return mi_launch_gauss_march(in1, out1, nullptr, sig1, 0.f, in0, out0a, sig0a, D, H, W, axis, s, box, nullptr, 0, nullptr, 0);
#define MI_TWO(R0)                                                                                   \
    switch (r1) {                                                                                    \
        case 8: if (R0 <= 8) return launch_march_two<(R0 <= 8 ? R0 : 8), 8>(in0, out0a, sig0a, in1, out1, sig1, g, s);      \
        case 12: if (R0 <= 12) return launch_march_two<(R0 <= 12 ? R0 : 12), 12>(in0, out0a, sig0a, in1, out1, sig1, g, s);  \
        case 16: if (R0 <= 16) return launch_march_two<(R0 <= 16 ? R0 : 16), 16>(in0, out0a, sig0a, in1, out1, sig1, g, s);  \
        default: return launch_march_two<R0, 20>(in0, out0a, sig0a, in1, out1, sig1, g, s);          \
    }
    switch (ra) {
        case 8: MI_TWO(8)
        case 12: MI_TWO(12)
        case 16: MI_TWO(16)
        default: MI_TWO(20)
    }
#undef MI_TWO
#undef MI_R4
}

// One axis of the separable filter.  axis: 0 = z, 1 = y, 2 = x.
int mi_launch_gauss_axis(const float* in, float* out, int D, int H, int W, int axis, float sigma,
                         hipStream_t s) {
    const int r = mi_gauss_radius(sigma);
    const int nt4 = ((2 * r + 1) + 3) & ~3;
    if (r > 64) return MI_E_UNSUPPORTED;   // sigma <= 16: LDS tile stays under 64 KiB
    if (axis == 2) {
        constexpr int LX = 256;
        int pitch = (LX + nt4 + 4 + 3) / 4;
        pitch += (8 - (pitch & 31) + 32) & 31;   // pitch % 32 == 8 -> conflict-free interleave
        size_t lds = sizeof(double) * nt4 + sizeof(float) * nt4 + sizeof(float) * 4 * 4 * pitch;
        long n_rows = (long)D * H;
        dim3 grid(mi_cdiv(W, LX), mi_cdiv(n_rows, 4));
        if (grid.y > 65535u * 32u) return MI_E_UNSUPPORTED;
        if (r <= 20 && !getenv("MI_GAUSS_NO_XFIXED")) {       // radius-specialised variant (taps from the host)
            if (r <= 8) return launch_x_fixed<8>(in, out, n_rows, W, sigma, s);
            if (r <= 12) return launch_x_fixed<12>(in, out, n_rows, W, sigma, s);
            if (r <= 16) return launch_x_fixed<16>(in, out, n_rows, W, sigma, s);
            return launch_x_fixed<20>(in, out, n_rows, W, sigma, s);
        }
        hipLaunchKernelGGL(gauss_x_kernel, grid, dim3(GT), lds, s, in, out, n_rows, W, r, nt4, pitch, sigma);
    } else {
        constexpr int OUTS = 16, L = 64;
        size_t lds = sizeof(double) * nt4 + sizeof(float) * nt4 + sizeof(float) * (L + nt4) * 64;
        int n_conv = axis == 0 ? D : H;
        long conv_stride = axis == 0 ? (long)H * W : (long)W;
        int n_other = axis == 0 ? H : D;
        long other_stride = axis == 0 ? (long)W : (long)H * W;
        dim3 grid(mi_cdiv(W, 64), mi_cdiv(n_conv, L), n_other);
        hipLaunchKernelGGL((gauss_strided_kernel<OUTS>), grid, dim3(GT), lds, s, in, out, n_conv,
                           conv_stride, other_stride, W, r, nt4, sigma);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_gauss3d_sep(const float* in, float* out, float* tmp, int D, int H, int W,
                              float sigma, mi_stream_t stream) {
    if (!in || !out || !tmp || D <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f)) return MI_E_ARG;
    if (tmp == in || tmp == out) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (out != in) {
        // z: in -> out, y: out -> tmp, x: tmp -> out   (one read + one write of the volume per pass; z and y
        // through the marching kernel when its radius fits)
        auto strided = [&](const float* a, float* b, int axis) -> int {
            int r2 = mi_launch_gauss_march(a, b, nullptr, sigma, 0.f, nullptr, nullptr, 0.f, D, H, W, axis, s, nullptr, nullptr, 0, nullptr, 0);
            return r2 == MI_E_UNSUPPORTED ? mi_launch_gauss_axis(a, b, D, H, W, axis, sigma, s) : r2;
        };
        if ((rc = strided(in, out, 0))) return rc;
        if ((rc = strided(out, tmp, 1))) return rc;
        if ((rc = mi_launch_gauss_axis(tmp, out, D, H, W, 2, sigma, s))) return rc;
    } else {
        // in place: no pass may write the buffer its neighbours still read, so one extra copy
        if ((rc = mi_launch_gauss_axis(in, tmp, D, H, W, 0, sigma, s))) return rc;
        if ((rc = mi_launch_gauss_axis(tmp, out, D, H, W, 1, sigma, s))) return rc;
        if ((rc = mi_launch_gauss_axis(out, tmp, D, H, W, 2, sigma, s))) return rc;
        MI_HIP(hipMemcpyAsync(out, tmp, sizeof(float) * (size_t)D * H * W, hipMemcpyDeviceToDevice, s));
    }
    return MI_OK;
}

extern "C" int mi_gauss2d_slices(const float* in, float* out, float* tmp, int D, int H, int W, float sigma,
                                 mi_stream_t stream) {
    if (!in || !out || !tmp || D <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f)) return MI_E_ARG;
    if (tmp == in || tmp == out) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = mi_launch_gauss_axis(in, tmp, D, H, W, 1, sigma, s))) return rc;     // y: in -> tmp
    if ((rc = mi_launch_gauss_axis(tmp, out, D, H, W, 2, sigma, s))) return rc;    // x: tmp -> out
    return MI_OK;
}
