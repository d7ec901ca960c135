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

__device__ __forceinline__ int reflect_idx(int i, int n) {
    // scipy 'reflect': d c b a | a b c d | d c b a
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

}  // namespace

int mi_gauss_radius(float sigma) { return (int)(4.0f * sigma + 0.5f); }

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
        // z: in -> out, y: out -> tmp, x: tmp -> out   (one read + one write of the volume per pass)
        if ((rc = mi_launch_gauss_axis(in, out, D, H, W, 0, sigma, s))) return rc;
        if ((rc = mi_launch_gauss_axis(out, tmp, D, H, W, 1, sigma, s))) return rc;
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
