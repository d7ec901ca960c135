// Tomogram loading on the device: the arithmetic of utils/loader.py `load_rec`
// (:27-88: axis reorder, optional z-pair max, z-score) and `preprocess` / `quantize` (:16-25, :90-121:
// z-score -> 8-bit quantisation on [mi, ma] -> min-max to [0, 1]).  All of it is HBM-bound streaming:
//   reorder      read N x {1,2,4} B, write 4 B per output voxel (LDS-tiled transpose when the file's
//                fastest axis is not the output's)
//   stats        read 4 B / voxel -> shifted fp64 sum, sum of squares, min, max (per slice or per volume)
//   z-score      read 4 B, write 4 B
//   preprocess   read 4 B, write 4 B: the min / max of the quantised volume follow from the min / max of
//                the input because the quantiser is monotone, so no third pass is needed
// The reference computes in float64 (np.zeros default dtype); here every per-voxel value is evaluated in
// fp64 registers from the exactly-representable fp32 / integer input and stored as fp32.
#include "common.h"
#include "../../include/cetpick_hip.h"

namespace {

template <class T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// out[j][a][b] = max_{t < pair} in[(pair*j + t)*sj + a*sa + b*sb]   (pair = 2 with `compress`)
// `fast` = which output axis is contiguous in the input: 2 (b: rows are copied) or 0 (j: transpose
// of the (b, j) plane through a 32 x 33 LDS tile)
struct ReorderParams {
    const void* src;
    float* dst;
    long sj, sa, sb;
    int Z, A, B;          // output extents (Z' , X, Y)
    int zin;              // input extent along j
    int pair;
};

template <class T>
__global__ __launch_bounds__(256) void reorder_rows_kernel(ReorderParams p) {
    const T* src = (const T*)p.src;
    const long total = (long)p.Z * p.A * p.B;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i % p.B);
        const long t = i / p.B;
        const int a = (int)(t % p.A);
        const int j = (int)(t / p.A);
        const long base = (long)a * p.sa + (long)b * p.sb;
        float v = to_f32(src[base + (long)(p.pair * j) * p.sj]);
        if (p.pair == 2 && 2 * j + 1 < p.zin) v = fmaxf(v, to_f32(src[base + (long)(2 * j + 1) * p.sj]));
        p.dst[i] = v;
    }
}

// grid: (ceil(B/32), ceil(Z/32), A); input contiguous along j (sj == 1)
template <class T>
__global__ __launch_bounds__(256) void reorder_transpose_kernel(ReorderParams p) {
    __shared__ float tile[32][33];
    const T* src = (const T*)p.src;
    const int a = blockIdx.z;
    const int b0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {                           // rows = b, cols = j (contiguous in)
        const int b = b0 + r, j = j0 + tx;
        float v = 0.f;
        if (b < p.B && j < p.Z) {
            const long base = (long)a * p.sa + (long)b * p.sb;
            v = to_f32(src[base + (long)(p.pair * j)]);
            if (p.pair == 2 && 2 * j + 1 < p.zin) v = fmaxf(v, to_f32(src[base + (long)(2 * j + 1)]));
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                           // rows = j, cols = b (contiguous out)
        const int j = j0 + r, b = b0 + tx;
        if (j < p.Z && b < p.B) p.dst[((long)j * p.A + a) * p.B + b] = tile[tx][r];
    }
}

// ---- statistics: per slice {sum (x-K), sum (x-K)^2, min, max, K} with K = first element of the slice ------
constexpr int STAT_CHUNK = 1 << 16;      // elements per workgroup

__global__ __launch_bounds__(256) void stats_partial_kernel(const float* x, long slice_elems, int chunks,
                                                            double* partials) {
    const long s = blockIdx.y;
    const float* xs = x + s * slice_elems;
    const double K = (double)xs[0];
    const long lo = (long)blockIdx.x * STAT_CHUNK;
    const long hi = lo + STAT_CHUNK < slice_elems ? lo + STAT_CHUNK : slice_elems;
    double s1 = 0, s2 = 0;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = xs[i];
        const double d = (double)v - K;
        s1 += d; s2 = fma(d, d, s2);
        mn = fminf(mn, v); mx = fmaxf(mx, v);
        if (v != v) { mn = v; mx = v; }                           // NaN poisons min / max like numpy
    }
    __shared__ double r1[256], r2[256];
    __shared__ float rmn[256], rmx[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2; rmn[threadIdx.x] = mn; rmx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o];
            const float a = rmn[threadIdx.x], b = rmn[threadIdx.x + o], c = rmx[threadIdx.x], d = rmx[threadIdx.x + o];
            rmn[threadIdx.x] = (a != a || b != b) ? NAN : fminf(a, b);
            rmx[threadIdx.x] = (c != c || d != d) ? NAN : fmaxf(c, d);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* dst = partials + (s * chunks + blockIdx.x) * 4;
        dst[0] = r1[0]; dst[1] = r2[0]; dst[2] = (double)rmn[0]; dst[3] = (double)rmx[0];
    }
}

// stats[s] = {mean, std (population, numpy default), min, max}
__global__ __launch_bounds__(256) void stats_final_kernel(const float* x, long slice_elems, int chunks,
                                                          const double* partials, double* stats) {
    const long s = blockIdx.x;
    double s1 = 0, s2 = 0, mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    for (int c = threadIdx.x; c < chunks; c += 256) {
        const double* src = partials + (s * chunks + c) * 4;
        s1 += src[0]; s2 += src[1];
        if (src[2] != src[2]) nan = true;
        mn = fmin(mn, src[2]); mx = fmax(mx, src[3]);
    }
    __shared__ double r1[256], r2[256], rmn[256], rmx[256];
    __shared__ int rnan[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2; rmn[threadIdx.x] = mn; rmx[threadIdx.x] = mx; rnan[threadIdx.x] = nan;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o];
            rmn[threadIdx.x] = fmin(rmn[threadIdx.x], rmn[threadIdx.x + o]);
            rmx[threadIdx.x] = fmax(rmx[threadIdx.x], rmx[threadIdx.x + o]);
            rnan[threadIdx.x] |= rnan[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)slice_elems, K = (double)x[s * slice_elems];
        const double m = r1[0] / n;
        double var = r2[0] / n - m * m;
        if (var < 0) var = 0;
        stats[4 * s + 0] = K + m;
        stats[4 * s + 1] = sqrt(var);
        stats[4 * s + 2] = rnan[0] ? (double)NAN : rmn[0];
        stats[4 * s + 3] = rnan[0] ? (double)NAN : rmx[0];
    }
}

__global__ __launch_bounds__(256) void zscore_kernel(const float* x, float* y, long slice_elems,
                                                     const double* stats) {
    const long s = blockIdx.y;
    const double mean = stats[4 * s], inv = 1.0 / stats[4 * s + 1];
    const float* xs = x + s * slice_elems;
    float* ys = y + s * slice_elems;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < slice_elems; i += (long)gridDim.x * 256)
        ys[i] = (float)(((double)xs[i] - mean) * inv);
}

// loader.py:16-25 with the z-score in front: q = rint(clip(255 * (z - mi) / (ma - mi), 0, 255))
__device__ __forceinline__ double quant8(double v, double mean, double std, double mi, double r) {
    const double z = (v - mean) / std;
    double t = 255.0 * (z - mi) / r;
    t = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);       // np.clip (a NaN stays NaN)
    return rint(t);                                     // np.round: half to even
}

// y = (q - qmin) / (qmax - qmin), qmin / qmax from the slice min / max (quant8 is monotone in v).
// flat_zero: a constant slice gives 0 (cv2.normalize NORM_MINMAX, the tilt branch) instead of 0/0 = NaN
// (the uint8 expression of the volume branch, loader.py:106,120)
__global__ __launch_bounds__(256) void preprocess_kernel(const float* x, float* y, long slice_elems,
                                                         const double* stats, double mi, double ma,
                                                         int flat_zero) {
    const long s = blockIdx.y;
    const double mean = stats[4 * s], std = stats[4 * s + 1], r = ma - mi;
    const double qmin = quant8(stats[4 * s + 2], mean, std, mi, r);
    const double qmax = quant8(stats[4 * s + 3], mean, std, mi, r);
    const double span = qmax - qmin;
    const float* xs = x + s * slice_elems;
    float* ys = y + s * slice_elems;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < slice_elems; i += (long)gridDim.x * 256) {
        const double q = quant8((double)xs[i], mean, std, mi, r);
        double o = (q - qmin) / span;
        if (flat_zero && !(span > 0.0)) o = 0.0;
        ys[i] = (float)o;
    }
}

int stat_chunks(long slice_elems) { return (int)((slice_elems + STAT_CHUNK - 1) / STAT_CHUNK); }
int stream_blocks(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 4096)); }

template <class T>
int launch_reorder(const ReorderParams& p, bool transpose, hipStream_t s) {
    if (transpose) {
        dim3 grid((p.B + 31) / 32, (p.Z + 31) / 32, p.A);
        hipLaunchKernelGGL((reorder_transpose_kernel<T>), grid, dim3(256), 0, s, p);
    } else {
        const long total = (long)p.Z * p.A * p.B;
        hipLaunchKernelGGL((reorder_rows_kernel<T>), dim3(stream_blocks(total)), dim3(256), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

extern "C" int mi_rec_reorder(const void* src, int mrc_mode, int d0, int d1, int d2, int order, int compress,
                              float* dst, mi_stream_t stream) {
    if (!src || !dst || d0 <= 0 || d1 <= 0 || d2 <= 0 || order < 0 || order > 3) return MI_E_ARG;
    ReorderParams p = {};
    p.src = src; p.dst = dst; p.pair = compress ? 2 : 1;
    const long D1 = d1, D2 = d2;
    int zin;
    switch (order) {                       // out[j][a][b] = file[...]  (loader.py:32-36, :62)
        case MI_ORDER_XYZ: p.sj = 1;       p.sa = D1 * D2; p.sb = D2;      zin = d2; p.A = d0; p.B = d1; break;
        case MI_ORDER_XZY: p.sj = D2;      p.sa = D1 * D2; p.sb = 1;       zin = d1; p.A = d0; p.B = d2; break;
        case MI_ORDER_YXZ: p.sj = 1;       p.sa = D2;      p.sb = D1 * D2; zin = d2; p.A = d1; p.B = d0; break;
        default:           p.sj = D1 * D2; p.sa = D2;      p.sb = 1;       zin = d0; p.A = d1; p.B = d2; break;
    }
    // zxy + compress allocates z//2 slices and walks range(0, z, 2): an odd z overruns (IndexError in the
    // reference, loader.py:64-75) -> argument error here
    if (order == MI_ORDER_ZXY && compress && (zin & 1)) return MI_E_ARG;
    p.zin = zin;
    p.Z = compress ? (zin + 1) / 2 : zin;
    if (p.A > 65535) return MI_E_UNSUPPORTED;
    const bool transpose = (p.sj == 1);
    hipStream_t s = (hipStream_t)stream;
    switch (mrc_mode) {
        case 0: return launch_reorder<int8_t>(p, transpose, s);
        case 1: return launch_reorder<int16_t>(p, transpose, s);
        case 2: return launch_reorder<float>(p, transpose, s);
        case 6: return launch_reorder<uint16_t>(p, transpose, s);
        default: return MI_E_UNSUPPORTED;
    }
}

extern "C" size_t mi_vol_stats_workspace_bytes(long n_slices, long slice_elems) {
    if (n_slices <= 0 || slice_elems <= 0) return 0;
    return sizeof(double) * 4 * (size_t)n_slices * stat_chunks(slice_elems);
}

extern "C" int mi_vol_stats(const float* x, long n_slices, long slice_elems, double* stats, void* ws,
                            size_t ws_bytes, mi_stream_t stream) {
    if (!x || !stats || n_slices <= 0 || slice_elems <= 0 || n_slices > 65535) return MI_E_ARG;
    if (!ws || ws_bytes < mi_vol_stats_workspace_bytes(n_slices, slice_elems)) return MI_E_WORKSPACE;
    const int chunks = stat_chunks(slice_elems);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(chunks, (unsigned)n_slices), dim3(256), 0, s, x, slice_elems,
                       chunks, (double*)ws);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(stats_final_kernel, dim3((unsigned)n_slices), dim3(256), 0, s, x, slice_elems, chunks,
                       (const double*)ws, stats);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_zscore(const float* x, float* y, long n_slices, long slice_elems, const double* stats,
                         mi_stream_t stream) {
    if (!x || !y || !stats || n_slices <= 0 || slice_elems <= 0 || n_slices > 65535) return MI_E_ARG;
    const int bx = (int)std::max<long>(1, std::min<long>((slice_elems + 255) / 256, n_slices > 1 ? 256 : 4096));
    hipLaunchKernelGGL(zscore_kernel, dim3(bx, (unsigned)n_slices), dim3(256), 0, (hipStream_t)stream, x, y,
                       slice_elems, stats);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_zscore_quantize_minmax(const float* x, float* y, long n_slices, long slice_elems,
                                         const double* stats, double mi, double ma, int flat_zero,
                                         mi_stream_t stream) {
    if (!x || !y || !stats || n_slices <= 0 || slice_elems <= 0 || n_slices > 65535 || !(ma > mi)) return MI_E_ARG;
    const int bx = (int)std::max<long>(1, std::min<long>((slice_elems + 255) / 256, n_slices > 1 ? 256 : 4096));
    hipLaunchKernelGGL(preprocess_kernel, dim3(bx, (unsigned)n_slices), dim3(256), 0, (hipStream_t)stream, x, y,
                       slice_elems, stats, mi, ma, flat_zero);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
