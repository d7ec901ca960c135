// Sub-tomogram crop + normalise, one workgroup per crop.
//
// Replaces the per-pick Python loops of the reference datasets:
//   datasets/tomo_pre_proj_angle_select_new3d_vol.py:117-128 `extract_subvols`
//        v[z-sz//2 : z+sz//2+1, y-sy//2 : y+sy//2, x-sx//2 : x+sx//2] -> sum over z -> min-max
//   :130-138 `extract_subvols_3d` (plain crop)
//   and the z-normalised 3-D crop that feeds the MoCo-3D encoder;
//   datasets/tomo_pre.py:57-60 the deterministic tail of the 3-D chain on a cutup window:
//        Crop -> ZNormalization -> RescaleIntensity(-3, 3) -> ZNormalization      (mode 3; the Crop is the window itself)
//   simsiam_test_hm_3d.py:45-51  ToPILImage -> ToTensor -> Normalize(mean, std)   (mi_u8_roundtrip_normalize)
// HBM-bound gather: rows of the crop are contiguous along x, so lanes read consecutive floats.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

enum { CROP_RAW = 0, CROP_SUMZ_MINMAX = 1, CROP_ZNORM = 2, CROP_ZNORM_RESCALE_ZNORM = 3 };

// block-wide (sum, sum of squares) in fp64 and (min, max) of per-thread partials; every thread gets the result
__device__ __forceinline__ void block_stats(double s, double ss, float mn, float mx, double* out_s, double* out_ss,
                                            float* out_mn, float* out_mx) {
    __shared__ double rs[2][4];
    __shared__ float rm[2][4];
    s = wave_sum(s); ss = wave_sum(ss);
    mn = -wave_max(-mn); mx = wave_max(mx);
    __syncthreads();                                  // (the buffers may still be read from the previous round)
    if ((threadIdx.x & 63) == 0) {
        rs[0][threadIdx.x >> 6] = s; rs[1][threadIdx.x >> 6] = ss;
        rm[0][threadIdx.x >> 6] = mn; rm[1][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    *out_s = rs[0][0] + rs[0][1] + rs[0][2] + rs[0][3];
    *out_ss = rs[1][0] + rs[1][1] + rs[1][2] + rs[1][3];
    *out_mn = fminf(fminf(rm[0][0], rm[0][1]), fminf(rm[0][2], rm[0][3]));
    *out_mx = fmaxf(fmaxf(rm[1][0], rm[1][1]), fmaxf(rm[1][2], rm[1][3]));
}

// mode 3: the crop sits in LDS; three in-place passes, statistics of each pass in fp64
__global__ __launch_bounds__(256) void crop_chain_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                        const int* __restrict__ centres, int cz, int cy, int cx,
                                                        int flip_x, float* __restrict__ out) {
    extern __shared__ float buf[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int x0 = centres[3 * n + 0] - cx / 2, y0 = centres[3 * n + 1] - cy / 2, z0 = centres[3 * n + 2] - cz / 2;
    const long HW = (long)H * W;
    const int vox = cz * cy * cx, pix = cy * cx;
    double s = 0, ss = 0, S, SS;
    float mn = INFINITY, mx = -INFINITY, MN, MX;
    for (int i = tid; i < vox; i += 256) {
        const int z = i / pix, y = (i / cx) % cy, x = i % cx;
        const int sx = flip_x ? (cx - 1 - x) : x;
        const float v = vol[(long)clampi(z0 + z, 0, D - 1) * HW + (long)clampi(y0 + y, 0, H - 1) * W + clampi(x0 + sx, 0, W - 1)];
        buf[i] = v;
        s += v; ss += (double)v * v;
    }
    block_stats(s, ss, 0.f, 0.f, &S, &SS, &MN, &MX);
    // ZNormalization: (x - mean) / std, unbiased
    double mean = S / vox, var = (SS - vox * mean * mean) / (vox - 1);
    float m = (float)mean, inv = (float)(1.0 / sqrt(var > 0 ? var : 0.0));
    for (int i = tid; i < vox; i += 256) {
        const float v = (buf[i] - m) * inv;
        buf[i] = v;
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    block_stats(0, 0, mn, mx, &S, &SS, &MN, &MX);
    // RescaleIntensity(out_min_max = (-3, 3)): (x - min) / (max - min) * 6 - 3
    const float sc = 6.0f / (MX - MN);
    s = 0; ss = 0;
    for (int i = tid; i < vox; i += 256) {
        const float v = (buf[i] - MN) * sc - 3.0f;
        buf[i] = v;
        s += v; ss += (double)v * v;
    }
    block_stats(s, ss, 0.f, 0.f, &S, &SS, &MN, &MX);
    mean = S / vox; var = (SS - vox * mean * mean) / (vox - 1);
    m = (float)mean; inv = (float)(1.0 / sqrt(var > 0 ? var : 0.0));
    float* o = out + (long)n * vox;
    for (int i = tid; i < vox; i += 256) o[i] = (buf[i] - m) * inv;
}

__global__ void u8_roundtrip_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float mean,
                                              float inv_std_unused, float std) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // ToPILImage on a float tensor: mul(255).byte() (truncation, values of a min-max'ed crop are in [0, 1]);
        // ToTensor: / 255; Normalize: (q - mean) / std
        const float q = floorf(fminf(fmaxf(x[i] * 255.0f, 0.0f), 255.0f)) / 255.0f;
        y[i] = (q - mean) / std;
    }
}


// window start along an axis: the reference slices [c - s//2, c - s//2 + s) (z windows are odd:
// c - s//2 .. c + s//2, xy windows even: c - s/2 .. c + s/2 - 1)
__device__ __forceinline__ void crop_body(const float* __restrict__ vol, int D, int H, int W, int c_x, int c_y, int c_z,
                                          int cz, int cy, int cx, int mode, int flip_x, float* __restrict__ out) {
    __shared__ float red[2][4];
    extern __shared__ float plane[];               // SUMZ: cy*cx floats
    const int n = blockIdx.x, tid = threadIdx.x;
    const int x0 = c_x - cx / 2, y0 = c_y - cy / 2, z0 = c_z - cz / 2;
    const long HW = (long)H * W;
    const int vox = cz * cy * cx, pix = cy * cx;
    auto at = [&](int z, int y, int x) {
        int sx = flip_x ? (cx - 1 - x) : x;
        return vol[(long)clampi(z0 + z, 0, D - 1) * HW + (long)clampi(y0 + y, 0, H - 1) * W + clampi(x0 + sx, 0, W - 1)];
    };
    if (mode == CROP_RAW) {
        float* o = out + (long)n * vox;
        for (int i = tid; i < vox; i += 256) o[i] = at(i / pix, (i / cx) % cy, i % cx);
        return;
    }
    float a = 0.f, b = 0.f;                        // SUMZ: min, max ; ZNORM: sum, sumsq
    if (mode == CROP_SUMZ_MINMAX) {
        a = INFINITY; b = -INFINITY;
        for (int i = tid; i < pix; i += 256) {
            float s = 0.f;
            for (int z = 0; z < cz; ++z) s += at(z, i / cx, i % cx);
            plane[i] = s;
            a = fminf(a, s); b = fmaxf(b, s);
        }
        a = -wave_max(-a); b = wave_max(b);
    } else {
        for (int i = tid; i < vox; i += 256) { float v = at(i / pix, (i / cx) % cy, i % cx); a += v; b = fmaf(v, v, b); }
        a = wave_sum(a); b = wave_sum(b);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
    __syncthreads();
    if (mode == CROP_SUMZ_MINMAX) {
        const float mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        const float mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        const float inv = 1.0f / (mx - mn);                      // reference divides by (max - min) as is
        float* o = out + (long)n * pix;
        for (int i = tid; i < pix; i += 256) o[i] = (plane[i] - mn) * inv;
    } else {
        const double s = (double)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double ss = (double)red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = s / vox;
        double var = (ss - vox * mean * mean) / (vox - 1);        // unbiased, as torch.std
        if (var < 0) var = 0;
        const float m = (float)mean, inv = (float)(1.0 / sqrt(var));
        float* o = out + (long)n * vox;
        for (int i = tid; i < vox; i += 256) o[i] = (at(i / pix, (i / cx) % cy, i % cx) - m) * inv;
    }
}

// z-normalised (or raw) crops whose rows fit a wave and whose voxels fit the registers of 1024 threads (32^3, 6 x 48 x 48,
// 6 x 64 x 64 ...): ONE read of the crop.  A wave takes 64 / cxp rows of the crop at a time (cxp = cx rounded up to a power
// of two: lanes run along x, a 32-wide row pair is one 256-byte access), a thread keeps its <= 32 values, the statistics go
// through one workgroup reduction, the normalised values leave as coalesced rows.  The generic kernel reads the crop twice
// with three integer divisions per element on 256 threads: 90 us for the 64 crops of a MoCo batch - a tenth of the training
// step it feeds (tools/ab/entry_host.py); this one takes a few microseconds.
constexpr int CF_T = 1024, CF_MAXIT = 32;
__device__ __forceinline__ void crop_fast_body(const float* __restrict__ vol, int D, int H, int W, int c_x, int c_y, int c_z,
                                               int cz, int cy, int cx, int lcxp, int znorm, int flip_x, float* __restrict__ out) {
    __shared__ double rs[2][CF_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cxp = 1 << lcxp, rpw = 64 >> lcxp;              // rows a wave takes per iteration
    const int x = lane & (cxp - 1), sub = lane >> lcxp;
    const int x0 = c_x - cx / 2, y0 = c_y - cy / 2, z0 = c_z - cz / 2;
    const int n_rows = cz * cy, step = (CF_T / 64) * rpw;
    const bool x_ok = x < cx;
    const int sx = clampi(x0 + (flip_x ? cx - 1 - x : x), 0, W - 1);
    const long HW = (long)H * W;
    float v[CF_MAXIT];
    int row = wv * rpw + sub;
    int z = row / cy, y = row - z * cy;                       // (one division per thread; the march below only adds)
    const int dz = step / cy, dy = step - dz * cy;
    double s = 0, ss = 0;
#pragma unroll
    for (int k = 0; k < CF_MAXIT; ++k) {
        v[k] = 0.f;
        if (row + k * step < n_rows && x_ok) {
            v[k] = vol[(long)clampi(z0 + z, 0, D - 1) * HW + (long)clampi(y0 + y, 0, H - 1) * W + sx];
            s += v[k]; ss += (double)v[k] * v[k];
        }
        z += dz; y += dy;
        if (y >= cy) { y -= cy; ++z; }
    }
    float m = 0.f, inv = 1.f;
    if (znorm) {
        s = wave_sum(s); ss = wave_sum(ss);
        if (lane == 0) { rs[0][wv] = s; rs[1][wv] = ss; }
        __syncthreads();
        double S = 0, SS = 0;
#pragma unroll
        for (int w = 0; w < CF_T / 64; ++w) { S += rs[0][w]; SS += rs[1][w]; }
        const int vox = n_rows * cx;
        const double mean = S / vox;
        double var = (SS - vox * mean * mean) / (vox - 1);    // unbiased, as torch.std
        if (var < 0) var = 0;
        m = (float)mean; inv = (float)(1.0 / sqrt(var));
    }
    float* o = out + (long)blockIdx.x * n_rows * cx;
#pragma unroll
    for (int k = 0; k < CF_MAXIT; ++k) {
        const int r = row + k * step;
        if (r < n_rows && x_ok) o[(long)r * cx + x] = znorm ? (v[k] - m) * inv : v[k];
    }
}

__global__ __launch_bounds__(CF_T) void crop_fast_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                        const int* __restrict__ centres, int cz, int cy, int cx, int lcxp,
                                                        int znorm, int flip_x, float* __restrict__ out) {
    const int n = blockIdx.x;
    crop_fast_body(vol, D, H, W, centres[3 * n + 0], centres[3 * n + 1], centres[3 * n + 2], cz, cy, cx, lcxp, znorm, flip_x, out);
}

// 0 when the fast kernel does not take the shape / mode, else log2 of the padded row width + 1
int crop_fast_shape(int cz, int cy, int cx, int mode) {
    if ((mode != CROP_RAW && mode != CROP_ZNORM) || cx > 64 || getenv("MI_CROP_GENERIC")) return 0;
    int l = 0;
    while ((1 << l) < cx) ++l;
    const int step = (CF_T / 64) * (64 >> l);
    if (((long)cz * cy + step - 1) / step > CF_MAXIT) return 0;
    return l + 1;
}

__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                  const int* __restrict__ centres, int cz, int cy,
                                                  int cx, int mode, int flip_x, float* __restrict__ out) {
    const int n = blockIdx.x;
    crop_body(vol, D, H, W, centres[3 * n + 0], centres[3 * n + 1], centres[3 * n + 2], cz, cy, cx, mode, flip_x, out);
}

// The same crops with everything a batch needs already on the device (mi_crop_normalize_table): crop n of the launch is
// sample order[first + n] of the dataset; its tomogram, centre and (second view) shift come from per-sample tables.  No
// per-batch host work, no index upload: the reference's DataLoader workers (datasets/particle_pre_3d_vol.py:70-85 under
// moco_main.py:122-130) become two launches per batch.
__global__ __launch_bounds__(256) void crop_table_kernel(const mi_vol_desc* __restrict__ vols, const int* __restrict__ owner,
                                                        const int* __restrict__ centres, const int* __restrict__ shift,
                                                        const long long* __restrict__ order, long long first, int cz, int cy,
                                                        int cx, int mode, int flip_x, float* __restrict__ out) {
    const long long i = order ? order[first + blockIdx.x] : first + blockIdx.x;
    const mi_vol_desc d = vols[owner ? owner[i] : 0];
    int c_x = centres[3 * i + 0], c_y = centres[3 * i + 1], c_z = centres[3 * i + 2];
    if (shift) { c_x += shift[3 * i + 0]; c_y += shift[3 * i + 1]; c_z += shift[3 * i + 2]; }
    crop_body(d.vol, d.D, d.H, d.W, c_x, c_y, c_z, cz, cy, cx, mode, flip_x, out);
}

__global__ __launch_bounds__(CF_T) void crop_table_fast_kernel(const mi_vol_desc* __restrict__ vols, const int* __restrict__ owner,
                                                              const int* __restrict__ centres, const int* __restrict__ shift,
                                                              const long long* __restrict__ order, long long first, int cz,
                                                              int cy, int cx, int lcxp, int znorm, int flip_x,
                                                              float* __restrict__ out) {
    const long long i = order ? order[first + blockIdx.x] : first + blockIdx.x;
    const mi_vol_desc d = vols[owner ? owner[i] : 0];
    int c_x = centres[3 * i + 0], c_y = centres[3 * i + 1], c_z = centres[3 * i + 2];
    if (shift) { c_x += shift[3 * i + 0]; c_y += shift[3 * i + 1]; c_z += shift[3 * i + 2]; }
    crop_fast_body(d.vol, d.D, d.H, d.W, c_x, c_y, c_z, cz, cy, cx, lcxp, znorm, flip_x, out);
}

}  // namespace

extern "C" int mi_crop_normalize(const float* vol, int D, int H, int W, const int32_t* centres_xyz,
                                 int n, int cz, int cy, int cx, int mode, int flip_x, float* out,
                                 mi_stream_t stream) {
    if (n == 0) return MI_OK;
    if (!vol || !centres_xyz || !out || D <= 0 || H <= 0 || W <= 0 || n < 0) return MI_E_ARG;
    if (cz <= 0 || cy <= 0 || cx <= 0 || mode < 0 || mode > 3) return MI_E_ARG;
    if (mode == CROP_ZNORM_RESCALE_ZNORM) {
        const size_t bytes = sizeof(float) * (size_t)cz * cy * cx;
        if (bytes > 128 * 1024) return MI_E_UNSUPPORTED;            // (6, 48, 48) = 54 KiB; 32^3 = 128 KiB
        static bool attr_set = false;
        if (!attr_set) {
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(crop_chain_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(crop_chain_kernel, dim3(n), dim3(256), bytes, (hipStream_t)stream, vol, D, H, W,
                           (const int*)centres_xyz, cz, cy, cx, flip_x, out);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    if (const int f = crop_fast_shape(cz, cy, cx, mode)) {
        hipLaunchKernelGGL(crop_fast_kernel, dim3(n), dim3(CF_T), 0, (hipStream_t)stream, vol, D, H, W, (const int*)centres_xyz,
                           cz, cy, cx, f - 1, mode == CROP_ZNORM ? 1 : 0, flip_x, out);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    size_t lds = (mode == CROP_SUMZ_MINMAX) ? sizeof(float) * (size_t)cy * cx : 0;
    if (lds > 64 * 1024) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL(crop_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, vol, D, H, W,
                       (const int*)centres_xyz, cz, cy, cx, mode, flip_x, out);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_crop_normalize_table(const mi_vol_desc* vols, const int32_t* owner, const int32_t* centres_xyz,
                                       const int32_t* shift_xyz, const int64_t* order, int64_t first, int n, int cz, int cy,
                                       int cx, int mode, int flip_x, float* out, mi_stream_t stream) {
    if (n == 0) return MI_OK;
    if (!vols || !centres_xyz || !out || n < 0 || first < 0) return MI_E_ARG;
    if (cz <= 0 || cy <= 0 || cx <= 0 || mode < 0 || mode > 2) return MI_E_ARG;      // (mode 3 keeps its own kernel)
    if (const int f = crop_fast_shape(cz, cy, cx, mode)) {
        hipLaunchKernelGGL(crop_table_fast_kernel, dim3(n), dim3(CF_T), 0, (hipStream_t)stream, vols, (const int*)owner,
                           (const int*)centres_xyz, (const int*)shift_xyz, (const long long*)order, (long long)first, cz, cy, cx,
                           f - 1, mode == CROP_ZNORM ? 1 : 0, flip_x, out);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    const size_t lds = (mode == CROP_SUMZ_MINMAX) ? sizeof(float) * (size_t)cy * cx : 0;
    if (lds > 64 * 1024) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL(crop_table_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, vols, (const int*)owner,
                       (const int*)centres_xyz, (const int*)shift_xyz, (const long long*)order, (long long)first, cz, cy, cx,
                       mode, flip_x, out);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_u8_roundtrip_normalize(const float* x, float* y, size_t n, float mean, float std, mi_stream_t stream) {
    if (n == 0) return MI_OK;
    if (!x || !y || !(std > 0.f)) return MI_E_ARG;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(u8_roundtrip_normalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, mean, 0.f, std);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
