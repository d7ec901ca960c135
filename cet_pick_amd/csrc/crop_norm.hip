// Sub-tomogram crop + normalise, one workgroup per crop.
//
// Replaces the per-pick Python loops of the reference datasets:
//   datasets/tomo_pre_proj_angle_select_new3d_vol.py:117-128 `extract_subvols`
//        v[z-sz//2 : z+sz//2+1, y-sy//2 : y+sy//2, x-sx//2 : x+sx//2] -> sum over z -> min-max
//   :130-138 `extract_subvols_3d` (plain crop)
//   and the z-normalised 3-D crop that feeds the MoCo-3D encoder.
// HBM-bound gather: rows of the crop are contiguous along x, so lanes read consecutive floats.
#include "common.h"

namespace {

enum { CROP_RAW = 0, CROP_SUMZ_MINMAX = 1, CROP_ZNORM = 2 };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// window start along an axis: the reference slices [c - s//2, c - s//2 + s) (z windows are odd:
// c - s//2 .. c + s//2, xy windows even: c - s/2 .. c + s/2 - 1)
__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ vol, int D, int H, int W,
                                                  const int* __restrict__ centres, int cz, int cy,
                                                  int cx, int mode, int flip_x, float* __restrict__ out) {
    __shared__ float red[2][4];
    extern __shared__ float plane[];               // SUMZ: cy*cx floats
    const int n = blockIdx.x, tid = threadIdx.x;
    const int x0 = centres[3 * n + 0] - cx / 2, y0 = centres[3 * n + 1] - cy / 2, z0 = centres[3 * n + 2] - cz / 2;
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

}  // namespace

extern "C" int mi_crop_normalize(const float* vol, int D, int H, int W, const int32_t* centres_xyz,
                                 int n, int cz, int cy, int cx, int mode, int flip_x, float* out,
                                 mi_stream_t stream) {
    if (n == 0) return MI_OK;
    if (!vol || !centres_xyz || !out || D <= 0 || H <= 0 || W <= 0 || n < 0) return MI_E_ARG;
    if (cz <= 0 || cy <= 0 || cx <= 0 || mode < 0 || mode > 2) return MI_E_ARG;
    size_t lds = (mode == CROP_SUMZ_MINMAX) ? sizeof(float) * (size_t)cy * cx : 0;
    if (lds > 64 * 1024) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL(crop_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, vol, D, H, W,
                       (const int*)centres_xyz, cz, cy, cx, mode, flip_x, out);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
