// Memory-bound glue of the detector network (models/networks/unet.py, unet_small.py)
// around the implicit-GEMM convolutions: 2-D max pooling with ceil_mode, the 2x2 stride-2 transposed
// convolution's pixel shuffle, the skip concatenation and the (3,1,1) heads with a handful of outputs.
// Channels-last fp32 everywhere; every kernel moves float4s and is bound by HBM bandwidth.
#include "common.h"
#include "../../include/cetpick_hip.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
int ew_blocks(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 4096)); }

// nn.MaxPool2d(k, stride=k, ceil_mode=True) per image (unet.py:231-233): windows may hang over the far edge
__global__ __launch_bounds__(256) void maxpool2d_kernel(const float* x, float* y, uint8_t* arg, int N, int Hi,
                                                       int Wi, int C, int Ho, int Wo, int k) {
    const int CV = C >> 2;
    const long total = (long)N * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long v = i / CV;
        const int xo = (int)(v % Wo); v /= Wo;
        const int yo = (int)(v % Ho);
        const int n = (int)(v / Ho);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int b = 0; b < k; ++b) {
            const int yi = yo * k + b;
            if (yi >= Hi) break;
            for (int c = 0; c < k; ++c) {
                const int xi = xo * k + c;
                if (xi >= Wi) break;
                const float4 t = ld4(x + (((long)n * Hi + yi) * Wi + xi) * C + 4 * cv);
                const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (first || tv[q] > best[q] || tv[q] != tv[q]) { best[q] = tv[q]; bi[q] = b * k + c; }
                first = false;
            }
        }
        st4(y + 4 * i, make_float4(best[0], best[1], best[2], best[3]));
        if (arg) *reinterpret_cast<uchar4*>(arg + 4 * i) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

// dx[n][yi][xi][c] = dy of the window that holds (yi, xi) when this element was its arg-max (windows are
// disjoint: stride == k)
__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const float* dy, const uint8_t* arg, float* dx, int N,
                                                           int Hi, int Wi, int C, int Ho, int Wo, int k) {
    const int CV = C >> 2;
    const long total = (long)N * Hi * Wi * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long v = i / CV;
        const int xi = (int)(v % Wi); v /= Wi;
        const int yi = (int)(v % Hi);
        const int n = (int)(v / Hi);
        const int yo = yi / k, xo = xi / k, tap = (yi - yo * k) * k + (xi - xo * k);
        const long o = (((long)n * Ho + yo) * Wo + xo) * C + 4 * cv;
        const uchar4 am = *reinterpret_cast<const uchar4*>(arg + o);
        const float4 d = ld4(dy + o);
        st4(dx + 4 * i, make_float4(am.x == tap ? d.x : 0.f, am.y == tap ? d.y : 0.f, am.z == tap ? d.z : 0.f,
                                    am.w == tap ? d.w : 0.f));
    }
}

// nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2) (unet.py:155-160) = a 1x1 convolution to 4*Co columns
// (column (a*2 + b)*Co + co, done by the implicit-GEMM kernel) followed by this shuffle:
//   y[n][2h + a][2w + b][co] = t[n][h][w][(a*2 + b)*Co + co] + bias[co],   cropped to (Ho, Wo) (autocrop, :253-266)
__global__ __launch_bounds__(256) void shuffle2x2_kernel(const float* t, const float* bias, float* y, int N, int H,
                                                        int W, int Co, int Ho, int Wo) {
    const int CV = Co >> 2;
    const long total = (long)N * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long v = i / CV;
        const int xo = (int)(v % Wo); v /= Wo;
        const int yo = (int)(v % Ho);
        const int n = (int)(v / Ho);
        const int h = yo >> 1, a = yo & 1, w = xo >> 1, b = xo & 1;
        float4 r = ld4(t + ((((long)n * H + h) * W + w) * 4 + (a * 2 + b)) * Co + 4 * cv);
        if (bias) { const float4 bb = ld4(bias + 4 * cv); r.x += bb.x; r.y += bb.y; r.z += bb.z; r.w += bb.w; }
        st4(y + 4 * i, r);
    }
}
// Inference tail of an up-convolution block in ONE pass (unet.py:319-399: ConvTranspose2d -> BatchNorm -> ReLU -> cat with the encoder
// feature): out[n][yo][xo][0:Co] = relu(scale * t[n][yo/2][xo/2][(a*2+b)*Co + co] + shift), out[..][Co:Co+Ce] = enc[n][yo][xo][:] - instead
// of the shuffle, the BatchNorm pass and the concatenation each reading and writing the feature map (scale / shift: the evaluation-mode
// BatchNorm with the transposed convolution's bias folded in, built by the caller).
__global__ __launch_bounds__(256) void upconv_tail_kernel(const float* t, const float* scale, const float* shift, const float* enc,
                                                         float* out, int N, int H, int W, int Co, int Ce, int Ho, int Wo) {
    const int CU = Co >> 2, CV = (Co + Ce) >> 2;
    const long total = (long)N * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long v = i / CV;
        float4 r;
        if (cv < CU) {
            const int xo = (int)(v % Wo); long q = v / Wo;
            const int yo = (int)(q % Ho);
            const int n = (int)(q / Ho);
            const int h = yo >> 1, a = yo & 1, w = xo >> 1, b = xo & 1;
            r = ld4(t + ((((long)n * H + h) * W + w) * 4 + (a * 2 + b)) * Co + 4 * cv);
            const float4 sc = ld4(scale + 4 * cv), sh = ld4(shift + 4 * cv);
            r.x = fmaxf(fmaf(r.x, sc.x, sh.x), 0.f); r.y = fmaxf(fmaf(r.y, sc.y, sh.y), 0.f);
            r.z = fmaxf(fmaf(r.z, sc.z, sh.z), 0.f); r.w = fmaxf(fmaf(r.w, sc.w, sh.w), 0.f);
        } else {
            r = ld4(enc + v * Ce + 4 * (cv - CU));
        }
        st4(out + 4 * i, r);
    }
}
// backward of the shuffle: dt[n][h][w][(a*2+b)*Co + co] = dy[n][2h+a][2w+b][co] (0 in the cropped rim)
__global__ __launch_bounds__(256) void unshuffle2x2_kernel(const float* dy, float* dt, int N, int H, int W, int Co,
                                                          int Ho, int Wo) {
    const int CV = Co >> 2;
    const long total = (long)N * H * W * 4 * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long v = i / CV;
        const int ab = (int)(v % 4); v /= 4;
        const int w = (int)(v % W); v /= W;
        const int h = (int)(v % H);
        const int n = (int)(v / H);
        const int yo = 2 * h + (ab >> 1), xo = 2 * w + (ab & 1);
        float4 r = make_float4(0, 0, 0, 0);
        if (yo < Ho && xo < Wo) r = ld4(dy + (((long)n * Ho + yo) * Wo + xo) * Co + 4 * cv);
        st4(dt + 4 * i, r);
    }
}

// torch.cat((a, b), 1) in channels-last: out[m][0:Ca] = a[m], out[m][Ca:Ca+Cb] = b[m]; b may be a crop
// (autocrop's centre crop is the identity for 'same' convolutions, so rows coincide)
__global__ __launch_bounds__(256) void concat_kernel(const float* a, int Ca, const float* b, int Cb, float* out,
                                                    long M) {
    const int CV = (Ca + Cb) >> 2, CA = Ca >> 2;
    const long total = M * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long m = i / CV;
        st4(out + 4 * i, cv < CA ? ld4(a + m * Ca + 4 * cv) : ld4(b + m * Cb + 4 * (cv - CA)));
    }
}
// backward: split d(out) into da, db
__global__ __launch_bounds__(256) void split_kernel(const float* dout, float* da, int Ca, float* db, int Cb, long M) {
    const int CV = (Ca + Cb) >> 2, CA = Ca >> 2;
    const long total = M * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long m = i / CV;
        const float4 v = ld4(dout + 4 * i);
        if (cv < CA) st4(da + m * Ca + 4 * cv, v); else st4(db + m * Cb + 4 * (cv - CA), v);
    }
}

// nn.Conv3d(C, K, kernel_size=(3,1,1), padding=(1,0,0), bias=False) with K <= 4 outputs (the `hm` head,
// unet_small.py:55-62): y[n][z][p][k] = sum_{dz, c} x[n][z+dz-1][p][c] * w[dz][c][k].  One thread per voxel,
// the 3*C*K weights in LDS; reads 3*C floats per voxel (2 of the 3 planes hit L2).
template <int K>
__global__ __launch_bounds__(256) void zhead_kernel(const float* x, const float* w, float* y, int N, int D, long P,
                                                   int C) {
    extern __shared__ float ws[];                      // [3][C][K]
    for (int i = threadIdx.x; i < 3 * C * K; i += 256) ws[i] = w[i];
    __syncthreads();
    const long total = (long)N * D * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pz = i / P;
        const int z = (int)(pz % D);
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
        for (int dz = 0; dz < 3; ++dz) {
            const int zz = z + dz - 1;
            if ((unsigned)zz >= (unsigned)D) continue;
            const float* xr = x + (i + (long)(dz - 1) * P) * C;
            const float* wr = ws + dz * C * K;
            for (int c = 0; c < C; c += 4) {
                const float4 v = ld4(xr + c);
#pragma unroll
                for (int k = 0; k < K; ++k)
                    acc[k] = fmaf(v.x, wr[c * K + k], fmaf(v.y, wr[(c + 1) * K + k],
                             fmaf(v.z, wr[(c + 2) * K + k], fmaf(v.w, wr[(c + 3) * K + k], acc[k]))));
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) y[i * K + k] = acc[k];
    }
}

// The same head with C / 4 lanes per voxel (C = 16 / 32 / 64; round 4): a thread per voxel reads its 128-byte rows with eight 16-byte loads
// whose lanes are a row apart - 64 cache lines per instruction (0.50 ms for the 1 GB feature volume of a 128 x 512 x 512 tomogram).  Here
// the G lanes of a voxel read one row as ONE coalesced line, keep partial dot products and add them with a butterfly.
template <int K, int G>
__global__ __launch_bounds__(256) void zhead_rows_kernel(const float* x, const float* w, float* y, int N, int D, long P) {
    constexpr int C = 4 * G;
    __shared__ float ws[3 * C * K];                    // [3][C][K]
    for (int i = threadIdx.x; i < 3 * C * K; i += 256) ws[i] = w[i];
    __syncthreads();
    const long total = (long)N * D * P;
    const int g = threadIdx.x % G;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) / G; i < total; i += (long)gridDim.x * (256 / G)) {
        const int z = (int)((i / P) % D);
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int zz = z + dz - 1;
            if ((unsigned)zz >= (unsigned)D) continue;
            const float4 v = ld4(x + (i + (long)(dz - 1) * P) * C + 4 * g);
            const float* wr = ws + (dz * C + 4 * g) * K;
#pragma unroll
            for (int k = 0; k < K; ++k)
                acc[k] = fmaf(v.x, wr[k], fmaf(v.y, wr[K + k], fmaf(v.z, wr[2 * K + k], fmaf(v.w, wr[3 * K + k], acc[k]))));
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) y[i * K + k] = acc[k];
        }
    }
}

// backward of the (3,1,1) head: dx[n][z][p][c] = sum_{dz, k} dy[n][z - dz + 1][p][k] * w[dz][c][k]
template <int K>
__global__ __launch_bounds__(256) void zhead_bwd_data_kernel(const float* dy, const float* w, float* dx, int N, int D,
                                                            long P, int C) {
    extern __shared__ float ws[];                      // [3][C][K]
    for (int i = threadIdx.x; i < 3 * C * K; i += 256) ws[i] = w[i];
    __syncthreads();
    const int CV = C >> 2;
    const long total = (long)N * D * P * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long v = i / CV;                         // voxel (n, z, p)
        const int z = (int)((v / P) % D);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int dz = 0; dz < 3; ++dz) {
            const int zo = z - dz + 1;
            if ((unsigned)zo >= (unsigned)D) continue;
            const float* dyr = dy + (v + (long)(1 - dz) * P) * K;
            const float* wr = ws + (dz * C + 4 * cv) * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float g = dyr[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(g, wr[q * K + k], acc[q]);
            }
        }
        st4(dx + 4 * i, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// dw[dz][c][k] = sum over voxels of x[n][z + dz - 1][p][c] * dy[n][z][p][k]: thread (c, row group) accumulates
// 3*K partial sums over its voxels; fixed-shape block tree, fp64 partials, deterministic finalize
template <int K>
__global__ __launch_bounds__(256) void zhead_bwd_weight_kernel(const float* x, const float* dy, int N, int D, long P,
                                                              int C, double* partials) {
    const int RG = 256 / C;                            // C in {16, 32, 64, 128, 256}
    const int c = threadIdx.x % C, rg = threadIdx.x / C;
    const long total = (long)N * D * P;
    float acc[3][K];
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[dz][k] = 0.f;
    for (long v = (long)blockIdx.x * RG + rg; v < total; v += (long)gridDim.x * RG) {
        const int z = (int)((v / P) % D);
        float g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = dy[v * K + k];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int zz = z + dz - 1;
            if ((unsigned)zz >= (unsigned)D) continue;
            const float xv = x[(v + (long)(dz - 1) * P) * C + c];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[dz][k] = fmaf(xv, g[k], acc[dz][k]);
        }
    }
    __shared__ double red[256];
    for (int dz = 0; dz < 3; ++dz)
        for (int k = 0; k < K; ++k) {
            red[threadIdx.x] = (double)acc[dz][k];
            __syncthreads();
            if (rg == 0) {
                double s = 0;
                for (int r = 0; r < RG; ++r) s += red[r * C + c];
                partials[((long)blockIdx.x * 3 + dz) * C * K + c * K + k] = s;
            }
            __syncthreads();
        }
}
__global__ __launch_bounds__(256) void zhead_bwd_weight_final_kernel(const double* partials, int n_part, int n_out,
                                                                    float* dw) {
    const int o = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double s = 0;
    if (o < n_out)
        for (int b = l; b < n_part; b += 32) s += partials[(long)b * n_out + o];
#pragma unroll
    for (int q = 16; q > 0; q >>= 1) s += __shfl_xor(s, q, 32);
    if (l == 0 && o < n_out) dw[o] = (float)s;
}
int zhead_w_blocks(long voxels, int C) { return (int)std::max<long>(1, std::min<long>(voxels / ((256 / C) * 64) + 1, 1024)); }

// ---- the detector's first convolution at inference: nn.Conv2d(1, 16, 7, stride 2, padding 3) per slice (unet_small.py:35) ----
// One input channel: 49 taps x 16 outputs.  The implicit GEMM gathers 49 single floats per row behind a tap table (1.9 ms per
// 128 x 512 x 512 tomogram, 7 TFLOP/s: 13 GFLOP that write 0.5 GB).  Here a workgroup owns a 16 x 16 tile of output pixels: the 37 x 37
// input patch goes through LDS once, a thread keeps its pixel's 16 accumulators and walks the window a row at a time, the weights come
// as scalar operands (they are the same for every lane); bias + ReLU in the epilogue, four 16-byte stores.
constexpr int S2D_T = 16, S2D_P = 2 * S2D_T + 5, S2D_PP = S2D_P + 1;          // tile, patch extent, patch pitch
__global__ __launch_bounds__(256) void stem2d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int relu,
                                                        int H, int W, int Ho, int Wo) {
    __shared__ float patch[S2D_P * S2D_PP];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, ty0 = blockIdx.y * S2D_T, tx0 = blockIdx.x * S2D_T;
    const float* img = x + (long)n * H * W;
    const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
    for (int i = tid; i < S2D_P * S2D_P; i += 256) {
        const int py = i / S2D_P, px = i % S2D_P;
        const int yy = iy0 + py, xx = ix0 + px;
        patch[py * S2D_PP + px] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[(long)yy * W + xx] : 0.f;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int yo = ty0 + ly, xo = tx0 + lx;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    // one window row at a time (a rolled loop: unrolled, the compiler hoisted all 784 weights into registers and spilled); the weights
    // are the same for every lane: scalar loads, a scalar operand per multiply-add - no LDS traffic for them
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float v[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) v[kx] = patch[(2 * ly + ky) * S2D_PP + 2 * lx + kx];
        const float* wr = w + ky * 7 * 16;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(v[kx], wr[kx * 16 + c], acc[c]);
    }
    if (yo >= Ho || xo >= Wo) return;
    float* out = y + (((long)n * Ho + yo) * Wo + xo) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 b4 = bias ? ld4(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o = make_float4(acc[4 * q] + b4.x, acc[4 * q + 1] + b4.y, acc[4 * q + 2] + b4.z, acc[4 * q + 3] + b4.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4(out + 4 * q, o);
    }
}

}  // namespace

/* y = act(conv2d(x, w, 7 x 7, stride 2, padding 3) + bias) for ONE input channel and 16 output channels per image: x (N, H, W),
 * w [7][7][1][16] (kernel layout), bias[16] or NULL, y (N, Ho, Wo, 16) with Ho = (H - 1) / 2 + 1.  Inference-path replacement of the
 * implicit GEMM for the detector's first layer (models/networks/unet_small.py:35, `conv1`). */
extern "C" int mi_stem2d_fwd_bias_f32(const float* x, const float* w, const float* bias, float* y, int relu, int N, int H, int W,
                                      mi_stream_t stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0) return MI_E_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (N > 65535 || (Ho + S2D_T - 1) / S2D_T > 65535) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL(stem2d_fwd_kernel, dim3((unsigned)((Wo + S2D_T - 1) / S2D_T), (unsigned)((Ho + S2D_T - 1) / S2D_T), (unsigned)N),
                       dim3(256), 0, (hipStream_t)stream, x, w, bias, y, relu, H, W, Ho, Wo);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_maxpool2d_ceil_fwd(const float* x, float* y, uint8_t* argmax, int N, int Hi, int Wi, int C, int k,
                                     mi_stream_t stream) {
    if (!x || !y || C % 4 || k <= 0 || k > 15 || N <= 0 || Hi <= 0 || Wi <= 0) return MI_E_ARG;
    const int Ho = (Hi + k - 1) / k, Wo = (Wi + k - 1) / k;
    const long total = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool2d_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, N,
                       Hi, Wi, C, Ho, Wo, k);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_maxpool2d_ceil_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int Hi, int Wi, int C,
                                     int k, mi_stream_t stream) {
    if (!dy || !argmax || !dx || C % 4 || k <= 0 || k > 15 || N <= 0 || Hi <= 0 || Wi <= 0) return MI_E_ARG;
    const int Ho = (Hi + k - 1) / k, Wo = (Wi + k - 1) / k;
    const long total = (long)N * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(maxpool2d_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx,
                       N, Hi, Wi, C, Ho, Wo, k);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_shuffle2x2_fwd(const float* t, const float* bias, float* y, int N, int H, int W, int Co, int Ho,
                                 int Wo, mi_stream_t stream) {
    if (!t || !y || Co % 4 || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Ho > 2 * H || Wo > 2 * W) return MI_E_ARG;
    const long total = (long)N * Ho * Wo * (Co / 4);
    hipLaunchKernelGGL(shuffle2x2_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, t, bias, y, N, H, W,
                       Co, Ho, Wo);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
/* out = cat(relu(scale * shuffle(t) + shift), enc) over the channel axis, channels-last: t (N, H, W, 4*Co) from the 1 x 1 product of the
 * transposed convolution, enc (N, Ho, Wo, Ce), out (N, Ho, Wo, Co + Ce); Ho in {2H - 1, 2H}, Wo likewise (the reference's autocrop). */
extern "C" int mi_upconv_tail_fwd(const float* t, const float* scale, const float* shift, const float* enc, float* out, int N, int H,
                                  int W, int Co, int Ce, int Ho, int Wo, mi_stream_t stream) {
    if (!t || !scale || !shift || !enc || !out || N <= 0 || H <= 0 || W <= 0 || Co <= 0 || Ce <= 0 || (Co & 3) || (Ce & 3)) return MI_E_ARG;
    if (Ho > 2 * H || Ho < 2 * H - 1 || Wo > 2 * W || Wo < 2 * W - 1) return MI_E_ARG;
    const long total = (long)N * Ho * Wo * ((Co + Ce) >> 2);
    hipLaunchKernelGGL(upconv_tail_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, t, scale, shift, enc, out, N, H, W,
                       Co, Ce, Ho, Wo);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// dst[m][c0 : c0 + Cs] = src[m][:] for the M rows of a (M, Ct) tensor: the encoder feature into the concatenation buffer whose first
// channels the up-convolution's fused epilogue writes (mi_conv_d32_upconv_fwd_f32)
__global__ __launch_bounds__(256) void copy_channels_into_kernel(const float* src, int Cs, float* dst, int Ct, int c0, long M) {
    const int CV = Cs >> 2;
    const long total = M * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / CV;
        const int cv = (int)(i - m * CV);
        st4(dst + m * Ct + c0 + 4 * cv, ld4(src + 4 * i));
    }
}
extern "C" int mi_copy_channels_into(const float* src, int Cs, float* dst, int Ct, int c0, long M, mi_stream_t stream) {
    if (!src || !dst || Cs <= 0 || (Cs & 3) || (Ct & 3) || (c0 & 3) || c0 < 0 || c0 + Cs > Ct || M <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(copy_channels_into_kernel, dim3(ew_blocks(M * (Cs >> 2))), dim3(256), 0, (hipStream_t)stream, src, Cs, dst, Ct, c0, M);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_shuffle2x2_bwd(const float* dy, float* dt, int N, int H, int W, int Co, int Ho, int Wo,
                                 mi_stream_t stream) {
    if (!dy || !dt || Co % 4 || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Ho > 2 * H || Wo > 2 * W) return MI_E_ARG;
    const long total = (long)N * H * W * Co;
    hipLaunchKernelGGL(unshuffle2x2_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dt, N, H, W,
                       Co, Ho, Wo);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_concat_channels(const float* a, int Ca, const float* b, int Cb, float* out, long M,
                                  mi_stream_t stream) {
    if (!a || !b || !out || Ca <= 0 || Cb <= 0 || Ca % 4 || Cb % 4 || M <= 0) return MI_E_ARG;
    const long total = M * ((Ca + Cb) / 4);
    hipLaunchKernelGGL(concat_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, Ca, b, Cb, out, M);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_split_channels(const float* dout, float* da, int Ca, float* db, int Cb, long M, mi_stream_t stream) {
    if (!dout || !da || !db || Ca <= 0 || Cb <= 0 || Ca % 4 || Cb % 4 || M <= 0) return MI_E_ARG;
    const long total = M * ((Ca + Cb) / 4);
    hipLaunchKernelGGL(split_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dout, da, Ca, db, Cb, M);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_zhead_fwd(const float* x, const float* w, float* y, int N, int D, long P, int C, int K,
                            mi_stream_t stream) {
    if (!x || !w || !y || N <= 0 || D <= 0 || P <= 0 || C <= 0 || C % 4 || K < 1 || K > 4) return MI_E_ARG;
    const long total = (long)N * D * P;
    const size_t lds = sizeof(float) * 3 * (size_t)C * K;
    if (lds > 48 * 1024) return MI_E_UNSUPPORTED;
    const dim3 grid(ew_blocks(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if ((C == 16 || C == 32 || C == 64) && total >= 4096 && !getenv("MI_ZHEAD_GENERIC")) {     // C / 4 lanes per voxel: coalesced rows
        const dim3 gr((unsigned)std::min<long>((total * (C / 4) + 255) / 256, 1l << 20));
#define ZH(Kv, Gv) hipLaunchKernelGGL((zhead_rows_kernel<Kv, Gv>), gr, block, 0, s, x, w, y, N, D, P)
#define ZHK(Gv) do { if (K == 1) ZH(1, Gv); else if (K == 2) ZH(2, Gv); else if (K == 3) ZH(3, Gv); else ZH(4, Gv); } while (0)
        if (C == 16) ZHK(4); else if (C == 32) ZHK(8); else ZHK(16);
#undef ZHK
#undef ZH
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    switch (K) {
        case 1: hipLaunchKernelGGL((zhead_kernel<1>), grid, block, lds, s, x, w, y, N, D, P, C); break;
        case 2: hipLaunchKernelGGL((zhead_kernel<2>), grid, block, lds, s, x, w, y, N, D, P, C); break;
        case 3: hipLaunchKernelGGL((zhead_kernel<3>), grid, block, lds, s, x, w, y, N, D, P, C); break;
        default: hipLaunchKernelGGL((zhead_kernel<4>), grid, block, lds, s, x, w, y, N, D, P, C); break;
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" size_t mi_zhead_bwd_workspace_bytes(int N, int D, long P, int C, int K) {
    if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || 256 % C || K < 1 || K > 4) return 0;
    return sizeof(double) * 3 * (size_t)C * K * zhead_w_blocks((long)N * D * P, C);
}

/* backward of mi_zhead_fwd: dx (may be NULL) and dw [3][C][K] (may be NULL) */
extern "C" int mi_zhead_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int N, int D, long P,
                            int C, int K, void* ws, size_t ws_bytes, mi_stream_t stream) {
    if (!x || !w || !dy || N <= 0 || D <= 0 || P <= 0 || C <= 0 || C % 4 || 256 % C || K < 1 || K > 4) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long vox = (long)N * D * P;
    if (dx) {
        const size_t lds = sizeof(float) * 3 * (size_t)C * K;
        if (lds > 48 * 1024) return MI_E_UNSUPPORTED;
        const dim3 grid(ew_blocks(vox * (C / 4))), block(256);
        switch (K) {
            case 1: hipLaunchKernelGGL((zhead_bwd_data_kernel<1>), grid, block, lds, s, dy, w, dx, N, D, P, C); break;
            case 2: hipLaunchKernelGGL((zhead_bwd_data_kernel<2>), grid, block, lds, s, dy, w, dx, N, D, P, C); break;
            case 3: hipLaunchKernelGGL((zhead_bwd_data_kernel<3>), grid, block, lds, s, dy, w, dx, N, D, P, C); break;
            default: hipLaunchKernelGGL((zhead_bwd_data_kernel<4>), grid, block, lds, s, dy, w, dx, N, D, P, C); break;
        }
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    if (dw) {
        if (!ws || ws_bytes < mi_zhead_bwd_workspace_bytes(N, D, P, C, K)) return MI_E_WORKSPACE;
        const int blocks = zhead_w_blocks(vox, C);
        double* part = (double*)ws;
        switch (K) {
            case 1: hipLaunchKernelGGL((zhead_bwd_weight_kernel<1>), dim3(blocks), dim3(256), 0, s, x, dy, N, D, P, C, part); break;
            case 2: hipLaunchKernelGGL((zhead_bwd_weight_kernel<2>), dim3(blocks), dim3(256), 0, s, x, dy, N, D, P, C, part); break;
            case 3: hipLaunchKernelGGL((zhead_bwd_weight_kernel<3>), dim3(blocks), dim3(256), 0, s, x, dy, N, D, P, C, part); break;
            default: hipLaunchKernelGGL((zhead_bwd_weight_kernel<4>), dim3(blocks), dim3(256), 0, s, x, dy, N, D, P, C, part); break;
        }
        MI_RETURN_IF_LAUNCH_FAILED();
        const int n_out = 3 * C * K;
        hipLaunchKernelGGL(zhead_bwd_weight_final_kernel, dim3((n_out + 7) / 8), dim3(256), 0, s, (const double*)part,
                           blocks, n_out, dw);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}
