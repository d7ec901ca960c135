// Non-GEMM kernels of the contrastive training step (channels-last fp32):
// BatchNorm (train/eval, fwd/bwd, split so SyncBN can all-reduce the per-channel sums),
// MaxPool3d, global average pool, bias, L2-normalise, MoCo logits, cross-entropy(label 0),
// momentum (EMA) update, SGD, queue enqueue.
//
// Replaces (reference, cet_pick/...): nn.BatchNorm3d/1d, nn.MaxPool3d, nn.AdaptiveAvgPool3d at
// models/networks/moco_encoder_3d.py:170-205; models/moco.py:31-52,111-141 (EMA, enqueue, logits);
// trains/tomo_moco_trainer.py:52,73 (CrossEntropyLoss); torch.optim.SGD at moco_main.py:79.
// All are HBM-bound elementwise / column-reduction passes.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// column reductions over [M][C] (C % 4 == 0, 256 % (C/4) == 0)
// ---------------------------------------------------------------------------------------------
enum { CR_STATS = 0, CR_BNBWD = 1, CR_SUM = 2, CR_BNBWD_X = 3 };

struct PoolGeom { int N, Di, Hi, Wi, Do, Ho, Wo, k, s, pad; };

struct ColReduceParams {
    const float* a;      // STATS: x ; BNBWD: dy ; SUM: dy
    const float* x;      // BNBWD: x
    const float* y;      // BNBWD: y (post-ReLU output) when relu != 0
    const float* save;   // BNBWD: mean[C], invstd[C]
    long M;
    int C, relu;
    double* partials;    // [gridDim.x][2][C]
    // BNBWD_X: the ReLU mask is recomputed from x (y = relu(bn(x)) was never stored): y > 0 <=> xhat*gamma + beta > 0
    const float* gamma;
    const float* beta;
    // a launch of ONE workgroup writes the final sums itself (no finalize launch): sums_direct[2][C], f32_direct[n_f32]
    double* sums_direct;
    float* f32_direct;
    int n_f32;
};

template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(ColReduceParams p) {
    const int CV = p.C >> 2;
    const int tcol = threadIdx.x % CV, trow = threadIdx.x / CV;
    const int RS = 256 / CV;
    float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float mean[4] = {0, 0, 0, 0}, inv[4] = {1, 1, 1, 1};
    if (MODE == CR_BNBWD || MODE == CR_BNBWD_X) {
        float4 m = ld4(p.save + 4 * tcol), iv = ld4(p.save + p.C + 4 * tcol);
        mean[0] = m.x; mean[1] = m.y; mean[2] = m.z; mean[3] = m.w;
        inv[0] = iv.x; inv[1] = iv.y; inv[2] = iv.z; inv[3] = iv.w;
    }
    for (long r = (long)blockIdx.x * RS + trow; r < p.M; r += (long)gridDim.x * RS) {
        const long o = r * p.C + 4 * tcol;
        float a[4];
        if (MODE == CR_BNBWD_X) {
            const float4 a4 = ld4(p.a + o);
            a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
            const float4 x4 = ld4(p.x + o);
            const float x[4] = {x4.x, x4.y, x4.z, x4.w};
            const float4 g4 = p.gamma ? ld4(p.gamma + 4 * tcol) : make_float4(1, 1, 1, 1);
            const float4 b4 = p.beta ? ld4(p.beta + 4 * tcol) : make_float4(0, 0, 0, 0);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (x[i] - mean[i]) * inv[i];
                const float d = fmaf(xh, g[i], b[i]) > 0.f ? a[i] : 0.f;       // ReLU mask from the recomputed y
                s0[i] += d; s1[i] = fmaf(d, xh, s1[i]);
            }
            continue;
        }
        float4 a4 = ld4(p.a + o);
        a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
        if (MODE == CR_STATS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] += a[i]; s1[i] = fmaf(a[i], a[i], s1[i]); }
        } else if (MODE == CR_BNBWD) {
            float4 x4 = ld4(p.x + o);
            float x[4] = {x4.x, x4.y, x4.z, x4.w};
            if (p.relu) {
                float4 y4 = ld4(p.y + o);
                float y[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = y[i] > 0.f ? a[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] += a[i]; s1[i] = fmaf(a[i], (x[i] - mean[i]) * inv[i], s1[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) s0[i] += a[i];
        }
    }
    __shared__ double red[2][256][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][threadIdx.x][i] = (double)s0[i]; red[1][threadIdx.x][i] = (double)s1[i]; }
    __syncthreads();
    if (trow == 0) {
        double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
        for (int rr = 0; rr < RS; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) { a0[i] += red[0][rr * CV + tcol][i]; a1[i] += red[1][rr * CV + tcol][i]; }
        double* dst = p.sums_direct ? p.sums_direct : p.partials + (long)blockIdx.x * 2 * p.C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[4 * tcol + i] = a0[i]; dst[p.C + 4 * tcol + i] = a1[i];
            if (p.f32_direct) {
                if (4 * tcol + i < p.n_f32) p.f32_direct[4 * tcol + i] = (float)a0[i];
                if (p.C + 4 * tcol + i < p.n_f32) p.f32_direct[p.C + 4 * tcol + i] = (float)a1[i];
            }
        }
    }
}

// 8 columns per workgroup, 32 lanes per column: fixed-shape tree -> deterministic
__global__ __launch_bounds__(256) void colreduce_finalize_kernel(const double* partials, int n_part,
                                                                int C, double* sums, float* sums_f32, int n_f32) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double s = 0;
    if (c < 2 * C) {
        // eight loads in flight, added in the rolled loop's order (same sums; rolled, each load waited for the one before it)
        int b = l;
        for (; b + 7 * 32 < n_part; b += 8 * 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(long)(b + 32 * u) * 2 * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < n_part; b += 32) s += partials[(long)b * 2 * C + c];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
    if (l == 0 && c < 2 * C) {
        sums[c] = s;
        if (sums_f32 && c < n_f32) sums_f32[c] = (float)s;       // (the column sums of a bias gradient leave as fp32 here)
    }
}

int colreduce_blocks(long M, int C) {
    int RS = 256 / (C / 4);
    long b = (M + (long)RS * 8 - 1) / ((long)RS * 8);
    long cap = 512;
    if (const char* v = getenv("MI_COLREDUCE_BLOCKS")) { const long cv = atol(v); if (cv >= 1 && cv <= 4096) cap = cv; }      // tuning
    return (int)std::max<long>(1, std::min<long>(b, cap));
}
bool colreduce_ok(int C) { return C >= 4 && (C % 4) == 0 && (C / 4) <= 256 && (256 % (C / 4)) == 0; }

template <int MODE>
int run_colreduce(ColReduceParams p, double* sums, void* ws, size_t ws_bytes, hipStream_t s, float* sums_f32 = nullptr,
                  int n_f32 = 0) {
    if (!colreduce_ok(p.C) || p.M <= 0) return MI_E_ARG;
    int blocks = colreduce_blocks(p.M, p.C);
    if (!ws || ws_bytes < sizeof(double) * 2 * p.C * (size_t)blocks) return MI_E_WORKSPACE;
    p.partials = (double*)ws;
    if (blocks == 1) {       // few rows: the one workgroup's partial sums ARE the sums (the finalize would add zeros to them)
        p.sums_direct = sums; p.f32_direct = sums_f32; p.n_f32 = n_f32;
        hipLaunchKernelGGL((colreduce_kernel<MODE>), dim3(1), dim3(256), 0, s, p);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    hipLaunchKernelGGL((colreduce_kernel<MODE>), dim3(blocks), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(colreduce_finalize_kernel, dim3((2 * p.C + 7) / 8), dim3(256), 0, s,
                       (const double*)ws, blocks, p.C, sums, sums_f32, n_f32);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm apply kernels
// ---------------------------------------------------------------------------------------------
// Training statistics -> (mean, invstd) for 4 channels, from the fp64 column sums
struct BnStat4 { float mean[4], inv[4]; double var[4]; };
__device__ __forceinline__ BnStat4 bn_stat4(const double* sums, double count, int C, int c, float eps) {
    BnStat4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mean = sums[c + k] / count;
        double var = sums[C + c + k] / count - mean * mean;
        if (var < 0) var = 0;
        r.mean[k] = (float)mean;
        r.inv[k] = (float)(1.0 / sqrt(var + (double)eps));
        r.var[k] = var;
    }
    return r;
}

// y = relu?((x - mean) * invstd * gamma + beta + res).  One launch does the whole forward BatchNorm:
//   sums != NULL (training): mean / invstd from the column sums (every thread recomputes its 4 channels: the
//     channel of a thread is loop-invariant because 256 % (C/4) == 0); workgroup 0 also writes
//     save = mean[C], invstd[C], the running statistics (momentum update, unbiased variance) and increments
//     num_batches_tracked.
//   sums == NULL (eval): mean / invstd from the running statistics.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, float* y, long n4, int C,
                                                      const double* sums, double count, float eps,
                                                      float momentum, float* running_mean,
                                                      float* running_var, long long* num_batches_tracked,
                                                      float* save, const float* gamma,
                                                      const float* beta, const float* res, int relu) {
    const int CV = C >> 2;
    const int c = (int)(threadIdx.x % CV) * 4;
    float m[4], iv[4];
    if (sums) {
        const BnStat4 st = bn_stat4(sums, count, C, c, eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = st.mean[k]; iv[k] = st.inv[k]; }
        if (blockIdx.x == 0 && threadIdx.x < CV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (save) { save[c + k] = m[k]; save[C + c + k] = iv[k]; }
                if (running_mean) {
                    const double unbiased = count > 1 ? st.var[k] * count / (count - 1.0) : st.var[k];
                    running_mean[c + k] = (1.f - momentum) * running_mean[c + k] + momentum * m[k];
                    running_var[c + k] = (1.f - momentum) * running_var[c + k] + momentum * (float)unbiased;
                }
            }
            if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = running_mean[c + k]; iv[k] = 1.0f / sqrtf(running_var[c + k] + eps); }
        if (save && blockIdx.x == 0 && threadIdx.x < CV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { save[c + k] = m[k]; save[C + c + k] = iv[k]; }
        }
    }
    const float4 g = gamma ? ld4(gamma + c) : make_float4(1, 1, 1, 1);
    const float4 b = beta ? ld4(beta + c) : make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = ld4(x + 4 * i);
        float4 o;
        o.x = fmaf((v.x - m[0]) * iv[0], g.x, b.x); o.y = fmaf((v.y - m[1]) * iv[1], g.y, b.y);
        o.z = fmaf((v.z - m[2]) * iv[2], g.z, b.z); o.w = fmaf((v.w - m[3]) * iv[3], g.w, b.w);
        if (res) { float4 r = ld4(res + 4 * i); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4(y + 4 * i, o);
    }
}

// ---- stem: BatchNorm + ReLU + MaxPool3d fused (moco_encoder_3d.py:170-172, :355-358).  y = relu(bn(x)) is
// never written: the forward pools bn(x) on the fly, the backward gathers d(pool) through the argmax taps and
// recomputes the ReLU mask from x.  Saves a write + three reads of the largest activation of the step (67 MB).
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float* x, float* y, uint8_t* arg, PoolGeom g,
                                                                 int C, const double* sums, double count, float eps,
                                                                 float momentum, float* running_mean,
                                                                 float* running_var, long long* num_batches_tracked,
                                                                 float* save, const float* gamma, const float* beta) {
    const int CV = C >> 2;
    const int c = (int)(threadIdx.x % CV) * 4;                  // loop-invariant: 256 % CV == 0
    float m[4], iv[4];
    if (sums) {
        const BnStat4 st = bn_stat4(sums, count, C, c, eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = st.mean[k]; iv[k] = st.inv[k]; }
        if (blockIdx.x == 0 && threadIdx.x < CV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (save) { save[c + k] = m[k]; save[C + c + k] = iv[k]; }
                if (running_mean) {
                    const double unbiased = count > 1 ? st.var[k] * count / (count - 1.0) : st.var[k];
                    running_mean[c + k] = (1.f - momentum) * running_mean[c + k] + momentum * m[k];
                    running_var[c + k] = (1.f - momentum) * running_var[c + k] + momentum * (float)unbiased;
                }
            }
            if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = running_mean[c + k]; iv[k] = 1.0f / sqrtf(running_var[c + k] + eps); }
    }
    const float4 g4 = gamma ? ld4(gamma + c) : make_float4(1, 1, 1, 1);
    const float4 b4 = beta ? ld4(beta + c) : make_float4(0, 0, 0, 0);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const long total = (long)g.N * g.Do * g.Ho * g.Wo * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        unsigned v = (unsigned)i / (unsigned)CV;                 // (the host checks total < 2^31)
        const int xo = (int)(v % (unsigned)g.Wo); v /= (unsigned)g.Wo;
        const int yo = (int)(v % (unsigned)g.Ho); v /= (unsigned)g.Ho;
        const int zo = (int)(v % (unsigned)g.Do);
        const int n = (int)(v / (unsigned)g.Do);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        if (g.k == 3) {
            // 3^3 window: the nine loads of a z-plane are issued together (addresses clamped into the tensor, validity
            // kept aside) instead of one dependent load per tap; same visiting order, so the same argmax on ties
            for (int a = 0; a < 3; ++a) {
                const int zi = zo * g.s - g.pad + a;
                const bool zok = (unsigned)zi < (unsigned)g.Di;
                float4 t9[9];
                bool ok9[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int yi = yo * g.s - g.pad + q / 3, xi = xo * g.s - g.pad + q % 3;
                    ok9[q] = zok & ((unsigned)yi < (unsigned)g.Hi) & ((unsigned)xi < (unsigned)g.Wi);
                    const int zc = min(max(zi, 0), g.Di - 1), yc = min(max(yi, 0), g.Hi - 1), xc = min(max(xi, 0), g.Wi - 1);
                    t9[q] = ld4(x + ((((long)n * g.Di + zc) * g.Hi + yc) * g.Wi + xc) * C + c);
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (!ok9[q]) continue;
                    const float xv[4] = {t9[q].x, t9[q].y, t9[q].z, t9[q].w};
                    const int tap = a * 9 + q;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = fmaxf(fmaf((xv[r] - m[r]) * iv[r], gg[r], bb[r]), 0.f);
                        const float tn = xv[r] != xv[r] ? xv[r] : t;          // a NaN input stays a NaN
                        if (first || tn > best[r] || tn != tn) { best[r] = tn; bi[r] = tap; }
                    }
                    first = false;
                }
            }
        } else
        for (int a = 0; a < g.k; ++a) {
            const int zi = zo * g.s - g.pad + a;
            if ((unsigned)zi >= (unsigned)g.Di) continue;
            for (int b = 0; b < g.k; ++b) {
                const int yi = yo * g.s - g.pad + b;
                if ((unsigned)yi >= (unsigned)g.Hi) continue;
                for (int cc = 0; cc < g.k; ++cc) {
                    const int xi = xo * g.s - g.pad + cc;
                    if ((unsigned)xi >= (unsigned)g.Wi) continue;
                    const float4 t4 = ld4(x + ((((long)n * g.Di + zi) * g.Hi + yi) * g.Wi + xi) * C + c);
                    const float xv[4] = {t4.x, t4.y, t4.z, t4.w};
                    const int tap = (a * g.k + b) * g.k + cc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = fmaxf(fmaf((xv[q] - m[q]) * iv[q], gg[q], bb[q]), 0.f);
                        const float tn = xv[q] != xv[q] ? xv[q] : t;          // a NaN input stays a NaN
                        if (first || tn > best[q] || tn != tn) { best[q] = tn; bi[q] = tap; }
                    }
                    first = false;
                }
            }
        }
        st4(y + 4 * i, make_float4(best[0], best[1], best[2], best[3]));
        if (arg) *reinterpret_cast<uchar4*>(arg + 4 * i) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

// dx = gamma*invstd*(dy' - sum_dy/count - xhat*sum_dyxhat/count) with dy' = dy * [xhat*gamma + beta > 0]: the backward of
// relu(bn(x)) when relu(bn(x)) was not stored
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_x_kernel(const float* dy, const float* x, float* dx, long n4, int C,
                                                                 const float* save, const float* gamma, const float* beta,
                                                                 const double* sums, double count, float* dgamma,
                                                                 float* dbeta) {
    const int CV = C >> 2;
    const int c = (int)(threadIdx.x % CV) * 4;
    const float rc = (float)(1.0 / count);
    float mean[4], inv[4], gi[4], gg[4], bb[4], sdy[4], sdx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mean[k] = save[c + k]; inv[k] = save[C + c + k];
        gg[k] = gamma ? gamma[c + k] : 1.f; bb[k] = beta ? beta[c + k] : 0.f;
        gi[k] = gg[k] * inv[k];
        sdy[k] = (float)sums[c + k] * rc; sdx[k] = (float)sums[C + c + k] * rc;
    }
    if (blockIdx.x == 0 && threadIdx.x < CV) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dbeta) dbeta[c + k] = (float)sums[c + k];
            if (dgamma) dgamma[c + k] = (float)sums[C + c + k];
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 d4 = ld4(dy + 4 * i), x4 = ld4(x + 4 * i);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mean[k]) * inv[k];
            const float dm = fmaf(xh, gg[k], bb[k]) > 0.f ? d[k] : 0.f;
            o[k] = gi[k] * (dm - sdy[k] - xh * sdx[k]);
        }
        st4(dx + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---- small-M BatchNorm in ONE launch (the BatchNorm1d layers of the projection MLP see M = batch rows, the
// feature_3d BatchNorm M = batch * 8): a workgroup owns 4 channels (one float4 column), its 256 threads are row lanes;
// sums in fp64 through a fixed-shape shuffle + LDS tree (deterministic), then the apply pass over the same rows (they
// are still in L1/L2).  Replaces colreduce + finalize + apply (3 launches).
constexpr int BNS_CH = 4;             // channels per workgroup

__device__ __forceinline__ void bns_block_sum8(double (&v)[8], double (*red)[8]) {
    // v[0..7] summed over the 256 threads of the workgroup; result in every thread
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* x, float* y, int M, int C, const float* gamma,
                                                          const float* beta, float eps, float momentum,
                                                          float* running_mean, float* running_var,
                                                          long long* num_batches_tracked, float* save,
                                                          const float* res, int relu, float* pooled, int V) {
    const int c = blockIdx.x * BNS_CH;
    __shared__ double red[4][8];
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (int r = threadIdx.x; r < M; r += 256) {
        const float4 v = ld4(x + (long)r * C + c);
        const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += (double)a[k]; s[4 + k] += (double)a[k] * (double)a[k]; }
    }
    bns_block_sum8(s, red);
    float m[4], iv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mean = s[k] / M;
        double var = s[4 + k] / M - mean * mean;
        if (var < 0) var = 0;
        m[k] = (float)mean; iv[k] = (float)(1.0 / sqrt(var + (double)eps));
        if (threadIdx.x == 0) {
            if (save) { save[c + k] = m[k]; save[C + c + k] = iv[k]; }
            if (running_mean) {
                const double unbiased = M > 1 ? var * M / (M - 1.0) : var;
                running_mean[c + k] = (1.f - momentum) * running_mean[c + k] + momentum * m[k];
                running_var[c + k] = (1.f - momentum) * running_var[c + k] + momentum * (float)unbiased;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    const float4 g = gamma ? ld4(gamma + c) : make_float4(1, 1, 1, 1);
    const float4 b = beta ? ld4(beta + c) : make_float4(0, 0, 0, 0);
    if (pooled) {
        // global average pool of the output over the V consecutive rows of a sample (V a power of two <= 64, M % V == 0):
        // the first lane of each group of V sums its group's values in row order - the order of avgpool_fwd_kernel
        const int lane = threadIdx.x & 63;
        for (int r0 = 0; r0 < M; r0 += 256) {
            const int r = r0 + threadIdx.x;
            const bool ok = r < M;
            const long o = (long)(ok ? r : 0) * C + c;
            const float4 v = ld4(x + o);
            float q[4];
            q[0] = fmaf((v.x - m[0]) * iv[0], g.x, b.x); q[1] = fmaf((v.y - m[1]) * iv[1], g.y, b.y);
            q[2] = fmaf((v.z - m[2]) * iv[2], g.z, b.z); q[3] = fmaf((v.w - m[3]) * iv[3], g.w, b.w);
            if (relu) { q[0] = fmaxf(q[0], 0.f); q[1] = fmaxf(q[1], 0.f); q[2] = fmaxf(q[2], 0.f); q[3] = fmaxf(q[3], 0.f); }
            if (ok) st4(y + o, make_float4(q[0], q[1], q[2], q[3]));
            float acc[4] = {q[0], q[1], q[2], q[3]};
            for (int k = 1; k < V; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += __shfl(q[e], (lane & ~(V - 1)) + k, 64);
            if (ok && (lane & (V - 1)) == 0) {
                const float rs = (float)V;
                st4(pooled + (long)(r / V) * C + c, make_float4(acc[0] / rs, acc[1] / rs, acc[2] / rs, acc[3] / rs));
            }
        }
        return;
    }
#pragma unroll 4
    for (int r = threadIdx.x; r < M; r += 256) {
        const long o = (long)r * C + c;
        const float4 v = ld4(x + o);
        float4 q;
        q.x = fmaf((v.x - m[0]) * iv[0], g.x, b.x); q.y = fmaf((v.y - m[1]) * iv[1], g.y, b.y);
        q.z = fmaf((v.z - m[2]) * iv[2], g.z, b.z); q.w = fmaf((v.w - m[3]) * iv[3], g.w, b.w);
        if (res) { const float4 t = ld4(res + o); q.x += t.x; q.y += t.y; q.z += t.z; q.w += t.w; }
        if (relu) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
        st4(y + o, q);
    }
}

// backward in one launch: sums of dy' and dy'*xhat per channel, then dx, dgamma, dbeta
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* dy, const float* x, const float* y, float* dx,
                                                          int M, int C, const float* save, const float* gamma,
                                                          int relu, float* dgamma, float* dbeta, int V) {
    // V > 0: dy is the gradient of the pooled output (M / V rows); the gradient of row r is dy[r / V] / V (avgpool_bwd_kernel)
    const int c = blockIdx.x * BNS_CH;
    __shared__ double red[4][8];
    const float4 ma = ld4(save + c), ia = ld4(save + C + c);
    const float mean[4] = {ma.x, ma.y, ma.z, ma.w}, inv[4] = {ia.x, ia.y, ia.z, ia.w};
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (int r = threadIdx.x; r < M; r += 256) {
        const long o = (long)r * C + c;
        const float4 x4 = ld4(x + o);
        float4 d4 = ld4(dy + (V > 0 ? (long)(r / V) * C + c : o));
        if (V > 0) { const float rs = (float)V; d4.x /= rs; d4.y /= rs; d4.z /= rs; d4.w /= rs; }
        float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        if (relu) {
            const float4 y4 = ld4(y + o);
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = yv[k] > 0.f ? d[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += (double)d[k]; s[4 + k] += (double)(d[k] * ((xv[k] - mean[k]) * inv[k])); }
    }
    bns_block_sum8(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dbeta) dbeta[c + k] = (float)s[k];
            if (dgamma) dgamma[c + k] = (float)s[4 + k];
        }
    }
    const float rc = 1.0f / (float)M;
    float gi[4], sdy[4], sdx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gi[k] = (gamma ? gamma[c + k] : 1.f) * inv[k];
        sdy[k] = (float)s[k] * rc; sdx[k] = (float)s[4 + k] * rc;
    }
#pragma unroll 4
    for (int r = threadIdx.x; r < M; r += 256) {
        const long o = (long)r * C + c;
        const float4 x4 = ld4(x + o);
        float4 d4 = ld4(dy + (V > 0 ? (long)(r / V) * C + c : o));
        if (V > 0) { const float rs = (float)V; d4.x /= rs; d4.y /= rs; d4.z /= rs; d4.w /= rs; }
        float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        if (relu) {
            const float4 y4 = ld4(y + o);
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = yv[k] > 0.f ? d[k] : 0.f;
        }
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = gi[k] * (d[k] - sdy[k] - (xv[k] - mean[k]) * inv[k] * sdx[k]);
        st4(dx + o, make_float4(q[0], q[1], q[2], q[3]));
    }
}

// dx = gamma * invstd * (dy' - sum_dy/count - xhat * sum_dyxhat/count); workgroup 0 also writes the affine
// gradients dbeta = sum_dy, dgamma = sum_dy*xhat when asked to (from `sums`, i.e. the sums of THIS rank's rows
// only when the caller has not all-reduced them)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, const float* x,
                                                          const float* y, float* dx, long n4, int C,
                                                          const float* save, const float* gamma,
                                                          const double* sums, double count,
                                                          int relu, float* dgamma, float* dbeta, float* dres) {
    // dres (may be null): the gradient behind the ReLU, dy * (y > 0), for a residual branch added in front of the activation -
    // written on the way instead of by a masking launch of its own in front of this one
    const int CV = C >> 2;
    const int c = (int)(threadIdx.x % CV) * 4;
    const float rc = (float)(1.0 / count);
    float mean[4], inv[4], gi[4], sdy[4], sdx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mean[k] = save[c + k]; inv[k] = save[C + c + k];
        gi[k] = (gamma ? gamma[c + k] : 1.f) * inv[k];
        sdy[k] = (float)sums[c + k] * rc; sdx[k] = (float)sums[C + c + k] * rc;
    }
    if (blockIdx.x == 0 && threadIdx.x < CV) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dbeta) dbeta[c + k] = (float)sums[c + k];
            if (dgamma) dgamma[c + k] = (float)sums[C + c + k];
        }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 d4 = ld4(dy + 4 * i), x4 = ld4(x + 4 * i);
        float d[4] = {d4.x, d4.y, d4.z, d4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
        if (relu) {
            float4 y4 = ld4(y + 4 * i);
            float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = yv[k] > 0.f ? d[k] : 0.f;
        }
        if (dres) st4(dres + 4 * i, make_float4(d[0], d[1], d[2], d[3]));
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xhat = (xv[k] - mean[k]) * inv[k];
            o[k] = gi[k] * (d[k] - sdy[k] - xhat * sdx[k]);
        }
        st4(dx + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

__global__ void bn_param_grad_kernel(const double* sums, int C, float* dgamma, float* dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] = (float)sums[c];
    if (dgamma) dgamma[c] = (float)sums[C + c];
}

__global__ void cast_sums_kernel(const double* sums, int n, float* out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) out[c] = (float)sums[c];
}

// ---------------------------------------------------------------------------------------------
// MaxPool3d (cubic window k, stride s, pad p), channels-last, argmax tap saved per element
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, float* y, uint8_t* arg,
                                                         int N, int Di, int Hi, int Wi, int C,
                                                         int Do, int Ho, int Wo, int k, int s, int pad) {
    const int CV = C >> 2;
    const long total = (long)N * Do * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int cv = (int)(i % CV);
        long v = i / CV;
        int xo = (int)(v % Wo); v /= Wo;
        int yo = (int)(v % Ho); v /= Ho;
        int zo = (int)(v % Do);
        int n = (int)(v / Do);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int a = 0; a < k; ++a) {
            int zi = zo * s - pad + a;
            if ((unsigned)zi >= (unsigned)Di) continue;
            for (int b = 0; b < k; ++b) {
                int yi = yo * s - pad + b;
                if ((unsigned)yi >= (unsigned)Hi) continue;
                for (int c = 0; c < k; ++c) {
                    int xi = xo * s - pad + c;
                    if ((unsigned)xi >= (unsigned)Wi) continue;
                    float4 t = ld4(x + ((((long)n * Di + zi) * Hi + yi) * Wi + xi) * C + 4 * cv);
                    float tv[4] = {t.x, t.y, t.z, t.w};
                    int tap = (a * k + b) * k + c;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (first || tv[q] > best[q] || tv[q] != tv[q]) { best[q] = tv[q]; bi[q] = tap; }
                    first = false;
                }
            }
        }
        st4(y + 4 * i, make_float4(best[0], best[1], best[2], best[3]));
        if (arg) *reinterpret_cast<uchar4*>(arg + 4 * i) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

// One workgroup per input row (n, zi, yi): the row's (xi, channel-quad) items are the threads, so the only per-item
// divisions are by the compile-time stride S (0 = run-time stride) - the element-wise form of this kernel spent its time
// in integer divisions (7 per item by run-time values), not in memory.
template <int S>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy, const uint8_t* arg,
                                                         float* dx, int N, int Di, int Hi, int Wi,
                                                         int C, int Do, int Ho, int Wo, int k, int s_rt,
                                                         int pad) {
    const int s = S ? S : s_rt;
    const int CV = C >> 2;
    const unsigned row = blockIdx.x;                       // wave-uniform decode
    const int yi = (int)(row % (unsigned)Hi);
    const unsigned r2 = row / (unsigned)Hi;
    const int zi = (int)(r2 % (unsigned)Di), n = (int)(r2 / (unsigned)Di);
    // the pooled windows that contain this voxel: o in [ceil((i + pad - k + 1)/s), floor((i + pad)/s)]
    const int zl = max(0, (zi + pad - k + s) / s), zh = min(Do - 1, (zi + pad) / s);
    const int yl = max(0, (yi + pad - k + s) / s), yh = min(Ho - 1, (yi + pad) / s);
    const long row_base = (((long)n * Di + zi) * Hi + yi) * Wi;
    const bool pow2 = (CV & (CV - 1)) == 0;
    const int sh = 31 - __builtin_clz((unsigned)CV);
    for (int t = threadIdx.x; t < Wi * CV; t += 256) {
        const int xi = pow2 ? t >> sh : t / CV, cv = pow2 ? t & (CV - 1) : t % CV;
        const int xl = max(0, (xi + pad - k + s) / s), xh = min(Wo - 1, (xi + pad) / s);
        float acc[4] = {0, 0, 0, 0};
        if (k <= 2 * s) {
            // at most two windows per axis (k = 3, s = 2).  The eight candidate argmax words (4 bytes each) are loaded
            // together; the gradient (16 bytes) is fetched only where a channel's argmax names this voxel
            uchar4 am[8];
            long off[8];
            int tp[8];
            const bool any = (zl <= zh) & (yl <= yh) & (xl <= xh);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int zo = zl + (q >> 2), yo = yl + ((q >> 1) & 1), xo = xl + (q & 1);
                const bool ok = any & (zo <= zh) & (yo <= yh) & (xo <= xh);
                off[q] = ok ? ((((long)n * Do + zo) * Ho + yo) * Wo + xo) * C + 4 * cv : 4 * cv;
                am[q] = *reinterpret_cast<const uchar4*>(arg + off[q]);
                tp[q] = ok ? ((zi + pad - zo * s) * k + (yi + pad - yo * s)) * k + (xi + pad - xo * s) : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool hx = am[q].x == tp[q], hy = am[q].y == tp[q], hz = am[q].z == tp[q], hw = am[q].w == tp[q];
                if (hx | hy | hz | hw) {
                    const float4 d = ld4(dy + off[q]);
                    if (hx) acc[0] += d.x;
                    if (hy) acc[1] += d.y;
                    if (hz) acc[2] += d.z;
                    if (hw) acc[3] += d.w;
                }
            }
        } else {
            for (int zo = zl; zo <= zh; ++zo)
                for (int yo = yl; yo <= yh; ++yo)
                    for (int xo = xl; xo <= xh; ++xo) {
                        const long o = ((((long)n * Do + zo) * Ho + yo) * Wo + xo) * C + 4 * cv;
                        const uchar4 am = *reinterpret_cast<const uchar4*>(arg + o);
                        const float4 d = ld4(dy + o);
                        const int tap = ((zi + pad - zo * s) * k + (yi + pad - yo * s)) * k + (xi + pad - xo * s);
                        if (am.x == tap) acc[0] += d.x;
                        if (am.y == tap) acc[1] += d.y;
                        if (am.z == tap) acc[2] += d.z;
                        if (am.w == tap) acc[3] += d.w;
                    }
        }
        st4(dx + 4 * ((row_base + xi) * CV + cv), make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool over S spatial positions: x [B][S][C] -> y [B][C]
// ---------------------------------------------------------------------------------------------
__global__ void avgpool_fwd_kernel(const float* x, float* y, int B, int S, int C) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    int c = (int)(i % C);
    long b = i / C;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += x[(b * S + k) * C + c];
    y[i] = s / (float)S;
}
__global__ void avgpool_bwd_kernel(const float* dy, float* dx, int B, int S, int C) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * C) return;
    int c = (int)(i % C);
    long b = i / ((long)S * C);
    dx[i] = dy[b * C + c] / (float)S;
}

__global__ void bias_add_kernel(float* y, const float* bias, long n, int C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] += bias[i % C];
}

// out = dy * (y > 0)  (+ add)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* dy, const float* y,
                                                       const float* add, float* out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 d = ld4(dy + 4 * i), v = ld4(y + 4 * i);
        if (add) { float4 a = ld4(add + 4 * i); d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w; }
        d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f;
        d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        st4(out + 4 * i, d);
    }
}

// ---------------------------------------------------------------------------------------------
// L2 normalise rows (F.normalize, eps 1e-12), MoCo logits, CE(label 0)
// one wave per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* inv_norm,
                                                        int B, int C) {
    int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= B) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { float v = x[(long)row * C + c]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < C; c += 64) y[(long)row * C + c] = x[(long)row * C + c] * inv;
    if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
// Rows of 16 / 32 / 64 channels (the detector's 32-channel projection head: 8.4 M rows per 128 x 512 x 512 tomogram, where a wave per
// row moved 1.3 TB/s): G = C / 4 lanes per row, a float4 each, 64 / G rows per wave.  The sum of squares is added in the order
// l2norm_fwd_kernel's butterfly adds it (elements 32, 16, 8, 4 apart across lanes, then 2 and 1 apart inside the lane): bit-identical.
template <int G>
__global__ __launch_bounds__(256) void l2norm_small_fwd_kernel(const float* x, float* y, float* inv_norm, long B) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / G;
    const int g = (int)(t % G);
    const bool ok = row < B;
    const float4 v = ok ? ld4(x + (row * G + g) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float a[4] = {fmaf(v.x, v.x, 0.f), fmaf(v.y, v.y, 0.f), fmaf(v.z, v.z, 0.f), fmaf(v.w, v.w, 0.f)};
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += __shfl_xor(a[j], o, 64);
    const float s = (a[0] + a[2]) + (a[1] + a[3]);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    if (!ok) return;
    st4(y + (row * G + g) * 4, make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv));
    if (g == 0 && inv_norm) inv_norm[row] = inv;
}
// dx = (dy - y * (y . dy)) * inv_norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* dy, const float* y,
                                                        const float* inv_norm, float* dx, int B,
                                                        int C) {
    int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= B) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(dy[(long)row * C + c], y[(long)row * C + c], s);
    s = wave_sum(s);
    float inv = inv_norm[row];
    for (int c = lane; c < C; c += 64) {
        long o = (long)row * C + c;
        dx[o] = (dy[o] - y[o] * s) * inv;
    }
}

// out = mean_b (a_b . b_b)   (SimSiam cosine term on normalised rows); one workgroup, fixed order
__global__ __launch_bounds__(256) void rowdot_mean_kernel(const float* a, const float* b, float* out,
                                                         int B, int C) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = threadIdx.x; i < (long)B * C; i += 256) s = fmaf(a[i], b[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}
// da = (g / B) * b
__global__ void scale_by_scalar_kernel(const float* b, const float* g, float mul, float* da, long n) {
    const float s = (*g) * mul;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) da[i] = s * b[i];
}
// mean over columns of the unbiased std over rows: torch.std(x, 0).mean()
__global__ __launch_bounds__(256) void column_std_mean_kernel(const float* x, float* out, int B, int C) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0, ss = 0;
        for (int r = 0; r < B; ++r) { double v = x[(long)r * C + c]; s += v; ss += v * v; }
        double mean = s / B, var = B > 1 ? (ss - B * mean * mean) / (B - 1) : 0.0;
        acc += (float)sqrt(var > 0 ? var : 0.0);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) / (float)C;
}

// logits[b][0] = (q_b . k_b)/T ; logits[b][1+j] = (q_b . queue[:,j])/T     queue is [C][R]
// grid (B, ceil(R/256)): one queue column per thread, q_b staged in LDS
__global__ __launch_bounds__(256) void moco_logits_fwd_kernel(const float* q, const float* k,
                                                             const float* queue, float* logits,
                                                             int C, int R, float invT) {
    extern __shared__ float qs[];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) qs[c] = q[(long)b * C + c];
    __syncthreads();
    float* out = logits + (long)b * (R + 1);
    const int j = blockIdx.y * 256 + tid;
    if (j < R) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) s = fmaf(qs[c], queue[(long)c * R + j], s);
        out[1 + j] = s * invT;
    }
    if (blockIdx.y == 0 && tid < 64) {
        float s = 0.f;
        for (int c = tid; c < C; c += 64) s = fmaf(qs[c], k[(long)b * C + c], s);
        s = wave_sum(s);
        if (tid == 0) out[0] = s * invT;
    }
}
// The same logits from the UN-normalised projections: every workgroup normalises its q row on the way into LDS (same
// summation pattern and same multiply as l2norm_fwd_kernel: bit-identical rows), the first workgroup of a row also
// writes q_hat / 1/|q| (the backward's inputs) and the normalised key row k_hat (the enqueue's input) - two l2norm
// launches fewer per step.  C <= 1024.
__global__ __launch_bounds__(256) void moco_logits_norm_fwd_kernel(const float* q_raw, const float* k_raw,
                                                                  const float* queue, float* logits, float* q_hat,
                                                                  float* q_inv, float* k_hat, int C, int R, float invT) {
    extern __shared__ float qs[];
    __shared__ float s_inv[2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = blockIdx.y == 0;
    if (wave < 2 && (wave == 0 || first)) {            // wave 0: |q|, wave 1 (first workgroup of the row): |k|
        const float* src = (wave == 0 ? q_raw : k_raw) + (long)b * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = src[c]; s = fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) s_inv[wave] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();
    const float qi = s_inv[0];
    for (int c = tid; c < C; c += 256) qs[c] = q_raw[(long)b * C + c] * qi;
    __syncthreads();
    float* out = logits + (long)b * (R + 1);
    const int j = blockIdx.y * 256 + tid;
    if (j < R) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) s = fmaf(qs[c], queue[(long)c * R + j], s);
        out[1 + j] = s * invT;
    }
    if (first) {
        const float ki = s_inv[1];
        for (int c = tid; c < C; c += 256) {
            q_hat[(long)b * C + c] = qs[c];
            k_hat[(long)b * C + c] = k_raw[(long)b * C + c] * ki;
        }
        if (tid == 0) q_inv[b] = qi;
        if (tid < 64) {
            float s = 0.f;
            for (int c = tid; c < C; c += 64) s = fmaf(qs[c], k_raw[(long)b * C + c] * ki, s);
            s = wave_sum(s);
            if (tid == 0) out[0] = s * invT;
        }
    }
}
// dq[b][c] = (dl[b][0]*k[b][c] + sum_j dl[b][1+j]*queue[c][j]) / T
// grid (B, ceil(C/4)): one wave per (b, c), the row of queue is read contiguously
__global__ __launch_bounds__(256) void moco_logits_bwd_kernel(const float* dlogits, const float* k,
                                                             const float* queue, float* dq, int C,
                                                             int R, float invT) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float* dl = dlogits + (long)b * (R + 1);
    const float* qr = queue + (long)c * R;
    float s = 0.f;
    for (int j = lane; j < R; j += 64) s = fmaf(dl[1 + j], qr[j], s);
    s = wave_sum(s);
    if (lane == 0) dq[(long)b * C + c] = (s + dl[0] * k[(long)b * C + c]) * invT;
}

// mean cross-entropy against label 0 and its gradient: dlogits = scale * (softmax - onehot0) / B
// one workgroup per row; row_loss[b] written, loss reduced by a second tiny kernel (fixed order)
__global__ __launch_bounds__(256) void ce0_kernel(const float* logits, float* row_loss,
                                                 float* dlogits, int B, int n, float scale) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (long)b * n;
    float m = -INFINITY;
    for (int j = tid; j < n; j += 256) m = fmaxf(m, l[j]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < n; j += 256) s += expf(l[j] - m);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    const float lse = m + logf(s);
    if (tid == 0) row_loss[b] = lse - l[0];
    if (dlogits) {
        const float g = scale / (float)B;
        for (int j = tid; j < n; j += 256) {
            float pj = expf(l[j] - lse);
            dlogits[(long)b * n + j] = g * (pj - (j == 0 ? 1.f : 0.f));
        }
    }
}
// One workgroup per row (as ce0_kernel); the workgroup that finishes LAST sums the row losses in row order and writes the
// mean - one launch, deterministic.  (A single 16-wave workgroup for all rows was measured at 29 us on the step's critical
// path against 5 us here.)  The arrival counter is the caller's (one word per stream, zero before the first call): it resets
// itself, so launches on different streams - the key branch's side stream, a second model, an eager step beside a graph
// replay - never share tickets.  A null counter falls back to one process-wide word (one call at a time per device).
__device__ unsigned g_ce0_arrivals = 0;
__global__ __launch_bounds__(256) void ce0_rows_kernel(const float* logits, float* loss, float* loss_copy, float* row_loss,
                                                      float* row_lse, int B, int n, unsigned* counter) {
    __shared__ float red[4];
    __shared__ unsigned s_ticket;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (long)b * n;
    float m = -INFINITY;
    for (int j = tid; j < n; j += 256) m = fmaxf(m, l[j]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < n; j += 256) s += expf(l[j] - m);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float lse = m + logf(red[0] + red[1] + red[2] + red[3]);
        row_lse[b] = lse;
        __hip_atomic_store(&row_loss[b], lse - l[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = atomicAdd(counter ? counter : &g_ce0_arrivals, 1u);
    }
    __syncthreads();
    if (s_ticket == (unsigned)B - 1 && tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float t = 0.f;
        for (int r = 0; r < B; ++r) t += __hip_atomic_load(&row_loss[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *loss = t / (float)B;
        if (loss_copy) *loss_copy = t / (float)B;
        __hip_atomic_store(counter ? counter : &g_ce0_arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// dlogits = g * (softmax - onehot0) / B with the upstream gradient g read on the device
__global__ __launch_bounds__(256) void ce0_bwd_kernel(const float* logits, const float* row_lse, const float* g_dev,
                                                     float* dlogits, int B, int n) {
    const int b = blockIdx.x;
    const float g = (*g_dev) / (float)B, lse = row_lse[b];
    const float* l = logits + (long)b * n;
    for (int j = threadIdx.x; j < n; j += 256) dlogits[(long)b * n + j] = g * (expf(l[j] - lse) - (j == 0 ? 1.f : 0.f));
}
__global__ void mean_kernel(const float* v, int n, float* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += v[i];
        *out = s / (float)n;
    }
}

// ---------------------------------------------------------------------------------------------
// optimiser-side passes over flat parameter arenas
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_kernel(float* k, const float* q, float m, long n) {
    const float om = 1.0f - m;
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ld4(k + 4 * i), b = ld4(q + 4 * i);
        a.x = a.x * m + b.x * om; a.y = a.y * m + b.y * om;
        a.z = a.z * m + b.z * om; a.w = a.w * m + b.w * om;
        st4(k + 4 * i, a);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) k[i] = k[i] * m + q[i] * om;
}
// p <- p - lr * (g + wd * p)   (torch.optim.SGD without momentum)
// gs: scale of the gradient (1 / world size when g holds the SUM over the data-parallel ranks: the averaging rides here
// instead of in a pass of its own over the 40 MB arena)
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, const float* lr_dev,
                                                 float lr_host, float wd, float gs, long n) {
    const float lr = lr_dev ? *lr_dev : lr_host;
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ld4(p + 4 * i), b = ld4(g + 4 * i);
        a.x -= lr * (gs * b.x + wd * a.x); a.y -= lr * (gs * b.y + wd * a.y);
        a.z -= lr * (gs * b.z + wd * a.z); a.w -= lr * (gs * b.w + wd * a.w);
        st4(p + 4 * i, a);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) p[i] -= lr * (gs * g[i] + wd * p[i]);
}

// queue[:, ptr:ptr+B] = keys.T ; ptr = (ptr + B) % R      (queue is [C][R], ptr is int64 on device)
__global__ void enqueue_kernel(float* queue, long long* ptr, const float* keys, int B, int C, int R) {
    const long long p0 = *ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * C; i += gridDim.x * blockDim.x) {
        int b = i % B, c = i / B;
        long long col = p0 + b;
        if (col < R) queue[(long)c * R + col] = keys[(long)b * C + c];
    }
}
__global__ void advance_ptr_kernel(long long* ptr, int B, int R) { *ptr = (*ptr + B) % R; }
// the same in one launch for the usual key batches (B * C <= 64 Ki elements): one workgroup copies, then moves the pointer
__global__ __launch_bounds__(1024) void enqueue_small_kernel(float* queue, long long* ptr, const float* keys, int B, int C, int R) {
    const long long p0 = *ptr;
    for (int i = threadIdx.x; i < B * C; i += 1024) {
        int b = i % B, c = i / B;
        long long col = p0 + b;
        if (col < R) queue[(long)c * R + col] = keys[(long)b * C + c];
    }
    __syncthreads();
    if (threadIdx.x == 0) *ptr = (p0 + B) % R;
}

int ew_blocks(long n) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, 2048)); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" size_t mi_colreduce_workspace_bytes(long M, int C) {
    if (!colreduce_ok(C)) return 0;
    return sizeof(double) * 2 * C * (size_t)colreduce_blocks(M, C);
}

extern "C" int mi_bn_stats(const float* x, long M, int C, double* sums, void* ws, size_t ws_bytes,
                           mi_stream_t stream) {
    if (!x || !sums) return MI_E_ARG;
    ColReduceParams p = {};
    p.a = x; p.M = M; p.C = C;
    return run_colreduce<CR_STATS>(p, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_bn_apply_fwd(const float* x, float* y, long M, int C, const double* sums,
                               double count, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float* save_mean_invstd, const float* res,
                               int relu, mi_stream_t stream) {
    if (!x || !y || !sums || !save_mean_invstd || !colreduce_ok(C) || M <= 0 || !(count > 0)) return MI_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, y, n4, C, sums, count, eps,
                       momentum, running_mean, running_var, num_batches_tracked, save_mean_invstd, gamma,
                       beta, res, relu);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

static bool pool_geom(int N, int Di, int Hi, int Wi, int k, int stride, int pad, PoolGeom* g) {
    if (N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || k <= 0 || k > 6 || stride <= 0 || pad < 0) return false;
    const int Do = (Di + 2 * pad - k) / stride + 1, Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return false;
    *g = PoolGeom{N, Di, Hi, Wi, Do, Ho, Wo, k, stride, pad};
    return true;
}

/* relu(bn(x)) pooled by MaxPool3d(k, stride, pad) without materialising relu(bn(x)).  `sums`/`count` as for
 * mi_bn_apply_fwd (sums == NULL: eval mode, running statistics). */
extern "C" int mi_bn_relu_maxpool3d_fwd(const float* x, float* y, uint8_t* argmax, int N, int Di, int Hi, int Wi, int C,
                                        int k, int stride, int pad, const double* sums, double count,
                                        const float* gamma, const float* beta, float eps, float momentum,
                                        float* running_mean, float* running_var, long long* num_batches_tracked,
                                        float* save_mean_invstd, mi_stream_t stream) {
    PoolGeom g;
    if (!x || !y || !colreduce_ok(C) || !pool_geom(N, Di, Hi, Wi, k, stride, pad, &g)) return MI_E_ARG;
    if (sums ? (!save_mean_invstd || !(count > 0)) : (!running_mean || !running_var)) return MI_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MI_E_ARG;
    const long total = (long)N * g.Do * g.Ho * g.Wo * (C / 4);
    if (total >= (1l << 31)) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax,
                       g, C, sums, count, eps, momentum, running_mean, running_var, num_batches_tracked,
                       save_mean_invstd, gamma, beta);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

/* backward of relu(bn(x)) when the activation was not stored (the ReLU mask is recomputed from x): the two halves,
 * with the SyncBN all-reduce of `sums` between them.  dy = gradient behind the ReLU (here: mi_maxpool3d_bwd's output). */
extern "C" int mi_bn_relu_bwd_reduce_x(const float* dy, const float* x, long M, int C, const float* save_mean_invstd,
                                       const float* gamma, const float* beta, double* sums, void* ws, size_t ws_bytes,
                                       mi_stream_t stream) {
    if (!dy || !x || !save_mean_invstd || !sums) return MI_E_ARG;
    ColReduceParams p = {};
    p.a = dy; p.x = x; p.save = save_mean_invstd; p.M = M; p.C = C; p.gamma = gamma; p.beta = beta;
    return run_colreduce<CR_BNBWD_X>(p, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_bn_relu_bwd_apply_x(const float* dy, const float* x, float* dx, long M, int C,
                                      const float* save_mean_invstd, const float* gamma, const float* beta,
                                      const double* sums, double count, float* dgamma, float* dbeta, mi_stream_t stream) {
    if (!dy || !x || !dx || !save_mean_invstd || !sums || !colreduce_ok(C) || M <= 0 || !(count > 0)) return MI_E_ARG;
    const long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_relu_bwd_apply_x_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n4, C,
                       save_mean_invstd, gamma, beta, sums, count, dgamma, dbeta);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

/* One-launch BatchNorm for small M (<= MI_BN_SMALL_MAX_ROWS rows), single process: statistics + running
 * statistics + affine (+res, ReLU).  Same arithmetic as mi_bn_stats + mi_bn_apply_fwd. */
extern "C" int mi_bn_small_fwd(const float* x, float* y, long M, int C, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float* save_mean_invstd, const float* res, int relu,
                               mi_stream_t stream) {
    if (!x || !y || !save_mean_invstd || M <= 0 || M > MI_BN_SMALL_MAX_ROWS || C <= 0 || C % 4) return MI_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MI_E_ARG;
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3((C + BNS_CH - 1) / BNS_CH), dim3(256), 0, (hipStream_t)stream, x, y,
                       (int)M, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                       save_mean_invstd, res, relu, (float*)nullptr, 0);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

/* global_avgpool(relu(bn(x))) over the V rows of each sample in the same launch (feature_3d + avgpool of the MoCo-3D trunk,
 * models/networks/moco_encoder_3d.py:385-388): y = relu(bn(x)) is still written (the backward reads its sign), pooled is
 * (M / V, C).  V a power of two <= 64 dividing M. */
extern "C" int mi_bn_small_pool_fwd(const float* x, float* y, float* pooled, long M, int C, int V, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* save_mean_invstd, mi_stream_t stream) {
    if (!x || !y || !pooled || !save_mean_invstd || M <= 0 || M > MI_BN_SMALL_MAX_ROWS || C <= 0 || C % 4) return MI_E_ARG;
    if (V < 1 || V > 64 || (V & (V - 1)) || M % V) return MI_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MI_E_ARG;
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3((C + BNS_CH - 1) / BNS_CH), dim3(256), 0, (hipStream_t)stream, x, y,
                       (int)M, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                       save_mean_invstd, (const float*)nullptr, 1, pooled, V);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_small_bwd(const float* dy, const float* x, const float* y, float* dx, long M, int C,
                               const float* save_mean_invstd, const float* gamma, int relu, float* dgamma,
                               float* dbeta, mi_stream_t stream) {
    if (!dy || !x || !dx || !save_mean_invstd || (relu && !y) || M <= 0 || M > MI_BN_SMALL_MAX_ROWS || C <= 0 || C % 4)
        return MI_E_ARG;
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3((C + BNS_CH - 1) / BNS_CH), dim3(256), 0, (hipStream_t)stream, dy, x, y,
                       dx, (int)M, C, save_mean_invstd, gamma, relu, dgamma, dbeta, 0);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_small_pool_bwd(const float* dpooled, const float* x, const float* y, float* dx, long M, int C, int V,
                                    const float* save_mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                                    mi_stream_t stream) {
    if (!dpooled || !x || !y || !dx || !save_mean_invstd || M <= 0 || M > MI_BN_SMALL_MAX_ROWS || C <= 0 || C % 4) return MI_E_ARG;
    if (V < 1 || V > 64 || (V & (V - 1)) || M % V) return MI_E_ARG;
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3((C + BNS_CH - 1) / BNS_CH), dim3(256), 0, (hipStream_t)stream, dpooled, x, y,
                       dx, (int)M, C, save_mean_invstd, gamma, 1, dgamma, dbeta, V);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_eval_fwd(const float* x, float* y, long M, int C, const float* running_mean,
                              const float* running_var, const float* gamma, const float* beta,
                              float eps, float* save_mean_invstd, const float* res, int relu,
                              mi_stream_t stream) {
    if (!x || !y || !running_mean || !running_var || !colreduce_ok(C) || M <= 0) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, y, n4, C,
                       (const double*)nullptr, 1.0, eps, 0.f, (float*)running_mean, (float*)running_var,
                       (long long*)nullptr, save_mean_invstd, gamma, beta, res, relu);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_bwd_reduce(const float* dy, const float* x, const float* y, long M, int C,
                                const float* save_mean_invstd, int relu, double* sums, void* ws,
                                size_t ws_bytes, mi_stream_t stream) {
    if (!dy || !x || !save_mean_invstd || !sums || (relu && !y)) return MI_E_ARG;
    ColReduceParams p = {};
    p.a = dy; p.x = x; p.y = y; p.save = save_mean_invstd; p.M = M; p.C = C; p.relu = relu;
    return run_colreduce<CR_BNBWD>(p, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_bn_bwd_apply(const float* dy, const float* x, const float* y, float* dx, long M,
                               int C, const float* save_mean_invstd, const float* gamma,
                               const double* sums, double count, int relu, float* dgamma,
                               float* dbeta, mi_stream_t stream) {
    if (!dy || !x || !dx || !save_mean_invstd || !sums || (relu && !y) || !colreduce_ok(C) || !(count > 0)) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, y, dx, n4, C,
                       save_mean_invstd, gamma, sums, count, relu, dgamma, dbeta, (float*)nullptr);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_bwd_apply_res(const float* dy, const float* x, const float* y, float* dx, float* dres, long M,
                                   int C, const float* save_mean_invstd, const float* gamma,
                                   const double* sums, double count, float* dgamma, float* dbeta, mi_stream_t stream) {
    if (!dy || !x || !y || !dx || !dres || !save_mean_invstd || !sums || !colreduce_ok(C) || !(count > 0)) return MI_E_ARG;
    long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, dy, x, y, dx, n4, C,
                       save_mean_invstd, gamma, sums, count, 1, dgamma, dbeta, dres);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, mi_stream_t stream) {
    if (!sums || C <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, dgamma, dbeta);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_colsum(const float* dy, long M, int C, float* out, double* sums_scratch, void* ws,
                         size_t ws_bytes, mi_stream_t stream) {
    if (!dy || !out || !sums_scratch) return MI_E_ARG;
    ColReduceParams p = {};
    p.a = dy; p.M = M; p.C = C;
    return run_colreduce<CR_SUM>(p, sums_scratch, ws, ws_bytes, (hipStream_t)stream, out, C);
}

extern "C" int mi_maxpool3d_fwd(const float* x, float* y, uint8_t* argmax, int N, int Di, int Hi,
                                int Wi, int C, int k, int stride, int pad, mi_stream_t stream) {
    if (!x || !y || C % 4 || k <= 0 || k > 6 || stride <= 0 || pad < 0) return MI_E_ARG;
    int Do = (Di + 2 * pad - k) / stride + 1, Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return MI_E_ARG;
    long total = (long)N * Do * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y,
                       argmax, N, Di, Hi, Wi, C, Do, Ho, Wo, k, stride, pad);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// Round 4: the stem's pool (k = 3, stride 2, padding 1) for planes of Wi * C / 4 = 256 vectors.  maxpool_bwd_kernel<2> spends ~500
// instructions per wave on ONE output vector per thread (281 scalar + 206 vector: row decode with runtime divisors, 64-bit offsets
// and tap indices of eight candidates) - 64 waves per SIMD x 2,100 issue cycles = the 55 us it takes, whatever its memory traffic
// (profiles/r04_experiments.txt item 24).  Here a workgroup owns an input plane (n, zi), stages the one or two pooled planes that
// contain it (argmax bytes + gradients) in LDS, and a thread keeps its (x, channel vector) column and walks y: which pooled windows
// contain a coordinate, and at which tap, depends only on the coordinate's PARITY (even: one window, tap 1; odd: two, taps 2 and 0),
// so the candidates are fixed per thread (x), per workgroup (z) and per unrolled half-iteration (y), a hit test is one XOR and a
// has-zero-byte test on the argmax word, and the gradient vector is read only behind a hit.  Candidates are visited in (zo, yo, xo)
// order as in the generic kernel: the sums are bit-identical.
// (blockIdx.y: the plane is cut into gridDim.y bands of rows; a band stages only the pooled rows it needs: smaller workgroups, more of
// them per CU and out of phase with each other - a workgroup's staging, candidate work and stores do not overlap by themselves)
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_kernel(const float* dy, const uint8_t* arg, float* dx, int Di, int Hi, int Wi,
                                                              int C, int Do, int Ho, int Wo, int band) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pool_lds[];
    const int yb = blockIdx.y * band, ye = min(Hi, yb + band);       // this workgroup's rows
    const int yo_lo = yb >> 1, yo_hi = min(Ho - 1, ye >> 1);           // pooled rows they can touch ((ye - 1 + 1) / 2)
    const int Hb = band / 2 + 1;                         // pooled rows staged per plane (>= yo_hi - yo_lo + 1)
    const int plane_o = Ho * Wo * C;                     // elements of one pooled plane
    const int plane_b = Hb * Wo * C;                     // ... of the staged band of it
    float* s_dy = reinterpret_cast<float*>(pool_lds);    // [2][Hb][Wo][C]
    uint8_t* s_arg = pool_lds + 2 * (size_t)plane_b * sizeof(float);      // [2][Hb][Wo][C]
    const int zi = blockIdx.x % Di, n = blockIdx.x / Di;
    // windows along z: zi even -> zo = zi / 2 at tap 1; odd -> zo = (zi - 1) / 2 at tap 2 and (zi + 1) / 2 at tap 0 (if inside)
    const int z0 = zi >> 1;
    const int nz = (zi & 1) ? ((z0 + 1 < Do) ? 2 : 1) : 1;
    const int rows_b = (yo_hi - yo_lo + 1) * Wo * C / 4;              // vectors of the band that exist
    for (int q = threadIdx.x; q < nz * rows_b; q += 256) {
        const int zq = q / rows_b, r = q % rows_b;
        const long src = ((long)n * Do + z0 + zq) * plane_o + (long)yo_lo * Wo * C + 4 * r;
        *reinterpret_cast<float4*>(s_dy + zq * plane_b + 4 * r) = ld4(dy + src);
        *reinterpret_cast<unsigned*>(s_arg + zq * plane_b + 4 * r) = *reinterpret_cast<const unsigned*>(arg + src);
    }
    __syncthreads();
    const int CV = C >> 2;
    const int cv = threadIdx.x % CV, xi = threadIdx.x / CV;          // (Wi * CV == 256: host check)
    // the thread's x candidates: byte offset of (xo, channels 4 cv ..) inside a pooled row, and the x tap
    const int x0 = xi >> 1;
    const int nx = (xi & 1) ? ((x0 + 1 < Wo) ? 2 : 1) : 1;
    const int xoff0 = x0 * C + 4 * cv, xoff1 = xoff0 + C;
    const int tx0 = (xi & 1) ? 2 : 1;                    // (tap of the second candidate: 0)
    float* out = dx + 4 * ((((long)n * Di + zi) * Hi * Wi + xi) * CV + cv);
    const long ystride = 4l * Wi * CV;
    for (int yi = yb; yi < ye; ++yi) {
        const int y0 = yi >> 1;
        const int ny = (yi & 1) ? ((y0 + 1 < Ho) ? 2 : 1) : 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nz; ++a) {
            const int tz = (zi & 1) ? (a ? 0 : 2) : 1;
            for (int b = 0; b < ny; ++b) {
                const int ty = (yi & 1) ? (b ? 0 : 2) : 1;
                const int rowb = (a * Hb + y0 + b - yo_lo) * Wo * C;  // byte offset of the pooled row in s_arg (x 4 in s_dy)
                const int tzy = (tz * 3 + ty) * 3;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c >= nx) break;
                    const int o = rowb + (c ? xoff1 : xoff0);
                    const unsigned tp = (unsigned)(tzy + (c ? 0 : tx0));
                    const unsigned x = *reinterpret_cast<const unsigned*>(s_arg + o) ^ (tp * 0x01010101u);     // zero byte = hit
                    if (((x - 0x01010101u) & ~x & 0x80808080u) != 0u) {
                        const float4 d = *reinterpret_cast<const float4*>(s_dy + o);
                        if ((x & 0x000000ffu) == 0u) acc[0] += d.x;
                        if ((x & 0x0000ff00u) == 0u) acc[1] += d.y;
                        if ((x & 0x00ff0000u) == 0u) acc[2] += d.z;
                        if ((x & 0xff000000u) == 0u) acc[3] += d.w;
                    }
                }
            }
        }
        st4(out + yi * ystride, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// (Round 5 built the stem's backward chain WITHOUT the dense pool gradient - both BatchNorm halves gathering the pooled gradient
// themselves: 8 us faster alone, 30 us slower inside the captured step, profiles/r05_experiments.txt item 6; removed in round 6.)
extern "C" int mi_maxpool3d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int Di,
                                int Hi, int Wi, int C, int k, int stride, int pad, mi_stream_t stream) {
    if (!dy || !argmax || !dx || C % 4 || k <= 0 || k > 6 || stride <= 0 || pad < 0) return MI_E_ARG;
    int Do = (Di + 2 * pad - k) / stride + 1, Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
    const long rows = (long)N * Di * Hi;
    if (rows >= (1l << 31)) return MI_E_UNSUPPORTED;
    // bands of rows per plane: 4 rows each by default (MI_MAXPOOL_BWD_BAND: tuning), the staged pooled rows of a band in LDS
    int band = 4;
    if (const char* v = getenv("MI_MAXPOOL_BWD_BAND")) { const int bv = atoi(v); if (bv >= 2 && bv % 2 == 0) band = bv; }
    if (band > Hi) band = (Hi + 1) & ~1;
    const size_t lds_k3s2 = 2 * (size_t)(band / 2 + 1) * Wo * C * 5;          // two pooled bands: gradients + argmax bytes
    if (stride == 2 && k == 3 && pad == 1 && Wi * (C / 4) == 256 && lds_k3s2 <= 64 * 1024 && (long)N * Di < (1l << 31) &&
        !getenv("MI_MAXPOOL_BWD_GENERIC"))
        hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel, dim3((unsigned)(N * Di), (unsigned)((Hi + band - 1) / band)), dim3(256), lds_k3s2,
                           (hipStream_t)stream, dy, argmax, dx, Di, Hi, Wi, C, Do, Ho, Wo, band);
    else if (stride == 2)
        hipLaunchKernelGGL(maxpool_bwd_kernel<2>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, N,
                           Di, Hi, Wi, C, Do, Ho, Wo, k, stride, pad);
    else if (stride == 1)
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, N,
                           Di, Hi, Wi, C, Do, Ho, Wo, k, stride, pad);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<0>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, N,
                           Di, Hi, Wi, C, Do, Ho, Wo, k, stride, pad);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_avgpool_fwd(const float* x, float* y, int B, int S, int C, mi_stream_t stream) {
    if (!x || !y || B <= 0 || S <= 0 || C <= 0) return MI_E_ARG;
    long n = (long)B * C;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, S, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_avgpool_bwd(const float* dy, float* dx, int B, int S, int C, mi_stream_t stream) {
    if (!dy || !dx || B <= 0 || S <= 0 || C <= 0) return MI_E_ARG;
    long n = (long)B * S * C;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, S, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// the step's two input batches into the engine's static buffers in ONE launch (two copy_ launches were 6.5 + 6.2 us and a 5 us gap in
// front of every graph replay): blockIdx.y = pair
__global__ __launch_bounds__(256) void copy_pair_kernel(float* d0, const float* s0, float* d1, const float* s1, long n4) {
    float* d = blockIdx.y ? d1 : d0;
    const float* s = blockIdx.y ? s1 : s0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st4(d + 4 * i, ld4(s + 4 * i));
}
extern "C" int mi_copy_pair_f32(float* dst0, const float* src0, float* dst1, const float* src1, long n, mi_stream_t stream) {
    if (!dst0 || !src0 || !dst1 || !src1 || n <= 0 || (n & 3)) return MI_E_ARG;
    if (((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) & 15) return MI_E_ARG;
    const long n4 = n / 4;
    hipLaunchKernelGGL(copy_pair_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048), 2), dim3(256), 0, (hipStream_t)stream,
                       dst0, src0, dst1, src1, n4);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_bias_add(float* y, const float* bias, long M, int C, mi_stream_t stream) {
    if (!y || !bias || M <= 0 || C <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(bias_add_kernel, dim3(ew_blocks(M * C)), dim3(256), 0, (hipStream_t)stream, y, bias, M * C, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_relu_mask(const float* dy, const float* y, const float* add, float* out, long n,
                            mi_stream_t stream) {
    if (!dy || !y || !out || n <= 0 || (n & 3)) return MI_E_ARG;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, add, out, n / 4);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int C, mi_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0) return MI_E_ARG;
    if ((C == 16 || C == 32 || C == 64) && B >= 4096 && !getenv("MI_L2NORM_GENERIC")) {
        const unsigned blocks = (unsigned)(((long)B * (C / 4) + 255) / 256);
        if (C == 16) hipLaunchKernelGGL(l2norm_small_fwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, (long)B);
        else if (C == 32) hipLaunchKernelGGL(l2norm_small_fwd_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, (long)B);
        else hipLaunchKernelGGL(l2norm_small_fwd_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, (long)B);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, B, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B,
                             int C, mi_stream_t stream) {
    if (!dy || !y || !inv_norm || !dx || B <= 0 || C <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx, B, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_rowdot_mean_fwd(const float* a, const float* b, float* out, int B, int C, mi_stream_t stream) {
    if (!a || !b || !out || B <= 0 || C <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(rowdot_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, out, B, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_rowdot_mean_bwd(const float* b, const float* grad_out, float* da, int B, int C, mi_stream_t stream) {
    if (!b || !grad_out || !da || B <= 0 || C <= 0) return MI_E_ARG;
    long n = (long)B * C;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, b, grad_out,
                       1.0f / (float)B, da, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_column_std_mean(const float* x, float* out, int B, int C, mi_stream_t stream) {
    if (!x || !out || B <= 0 || C <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(column_std_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, B, C);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_moco_logits_fwd(const float* q, const float* k, const float* queue, float* logits,
                                  int B, int C, int R, float T, mi_stream_t stream) {
    if (!q || !k || !queue || !logits || B <= 0 || C <= 0 || R <= 0 || !(T > 0.f) || C > 8192) return MI_E_ARG;
    hipLaunchKernelGGL(moco_logits_fwd_kernel, dim3(B, (R + 255) / 256), dim3(256), sizeof(float) * C,
                       (hipStream_t)stream, q, k, queue, logits, C, R, 1.0f / T);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_moco_logits_norm_fwd(const float* q_raw, const float* k_raw, const float* queue, float* logits,
                                       float* q_hat, float* q_inv, float* k_hat, int B, int C, int R, float T,
                                       mi_stream_t stream) {
    if (!q_raw || !k_raw || !queue || !logits || !q_hat || !q_inv || !k_hat || B <= 0 || C <= 0 || R <= 0 || !(T > 0.f) ||
        C > 1024)
        return MI_E_ARG;
    hipLaunchKernelGGL(moco_logits_norm_fwd_kernel, dim3(B, (R + 255) / 256), dim3(256), sizeof(float) * C,
                       (hipStream_t)stream, q_raw, k_raw, queue, logits, q_hat, q_inv, k_hat, C, R, 1.0f / T);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_moco_logits_bwd(const float* dlogits, const float* k, const float* queue, float* dq,
                                  int B, int C, int R, float T, mi_stream_t stream) {
    if (!dlogits || !k || !queue || !dq || B <= 0 || C <= 0 || R <= 0 || !(T > 0.f)) return MI_E_ARG;
    hipLaunchKernelGGL(moco_logits_bwd_kernel, dim3(B, (C + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       dlogits, k, queue, dq, C, R, 1.0f / T);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_ce_label0(const float* logits, float* loss, float* row_loss, float* dlogits, int B,
                            int n, float grad_scale, mi_stream_t stream) {
    if (!logits || !loss || !row_loss || B <= 0 || n <= 0) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce0_kernel, dim3(B), dim3(256), 0, s, logits, row_loss, dlogits, B, n, grad_scale);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, s, (const float*)row_loss, B, loss);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_ce_label0_fwd(const float* logits, float* loss, float* loss_copy, float* row_loss, float* row_lse, int B,
                                int n, unsigned* counter, mi_stream_t stream) {
    if (!logits || !loss || !row_loss || !row_lse || B <= 0 || n <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(ce0_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, loss, loss_copy, row_loss, row_lse,
                       B, n, counter);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_ce_label0_bwd(const float* logits, const float* row_lse, const float* grad_loss, float* dlogits, int B,
                                int n, mi_stream_t stream) {
    if (!logits || !row_lse || !grad_loss || !dlogits || B <= 0 || n <= 0) return MI_E_ARG;
    hipLaunchKernelGGL(ce0_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, row_lse, grad_loss, dlogits, B, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_ema_update(float* k, const float* q, float m, long n, mi_stream_t stream) {
    if (!k || !q || n <= 0 || ((uintptr_t)k & 15) || ((uintptr_t)q & 15)) return MI_E_ARG;
    hipLaunchKernelGGL(ema_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, k, q, m, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_sgd_step(float* p, const float* g, const float* lr_dev, float lr, float weight_decay,
                           float grad_scale, long n, mi_stream_t stream) {
    if (!p || !g || n <= 0 || ((uintptr_t)p & 15) || ((uintptr_t)g & 15)) return MI_E_ARG;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, lr_dev, lr, weight_decay,
                       grad_scale, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// Round 6 (SimSiamStepEngine): the same step for a model whose two views wrote their parameter gradients into two arenas (the
// second view's contribution lands in g2 instead of being added to g by a launch per parameter): p <- p - lr (gs (g + g2) + wd p)
__global__ __launch_bounds__(256) void sgd2_kernel(float* p, const float* g, const float* g2, const float* lr_dev, float lr_host, float wd,
                                                  float gs, long n) {
    const float lr = lr_dev ? *lr_dev : lr_host;
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ld4(p + 4 * i), b = ld4(g + 4 * i), c = ld4(g2 + 4 * i);
        b.x += c.x; b.y += c.y; b.z += c.z; b.w += c.w;
        a.x -= lr * (gs * b.x + wd * a.x); a.y -= lr * (gs * b.y + wd * a.y);
        a.z -= lr * (gs * b.z + wd * a.z); a.w -= lr * (gs * b.w + wd * a.w);
        st4(p + 4 * i, a);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) p[i] -= lr * (gs * (g[i] + g2[i]) + wd * p[i]);
}
extern "C" int mi_sgd_step2(float* p, const float* g, const float* g2, const float* lr_dev, float lr, float weight_decay,
                            float grad_scale, long n, mi_stream_t stream) {
    if (!p || !g || !g2 || n <= 0 || ((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)g2 & 15)) return MI_E_ARG;
    hipLaunchKernelGGL(sgd2_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, g2, lr_dev, lr, weight_decay,
                       grad_scale, n);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// sums[i] += *src[i], i < n <= 4: the loss meters of a training step, accumulated by one launch INSIDE the step (a graph node)
__global__ void scalar_accumulate_kernel(float* sums, const float* a, const float* b, const float* c, const float* d) {
    const float* src[4] = {a, b, c, d};
    const int i = threadIdx.x;
    if (i < 4 && src[i]) sums[i] += *src[i];
}
extern "C" int mi_scalar_accumulate(float* sums, const float* a, const float* b, const float* c, const float* d, mi_stream_t stream) {
    if (!sums) return MI_E_ARG;
    hipLaunchKernelGGL(scalar_accumulate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, a, b, c, d);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_queue_enqueue(float* queue, int64_t* queue_ptr, const float* keys, int B, int C, int R,
                                mi_stream_t stream) {
    if (!queue || !queue_ptr || !keys || B <= 0 || C <= 0 || R <= 0 || (R % B)) return MI_E_ARG;  // moco.py:47
    hipStream_t s = (hipStream_t)stream;
    if ((long)B * C <= 65536) {
        hipLaunchKernelGGL(enqueue_small_kernel, dim3(1), dim3(1024), 0, s, queue, (long long*)queue_ptr, keys, B, C, R);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    hipLaunchKernelGGL(enqueue_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, queue, (long long*)queue_ptr, keys, B, C, R);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(advance_ptr_kernel, dim3(1), dim3(1), 0, s, (long long*)queue_ptr, B, R);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
