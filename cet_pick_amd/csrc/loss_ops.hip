// Detector-training losses, reference cet_pick/models/loss.py:
//   _pu_neg_loss   :255-308   positive-unlabeled focal risk (PULoss :310-324)
//   _neg_loss      :378-411   CornerNet focal loss (FocalLoss)
//   ConsistencyLoss:701-712   mean squared error between the two views' heat-maps
//   UnbiasedConLoss:571-699   debiased contrastive regulariser over a (2N x 2N) similarity matrix
// The voxel losses are one streaming pass (fp64 partial sums, deterministic two-level tree) plus a
// one-thread finalize that also takes the data-dependent branch of the PU risk on the device, and an
// elementwise backward.  The contrastive loss never materialises the (2N)^2 matrix (2.4 GB at N = 12,288):
// a row block keeps its features in registers, walks the column tiles through LDS, forms each 32x32 tile of
// similarities on the matrix cores (v_mfma_f32_32x32x2_f32, K = feature dim) and folds exp() of it straight
// into the four row sums the loss needs (online max, flash-attention style).  The backward recomputes the
// tiles and contracts them with the features again on the matrix cores.
#include "common.h"
#include "../../include/cetpick_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------------------
// voxel losses
// ------------------------------------------------------------------------------------------------
constexpr int NQ = 8;
enum { VL_PU = 0, VL_FOCAL = 1, VL_MSE = 2 };

// q0 #pos  q1 #soft  q2 #unlabeled  q3 sum log(p)(1-p)^2 [pos]  q4 sum log(1-p) p^2 (1-g)^4 [soft]
// q5 sum log(1-p) p^2 [pos]  q6 sum log(p)(1-p)^2 g^4 [soft]  q7 sum p^2 log(1-p) [unlabeled]
// (MSE: q3 = sum (a-b)^2)
template <int MODE>
__global__ __launch_bounds__(256) void voxel_loss_partial_kernel(const float* pred, const float* gt, long n,
                                                                double* partials) {
    double q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float p = pred[i], g = gt[i];
        if (MODE == VL_MSE) { const float d = p - g; q[3] += (double)d * d; continue; }
        const bool pos = (g == 1.f), soft = (g > -1.f) && (g < 1.f), unl = (g == -1.f);
        const float lp = logf(p), l1p = logf(1.f - p);
        const float a = lp * (1.f - p) * (1.f - p);          // log(p) (1-p)^2
        const float b = l1p * p * p;                         // log(1-p) p^2
        const float w = (1.f - g) * (1.f - g) * (1.f - g) * (1.f - g), w2 = g * g * g * g;
        if (pos) { q[0] += 1.0; q[3] += a; q[5] += b; }
        if (soft) { q[1] += 1.0; q[4] += (double)(b * w); q[6] += (double)(a * w2); }
        if (unl) { q[2] += 1.0; q[7] += b; }
    }
    __shared__ double red[256];
    for (int k = 0; k < NQ; ++k) {
        red[threadIdx.x] = q[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partials[(long)blockIdx.x * NQ + k] = red[0];
        __syncthreads();
    }
}

// sums[0..7] = q, sums[8] = loss, sums[9] = 1 when the negative risk is kept (PU), sums[10] = n
__global__ __launch_bounds__(256) void voxel_loss_final_kernel(const double* partials, int n_part, int mode, long n,
                                                              double tau, double beta, double* sums, float* loss) {
    __shared__ double red[256];
    __shared__ double q[NQ];
    for (int k = 0; k < NQ; ++k) {
        double s = 0;
        for (int b = threadIdx.x; b < n_part; b += 256) s += partials[(long)b * NQ + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) q[k] = red[0];
        __syncthreads();
    }
    if (threadIdx.x) return;
    double L = 0, keep = 1;
    if (mode == VL_MSE) {
        L = q[3] / (double)n;
    } else if (mode == VL_FOCAL) {              // loss.py:403-409
        L = q[0] == 0 ? -q[4] : -(q[3] + q[4]) / q[0];
    } else {                                    // loss.py:283-308
        const double np = q[0], ns = q[1], nu = q[2];
        double pos_tot = -q[3] / np, negpos_tot = -q[5] / np;
        if (ns > 0) { pos_tot -= q[4] / ns; negpos_tot -= q[6] / ns; }
        const double pos_risk = pos_tot * tau;
        const double neg_total = -tau * negpos_tot + (-q[7]) / nu;
        keep = (neg_total < -beta) ? 0.0 : 1.0;
        L = keep != 0.0 ? pos_risk + neg_total : pos_risk;
    }
    for (int k = 0; k < NQ; ++k) sums[k] = q[k];
    sums[8] = L; sums[9] = keep; sums[10] = (double)n;
    *loss = (float)L;
}

// d loss / d pred (and, for MSE, d loss / d gt = -that)
template <int MODE>
__global__ __launch_bounds__(256) void voxel_loss_bwd_kernel(const float* pred, const float* gt, long n, double tau,
                                                            const double* sums, const float* dloss, float* dpred,
                                                            float* dgt) {
    const float up = *dloss;
    const double np = sums[0], ns = sums[1], nu = sums[2], keep = sums[9];
    // loss = c3 q3 + c4 q4 + c5 q5 + c6 q6 + c7 q7
    float c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    if (MODE == VL_FOCAL) {
        if (np == 0) c4 = -1.f; else { c3 = (float)(-1.0 / np); c4 = c3; }
    } else if (MODE == VL_PU) {
        c3 = (float)(-tau / np);
        if (ns > 0) c4 = (float)(-tau / ns);
        if (keep != 0.0) {
            c5 = (float)(tau / np);
            if (ns > 0) c6 = (float)(tau / ns);
            c7 = (float)(-1.0 / nu);
        }
    }
    const float mse_c = (float)(2.0 / (double)n);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float p = pred[i], g = gt[i];
        if (MODE == VL_MSE) {
            const float d = up * mse_c * (p - g);
            dpred[i] = d;
            if (dgt) dgt[i] = -d;
            continue;
        }
        const bool pos = (g == 1.f), soft = (g > -1.f) && (g < 1.f), unl = (g == -1.f);
        const float lp = logf(p), l1p = logf(1.f - p);
        const float da = (1.f - p) * (1.f - p) / p - 2.f * (1.f - p) * lp;        // d/dp log(p)(1-p)^2
        const float db = -p * p / (1.f - p) + 2.f * p * l1p;                      // d/dp log(1-p) p^2
        const float w = (1.f - g) * (1.f - g) * (1.f - g) * (1.f - g), w2 = g * g * g * g;
        float d = 0.f;
        if (pos) d += c3 * da + c5 * db;
        if (soft) d += c4 * db * w + c6 * da * w2;
        if (unl) d += c7 * db;
        dpred[i] = up * d;
    }
}

int vl_blocks(long n) { return (int)std::max<long>(1, std::min<long>((n + 256 * 8 - 1) / (256 * 8), 1024)); }

// ------------------------------------------------------------------------------------------------
// debiased contrastive loss: row sums of E = exp((S - rowmax) * (1 - I)),  S = F F^T / T
// ------------------------------------------------------------------------------------------------
constexpr int UB = 64;            // rows per workgroup and columns per tile (2 x 2 waves of 32 x 32)
constexpr int MAXDIM = 64;

// per-lane online state of one row
struct RowAcc { float m, sa, sp, so, spair; };

template <int DIM>
__global__ __launch_bounds__(256) void ucl_fwd_kernel(const float* feat, const uint8_t* cls, int n2, int n_half,
                                                     float inv_T, float* rowmax, float* s_all, float* s_pos,
                                                     float* s_other, float* e_pair) {
    constexpr int LD = DIM + 1;                           // conflict-free fragment reads
    __shared__ float colf[UB * LD];
    __shared__ uint8_t colc[UB];
    __shared__ float mrg[2][UB][5];                       // merge of the two column halves (wn)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l32 = lane & 31;
    const int row0 = blockIdx.x * UB;

    // A fragments: this wave's 32 rows, constant over the column walk
    float af[DIM / 2];
    {
        const int r = row0 + wm * 32 + l32;
#pragma unroll
        for (int t = 0; t < DIM / 2; ++t) af[t] = r < n2 ? feat[(long)r * DIM + 2 * t + h] : 0.f;
    }
    RowAcc st[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = {-INFINITY, 0.f, 0.f, 0.f, -INFINITY};

    for (int col0 = 0; col0 < n2; col0 += UB) {
        __syncthreads();
        for (int q = tid; q < UB * (DIM / 4); q += 256) {
            const int c = q / (DIM / 4), k4 = (q % (DIM / 4)) * 4;
            float4 v = make_float4(0, 0, 0, 0);
            if (col0 + c < n2) v = ld4(feat + (long)(col0 + c) * DIM + k4);
            float* d = colf + c * LD + k4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (tid < UB) colc[tid] = col0 + tid < n2 ? cls[col0 + tid] : 0;
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bsrc = colf + (wn * 32 + l32) * LD + h;
#pragma unroll
        for (int t = 0; t < DIM / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bsrc[2 * t], acc, 0, 0, 0);
        const int col = col0 + wn * 32 + l32;
        const bool colok = col < n2;
        const uint8_t cc = colc[wn * 32 + l32];
        const float fp = (cc & 1) ? 1.f : 0.f, fo = (cc & 2) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!colok) continue;
            const float s = acc[r] * inv_T;
            RowAcc& a = st[r];
            if (s > a.m) {                                // rescale the running sums to the new maximum
                const float sc = __expf(a.m - s);          // exp(-inf) = 0 on the first element
                a.sa *= sc; a.sp *= sc; a.so *= sc; a.m = s;
            }
            if (col != row) {                             // the diagonal only takes part in the maximum
                const float e = __expf(s - a.m);
                a.sa += e; a.sp += e * fp; a.so += e * fo;
            }
            const int pr = row < n_half ? row + n_half : row - n_half;
            if (col == pr) a.spair = s;
        }
    }
    // merge the 32 lanes that share a row (same h), then the two column halves
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        RowAcc a = st[r];
        float m = a.m, sp_raw = a.spair;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); sp_raw = fmaxf(sp_raw, __shfl_xor(sp_raw, o, 64)); }
        const float sc = __expf(a.m - m);
        float sa = a.sa * sc, sp = a.sp * sc, so = a.so * sc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sp += __shfl_xor(sp, o, 64); so += __shfl_xor(so, o, 64); }
        if (l32 == 0) {
            const int tr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            mrg[wn][tr][0] = m; mrg[wn][tr][1] = sa; mrg[wn][tr][2] = sp; mrg[wn][tr][3] = so; mrg[wn][tr][4] = sp_raw;
        }
    }
    __syncthreads();
    if (tid < UB && row0 + tid < n2) {
        const float m0 = mrg[0][tid][0], m1 = mrg[1][tid][0], m = fmaxf(m0, m1);
        const float c0 = __expf(m0 - m), c1 = __expf(m1 - m);
        const int row = row0 + tid;
        rowmax[row] = m;
        // the masked diagonal contributes exp(0) = 1 to every column sum it belongs to (loss.py:622-624)
        const uint8_t rc = cls[row];
        s_all[row] = mrg[0][tid][1] * c0 + mrg[1][tid][1] * c1 + 1.f;
        s_pos[row] = mrg[0][tid][2] * c0 + mrg[1][tid][2] * c1 + ((rc & 1) ? 1.f : 0.f);
        s_other[row] = mrg[0][tid][3] * c0 + mrg[1][tid][3] * c1 + ((rc & 2) ? 1.f : 0.f);
        e_pair[row] = __expf(fmaxf(mrg[0][tid][4], mrg[1][tid][4]) - m);
    }
}

// dF[row] = inv_T * sum_col W[row][col] F[col],
//   TRANS == 0:  W = E[row][col] * c(row; col),   E = exp(S - rowmax[row])          (d/d row-side features)
//   TRANS == 1:  W = E[col][row] * c(col; row),   E = exp(S - rowmax[col])          (d/d column-side features)
// with c(i; j) = g_all[i] + g_pos[i] [pos j] + g_other[i] [other j] + g_pair[i] [j == pair(i)], zero on the diagonal.
// The W tile goes through LDS to become the A operand of the second product.
template <int DIM, int TRANS>
__global__ __launch_bounds__(256) void ucl_bwd_kernel(const float* feat, const uint8_t* cls, int n2, int n_half,
                                                     float inv_T, const float* rowmax, const float* g_all,
                                                     const float* g_pos, const float* g_other, const float* g_pair,
                                                     float* dfeat, int accumulate) {
    constexpr int LD = DIM + 1;
    constexpr int WL = UB + 1;
    constexpr int NT = DIM / 32;                          // 32-wide output column tiles of the second product
    static_assert(DIM % 32 == 0, "feature dim");
    // one LDS arena: [column-tile features UB x LD][4 waves x (32 x 33) W tiles]; after the walk the same
    // memory holds the two column halves' partial dF (2 x UB x LD)
    __shared__ float arena[UB * LD + 4 * 32 * 33];
    static_assert(2 * UB * LD <= UB * LD + 4 * 32 * 33, "epilogue does not fit the arena");
    float* const colf = arena;
    float (*const wt)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(arena + UB * LD);
    __shared__ float cmeta[UB][5];                        // TRANS: rowmax, g_* of the tile's columns
    __shared__ uint8_t colc[UB];
    (void)WL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l32 = lane & 31;
    const int row0 = blockIdx.x * UB;

    float af[DIM / 2];
    {
        const int r = row0 + wm * 32 + l32;
#pragma unroll
        for (int t = 0; t < DIM / 2; ++t) af[t] = r < n2 ? feat[(long)r * DIM + 2 * t + h] : 0.f;
    }
    // per-row metadata of the 16 rows this lane sees in the C layout
    float rm[16], ra[16], rp[16], ro[16], rq[16];
    uint8_t rcl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = row < n2;
        rm[r] = ok ? rowmax[row] : 0.f;
        ra[r] = ok ? g_all[row] : 0.f; rp[r] = ok ? g_pos[row] : 0.f;
        ro[r] = ok ? g_other[row] : 0.f; rq[r] = ok ? g_pair[row] : 0.f;
        rcl[r] = ok ? cls[row] : 0;
    }
    f32x16 out[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[j][r] = 0.f;

    for (int col0 = 0; col0 < n2; col0 += UB) {
        __syncthreads();
        for (int q = tid; q < UB * (DIM / 4); q += 256) {
            const int c = q / (DIM / 4), k4 = (q % (DIM / 4)) * 4;
            float4 v = make_float4(0, 0, 0, 0);
            if (col0 + c < n2) v = ld4(feat + (long)(col0 + c) * DIM + k4);
            float* d = colf + c * LD + k4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (tid < UB) {
            const int c = col0 + tid;
            const bool ok = c < n2;
            colc[tid] = ok ? cls[c] : 0;
            if (TRANS) {
                cmeta[tid][0] = ok ? rowmax[c] : 0.f; cmeta[tid][1] = ok ? g_all[c] : 0.f;
                cmeta[tid][2] = ok ? g_pos[c] : 0.f; cmeta[tid][3] = ok ? g_other[c] : 0.f;
                cmeta[tid][4] = ok ? g_pair[c] : 0.f;
            }
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bsrc = colf + (wn * 32 + l32) * LD + h;
#pragma unroll
        for (int t = 0; t < DIM / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bsrc[2 * t], acc, 0, 0, 0);
        const int lc = wn * 32 + l32, col = col0 + lc;
        const bool colok = col < n2;
        const uint8_t cc = colc[lc];
        float* wrow = wt[wave];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = row0 + wm * 32 + tr;
            float w = 0.f;
            if (colok && row < n2 && col != row) {
                const float s = acc[r] * inv_T;
                const int pr = row < n_half ? row + n_half : row - n_half;
                if (!TRANS) {
                    w = __expf(s - rm[r]) * (ra[r] + ((cc & 1) ? rp[r] : 0.f) + ((cc & 2) ? ro[r] : 0.f) +
                                             (col == pr ? rq[r] : 0.f));
                } else {
                    const float* cm = cmeta[lc];
                    w = __expf(s - cm[0]) * (cm[1] + ((rcl[r] & 1) ? cm[2] : 0.f) + ((rcl[r] & 2) ? cm[3] : 0.f) +
                                             (col == pr ? cm[4] : 0.f));
                }
            }
            wrow[tr * 33 + l32] = w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes are visible to it
        // second product: out[32 rows][DIM] += W[32 x 32] * Fcol[32 x DIM]; A = W (rows x k = tile cols)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float a = wrow[l32 * 33 + 2 * t + h];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float b = colf[(wn * 32 + 2 * t + h) * LD + j * 32 + l32];
                out[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, out[j], 0, 0, 0);
            }
        }
    }
    // sum the two column halves and write
    __syncthreads();                                      // the arena is free
    float* const outm = arena;                            // [2][UB][LD]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            outm[(wn * UB + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LD + j * 32 + l32] = out[j][r];
    __syncthreads();
    for (int q = tid; q < UB * DIM; q += 256) {
        const int tr = q / DIM, k = q % DIM;
        const int row = row0 + tr;
        if (row < n2) {
            const float v = (outm[tr * LD + k] + outm[(UB + tr) * LD + k]) * inv_T;
            dfeat[(long)row * DIM + k] = accumulate ? dfeat[(long)row * DIM + k] + v : v;
        }
    }
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------
extern "C" size_t mi_voxel_loss_workspace_bytes(long n) {
    return n > 0 ? sizeof(double) * NQ * (size_t)vl_blocks(n) : 0;
}

static int voxel_loss_fwd(int mode, const float* pred, const float* gt, long n, double tau, double beta, double* sums,
                          float* loss, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!pred || !gt || !sums || !loss || n <= 0) return MI_E_ARG;
    if (!ws || ws_bytes < mi_voxel_loss_workspace_bytes(n)) return MI_E_WORKSPACE;
    const int blocks = vl_blocks(n);
    double* part = (double*)ws;
    if (mode == VL_PU) hipLaunchKernelGGL((voxel_loss_partial_kernel<VL_PU>), dim3(blocks), dim3(256), 0, s, pred, gt, n, part);
    else if (mode == VL_FOCAL) hipLaunchKernelGGL((voxel_loss_partial_kernel<VL_FOCAL>), dim3(blocks), dim3(256), 0, s, pred, gt, n, part);
    else hipLaunchKernelGGL((voxel_loss_partial_kernel<VL_MSE>), dim3(blocks), dim3(256), 0, s, pred, gt, n, part);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(voxel_loss_final_kernel, dim3(1), dim3(256), 0, s, (const double*)part, blocks, mode, n, tau, beta,
                       sums, loss);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
static int voxel_loss_bwd(int mode, const float* pred, const float* gt, long n, double tau, const double* sums,
                          const float* dloss, float* dpred, float* dgt, hipStream_t s) {
    if (!pred || !gt || !sums || !dloss || !dpred || n <= 0) return MI_E_ARG;
    const int blocks = (int)std::max<long>(1, std::min<long>((n + 255) / 256, 4096));
    if (mode == VL_PU) hipLaunchKernelGGL((voxel_loss_bwd_kernel<VL_PU>), dim3(blocks), dim3(256), 0, s, pred, gt, n, tau, sums, dloss, dpred, dgt);
    else if (mode == VL_FOCAL) hipLaunchKernelGGL((voxel_loss_bwd_kernel<VL_FOCAL>), dim3(blocks), dim3(256), 0, s, pred, gt, n, tau, sums, dloss, dpred, dgt);
    else hipLaunchKernelGGL((voxel_loss_bwd_kernel<VL_MSE>), dim3(blocks), dim3(256), 0, s, pred, gt, n, tau, sums, dloss, dpred, dgt);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_pu_focal_loss_fwd(const float* pred, const float* gt, long n, double tau, double beta, double* sums,
                                    float* loss, void* ws, size_t ws_bytes, mi_stream_t stream) {
    return voxel_loss_fwd(VL_PU, pred, gt, n, tau, beta, sums, loss, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mi_pu_focal_loss_bwd(const float* pred, const float* gt, long n, double tau, const double* sums,
                                    const float* dloss, float* dpred, mi_stream_t stream) {
    return voxel_loss_bwd(VL_PU, pred, gt, n, tau, sums, dloss, dpred, nullptr, (hipStream_t)stream);
}
extern "C" int mi_focal_loss_fwd(const float* pred, const float* gt, long n, double* sums, float* loss, void* ws,
                                 size_t ws_bytes, mi_stream_t stream) {
    return voxel_loss_fwd(VL_FOCAL, pred, gt, n, 0, 0, sums, loss, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mi_focal_loss_bwd(const float* pred, const float* gt, long n, const double* sums, const float* dloss,
                                 float* dpred, mi_stream_t stream) {
    return voxel_loss_bwd(VL_FOCAL, pred, gt, n, 0, sums, dloss, dpred, nullptr, (hipStream_t)stream);
}
extern "C" int mi_mse_loss_fwd(const float* a, const float* b, long n, double* sums, float* loss, void* ws,
                               size_t ws_bytes, mi_stream_t stream) {
    return voxel_loss_fwd(VL_MSE, a, b, n, 0, 0, sums, loss, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mi_mse_loss_bwd(const float* a, const float* b, long n, const double* sums, const float* dloss,
                               float* da, float* db, mi_stream_t stream) {
    return voxel_loss_bwd(VL_MSE, a, b, n, 0, sums, dloss, da, db, (hipStream_t)stream);
}

extern "C" int mi_ucl_rowsums_fwd(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T, float* rowmax,
                                  float* s_all, float* s_pos, float* s_other, float* e_pair, mi_stream_t stream) {
    if (!feat || !cls || !rowmax || !s_all || !s_pos || !s_other || !e_pair || n2 <= 0 || (n2 & 1)) return MI_E_ARG;
    const dim3 grid((n2 + UB - 1) / UB), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dim == 32) hipLaunchKernelGGL((ucl_fwd_kernel<32>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, s_all, s_pos, s_other, e_pair);
    else if (dim == 64) hipLaunchKernelGGL((ucl_fwd_kernel<64>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, s_all, s_pos, s_other, e_pair);
    else return MI_E_UNSUPPORTED;
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_ucl_rowsums_bwd(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T,
                                  const float* rowmax, const float* g_all, const float* g_pos, const float* g_other,
                                  const float* g_pair, float* dfeat, mi_stream_t stream) {
    if (!feat || !cls || !rowmax || !g_all || !g_pos || !g_other || !g_pair || !dfeat || n2 <= 0 || (n2 & 1)) return MI_E_ARG;
    const dim3 grid((n2 + UB - 1) / UB), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MI_UCL_BWD(D_)                                                                                              \
    hipLaunchKernelGGL((ucl_bwd_kernel<D_, 0>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, \
                       g_other, g_pair, dfeat, 0);                                                                  \
    hipLaunchKernelGGL((ucl_bwd_kernel<D_, 1>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, \
                       g_other, g_pair, dfeat, 1)
    if (dim == 32) { MI_UCL_BWD(32); }
    else if (dim == 64) { MI_UCL_BWD(64); }
    else return MI_E_UNSUPPORTED;
#undef MI_UCL_BWD
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
