// Detector-training losses, reference cet_pick/models/loss.py:
//   _pu_neg_loss   :255-308   positive-unlabeled focal risk (PULoss :310-324)
//   _neg_loss      :378-411   CornerNet focal loss (FocalLoss)
//   ConsistencyLoss:701-712   mean squared error between the two views' heat-maps
//   UnbiasedConLoss:571-699   debiased contrastive regulariser over a (2N x 2N) similarity matrix
// The voxel losses are one streaming pass (fp64 partial sums, deterministic two-level tree) plus a
// one-thread finalize that also takes the data-dependent branch of the PU risk on the device, and an
// elementwise backward.  The contrastive loss never materialises the (2N)^2 matrix (2.4 GB at N = 12,288):
// a row block keeps its features in registers, walks the column tiles through LDS, forms each 32x32 tile of
// similarities on the matrix cores (round 4: f32-equivalent products as six bf16 MFMAs of a 3-way cut, UclS) and folds exp() of it straight
// into the four row sums the loss needs (online max, flash-attention style).  The backward recomputes the
// tiles and contracts them with the features again on the matrix cores.
// (the SLP vectoriser pairs the row bookkeeping of the contrastive loss into v_pk_mul_f32 / v_pk_add_f32: 14 cycles of matrix-pipe
// throughput each next to the MFMAs where the scalar FP32 operations they replace are hidden - tools/probes/mfma_coissue.hip)
// hipcc-flags: -fno-slp-vectorize
#include "common.h"
#include "../../include/cetpick_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4w lds_bf16x4w;
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Exact 3-way bf16 cut of eight f32 values (a = a0 + a1 + a2, truncation: every subtraction is exact), packed as the three
// MFMA operand planes - the arithmetic of the convolution kernels (conv_cube2.hip cut8r): six bf16 products of
// weight <= 2 accumulated in f32 are an f32-equivalent product on the bf16 matrix pipe (16x the f32 MFMA rate).
__device__ __forceinline__ void ucl_cut8(const float (&v)[8], bf16x8 (&o)[3]) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        u0[t] = __float_as_uint(v[t]);
        const float r1 = v[t] - __uint_as_float(u0[t] & 0xffff0000u);
        u1[t] = __float_as_uint(r1);
        u2[t] = __float_as_uint(r1 - __uint_as_float(u1[t] & 0xffff0000u));
    }
    constexpr unsigned HI2 = 0x07060302u;
    u32x4 p0, p1, p2;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        p0[d] = __builtin_amdgcn_perm(u0[2 * d + 1], u0[2 * d], HI2);
        p1[d] = __builtin_amdgcn_perm(u1[2 * d + 1], u1[2 * d], HI2);
        p2[d] = __builtin_amdgcn_perm(u2[2 * d + 1], u2[2 * d], HI2);
    }
    o[0] = __builtin_bit_cast(bf16x8, p0); o[1] = __builtin_bit_cast(bf16x8, p1); o[2] = __builtin_bit_cast(bf16x8, p2);
}

// The similarity tile S = F_rows . F_cols^T of the contrastive loss on the bf16 pipe.  A (this wave's 32 rows, constant over
// the column walk): lane = (row l32, features 16 ks + 8 h .. + 7), cut once into registers.  B (the 64 columns of a tile):
// cut once per tile while it is staged, three planes of [column][DIM] bf16 rows in LDS (row pitch DIM * 2 + 16 bytes: the
// 16-byte fragment reads of eight consecutive lanes fall on disjoint banks).
template <int DIM> struct UclS {
    static constexpr int KS = DIM / 16;                  // k-steps of v_mfma_f32_32x32x16_bf16
    static constexpr int PITCH = DIM * 2 + 16;           // bytes per column row of a plane
    static constexpr int PLANE = 64 * PITCH;
    static constexpr int BYTES = 3 * PLANE;
    // rows [row0, row0 + 32) of feat -> A fragments
    // fs: factor applied to every feature before the cut (the forward pass folds 1 / T and log2(e) into the operands)
    static __device__ __forceinline__ void load_a(const float* feat, int row, bool ok, int h, bf16x8 (&af)[KS][3], float fs = 1.f) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
            const float4 a = ok ? ld4(feat + (long)row * DIM + 16 * ks + 8 * h) : make_float4(0, 0, 0, 0);
            const float4 b = ok ? ld4(feat + (long)row * DIM + 16 * ks + 8 * h + 4) : make_float4(0, 0, 0, 0);
            v[0] = a.x * fs; v[1] = a.y * fs; v[2] = a.z * fs; v[3] = a.w * fs; v[4] = b.x * fs; v[5] = b.y * fs; v[6] = b.z * fs; v[7] = b.w * fs;
            ucl_cut8(v, af[ks]);
        }
    }
    // one unit of the tile (column c, features k8 .. k8 + 7) -> the three planes
    static __device__ __forceinline__ void stage(unsigned char* planes, int c, int k8, const float (&v)[8]) {
        bf16x8 o[3];
        ucl_cut8(v, o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<u32x4*>(planes + pl * PLANE + c * PITCH + k8 * 2) = __builtin_bit_cast(u32x4, o[pl]);
    }
    // acc += A . B^T for the 32 columns [cb, cb + 32) of the tile
    static __device__ __forceinline__ f32x16 product(const unsigned char* planes, const bf16x8 (&af)[KS][3], int cb, int l32,
                                                     int h) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest terms first
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(planes + pl * PLANE + (cb + l32) * PITCH +
                                                                                     (16 * ks + 8 * h) * 2));
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][PA[pr]], bf[PB[pr]], acc, 0, 0, 0);
        }
        return acc;
    }
};

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
constexpr int UCL_FLUSH = 16;     // column tiles per first-level accumulator of the backward's second product

// per-lane online state of one row
struct RowAcc { float m, ref, sa, sp, so; };      // m: running maximum; ref: what the three sums are relative to

template <int DIM>
// (three waves per SIMD: 168 registers, seven dwords of scratch outside the column walk - 118 ms against 127 for the C5 step)
__global__ __launch_bounds__(256, 3) void ucl_fwd_kernel(const float* feat, const uint8_t* cls, int n2, int n_half,
                                                     float inv_T, float* rowmax, float* s_all, float* s_pos,
                                                     float* s_other, float* e_pair) {
    typedef UclS<DIM> SP;
    // (round 6: two column tiles resident - tile t + 1 is fetched, cut and stored while tile t's product and exponentials run: ONE
    // barrier per tile instead of two)
    constexpr bool DB = DIM == 32;                        // (64-wide features: the second buffer would cost the third resident workgroup)
    __shared__ __attribute__((aligned(16))) unsigned char colb_[DB ? 2 : 1][SP::BYTES];
    __shared__ uint8_t colc_[DB ? 2 : 1][UB];
    __shared__ float mrg[2][UB][4];                       // merge of the two column halves (wn)
    __shared__ float s_pair[UB];                          // S[row][pair(row)]: one writer per row in the whole walk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l32 = lane & 31;
    const int row0 = blockIdx.x * UB;

    // The walk runs in the exponent's own units: both operands are scaled by sqrt(log2(e) / T) before the cut, so a tile element IS
    // log2(e) S / T and exp() is one v_exp_f32 of a difference - the multiplication by 1 / T and the one inside __expf, per tile
    // element and next to the MFMAs become eight multiplications per
    // thread and tile in the staging.  Row maxima leave the kernel in natural units (x ln 2).
    const float fs = sqrtf(inv_T * 1.4426950408889634f);
    // A fragments: this wave's 32 rows, constant over the column walk (bf16x3 cut, once)
    bf16x8 af[SP::KS][3];
    SP::load_a(feat, row0 + wm * 32 + l32, row0 + wm * 32 + l32 < n2, h, af, fs);
    RowAcc st[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = {-INFINITY, -INFINITY, 0.f, 0.f, 0.f};
    if (tid < UB) s_pair[tid] = -INFINITY;

    auto stage_tile = [&](int b, int c0) {
        for (int q = tid; q < UB * (DIM / 8); q += 256) {
            const int c = q / (DIM / 8), k8 = (q % (DIM / 8)) * 8;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c0 + c < n2) {
                const float4 a = ld4(feat + (long)(c0 + c) * DIM + k8), b4 = ld4(feat + (long)(c0 + c) * DIM + k8 + 4);
                v[0] = a.x * fs; v[1] = a.y * fs; v[2] = a.z * fs; v[3] = a.w * fs; v[4] = b4.x * fs; v[5] = b4.y * fs; v[6] = b4.z * fs; v[7] = b4.w * fs;
            }
            SP::stage(colb_[b], c, k8, v);
        }
        if (tid < UB) colc_[b][tid] = c0 + tid < n2 ? cls[c0 + tid] : 0;
    };
    if (DB) {
        stage_tile(0, 0);
        __syncthreads();
    }
    for (int col0 = 0; col0 < n2; col0 += UB) {
        const int cur = DB ? (col0 / UB) & 1 : 0;
        if (DB) {
            if (col0 + UB < n2) stage_tile(cur ^ 1, col0 + UB);
        } else {
            __syncthreads();
            stage_tile(0, col0);
            __syncthreads();
        }
        const unsigned char* colb = colb_[cur];
        const uint8_t* colc = colc_[cur];
        const f32x16 acc = SP::product(colb, af, wn * 32, l32, h);
        const int col = col0 + wn * 32 + l32;
        const bool colok = col < n2;
        const uint8_t cc = colc[wn * 32 + l32];
        const float fp = (cc & 1) ? 1.f : 0.f, fo = (cc & 2) ? 1.f : 0.f;
        // (the pair element (row, pair(row)) exists in ONE lane of the whole walk: it goes to LDS, not into per-row registers)
        const int rlo = row0 + wm * 32, clo = col0 + wn * 32;
        const bool pair_tile = !((clo + 32 <= rlo + n_half || clo >= rlo + 32 + n_half) &&
                                 (clo + 32 <= rlo - n_half || clo >= rlo + 32 - n_half));
        // The sums of a row are kept relative to a REFERENCE that only moves when an element exceeds it by more than 2^64 (the
        // first element, then almost never), not to the running maximum: per element one v_max for the maximum, one v_sub + v_exp
        // for the term and three accumulations - the rescale-by-exp of the textbook online softmax (a second exp, a compare and
        // four selects per element once the compiler has if-converted it) only runs when some lane of the wave needs it.
        if (colok) {
            float d[16], dmax;                            // element - reference, and the largest of the lane's sixteen
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = acc[r] - st[r].ref;
            dmax = fmaxf(fmaxf(d[0], d[1]), d[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) dmax = fmaxf(fmaxf(dmax, d[r]), d[r + 1]);
            dmax = fmaxf(dmax, d[15]);
            if (__builtin_amdgcn_ballot_w64(dmax > 64.f) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    RowAcc& a = st[r];
                    if (d[r] > 64.f) {
                        const float sc = __builtin_amdgcn_exp2f(-d[r]);             // 2^(-inf) = 0 on the first element
                        a.sa *= sc; a.sp *= sc; a.so *= sc; a.ref = acc[r]; d[r] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float s = acc[r];                   // log2(e) S / T
                RowAcc& a = st[r];
                a.m = fmaxf(a.m, s);
                if (col != row) {                         // the diagonal only takes part in the maximum
                    const float e = __builtin_amdgcn_exp2f(d[r]);
                    a.sa += e; a.sp += e * fp; a.so += e * fo;
                }
                if (pair_tile) {                          // (wave-uniform)
                    const int pr = row < n_half ? row + n_half : row - n_half;
                    if (col == pr) s_pair[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = s;
                }
            }
        }
        if (DB) __syncthreads();                           // the next tile is in place; this one's buffer is free
    }
    // merge the 32 lanes that share a row (same h), then the two column halves
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        RowAcc a = st[r];
        float m = a.m;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float sc = __builtin_amdgcn_exp2f(a.ref - m);          // (a lane that saw no column: ref = -inf, sums 0)
        float sa = a.sa * sc, sp = a.sp * sc, so = a.so * sc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sp += __shfl_xor(sp, o, 64); so += __shfl_xor(so, o, 64); }
        if (l32 == 0) {
            const int tr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            mrg[wn][tr][0] = m; mrg[wn][tr][1] = sa; mrg[wn][tr][2] = sp; mrg[wn][tr][3] = so;
        }
    }
    __syncthreads();
    if (tid < UB && row0 + tid < n2) {
        const float m0 = mrg[0][tid][0], m1 = mrg[1][tid][0], m = fmaxf(m0, m1);
        const float c0 = __builtin_amdgcn_exp2f(m0 - m), c1 = __builtin_amdgcn_exp2f(m1 - m);
        const int row = row0 + tid;
        rowmax[row] = m * 0.6931471805599453f;           // back to natural units
        // the masked diagonal contributes exp(0) = 1 to every column sum it belongs to (loss.py:622-624)
        const uint8_t rc = cls[row];
        s_all[row] = mrg[0][tid][1] * c0 + mrg[1][tid][1] * c1 + 1.f;
        s_pos[row] = mrg[0][tid][2] * c0 + mrg[1][tid][2] * c1 + ((rc & 1) ? 1.f : 0.f);
        s_other[row] = mrg[0][tid][3] * c0 + mrg[1][tid][3] * c1 + ((rc & 2) ? 1.f : 0.f);
        e_pair[row] = __builtin_amdgcn_exp2f(s_pair[tid] - m);
    }
}

// dF[row] = inv_T * sum_col W[row][col] F[col],
//   TRANS == 0:  W = E[row][col] * c(row; col),   E = exp(S - rowmax[row])          (d/d row-side features)
//   TRANS == 1:  W = E[col][row] * c(col; row),   E = exp(S - rowmax[col])          (d/d column-side features)
//   TRANS == 2:  both at once (round 6): S is symmetric, so the tile (rows R, columns C) a row block forms for its row-side term IS the
//                transpose of the tile (C, R) its column-side term needs - W = E_R c(row; col) + E_C c(col; row) from ONE similarity
//                product, ONE staged column tile and ONE contraction with F[col] (before: two kernels, each with both products)
// with c(i; j) = g_all[i] + g_pos[i] [pos j] + g_other[i] [other j] + g_pair[i] [j == pair(i)], zero on the diagonal.
// The W tile goes through LDS to become the A operand of the second product.
template <int DIM, int TRANS>
__global__ __launch_bounds__(256, (DIM == 64 || TRANS == 2) ? 2 : 3) void ucl_bwd_kernel(const float* feat, const uint8_t* cls, int n2, int n_half,
                                                     float inv_T, const float* rowmax, const float* g_all,
                                                     const float* g_pos, const float* g_other, const float* g_pair,
                                                     float* dfeat, int accumulate, const float* range = nullptr) {
    // TRANS == 3 (round 6): TRANS == 2 with ONE exponential per similarity.  exp(S - max_row) = exp(S - M) exp(M - max_row) for any
    // reference M; with M = the largest row maximum (range[0], base-2 units) the second factor is a per-row constant folded into the
    // row's g_* once, and a tile element costs one exp2 for both of its terms.  Safe when the row maxima lie within 2^16 of each other
    // (range[1] != 0: the factors stay far from overflow and what underflows in exp(S - M) is < 2^-100 of its row's largest term) - always
    // so for L2-normalised features (the diagonal 1 / T is every row's maximum), which is what the detector's projection head emits;
    // otherwise this kernel returns at once and the TRANS == 2 launch behind it does the work (and vice versa).
    if (TRANS >= 2 && range != nullptr && (range[1] != 0.f) != (TRANS == 3)) return;
    constexpr int LD = DIM + 1;
    constexpr int WL = UB + 1;
    constexpr int NT = DIM / 32;                          // 32-wide output column tiles of the second product
    static_assert(DIM % 32 == 0, "feature dim");
    // one LDS arena: [column-tile features UB x LD][4 waves x (32 x 33) W tiles]; after the walk the same
    // memory holds the two column halves' partial dF (2 x UB x LD)
    typedef UclS<DIM> SP;
    constexpr int WP = 36;                                // W tile row pitch (floats): 16-byte rows, bank-disjoint b128 reads
    // (TRANS >= 2: the epilogue's 2 x UB x LD floats alias the two column-tile buffers instead - 8.4 KB less, which with the register
    // diet below lets THREE workgroups share a CU: the per-tile chain of a wave is serial and two waves per SIMD did not cover it)
    constexpr int COLF = TRANS >= 2 ? 0 : ((UB * LD + 3) & ~3);   // (the W tiles start 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float arena[COLF + 4 * 32 * WP];
    static_assert(TRANS >= 2 || 2 * UB * LD <= COLF + 4 * 32 * WP, "epilogue does not fit the arena");
    float (*const wt)[32 * WP] = reinterpret_cast<float (*)[32 * WP]>(arena + COLF);
    // (round 6, the merged forms: two column tiles resident - tile t + 1 is cut and stored while tile t's products
    // run, ONE barrier per tile instead of two)
    constexpr bool DB = TRANS >= 2;
    __shared__ __attribute__((aligned(16))) unsigned char colb_[DB ? 2 : 1][SP::BYTES];      // the tiles' bf16x3 planes (first product)
    __shared__ float cmeta_[DB ? 2 : 1][UB][5];           // TRANS: rowmax, g_* of the tile's columns
    __shared__ uint8_t colc_[DB ? 2 : 1][UB];
    (void)WL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l32 = lane & 31;
    const int row0 = blockIdx.x * UB;

    // (as the forward pass: operands scaled by sqrt(log2(e) / T), a tile element is the exponent in base-2 units; the row maxima
    // arrive in natural units and are converted once per row / column; the second product contracts W with the SCALED features, so
    // the epilogue's factor is 1 / (T fs) instead of 1 / T)
    const float fs = sqrtf(inv_T * 1.4426950408889634f), LOG2E = 1.4426950408889634f;
    bf16x8 af[SP::KS][3];
    SP::load_a(feat, row0 + wm * 32 + l32, row0 + wm * 32 + l32 < n2, h, af, fs);
    // per-row metadata of the 16 rows this lane sees in the C layout
    // TRANS == 0 needs five numbers per row and element: rowmax and g_all stay in registers, (g_pos, g_other, g_pair) of the
    // workgroup's 64 rows sit in LDS and come as ONE 16-byte broadcast read per row (all five in registers: 198 VGPRs, two
    // waves per SIMD - the kernel's vector work hides behind a third wave's MFMAs)
    __shared__ __attribute__((aligned(16))) float rmeta[UB][4];
    const float gmax = TRANS == 3 ? range[0] : 0.f;         // M, base-2 units
    if (TRANS != 1 && tid < UB) {
        const int row = row0 + tid;
        const bool ok = row < n2;
        const float sg = (TRANS == 3 && ok) ? __builtin_amdgcn_exp2f(gmax - rowmax[row] * LOG2E) : 1.f;
        rmeta[tid][0] = ok ? g_pos[row] * sg : 0.f; rmeta[tid][1] = ok ? g_other[row] * sg : 0.f;
        rmeta[tid][2] = ok ? g_pair[row] * sg : 0.f; rmeta[tid][3] = (TRANS == 3 && ok) ? g_all[row] * sg : 0.f;
    }
    __shared__ uint8_t rcls[UB];                          // TRANS == 3: the rows' class bytes (the rare diagonal / pair / ragged tiles read them)
    if (TRANS == 3 && tid < UB) rcls[tid] = row0 + tid < n2 ? cls[row0 + tid] : 0;
    // (TRANS == 3 keeps none of these: its row maximum is the constant M, g_all' and the class bytes sit in LDS for the rare tiles that
    // need them - 40 registers less, the third resident workgroup)
    float rm[TRANS == 3 ? 1 : 16], ra[TRANS == 3 ? 1 : 16];
    uint8_t rcl[TRANS == 3 ? 1 : 16];
    if constexpr (TRANS != 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = row < n2;
            rm[r] = (TRANS != 1 && ok) ? rowmax[row] * LOG2E : 0.f;
            ra[r] = (TRANS != 1 && ok) ? g_all[row] : 0.f;
            rcl[r] = ok ? cls[row] : 0;
        }
    } else { rm[0] = gmax; ra[0] = 0.f; rcl[0] = 0; }
    // Two levels of accumulation: `out` collects UCL_FLUSH column tiles on the matrix pipe and is then added into `tot` by the
    // vector unit.  One accumulator for the whole walk (2N / 32 tiles x 12 MFMAs = 73,728 dependent accumulations at 2N =
    // 196,608) showed a bias of -2e-5 of the sum, growing linearly with N - the MFMA aligns its products to the (large)
    // accumulator and drops what falls below it, always towards zero (tools/ab/ucl_bwd_diag.py: -1.9e-5 / -6e-6 at 196,608 /
    // 24,576 rows against dense float64 rows; a row of dF is ~1/400 of its terms' magnitudes, so that was 4e-4 of the row).
    // (64-wide features - two accumulators per wave - keep the single level: the second pair of totals would cost the kernel a
    // resident wave per SIMD; the detector's projection head is 32 wide)
    // TRANS == 3 (round 6): the coefficient of a tile element is BILINEAR in a row vector and a column vector,
    //   cf(row, col) = g_all'[row] 1 + g_pos'[row] [pos col] + g_other'[row] [other col] + 1 g_all'[col] + [pos row] g_pos'[col] + [other row] g_other'[col]
    // (' = scaled by the row's / column's exp(M - max)), i.e. a rank-6 product U V^T: three v_mfma_f32_32x32x2_f32 per tile on the
    // idle matrix pipe instead of ~12 vector operations per element (the kernel was bound by vector issue: ~350 VALU instructions per
    // tile and wave against 24 MFMAs).  A operand: this wave's row l32, k = h of each of the three k-steps - constant over the walk.
    float ua[3] = {0.f, 0.f, 0.f};
    if (TRANS == 3) {
        const int row = row0 + wm * 32 + l32;
        const bool ok = row < n2;
        const float sg = ok ? __builtin_amdgcn_exp2f(gmax - rowmax[row] * LOG2E) : 0.f;
        const uint8_t rc = ok ? cls[row] : 0;
        ua[0] = ok ? (h ? g_pos[row] : g_all[row]) * sg : 0.f;
        ua[1] = h ? (ok ? 1.f : 0.f) : (ok ? g_other[row] * sg : 0.f);
        ua[2] = h ? (float)((rc >> 1) & 1) : (float)(rc & 1);
    }
    constexpr bool TWO_LEVEL = NT == 1;
    f32x16 out[NT], tot[TWO_LEVEL ? NT : 1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { out[j][r] = 0.f; if (TWO_LEVEL) tot[j][r] = 0.f; }

    constexpr int NPRE = UB * (DIM / 8) / 256;            // staging units (column, 8 features) per thread and tile
    static_assert(UB * (DIM / 8) % 256 == 0, "staging units");
    float4 pre[NPRE][2];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int q = tid + 256 * u;
            const int c = q / (DIM / 8), k8 = (q % (DIM / 8)) * 8;
            const bool ok = c0 + c < n2;
            pre[u][0] = ok ? ld4(feat + (long)(c0 + c) * DIM + k8) : make_float4(0, 0, 0, 0);
            pre[u][1] = ok ? ld4(feat + (long)(c0 + c) * DIM + k8 + 4) : make_float4(0, 0, 0, 0);
        }
    };
    constexpr bool PRE = DB || DIM == 64;                 // (the two-launch 32-wide forms run three waves per SIMD: no registers to spare)
    // cut + store the tile held in `pre` (columns c0 ..) into buffer b, with its columns' metadata
    auto stage_tile = [&](int b, int c0) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int q = tid + 256 * u;
            const int c = q / (DIM / 8), k8 = (q % (DIM / 8)) * 8;
            float v[8];
            v[0] = pre[u][0].x * fs; v[1] = pre[u][0].y * fs; v[2] = pre[u][0].z * fs; v[3] = pre[u][0].w * fs;
            v[4] = pre[u][1].x * fs; v[5] = pre[u][1].y * fs; v[6] = pre[u][1].z * fs; v[7] = pre[u][1].w * fs;
            SP::stage(colb_[b], c, k8, v);                 // (both products read the planes)
        }
        if (tid < UB) {
            const int c = c0 + tid;
            const bool ok = c < n2;
            colc_[b][tid] = ok ? cls[c] : 0;
            if (TRANS != 0) {
                const float cmxv = ok ? rowmax[c] * LOG2E : 0.f;
                const float sg = (TRANS == 3 && ok) ? __builtin_amdgcn_exp2f(gmax - cmxv) : 1.f;
                cmeta_[b][tid][0] = TRANS == 3 ? gmax : cmxv; cmeta_[b][tid][1] = ok ? g_all[c] * sg : 0.f;
                cmeta_[b][tid][2] = ok ? g_pos[c] * sg : 0.f; cmeta_[b][tid][3] = ok ? g_other[c] * sg : 0.f;
                cmeta_[b][tid][4] = ok ? g_pair[c] * sg : 0.f;
            }
        }
    };
    if (DB) {
        prefetch(0);
        stage_tile(0, 0);
        prefetch(UB);
        __syncthreads();
    } else if (PRE) prefetch(0);
    for (int col0 = 0; col0 < n2; col0 += UB) {
        if (TWO_LEVEL && (col0 / UB) % UCL_FLUSH == UCL_FLUSH - 1) {     // (uniform)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot[j][r] += out[j][r]; out[j][r] = 0.f; }
        }
        const int cur = DB ? (col0 / UB) & 1 : 0;
        if (!DB) {
            __syncthreads();
            if (!PRE) prefetch(col0);                      // (PRE: fetched during the previous tile's products)
            stage_tile(0, col0);
            if (PRE) prefetch(col0 + UB);
            __syncthreads();
        } else {
            // the next tile goes into the other buffer (its features arrived during the previous tile), the one after it is fetched
            if (col0 + UB < n2) stage_tile(cur ^ 1, col0 + UB);
            prefetch(col0 + 2 * UB);
        }
        const unsigned char* colb = colb_[cur];
        const float (*cmeta)[5] = cmeta_[cur];
        const uint8_t* colc = colc_[cur];
        const f32x16 acc = SP::product(colb, af, wn * 32, l32, h);
        const int lc = wn * 32 + l32, col = col0 + lc;
        const bool colok = col < n2;
        const uint8_t cc = colc[lc];
        float* wrow = wt[wave];
        const int rlo = row0 + wm * 32, clo = col0 + wn * 32;
        const bool plain = clo + 32 <= n2 && rlo + 32 <= n2 && (clo + 32 <= rlo || clo >= rlo + 32) &&
                           (clo + 32 <= rlo + n_half || clo >= rlo + 32 + n_half) &&
                           (clo + 32 <= rlo - n_half || clo >= rlo + 32 - n_half);
        if (plain) {                                       // (wave-uniform) no diagonal, no pair element, nothing ragged
            float c0, c1, c2, cmx;
            if (TRANS == 1) {
                const float* cm = cmeta[lc];
                cmx = cm[0]; c0 = cm[1]; c1 = cm[2]; c2 = cm[3];
            } else {
                c0 = (cc & 1) ? 1.f : 0.f; c1 = (cc & 2) ? 1.f : 0.f; c2 = 0.f; cmx = 0.f;
            }
            float k0 = 0.f, k1 = 0.f, k2 = 0.f, kmx = 0.f;             // TRANS == 2: the column's g_all, g_pos, g_other and maximum
            if (TRANS >= 2) { const float* cm = cmeta[lc]; kmx = cm[0]; k0 = cm[1]; k1 = cm[2]; k2 = cm[3]; }
            f32x16 cf3;
            if (TRANS == 3) {                             // the tile's coefficients: U V^T, B operand = this lane's column, k = h
#pragma unroll
                for (int r = 0; r < 16; ++r) cf3[r] = 0.f;
                cf3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[0], h ? c0 : 1.f, cf3, 0, 0, 0);
                cf3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[1], h ? k0 : c1, cf3, 0, 0, 0);
                cf3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2], h ? k2 : k1, cf3, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float s = acc[r];
                float w;
                if constexpr (TRANS == 3) {
                    w = __builtin_amdgcn_exp2f(s - gmax) * cf3[r];
                } else if constexpr (TRANS == 2) {
                    const float4 mq = *reinterpret_cast<const float4*>(rmeta[wm * 32 + tr]);
                    const float wr = __builtin_amdgcn_exp2f(s - rm[r]) * fmaf(c1, mq.y, fmaf(c0, mq.x, ra[r]));
                    const float wc = __builtin_amdgcn_exp2f(s - kmx) * fmaf((float)((rcl[r] >> 1) & 1), k2, fmaf((float)(rcl[r] & 1), k1, k0));
                    w = wr + wc;
                } else if constexpr (TRANS == 0) {
                    const float4 mq = *reinterpret_cast<const float4*>(rmeta[wm * 32 + tr]);
                    w = __builtin_amdgcn_exp2f(s - rm[r]) * fmaf(c1, mq.y, fmaf(c0, mq.x, ra[r]));
                }
                else    // (the row's class bits as 0 / 1 factors of two multiply-adds: a compare + select pair costs 5 cycles of matrix-pipe throughput)
                    w = __builtin_amdgcn_exp2f(s - cmx) * fmaf((float)((rcl[r] >> 1) & 1), c2, fmaf((float)(rcl[r] & 1), c1, c0));
                wrow[tr * WP + l32] = w;
            }
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = row0 + wm * 32 + tr;
            float w = 0.f;
            if (colok && row < n2 && col != row) {
                const float s = acc[r];
                const int pr = row < n_half ? row + n_half : row - n_half;
                if constexpr (TRANS == 3) {                   // (row maximum = cm[0] = M, the coefficients arrive scaled; g_all' and the class from LDS)
                    const float4 mq = *reinterpret_cast<const float4*>(rmeta[wm * 32 + tr]);
                    const float* cm = cmeta[lc];
                    const uint8_t rc = rcls[wm * 32 + tr];
                    w = __builtin_amdgcn_exp2f(s - gmax) * (mq.w + ((cc & 1) ? mq.x : 0.f) + ((cc & 2) ? mq.y : 0.f) + (col == pr ? mq.z : 0.f) +
                                                            cm[1] + ((rc & 1) ? cm[2] : 0.f) + ((rc & 2) ? cm[3] : 0.f) + (col == pr ? cm[4] : 0.f));
                } else if constexpr (TRANS == 2) {
                    const float4 mq = *reinterpret_cast<const float4*>(rmeta[wm * 32 + tr]);
                    const float* cm = cmeta[lc];
                    w = __builtin_amdgcn_exp2f(s - rm[r]) * (ra[r] + ((cc & 1) ? mq.x : 0.f) + ((cc & 2) ? mq.y : 0.f) +
                                             (col == pr ? mq.z : 0.f)) +
                        __builtin_amdgcn_exp2f(s - cm[0]) * (cm[1] + ((rcl[r] & 1) ? cm[2] : 0.f) + ((rcl[r] & 2) ? cm[3] : 0.f) +
                                             (col == pr ? cm[4] : 0.f));
                } else if constexpr (TRANS == 0) {
                    const float4 mq = *reinterpret_cast<const float4*>(rmeta[wm * 32 + tr]);
                    w = __builtin_amdgcn_exp2f(s - rm[r]) * (ra[r] + ((cc & 1) ? mq.x : 0.f) + ((cc & 2) ? mq.y : 0.f) +
                                             (col == pr ? mq.z : 0.f));
                } else {
                    const float* cm = cmeta[lc];
                    w = __builtin_amdgcn_exp2f(s - cm[0]) * (cm[1] + ((rcl[r] & 1) ? cm[2] : 0.f) + ((rcl[r] & 2) ? cm[3] : 0.f) +
                                             (col == pr ? cm[4] : 0.f));
                }
            }
            wrow[tr * WP + l32] = w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes are visible to it
        // second product: out[32 rows][DIM] += W[32 x 32] * Fcol[32 x DIM]; A = W (rows x k = tile cols), on the bf16 pipe as
        // well (round 4): the lane's eight W values of a k-step are cut here (they exist only now), the tile's features are the
        // planes staged for the first product - read through the TRANSPOSING LDS read (ds_read_b64_tr_b16: the planes are
        // [column = k][feature = n] rows, the fragment wants n per lane and k along its registers; a lane names row q4 of its
        // 16-lane group's 4 x 16 block and gets the block's column i16 - conv_cube2.hip pair_wgrad_kernel has the same read)
        {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            const int i16 = lane & 15, g16 = (lane >> 4) & 1, q4 = i16 >> 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
                const float4 a0 = *reinterpret_cast<const float4*>(wrow + l32 * WP + 16 * ks + 8 * h);
                const float4 a1 = *reinterpret_cast<const float4*>(wrow + l32 * WP + 16 * ks + 8 * h + 4);
                v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                bf16x8 aw[3];
                ucl_cut8(v, aw);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bf16x8 bf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned char* bp = colb + pl * SP::PLANE + (wn * 32 + 16 * ks + 8 * h + q4) * SP::PITCH +
                                                  (32 * j + 16 * g16 + 4 * (i16 & 3)) * 2;
                        const bf16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(bp));
                        const bf16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(bp + 4 * SP::PITCH));
                        bf[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        out[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[PA[pr]], bf[PB[pr]], out[j], 0, 0, 0);
                }
            }
        }
        if (DB) __syncthreads();                           // the next tile is in place; this one's buffer is free
    }
    if (TWO_LEVEL) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[j][r] += tot[j][r];
    }
    // sum the two column halves and write
    __syncthreads();                                      // the arena / the column buffers are free
    static_assert(TRANS < 2 || 2 * UB * LD * 4 <= 2 * SP::BYTES, "epilogue does not fit the column buffers");
    float* const outm = TRANS >= 2 ? reinterpret_cast<float*>(&colb_[0][0]) : arena;      // [2][UB][LD]
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
            const float v = (outm[tr * LD + k] + outm[(UB + tr) * LD + k]) * (inv_T / fs);
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

// range[0] = the largest row maximum (base-2 units), range[1] = 1 when all row maxima lie within 2^16 of it (ucl_bwd_kernel TRANS == 3)
__global__ __launch_bounds__(1024) void ucl_rowmax_range_kernel(const float* rowmax, int n2, float* range) {
    __shared__ float smx[16], smn[16];
    float mx = -INFINITY, mn = INFINITY;
    for (int i = threadIdx.x; i < n2; i += 1024) { const float v = rowmax[i]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
    mx = wave_max(mx); mn = -wave_max(-mn);
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) { mx = fmaxf(mx, smx[k]); mn = fminf(mn, smn[k]); }
        const float L2E = 1.4426950408889634f;
        range[0] = mx * L2E;
        range[1] = ((mx - mn) * L2E <= 16.f && mx == mx && mn == mn) ? 1.f : 0.f;      // (NaN anywhere: the general kernel)
    }
}

extern "C" int mi_ucl_rowsums_bwd(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T,
                                  const float* rowmax, const float* g_all, const float* g_pos, const float* g_other,
                                  const float* g_pair, float* dfeat, mi_stream_t stream) {
    if (!feat || !cls || !rowmax || !g_all || !g_pos || !g_other || !g_pair || !dfeat || n2 <= 0 || (n2 & 1)) return MI_E_ARG;
    const dim3 grid((n2 + UB - 1) / UB), block(256);
    hipStream_t s = (hipStream_t)stream;
    // round 6: one launch forms both terms from one similarity tile (TRANS == 2); MI_UCL_BWD_SPLIT=1: the two launches of rounds 2-5
    // (the 64-wide form stays split: its merged kernel would hold both sides' metadata at one resident wave less per SIMD)
    const bool split = getenv("MI_UCL_BWD_SPLIT") != nullptr;
#define MI_UCL_BWD(D_)                                                                                              \
    hipLaunchKernelGGL((ucl_bwd_kernel<D_, 0>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, \
                       g_other, g_pair, dfeat, 0);                                                                  \
    hipLaunchKernelGGL((ucl_bwd_kernel<D_, 1>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, \
                       g_other, g_pair, dfeat, 1)
    if (dim == 32 && !split) {
        hipLaunchKernelGGL((ucl_bwd_kernel<32, 2>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, g_other, g_pair,
                           dfeat, 0, (const float*)nullptr);
    } else if (dim == 32) { MI_UCL_BWD(32); }
    else if (dim == 64) { MI_UCL_BWD(64); }
    else return MI_E_UNSUPPORTED;
#undef MI_UCL_BWD
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ... with a two-float scratch `range`: the one-exponential form where the row maxima allow it (ucl_bwd_kernel TRANS == 3), decided on the
// device - a range kernel, then both forms' launches, of which one returns at once
extern "C" int mi_ucl_rowsums_bwd_ranged(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T, const float* rowmax,
                                         const float* g_all, const float* g_pos, const float* g_other, const float* g_pair, float* dfeat,
                                         float* range, mi_stream_t stream) {
    if (!range || dim != 32 || getenv("MI_UCL_BWD_SPLIT") || getenv("MI_UCL_BWD_TWO_EXP"))
        return mi_ucl_rowsums_bwd(feat, cls, n2, dim, inv_T, rowmax, g_all, g_pos, g_other, g_pair, dfeat, stream);
    if (!feat || !cls || !rowmax || !g_all || !g_pos || !g_other || !g_pair || !dfeat || n2 <= 0 || (n2 & 1)) return MI_E_ARG;
    const dim3 grid((n2 + UB - 1) / UB), block(256);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ucl_rowmax_range_kernel, dim3(1), dim3(1024), 0, s, rowmax, n2, range);
    hipLaunchKernelGGL((ucl_bwd_kernel<32, 3>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, g_other, g_pair, dfeat,
                       0, (const float*)range);
    hipLaunchKernelGGL((ucl_bwd_kernel<32, 2>), grid, block, 0, s, feat, cls, n2, n2 / 2, inv_T, rowmax, g_all, g_pos, g_other, g_pair, dfeat,
                       0, (const float*)range);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
