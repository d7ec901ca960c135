// 3-D convolution as implicit GEMM on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32:
// f32 in, f32 accumulate - bit-for-bit an fmaf chain, so parity with the fp32 reference holds).
//
// Replaces the nn.Conv3d / nn.Linear calls of the reference encoders
// (cet_pick/models/networks/moco_encoder_3d.py:40-84,156-236) in forward, data-gradient and
// weight-gradient form.  Activations are channels-last (N,D,H,W,C); weights live as
// [tap][Cin][Cout] (tap = (kd,kh,kw) flattened) - the Python side exposes that storage to
// state_dict() as a permuted (Cout,Cin,kd,kh,kw) view, so checkpoints keep the reference layout.
//
//   FWD   Y[m,co]  = sum_{tap,ci} X[src(m,tap),ci] * W[tap][ci][co]          m = output voxel
//   DGRAD dX[m,ci] = sum_{tap,co} dY[srcT(m,tap),co] * W[tap][ci][co]        m = input voxel
//   WGRAD dW[tap][ci][co] = sum_m X[src(m,tap),ci] * dY[m,co]                m = output voxel
//
// One 256-thread workgroup (4 waves, WM x WN) owns a BM x BN tile of the GEMM output and walks the
// reduction in 16-deep slices, double-buffered in LDS with register-staged prefetch (one barrier
// per slice).  Each operand is kept in LDS in the orientation its global layout is contiguous in:
//   "RowK" [row][k] (+4 pad): fragments by 2 x ds_read_b128 per 32 rows   (im2col rows, W in DGRAD)
//   "KRow" [k][row]:          fragments by 8 x ds_read_b32  per 32 rows   (W in FWD, dY/X in WGRAD)
// Within a slice lane-half h of the wave owns k = 8h..8h+7 for BOTH operands, which turns the
// 32x32x2 MFMA's (k = lane>>5) operand map into contiguous LDS reads; the reduction order inside a
// slice is therefore permuted (irrelevant beyond fp32 rounding).
// Small-M layers are split along the reduction (grid.z) into fp32 slabs, summed by a second
// kernel that also applies the epilogue (deterministic, no atomics).
#include "common.h"
#include <algorithm>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 16;
constexpr int LDK = BK + 4;   // RowK row stride in floats (conflict-free ds_read_b128)
constexpr int NTHREADS = 256;

struct ConvParams {
    const float* a_src;   // gathered tensor: X (FWD, WGRAD) or dY (DGRAD)
    const float* b_src;   // W (FWD, DGRAD) or dY (WGRAD)
    float* out;           // Y / dX / dW, or the split-K slabs
    const float* res;     // epilogue: out = act(acc + res)         (may be null)
    const float* mask;    // epilogue: out *= (mask > 0)            (may be null)
    int relu;
    // gathered tensor grid / channels, row grid (see header comment)
    int N, Dg, Hg, Wg, Cg;
    int Dr, Hr, Wr;
    int lDr, lHr, lWr;    // log2 of the row grid dims, or -1 when not powers of two
    int kd, kh, kw, stride, pad;
    int Ci, Co;           // conv channels (weights are [tap][Ci][Co])
    long M;               // GEMM rows
    int Ncols;            // GEMM cols
    int nk;               // 16-deep reduction slices in total
    int nk_per_split;
    long slab_stride;     // elements between split-K slabs (0: direct epilogue)
};

__device__ __forceinline__ void decode_row(const ConvParams& p, long m, int& n, int& z, int& y, int& x) {
    if (p.lWr >= 0) {
        x = (int)(m & (p.Wr - 1)); m >>= p.lWr;
        y = (int)(m & (p.Hr - 1)); m >>= p.lHr;
        z = (int)(m & (p.Dr - 1)); n = (int)(m >> p.lDr);
    } else {
        x = (int)(m % p.Wr); m /= p.Wr;
        y = (int)(m % p.Hr); m /= p.Hr;
        z = (int)(m % p.Dr); n = (int)(m / p.Dr);
    }
}

// forward gather: source voxel of (row voxel, tap) in the gathered (input) grid, or -1
__device__ __forceinline__ long src_fwd(const ConvParams& p, int n, int z, int y, int x, int a, int b, int c) {
    int zi = z * p.stride - p.pad + a, yi = y * p.stride - p.pad + b, xi = x * p.stride - p.pad + c;
    bool ok = (unsigned)zi < (unsigned)p.Dg && (unsigned)yi < (unsigned)p.Hg && (unsigned)xi < (unsigned)p.Wg;
    return ok ? ((((long)n * p.Dg + zi) * p.Hg + yi) * p.Wg + xi) : -1;
}
// transposed gather (DGRAD): output voxel that input voxel (z,y,x) feeds through tap (a,b,c)
__device__ __forceinline__ long src_bwd(const ConvParams& p, int n, int z, int y, int x, int a, int b, int c) {
    int tz = z + p.pad - a, ty = y + p.pad - b, tx = x + p.pad - c;
    if ((tz | ty | tx) < 0) return -1;
    int s = p.stride;
    if (s == 2) {
        if ((tz | ty | tx) & 1) return -1;
        tz >>= 1; ty >>= 1; tx >>= 1;
    } else if (s != 1) {
        if (tz % s || ty % s || tx % s) return -1;
        tz /= s; ty /= s; tx /= s;
    }
    bool ok = tz < p.Dg && ty < p.Hg && tx < p.Wg;
    return ok ? ((((long)n * p.Dg + tz) * p.Hg + ty) * p.Wg + tx) : -1;
}

__device__ __forceinline__ float4 ld4(const float* ptr) { return *reinterpret_cast<const float4*>(ptr); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// STEM: Cin == 1 (the 7x7x7 stride-2 stem, moco_encoder_3d.py:163-169): the reduction index is
// the tap itself and each of a chunk's 4 taps is gathered separately.
template <int MODE, int BM, int BN, int WM, int WN, bool STEM>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_kernel(ConvParams p) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    static_assert(MT >= 1 && NT >= 1, "wave tile >= 32x32");
    constexpr bool A_ROWK = (MODE != MODE_WGRAD);
    constexpr bool B_ROWK = (MODE == MODE_DGRAD);
    constexpr int A_ELEMS = A_ROWK ? BM * LDK : BK * BM;
    constexpr int B_ELEMS = B_ROWK ? BN * LDK : BK * BN;
    constexpr int A_CH = BM * 4 / NTHREADS;          // 16-B chunks per thread per slice
    constexpr int B_CH = BN * 4 / NTHREADS;
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for 256 threads");

    __shared__ __attribute__((aligned(16))) float lds[2 * (A_ELEMS + B_ELEMS)];
    constexpr int STAGE = A_ELEMS + B_ELEMS;         // buffer b: A at lds + b*STAGE, B right after

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l32 = lane & 31;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int kt0 = blockIdx.z * p.nk_per_split;
    const int kt1 = min(kt0 + p.nk_per_split, p.nk);
    const int taps_hw = p.kh * p.kw;
    const int taps = p.kd * taps_hw;

    // ---- per-thread staging state ------------------------------------------------------------
    // A operand
    int a_n[A_CH], a_z[A_CH], a_y[A_CH], a_x[A_CH];   // RowK: row voxel ; KRow (WGRAD): tap a,b,c + ci
    bool a_ok[A_CH];
    int a_lds[A_CH];
    if (A_ROWK) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int q = tid + i * NTHREADS;
            int row = q >> 2, c = q & 3;
            long m = m0 + row;
            a_ok[i] = m < p.M;
            decode_row(p, a_ok[i] ? m : 0, a_n[i], a_z[i], a_y[i], a_x[i]);
            a_lds[i] = row * LDK + 4 * c;
        }
    } else {
        // WGRAD: GEMM rows are (tap, ci); this thread's 4 rows share one tap
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int q = tid + i * NTHREADS;
            int kk = q / (BM / 4), j = q % (BM / 4);
            long row = m0 + 4 * j;
            a_ok[i] = row < p.M;
            int tap = STEM ? (int)row : (int)(row / p.Ci);   // STEM: rows ARE taps (4 consecutive)
            a_n[i] = STEM ? 0 : (int)(row % p.Ci);           // ci of the first of the 4 rows
            a_z[i] = tap / taps_hw; a_y[i] = (tap / p.kw) % p.kh; a_x[i] = tap % p.kw;
            a_lds[i] = kk * BM + 4 * j;
        }
    }
    // B operand
    int b_lds[B_CH];
    int b_row[B_CH], b_col[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int q = tid + i * NTHREADS;
        if (B_ROWK) {            // DGRAD: LDS [n][k]; global W[tap][ci = n][co = k]
            b_row[i] = q >> 2; b_col[i] = (q & 3) * 4;
            b_lds[i] = b_row[i] * LDK + b_col[i];
        } else {                 // LDS [k][n]; global rows k, cols n contiguous
            b_row[i] = q / (BN / 4); b_col[i] = (q % (BN / 4)) * 4;
            b_lds[i] = b_row[i] * BN + b_col[i];
        }
    }

    float4 a_reg[A_CH], b_reg[B_CH];

    auto load_tile = [&](int kt) {
        if (MODE == MODE_WGRAD) {
            // reduction index = output voxel m = kt*16 + kk
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int q = tid + i * NTHREADS;
                int kk = q / (BM / 4);
                long mv = (long)kt * BK + kk;
                float4 v = zero4();
                if (a_ok[i] && mv < (long)p.N * p.Dr * p.Hr * p.Wr) {
                    int n, z, y, x;
                    decode_row(p, mv, n, z, y, x);
                    if (STEM) {
                        int j = q % (BM / 4);
                        float e[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            int tap = (int)m0 + 4 * j + u;
                            e[u] = 0.f;
                            if (tap < taps) {
                                long s = src_fwd(p, n, z, y, x, tap / taps_hw, (tap / p.kw) % p.kh, tap % p.kw);
                                if (s >= 0) e[u] = p.a_src[s];
                            }
                        }
                        v = make_float4(e[0], e[1], e[2], e[3]);
                    } else {
                        long s = src_fwd(p, n, z, y, x, a_z[i], a_y[i], a_x[i]);
                        if (s >= 0) v = ld4(p.a_src + s * p.Cg + a_n[i]);
                    }
                }
                a_reg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                long mv = (long)kt * BK + b_row[i];
                int col = n0 + b_col[i];
                float4 v = zero4();
                if (mv < (long)p.N * p.Dr * p.Hr * p.Wr && col < p.Ncols) v = ld4(p.b_src + mv * p.Co + col);
                b_reg[i] = v;
            }
        } else if (STEM) {
            // FWD stem: slice kt covers taps kt*16 .. kt*16+15
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int q = tid + i * NTHREADS;
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                if (a_ok[i]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int tap = kt * BK + 4 * (q & 3) + u;
                        if (tap < taps) {
                            long s = src_fwd(p, a_n[i], a_z[i], a_y[i], a_x[i], tap / taps_hw,
                                             (tap / p.kw) % p.kh, tap % p.kw);
                            if (s >= 0) e[u] = p.a_src[s];
                        }
                    }
                }
                a_reg[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                int tap = kt * BK + b_row[i];
                int col = n0 + b_col[i];
                float4 v = zero4();
                if (tap < taps && col < p.Ncols) v = ld4(p.b_src + (long)tap * p.Co + col);
                b_reg[i] = v;
            }
        } else {
            const int cred = (MODE == MODE_FWD) ? p.Ci : p.Co;    // reduction channels
            const int per_tap = cred / BK;
            const int tap = kt / per_tap;
            const int c0 = (kt - tap * per_tap) * BK;
            const int ta = tap / taps_hw, tb = (tap / p.kw) % p.kh, tc = tap % p.kw;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                float4 v = zero4();
                if (a_ok[i]) {
                    long s = (MODE == MODE_FWD) ? src_fwd(p, a_n[i], a_z[i], a_y[i], a_x[i], ta, tb, tc)
                                                : src_bwd(p, a_n[i], a_z[i], a_y[i], a_x[i], ta, tb, tc);
                    int q = tid + i * NTHREADS;
                    if (s >= 0) v = ld4(p.a_src + s * p.Cg + c0 + 4 * (q & 3));
                }
                a_reg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                float4 v = zero4();
                if (MODE == MODE_FWD) {
                    int col = n0 + b_col[i];
                    if (col < p.Ncols) v = ld4(p.b_src + ((long)tap * p.Ci + c0 + b_row[i]) * p.Co + col);
                } else {
                    int ci = n0 + b_row[i];
                    if (ci < p.Ncols) v = ld4(p.b_src + ((long)tap * p.Ci + ci) * p.Co + c0 + b_col[i]);
                }
                b_reg[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<float4*>(lds + buf * STAGE + a_lds[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<float4*>(lds + buf * STAGE + A_ELEMS + b_lds[i]) = b_reg[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();

    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1 < kt1);
        if (more) load_tile(kt + 1);

        float af[MT][8], bf[NT][8];
        const float* Ab = lds + buf * STAGE;
        const float* Bb = Ab + A_ELEMS;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * WTM + i * 32 + l32;
            if (A_ROWK) {
                float4 v0 = *reinterpret_cast<const float4*>(Ab + r * LDK + h * 8);
                float4 v1 = *reinterpret_cast<const float4*>(Ab + r * LDK + h * 8 + 4);
                af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
                af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) af[i][t] = Ab[(h * 8 + t) * BM + r];
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = wn * WTN + j * 32 + l32;
            if (B_ROWK) {
                float4 v0 = *reinterpret_cast<const float4*>(Bb + c * LDK + h * 8);
                float4 v1 = *reinterpret_cast<const float4*>(Bb + c * LDK + h * 8 + 4);
                bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
                bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) bf[j][t] = Bb[(h * 8 + t) * BN + c];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);

        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* outp = p.out + (long)blockIdx.z * p.slab_stride;
    const bool direct = (p.slab_stride == 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WTN + j * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M && col < p.Ncols) {
                    const long o = row * p.Ncols + col;
                    float v = acc[i][j][r];
                    if (direct) {
                        if (p.res) v += p.res[o];
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (p.mask) v = (p.mask[o] > 0.f) ? v : 0.f;
                    }
                    outp[o] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int n_slabs,
                                                           long slab_stride, float* out,
                                                           const float* res, const float* mask,
                                                           int relu, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 s = ld4(slabs + 4 * i);
        for (int z = 1; z < n_slabs; ++z) {
            float4 v = ld4(slabs + (long)z * slab_stride + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (res) { float4 v = ld4(res + 4 * i); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        if (mask) {
            float4 v = ld4(mask + 4 * i);
            s.x = v.x > 0.f ? s.x : 0.f; s.y = v.y > 0.f ? s.y : 0.f;
            s.z = v.z > 0.f ? s.z : 0.f; s.w = v.w > 0.f ? s.w : 0.f;
        }
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct Plan { int bm, bn, splits, nk_per_split; };

Plan make_plan(long M, int Ncols, int nk, int force_splits) {
    Plan pl;
    pl.bn = 64;
    pl.bm = (M >= 16384) ? 128 : 64;
    long tiles = ((M + pl.bm - 1) / pl.bm) * ((Ncols + pl.bn - 1) / pl.bn);
    int splits = 1;
    if (force_splits > 0) {
        splits = force_splits;
    } else if (tiles < 384) {
        splits = (int)((512 + tiles - 1) / tiles);
        int max_splits = nk / 8 > 0 ? nk / 8 : 1;     // at least 8 slices per split
        if (splits > max_splits) splits = max_splits;
        if (splits > 64) splits = 64;
    }
    if (splits < 1) splits = 1;
    pl.nk_per_split = (nk + splits - 1) / splits;
    pl.splits = (nk + pl.nk_per_split - 1) / pl.nk_per_split;
    return pl;
}

template <int MODE, bool STEM>
int launch_mode(ConvParams p, const Plan& pl, hipStream_t s) {
    dim3 grid((unsigned)((p.M + pl.bm - 1) / pl.bm), (unsigned)((p.Ncols + pl.bn - 1) / pl.bn), pl.splits);
    if (pl.bm == 128)
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 128, 64, 2, 2, STEM>), grid, dim3(NTHREADS), 0, s, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 64, 64, 2, 2, STEM>), grid, dim3(NTHREADS), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

struct Geom {
    int N, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, k, stride, pad;
};

bool geom_ok(const Geom& g) {
    if (g.N <= 0 || g.Di <= 0 || g.Hi <= 0 || g.Wi <= 0 || g.Ci <= 0 || g.Co <= 0) return false;
    if (g.k <= 0 || g.stride <= 0 || g.pad < 0) return false;
    if ((g.Ci % BK && g.Ci != 1) || g.Co % BK) return false;
    return true;
}
Geom make_geom(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
    Geom g{N, Di, Hi, Wi, Ci, 0, 0, 0, Co, k, stride, pad};
    g.Do = (Di + 2 * pad - k) / stride + 1;
    g.Ho = (Hi + 2 * pad - k) / stride + 1;
    g.Wo = (Wi + 2 * pad - k) / stride + 1;
    return g;
}

int run_conv(int mode, const Geom& g, const float* a_src, const float* b_src, float* out,
             const float* res, const float* mask, int relu, void* ws, size_t ws_bytes,
             int force_splits, hipStream_t s) {
    ConvParams p = {};
    p.a_src = a_src; p.b_src = b_src; p.res = res; p.mask = mask; p.relu = relu;
    p.N = g.N; p.kd = p.kh = p.kw = g.k; p.stride = g.stride; p.pad = g.pad; p.Ci = g.Ci; p.Co = g.Co;
    const int taps = g.k * g.k * g.k;
    const bool stem = (g.Ci == 1);
    const long Mout = (long)g.N * g.Do * g.Ho * g.Wo, Min = (long)g.N * g.Di * g.Hi * g.Wi;
    if (mode == MODE_FWD) {
        p.Dg = g.Di; p.Hg = g.Hi; p.Wg = g.Wi; p.Cg = g.Ci;
        p.Dr = g.Do; p.Hr = g.Ho; p.Wr = g.Wo;
        p.M = Mout; p.Ncols = g.Co; p.nk = stem ? (taps + BK - 1) / BK : taps * g.Ci / BK;
    } else if (mode == MODE_DGRAD) {
        if (stem) return MI_E_UNSUPPORTED;    // the stem's input is the image: no data gradient
        p.Dg = g.Do; p.Hg = g.Ho; p.Wg = g.Wo; p.Cg = g.Co;
        p.Dr = g.Di; p.Hr = g.Hi; p.Wr = g.Wi;
        p.M = Min; p.Ncols = g.Ci; p.nk = taps * g.Co / BK;
    } else {
        p.Dg = g.Di; p.Hg = g.Hi; p.Wg = g.Wi; p.Cg = g.Ci;
        p.Dr = g.Do; p.Hr = g.Ho; p.Wr = g.Wo;
        p.M = (long)taps * g.Ci; p.Ncols = g.Co; p.nk = (int)((Mout + BK - 1) / BK);
    }
    p.lDr = ilog2_exact(p.Dr); p.lHr = ilog2_exact(p.Hr); p.lWr = ilog2_exact(p.Wr);
    if (p.lDr < 0 || p.lHr < 0 || p.lWr < 0) p.lDr = p.lHr = p.lWr = -1;
    Plan pl = make_plan(p.M, p.Ncols, p.nk, force_splits);
    p.nk_per_split = pl.nk_per_split;
    const long out_elems = p.M * p.Ncols;
    if (pl.splits > 1) {
        size_t need = sizeof(float) * (size_t)out_elems * pl.splits;
        if (!ws || ws_bytes < need) {
            // not enough scratch for slabs: fall back to the unsplit schedule (same result)
            pl.splits = 1; pl.nk_per_split = p.nk; p.nk_per_split = p.nk;
        }
    }
    if (pl.splits > 1) {
        p.out = (float*)ws; p.slab_stride = out_elems;
    } else {
        p.out = out; p.slab_stride = 0;
    }
    int rc;
    if (mode == MODE_FWD) rc = stem ? launch_mode<MODE_FWD, true>(p, pl, s) : launch_mode<MODE_FWD, false>(p, pl, s);
    else if (mode == MODE_DGRAD) rc = launch_mode<MODE_DGRAD, false>(p, pl, s);
    else rc = stem ? launch_mode<MODE_WGRAD, true>(p, pl, s) : launch_mode<MODE_WGRAD, false>(p, pl, s);
    if (rc) return rc;
    if (pl.splits > 1) {
        long n4 = out_elems / 4;
        int blocks = (int)std::min<long>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws,
                           pl.splits, out_elems, out, res, mask, relu, n4);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}

}  // namespace

extern "C" size_t mi_conv3d_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int k,
                                            int stride, int pad) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!geom_ok(g)) return 0;
    const int taps = k * k * k;
    const long Mout = (long)N * g.Do * g.Ho * g.Wo, Min = (long)N * Di * Hi * Wi;
    size_t best = 0;
    struct { long M; int Nc; int nk; } cases[3] = {
        {Mout, Co, Ci == 1 ? (taps + BK - 1) / BK : taps * Ci / BK}, {Min, Ci, taps * Co / BK},
        {(long)taps * Ci, Co, (int)((Mout + BK - 1) / BK)}};
    for (auto& c : cases) {
        Plan pl = make_plan(c.M, c.Nc, c.nk, 0);
        size_t b = pl.splits > 1 ? sizeof(float) * (size_t)c.M * c.Nc * pl.splits : 0;
        best = std::max(best, b);
    }
    return best + 256;
}

extern "C" int mi_conv3d_fwd_f32(const float* x, const float* w, float* y, const float* res,
                                 int relu, int N, int Di, int Hi, int Wi, int Ci, int Co, int k,
                                 int stride, int pad, void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!x || !w || !y || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, res, nullptr, relu, ws, ws_bytes, 0, (hipStream_t)stream);
}

extern "C" int mi_conv3d_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                                   const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co,
                                   int k, int stride, int pad, void* ws, size_t ws_bytes,
                                   mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!dy || !w || !dx || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_DGRAD, g, dy, w, dx, res, mask, 0, ws, ws_bytes, 0, (hipStream_t)stream);
}

extern "C" int mi_conv3d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int Di,
                                   int Hi, int Wi, int Ci, int Co, int k, int stride, int pad,
                                   void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!x || !dy || !dw || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_WGRAD, g, x, dy, dw, nullptr, nullptr, 0, ws, ws_bytes, 0, (hipStream_t)stream);
}
