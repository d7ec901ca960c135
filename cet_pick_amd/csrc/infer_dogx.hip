// DoG particle picker, last stage fused: x pass of BOTH Gaussians + DoG + border zeroing + `_nms_xy` (3x3) + the
// statistics of the positive survivors + candidate compaction, in one pass over the two y-filtered volumes.
//
// Replaces (reference, cet_pick/...): the x axis of the two scipy.ndimage.gaussian_filter calls and
// `rec_i = gaussian(s2) - gaussian(s1)`, the border zeroing, `_nms_xy(.., kernel=3)` and the `mean + 0.5 std` statistics
// of utils/image.py:152-179.
//
// Before: x pass (read + write) per Gaussian, then the DoG/NMS march reading both results: 6 volume passes, 449 us on
// 256x512x512.  Here the two x-filtered rows exist only in registers: 2 volume reads, candidates out.
//   * a WAVE owns a row segment of 512 x (64 lanes x 8 consecutive x) of one z plane and marches down a chunk of y;
//   * the two input rows (+ halo R, scipy 'reflect') are staged in a wave-private LDS tile, 8-way interleaved so the
//     2R+8 window reads of a lane are conflict-free `ds_read_b32` at (lane base + immediate); the wave alone writes and
//     reads its tile (LDS operations of one wave execute in order): no barrier anywhere; the next row's global loads
//     are in flight (registers) while the current row is filtered;
//   * taps are symmetric: w[|d|] as scalar operands, 21 + 13 of them for sigma (5, 3) - the 82 two-sided taps of a
//     two-sigma kernel do not fit the scalar register file;
//   * x neighbours of the 3x3 window come from the adjacent lanes by DPP wave shifts, y neighbours from a two-row
//     register ring; rows and planes inside the zeroed border are not filtered at all;
//   * survivors leave through an LDS ring 64 at a time into a candidate segment private to the wave (no global counter),
//     fp64 (count, sum, sum of squares) partials go out once per wave.
// VALU-bound by the direct filter: (R1 + R2 + 2) multiply-adds per voxel = 66 wave instructions per 64 voxels for
// sigma (3, 5), 56 us of issue at the vector peak; algorithmic bytes 8 B per voxel (two reads).
// hipcc-flags: -fno-slp-vectorize
// (the SLP vectoriser packs the filter's FMA chains into v_pk_fma_f32 / v_pk_add_f32 register pairs: no more FLOP per
// clock on this part, ~290 v_mov per row to build the pairs, and enough extra live registers to spill in the row loop)
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int XL = 8, SEGW = 64 * XL, DWPB = 4, DNT = 64 * DWPB;
constexpr int DRING = 512;

template <int R>
struct SymTaps { float w[R + 1]; };                 // w[t] = tap at distance t from the centre

__device__ __forceinline__ int reflect1(int i, int n) {      // one reflection is enough here (R < n)
    if (i < 0) return -i - 1;
    if (i >= n) return 2 * n - 1 - i;
    return i;
}
__device__ __forceinline__ float dx_from_lower(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dx_from_upper(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float mx3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int R>
struct RowRegs {
    float c[XL];       // centre elements x = 8 lane .. + 7
    float h;           // halo element of lanes < 2R
};

// LDS tile of one row: element e (x = e - R) at (e & 7) * P + (e >> 3)
template <int R>
struct Tile {
    static constexpr int P = 64 + (2 * R + 7) / 8;
    float v[8 * P];
};

template <int R>
__device__ __forceinline__ void fetch_row(RowRegs<R>& rr, const float* __restrict__ row, int W, int lane) {
    const int x0 = XL * lane;
    if (x0 + XL <= W) {
        const float4 a = *reinterpret_cast<const float4*>(row + x0);
        const float4 b = *reinterpret_cast<const float4*>(row + x0 + 4);
        rr.c[0] = a.x; rr.c[1] = a.y; rr.c[2] = a.z; rr.c[3] = a.w;
        rr.c[4] = b.x; rr.c[5] = b.y; rr.c[6] = b.z; rr.c[7] = b.w;
    } else if (x0 < W + R) {                         // past the right edge but inside the reflected halo
#pragma unroll
        for (int k = 0; k < XL; ++k) rr.c[k] = row[reflect1(x0 + k, W)];
    }
    if (lane < 2 * R) {
        const int x = lane < R ? lane - R : SEGW + lane - R;      // left halo / right halo of the 512-wide segment
        rr.h = row[reflect1(x < W + R ? x : W - 1, W)];
    }
}

template <int R>
__device__ __forceinline__ void stage_row(const RowRegs<R>& rr, Tile<R>& t, int lane) {
#pragma unroll
    for (int k = 0; k < XL; ++k) {
        constexpr int P = Tile<R>::P;
        const int e0 = R + k;                        // element of lane 0; lane adds 8 -> +1 in the sub-array
        t.v[(e0 & 7) * P + (e0 >> 3) + lane] = rr.c[k];
    }
    if (lane < 2 * R) {
        const int e = lane < R ? lane : SEGW + lane;
        t.v[(e & 7) * Tile<R>::P + (e >> 3)] = rr.h;
    }
}

// g[i] = sum_d w[|d|] * in[x_i + d] for the lane's 8 outputs
template <int R>
__device__ __forceinline__ void filter_row(const Tile<R>& t, const SymTaps<R>& w, int lane, float g[XL]) {
    constexpr int P = Tile<R>::P;
    float val[2 * R + XL];
#pragma unroll
    for (int j = 0; j < 2 * R + XL; ++j) val[j] = t.v[(j & 7) * P + (j >> 3) + lane];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        float acc = w.w[0] * val[i + R];
#pragma unroll
        for (int d = 1; d <= R; ++d) acc = fmaf(w.w[d], val[i + R - d] + val[i + R + d], acc);
        g[i] = acc;
    }
}

template <int R1, int R2>
__global__ __launch_bounds__(DNT, 2) void dogx_nms_kernel(DogxParams p, SymTaps<R1> w1, SymTaps<R2> w2) {
    __shared__ Tile<R1> tile1[DWPB];
    __shared__ Tile<R2> tile2[DWPB];
    __shared__ uint2 ring_all[DWPB][DRING];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long g = (long)blockIdx.x * DWPB + wv;                  // wave = (plane z, y chunk)
    if (g >= (long)p.D * p.n_ychunks) return;
    const int z = (int)(g / p.n_ychunks), yc = (int)(g % p.n_ychunks);
    const int ya = yc * p.ychunk, yb = min(ya + p.ychunk, p.H);
    const float NEG = -INFINITY;
    const int x0 = XL * lane;
    const bool lane_in = x0 < p.W;                                // W % 8 == 0: the lane's 8 voxels are all inside
    uint2* seg_base = p.cands + (size_t)g * p.seg_cap;
    uint2* ring = ring_all[wv];
    unsigned cnt = 0, flushed = 0;
    double st_n = 0, st_s = 0, st_ss = 0;

    const bool plane_zero = z < p.bz || z >= p.D - p.bz;          // whole plane inside the zeroed z border
    if (!plane_zero || p.nms_out) {
        const long plane = (long)z * p.H * p.W;
        const float* y1p = p.y1 + plane;
        const float* y2p = p.y2 + plane;
        auto row_live = [&](int r) { return !plane_zero && r >= p.by && r < p.H - p.by && r >= 0 && r < p.H; };
        bool xzero[XL];
#pragma unroll
        for (int i = 0; i < XL; ++i) xzero[i] = (x0 + i < p.bx) || (x0 + i >= p.W - p.bx);

        RowRegs<R1> r1;
        RowRegs<R2> r2;
        float xm_pp[XL], xm_p[XL], c_p[XL];
#pragma unroll
        for (int i = 0; i < XL; ++i) { xm_pp[i] = NEG; xm_p[i] = NEG; c_p[i] = 0.f; }
        if (row_live(ya - 1)) {
            fetch_row<R1>(r1, y1p + (long)(ya - 1) * p.W, p.W, lane);
            fetch_row<R2>(r2, y2p + (long)(ya - 1) * p.W, p.W, lane);
        }
        for (int r = ya - 1; r <= yb; ++r) {
            const bool in_img = r >= 0 && r < p.H;
            float dog[XL];
            if (row_live(r)) {                                     // (wave-uniform)
                stage_row<R1>(r1, tile1[wv], lane);
                stage_row<R2>(r2, tile2[wv], lane);
                if (r + 1 <= yb && row_live(r + 1)) {              // next row in flight while this one is filtered
                    fetch_row<R1>(r1, y1p + (long)(r + 1) * p.W, p.W, lane);
                    fetch_row<R2>(r2, y2p + (long)(r + 1) * p.W, p.W, lane);
                }
                float g1[XL], g2[XL];
                // one filter at a time: without the fences the scheduler hoists the second window's 32 LDS reads above the
                // first filter's arithmetic and the row loop spills
                __builtin_amdgcn_sched_barrier(0);
                filter_row<R2>(tile2[wv], w2, lane, g2);
                __builtin_amdgcn_sched_barrier(0);
                filter_row<R1>(tile1[wv], w1, lane, g1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < XL; ++i) dog[i] = !lane_in ? NEG : (xzero[i] ? 0.f : g2[i] - g1[i]);
            } else {
                if (r + 1 <= yb && row_live(r + 1)) {
                    fetch_row<R1>(r1, y1p + (long)(r + 1) * p.W, p.W, lane);
                    fetch_row<R2>(r2, y2p + (long)(r + 1) * p.W, p.W, lane);
                }
#pragma unroll
                for (int i = 0; i < XL; ++i) dog[i] = (in_img && lane_in) ? 0.f : NEG;
            }
            // 3x3 window: x neighbours from the adjacent lanes, y neighbours from the ring
            float xm[XL];
            const float left = dx_from_lower(dog[XL - 1], NEG), right = dx_from_upper(dog[0], NEG);
            xm[0] = mx3(left, dog[0], dog[1]);
#pragma unroll
            for (int i = 1; i < XL - 1; ++i) xm[i] = mx3(dog[i - 1], dog[i], dog[i + 1]);
            xm[XL - 1] = mx3(dog[XL - 2], dog[XL - 1], right);
            const int ro = r - 1;                                  // row that is complete now
            if (ro >= ya && ro < yb) {
                float out[XL];
                unsigned flags = 0;
#pragma unroll
                for (int i = 0; i < XL; ++i) {
                    const float hm = mx3(xm_pp[i], xm_p[i], xm[i]);
                    out[i] = (hm == c_p[i]) ? c_p[i] : 0.f;
                    if (lane_in && out[i] > 0.f) {
                        flags |= 1u << i;
                        const double dv = (double)out[i];
                        st_n += 1.0; st_s += dv; st_ss += dv * dv;
                    }
                }
                const long obase = plane + (long)ro * p.W + x0;
                if (p.nms_out && lane_in) {
                    *reinterpret_cast<float4*>(p.nms_out + obase) = make_float4(out[0], out[1], out[2], out[3]);
                    *reinterpret_cast<float4*>(p.nms_out + obase + 4) = make_float4(out[4], out[5], out[6], out[7]);
                }
                if (__ballot(flags != 0)) {                        // (wave-uniform; survivors are sparse)
#pragma unroll
                    for (int i = 0; i < XL; ++i) {
                        const bool is = (flags >> i) & 1u;
                        const unsigned long long mask = __ballot(is);
                        if (mask) {
                            if (is) {
                                const unsigned pos = cnt + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                                ring[pos & (DRING - 1)] = make_uint2(__float_as_uint(out[i]), (unsigned)(obase + i));
                            }
                            cnt += (unsigned)__popcll(mask);
                        }
                    }
                    while (cnt - flushed >= 64u) {
                        const uint2 e = ring[(flushed + lane) & (DRING - 1)];
                        if (flushed + lane < p.seg_cap) seg_base[flushed + lane] = e;
                        flushed += 64;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < XL; ++i) { xm_pp[i] = xm_p[i]; xm_p[i] = xm[i]; c_p[i] = dog[i]; }
        }
    }
    if (cnt > flushed && flushed + lane < cnt && flushed + lane < p.seg_cap)
        seg_base[flushed + lane] = ring[(flushed + lane) & (DRING - 1)];
    const double a = wave_sum(st_n), s = wave_sum(st_s), ss = wave_sum(st_ss);
    if (lane == 0) {
        p.seg_count[g] = min(cnt, p.seg_cap);
        if (cnt > p.seg_cap) atomicOr(p.overflow, 1u);
        p.stats[3 * g + 0] = a; p.stats[3 * g + 1] = s; p.stats[3 * g + 2] = ss;
    }
}

inline int dogx_radius(float sigma) { return (int)(4.0f * sigma + 0.5f); }

template <int R>
void fill_sym(float sigma, SymTaps<R>& t) {
    const int rs = dogx_radius(sigma);
    double tmp[R + 1], sum = 0;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int d = 0; d <= R; ++d) {
        tmp[d] = d <= rs ? exp(c * (double)d * (double)d) : 0.0;
        sum += d == 0 ? tmp[d] : 2.0 * tmp[d];
    }
    for (int d = 0; d <= R; ++d) t.w[d] = (float)(tmp[d] / sum);
}

template <int R1, int R2>
int launch_dogx(const DogxParams& p, float s1, float s2, hipStream_t st) {
    SymTaps<R1> w1;
    SymTaps<R2> w2;
    fill_sym<R1>(s1, w1);
    fill_sym<R2>(s2, w2);
    const long waves = (long)p.D * p.n_ychunks;
    hipLaunchKernelGGL((dogx_nms_kernel<R1, R2>), dim3((unsigned)((waves + DWPB - 1) / DWPB)), dim3(DNT), 0, st, p, w1, w2);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

DogxGrid mi_dogx_grid(int D, int H, int W) {
    (void)W;
    DogxGrid g;
    // about 8 waves per CU, chunks of >= 32 rows (a chunk filters its two halo rows again)
    int chunk = H;
    while (chunk > 32 && (long)D * mi_cdiv(H, chunk) < 2048) chunk = (chunk + 1) / 2;
    g.ychunk = chunk;
    g.n_ychunks = mi_cdiv(H, chunk);
    g.n_seg = (unsigned)((long)D * g.n_ychunks);
    g.seg_cap = (unsigned)(chunk * SEGW / 4 + 64);       // xy-NMS survivors: at most one per 2x2 patch without plateaus
    return g;
}

bool mi_dogx_usable(const float* y1, const float* y2, const float* nms_out, int D, int H, int W, float s1, float s2,
                    int k) {
    if (getenv("MI_NO_DOGX")) return false;
    auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const int r1 = dogx_radius(s1), r2 = dogx_radius(s2);
    return k == 3 && (W & 7) == 0 && W >= 64 && W <= SEGW && H >= 1 && D >= 1 && r1 <= 12 && r2 <= 20 && r1 >= 1 && r2 >= 1 &&
           al(y1) && al(y2) && al(nms_out);
}

int mi_launch_dogx(DogxParams p, const DogxGrid& g, float s1, float s2, hipStream_t st) {
    p.ychunk = g.ychunk; p.n_ychunks = g.n_ychunks; p.seg_cap = g.seg_cap;
    const int r1 = dogx_radius(s1), r2 = dogx_radius(s2);
    if (r1 <= 8 && r2 <= 16) return launch_dogx<8, 16>(p, s1, s2, st);
    if (r1 <= 8) return launch_dogx<8, 20>(p, s1, s2, st);
    if (r2 <= 16) return launch_dogx<12, 16>(p, s1, s2, st);
    return launch_dogx<12, 20>(p, s1, s2, st);
}
