// 3x3x3 pooled NMS of the detector decode as a register march: no LDS staging, no barrier in the loop.
//
// Replaces (reference, cet_pick/...): models/utils.py:167-169 `_sigmoid` + models/decode.py:27-33 `_nms` (window
// (3,3,3)) for the fused decode and for mi_nms3d(kd=3, kh=3).
//
// HBM-bound: 4 B read + 4 B written per voxel (the sigmoid heat-map is an output).  The older LDS-tile march kept 8 KB
// of loads in flight per CU (one 16-B load per thread and plane, two workgroups per CU) and ran at 1.5 TB/s on a
// 128x256x256 volume; its 64-wide tiles also fetched two extra cache lines per row for the side halo.  Here
//   * a WAVE owns a strip of RY = 4 rows x 256 x (64 lanes x 4 consecutive x) and marches over a z-chunk;
//   * a lane loads RY + 2 rows per plane as 16-B loads (96 B per lane and plane), two planes ahead (PD = 2):
//     ~48 KB in flight per CU at one wave per SIMD;
//   * x neighbours come from the adjacent lanes by DPP wave shifts (the strip spans the whole row when W <= 256; wider
//     rows add one predicated 4-B load per row for the two strip-edge lanes), y neighbours are the lane's own rows
//     (rows y0-1 and y0+4 are fetched again by the neighbouring strip: L1/L2 hits), the z window lives in registers;
//   * local maxima are compacted by ballot into a small LDS ring per wave and leave it 64 at a time as one coalesced
//     512-B store into a candidate segment private to the wave (no global counter; storing them straight from the
//     ballot rounds - 16 sparse store instructions per plane - cost more than the whole rest of the kernel); their score
//     histogram is kept per workgroup in LDS (one LDS atomic per flushed entry) and merged once at the end.
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int RY = 4, WX = 256, WPB = 8, NT3 = 64 * WPB;      // (8 waves per workgroup: half the histogram merges of 4; 16: 3x slower)
constexpr int NROW = RY + 2;
#ifndef MI_PEAK3_NB
#define MI_PEAK3_NB 3
#endif
constexpr int NB = MI_PEAK3_NB;   // planes in flight per wave
constexpr int RING = 512;          // candidate ring per wave: < 64 pending + one row of the strip (<= 256 new entries)

__device__ __forceinline__ float dpp_from_lower(float v, float edge) {     // lane i <- lane i-1; lane 0 <- edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper(float v, float edge) {     // lane i <- lane i+1; lane 63 <- edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <bool SIGMOID>
__device__ __forceinline__ float xform(float v) {
    if (SIGMOID) {
        // v_exp_f32 + v_rcp_f32 (about 1e-6 relative on the clamped range, monotone), as in nms_march_kernel
        float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        return sg != sg ? sg : fminf(fmaxf(sg, 1e-4f), 1.0f - 1e-4f);       // torch.clamp keeps a NaN
    }
    return v;
}

struct RawPlane {
    float4 row[NROW];
    float halo[NROW];        // XHALO only: lane 0 holds x = xs-1, lane 63 holds x = xs+256
};

template <bool SIGMOID, bool XHALO>
__global__ __launch_bounds__(NT3) void peak3_march_kernel(Peak3Params p) {
    __shared__ unsigned lhist[MI_HIST_BINS];
    __shared__ uint2 ring_all[WPB][RING];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float NEG = -INFINITY;
    const int xs = blockIdx.x * WX;
    const int x0 = xs + 4 * lane;
    const int y0 = (blockIdx.y * WPB + wv) * RY;
    const int z0 = blockIdx.z * p.zchunk;
    const int zend = min(z0 + p.zchunk, p.D);
    const long HW = (long)p.H * p.W;
    const bool lane_ok = x0 < p.W;                         // W % 4 == 0: the whole float4 is inside
    const bool strip_ok = y0 < p.H;                        // (wave-uniform)
    const bool emit = p.cands != nullptr;
    if (p.hist) for (int i = tid; i < MI_HIST_BINS; i += NT3) lhist[i] = 0;
    __syncthreads();

    // segment of this wave
    const unsigned seg = (unsigned)(((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WPB + wv;
    uint2* seg_base = emit ? p.cands + (size_t)seg * p.seg_cap : nullptr;
    unsigned cnt = 0, flushed = 0;
    uint2* ring = ring_all[wv];
    auto flush64 = [&]() {               // entries [flushed, flushed + 64) of the ring -> the segment, coalesced
        const uint2 e = ring[(flushed + lane) & (RING - 1)];
        seg_base[flushed + lane] = e;
        if (p.hist) atomicAdd(&lhist[e.x >> MI_HIST_SHIFT], 1u);
        flushed += 64;
    };

    // rows of the strip with halo: r = 0 .. RY+1  <->  y = y0 - 1 + r
    bool row_in[NROW];
#pragma unroll
    for (int r = 0; r < NROW; ++r) { const int gy = y0 - 1 + r; row_in[r] = gy >= 0 && gy < p.H; }
    const bool has_left = XHALO && xs > 0, has_right = XHALO && xs + WX < p.W;
    const bool halo_lane = XHALO && ((lane == 0 && has_left) || (lane == 63 && has_right));
    const int halo_x = lane == 0 ? xs - 1 : xs + WX;

    auto fetch = [&](RawPlane& raw, int zz) {
        if (zz < 0 || zz >= p.D || zz > zend || !strip_ok) return;          // (wave-uniform)
        const float* pl = p.in + (long)zz * HW;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            if (!row_in[r]) continue;
            const float* rowp = pl + (long)(y0 - 1 + r) * p.W;
            if (lane_ok) raw.row[r] = *reinterpret_cast<const float4*>(rowp + x0);
            if (XHALO && halo_lane) raw.halo[r] = rowp[halo_x];
        }
    };

    float ringM[2][RY][4];      // xy-pooled planes z-2, z-1 (relative to the plane being processed)
    float cprev[RY][4];         // centre values of plane z-1
#pragma unroll
    for (int q = 0; q < RY; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ringM[0][q][i] = NEG; ringM[1][q][i] = NEG; cprev[q][i] = 0.f; }

    auto process = [&](RawPlane& raw, int zz) {
        const bool plane_in = zz >= 0 && zz < p.D;
        float m[RY][4], ccur[RY][4];
        if (plane_in) {
            float c[NROW][4], xm[NROW][4];
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const bool ok = row_in[r] && lane_ok;
                c[r][0] = ok ? xform<SIGMOID>(raw.row[r].x) : NEG;
                c[r][1] = ok ? xform<SIGMOID>(raw.row[r].y) : NEG;
                c[r][2] = ok ? xform<SIGMOID>(raw.row[r].z) : NEG;
                c[r][3] = ok ? xform<SIGMOID>(raw.row[r].w) : NEG;
                float edge = NEG;
                if (XHALO) edge = (halo_lane && row_in[r]) ? xform<SIGMOID>(raw.halo[r]) : NEG;
                const float left = dpp_from_lower(c[r][3], edge);
                const float right = dpp_from_upper(c[r][0], edge);
                xm[r][0] = max3f(left, c[r][0], c[r][1]);
                xm[r][1] = max3f(c[r][0], c[r][1], c[r][2]);
                xm[r][2] = max3f(c[r][1], c[r][2], c[r][3]);
                xm[r][3] = max3f(c[r][2], c[r][3], right);
            }
#pragma unroll
            for (int q = 0; q < RY; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m[q][i] = max3f(xm[q][i], xm[q + 1][i], xm[q + 2][i]);
                    ccur[q][i] = c[q + 1][i];
                }
            // the pre-NMS value (sigmoid heat-map) is an output of the fused decode
            if (p.val_out && zz >= z0 && zz < zend && lane_ok) {
#pragma unroll
                for (int q = 0; q < RY; ++q)
                    if (y0 + q < p.H)
                        *reinterpret_cast<float4*>(p.val_out + (long)zz * HW + (long)(y0 + q) * p.W + x0) =
                            make_float4(ccur[q][0], ccur[q][1], ccur[q][2], ccur[q][3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < RY; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { m[q][i] = NEG; ccur[q][i] = 0.f; }
        }
        // ---- emit plane zo = zz - 1: window = ringM[0] (zo-1), ringM[1] (zo), m (zo+1)
        const int zo = zz - 1;
        if (zo >= z0 && zo < zend) {
#pragma unroll
            for (int q = 0; q < RY; ++q) {
                const bool rok = (y0 + q < p.H) && lane_ok;
                const long obase = (long)zo * HW + (long)(y0 + q) * p.W + x0;
                float out[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hm = max3f(ringM[0][q][i], ringM[1][q][i], m[q][i]);
                    const float cc = cprev[q][i];
                    out[i] = (hm == cc) ? cc : 0.f;
                }
                if (p.nms_out && rok)
                    *reinterpret_cast<float4*>(p.nms_out + obase) = make_float4(out[0], out[1], out[2], out[3]);
                if (emit) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool is = rok && out[i] > 0.f;
                        const unsigned long long mask = __ballot(is);
                        if (mask) {                                             // (wave-uniform)
                            if (is) {
                                const unsigned pos = cnt + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                                ring[pos & (RING - 1)] = make_uint2(__float_as_uint(out[i]), (unsigned)(obase + i));
                            }
                            cnt += (unsigned)__popcll(mask);
                        }
                    }
                    while (cnt - flushed >= 64u) flush64();                     // (wave-uniform)
                }
            }
        }
        // ---- shift the z window
#pragma unroll
        for (int q = 0; q < RY; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ringM[0][q][i] = ringM[1][q][i];
                ringM[1][q][i] = m[q][i];
                cprev[q][i] = ccur[q][i];
            }
    };

    if (strip_ok) {
        // NB planes in flight (a ring of statically indexed register buffers): a wave lives for zchunk + 2 planes only, so
        // with two planes ahead the pipeline's fill - one memory latency with nothing to do - was most of its life
        // (SQ counters: 59 % of the wave cycles waiting, the vector unit 20 - 35 % busy)
        RawPlane raw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) fetch(raw[b], z0 - 1 + b);
        for (int zz = z0 - 1; zz <= zend; zz += NB) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (zz + b <= zend) {                       // (wave-uniform)
                    process(raw[b], zz + b);
                    fetch(raw[b], zz + b + NB);
                }
        }
    }
    if (emit) {
        if (cnt > flushed) {                                                    // the tail: fewer than 64 entries
            const bool has = flushed + lane < cnt;
            if (has) {
                const uint2 e = ring[(flushed + lane) & (RING - 1)];
                seg_base[flushed + lane] = e;
                if (p.hist) atomicAdd(&lhist[e.x >> MI_HIST_SHIFT], 1u);
            }
        }
        if (lane == 0) p.seg_count[seg] = cnt;
    }
    if (p.hist) {
        __syncthreads();
        for (int i = tid; i < MI_HIST_BINS; i += NT3) {
            const unsigned hcount = lhist[i];
            if (hcount) atomicAdd(&p.hist[i], hcount);
        }
    }
}

}  // namespace

// grid of the register march; every wave owns one candidate segment of seg_cap = zchunk * RY * WX entries
Peak3Grid mi_peak3_grid(int D, int H, int W) {
    Peak3Grid g;
    g.gx = mi_cdiv(W, WX);
    g.gy = mi_cdiv(mi_cdiv(H, RY), WPB);
    // z-chunks: about 8 waves per CU (2 per SIMD) when the volume has them, chunks of >= 4 planes.  A chunk re-reads its
    // two halo planes (from L2 / the Infinity Cache), yet on 128x256x256 chunks of 4 planes (2048 waves) take 16 us
    // against 23 us for chunks of 8 (1024 waves, one per SIMD) and 18 us for chunks of 2: the march is bound by the bytes
    // it keeps in flight, not by the bytes it moves
    const long strips = (long)g.gx * g.gy * WPB;
    int zc = D;
    int target = 2048;
    if (const char* e = getenv("MI_PEAK3_WAVES")) target = atoi(e);
    int minz = 4;
    if (const char* e = getenv("MI_PEAK3_MINZ")) minz = std::max(1, atoi(e));
    while (zc > minz && strips * mi_cdiv(D, zc) < target) zc = (zc + 1) / 2;
    g.zchunk = zc;
    g.gz = mi_cdiv(D, zc);
    g.n_seg = (unsigned)((long)g.gx * g.gy * g.gz * WPB);
    g.seg_cap = (unsigned)zc * RY * WX;
    return g;
}

bool mi_peak3_usable(const float* in, const float* val_out, const float* nms_out, int D, int H, int W) {
    if (getenv("MI_NO_PEAK3")) return false;
    auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    return (W & 3) == 0 && al(in) && al(val_out) && al(nms_out) && D > 0 && H > 0 && W > 0 &&
           (size_t)D * H * W < (1ull << 32);
}

int mi_launch_peak3(Peak3Params p, const Peak3Grid& g, bool sigmoid, hipStream_t s) {
    p.zchunk = g.zchunk;
    p.seg_cap = g.seg_cap;
    const dim3 grid(g.gx, g.gy, g.gz);
    const bool xhalo = p.W > WX;
    if (sigmoid) {
        if (xhalo) hipLaunchKernelGGL((peak3_march_kernel<true, true>), grid, dim3(NT3), 0, s, p);
        else hipLaunchKernelGGL((peak3_march_kernel<true, false>), grid, dim3(NT3), 0, s, p);
    } else {
        if (xhalo) hipLaunchKernelGGL((peak3_march_kernel<false, true>), grid, dim3(NT3), 0, s, p);
        else hipLaunchKernelGGL((peak3_march_kernel<false, false>), grid, dim3(NT3), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
