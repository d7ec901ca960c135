// Detector decode in ONE launch (round 5): sigmoid + (3,3,3) pooled NMS + top-K + the (K,5) rows.
//
// Replaces (reference, cet_pick/...): models/utils.py:167-169 `_sigmoid`, models/decode.py:27-33 `_nms` (window
// (3,3,3)), models/decode.py:82-92 `_topk`, models/decode.py:35-41 `_convert_1d_to_3d`, models/decode.py:123-155
// `tomo_decode` - the chain peak3_march_kernel | topk_filter_seg_kernel | topk_final_kernel of rounds 2-4 (kept in
// infer_peak3.hip / infer_nms.hip for the shapes this kernel does not take and as the A/B, MI_DECODE_CHAIN=1).
//
// What the three-launch chain spent (rocprofv3, 128x256x256, K = 900): march 29.2 us, of which 6.8 us is the candidate
// emission (a ballot, a popcount and an LDS ring store per output register, 16 per plane and wave, + 1.7 us of
// histogram atomics) - for 310 k local maxima of which 900 matter -, then 8 us to filter the 2.5 MB of candidates
// back in and 10 us to rank ~1100 survivors on 16 workgroups.  Here:
//   * the march is the register march of infer_peak3.hip (a wave = 4 rows x 256 x, z-chunk in registers, x neighbours
//     by DPP); a local maximum goes into a list PRIVATE TO ITS LANE in LDS (one predicated ds_write_b64 and an add: no
//     ballot, no popcount, no ring), capacity 8 per lane and wave life - a lane sees 64 output voxels, and without
//     plateaus at most 8 of them can be maxima of a 3x3x3 window; a lane that fills up makes its wave spill;
//   * at the end of a workgroup its ~1200 candidates are still in LDS: a 2048-bin histogram of the score bits (+ a
//     256-bin refinement of the boundary bin) gives the workgroup's M best (M >= 32) exactly, which it appends to a
//     packed table (ONE returning atomic per workgroup) together with the bound below which it dropped; every
//     candidate also goes to the wave's segment, as before (the exact fall-back reads those);
//   * the workgroup that draws the last ticket selects: histogram of the table (~8 k entries) -> the K-th score's 20-bit
//     prefix -> ~K survivors in LDS -> rank = start of the survivor's bucket (2048 monotone buckets over the survivors'
//     score range) + the larger keys of its own bucket -> row `rank` of the output.  It is exact iff no workgroup
//     dropped a candidate at or above the survivors' threshold - checked against the workgroups' bounds; otherwise
//     (dense plateaus, more than M of the K best in one workgroup, fewer than K candidates in the tables) the same
//     workgroup runs the exact 64-bit radix select over ALL candidate segments: slower, never wrong, bounded.
// Hand-over without a device-scope fence (MI355X_MICROARCH.md, inter-workgroup visibility, the sc1 form): table, bounds
// and segments are stored past the caches (agent-scope relaxed atomic stores = sc1), every storing wave drains its
// stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane draws the ticket (agent-scope atomic add); the
// workgroup whose add came last loads everything the same way, after a barrier behind the add.  Nobody waits for
// anybody: a workgroup that is not last returns.  The last one leaves the header zeroed for the next call.
#include "common.h"
#include "infer_common.h"

namespace {

constexpr int RY = 4, WX = 256, WPB = 8, NT = 64 * WPB, NROW = RY + 2, NB = 3;
constexpr int LCAP = 8;                  // candidates a lane can hold
constexpr int TSLOT = 32;                // table slots of a workgroup (unused ones are written as zeros)
constexpr int SCAP = 2048;               // survivors the selecting workgroup ranks in LDS
constexpr int NBUCKET = 2048;
constexpr unsigned FLAG_INCOMPLETE = 1u; // a wave spilled / the workgroup's keep list overflowed: its bound is not valid

__device__ __forceinline__ float d_from_lower(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float d_from_upper(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float d_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// Wave-wide scans and reductions on the DPP path (row shifts + row broadcasts: a few cycles each; the __shfl forms are
// ds_bpermute round trips through the LDS crossbar, ~100 cycles apiece in a dependent chain)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp0(unsigned v) {      // lanes without a source (or masked out) read 0
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
    v += dpp0<0x111, 0xf>(v);        // row_shr:1
    v += dpp0<0x112, 0xf>(v);        // row_shr:2
    v += dpp0<0x114, 0xf>(v);        // row_shr:4
    v += dpp0<0x118, 0xf>(v);        // row_shr:8   -> inclusive within each row of 16
    v += dpp0<0x142, 0xa>(v);        // row_bcast:15 into rows 1, 3
    v += dpp0<0x143, 0xc>(v);        // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned wave_last(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) { return wave_last(wave_incl_scan_u32(v)); }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp0<0x111, 0xf>(v));
    v = max(v, dpp0<0x112, 0xf>(v));
    v = max(v, dpp0<0x114, 0xf>(v));
    v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v));
    v = max(v, dpp0<0x143, 0xc>(v));
    return wave_last(v);
}
template <bool SIGMOID>
__device__ __forceinline__ float d_xform(float v) {
    if (SIGMOID) {
        // v_exp_f32 + v_rcp_f32 (about 1e-6 relative on the clamped range, monotone), as in peak3_march_kernel
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        return sg != sg ? sg : __builtin_amdgcn_fmed3f(sg, 1e-4f, 1.0f - 1e-4f);   // torch.clamp keeps a NaN
    }
    return v;
}

// stores / loads that other workgroups read or that read other workgroups' data: past the caches (sc1)
__device__ __forceinline__ void st_u64(uint2* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_u64(const uint2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct RawPlane {
    float4 row[NROW];
    float halo[NROW];        // XHALO only: lane 0 holds x = xs-1, lane 63 holds x = xs+256
};

// Highest 12-bit bin T (score bits >> 20) with count(bins >= T) >= want, by ONE wave, from a 2048-bin histogram in LDS.
// *above_out = count(bins > T).  Fewer than `want` entries in all: T = 0 (callers treat T = 0 as "keep everything").
// Lane l reads four consecutive bins of each 256-bin row j (bins 256 j + 4 l ..+3: 16-byte lane stride, conflict-free -
// a lane owning 32 consecutive bins reads at a 128-byte stride, every lane of a read group on the same banks, and all
// eight waves of the workgroup do this at once); sums across lanes by DPP; branch-free.
__device__ __forceinline__ int lds_threshold_bin(const unsigned* hist, unsigned want, int lane, unsigned* above_out) {
    constexpr int ROWS = MI_HIST_BINS / 256;
    const uint4* src = reinterpret_cast<const uint4*>(hist) + lane;
    uint4 r[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) r[j] = src[64 * j];
    unsigned pre[ROWS], tot[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        pre[j] = wave_incl_scan_u32(r[j].x + r[j].y + r[j].z + r[j].w);
        tot[j] = wave_last(pre[j]);
    }
    unsigned above_row = 0;                                 // entries in the rows above row j (scalar)
    int T = -1;
    unsigned ab_sel = 0;
#pragma unroll
    for (int j = ROWS - 1; j >= 0; --j) {
        const unsigned mine = r[j].x + r[j].y + r[j].z + r[j].w;
        const unsigned above = above_row + tot[j] - pre[j]; // entries in bins above this lane's four
        const unsigned h[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
        unsigned acc = above, nb = 0, ab = above;
#pragma unroll
        for (int e = 3; e >= 0; --e) {
            acc += h[e];
            const bool ge = acc >= want;
            nb += ge ? 1u : 0u;
            ab = ge ? ab : acc;                             // the last suffix sum below `want`
        }
        const bool holds = above < want && above + mine >= want;    // one (row, lane) in all, or none
        T = holds ? (256 * j + 4 * lane + (int)nb - 1) : T;
        ab_sel = holds ? ab : ab_sel;
        above_row += tot[j];
    }
    const unsigned long long m = __ballot(T >= 0);
    if (!m) { *above_out = 0u; return 0; }
    const int src_lane = __ffsll((long long)m) - 1;
    *above_out = (unsigned)__builtin_amdgcn_readlane((int)ab_sel, src_lane);
    return __builtin_amdgcn_readlane(T, src_lane);
}

// the same over a 256-bin histogram: highest sub-bin S with base + count(sub-bins >= S) >= want (0 if none)
__device__ __forceinline__ int lds_threshold_sub(const unsigned* sub, unsigned base, unsigned want, int lane) {
    const uint4 v = reinterpret_cast<const uint4*>(sub)[lane];
    const unsigned h[4] = {v.x, v.y, v.z, v.w};
    const unsigned mine = h[0] + h[1] + h[2] + h[3];
    const unsigned pre = wave_incl_scan_u32(mine);
    const unsigned above = base + wave_last(pre) - pre;
    unsigned acc = above, nb = 0;
#pragma unroll
    for (int b = 3; b >= 0; --b) {
        acc += h[b];
        nb += acc >= want ? 1u : 0u;
    }
    const bool holds = above < want && above + mine >= want;
    const unsigned long long m = __ballot(holds);
    if (!m) return 0;
    const int src_lane = __ffsll((long long)m) - 1;
    return src_lane * 4 + (int)__builtin_amdgcn_readlane((int)nb, src_lane) - 1;
}

__device__ __forceinline__ void d_emit_det(float* dets, int r, unsigned long long key, int H, int W, bool valid) {
    float* o = dets + 5 * (long)r;
    if (!valid) { o[0] = 0.25f; o[1] = 0.25f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; return; }
    const float score = __uint_as_float((unsigned)(key >> 32));
    const unsigned idx = ~(unsigned)(key & 0xffffffffu);
    // `_convert_1d_to_3d` (decode.py:35-41): float32 division, then integer remainder
    const int hw = H * W;
    const int z = (int)floorf(__fdiv_rn((float)idx, (float)hw));
    const int t = (int)idx - z * hw;
    const float yf = floorf(__fdiv_rn((float)t, (float)W));
    int x = t % W;
    if (x < 0) x += W;
    o[0] = (float)x + 0.25f; o[1] = yf + 0.25f; o[2] = (float)z; o[3] = score; o[4] = score;
}

struct Decode1Params {
    const float* in;
    float* val_out;          // sigmoid heat-map or null
    int D, H, W, zchunk;
    uint2* cands;            // segment of wave s = cands + s * seg_cap: every candidate of the wave
    unsigned* seg_count;
    unsigned seg_cap, n_seg, n_wg;
    DecodeHeader* hdr;       // pad0: ticket, cand_count: candidates in all (both zero between calls)
    uint2* table;            // n_wg * TSLOT slots; workgroup w owns [w * TSLOT, (w + 1) * TSLOT)
    uint2* meta;             // per workgroup {bound (20-bit score prefix below which it dropped), flags}
    int K, M;
    float* dets;
    int* n_valid_out;
    int self_clean;
};

// LDS: the march's per-lane lists and the selection's arrays share one arena
struct alignas(16) D1Lds {
    union {
        uint2 lists[WPB][LCAP + 1][64];                // 36 KB: slot j of lane l of wave w (slot LCAP: overflow dummy)
        struct { unsigned long long S[SCAP]; unsigned long long P[SCAP]; } fin;     // 32 KB
    } a;
    unsigned hist[MI_HIST_BINS];                        // 8 KB (the final's bucket counts alias it: NBUCKET == MI_HIST_BINS)
    unsigned start[NBUCKET];                            // 8 KB
    unsigned sub[256];
    uint2 stage[TSLOT];
    unsigned keep_n, spilled, base, last, wg_total, hi_bits, n_s, prefix_lo, fallback;
    unsigned long long r_prefix;
    unsigned r_remaining;
    unsigned wtot[WPB];
};
static_assert(NBUCKET == MI_HIST_BINS, "bucket counts alias the histogram");

template <bool SIGMOID, bool XHALO>
__global__ __launch_bounds__(NT) void decode1_kernel(Decode1Params p) {
    __shared__ D1Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float NEG = -INFINITY;
    const int xs = blockIdx.x * WX;
    const int x0 = xs + 4 * lane;
    const int y0 = (blockIdx.y * WPB + wv) * RY;
    const int z0 = blockIdx.z * p.zchunk;
    const int zend = min(z0 + p.zchunk, p.D);
    const long HW = (long)p.H * p.W;
    const bool lane_ok = x0 < p.W;
    const bool strip_ok = y0 < p.H;
    const unsigned wg = (unsigned)(((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    const unsigned seg = wg * WPB + wv;
    uint2* seg_base = p.cands + (size_t)seg * p.seg_cap;
    if (tid == 0) { L.keep_n = 0; L.spilled = 0; L.wg_total = 0; }
    for (int i = tid; i < MI_HIST_BINS; i += NT) L.hist[i] = 0u;       // (used at the end of the workgroup only)
    if (tid < 256) L.sub[tid] = 0u;
    if (tid < TSLOT) L.stage[tid] = make_uint2(0u, 0u);
    __syncthreads();

    uint2* mylist = &L.a.lists[wv][0][lane];            // slot j at mylist + 64 j
    unsigned n_l = 0;                                   // entries of this lane's list
    unsigned seg_n = 0;                                 // entries of the wave already in its segment (wave-uniform)
    bool spilled = false;

    // all lanes' lists -> the segment, compacted (lane order, slot order); lists emptied
    auto spill = [&]() {
        const unsigned incl = wave_incl_scan_u32(n_l);
        const unsigned excl = incl - n_l, total = wave_last(incl);
        for (unsigned j = 0; j < n_l; ++j) {
            const unsigned pos = seg_n + excl + j;
            if (pos < p.seg_cap) st_u64(seg_base + pos, mylist[64 * j]);
        }
        seg_n += total;
        n_l = 0;
    };

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
                c[r][0] = ok ? d_xform<SIGMOID>(raw.row[r].x) : NEG;
                c[r][1] = ok ? d_xform<SIGMOID>(raw.row[r].y) : NEG;
                c[r][2] = ok ? d_xform<SIGMOID>(raw.row[r].z) : NEG;
                c[r][3] = ok ? d_xform<SIGMOID>(raw.row[r].w) : NEG;
                float edge = NEG;
                if (XHALO) edge = (halo_lane && row_in[r]) ? d_xform<SIGMOID>(raw.halo[r]) : NEG;
                const float left = d_from_lower(c[r][3], edge);
                const float right = d_from_upper(c[r][0], edge);
                xm[r][0] = d_max3(left, c[r][0], c[r][1]);
                xm[r][1] = d_max3(c[r][0], c[r][1], c[r][2]);
                xm[r][2] = d_max3(c[r][1], c[r][2], c[r][3]);
                xm[r][3] = d_max3(c[r][2], c[r][3], right);
            }
#pragma unroll
            for (int q = 0; q < RY; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m[q][i] = d_max3(xm[q][i], xm[q + 1][i], xm[q + 2][i]);
                    ccur[q][i] = c[q + 1][i];
                }
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
        // ---- plane zo = zz - 1 is complete: window = ringM[0] (zo-1), ringM[1] (zo), m (zo+1)
        const int zo = zz - 1;
        if (zo >= z0 && zo < zend) {
            const unsigned n_l0 = n_l;
#pragma unroll
            for (int q = 0; q < RY; ++q) {
                const bool rok = (y0 + q < p.H) && lane_ok;
                const unsigned obase = (unsigned)((long)zo * HW + (long)(y0 + q) * p.W + x0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hm = d_max3(ringM[0][q][i], ringM[1][q][i], m[q][i]);
                    const float cc = cprev[q][i];
                    if (rok && hm == cc && cc > 0.f) {
                        mylist[64 * min(n_l, (unsigned)LCAP)] = make_uint2(__float_as_uint(cc), obase + i);   // slot LCAP: overflow dummy
                        ++n_l;
                    }
                }
            }
            if (__ballot(n_l > (unsigned)LCAP) != 0ull) {
                // a lane ran out of slots inside this plane (plateaus, dense maxima): the lists as they were before the
                // plane go to the segment, and so does the plane itself, straight from the registers
                n_l = n_l0;
                spill();
                unsigned add = 0;
#pragma unroll
                for (int q = 0; q < RY; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float hm = d_max3(ringM[0][q][i], ringM[1][q][i], m[q][i]);
                        add += ((y0 + q < p.H) && lane_ok && hm == cprev[q][i] && cprev[q][i] > 0.f) ? 1u : 0u;
                    }
                const unsigned incl = wave_incl_scan_u32(add);
                unsigned pos = seg_n + incl - add;
                seg_n += wave_last(incl);
#pragma unroll
                for (int q = 0; q < RY; ++q) {
                    const unsigned obase = (unsigned)((long)zo * HW + (long)(y0 + q) * p.W + x0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float hm = d_max3(ringM[0][q][i], ringM[1][q][i], m[q][i]);
                        const float cc = cprev[q][i];
                        if ((y0 + q < p.H) && lane_ok && hm == cc && cc > 0.f) {
                            if (pos < p.seg_cap) st_u64(seg_base + pos, make_uint2(__float_as_uint(cc), obase + i));
                            ++pos;
                        }
                    }
                }
                spilled = true;
            }
        }
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

    // ------------------------------------------------------------------------------------------------------------
    // end of the workgroup: its best M candidates -> its slots of the table; every candidate -> its wave's segment
    // ------------------------------------------------------------------------------------------------------------
    // (hist / sub / stage were zeroed at kernel entry: the march does not touch them)
    // arguments of the selection code
    unsigned a_M = (unsigned)p.M, a_K = (unsigned)p.K, a_nwg = p.n_wg, a_segcap = p.seg_cap, a_nseg = p.n_seg;
    int a_H = p.H, a_W = p.W;
    uint2* a_meta = p.meta;
    uint2* a_table = p.table;
    uint2* a_cands = p.cands;
    unsigned* a_segcount = p.seg_count;
    DecodeHeader* a_hdr = p.hdr;
    float* a_dets = p.dets;
    a_K = (unsigned)__builtin_amdgcn_readfirstlane((int)a_K);
    a_M = (unsigned)__builtin_amdgcn_readfirstlane((int)a_M);
    uint2 le[LCAP];                                         // this lane's list, in registers for the three passes below
#pragma unroll
    for (int j = 0; j < LCAP; ++j) le[j] = (unsigned)j < n_l ? mylist[64 * j] : make_uint2(0u, 0u);
    {
        const unsigned incl = wave_incl_scan_u32(n_l);
        const unsigned excl = seg_n + incl - n_l, total = seg_n + wave_last(incl);
#pragma unroll
        for (int j = 0; j < LCAP; ++j)
            if ((unsigned)j < n_l) {
                atomicAdd(&L.hist[le[j].x >> MI_HIST_SHIFT], 1u);
                if (excl + j < a_segcap) st_u64(seg_base + excl + j, le[j]);
            }
        if (lane == 0) {
            st_u32(a_segcount + seg, min(total, a_segcap));
            if (total) atomicAdd(&L.wg_total, total);
            if (spilled || total > a_segcap) L.spilled = 1u;
        }
    }
    __syncthreads();
    unsigned above = 0;
    const int T = lds_threshold_bin(L.hist, a_M, lane, &above);      // (every wave for itself: no barrier)
    if (T > 0) {
#pragma unroll
        for (int j = 0; j < LCAP; ++j)
            if ((unsigned)j < n_l && (int)(le[j].x >> MI_HIST_SHIFT) == T) atomicAdd(&L.sub[(le[j].x >> 12) & 255u], 1u);
    }
    __syncthreads();
    const unsigned bound = T > 0 ? (((unsigned)T << 8) | (unsigned)lds_threshold_sub(L.sub, above, a_M, lane)) : 0u;
    {
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < LCAP; ++j) c += ((unsigned)j < n_l && (le[j].x >> 12) >= bound) ? 1u : 0u;
        const unsigned incl = wave_incl_scan_u32(c);
        const unsigned wtotal = wave_last(incl);
        unsigned base = 0;
        if (lane == 63 && wtotal) base = atomicAdd(&L.keep_n, wtotal);      // one returning LDS atomic per wave
        base = wave_last(base);
        unsigned slot = base + incl - c;
#pragma unroll
        for (int j = 0; j < LCAP; ++j)
            if ((unsigned)j < n_l && (le[j].x >> 12) >= bound) {
                if (slot < (unsigned)TSLOT) L.stage[slot] = le[j];
                ++slot;
            }
    }
    __syncthreads();
    if (wv == 0) {
        // all TSLOT slots are written, the unused ones as zeros: the selecting workgroup loads them blindly
        if (lane < TSLOT) st_u64(a_table + (size_t)wg * TSLOT + lane, L.stage[lane]);
        if (lane == 0) {
            if (L.wg_total) atomicAdd(&a_hdr->cand_count, L.wg_total);
            const unsigned flags = (L.spilled || L.keep_n > (unsigned)TSLOT) ? FLAG_INCOMPLETE : 0u;
            st_u64(a_meta + wg, make_uint2(bound, flags));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's table / segment / bound stores have left
    __syncthreads();
    if (tid == 0) L.last = (__hip_atomic_fetch_add(&a_hdr->pad0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a_nwg - 1) ? 1u : 0u;
    __syncthreads();
    if (!L.last) return;

    // ------------------------------------------------------------------------------------------------------------
    // the last workgroup: select and emit
    // ------------------------------------------------------------------------------------------------------------
    // Table chunk c = slots [c * EPT * NT, (c + 1) * EPT * NT): EPT per thread, all loads issued before the first use (a
    // loop of load -> use is one memory round trip per entry).  One chunk (<= 256 workgroups) stays in registers for the
    // second pass; larger grids load their chunks again.
    constexpr int EPT = 16;                                 // entries per thread and chunk: eight 16-byte loads of two slots
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned n_slots = a_nwg * (unsigned)TSLOT;       // (even)
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a_table, 0, (int)(n_slots * 8u), 0x00020000);
    const unsigned n_chunks = (n_slots + EPT * NT - 1) / (EPT * NT);
    uint2 ent[EPT];
    auto load_chunk = [&](unsigned c) {
        // raw buffer loads past the caches (aux 0x10 = sc1): a pair past the end of the table reads zeros, no branch
#pragma unroll
        for (int u = 0; u < EPT / 2; ++u) {
            const unsigned pair = (c * (EPT / 2) + (unsigned)u) * NT + tid;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(pair * 16u), 0, 0x10);
            ent[2 * u] = make_uint2(v.x, v.y);
            ent[2 * u + 1] = make_uint2(v.z, v.w);
        }
    };
    load_chunk(0);
    unsigned mx = 0, bad = 0;
    for (unsigned w = tid; w < a_nwg; w += NT) {           // the workgroups' bounds: the largest one, any incomplete
        const uint2 mt = ld_u64(a_meta + w);
        mx = max(mx, mt.x);
        bad |= mt.y;
    }
    const unsigned n_total = ld_u32(&a_hdr->cand_count);
    const int K = (int)a_K;
    const int n_valid = (int)min(n_total, (unsigned)K);
    for (int i = tid; i < MI_HIST_BINS; i += NT) L.hist[i] = 0u;
    if (tid < 256) L.sub[tid] = 0u;
    if (tid == 0) { L.fallback = 0u; L.n_s = 0u; L.hi_bits = 0u; L.prefix_lo = 0u; L.keep_n = 0u; }
    __syncthreads();
    if (bad & FLAG_INCOMPLETE) atomicOr(&L.fallback, 1u);
    mx = wave_max_u32(mx);
    if (lane == 0 && mx) atomicMax(&L.prefix_lo, mx);       // (prefix_lo holds the largest bound)
    unsigned n_t_mine = 0;
    for (unsigned c = 0; c < n_chunks; ++c) {
        if (c) load_chunk(c);
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (ent[u].x) { atomicAdd(&L.hist[ent[u].x >> MI_HIST_SHIFT], 1u); ++n_t_mine; }       // (scores are > 0)
    }
    n_t_mine = wave_sum_u32(n_t_mine);
    if (lane == 0 && n_t_mine) atomicAdd(&L.keep_n, n_t_mine);
    __syncthreads();
    const unsigned n_t = L.keep_n;                          // entries in the tables
    const unsigned max_bound = L.prefix_lo;
    unsigned sel_prefix = 0;                                // survivors: score bits >> 12 >= sel_prefix
    bool fallback = L.fallback != 0u;
    if (n_t < (unsigned)K) {
        // fewer than K entries in the tables: exact only if the tables hold every candidate
        if (n_t != n_total) fallback = true;
    } else if (!fallback) {
        unsigned ab = 0;
        const int Tg = lds_threshold_bin(L.hist, (unsigned)K, lane, &ab);
        sel_prefix = (unsigned)Tg << 8;
        if (Tg > 0 && ab + L.hist[Tg] > (unsigned)SCAP) {   // (uniform) the boundary bin is too full for the ranking arrays: refine
            for (unsigned c = 0; c < n_chunks; ++c) {
                if (n_chunks > 1) load_chunk(c);
#pragma unroll
                for (int u = 0; u < EPT; ++u)
                    if (ent[u].x && (int)(ent[u].x >> MI_HIST_SHIFT) == Tg) atomicAdd(&L.sub[(ent[u].x >> 12) & 255u], 1u);
            }
            __syncthreads();
            sel_prefix |= (unsigned)lds_threshold_sub(L.sub, ab, (unsigned)K, lane);
        }
        // every dropped candidate has a prefix below its workgroup's bound: nothing at or above sel_prefix was dropped
        if (max_bound > sel_prefix) fallback = true;
    }
    __syncthreads();
    if (!fallback) {
        for (unsigned c = 0; c < n_chunks; ++c) {
            if (n_chunks > 1) load_chunk(c);
            // this thread's survivors of the chunk, then one prefix over the workgroup: no atomic per entry
            unsigned cnt_mine = 0, hi = 0;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const bool keep = ent[u].x != 0u && (ent[u].x >> 12) >= sel_prefix;
                cnt_mine += keep ? 1u : 0u;
                hi = max(hi, keep ? ent[u].x : 0u);
            }
            const unsigned incl = wave_incl_scan_u32(cnt_mine);
            const unsigned wtotal = wave_last(incl);
            unsigned base = 0;
            if (lane == 63 && wtotal) base = atomicAdd(&L.n_s, wtotal);
            base = wave_last(base);
            hi = wave_max_u32(hi);
            if (lane == 0 && hi) atomicMax(&L.hi_bits, hi);
            unsigned slot = base + incl - cnt_mine;
#pragma unroll
            for (int u = 0; u < EPT; ++u)
                if (ent[u].x != 0u && (ent[u].x >> 12) >= sel_prefix) {
                    if (slot < (unsigned)SCAP) L.a.fin.S[slot] = ((unsigned long long)ent[u].x << 32) | (unsigned long long)(~ent[u].y);
                    ++slot;
                }
        }
        __syncthreads();
        if (L.n_s > (unsigned)SCAP) fallback = true;        // (a boundary sub-bin with thousands of entries: plateaus)
    }
    if (!fallback) {
        const unsigned n_s = L.n_s;
        const unsigned lo = sel_prefix << 12, span = L.hi_bits - lo;
        int sh = 0;
        while ((span >> sh) >= (unsigned)NBUCKET) ++sh;
        unsigned* cnt = L.hist;                             // bucket counts
        for (int i = tid; i < NBUCKET; i += NT) cnt[i] = 0u;
        __syncthreads();
        for (unsigned i = tid; i < n_s; i += NT) atomicAdd(&cnt[((unsigned)(L.a.fin.S[i] >> 32) - lo) >> sh], 1u);
        __syncthreads();
        {   // start[b] = keys in buckets above b: suffix sums, 4 consecutive buckets per thread
            const uint4 c4 = reinterpret_cast<const uint4*>(cnt)[tid];
            const unsigned mine = c4.x + c4.y + c4.z + c4.w;
            const unsigned pre = wave_incl_scan_u32(mine);
            const unsigned wtotal = wave_last(pre);
            if (lane == 0) L.wtot[wv] = wtotal;
            __syncthreads();
            unsigned higher = 0;
            for (int w = wv + 1; w < WPB; ++w) higher += L.wtot[w];
            const unsigned ab3 = higher + wtotal - pre;     // keys in buckets above this thread's four
            reinterpret_cast<uint4*>(L.start)[tid] = make_uint4(ab3 + c4.w + c4.z + c4.y, ab3 + c4.w + c4.z, ab3 + c4.w, ab3);
        }
        __syncthreads();
        // place: a bucket's keys end up in P[start_initial, start_initial + cnt) (in any order); start advances to the end
        for (unsigned i = tid; i < n_s; i += NT) {
            const unsigned long long k = L.a.fin.S[i];
            const unsigned b = ((unsigned)(k >> 32) - lo) >> sh;
            L.a.fin.P[atomicAdd(&L.start[b], 1u)] = k;
        }
        __syncthreads();
        // SCAP / NT = 4 keys per thread at most: their LDS reads are issued together, then the (short) bucket walks, then the rows
        constexpr int KPT = SCAP / NT;
        unsigned long long kk[KPT];
        unsigned e0[KPT], e1[KPT];
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const unsigned i = (unsigned)u * NT + tid;
            kk[u] = L.a.fin.S[min(i, (unsigned)SCAP - 1u)];
        }
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const unsigned i = (unsigned)u * NT + tid;
            const unsigned b = i < n_s ? ((unsigned)(kk[u] >> 32) - lo) >> sh : 0u;
            e1[u] = L.start[b];
            e0[u] = e1[u] - cnt[b];
            if (i >= n_s) e0[u] = e1[u] = 0u;
        }
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            unsigned rank = e0[u];
            for (unsigned j = e0[u]; j < e1[u]; ++j) rank += L.a.fin.P[j] > kk[u] ? 1u : 0u;
            if ((unsigned)u * NT + tid < n_s && rank < (unsigned)n_valid) d_emit_det(a_dets, (int)rank, kk[u], a_H, a_W, true);
        }
    } else {
        // exact 64-bit radix select of the K-th largest key over every candidate segment, then sort the K keys (LDS)
        // a wave takes 64 segments at a time: their counts in one load (a lane each), then segment by segment
        auto for_each = [&](auto&& f) {
            for (unsigned g0 = (unsigned)wv * 64u; g0 < a_nseg; g0 += WPB * 64u) {
                const unsigned mine = g0 + lane < a_nseg ? min(ld_u32(a_segcount + g0 + lane), a_segcap) : 0u;
                for (int sl = 0; sl < 64; ++sl) {
                    const unsigned c = __shfl(mine, sl, 64);
                    const uint2* base = a_cands + (size_t)(g0 + sl) * a_segcap;
                    for (unsigned i = lane; i < c; i += 64) f(ld_u64(base + i));
                }
            }
        };
        unsigned* s_hist = L.sub;
        if (tid == 0) { L.r_prefix = 0ull; L.r_remaining = (unsigned)max(n_valid, 1); L.n_s = 0u; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = L.r_prefix;
            const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
            for_each([&](uint2 c) {
                const unsigned long long k = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y);
                if ((k & himask) == prefix) atomicAdd(&s_hist[(unsigned)(k >> shift) & 255u], 1u);
            });
            __syncthreads();
            if (tid == 0) {
                const unsigned rem = L.r_remaining;
                unsigned acc = 0;
                int d = 255;
                for (; d > 0; --d) {
                    if (acc + s_hist[d] >= rem) break;
                    acc += s_hist[d];
                }
                L.r_remaining = rem - acc;
                L.r_prefix = prefix | ((unsigned long long)d << shift);
            }
            __syncthreads();
        }
        const unsigned long long thr = L.r_prefix;           // the n_valid-th largest key
        for_each([&](uint2 c) {
            const unsigned long long k = ((unsigned long long)c.x << 32) | (unsigned long long)(~c.y);
            if (k >= thr) {
                const unsigned slot = atomicAdd(&L.n_s, 1u);
                if (slot < (unsigned)SCAP) L.a.fin.S[slot] = k;
            }
        });
        __syncthreads();
        const int n = (int)min(L.n_s, (unsigned)SCAP);       // == n_valid (keys are unique)
        int P2 = NT;
        while (P2 < n) P2 <<= 1;
        for (int i = n + tid; i < P2; i += NT) L.a.fin.S[i] = 0ull;
        block_sort_desc_fast(L.a.fin.S, P2, tid, NT);
        for (int r = tid; r < n_valid; r += NT) d_emit_det(a_dets, r, L.a.fin.S[r], a_H, a_W, r < n);
    }
    for (int r = n_valid + tid; r < K; r += NT) d_emit_det(a_dets, r, 0ull, a_H, a_W, false);
    if (tid == 0) {
        if (p.n_valid_out) *p.n_valid_out = n_valid;
        // the header is left as it was found: zero
        st_u32(&a_hdr->cand_count, 0u);
        st_u32(&a_hdr->pad0, 0u);
    }
}

}  // namespace

size_t mi_decode1_extra_bytes(int D, int H, int W) {
    const Peak3Grid g = mi_peak3_grid(D, H, W);
    const size_t n_wg = (size_t)g.gx * g.gy * g.gz;
    return mi_align_up(n_wg * TSLOT * sizeof(uint2), 256) + mi_align_up(n_wg * sizeof(uint2), 256);
}

bool mi_decode1_usable(const float* in, const float* val_out, int D, int H, int W, int K) {
    if (getenv("MI_DECODE_CHAIN")) return false;
    const Peak3Grid g = mi_peak3_grid(D, H, W);
    // the selecting workgroup ranks K + the occupants of one boundary sub-bin in SCAP slots; the segments must hold a
    // wave's every candidate (seg_cap = all its voxels)
    return K >= 1 && K <= SCAP - 512 && (long)g.gx * g.gy * g.gz < (1l << 20) && mi_peak3_usable(in, val_out, nullptr, D, H, W);
}

// extra = the table and bound arrays (mi_decode1_extra_bytes); hdr: zero on entry, left zero
int mi_launch_decode1(const float* in, float* val_out, int D, int H, int W, bool sigmoid, int K, float* dets, int* n_valid_out,
                      DecodeHeader* hdr, uint2* cands, unsigned* seg_count, void* extra, hipStream_t s) {
    const Peak3Grid g = mi_peak3_grid(D, H, W);
    Decode1Params p = {};
    p.in = in; p.val_out = val_out; p.D = D; p.H = H; p.W = W; p.zchunk = g.zchunk;
    p.cands = cands; p.seg_count = seg_count; p.seg_cap = g.seg_cap; p.n_seg = g.n_seg;
    p.n_wg = (unsigned)((long)g.gx * g.gy * g.gz);
    p.hdr = hdr;
    p.table = (uint2*)extra;
    p.meta = (uint2*)((char*)extra + mi_align_up((size_t)p.n_wg * TSLOT * sizeof(uint2), 256));
    p.K = K;
    // entries a workgroup contributes: four times its share of K (clustered picks), at least 20 (K = 900 over 256 workgroups
    // is 3.5 each: Poisson tail beyond 20 ~ 1e-9); its slots leave room for the other occupants of the boundary sub-bin
    int M = (int)((4l * K + p.n_wg - 1) / p.n_wg);
    if (const char* e = getenv("MI_DECODE1_M")) M = atoi(e);
    p.M = std::min(std::max(M, 20), TSLOT - 8);
    p.dets = dets; p.n_valid_out = n_valid_out;
    const dim3 grid(g.gx, g.gy, g.gz);
    const bool xhalo = W > WX;
    if (sigmoid) {
        if (xhalo) hipLaunchKernelGGL((decode1_kernel<true, true>), grid, dim3(NT), 0, s, p);
        else hipLaunchKernelGGL((decode1_kernel<true, false>), grid, dim3(NT), 0, s, p);
    } else {
        if (xhalo) hipLaunchKernelGGL((decode1_kernel<false, true>), grid, dim3(NT), 0, s, p);
        else hipLaunchKernelGGL((decode1_kernel<false, false>), grid, dim3(NT), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
