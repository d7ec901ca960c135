// Greedy 3-D ball NMS with the reference's sequential semantics, run in parallel, and the DoG
// particle-picker pipeline built on it.
//
// Replaces (reference, cet_pick/...): models/decode.py:42-79 == utils/image.py:42-79
// `non_maximum_suppression_3d`, utils/image.py:138-183 `get_potential_coords_pyramid`.
//
// The reference visits ALL voxels in descending value order; a voxel is a pick iff no earlier pick
// has it in its ball (flat-offset ball: i_pick + delta == i, no bounds check).  Only voxels above
// the threshold can be picks, and a voxel can only be suppressed by a PICK, so the sequential
// result equals the unique fixed point of
//      pick(i)  <=>  no higher-priority candidate j with (i - j) in ball is a pick
// over the candidate set {value > threshold}.  We resolve it in rounds: a candidate is decided as
// soon as all its higher-priority ball-neighbours are decided.  Priority = (value, flat index),
// both descending - what a stable ascending argsort reversed yields.
#include "common.h"
#include "infer_common.h"

int mi_launch_gauss_axis(const float* in, float* out, int D, int H, int W, int axis, float sigma,
                         hipStream_t s);
int mi_launch_gauss_march(const float* in0, float* out0a, float* out0b, float sig0a, float sig0b, const float* in1,
                          float* out1, float sig1, int D, int H, int W, int axis, hipStream_t s, const int* box = nullptr,
                          unsigned* clr0 = nullptr, unsigned clr0_n = 0, unsigned* clr1 = nullptr, unsigned clr1_n = 0);
int mi_gauss_radius(float sigma);

namespace {

constexpr int CAPN = 32;          // stored higher-priority neighbours per candidate
constexpr int MAX_RUNS = 33 * 33; // (dz, dy) rows of the ball
constexpr int NB_MAX_RANGES = 1024;   // workgroups of cand_filter_seg_kernel (one list range each)

struct GreedyHeader {
    unsigned cand_count;     // positive NMS survivors (march kernel)
    unsigned n;              // candidates above the cutoff
    unsigned n_kept;
    unsigned n_left;         // candidates the chip-wide passes left open (rounds_all_kernel -> its last workgroup)
    unsigned overflow;       // bit0: candidate buffer overflow, bit1: too many picks for max_out
    float cutoff;
    unsigned n_runs;         // rows of the ball: runs of consecutive flat offsets
    unsigned ticket;         // workgroups of rounds_all_kernel that have finished their passes
    unsigned trace[16];      // candidates still undecided at the start of each chip-wide round launch (diagnostics)
};

// A row (dz, dy) of the ball is a run of consecutive flat offsets (the reference's ball lives in flat index space, no
// bounds check): first offset and length.
struct BallRun { int start; int len; };       // (|start| < 2^31: the volume has fewer voxels than that)

__device__ __forceinline__ unsigned order_bits(float v) {
    unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned u) {
    unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// ---- cutoff = mean(pos) + 0.5 * std_unbiased(pos)   (utils/image.py:177-179) -------------------
__global__ void stats_finalize_kernel(const double* partials, int n_part, GreedyHeader* hdr,
                                      float* cutoff_out) {
    __shared__ double r[3][256];
    int tid = threadIdx.x;
    double a = 0, s = 0, ss = 0;
    for (int i = tid; i < n_part; i += 256) {   // fixed order -> deterministic
        a += partials[3 * i]; s += partials[3 * i + 1]; ss += partials[3 * i + 2];
    }
    r[0][tid] = a; r[1][tid] = s; r[2][tid] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r[0][tid] += r[0][tid + o]; r[1][tid] += r[1][tid + o]; r[2][tid] += r[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        double n = r[0][0], mean = n > 0 ? r[1][0] / n : 0.0;
        double var = n > 1 ? (r[2][0] - n * mean * mean) / (n - 1.0) : 0.0;
        if (var < 0) var = 0;
        float c = (float)(mean + 0.5 * sqrt(var));
        if (!(n > 0)) c = INFINITY;     // no positive voxel: the reference would raise on mean() of empty
        hdr->cutoff = c;
        if (cutoff_out) *cutoff_out = c;
    }
}

__global__ void set_cutoff_kernel(GreedyHeader* hdr, float v) { hdr->cutoff = v; }

// ---- ball offsets (decode.py:43-54) ------------------------------------------------------------
// The ball of radius r as rows: row (dz, dy) is the run of consecutive flat offsets dz * zs + dy * ys + [-cm, cm] (the
// reference's ball lives in flat index space, no bounds check).  Every workgroup that needs the rows builds them in its own
// LDS (side^2 <= 1089 square roots) - round 3 spent a launch of one workgroup on a global copy.  Rows come out in (dz, dy)
// order: consecutive lanes of the neighbour search then read adjacent rows of the bitmap.  Returns the number of rows.
__device__ int ball_rows_lds(BallRun* s_runs, int* s_scan, double r, int width, long zs, long ys) {
    const int side = 2 * width + 1, total = side * side;
    const double r2 = r * r;
    const int nt = blockDim.x, tid = threadIdx.x;
    const int per = (total + nt - 1) / nt;                 // consecutive entries per thread
    int cm[5];                                             // (total <= 1089, nt >= 256: per <= 5)
    int mine = 0;
    for (int k = 0; k < per && k < 5; ++k) {
        const int t = tid * per + k;
        int c = -1;
        if (t < total) {
            const int a = t / side - width, b = t % side - width;
            const double rest = r2 - (double)(a * a + b * b);
            if (rest >= 0.0) {
                c = (int)sqrt(rest);
                while ((double)((c + 1) * (c + 1)) <= rest) ++c;         // exact integer bound of c^2 <= rest
                while ((double)(c * c) > rest) --c;
                if (c > width) c = width;
            }
        }
        cm[k] = c;
        mine += c >= 0;
    }
    s_scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < nt; o <<= 1) {                     // inclusive scan (nt <= 1024 entries)
        const int v = tid >= o ? s_scan[tid - o] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int pos = s_scan[tid] - mine;
    for (int k = 0; k < per && k < 5; ++k) {
        const int t = tid * per + k;
        if (cm[k] >= 0) {
            const int a = t / side - width, b = t % side - width;
            s_runs[pos++] = BallRun{(int)((long)a * zs + (long)b * ys - (long)cm[k]), 2 * cm[k] + 1};
        }
    }
    const int n = s_scan[nt - 1];
    __syncthreads();
    return n;
}

// ---- dense volume -> candidate list (values > cutoff) ------------------------------------------
__global__ __launch_bounds__(256) void dense_filter_kernel(const float* vol, size_t n_vox,
                                                          GreedyHeader* hdr,
                                                          unsigned long long* G, int* map, unsigned* bits,
                                                          unsigned cap) {
    const float cut = hdr->cutoff;
    const size_t stride = (size_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count: the slot counter is bumped once per wave (one atomic for all its candidates), not once
    // per candidate
    for (size_t i0 = (size_t)blockIdx.x * 256; i0 < n_vox; i0 += stride) {
        const size_t i = i0 + threadIdx.x;
        const float v = i < n_vox ? vol[i] : 0.f;
        const bool keep = i < n_vox && v > cut;
        const unsigned long long km = __ballot(keep);
        if (!km) continue;
        unsigned base = 0;
        const int leader = __ffsll((long long)km) - 1;
        if (lane == leader) base = atomicAdd(&hdr->n, (unsigned)__popcll(km));
        base = __shfl(base, leader, 64);
        if (keep) {
            const unsigned slot = base + (unsigned)__popcll(km & ((1ull << lane) - 1ull));
            if (slot < cap) {
                G[slot] = ((unsigned long long)order_bits(v) << 32) | (unsigned long long)i;
                map[i] = (int)slot;
                atomicOr(&bits[i >> 5], 1u << (i & 31));
            } else {
                atomicOr(&hdr->overflow, 1u);
            }
        }
    }
}

// ---- sparse candidates (score bits, idx) -> candidate list (values > cutoff) -------------------
__global__ __launch_bounds__(256) void cand_filter_kernel(const uint2* cands, unsigned cand_cap,
                                                         GreedyHeader* hdr, unsigned long long* G,
                                                         int* map, unsigned* vmap, unsigned* bits, unsigned cap) {
    const float cut = hdr->cutoff;
    unsigned total = hdr->cand_count;
    if (total > cand_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&hdr->overflow, 1u); total = cand_cap; }
    const int lane = threadIdx.x & 63;
    for (unsigned i0 = blockIdx.x * 256; i0 < total; i0 += gridDim.x * 256) {     // wave-uniform trip count
        const unsigned i = i0 + threadIdx.x;
        const uint2 c = i < total ? cands[i] : make_uint2(0u, 0u);
        const float v = __uint_as_float(c.x);
        const bool keep = i < total && v > cut;
        const unsigned long long km = __ballot(keep);
        if (!km) continue;
        unsigned base = 0;
        const int leader = __ffsll((long long)km) - 1;
        if (lane == leader) base = atomicAdd(&hdr->n, (unsigned)__popcll(km));   // one atomic per wave
        base = __shfl(base, leader, 64);
        if (keep) {
            const unsigned slot = base + (unsigned)__popcll(km & ((1ull << lane) - 1ull));
            if (slot < cap) {
                G[slot] = ((unsigned long long)order_bits(v) << 32) | (unsigned long long)c.y;
                map[c.y] = (int)slot;
                vmap[c.y] = order_bits(v);
                atomicOr(&bits[c.y >> 5], 1u << (c.y & 31));
            } else {
                atomicOr(&hdr->overflow, 1u);
            }
        }
    }
}

// ---- segmented candidates (infer_dogx.hip: one segment per wave of the fused kernel) -> candidate list ----------
// The survivors of a workgroup are staged in LDS and take their slots in G with ONE returning atomic per workgroup: a
// returning atomic on a single word costs ~11 ns, and one per wave and 64-candidate step (~10^4 of them for 83 k
// survivors) made the linear-list kernel above 58 us for 2.4 MB of candidates.
// FIN: the cutoff (mean + 0.5 std of the positive survivors, stats_finalize_kernel's arithmetic) is computed by EVERY
// workgroup from the partial sums in one fixed order - the same bits everywhere - instead of by a launch of its own in
// front of this one; workgroup 0 publishes it.
template <bool FIN>
__global__ __launch_bounds__(256) void cand_filter_seg_kernel(const uint2* cands, const unsigned* seg_count,
                                                             unsigned n_seg, unsigned seg_cap, GreedyHeader* hdr,
                                                             unsigned long long* G, int* map, unsigned* vmap,
                                                             unsigned* bits, unsigned cap, const double* partials,
                                                             int n_part, float* cutoff_out, uint2* ranges) {
    __shared__ double s_red[3 * 256];
    __shared__ unsigned s_n, s_base;
    __shared__ float s_cut;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0;
    if (FIN) {
        double* r = s_red;
        double a = 0, s = 0, ss = 0;
        for (int i = tid; i < n_part; i += 256) { a += partials[3 * i]; s += partials[3 * i + 1]; ss += partials[3 * i + 2]; }
        r[tid] = a; r[256 + tid] = s; r[512 + tid] = ss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { r[tid] += r[tid + o]; r[256 + tid] += r[256 + tid + o]; r[512 + tid] += r[512 + tid + o]; }
            __syncthreads();
        }
        if (tid == 0) {
            const double n = r[0], mean = n > 0 ? r[256] / n : 0.0;
            double var = n > 1 ? (r[512] - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0) var = 0;
            float c = (float)(mean + 0.5 * sqrt(var));
            if (!(n > 0)) c = INFINITY;
            s_cut = c;
            if (blockIdx.x == 0) { hdr->cutoff = c; if (cutoff_out) *cutoff_out = c; }
        }
    }
    __syncthreads();
    const float cut = FIN ? s_cut : hdr->cutoff;
    auto place = [&](uint2 c, unsigned slot) {
        if (slot < cap) {
            G[slot] = ((unsigned long long)order_bits(__uint_as_float(c.x)) << 32) | (unsigned long long)c.y;
            map[c.y] = (int)slot;
            vmap[c.y] = order_bits(__uint_as_float(c.x));
            atomicOr(&bits[c.y >> 5], 1u << (c.y & 31));
        } else {
            atomicOr(&hdr->overflow, 1u);
        }
    };
    // A workgroup takes CONSECUTIVE segments (= neighbouring strips / row chunks / z planes) and its survivors ONE contiguous
    // range of the list, recorded in `ranges[workgroup]`: walked range by range the list is in plane order whatever order
    // the workgroups finished in - the neighbour search then works on one slab of the volume at a time (its bitmap lines stay
    // in L2; in completion order the whole 8 MB bitmap was the working set: 0.32 GB of fetches).  Two passes over the
    // segments: count, take the range with one atomic, place (the second read comes from L2).
    const unsigned per_wg = (n_seg + gridDim.x - 1) / gridDim.x;
    const unsigned sg_end = min((blockIdx.x + 1) * per_wg, n_seg);
    constexpr int PF = 4;
    unsigned mine = 0;
    for (unsigned sg = blockIdx.x * per_wg + (tid >> 6); sg < sg_end; sg += 4) {
        const unsigned cnt = min(seg_count[sg], seg_cap);
        const uint2* base = cands + (size_t)sg * seg_cap;
        for (unsigned i0 = 0; i0 < cnt; i0 += 64 * PF) {
            uint2 c[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {                        // all loads of the step in flight together
                const unsigned i = i0 + 64 * u + lane;
                c[u] = i < cnt ? base[i] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) mine += (i0 + 64 * u + lane < cnt && __uint_as_float(c[u].x) > cut) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&s_n, mine);               // LDS
    __syncthreads();
    if (tid == 0) {
        const unsigned n = s_n;
        s_base = n ? atomicAdd(&hdr->n, n) : 0u;                // ONE returning atomic per workgroup (~11 ns each on one word)
        if (ranges) ranges[blockIdx.x] = make_uint2(s_base, n);
        s_n = s_base;                                            // becomes the placement cursor
    }
    __syncthreads();
    for (unsigned sg = blockIdx.x * per_wg + (tid >> 6); sg < sg_end; sg += 4) {
        const unsigned cnt = min(seg_count[sg], seg_cap);
        const uint2* base = cands + (size_t)sg * seg_cap;
        for (unsigned i0 = 0; i0 < cnt; i0 += 64 * PF) {
            uint2 c[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const unsigned i = i0 + 64 * u + lane;
                c[u] = i < cnt ? base[i] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (i0 + 64 * u >= cnt) break;                    // (wave-uniform)
                const unsigned i = i0 + 64 * u + lane;
                const bool keep = i < cnt && __uint_as_float(c[u].x) > cut;
                const unsigned long long km = __ballot(keep);
                if (!km) continue;
                unsigned b0 = 0;
                const int leader = __ffsll((long long)km) - 1;
                if (lane == leader) b0 = atomicAdd(&s_n, (unsigned)__popcll(km));     // LDS
                b0 = __shfl(b0, leader, 64);
                if (keep) place(c[u], b0 + (unsigned)__popcll(km & ((1ull << lane) - 1ull)));
            }
        }
    }
}

// ---- the fused picker's candidates as a SPATIAL INDEX (round 4) ------------------------------------------------------
// The y march (infer_dogf.hip) leaves its survivors in segments: one per (plane, chunk of rows, strip of 62 columns), in
// row order.  Filtered by the cutoff IN that order, every segment becomes a contiguous, row-sorted run of the candidate
// list, and (start, count) per segment is a grid index of the candidates: the ball of a candidate (15 planes, radius 7)
// overlaps ~40 segments and a binary search on the row finds the one or two entries of each that can matter.  The bitmap
// search of round 3 probed every row of the ball - 149 scattered 8-byte loads per candidate, 97 % of them empty - through
// an 8 MB bitmap, a dense id map and a dense value map (60 - 70 us, 0.33 GB of fetches, all three cleared / scattered per
// call); none of the three exists on this path.  Exactness: the reference's ball lives in flat index space; with the picker's
// zeroed border (>= 30 voxels in x / y) wider than the ball no row of it wraps, so the coordinate test is the same set.
struct SegIndex {
    const uint2* seg_range;      // per segment: (first slot, count) in the candidate list; null = no index (bitmap search)
    const unsigned* coord;       // per slot: (y << 16) | x
    const unsigned* coordz;      // per slot: z
    const unsigned* sub;         // per segment nb + 1 slots: sub[r] = first entry of the segment in row block r or later (8 rows
    int nb;                      // per block; sub[nb] = end of the segment): a row window is two loads, not a binary search
    int D, H, W, bz, by, bx;
    int ychunk, n_ychunks, n_strips, own;
    int width;
    double r2;
};

// calls fn(slot) for every candidate of HIGHER priority inside the ball of candidate (ki, z, y, x)
template <typename F>
__device__ __forceinline__ void seg_for_each_higher(const SegIndex& sx, const unsigned long long* G, unsigned long long ki,
                                                    int z, int y, int x, F&& fn) {
    const int w = sx.width;
    const int z_lo = max(z - w, sx.bz), z_hi = min(z + w, sx.D - sx.bz - 1);
    const int y_lo = max(y - w, sx.by), y_hi = min(y + w, sx.H - sx.by - 1);
    const int x_lo = max(x - w, sx.bx), x_hi = min(x + w, sx.W - sx.bx - 1);
    const int c_lo = (y_lo - sx.by) / sx.ychunk, c_hi = (y_hi - sx.by) / sx.ychunk;
    const int s_lo = (x_lo - sx.bx) / sx.own, s_hi = (x_hi - sx.bx) / sx.own;
    for (int zz = z_lo; zz <= z_hi; ++zz) {
        const int dz = zz - z;
        const double rest = sx.r2 - (double)(dz * dz);
        if (rest < 0.0) continue;
        for (int c = c_lo; c <= c_hi; ++c)
            for (int st = s_lo; st <= s_hi; ++st) {
                const uint2 rg = sx.seg_range[((size_t)(zz - sx.bz) * sx.n_ychunks + c) * sx.n_strips + st];
                if (rg.y == 0) continue;
                // first entry with row >= y_lo (entries are in row order)
                unsigned lo = 0, hi = rg.y;
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if ((int)(sx.coord[rg.x + mid] >> 16) < y_lo) lo = mid + 1; else hi = mid;
                }
                for (unsigned e = lo; e < rg.y; ++e) {
                    const unsigned cj = sx.coord[rg.x + e];
                    const int yj = (int)(cj >> 16), xj = (int)(cj & 0xffffu);
                    if (yj > y_hi) break;
                    const int dy = yj - y, dx = xj - x;
                    if ((double)(dy * dy + dx * dx) > rest) continue;
                    if (G[rg.x + e] > ki) fn((int)(rg.x + e));
                }
            }
    }
}

// ---- segmented candidates -> row-sorted runs of the list + the index ------------------------------------------------------
// As cand_filter_seg_kernel<true> (cutoff in the prologue, one list range per workgroup), but a SEGMENT's survivors stay
// together and in order, and nothing else is written: no bitmap, no id map, no value map.
constexpr int CI_MAXSEG = 64;        // segments per workgroup (the host sizes the grid accordingly)
__global__ __launch_bounds__(256) void cand_index_kernel(const uint2* cands, const unsigned* seg_count, unsigned n_seg,
                                                        unsigned seg_cap, GreedyHeader* hdr, unsigned long long* G,
                                                        unsigned* coord, unsigned* coordz, uint2* seg_range, unsigned cap,
                                                        const double* partials, int n_part, float* cutoff_out,
                                                        uint2* ranges, int H, int W, unsigned* sub, int nb, int by,
                                                        int ychunk, int n_ychunks, int n_strips) {
    __shared__ double s_red[3 * 256];
    __shared__ unsigned s_cnt[CI_MAXSEG], s_start[CI_MAXSEG];
    __shared__ float s_cut;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        double a = 0, s = 0, ss = 0;
        // four partial rows in flight, added in the rolled loop's order (same sums)
        int i = tid;
        for (; i + 3 * 256 < n_part; i += 4 * 256) {
            double v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) v[u][q] = partials[3 * (i + 256 * u) + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += v[u][0]; s += v[u][1]; ss += v[u][2]; }
        }
        for (; i < n_part; i += 256) { a += partials[3 * i]; s += partials[3 * i + 1]; ss += partials[3 * i + 2]; }
        s_red[tid] = a; s_red[256 + tid] = s; s_red[512 + tid] = ss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { s_red[tid] += s_red[tid + o]; s_red[256 + tid] += s_red[256 + tid + o]; s_red[512 + tid] += s_red[512 + tid + o]; }
            __syncthreads();
        }
        if (tid == 0) {
            const double n = s_red[0], mean = n > 0 ? s_red[256] / n : 0.0;
            double var = n > 1 ? (s_red[512] - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0) var = 0;
            float c = (float)(mean + 0.5 * sqrt(var));
            if (!(n > 0)) c = INFINITY;
            s_cut = c;
            if (blockIdx.x == 0) { hdr->cutoff = c; if (cutoff_out) *cutoff_out = c; }
        }
        __syncthreads();
    }
    const float cut = s_cut;
    const unsigned per_wg = (n_seg + gridDim.x - 1) / gridDim.x;            // <= CI_MAXSEG
    const unsigned sg0 = blockIdx.x * per_wg, sg_end = min(sg0 + per_wg, n_seg);
    constexpr int PF = 4;
    // pass 1: survivors per segment (a wave per segment)
    for (unsigned sg = sg0 + wv; sg < sg_end; sg += 4) {
        const unsigned cnt = min(seg_count[sg], seg_cap);
        const uint2* base = cands + (size_t)sg * seg_cap;
        unsigned mine = 0;
        for (unsigned i0 = 0; i0 < cnt; i0 += 64 * PF) {
            uint2 c[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {                        // all loads of the step in flight together
                const unsigned i = i0 + 64 * u + lane;
                c[u] = i < cnt ? base[i] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) mine += (i0 + 64 * u + lane < cnt && __uint_as_float(c[u].x) > cut) ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if (lane == 0) s_cnt[sg - sg0] = mine;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned tot = 0;
        for (unsigned q = 0; q < sg_end - sg0 && sg0 < sg_end; ++q) tot += s_cnt[q];
        unsigned base = tot ? atomicAdd(&hdr->n, tot) : 0u;     // ONE returning atomic per workgroup
        if (base + tot > cap) { atomicOr(&hdr->overflow, 1u); }
        if (ranges) ranges[blockIdx.x] = make_uint2(base, tot);
        for (unsigned q = 0; sg0 + q < sg_end; ++q) { s_start[q] = base; base += s_cnt[q]; }
    }
    __syncthreads();
    if (tid < (int)(sg_end > sg0 ? sg_end - sg0 : 0u)) {
        const unsigned st = s_start[tid], ct = s_cnt[tid];
        seg_range[sg0 + tid] = (st + ct <= cap) ? make_uint2(st, ct) : make_uint2(0u, 0u);
    }
    // pass 2: place, in order (the second read comes from L2)
    const unsigned hw = (unsigned)H * (unsigned)W;         // (flat indices are 32-bit here: 32-bit divisions, a 64-bit one is ~150 instructions)
    for (unsigned sg = sg0 + wv; sg < sg_end; sg += 4) {
        const unsigned cnt = min(seg_count[sg], seg_cap);
        const uint2* base = cands + (size_t)sg * seg_cap;
        unsigned pos = s_start[sg - sg0];
        unsigned* sb = sub + (size_t)sg * (nb + 1);
        if (pos + s_cnt[sg - sg0] > cap) {                      // (overflow: flagged above, the segment is dropped)
            for (int r = lane; r <= nb; r += 64) sb[r] = 0u;
            continue;
        }
        const int ya = by + (int)((sg / n_strips) % n_ychunks) * ychunk;      // first row of the segment's chunk
        int rb_last = -1;                                       // row block of the last entry placed so far (wave-uniform)
        for (unsigned i0 = 0; i0 < cnt; i0 += 64 * PF) {
            uint2 c[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const unsigned i = i0 + 64 * u + lane;
                c[u] = i < cnt ? base[i] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (i0 + 64 * u >= cnt) break;                    // (wave-uniform)
                const unsigned i = i0 + 64 * u + lane;
                const bool keep = i < cnt && __uint_as_float(c[u].x) > cut;
                const unsigned long long km = __ballot(keep);
                if (!km) continue;
                const unsigned long long below = km & ((1ull << lane) - 1ull);
                const unsigned zq = c[u].y / hw, rem = c[u].y - zq * hw;
                const int yy = (int)(rem / (unsigned)W);
                int rb = keep ? min(max((yy - ya) >> 3, 0), nb - 1) : 0;
                // the row block of the previous survivor: the nearest kept lane below, or the last one of the step before
                const int prev_lane = below ? 63 - __builtin_clzll(below) : 0;
                const int rb_from = __shfl(rb, prev_lane, 64);
                const int rb_prev = below ? rb_from : rb_last;
                if (keep) {
                    const unsigned slot = pos + (unsigned)__popcll(below);
                    G[slot] = ((unsigned long long)order_bits(__uint_as_float(c[u].x)) << 32) | (unsigned long long)c[u].y;
                    coord[slot] = ((unsigned)yy << 16) | (rem - (unsigned)yy * (unsigned)W);
                    coordz[slot] = zq;
                    for (int r = rb_prev + 1; r <= rb; ++r) sb[r] = slot;       // blocks that start at this entry
                }
                rb_last = __shfl(rb, 63 - __builtin_clzll(km), 64);
                pos += (unsigned)__popcll(km);
            }
        }
        for (int r = rb_last + 1 + lane; r <= nb; r += 64) sb[r] = pos;        // the blocks behind the last entry, and the end
    }
}

// ---- neighbour lists from the index: a WAVE per candidate, a LANE per segment of its ball --------------------------------
// The ball of a candidate overlaps (2 w + 1) planes x <= 2 chunks x <= 2 strips = at most 60 segments at w = 7: lane v takes
// segment v, finds the candidate's row window in it with two loads of the sub-index, and the lanes then walk their windows
// in lockstep - entry t of every lane in one step, coordinate and key loaded together - compacting the hits into the row of
// `nbr` by ballot.  A candidate costs 1 + (longest window, ~4) memory round trips, its successor's key and coordinates are
// in flight meanwhile.  (A thread per candidate walking its ~40 segments by itself: 129 us - 280 dependent loads in a row.)
__global__ __launch_bounds__(256) void neighbors_seg_kernel(GreedyHeader* hdr, const unsigned long long* G, SegIndex sx,
                                                           unsigned cap, int* nbr, unsigned char* state) {
    const unsigned n = min(hdr->n, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr->n_runs = 0; hdr->n_kept = 0; hdr->n_left = 0; hdr->ticket = 0; }
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), n_waves = (gridDim.x * 256) >> 6;
    const int w = sx.width;
    // (no integer division in the loop: the kernel was bound by its own instruction stream - ~8 emulated divisions per
    // candidate; rows and columns are < 2^16, so n / d = (n * ceil(2^32 / d)) >> 32 exactly)
    const unsigned m_chunk = (unsigned)((0x100000000ull + (unsigned)sx.ychunk - 1) / (unsigned)sx.ychunk);
    const unsigned m_own = (unsigned)((0x100000000ull + (unsigned)sx.own - 1) / (unsigned)sx.own);
    unsigned long long k_next = wave < n ? G[wave] : 0ull;
    unsigned c_next = wave < n ? sx.coord[wave] : 0u, z_next = wave < n ? sx.coordz[wave] : 0u;
    for (unsigned i = wave; i < n; i += n_waves) {
        // the candidate is the same for the whole wave: scalar registers
        const unsigned kilo = __builtin_amdgcn_readfirstlane((unsigned)k_next), kihi = __builtin_amdgcn_readfirstlane((unsigned)(k_next >> 32));
        const unsigned long long ki = ((unsigned long long)kihi << 32) | kilo;
        const unsigned ci = __builtin_amdgcn_readfirstlane(c_next);
        const int z = (int)__builtin_amdgcn_readfirstlane(z_next);
        if (i + n_waves < n) { k_next = G[i + n_waves]; c_next = sx.coord[i + n_waves]; z_next = sx.coordz[i + n_waves]; }
        const int y = (int)(ci >> 16), x = (int)(ci & 0xffffu);
        const int z_lo = max(z - w, sx.bz), z_hi = min(z + w, sx.D - sx.bz - 1);
        const int y_lo = max(y - w, sx.by), y_hi = min(y + w, sx.H - sx.by - 1);
        const int x_lo = max(x - w, sx.bx), x_hi = min(x + w, sx.W - sx.bx - 1);
        const int c_lo = (int)__umulhi((unsigned)(y_lo - sx.by), m_chunk), nc = (int)__umulhi((unsigned)(y_hi - sx.by), m_chunk) - c_lo + 1;
        const int s_lo = (int)__umulhi((unsigned)(x_lo - sx.bx), m_own), ns = (int)__umulhi((unsigned)(x_hi - sx.bx), m_own) - s_lo + 1;
        // nc, ns are 1 or 2 (a chunk is >= 48 rows, a strip 62 columns, the window <= 33)
        const int sh_s = ns - 1, sh_c = nc - 1;
        const int n_visits = ((z_hi - z_lo + 1) << sh_c) << sh_s;
        int count = 0;
        int* row = nbr + (size_t)i * CAPN;
        for (int v0 = 0; v0 < n_visits; v0 += 64) {
            const int v = v0 + lane;
            unsigned e0 = 0, e1 = 0;
            int rest = -1;                                   // floor(r^2 - dz^2): dy^2 + dx^2 is an integer
            if (v < n_visits) {
                const int st = s_lo + (v & sh_s), c = c_lo + ((v >> sh_s) & sh_c), zz = z_lo + (v >> (sh_s + sh_c));
                const int dz = zz - z;
                const double rd = sx.r2 - (double)(dz * dz);
                if (rd >= 0.0) {
                    rest = (int)rd;
                    const unsigned sg = (unsigned)((zz - sx.bz) * sx.n_ychunks + c) * (unsigned)sx.n_strips + (unsigned)st;
                    const int ya = sx.by + c * sx.ychunk;
                    const int rb0 = min(max((y_lo - ya) >> 3, 0), sx.nb - 1), rb1 = min(max((y_hi - ya) >> 3, 0), sx.nb - 1);
                    const unsigned* sb = sx.sub + (size_t)sg * (unsigned)(sx.nb + 1);
                    e0 = sb[rb0]; e1 = sb[rb1 + 1];
                }
            }
            for (unsigned t = 0; __ballot(e0 + t < e1); ++t) {
                int m = -1;
                const unsigned slot = e0 + t;
                bool near = false;
                if (slot < e1) {
                    const unsigned cj = sx.coord[slot];
                    const int dy = (int)(cj >> 16) - y, dx = (int)(cj & 0xffffu) - x;
                    near = dy * dy + dx * dx <= rest;
                }
                if (!__ballot(near)) continue;               // (nine entries of ten lie outside the ball: no key is fetched for them)
                if (near && G[slot] > ki) m = (int)slot;
                const unsigned long long ball = __ballot(m >= 0);
                if (ball) {
                    if (m >= 0) {
                        const int pos = count + __popcll(ball & ((1ull << lane) - 1ull));
                        if (pos < CAPN - 1) row[1 + pos] = m;
                    }
                    count += __popcll(ball);
                }
            }
        }
        if (lane == 0) { row[0] = count; state[i] = 0; }       // count > CAPN-1 -> overflow: re-probe in the rounds
    }
}

// ---- higher-priority ball neighbours of every candidate (one wave per candidate) ---------------
// The candidates are sparse (one voxel in ~1000), so "is there a candidate at voxel j" is answered by a BITMAP of the
// volume (1 bit per voxel: 8 MB for 256 x 512 x 512, resident in L2) and the dense id map - which is then never cleared
// and only read where a bit is set - is touched for actual candidates only.  A lane takes one row of the ball (a run of
// <= 33 consecutive flat offsets = one or two bitmap words) instead of one offset.
// Row i of `nbr` = [count, neighbour slots ...] (CAPN ints: count and the first three neighbours come with one 16-byte
// load).  The priority of a hit voxel is read from a dense value array (the volume itself, or order_bits scattered next
// to the id map) - independent of the id-map load, not behind it - and the candidate's own voxel is masked out before
// the hits are popped: half of the candidates have no other candidate in their ball and skip the loop altogether.
// The kernel also opens the rounds (state = undecided, counters).
__global__ __launch_bounds__(256) void neighbors_kernel(GreedyHeader* hdr,
                                                       const unsigned long long* G, const int* map,
                                                       const unsigned* vmap, const float* vol,
                                                       const unsigned* bits, BallRun* runs, long n_vox,
                                                       unsigned cap, int* nbr, unsigned char* state, double ball_r,
                                                       int ball_width, long zs, long ys, const uint2* ranges, int n_ranges) {
    const unsigned n = min(hdr->n, cap);
    const int lane = threadIdx.x & 63;
    // the list range by range (cand_filter_seg_kernel: plane order); s_pref[b] = candidates in the ranges before b
    __shared__ unsigned s_pref[NB_MAX_RANGES + 1], s_rbase[NB_MAX_RANGES];
    if (n_ranges > 0) {
        for (int b = threadIdx.x; b < n_ranges; b += 256) { const uint2 rg = ranges[b]; s_rbase[b] = rg.x; s_pref[b + 1] = rg.y; }
        if (threadIdx.x == 0) s_pref[0] = 0;
        __syncthreads();
        if (threadIdx.x < 64) {                             // inclusive scan of <= 1024 counts by one wave
            unsigned carry = 0;
            for (int b0 = 0; b0 < n_ranges; b0 += 64) {
                const int b = b0 + lane;
                unsigned v = b < n_ranges ? s_pref[b + 1] : 0u;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
                if (b < n_ranges) s_pref[b + 1] = v + carry;
                carry += __shfl(v, 63, 64);
            }
        }
        __syncthreads();
    }
    auto list_index = [&](unsigned k) -> unsigned {         // k-th candidate in range order -> its slot in G
        if (n_ranges <= 0) return k;
        int lo = 0, hi = n_ranges;                          // last b with s_pref[b] <= k
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= k) lo = mid; else hi = mid; }
        return s_rbase[lo] + (k - s_pref[lo]);
    };
    // Candidates are dealt round-robin to the waves: the list is filled roughly plane by plane, so at any moment the whole
    // chip works on one slab of the volume and that slab of the 8 MB bitmap is hot in every XCD's L2.  (Contiguous ranges
    // per wave with one z-slab per XCD were measured slower: 85 us against 59 us.)
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned n_waves = (gridDim.x * 256) >> 6;
    // the ball's rows, built by every workgroup in its own LDS; workgroup 0 publishes them for the rounds' re-probe of the
    // (rare) candidates whose neighbour list overflowed
    __shared__ BallRun s_runs[MAX_RUNS];
    __shared__ int s_scan[256];
    const int nr = ball_rows_lds(s_runs, s_scan, ball_r, ball_width, zs, ys);
    if (blockIdx.x == 0) {
        for (int q = threadIdx.x; q < nr; q += 256) runs[q] = s_runs[q];
        if (threadIdx.x == 0) { hdr->n_runs = (unsigned)nr; hdr->n_kept = 0; hdr->n_left = 0; hdr->ticket = 0; }
    }
    constexpr int QB = 4;                         // row groups whose bitmap words are fetched together
    unsigned i_next = wave < n ? list_index(wave) : 0u;
    unsigned long long k_next = wave < n ? G[i_next] : 0ull;
    for (unsigned kk = wave; kk < n; kk += n_waves) {
        const unsigned i = i_next;
        const unsigned long long ki = k_next;              // (the next candidate's key is in flight during this one's search)
        if (kk + n_waves < n) { i_next = list_index(kk + n_waves); k_next = G[i_next]; }
        const long idx = (long)(ki & 0xffffffffull);
        const unsigned vi = (unsigned)(ki >> 32);
        int count = 0;
        for (int q0 = 0; q0 < nr; q0 += 64 * QB) {
            // this lane's rows, clipped to the volume: voxels [lo, hi); candidate bits of [lo, lo + 64) - a row covers
            // at most 33 voxels.  All QB rows' words are in flight before the first is used.
            // (a row is at most 33 voxels and starts at most 31 bits into its first word: two words always hold it -
            // one 8-byte load; the bitmap has two words of slack behind the volume)
            long lo[QB];
            unsigned long long pend[QB];
            uint2 bw[QB];
            int len[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int q = q0 + 64 * u + lane;
                lo[u] = 0; len[u] = 0; bw[u] = make_uint2(0u, 0u);
                if (q < nr) {
                    const BallRun rn = s_runs[q];
                    const long l = max(idx + rn.start, 0l), h = min(idx + rn.start + rn.len, n_vox);
                    if (h > l) {
                        lo[u] = l; len[u] = (int)(h - l);
                        typedef unsigned u2a4 __attribute__((ext_vector_type(2), aligned(4)));
                        const u2a4 t2 = *reinterpret_cast<const u2a4*>(bits + (l >> 5));     // one dwordx2 load
                        bw[u] = make_uint2(t2.x, t2.y);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int sh = (int)(lo[u] & 31);
                const unsigned long long win = ((unsigned long long)bw[u].x | ((unsigned long long)bw[u].y << 32)) >> sh;
                pend[u] = len[u] > 0 ? (win & ((1ull << len[u]) - 1ull)) : 0ull;     // len <= 33 <= 64 - sh
                if (idx >= lo[u] && idx < lo[u] + len[u]) pend[u] &= ~(1ull << (idx - lo[u]));   // not its own neighbour
            }
            // pop the set bits in wave-uniform steps, compacting the hits into the neighbour list.  A step takes up to TWO
            // hits per lane out of ANY of its rows and has all four loads (id, priority) in flight together: the steps of a
            // candidate are its longest lane's hits / 2 - round 3 ran one loop per row group with one hit per trip, each
            // trip a memory round trip of its own (3 - 6 per candidate)
            while (true) {
                long j[2] = {-1, -1};
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < QB; ++u)
                        if (j[t] < 0 && pend[u]) {
                            const int b = __ffsll((long long)pend[u]) - 1;
                            pend[u] &= pend[u] - 1ull;
                            j[t] = lo[u] + b;
                        }
                if (!__ballot(j[0] >= 0)) break;
                int mm[2] = {-1, -1};
                unsigned vj[2] = {0u, 0u};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (j[t] >= 0) { mm[t] = map[j[t]]; vj[t] = vol ? order_bits(vol[j[t]]) : vmap[j[t]]; }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int m = (j[t] >= 0 && (vj[t] > vi || (vj[t] == vi && j[t] > idx))) ? mm[t] : -1;   // (value, index) both descending
                    const unsigned long long ball = __ballot(m >= 0);
                    if (m >= 0) {
                        const int pos = count + __popcll(ball & ((1ull << lane) - 1ull));
                        if (pos < CAPN - 1) nbr[(size_t)i * CAPN + 1 + pos] = m;
                    }
                    count += __popcll(ball);
                }
            }
        }
        if (lane == 0) {
            nbr[(size_t)i * CAPN] = count; state[i] = 0;        // count > CAPN-1 -> overflow: re-probe in the rounds
            if (count > CAPN - 1) atomicAdd(&hdr->trace[13], 1u);   // (diagnostics: lists that did not fit - rare.  An
            // atomicMax of every count on one word here made this kernel 952 us instead of 60: ~11 ns per returning atomic)
        }
    }
}

// ---- rounds: single workgroup, all candidates --------------------------------------------------
// state: 0 undecided, 1 pick, 2 suppressed
__device__ __forceinline__ int decide(unsigned i, int4 r0, const unsigned long long* G, const int* map, const unsigned* bits,
                                      const BallRun* runs, int nr, long n_vox, const int* nbr,
                                      const volatile unsigned char* state, const SegIndex& sx) {
    const int cnt = r0.x;                                  // r0 = the first 16 bytes of row i: count + three neighbours
    bool all_decided = true;
    if (cnt <= CAPN - 1) {
        // the states of the first three neighbours are independent loads; the rare longer lists walk the row
        const unsigned char s0 = cnt > 0 ? state[r0.y] : 2, s1 = cnt > 1 ? state[r0.z] : 2, s2 = cnt > 2 ? state[r0.w] : 2;
        if (s0 == 1 || s1 == 1 || s2 == 1) return 2;
        all_decided = s0 != 0 && s1 != 0 && s2 != 0;
        // longer lists (up to 18 neighbours on the 256x512x512 benchmark volume): eight at a time, their ids with two
        // 16-byte loads and their eight states in flight together - a wave waits for its longest list, and one id load
        // plus one dependent state load per neighbour was most of a round's 38 us
        const int4* row4 = reinterpret_cast<const int4*>(nbr + (size_t)i * CAPN);
        for (int q0 = 3; q0 < cnt; q0 += 8) {
            const int4 a = row4[(q0 + 1) >> 2];
            const int4 b = (q0 + 4 < cnt) ? row4[((q0 + 1) >> 2) + 1] : make_int4(0, 0, 0, 0);     // (stays inside the row)
            const int ids[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            unsigned char st[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) st[u] = (q0 + u < cnt) ? state[ids[u]] : (unsigned char)2;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (st[u] == 1) return 2;
                if (st[u] == 0) all_decided = false;
            }
        }
    } else if (sx.seg_range) {
        const unsigned long long ki = G[i];
        const long idx = (long)(ki & 0xffffffffull);
        const unsigned cj = sx.coord[i];
        bool pick_near = false;
        (void)idx;
        seg_for_each_higher(sx, G, ki, (int)sx.coordz[i], (int)(cj >> 16), (int)(cj & 0xffffu), [&](int slot) {
            const unsigned char st = state[slot];
            if (st == 1) pick_near = true;
            if (st == 0) all_decided = false;
        });
        if (pick_near) return 2;
    } else {
        const unsigned long long ki = G[i];
        const long idx = (long)(ki & 0xffffffffull);
        for (int q = 0; q < nr; ++q) {
            const BallRun rn = runs[q];
            for (int c = 0; c < rn.len; ++c) {
                const long j = idx + rn.start + c;
                if (j < 0 || j >= n_vox) continue;
                if (!((bits[j >> 5] >> (j & 31)) & 1u)) continue;   // (the id map is only valid where a bit is set)
                const int mm = map[j];
                if ((unsigned)mm == i || !(G[mm] > ki)) continue;
                const unsigned char st = state[mm];
                if (st == 1) return 2;
                if (st == 0) all_decided = false;
            }
        }
    }
    return all_decided ? 1 : 0;
}

// ---- rounds: ONE launch ------------------------------------------------------------------------------------------
// Round 3 resolved the fixed point in three chip-wide launches + a one-workgroup finisher, every wave taking its picks'
// slots and its share of an `undecided` counter with returning atomics on two words (~11 ns each, serialised: 25 of the
// first launch's 42 us).  Here a workgroup owns a contiguous chunk of the candidate list and makes RA_PASSES passes over it
// (states are read past the L1 - volatile - so decisions of other workgroups are seen as they land; a stale read only
// defers a decision), compacts its picks with ONE returning atomic per workgroup, hands what is still open (a few dozen
// candidates chip-wide) to a list, and the workgroup that finishes last resolves that list alone.
// RA_LMAX: leftovers the last workgroup resolves inside LDS.  A few dozen in the steady state - but on the FIRST launch of the chain
// in a process (workgroups start far apart, the early ones' passes run out) there are hundreds, and with a 256-entry table they
// fell through to the global loop: 572 - 767 us once against 34 us with 2,048 entries (profiles/r04_experiments.txt item 15).
constexpr int RA_T = 256, RA_PASSES = 16, RA_LBITS = 11, RA_LMAX = 1 << RA_LBITS, RA_REG = 11;   // RA_REG neighbour ids of a candidate live in registers

// one pass over a candidate whose list (<= RA_REG entries) sits in registers: ONE round trip (the neighbours' states)
__device__ __forceinline__ int decide_reg(const int (&ids)[RA_REG], int cnt, const volatile unsigned char* state) {
    unsigned char st[RA_REG];
#pragma unroll
    for (int u = 0; u < RA_REG; ++u) st[u] = u < cnt ? state[ids[u]] : (unsigned char)2;
    bool all_decided = true, pick_near = false;
#pragma unroll
    for (int u = 0; u < RA_REG; ++u) { pick_near |= st[u] == 1; all_decided &= st[u] != 0; }
    return pick_near ? 2 : (all_decided ? 1 : 0);
}

__global__ __launch_bounds__(RA_T) void rounds_all_kernel(GreedyHeader* hdr, const unsigned long long* G, const int* map,
                                                         const unsigned* bits, const BallRun* runs, long n_vox,
                                                         unsigned cap, const int* nbr, volatile unsigned char* state,
                                                         unsigned* left, unsigned* act_b, unsigned long long* kept,
                                                         unsigned kept_cap, SegIndex sx) {
    __shared__ unsigned s_wcnt[2][RA_T / 64], s_base[2], s_last, s_next, s_kept;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned n = min(hdr->n, cap);
    const int nr = (int)hdr->n_runs;
    const unsigned chunk = (n + gridDim.x - 1) / gridDim.x;
    const unsigned lo = min(blockIdx.x * chunk, n), hi = min(lo + chunk, n);
    volatile unsigned* vleft = left;
    auto load_list = [&](unsigned i, int (&ids)[RA_REG], int& cnt) {
        const int4* row4 = reinterpret_cast<const int4*>(nbr + (size_t)i * CAPN);
        const int4 a = row4[0];
        cnt = a.x;
        ids[0] = a.y; ids[1] = a.z; ids[2] = a.w;
        if (cnt > 3 && cnt <= RA_REG) {
            const int4 b = row4[1], c = row4[2];
            ids[3] = b.x; ids[4] = b.y; ids[5] = b.z; ids[6] = b.w; ids[7] = c.x; ids[8] = c.y; ids[9] = c.z; ids[10] = c.w;
        }
    };
    for (unsigned i0 = lo; i0 < hi; i0 += RA_T) {           // (a chunk is 162 candidates at 83 k: one trip)
        const unsigned i = i0 + tid;
        int ids[RA_REG], cnt = 0, my = 2;
        if (i < hi) { load_list(i, ids, cnt); my = 0; }
        // every thread polls for itself: a pass is ONE memory round trip (the neighbours' states, read past the caches),
        // its own state goes out as a store nobody waits for - with a workgroup barrier between the passes every pass also
        // paid the store's way to memory (59 us for 12 passes)
        for (int pass = 0; pass < RA_PASSES && my == 0; ++pass) {
            const int4 r0 = make_int4(cnt, ids[0], ids[1], ids[2]);
            my = cnt <= RA_REG ? decide_reg(ids, cnt, state) : decide(i, r0, G, map, bits, runs, nr, n_vox, nbr, state, sx);
            if (my != 0) state[i] = (unsigned char)my;
        }
        // picks -> kept, still open -> left: ballot prefix in the wave, wave totals through LDS, one atomic per list
        const unsigned long long mk = __ballot(my == 1), ml = __ballot(my == 0);
        if (lane == 0) { s_wcnt[0][wv] = (unsigned)__popcll(mk); s_wcnt[1][wv] = (unsigned)__popcll(ml); }
        __syncthreads();
        if (tid < 2) {
            unsigned tot = 0;
            for (int w2 = 0; w2 < RA_T / 64; ++w2) tot += s_wcnt[tid][w2];
            s_base[tid] = tot ? atomicAdd(tid == 0 ? &hdr->n_kept : &hdr->n_left, tot) : 0u;
        }
        __syncthreads();
        unsigned offk = s_base[0], offl = s_base[1];
        for (int w2 = 0; w2 < wv; ++w2) { offk += s_wcnt[0][w2]; offl += s_wcnt[1][w2]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (my == 1) { const unsigned slot = offk + (unsigned)__popcll(mk & below); if (slot < kept_cap) kept[slot] = G[i]; }
        if (my == 0) vleft[offl + (unsigned)__popcll(ml & below)] = i;          // (left holds `cap` entries)
        __syncthreads();
    }
    // The last workgroup to get here resolves the leftovers (its own included).  What it reads from the others - their
    // states and list entries - was stored past the caches (volatile = sc0 sc1 accesses) and is loaded the same way: no
    // device-scope fence (the L2 write-back / invalidate pair costs tens of microseconds on this part -
    // MI355X_MICROARCH.md, inter-workgroup visibility, the sc1 form).  Every storing wave drains its stores explicitly
    // (on gfx9 a workgroup barrier emits no vmcnt(0) by itself; the volatile stores' own waits are the compiler's choice,
    // not a contract) before the barrier in front of the owner's ticket, an agent-scope atomic.  `kept` is a plain store:
    // no workgroup reads it inside this kernel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(&hdr->ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    unsigned n_act = *reinterpret_cast<volatile unsigned*>(&hdr->n_left);
    if (tid == 0) { s_next = 0; s_kept = *reinterpret_cast<volatile unsigned*>(&hdr->n_kept); hdr->trace[14] = n_act; }
    __syncthreads();
    // Up to RA_LMAX leftovers with register-sized lists are resolved INSIDE LDS: an open candidate only waits for open
    // candidates, and every open candidate is in this list - so after ONE look at the global states (decided neighbours
    // drop out or suppress at once) the rest of the fixed point needs no memory round trips, only LDS passes.
    __shared__ unsigned s_id[RA_LMAX];
    __shared__ short s_nb[RA_LMAX][RA_REG];
    __shared__ unsigned char s_cnt[RA_LMAX], s_st[RA_LMAX];
    __shared__ unsigned s_hash[2 * RA_LMAX], s_big;
    bool local_ok = n_act <= RA_LMAX;
    if (local_ok) {
        if (tid == 0) s_big = 0;
        for (unsigned q = tid; q < 2 * RA_LMAX; q += RA_T) s_hash[q] = 0xffffffffu;
        __syncthreads();
        for (unsigned t = tid; t < n_act; t += RA_T) {
            const unsigned i = vleft[t];
            s_id[t] = i; s_st[t] = 0;
            unsigned h = (i * 2654435761u) >> (31 - RA_LBITS);         // table of 2 RA_LMAX slots
            while (atomicCAS(&s_hash[h], 0xffffffffu, t) != 0xffffffffu) h = (h + 1) & (2 * RA_LMAX - 1);
        }
        __syncthreads();
        for (unsigned t = tid; t < n_act; t += RA_T) {
            const unsigned i = s_id[t];
            int ids[RA_REG], cnt = 0;
            load_list(i, ids, cnt);
            if (cnt > RA_REG) { s_big = 1; continue; }
            int m = 0, st0 = 0;
            for (int u = 0; u < cnt; ++u) {
                const unsigned char sg = state[ids[u]];
                if (sg == 1) st0 = 2;
                if (sg == 0) {                              // open: one of the leftovers - its local slot
                    unsigned h = ((unsigned)ids[u] * 2654435761u) >> (31 - RA_LBITS);
                    int slot = -1;
                    for (int probe = 0; probe < 2 * RA_LMAX; ++probe) {
                        const unsigned e = s_hash[h];
                        if (e == 0xffffffffu) break;
                        if (s_id[e] == (unsigned)ids[u]) { slot = (int)e; break; }
                        h = (h + 1) & (2 * RA_LMAX - 1);
                    }
                    if (slot < 0) { s_big = 1; }            // (cannot happen: every open candidate is a leftover)
                    else s_nb[t][m++] = (short)slot;
                }
            }
            s_cnt[t] = (unsigned char)m;
            if (st0 == 2) s_st[t] = 2;
        }
        __syncthreads();
        local_ok = s_big == 0;
    }
    if (local_ok) {
        for (int pass = 0; pass < 4 * RA_LMAX; ++pass) {
            int progress = 0;
            for (unsigned t = tid; t < n_act; t += RA_T) {
                if (s_st[t] != 0) continue;
                bool pick_near = false, all_dec = true;
                for (int u = 0; u < (int)s_cnt[t]; ++u) {
                    const unsigned char sg = s_st[s_nb[t][u]];
                    pick_near |= sg == 1; all_dec &= sg != 0;
                }
                const int d = pick_near ? 2 : (all_dec ? 1 : 0);
                if (d) { s_st[t] = (unsigned char)d; progress = 1; }
            }
            if (!__syncthreads_or(progress)) break;
        }
        for (unsigned t = tid; t < n_act; t += RA_T) {
            const unsigned i = s_id[t];
            state[i] = s_st[t];
            if (s_st[t] == 1) { const unsigned slot = atomicAdd(&s_kept, 1u); if (slot < kept_cap) kept[slot] = G[i]; }
        }
        __syncthreads();
        n_act = 0;
    }
    const volatile unsigned* cur = left;
    unsigned* nxt = act_b;
    while (n_act > 0) {
        // (more leftovers than the LDS tables hold, or an overflowed list among them: passes through memory)
        for (unsigned t0 = 0; t0 < n_act; t0 += RA_T) {
            const unsigned t = t0 + tid;
            int ids[RA_REG], cnt = 0, my = 2;
            unsigned i = 0;
            if (t < n_act) { i = cur[t]; load_list(i, ids, cnt); my = 0; }
            for (int pass = 0; pass < 64; ++pass) {
                int before = my;
                if (my == 0) {
                    const int4 r0 = make_int4(cnt, ids[0], ids[1], ids[2]);
                    my = cnt <= RA_REG ? decide_reg(ids, cnt, state) : decide(i, r0, G, map, bits, runs, nr, n_vox, nbr, state, sx);
                    if (my != 0) state[i] = (unsigned char)my;
                }
                // go on while somebody of this trip made progress (an open candidate may wait for one of another trip)
                if (!__syncthreads_or((before == 0 && my != 0) ? 1 : 0)) break;
            }
            if (t < n_act) {
                if (my == 0) nxt[atomicAdd(&s_next, 1u)] = i;
                else if (my == 1) { const unsigned slot = atomicAdd(&s_kept, 1u); if (slot < kept_cap) kept[slot] = G[i]; }
            }
        }
        __threadfence_block();
        __syncthreads();
        n_act = s_next;
        __syncthreads();
        if (tid == 0) s_next = 0;
        const volatile unsigned* tmp = cur; cur = nxt; nxt = const_cast<unsigned*>(tmp);
        __syncthreads();
    }
    if (tid == 0) {
        unsigned k = s_kept;
        if (k > kept_cap) { atomicOr(&hdr->overflow, 2u); k = kept_cap; }
        hdr->n_kept = k;
    }
}

// ---- picks in descending priority: rank by counting ----------------------------------------------------------------
// Keys are unique (score bits, flat index), so a pick's output row is the number of greater keys.  Round 3 sorted
// 2048-key tiles in LDS and merged by binary searches (two launches, 56 us for 16 k picks).  Here a WAVE owns ER_PPW picks
// (their keys in registers) and its lanes scan ALL keys, 64 at a time out of an LDS chunk the workgroup staged once for
// its 16 waves: one 8-byte LDS read feeds ER_PPW compares, the lanes' counts are summed by shuffles at the end.  n^2
// compares spread over the chip, no sort, one launch.
constexpr int ER_T = 1024, ER_CHUNK = 4096, ER_PPW = 4, ER_PPG = ER_PPW * (ER_T / 64);   // 64 picks per workgroup
constexpr int ER_KPT = 16;                           // keys a thread keeps in registers (16 x 1,024 = the usual number of picks)
constexpr int ER_FINE = 4096, ER_MCAP = 2048;        // score cells; the largest cell the interval form accepts (a workgroup's picks stay in LDS)
static_assert(2 * ER_MCAP <= ER_CHUNK && ER_FINE % ER_T == 0, "members alias the key chunk: n / G + the largest cell");

__global__ __launch_bounds__(ER_T) void emit_rank_kernel(GreedyHeader* hdr, const unsigned long long* kept, unsigned cap,
                                                        int H, int W, float* scores, int32_t* coords, int32_t* n_out,
                                                        int max_out, int ER_NO_INTERVALS) {
    __shared__ unsigned long long keys[ER_CHUNK];
    const unsigned n = min(hdr->n_kept, cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long hw = (long)H * W;
    // Round 5: rank by INTERVALS first.  The score range of the picks [lo, hi] is cut into ER_FINE equal cells; every workgroup counts
    // all cells (the same counts everywhere, so all take the same decisions), and runs of consecutive cells are dealt to the
    // workgroups by the running count - workgroup b owns the cells whose first pick has sorted position in [b n / G, (b + 1) n / G) -
    // so that everybody ranks ~n / G picks whatever the score distribution (DoG picks crowd at the cut-off: equal-width intervals per
    // workgroup left one of them 2,000 picks, 21 us).  A pick's row = the picks in higher cells than its workgroup's + the greater
    // picks among the workgroup's own (kept in LDS): ~3 n + (n / G)^2 operations per workgroup instead of 64 n compares.  Keys are
    // unique (score bits, flat index) and the cell of a key is monotone in the key: the rows are the sorted order, exactly.  A cell
    // with more than ER_MCAP picks (a plateau of equal scores) keeps the n^2 form below.
    if (!ER_NO_INTERVALS && n > 0) {
        __shared__ unsigned s_cnt[ER_FINE], s_wsum[ER_T / 64], s_lo[ER_T / 64], s_hi[ER_T / 64], s_nm, s_f0, s_f1, s_above, s_big;
        unsigned long long* members = keys;              // ER_MCAP <= ER_CHUNK
        const unsigned G = gridDim.x;
        for (int i = tid; i < ER_FINE; i += ER_T) s_cnt[i] = 0u;
        if (tid == 0) { s_nm = 0u; s_f0 = ER_FINE; s_f1 = 0u; s_above = 0u; s_big = 0u; }
        // the first ER_KPT keys of a thread stay in registers over the three passes (n <= 16 k: all of them); the rest comes from L2 again
        unsigned long long kr[ER_KPT];
#pragma unroll
        for (int e = 0; e < ER_KPT; ++e) { const unsigned q = tid + e * ER_T; kr[e] = q < n ? kept[q] : 0ull; }
        auto for_keys = [&](auto&& f) {
#pragma unroll
            for (int e = 0; e < ER_KPT; ++e) if ((unsigned)(tid + e * ER_T) < n) f(kr[e]);
            for (unsigned q = tid + ER_KPT * ER_T; q < n; q += ER_T) f(kept[q]);
        };
        unsigned lo = 0xffffffffu, hi = 0u;
        for_keys([&](unsigned long long k) { const unsigned h = (unsigned)(k >> 32); lo = min(lo, h); hi = max(hi, h); });
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64)); hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64)); }
        if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ER_T / 64; ++i) { lo = min(lo, s_lo[i]); hi = max(hi, s_hi[i]); }
        const unsigned long long span = (unsigned long long)(hi - lo) + 1ull;
        // cell = floor((h - lo) * cscale / 2^32), cscale = floor(ER_FINE * 2^32 / span): monotone in h, < ER_FINE; ONE 64-bit division per
        // thread instead of one per key (a 64-bit division is ~150 instructions: 32 of them per thread were 40 us)
        const unsigned long long cscale = ((unsigned long long)ER_FINE << 32) / span;
        auto cell_of = [&](unsigned long long k) {
            return (unsigned)((((unsigned long long)((unsigned)(k >> 32) - lo)) * cscale) >> 32);
        };
        for_keys([&](unsigned long long k) { atomicAdd(&s_cnt[cell_of(k)], 1u); });
        __syncthreads();
        // exclusive running count of the cells in ASCENDING cell order (thread t: cells 4 t .. 4 t + 3), the owner of a cell from it
        constexpr int CPT = ER_FINE / ER_T;
        unsigned c[CPT], tsum = 0u;
#pragma unroll
        for (int e = 0; e < CPT; ++e) { c[e] = s_cnt[CPT * tid + e]; tsum += c[e]; }
        unsigned incl = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        unsigned base = incl - tsum;
        for (int i = 0; i < wv; ++i) base += s_wsum[i];
        unsigned big = 0u;
        const unsigned long long gscale = ((unsigned long long)G << 32) / n;
#pragma unroll
        for (int e = 0; e < CPT; ++e) {
            if (c[e]) {
                const unsigned owner = (unsigned)(((unsigned long long)base * gscale) >> 32);       // base < n: owner < G, monotone
                if (owner == blockIdx.x) {                 // (few threads: a workgroup owns a handful of cells)
                    atomicMin(&s_f0, (unsigned)(CPT * tid + e)); atomicMax(&s_f1, (unsigned)(CPT * tid + e));
                    atomicMax(&s_above, base + c[e]);      // the running count behind the workgroup's last cell
                }
                big = max(big, c[e]);
            }
            base += c[e];
        }
        if (big > ER_MCAP) s_big = 1u;
        __syncthreads();
        // (a workgroup's picks: at most n / G + the largest cell; more than ER_MCAP only with a cell of that size)
        if (!s_big && n / G + 4 + ER_MCAP <= ER_CHUNK) {      // (uniform over the grid)
            const unsigned f0 = s_f0, f1 = s_f1, above = n - s_above;
            if (f0 <= f1) {
                for_keys([&](unsigned long long k) {
                    const unsigned f = cell_of(k);
                    if (f >= f0 && f <= f1) { const unsigned slot = atomicAdd(&s_nm, 1u); if (slot < ER_CHUNK) members[slot] = k; }
                });
            }
            __syncthreads();
            const unsigned nm = min(s_nm, (unsigned)ER_CHUNK);      // (<= n / G + 1 + the largest cell <= ER_CHUNK)
            {
                for (unsigned q = tid; q < nm; q += ER_T) {
                    const unsigned long long k = members[q];
                    unsigned r = above;
                    for (unsigned j = 0; j < nm; ++j) r += members[j] > k ? 1u : 0u;      // (LDS broadcast reads)
                    if (r < (unsigned)max_out) {
                        const long idx = (long)(k & 0xffffffffull);
                        scores[r] = unorder_bits((unsigned)(k >> 32));
                        const long z = idx / hw, t2 = idx - z * hw;
                        coords[3 * r + 0] = (int)(t2 % W);
                        coords[3 * r + 1] = (int)(t2 / W);
                        coords[3 * r + 2] = (int)z;
                    }
                }
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    unsigned ov = hdr->overflow;
                    if (hdr->n_kept > cap || hdr->n_kept > (unsigned)max_out) ov |= 2u;
                    int v = (int)min(n, (unsigned)max_out);
                    if (ov) v = -(int)ov;   // -1: candidate overflow, -2: pick overflow
                    *n_out = v;
                }
                return;
            }
        }
        __syncthreads();                                  // (the n^2 form reuses `keys`)
    }
    for (unsigned g0 = blockIdx.x * ER_PPG; g0 < n; g0 += gridDim.x * ER_PPG) {      // (workgroup-uniform trip count)
        unsigned long long key[ER_PPW];
        unsigned rank[ER_PPW];
#pragma unroll
        for (int u = 0; u < ER_PPW; ++u) {
            const unsigned i = g0 + wv * ER_PPW + u;
            key[u] = i < n ? kept[i] : ~0ull;
            rank[u] = 0;
        }
        for (unsigned c0 = 0; c0 < n; c0 += ER_CHUNK) {
            const unsigned m = min((unsigned)ER_CHUNK, n - c0);
            __syncthreads();
#pragma unroll
            for (int q = tid; q < ER_CHUNK; q += ER_T) keys[q] = (unsigned)q < m ? kept[c0 + q] : 0ull;   // zero keys count for nobody
            __syncthreads();
            // eight keys per lane and step, all eight LDS reads in flight before the first compare (left to the compiler the
            // loop waited for every read by itself: 30 us); the chunk's tail is zero keys
            for (unsigned q0 = 0; q0 < m; q0 += 512) {
                unsigned long long kj[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) kj[e] = keys[q0 + 64 * e + lane];
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int u = 0; u < ER_PPW; ++u) rank[u] += kj[e] > key[u] ? 1u : 0u;
                    // (measured and rejected: the borrow of key - kj as v_sub_co / v_subb_co / v_addc_co in inline assembly -
                    // every chain goes through vcc, nothing overlaps: 38 us against 28)
            }
        }
#pragma unroll
        for (int u = 0; u < ER_PPW; ++u) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rank[u] += __shfl_xor(rank[u], o, 64);
            const unsigned i = g0 + wv * ER_PPW + u;
            if (lane == 0 && i < n && rank[u] < (unsigned)max_out) {
                const long idx = (long)(key[u] & 0xffffffffull);
                scores[rank[u]] = unorder_bits((unsigned)(key[u] >> 32));
                const long z = idx / hw, t2 = idx - z * hw;
                coords[3 * rank[u] + 0] = (int)(t2 % W);
                coords[3 * rank[u] + 1] = (int)(t2 / W);
                coords[3 * rank[u] + 2] = (int)z;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned ov = hdr->overflow;
        if (hdr->n_kept > cap || hdr->n_kept > (unsigned)max_out) ov |= 2u;
        int v = (int)min(n, (unsigned)max_out);
        if (ov) v = -(int)ov;   // -1: candidate overflow, -2: pick overflow
        *n_out = v;
    }
}

struct GreedyWs {
    GreedyHeader* hdr;
    unsigned long long* G;
    unsigned long long* kept;
    int* nbr;
    unsigned char* state;
    unsigned* act_a;
    unsigned* act_b;
    int* map;          // dense, n_vox ints; valid only where `bits` is set (never cleared)
    unsigned* vmap;    // dense order_bits(value) of the candidates (valid where `bits` is set), or null when
    const float* vol;  // ... the dense value volume itself is at hand (mi_greedy_nms3d)
    unsigned* bits;    // candidate bitmap of the volume, (n_vox + 31) / 32 words (+2 words of slack)
    BallRun* runs;
    uint2* ranges;     // list range of every workgroup of cand_filter_seg_kernel
    int n_ranges;      // 0: the list is walked in slot order
    SegIndex sx;       // seg_range != null: the list is indexed by segment (fused picker) - no bitmap / id map / value map
    unsigned cap, kept_cap;
};

// candidate capacity.  DoG picker: every 4th voxel (xy-NMS survivors are at most 1 per 2x2 patch
// without plateaus).  Dense API: all voxels for small volumes, every 4th above 4 Mi voxels.
size_t greedy_default_cap(size_t n_vox, bool dense_api) {
    if (dense_api && n_vox <= (1u << 22)) return n_vox;
    return n_vox / 4 + 1024;
}

size_t greedy_ws_layout(size_t n_vox, size_t cap, GreedyWs* w, char* base, bool with_map) {
    size_t kept_cap = 2048;
    while (kept_cap < cap) kept_cap <<= 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += mi_align_up(bytes, 256); return base ? base + o : nullptr; };
    char* p;
    p = take(sizeof(GreedyHeader)); if (w) w->hdr = (GreedyHeader*)p;
    p = take(sizeof(unsigned long long) * cap); if (w) w->G = (unsigned long long*)p;
    p = take(sizeof(unsigned long long) * kept_cap); if (w) w->kept = (unsigned long long*)p;
    p = take(sizeof(int) * cap * CAPN); if (w) w->nbr = (int*)p;
    p = take(cap); if (w) w->state = (unsigned char*)p;
    p = take(sizeof(unsigned) * cap); if (w) w->act_a = (unsigned*)p;
    p = take(sizeof(unsigned) * cap); if (w) w->act_b = (unsigned*)p;
    p = take(sizeof(BallRun) * MAX_RUNS); if (w) w->runs = (BallRun*)p;
    p = take(sizeof(uint2) * NB_MAX_RANGES); if (w) { w->ranges = (uint2*)p; w->n_ranges = 0; w->sx = SegIndex{}; }
    p = take(sizeof(unsigned) * ((n_vox + 31) / 32 + 2)); if (w) w->bits = (unsigned*)p;
    if (with_map) { p = take(sizeof(int) * n_vox); if (w) w->map = (int*)p; }
    if (w) { w->cap = (unsigned)cap; w->kept_cap = (unsigned)kept_cap; }
    return off;
}

// everything after the candidate list G (and map) is filled: neighbours, rounds, ranked output - three launches
int greedy_tail(const GreedyWs& w, int D, int H, int W, float d, float scale, float* scores,
                int32_t* coords, int32_t* n_out, int max_out, hipStream_t s) {
    const long n_vox = (long)D * H * W;
    double r = (double)scale * (double)d / 2.0;
    int width = (int)ceil(r);
    if (width > 16 || width < 0) return MI_E_UNSUPPORTED;
    SegIndex sx = w.sx;
    if (sx.seg_range) {
        sx.width = width; sx.r2 = r * r;
        hipLaunchKernelGGL(neighbors_seg_kernel, dim3(2048), dim3(256), 0, s, w.hdr, w.G, sx, w.cap, w.nbr, w.state);
    } else {
        hipLaunchKernelGGL(neighbors_kernel, dim3(2048), dim3(256), 0, s, w.hdr, w.G, w.map, w.vmap, w.vol, w.bits, w.runs,
                           n_vox, w.cap, w.nbr, w.state, r, width, (long)H * W, (long)W, w.ranges, w.n_ranges);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(rounds_all_kernel, dim3(512), dim3(RA_T), 0, s, w.hdr, w.G, w.map, w.bits, w.runs, n_vox, w.cap,
                       w.nbr, w.state, w.act_a, w.act_b, w.kept, w.kept_cap, sx);
    MI_RETURN_IF_LAUNCH_FAILED();
    const int er_n2 = getenv("MI_EMIT_RANK_N2") ? 1 : 0;                 // A/B: the round-4 n^2 form only
    hipLaunchKernelGGL(emit_rank_kernel, dim3(256), dim3(ER_T), 0, s, w.hdr, w.kept, w.kept_cap, H, W, scores, coords,
                       n_out, max_out, er_n2);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

extern "C" size_t mi_greedy_nms3d_workspace_bytes(int D, int H, int W) {
    size_t n = (size_t)D * H * W;
    return greedy_ws_layout(n, greedy_default_cap(n, true), nullptr, nullptr, true);
}

extern "C" int mi_greedy_nms3d(const float* vol, int D, int H, int W, float d, float scale,
                               float threshold, float* scores, int32_t* coords, int32_t* n_out,
                               int max_out, void* workspace, size_t workspace_bytes,
                               mi_stream_t stream) {
    if (!vol || !scores || !coords || !n_out || !workspace || max_out <= 0) return MI_E_ARG;
    if (D <= 0 || H <= 0 || W <= 0) return MI_E_ARG;
    size_t n_vox = (size_t)D * H * W;
    if (n_vox >= (1ull << 31)) return MI_E_UNSUPPORTED;
    if (workspace_bytes < mi_greedy_nms3d_workspace_bytes(D, H, W)) return MI_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    GreedyWs w;
    greedy_ws_layout(n_vox, greedy_default_cap(n_vox, true), &w, (char*)workspace, true);
    w.vmap = nullptr; w.vol = vol;                          // the dense values are the volume itself
    MI_HIP(hipMemsetAsync(w.hdr, 0, sizeof(GreedyHeader), s));
    MI_HIP(hipMemsetAsync(w.bits, 0, sizeof(unsigned) * ((n_vox + 31) / 32 + 2), s));
    hipLaunchKernelGGL(set_cutoff_kernel, dim3(1), dim3(1), 0, s, w.hdr, threshold);
    MI_RETURN_IF_LAUNCH_FAILED();
    int blocks = (int)std::min<size_t>((n_vox + 255) / 256, 4096);
    hipLaunchKernelGGL(dense_filter_kernel, dim3(blocks), dim3(256), 0, s, vol, n_vox, w.hdr, w.G,
                       w.map, w.bits, w.cap);
    MI_RETURN_IF_LAUNCH_FAILED();
    return greedy_tail(w, D, H, W, d, scale, scores, coords, n_out, max_out, s);
}

// ---------------------------------------------------------------------------------------------
// DoG particle picker
// ---------------------------------------------------------------------------------------------
namespace {
struct DogWs {
    float* g[2];
    float* tmp;
    float* heat;
    uint2* cands;
    double* stats;
    unsigned* seg_count;
    uint2* seg_range;
    unsigned* sub;
    size_t sub_room;
    unsigned cand_cap;
    size_t n_stats, cand_room, seg_room;
    DogxGrid xg;
    GreedyWs gw;
};

size_t dog_ws_layout(int D, int H, int W, DogWs* w, char* base) {
    size_t n_vox = (size_t)D * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += mi_align_up(bytes, 256); return base ? base + o : nullptr; };
    char* p;
    p = take(sizeof(float) * n_vox); if (w) w->g[0] = (float*)p;
    p = take(sizeof(float) * n_vox); if (w) w->g[1] = (float*)p;
    p = take(sizeof(float) * n_vox); if (w) w->tmp = (float*)p;
    p = take(sizeof(float) * n_vox); if (w) w->heat = (float*)p;
    // candidates: the linear list of the generic march, or one segment per wave of the fused kernels (infer_dogx.hip: x pass;
    // infer_dogf.hip: y march - its grid depends on the z border, unknown here: bounds over every border)
    const DogxGrid xg = mi_dogx_grid(D, H, W);
    size_t cand_cap = n_vox / 4 + 1024;
    const int bxy_l = (H > 512 && W > 512) ? 60 : 30;
    size_t f_seg = 0, f_ent = 0;
    if (2 * bxy_l < H && 2 * bxy_l < W) {
        const DogfGrid fg0 = mi_dogf_grid(D, H, W, 0, bxy_l);
        f_seg = std::max<size_t>(4 * 12288, (size_t)D * fg0.n_strips) + 64;      // (mi_dogf_grid: chunks double while waves < 6144; x4 for the MI_DOGF_NYC knob)
        f_ent = (size_t)D * fg0.n_strips * (size_t)(H - 2 * bxy_l) * 16 + 96 * f_seg;
    }
    const size_t cand_room = std::max(std::max(cand_cap, (size_t)xg.n_seg * xg.seg_cap), f_ent);
    p = take(sizeof(uint2) * cand_room);
    if (w) { w->cands = (uint2*)p; w->cand_cap = (unsigned)cand_cap; w->xg = xg; w->cand_room = cand_room; }
    int zc;
    dim3 grid = mi_march_grid(D, H, W, &zc);
    size_t n_stats = (size_t)grid.x * grid.y * grid.z;
    const size_t seg_room = std::max((size_t)xg.n_seg, f_seg);
    p = take(sizeof(double) * 3 * std::max(n_stats, seg_room)); if (w) { w->stats = (double*)p; w->n_stats = n_stats; }
    p = take(sizeof(unsigned) * seg_room); if (w) { w->seg_count = (unsigned*)p; w->seg_room = seg_room; }
    p = take(sizeof(uint2) * seg_room); if (w) w->seg_range = (uint2*)p;
    // sub-index of the fused picker's segments: (rows of a chunk / 8 + 1) slots per segment
    const size_t sub_room = 3 * seg_room + (2 * bxy_l < H && 2 * bxy_l < W
                                            ? (size_t)D * mi_dogf_grid(D, H, W, 0, bxy_l).n_strips * (size_t)((H - 2 * bxy_l) / 8 + 2) : 0);
    p = take(sizeof(unsigned) * sub_room); if (w) { w->sub = (unsigned*)p; w->sub_room = sub_room; }
    // the dense candidate map reuses a Gaussian buffer (free once the last DoG level is consumed)
    off += greedy_ws_layout(n_vox, greedy_default_cap(n_vox, false), w ? &w->gw : nullptr, base ? base + off : nullptr, false);
    return off;
}
}  // namespace

extern "C" size_t mi_dog_pick_workspace_bytes(int D, int H, int W, int n_sigmas) {
    (void)n_sigmas;
    return dog_ws_layout(D, H, W, nullptr, nullptr);
}

extern "C" int mi_dog_pick(const float* rec, int D, int H, int W, const float* sigmas_host,
                           int n_sigmas, int k, int border_z, int nms_d, float* heat_out,
                           float* scores, int32_t* coords, int32_t* n_out, int max_out,
                           float* cutoff_out, void* workspace, size_t workspace_bytes,
                           mi_stream_t stream) {
    if (!rec || !sigmas_host || n_sigmas < 2 || !scores || !coords || !n_out || !workspace) return MI_E_ARG;
    if (D <= 0 || H <= 0 || W <= 0 || max_out <= 0 || border_z < 0) return MI_E_ARG;
    size_t n_vox = (size_t)D * H * W;
    if (n_vox >= (1ull << 31)) return MI_E_UNSUPPORTED;
    if (workspace_bytes < mi_dog_pick_workspace_bytes(D, H, W, n_sigmas)) return MI_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    DogWs w;
    dog_ws_layout(D, H, W, &w, (char*)workspace);
    GreedyWs& gw = w.gw;
    const unsigned bits_words = (unsigned)((n_vox + 31) / 32 + 2);
    // Round 4: z + x passes | y pass + DoG + NMS (infer_dogf.hip) - two launches, no y-pass intermediates; the first launch
    // also zeroes the header and the candidate bitmap, the candidate filter computes the cutoff in its prologue.
    {
        const int bxy_f = (H > 512 && W > 512) ? 60 : 30;
        // (round 5 built the same two launches on the matrix cores - banded-Toeplitz bf16x3 products, picks identical - and measured
        // them slower, 290 + 248 us against 203 + 152: profiles/r05_experiments.txt; the kernels left the library in round 6)
        if (n_sigmas == 2 && mi_dogf_usable(rec, w.g[0], w.g[1], heat_out, D, H, W, sigmas_host[0], sigmas_host[1], k, border_z, bxy_f)) {
            const DogfGrid fg = mi_dogf_grid(D, H, W, border_z, bxy_f);
            if (fg.n_seg > 0 && (size_t)fg.n_seg * fg.seg_cap <= w.cand_room && fg.n_seg <= w.seg_room) {
                DogfParams q = {};
                q.rec = rec; q.g1 = w.g[0]; q.g2 = w.g[1]; q.nms_out = heat_out;
                q.D = D; q.H = H; q.W = W; q.bz = border_z; q.by = bxy_f; q.bx = bxy_f;
                q.cands = w.cands; q.seg_count = w.seg_count; q.overflow = &gw.hdr->overflow; q.stats = w.stats;
                q.clr[0] = reinterpret_cast<unsigned*>(gw.hdr); q.clr_n[0] = (unsigned)(sizeof(GreedyHeader) / 4);
                const int nb = (fg.ychunk + 7) / 8;
                const bool use_index = !getenv("MI_DOG_NO_INDEX") &&
                                       (fg.n_seg + CI_MAXSEG - 1) / CI_MAXSEG <= (unsigned)NB_MAX_RANGES && H < 65536 && W < 65536 &&
                                       (size_t)fg.n_seg * (nb + 1) <= w.sub_room;
                if (!use_index) { q.clr[1] = gw.bits; q.clr_n[1] = bits_words; }     // (the bitmap search: A/B and fallback)
                if (heat_out) MI_HIP(hipMemsetAsync(heat_out, 0, sizeof(float) * n_vox, s));      // the zeroed border
                int rcf = mi_launch_dogf(q, fg, sigmas_host[0], sigmas_host[1], s);
                if (rcf) return rcf;
                gw.map = reinterpret_cast<int*>(w.tmp);               // (g[0] / g[1] are read by the y march only: free as
                gw.vmap = reinterpret_cast<unsigned*>(w.heat); gw.vol = nullptr;    // well, but these two are never touched)
                if (use_index) {
                    // survivors above the cutoff, segment by segment and in row order: the list IS a grid index
                    const unsigned fb = std::max<unsigned>(std::min<unsigned>((fg.n_seg + 7) / 8, (unsigned)NB_MAX_RANGES),
                                                           (fg.n_seg + CI_MAXSEG - 1) / CI_MAXSEG);
                    unsigned* coord = reinterpret_cast<unsigned*>(w.tmp);       // (free volumes of the workspace:
                    unsigned* coordz = reinterpret_cast<unsigned*>(w.heat);     // `cap` entries each fit four times)
                    hipLaunchKernelGGL(cand_index_kernel, dim3(fb), dim3(256), 0, s, w.cands, w.seg_count, fg.n_seg, fg.seg_cap,
                                       gw.hdr, gw.G, coord, coordz, w.seg_range, gw.cap, (const double*)w.stats, (int)fg.n_wg,
                                       cutoff_out, gw.ranges, H, W, w.sub, nb, bxy_f, fg.ychunk, fg.n_ychunks, fg.n_strips);
                    MI_RETURN_IF_LAUNCH_FAILED();
                    gw.n_ranges = (int)fb;
                    SegIndex sx = {};
                    sx.seg_range = w.seg_range; sx.coord = coord; sx.coordz = coordz; sx.sub = w.sub; sx.nb = nb;
                    sx.D = D; sx.H = H; sx.W = W; sx.bz = border_z; sx.by = bxy_f; sx.bx = bxy_f;
                    sx.ychunk = fg.ychunk; sx.n_ychunks = fg.n_ychunks; sx.n_strips = fg.n_strips; sx.own = mi_dogf_own();
                    gw.sx = sx;
                } else {
                    const unsigned fb = std::min<unsigned>((fg.n_seg + 7) / 8, 1024u);
                    hipLaunchKernelGGL(cand_filter_seg_kernel<true>, dim3(fb), dim3(256), 0, s, w.cands, w.seg_count, fg.n_seg,
                                       fg.seg_cap, gw.hdr, gw.G, gw.map, gw.vmap, gw.bits, gw.cap, (const double*)w.stats,
                                       (int)fg.n_wg, cutoff_out, gw.ranges);
                    MI_RETURN_IF_LAUNCH_FAILED();
                    gw.n_ranges = (int)fb;
                }
                return greedy_tail(gw, D, H, W, (float)nms_d, 1.0f, scores, coords, n_out, max_out, s);
            }
        }
    }
    // Every other chain (volumes the fused kernels do not take: fewer planes than a ring, rows wider than 512, more than two
    // sigmas, other windows) clears the header and the candidate bitmap with two fill passes here - ONE place, whatever branch
    // runs below (round 3 skipped them for its fused chain by re-deriving that chain's own launch condition by hand).
    MI_HIP(hipMemsetAsync(gw.hdr, 0, sizeof(GreedyHeader), s));
    MI_HIP(hipMemsetAsync(gw.bits, 0, sizeof(unsigned) * bits_words, s));

    // utils/image.py:141-143: 30-voxel xy border, doubled when both H and W exceed 512
    int bxy = (H > 512 && W > 512) ? 60 : 30;
    const bool no_march = getenv("MI_GAUSS_NO_MARCH") != nullptr;
    auto gauss = [&](float sigma, float* dst) -> int {
        int rc;
        // z and y: the marching kernel (each element read once); x: the contiguous-axis kernel
        rc = no_march ? MI_E_UNSUPPORTED : mi_launch_gauss_march(rec, dst, nullptr, sigma, 0.f, nullptr, nullptr, 0.f, D, H, W, 0, s);
        if (rc == MI_E_UNSUPPORTED) rc = mi_launch_gauss_axis(rec, dst, D, H, W, 0, sigma, s);
        if (rc) return rc;
        rc = no_march ? MI_E_UNSUPPORTED : mi_launch_gauss_march(dst, w.tmp, nullptr, sigma, 0.f, nullptr, nullptr, 0.f, D, H, W, 1, s);
        if (rc == MI_E_UNSUPPORTED) rc = mi_launch_gauss_axis(dst, w.tmp, D, H, W, 1, sigma, s);
        if (rc) return rc;
        return mi_launch_gauss_axis(w.tmp, dst, D, H, W, 2, sigma, s);
    };
    // first DoG level: the z passes of both Gaussians in one launch and the y passes in another (two jobs each;
    // w.heat is free until the first NMS pass writes it).  The one-read-two-sigmas form of the z pass was measured
    // slower than two single-sigma jobs (its 82 scalar taps do not fit the SGPR file).
    auto gauss_pair = [&](float sa, float sb, float* ga, float* gb) -> int {
        int rc = no_march ? MI_E_UNSUPPORTED
                 : sa <= sb ? mi_launch_gauss_march(rec, ga, gb, sa, sb, nullptr, nullptr, 0.f, D, H, W, 0, s)
                            : mi_launch_gauss_march(rec, ga, nullptr, sa, 0.f, rec, gb, sb, D, H, W, 0, s);
        if (rc == MI_E_UNSUPPORTED) {
            if ((rc = gauss(sa, ga))) return rc;
            return gauss(sb, gb);
        }
        if (rc) return rc;
        rc = mi_launch_gauss_march(ga, w.tmp, nullptr, sa, 0.f, gb, w.heat, sb, D, H, W, 1, s);
        if (rc == MI_E_UNSUPPORTED) {
            if ((rc = mi_launch_gauss_axis(ga, w.tmp, D, H, W, 1, sa, s))) return rc;
            rc = mi_launch_gauss_axis(gb, w.heat, D, H, W, 1, sb, s);
        }
        if (rc) return rc;
        if ((rc = mi_launch_gauss_axis(w.tmp, ga, D, H, W, 2, sa, s))) return rc;
        return mi_launch_gauss_axis(w.heat, gb, D, H, W, 2, sb, s);
    };
    int rc;
    int cur = 0;
    // Two sigmas, 3x3 window, one 512-wide row segment: z and y passes, then ONE kernel for the x passes of both
    // Gaussians + DoG + border + xy-NMS + statistics + candidates (infer_dogx.hip) instead of two x passes and a march.
    if (n_sigmas == 2 && !no_march &&
        mi_dogx_usable(w.tmp, w.heat, heat_out, D, H, W, sigmas_host[0], sigmas_host[1], k) &&
        (bxy >= mi_gauss_radius(sigmas_host[1]) || W > 2 * mi_gauss_radius(sigmas_host[1]))) {
        const float sa = sigmas_host[0], sb = sigmas_host[1];
        // z: ONE read of the tomogram feeds both sigmas; y: two jobs, each with its own radius.  The DoG is zeroed
        // inside its border (z: border_z planes, x / y: bxy voxels), so only the part of each pass that the next one reads
        // is produced: planes [bz, D - bz), and on y / x the live range widened by the larger radius (bxy >= radius: the
        // fused kernel's windows of the live outputs stay inside it).  -11 % (z pass) / -19 % (y pass) of the writes.
        const int rmax = std::max(mi_gauss_radius(sa), mi_gauss_radius(sb));
        int boxz[6] = {0, D, 0, H, 0, W}, boxy[6] = {0, D, 0, H, 0, W};
        if (bxy >= rmax && 2 * border_z < D && 2 * bxy < H && 2 * bxy < W) {
            const int z0 = border_z, z1 = D - border_z;
            const int ylo = std::max(0, bxy - rmax), yhi = std::min(H, H - bxy + rmax);
            // (x stays whole: a box that starts at x = 10 and is 492 wide costs the marches more in split cache lines and
            // rows straddled by a wave than its 8 % of columns saves - measured 269 us against 231 us for the y pass)
            const int bz_[6] = {z0, z1, ylo, yhi, 0, W}, by_[6] = {z0, z1, bxy, H - bxy, 0, W};
            for (int i = 0; i < 6; ++i) { boxz[i] = bz_[i]; boxy[i] = by_[i]; }
        }
        rc = sa <= sb ? mi_launch_gauss_march(rec, w.g[0], w.g[1], sa, sb, nullptr, nullptr, 0.f, D, H, W, 0, s, boxz)
                      : mi_launch_gauss_march(rec, w.g[0], nullptr, sa, 0.f, rec, w.g[1], sb, D, H, W, 0, s, boxz);
        if (rc == MI_OK) rc = mi_launch_gauss_march(w.g[0], w.tmp, nullptr, sa, 0.f, w.g[1], w.heat, sb, D, H, W, 1, s, boxy);
        if (rc == MI_OK) {
            DogxParams q = {};
            q.y1 = w.tmp; q.y2 = w.heat; q.nms_out = heat_out;
            q.D = D; q.H = H; q.W = W; q.bz = border_z; q.by = bxy; q.bx = bxy;
            q.cands = w.cands; q.seg_count = w.seg_count; q.overflow = &gw.hdr->overflow; q.stats = w.stats;
            if ((rc = mi_launch_dogx(q, w.xg, sa, sb, s))) return rc;
            hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(256), 0, s, w.stats, (int)w.xg.n_seg, gw.hdr, cutoff_out);
            MI_RETURN_IF_LAUNCH_FAILED();
            gw.map = reinterpret_cast<int*>(w.g[0]);              // both z-pass outputs are consumed
            gw.vmap = reinterpret_cast<unsigned*>(w.g[1]); gw.vol = nullptr;
            const unsigned fb = std::min<unsigned>((w.xg.n_seg + 7) / 8, 1024u);
            hipLaunchKernelGGL(cand_filter_seg_kernel<false>, dim3(fb), dim3(256), 0, s, w.cands, w.seg_count, w.xg.n_seg,
                               w.xg.seg_cap, gw.hdr, gw.G, gw.map, gw.vmap, gw.bits, gw.cap, (const double*)nullptr, 0,
                               (float*)nullptr, gw.ranges);
            gw.n_ranges = (int)fb;
            MI_RETURN_IF_LAUNCH_FAILED();
            return greedy_tail(gw, D, H, W, (float)nms_d, 1.0f, scores, coords, n_out, max_out, s);
        }
        if (rc != MI_E_UNSUPPORTED) return rc;                    // radius not instantiated: the generic chain below
    }
    if ((rc = gauss_pair(sigmas_host[0], sigmas_host[1], w.g[0], w.g[1]))) return rc;
    float* dense = heat_out ? heat_out : ((n_sigmas > 2) ? w.heat : nullptr);
    for (int i = 1; i < n_sigmas; ++i) {
        int nxt = cur ^ 1;
        if (i > 1 && (rc = gauss(sigmas_host[i], w.g[nxt]))) return rc;      // level 1 came with the pair above
        const bool last = (i == n_sigmas - 1);
        MarchParams p = {};
        p.in = w.g[cur]; p.in2 = w.g[nxt]; p.mode = MI_LOAD_DOG;
        p.nms_out = dense; p.accumulate = (i > 1);
        p.D = D; p.H = H; p.W = W;
        p.bz = border_z; p.by = bxy; p.bx = bxy;
        if (last) {
            p.cands = w.cands; p.cand_count = &gw.hdr->cand_count; p.cand_cap = w.cand_cap;
            p.stats = w.stats;
        }
        if ((rc = mi_launch_march(p, 1, k, s))) return rc;
        cur = nxt;
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(256), 0, s, w.stats, (int)w.n_stats,
                       gw.hdr, cutoff_out);
    MI_RETURN_IF_LAUNCH_FAILED();
    // dense candidate-id map in the Gaussian buffer that is no longer needed
    gw.map = reinterpret_cast<int*>(w.g[cur ^ 1]);
    gw.vmap = reinterpret_cast<unsigned*>(w.tmp); gw.vol = nullptr;      // the x-pass scratch volume is free by now
    // (many short workgroups: the loop is one dependent load per trip)
    hipLaunchKernelGGL(cand_filter_kernel, dim3(4096), dim3(256), 0, s, w.cands, w.cand_cap, gw.hdr,
                       gw.G, gw.map, gw.vmap, gw.bits, gw.cap);
    MI_RETURN_IF_LAUNCH_FAILED();
    return greedy_tail(gw, D, H, W, (float)nms_d, 1.0f, scores, coords, n_out, max_out, s);
}
