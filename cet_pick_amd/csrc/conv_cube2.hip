// 3x3x3 / stride 1 / padding 1 convolutions on 2 x 2 x 2 volumes (layer3 and feature_3d of the MoCo-3D encoder,
// cet_pick/models/networks/moco_encoder_3d.py:55-84,172,178: 256 -> 256 channels, four per encoder pass), forward and data
// gradient, bf16x3 arithmetic.
//
// On a 2^3 volume every (input voxel i, output voxel o) pair is connected by exactly ONE tap (per axis t = i - o + 1 is
// 0, 1 or 2), so the convolution of a sample is a dense product with no padding at all:
//   Y[n][(o, co)] = sum over (i, ci) of X[n][(i, ci)] * W[tap(i, o)][ci][co]          - a (N x 8C) . (8C x 8C) GEMM
// whose A operand is the activation tensor as it lies in memory.  N is the batch (64 rows): far too few rows for the
// implicit GEMM's tile pipeline (12 - 19 us per launch + a 4 us split-K reduce for 0.5 GFLOP: launch, prologue and
// per-slice barrier latency).  Here nothing is staged: a wave owns 64 rows x 32 columns x a sixteenth of the reduction,
// loads its A fragments (8 consecutive channels of a row: two 16-byte loads) and B fragments (forward: 8 strided dwords;
// data gradient: the same 8 values are contiguous in W) straight from L2 into registers, one k-step ahead, cuts them there
// (three bf16 planes) and issues the six products.  The four waves of a workgroup split a quarter of the reduction and
// add their tiles through LDS; 8C/32 column blocks x 4 quarters = 256 workgroups write four slabs that the caller's reduce
// launch sums (with the convolution's epilogue).  No weight image, no LDS tile, one barrier.
// (The weight gradient in the same style - a workgroup per (tap, 64 x 32 tile), the batch as the reduction, every fragment 8
// dwords 32 KB apart - was built and measured: 17.1 us against 19.0 us for the implicit GEMM + its share of the batch
// reduce; it is bound by the number of 4-byte load instructions and was not kept.)  (Sixteen waves per workgroup
// reducing a whole tile in one launch - no slabs, epilogue in the kernel - were measured: 64 workgroups on 64 CUs take
// 20.7 / 24.5 us against 10.5 + 4.2 us for this form.)
#include "common.h"
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KQ = 4;                       // reduction quarters (split-K slabs); x 4 waves = 16 parts

struct Cube2Params {
    const float* a;           // X (forward) or dY (data gradient): (N, 2, 2, 2, C) = N rows of 8 C
    const float* w;           // [27][C][C] f32, kernel layout [tap][ci][co]
    float* slabs;             // KQ slabs of (N * 8, C) floats
    int N, C;
    unsigned a_bytes, w_bytes;
    // FINAL form (round 4): the last of a tile's KQ workgroups sums the slabs and applies the epilogue
    unsigned* tickets;        // one arrival counter per output tile, zero between launches
    float* out;               // (N * 8, C)
    const float* res;         // out = act(sum + res)          (may be null)
    const float* mask;        // out *= (mask > 0)             (may be null)
    int relu;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of2(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// exact three-way bf16 cut of 8 f32 (truncation, as conv_igemm.hip / conv_direct3.hip)
__device__ __forceinline__ void cut8r(const float (&v)[8], bf16x8 (&o)[3]) {
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

// tap index of the pair (voxel on the reduction side kv, voxel on the column side cv): per axis t = in - out + 1
__device__ __forceinline__ int pair_tap(int in_v, int out_v) {
    const int tz = ((in_v >> 2) & 1) - ((out_v >> 2) & 1) + 1, ty = ((in_v >> 1) & 1) - ((out_v >> 1) & 1) + 1,
              tx = (in_v & 1) - (out_v & 1) + 1;
    return (tz * 3 + ty) * 3 + tx;
}

// FINAL: no reduce launch behind the kernel.  Each workgroup stores its quarter's tile into its slab PAST the caches
// (agent-scope stores: the four quarters of a tile run on different XCDs, whose L2s do not see each other's dirty lines),
// drains them (vmcnt(0) + barrier) and takes a ticket of the tile (agent-scope atomic); the workgroup that draws the last
// one loads the other three quarters the same way, adds the four in slab order ((0 + 1) + 2) + 3 - what the reduce launch
// computed, so the values do not depend on who arrives last -, applies the epilogue and re-arms the ticket.  No workgroup
// waits for another; no device-scope fence (its L2 write-back costs tens of microseconds, MI355X_MICROARCH.md).
template <bool DGRAD, int NS, bool FINAL>   // NS = k-steps per wave = (8 C / 16) / (4 quarters x 4 waves) = C / 32
__global__ __launch_bounds__(256, 2) void cube2_kernel(Cube2Params p) {
    __shared__ float red[4][32][64];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int C = p.C, cpb = C >> 5;                     // column blocks per voxel
    const int cbi = blockIdx.x, q = blockIdx.y, mt = blockIdx.z;
    const int cv = cbi / cpb, cc0 = (cbi % cpb) * 32;    // column side: voxel cv, channels cc0 .. cc0 + 31
    const int n0 = mt * 64;
    const __amdgpu_buffer_rsrc_t ars = rsrc_of2(p.a, p.a_bytes), wrs = rsrc_of2(p.w, p.w_bytes);

    // k-steps of this wave: global k-step g = (q * 4 + wave) * NS + s covers reduction voxel kv = g / (C / 16), channels
    // kc0 = (g % (C / 16)) * 16 .. + 15; this lane's 8 are kc0 + 8 h ..
    const int g0 = (q * 4 + wave) * NS;
    unsigned a_row[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int n = n0 + rb * 32 + l32;
        a_row[rb] = n < p.N ? 4u * (unsigned)((long)n * 8 * C) : 0x80000000u;
    }
    constexpr int PD = NS < 4 ? NS : 4;   // k-steps in flight: the loop is a chain of L2 round trips, so the fetches run
                                          // four k-steps ahead of the MFMAs (one ahead: 12.3 us per launch)
    u32x4 araw[PD][2][2];              // [set][row block][16-byte half]
    unsigned braw[PD][8];              // [set][k]
    auto fetch = [&](int s, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int g = g0 + s, kv = g / (C >> 4), kc = (g % (C >> 4)) * 16 + 8 * h;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned off = a_row[rb] + 4u * (unsigned)(kv * C + kc);
            araw[SET][rb][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)off, 0, 0);
            araw[SET][rb][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(off + 16u), 0, 0);
        }
        if (DGRAD) {
            // reduction side = output voxel kv, channels co; column side = input voxel cv, channel ci = cc0 + l32:
            // B[k][col] = W[tap(cv, kv)][ci][co]: the lane's 8 k's are contiguous
            const int tap = pair_tap(cv, kv);
            const unsigned off = 4u * (unsigned)(((long)tap * C + cc0 + l32) * C + kc);
            const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0);
            const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(off + 16u), 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { braw[SET][e] = lo[e]; braw[SET][4 + e] = hi[e]; }
        } else {
            // reduction side = input voxel kv, channels ci; column side = output voxel cv, channel co = cc0 + l32:
            // B[k][col] = W[tap(kv, cv)][ci][co]: the lane's 8 k's are C floats apart
            const int tap = pair_tap(kv, cv);
            const unsigned off = 4u * (unsigned)(((long)tap * C + kc) * C + cc0 + l32);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                braw[SET][e] = __builtin_amdgcn_raw_buffer_load_b32(wrs, (int)(off + 4u * (unsigned)(e * C)), 0, 0);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    auto fetch_dyn = [&](int s) {
        switch (s % PD) {
            case 0: fetch(s, std::integral_constant<int, 0>{}); break;
            case 1: fetch(s, std::integral_constant<int, 1 % PD>{}); break;
            case 2: fetch(s, std::integral_constant<int, 2 % PD>{}); break;
            default: fetch(s, std::integral_constant<int, 3 % PD>{}); break;
        }
    };
#pragma unroll
    for (int s = 0; s < PD - 1; ++s) fetch_dyn(s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + PD - 1 < NS) fetch_dyn(s + PD - 1);
        bf16x8 af[2][3], bf[3];
        float v[8];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(araw[s % PD][rb][0][e]); v[4 + e] = __uint_as_float(araw[s % PD][rb][1][e]); }
            cut8r(v, af[rb]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(braw[s % PD][e]);
        cut8r(v, bf);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][PA[pr]], bf[PB[pr]], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][PA[pr]], bf[PB[pr]], acc[1], 0, 0, 0);
        }
    }

    // ---- the four waves' tiles added through LDS (fixed order: wave 0 + 1 + 2 + 3), wave w finishing a quarter ----
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][rb * 16 + r][lane] = acc[rb][r];
    __syncthreads();
    const long slab = (long)p.N * 8 * C;
    float* out = p.slabs + (long)q * slab;
    float tv[8];
    long o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = wave * 8 + j, rb = idx >> 4, r = idx & 15;
        tv[j] = ((red[0][idx][lane] + red[1][idx][lane]) + red[2][idx][lane]) + red[3][idx][lane];
        const int n = n0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;          // C/D layout: row of the 32 x 32 block
        o[j] = n < p.N ? ((long)n * 8 + cv) * C + cc0 + l32 : -1;
        if (!FINAL) { if (o[j] >= 0) out[o[j]] = tv[j]; }
        else if (o[j] >= 0) __hip_atomic_store(out + o[j], tv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!FINAL) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the slab stores have left the workgroup
    __syncthreads();
    unsigned* ticket = p.tickets + (mt * gridDim.x + cbi);
    if (tid == 0) s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == KQ - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // armed for the next launch
    float part[KQ][8];
#pragma unroll
    for (int z = 0; z < KQ; ++z)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            part[z][j] = (z == q || o[j] < 0) ? tv[j] : __hip_atomic_load(p.slabs + (long)z * slab + o[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (o[j] < 0) continue;
        float t = ((part[0][j] + part[1][j]) + part[2][j]) + part[3][j];
        if (p.res) t += p.res[o[j]];
        if (p.relu) t = fmaxf(t, 0.f);
        if (p.mask) t = (p.mask[o[j]] > 0.f) ? t : 0.f;
        p.out[o[j]] = t;
    }
}



// ---- weight gradient of the convolutions whose OUTPUT is a 2 x 2 x 2 volume (round 3) -----------------------------------------
// layer3 / feature_3d (256 -> 256 on 2^3, stride 1: four launches per step) and layer3.0.conv1 (128 -> 256, 4^3 -> 2^3, stride 2)
// of the MoCo-3D encoder (moco_encoder_3d.py:55-84,172,178).  dW[tap][ci][co] = sum over samples n and over the (input voxel vi,
// output voxel vo) PAIRS the tap connects - at most 8, one per output voxel, 64 / 27 = 2.4 on average on a stride-1 2^3 volume -
// of X[n][vi][ci] dY[n][vo][co]: per tap a few (ci x n) . (n x co) products whose reduction is the BATCH.  The implicit GEMM
// walks 16 slices per 64 x 64 tile behind a per-slice gather (19 - 20 us per launch for 0.5 GFLOP of real work).  Here a
// workgroup owns one (tap, 64 ci, 64 co) tile and loops over the tap's pairs: the two 64-sample x 64-channel blocks of a pair
// are fetched with 16-byte loads along the channels (their memory order), cut ONCE into bf16x3 rows [sample][32 channels] in
// LDS, and the fragments - which need the sample axis contiguous - come out through the transposing LDS read
// (ds_read_b64_tr_b16), as in direct3_wgrad_kernel.  The next pair's loads are in flight during the MFMAs of the current one;
// eight waves (four 32 x 32 sub-tiles x two halves of a block's k-steps, summed through LDS at the end: a workgroup's time is
// its chain of pair iterations); 48 KB of LDS, two workgroups per CU.  Output tiles are final: no split-K slabs,
// no reduce launch.  Taps are launched heaviest first (the centre tap has 8 pairs, a corner tap 1).
constexpr int PW_ROW = 64;                      // bytes of a (sample, 32 channels) row of one bf16 plane
constexpr int PW_HALF = 64 * PW_ROW;            // 64 samples: one 32-channel half
constexpr int PW_PLANE = 2 * PW_HALF;
constexpr int PW_OP = 3 * PW_PLANE;             // one operand block (64 samples x 64 channels x 3 planes): 24,576 bytes
constexpr int PW_MAXTAP = 27;

constexpr int PW_MAXPROB = 4;     // round 5: problems of one geometry per launch (blockIdx.y)
struct PairWgradParams {
    const float* x[PW_MAXPROB];       // (N, VI voxels, CI)
    const float* dy[PW_MAXPROB];      // (N, VO voxels, CO)
    float* dw[PW_MAXPROB];            // [ntaps][CI][CO]: final (S == 1) or slab 0 of S slabs
    int N, VI, VO, CI, CO, ntaps;
    unsigned x_bytes, dy_bytes;
    // geometry of the pairs (round 4: any cubic volume): per axis, tap t connects outputs lo .. lo + len - 1 to inputs stride * o + t - pad
    int Di, Do, stride, pad, ks;
    int S;                    // segments of a tap's chain of (pair, sample chunk) iterations: S slabs [S][ntaps][CI][CO], summed by the caller
    long slab_stride;         // floats between slabs
    unsigned char order[PW_MAXTAP];       // taps, heaviest first
    unsigned short cnt[PW_MAXTAP];        // pairs of a tap
    unsigned char lo[PW_MAXTAP][3], len[PW_MAXTAP][3];   // [tap][z, y, x]
};

typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4w lds_bf16x4w;

__global__ __launch_bounds__(512, 2) void pair_wgrad_kernel(PairWgradParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PW_OP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
    // eight waves: wave & 3 = the 32 x 32 sub-tile (ci half wm x co half wn), wave >> 2 = the half of a block's four k-steps it
    // takes - a workgroup's time is its chain of pair iterations (the centre tap has 8), so an iteration is cut in two
    const int wm = (wave >> 1) & 1, wn = wave & 1, kh = wave >> 2;
    const int ncb = p.CO >> 6, nib = p.CI >> 6;
    const int tiles = nib * ncb;
    const int rem = blockIdx.x % tiles, ts = blockIdx.x / tiles;
    const int tap = p.order[ts / p.S];
    const int seg = ts % p.S;
    const int ib = rem / ncb, cb = rem % ncb;
    const int npairs = p.cnt[tap], nchunks = (p.N + 63) >> 6;
    // this workgroup's share of the tap's chain: iterations it0 .. total - 1 (S == 1: the whole chain, the tile is final)
    const int all = npairs * nchunks, per = (all + p.S - 1) / p.S;
    const int it0 = seg * per, total = min(all, it0 + per);
    const int tz = tap / (p.ks * p.ks), ty = (tap / p.ks) % p.ks, tx = tap % p.ks;
    const int lz = p.lo[tap][0], ly = p.lo[tap][1], lx = p.lo[tap][2], ny = p.len[tap][1], nx = p.len[tap][2];

    // staging: thread = (sample row, 8 channels): one unit of X and one of dY per block
    const __amdgpu_buffer_rsrc_t xrs = rsrc_of2(p.x[blockIdx.y], p.x_bytes), yrs = rsrc_of2(p.dy[blockIdx.y], p.dy_bytes);
    const int srow = tid >> 3, cg = tid & 7;
    const int st_lds = (cg >> 2) * PW_HALF + srow * PW_ROW + (cg & 3) * 16;
    // three blocks in flight: an iteration (12 MFMAs per wave) is far shorter than an L2 / HBM round trip
    u32x4 ldx[3][2], ldy[3][2];
    // the fetches walk the chain in order: (pair, sample chunk) of the NEXT fetch as counters, advanced like an odometer (pairs of a tap:
    // output voxels of the tap's box, z-major - ascending output voxel).  (Divisions per fetch cost every wave ~170 scalar instructions.)
    int f_it = it0, f_chunk = it0 % nchunks, f_px, f_py, f_pz;
    { const int pr = it0 / nchunks; f_px = pr % nx; f_py = (pr / nx) % ny; f_pz = pr / (nx * ny); }
    auto fetch = [&](auto Sc, int) {
        constexpr int S = decltype(Sc)::value;
        const int n = f_chunk * 64 + srow;
        const bool ok = f_it < total && n < p.N;
        const int oz = lz + f_pz, oy = ly + f_py, ox = lx + f_px;
        const int vo = (oz * p.Do + oy) * p.Do + ox;
        const int vi = ((p.stride * oz + tz - p.pad) * p.Di + p.stride * oy + ty - p.pad) * p.Di + p.stride * ox + tx - p.pad;
        ++f_it;
        if (++f_chunk == nchunks) { f_chunk = 0; if (++f_px == nx) { f_px = 0; if (++f_py == ny) { f_py = 0; ++f_pz; } } }
        const unsigned xo = ok ? 4u * (unsigned)(((long)n * p.VI + vi) * p.CI + ib * 64 + cg * 8) : 0x80000000u;
        const unsigned yo = ok ? 4u * (unsigned)(((long)n * p.VO + vo) * p.CO + cb * 64 + cg * 8) : 0x80000000u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ldx[S][q] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(xo + 16u * q), 0, 0);
            ldy[S][q] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)(yo + 16u * q), 0, 0);
        }
    };
    auto store = [&](auto Sc, int boff) {
        constexpr int S = decltype(Sc)::value;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __uint_as_float(op ? ldy[S][0][e] : ldx[S][0][e]);
                v[4 + e] = __uint_as_float(op ? ldy[S][1][e] : ldx[S][1][e]);
            }
            bf16x8 o[3];
            cut8r(v, o);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<u32x4*>(lds + boff + op * PW_OP + pl * PW_PLANE + st_lds) = __builtin_bit_cast(u32x4, o[pl]);
        }
    };
    // fragment addresses (transposing read: this lane names row q4 of its 16-lane group's 4-row block; see
    // direct3_wgrad_kernel): k-step ks covers samples 16 ks .. 16 ks + 15, MFMA k = 8 h + e
    const int q4 = i16 >> 2;
    const int coloff = (16 * g16 + 4 * (i16 & 3)) * 2;
    const int a_base = wm * PW_HALF + (8 * h + q4 + 32 * kh) * PW_ROW + coloff;
    const int b_base = PW_OP + wn * PW_HALF + (8 * h + q4 + 32 * kh) * PW_ROW + coloff;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    auto multiply = [&](int boff) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[3], bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned char* ap = lds + boff + a_base + pl * PW_PLANE + ks * 16 * PW_ROW;
                const unsigned char* bp = lds + boff + b_base + pl * PW_PLANE + ks * 16 * PW_ROW;
                const bf16x4w alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(ap));
                const bf16x4w ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(ap + 4 * PW_ROW));
                const bf16x4w blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(bp));
                const bf16x4w bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4w*)(bp + 4 * PW_ROW));
                af[pl] = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
                bf[pl] = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bf[PB[pr]], acc, 0, 0, 0);
        }
    };
    auto body = [&](auto Sc, auto, int it) {
        store(Sc, 0);
        __syncthreads();
        fetch(Sc, it + 3);                               // (behind the last block: out of range, zeros, never stored)
        multiply(0);
        __syncthreads();                                 // every wave has read this block before the next one is stored
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    fetch(I0{}, it0);
    fetch(I1{}, it0 + 1);
    fetch(I2{}, it0 + 2);
    // (whole rounds of three without a branch between the bodies: the compiler's wait counts then name the oldest block only - with the
    // tail's conditions inside the loop it drained the two younger blocks at every loop head)
    int it = it0;
    for (; it + 3 <= total; it += 3) {
        body(I0{}, I1{}, it);
        body(I1{}, I2{}, it + 1);
        body(I2{}, I0{}, it + 2);
    }
    if (it < total) {
        body(I0{}, I1{}, it);
        if (it + 1 < total) body(I1{}, I2{}, it + 1);
    }
    // the two k-halves meet in LDS (staging is over): waves 4..7 publish, waves 0..3 add (first half + second half) and store
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds);
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave & 3][r][lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
        // final tile: C/D layout col = lane & 31 (co), row = ci
        float* out = p.dw[blockIdx.y] + seg * p.slab_stride + ((long)tap * p.CI + ib * 64 + 32 * wm) * p.CO + cb * 64 + 32 * wn + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(long)((r & 3) + 8 * (r >> 2) + 4 * h) * p.CO] = acc[r] + red[wave][r][lane];
    }
}

// ---- small dense products: the Linear layers of the encoder head (fc, projection MLP: 64 rows) ---------------------------
//   C[m][n] = sum_k A(m, k) * B(k, n) (+ bias[n]),   A(m, k) = a[m * lda_m + k * lda_k],  B(k, n) = b[k * ldb_k + n * ldb_n]
// covers y = x W + b (A k-contiguous, B strided), dx = dy W^T (both k-contiguous) and dW = x^T dy (both strided: the batch
// is the reduction).  Same register-staged form as cube2_kernel: a workgroup owns a 64 x 32 tile, its four waves take the
// k-steps round-robin and add their tiles through LDS; the launch is final (bias included): no split-K slabs, no reduce
// launch.  The implicit GEMM spends 8 - 14 us + a reduce launch on each of these 16 calls per step.
struct SmallGemmParams {
    const float* a;
    const float* b;
    const float* bias;        // may be null
    float* c;
    int M, N, K;
    long lda_m, lda_k, ldb_k, ldb_n;
    unsigned a_bytes, b_bytes;
    // training-mode BatchNorm1d (+ ReLU) of C in the same launch (M <= 64: the workgroup's tile holds every row of its 32
    // columns, so the batch statistics are its own): bn_y = act(bn(C)); C itself is still written (the backward reads it)
    float* bn_y;              // null: plain product
    const float* gamma;       // may be null
    const float* beta;
    float eps, momentum;
    float* running_mean;      // may be null (with running_var)
    float* running_var;
    long long* num_batches_tracked;     // may be null
    float* save;              // mean[N], invstd[N]
    int relu;
    double* sums_only;        // not null (with bn_y null): write the column sums of C instead (SyncBN's statistics pass)
};

template <bool A_KC, bool B_KC, int NS>     // operand is k-contiguous (16-byte loads) or strided (8 dwords); NS k-steps per wave
__global__ __launch_bounds__(256, 2) void small_gemm_kernel(SmallGemmParams p) {
    __shared__ float red[4][32][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
    const int KT = (p.K + 15) >> 4;
    const __amdgpu_buffer_rsrc_t ars = rsrc_of2(p.a, p.a_bytes), brs = rsrc_of2(p.b, p.b_bytes);
    constexpr int PD = NS < 4 ? NS : 4;
    unsigned araw[PD][2][8], braw[PD][8];
    const bool col_ok = n0 + l32 < p.N;
    auto fetch = [&](auto Uc) {
        constexpr int U = decltype(Uc)::value, SET = U % PD;
        if constexpr (U < NS) {
            const int ks = wave + 4 * U;
            const int k0 = ks * 16 + 8 * h;
            const bool live = ks < KT;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int m = m0 + rb * 32 + l32;
                const bool ok = live && m < p.M;
                if (A_KC) {
                    const unsigned off = (ok && k0 + 8 <= p.K) ? 4u * (unsigned)(m * p.lda_m + k0) : 0x80000000u;
                    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)off, 0, 0);
                    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(off + 16u), 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { araw[SET][rb][e] = lo[e]; araw[SET][rb][4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned off = (ok && k0 + e < p.K) ? 4u * (unsigned)(m * p.lda_m + (long)(k0 + e) * p.lda_k) : 0x80000000u;
                        araw[SET][rb][e] = __builtin_amdgcn_raw_buffer_load_b32(ars, (int)off, 0, 0);
                    }
                }
            }
            const bool okb = live && col_ok;
            if (B_KC) {
                const unsigned off = (okb && k0 + 8 <= p.K) ? 4u * (unsigned)((long)(n0 + l32) * p.ldb_n + k0) : 0x80000000u;
                const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)off, 0, 0);
                const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)(off + 16u), 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { braw[SET][e] = lo[e]; braw[SET][4 + e] = hi[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned off = (okb && k0 + e < p.K) ? 4u * (unsigned)((long)(k0 + e) * p.ldb_k + (long)(n0 + l32) * p.ldb_n) : 0x80000000u;
                    braw[SET][e] = __builtin_amdgcn_raw_buffer_load_b32(brs, (int)off, 0, 0);
                }
            }
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    auto step = [&](auto Uc) {
        constexpr int U = decltype(Uc)::value, SET = U % PD;
        if constexpr (U < NS) {
            fetch(std::integral_constant<int, U + PD - 1>{});
            if (wave + 4 * U < KT) {                               // (wave-uniform)
                bf16x8 af[2][3], bf[3];
                float v[8];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(araw[SET][rb][e]);
                    cut8r(v, af[rb]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(braw[SET][e]);
                cut8r(v, bf);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][PA[pr]], bf[PB[pr]], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][PA[pr]], bf[PB[pr]], acc[1], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (PD > 1) fetch(std::integral_constant<int, 0>{});
    if constexpr (PD > 2) fetch(std::integral_constant<int, 1>{});
    if constexpr (PD > 3) fetch(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][rb * 16 + r][lane] = acc[rb][r];
    __syncthreads();
    const float bv = (p.bias && col_ok) ? p.bias[n0 + l32] : 0.f;
    float tv[8];
    double cs = 0, css = 0;                              // column sums over this thread's rows (BatchNorm: doubles, as bn_small_fwd)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = wave * 8 + j, rb = idx >> 4, r = idx & 15;
        const float t = ((red[0][idx][lane] + red[1][idx][lane]) + red[2][idx][lane]) + red[3][idx][lane];
        const int m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        tv[j] = t + bv;
        if (m < p.M && col_ok) {
            p.c[(long)m * p.N + n0 + l32] = tv[j];
            cs += (double)tv[j]; css += (double)tv[j] * (double)tv[j];
        }
    }
    if (p.bn_y || p.sums_only) {                         // (uniform: launch arguments)
        __shared__ double bred[4][2][32];
        cs += __shfl_xor(cs, 32, 64); css += __shfl_xor(css, 32, 64);
        if (h == 0) { bred[wave][0][l32] = cs; bred[wave][1][l32] = css; }
        __syncthreads();
        const double S = (bred[0][0][l32] + bred[1][0][l32]) + (bred[2][0][l32] + bred[3][0][l32]);
        const double SS = (bred[0][1][l32] + bred[1][1][l32]) + (bred[2][1][l32] + bred[3][1][l32]);
        if (p.sums_only) {                               // SyncBN: this rank's sums; the all-reduce and the apply follow
            if (wave == 0 && h == 0 && col_ok) { p.sums_only[n0 + l32] = S; p.sums_only[p.N + n0 + l32] = SS; }
            return;
        }
        const double mean = S / p.M;
        double var = SS / p.M - mean * mean;
        if (var < 0) var = 0;
        const float mf = (float)mean, iv = (float)(1.0 / sqrt(var + (double)p.eps));
        const int col = n0 + l32;
        if (wave == 0 && h == 0 && col_ok) {
            p.save[col] = mf; p.save[p.N + col] = iv;
            if (p.running_mean) {
                const double unbiased = p.M > 1 ? var * p.M / (p.M - 1.0) : var;
                p.running_mean[col] = (1.f - p.momentum) * p.running_mean[col] + p.momentum * mf;
                p.running_var[col] = (1.f - p.momentum) * p.running_var[col] + p.momentum * (float)unbiased;
            }
        }
        if (blockIdx.x == 0 && tid == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
        const float g = (p.gamma && col_ok) ? p.gamma[col] : 1.f, b = (p.beta && col_ok) ? p.beta[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = wave * 8 + j, rb = idx >> 4, r = idx & 15;
            const int m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float q = fmaf((tv[j] - mf) * iv, g, b);
            if (p.relu) q = fmaxf(q, 0.f);
            if (m < p.M && col_ok) p.bn_y[(long)m * p.N + col] = q;
        }
    }
}

}  // namespace

bool mi_cube2_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw,
                     int dd, int dh, int dw) {
    const char* off = getenv("MI_CONV_NO_DIRECT");
    if (off && atoi(off) != 0) return false;
    if (kd != 3 || kh != 3 || kw != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1 || dd != 1 || dh != 1 || dw != 1) return false;
    if (Di != 2 || Hi != 2 || Wi != 2 || Ci != Co) return false;
    if (Ci != 128 && Ci != 256 && Ci != 512) return false;                    // k-steps per wave = C / 32: 4, 8, 16
    return N >= 1 && 4l * N * 8 * Ci < 0x7fff0000l;
}


// weight gradient through pair_wgrad_kernel: k^3 window (k = 3, pad 1 or k = 1, pad 0), cubic volumes, stride 1 or 2, channels multiples
// of 64.  Round 3: outputs of 2 x 2 x 2 (layer3).  Round 4: any cubic output up to 8^3 - layer2 (4^3, up to 64 pairs per tap), layer2.0's
// stride-2 convolution and 1 x 1 x 1 shortcut - with a tap's chain of pairs cut into S segments (S slabs, summed by the caller's reduce).
namespace {
struct PairGeom { int ntaps, maxcnt; long pairs; };
PairGeom pair_geom(int Di, int k, int stride, PairWgradParams* p) {
    const int pad = k == 3 ? 1 : 0, Do = (Di + 2 * pad - k) / stride + 1;
    PairGeom r = {k * k * k, 0, 0};
    for (int t = 0; t < r.ntaps; ++t) {
        const int tt[3] = {t / (k * k), (t / k) % k, t % k};
        int c = 1;
        for (int a = 0; a < 3; ++a) {
            int lo = Do, hi = -1;
            for (int o = 0; o < Do; ++o) {
                const int i = stride * o + tt[a] - pad;
                if (i >= 0 && i < Di) { lo = std::min(lo, o); hi = std::max(hi, o); }
            }
            const int len = hi >= lo ? hi - lo + 1 : 0;
            if (p) { p->lo[t][a] = (unsigned char)(len ? lo : 0); p->len[t][a] = (unsigned char)std::max(len, 1); }
            c *= len;
        }
        if (p) p->cnt[t] = (unsigned short)c;
        r.maxcnt = std::max(r.maxcnt, c);
        r.pairs += c;
    }
    return r;
}
}  // namespace

bool mi_pair_wgrad_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw,
                          int dd, int dh, int dw) {
    const char* off = getenv("MI_CONV_NO_DIRECT");
    if (off && atoi(off) != 0) return false;
    if (kd != kh || kd != kw || pd != ph || pd != pw || dd != 1 || dh != 1 || dw != 1) return false;
    if (!((kd == 3 && pd == 1) || (kd == 1 && pd == 0)) || (stride != 1 && stride != 2)) return false;
    if (Hi != Di || Wi != Di || Di < 2) return false;
    const int Do = (Di + 2 * pd - kd) / stride + 1;
    // largest output extent taken: 2 (layer3) by default - on 4^3 (layer2) the segmented form is no faster than the implicit GEMM
    // (r04_experiments.txt item 26); MI_PAIR_WGRAD_MAXOUT=4 (or 8) takes those too
    const char* wide = getenv("MI_PAIR_WGRAD_MAXOUT");
    if (Do < 1 || Do > (wide ? atoi(wide) : 2) || Do > 8) return false;
    if (Ci < 64 || Co < 64 || (Ci & 63) || (Co & 63)) return false;
    return N >= 1 && 4l * N * Di * Hi * Wi * Ci < 0x7fff0000l && 4l * N * Do * Do * Do * Co < 0x7fff0000l;
}

// segments of a tap's chain: about eight iterations per workgroup (four while that leaves CUs without one), at most 16
int mi_pair_wgrad_splits(int N, int Di, int Ci, int Co, int k, int stride) {
    const PairGeom g = pair_geom(Di, k, stride, nullptr);
    if (g.maxcnt <= 8) return 1;                                   // 2^3 outputs: final in one launch (round 3)
    const char* e = getenv("MI_PAIR_WGRAD_SPLITS");
    if (e && atoi(e) > 0) return std::min(atoi(e), 64);
    const int iters = g.maxcnt * ((N + 63) / 64), tiles = (Ci / 64) * (Co / 64);
    int S = (iters + 7) / 8;
    if (g.ntaps * tiles * S < 256) S = (iters + 3) / 4;
    return std::max(1, std::min(S, 16));
}
size_t mi_pair_wgrad_slab_bytes(int N, int Di, int Ci, int Co, int k, int stride) {
    const int S = mi_pair_wgrad_splits(N, Di, Ci, Co, k, stride);
    return S > 1 ? sizeof(float) * (size_t)S * k * k * k * Ci * Co : 0;
}

// S == 1: `dwt` is final.  S > 1 (mi_pair_wgrad_splits): S slabs of [k^3][Ci][Co] floats into `slabs`; the caller sums them
int mi_pair_wgrad_batch_max() { return PW_MAXPROB; }
int mi_pair_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* dws, float* const* slabs, int nb, int N,
                               int Di, int Ci, int Co, int k, int stride, hipStream_t s) {
    if (nb < 1 || nb > PW_MAXPROB) return MI_E_ARG;
    PairWgradParams p = {};
    const PairGeom g = pair_geom(Di, k, stride, &p);
    const int pad = k == 3 ? 1 : 0, Do = (Di + 2 * pad - k) / stride + 1;
    const int S = mi_pair_wgrad_splits(N, Di, Ci, Co, k, stride);
    for (int i = 0; i < nb; ++i) {
        if (S > 1 && !slabs[i]) return MI_E_ARG;
        p.x[i] = xs[i]; p.dy[i] = dys[i]; p.dw[i] = S > 1 ? slabs[i] : dws[i];
    }
    p.N = N; p.VI = Di * Di * Di; p.VO = Do * Do * Do; p.CI = Ci; p.CO = Co; p.ntaps = g.ntaps;
    p.x_bytes = (unsigned)(4l * N * p.VI * Ci); p.dy_bytes = (unsigned)(4l * N * p.VO * Co);
    p.Di = Di; p.Do = Do; p.stride = stride; p.pad = pad; p.ks = k; p.S = S; p.slab_stride = (long)g.ntaps * Ci * Co;
    // heaviest taps first (stable); a tap without a pair (none with these geometries) writes zeros
    int n = 0;
    for (int want = g.maxcnt; want >= 0; --want)
        for (int t = 0; t < g.ntaps; ++t)
            if (p.cnt[t] == want) p.order[n++] = (unsigned char)t;
    hipLaunchKernelGGL(pair_wgrad_kernel, dim3((unsigned)(g.ntaps * (Ci / 64) * (Co / 64) * S), (unsigned)nb), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
int mi_pair_wgrad_launch(const float* x, const float* dy, float* dwt, float* slabs, int N, int Di, int Ci, int Co, int k, int stride,
                         hipStream_t s) {
    const float* xs[1] = {x};
    const float* dys[1] = {dy};
    float* dws[1] = {dwt};
    float* sl[1] = {slabs};
    return mi_pair_wgrad_launch_batch(xs, dys, dws, sl, 1, N, Di, Ci, Co, k, stride, s);
}

size_t mi_cube2_slab_bytes(int N, int C) { return sizeof(float) * (size_t)KQ * N * 8 * C; }
int mi_cube2_splits() { return KQ; }

// KQ partial slabs of (N * 8, C) floats into `slabs`; the caller sums them (+ epilogue) - or, with `fin`, the launch is
// final: tickets (MI_CUBE2_TICKET_BYTES, zero between launches; mi_cube2_final_usable) + out / res / mask / relu of the epilogue
bool mi_cube2_final_usable(int N, int C) { return sizeof(unsigned) * (size_t)(8 * C / 32) * ((N + 63) / 64) <= MI_CUBE2_TICKET_BYTES; }
int mi_cube2_launch(int dgrad, const float* a, const float* w, float* slabs, int N, int C, hipStream_t s, const Cube2Final* fin) {
    Cube2Params p = {a, w, slabs, N, C, (unsigned)(4l * N * 8 * C), (unsigned)(4l * 27 * C * C), nullptr, nullptr, nullptr, nullptr, 0};
    if (fin) { p.tickets = fin->tickets; p.out = fin->out; p.res = fin->res; p.mask = fin->mask; p.relu = fin->relu; }
    const dim3 grid((unsigned)(8 * C / 32), KQ, (unsigned)((N + 63) / 64));
#define CUBE2_LAUNCH(D, NSV) do { if (fin) hipLaunchKernelGGL((cube2_kernel<D, NSV, true>), grid, dim3(256), 0, s, p); \
                                  else hipLaunchKernelGGL((cube2_kernel<D, NSV, false>), grid, dim3(256), 0, s, p); } while (0)
    if (C == 128) { if (dgrad) CUBE2_LAUNCH(true, 4); else CUBE2_LAUNCH(false, 4); }
    else if (C == 256) { if (dgrad) CUBE2_LAUNCH(true, 8); else CUBE2_LAUNCH(false, 8); }
    else if (C == 512) { if (dgrad) CUBE2_LAUNCH(true, 16); else CUBE2_LAUNCH(false, 16); }
    else return MI_E_UNSUPPORTED;
#undef CUBE2_LAUNCH
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// C[M][N] = A . B (+ bias) with strided operand views (see small_gemm_kernel); K <= 1024, tensors < 2 GiB
bool mi_small_gemm_usable(long M, long N, long K) {
    const char* off = getenv("MI_CONV_NO_DIRECT");
    if (off && atoi(off) != 0) return false;
    return M >= 1 && M <= 1024 && N >= 1 && N <= 4096 && K >= 1 && K <= 1024;
}
int mi_small_gemm_launch(const float* a, long lda_m, long lda_k, long a_elems, const float* b, long ldb_k, long ldb_n,
                         long b_elems, const float* bias, float* c, int M, int N, int K, hipStream_t s, const MiSmallGemmBN* bn) {
    if (4 * a_elems >= 0x7fff0000l || 4 * b_elems >= 0x7fff0000l) return MI_E_UNSUPPORTED;
    if (bn && M > 64) return MI_E_UNSUPPORTED;
    if (bn && !bn->sums_only && (!bn->y || !bn->save || (bn->running_mean == nullptr) != (bn->running_var == nullptr))) return MI_E_UNSUPPORTED;
    SmallGemmParams p = {a, b, bias, c, M, N, K, lda_m, lda_k, ldb_k, ldb_n, (unsigned)(4 * a_elems), (unsigned)(4 * b_elems)};
    p.bn_y = nullptr; p.sums_only = nullptr;
    if (bn && bn->sums_only) p.sums_only = bn->sums_only;
    else if (bn) {
        p.bn_y = bn->y; p.gamma = bn->gamma; p.beta = bn->beta; p.eps = bn->eps; p.momentum = bn->momentum;
        p.running_mean = bn->running_mean; p.running_var = bn->running_var; p.num_batches_tracked = bn->num_batches_tracked;
        p.save = bn->save; p.relu = bn->relu;
    }
    const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 63) / 64));
    const int kt = (K + 15) / 16, per = (kt + 3) / 4;
    const int ns = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 16;
    const bool akc = lda_k == 1 && (lda_m % 4) == 0 && (K % 8) == 0, bkc = ldb_k == 1 && (ldb_n % 4) == 0 && (K % 8) == 0;
#define SG_LAUNCH(AK, BK, NSV) hipLaunchKernelGGL((small_gemm_kernel<AK, BK, NSV>), grid, dim3(256), 0, s, p)
#define SG_NS(AK, BK)                                                                                   \
    switch (ns) {                                                                                       \
        case 1: SG_LAUNCH(AK, BK, 1); break;                                                            \
        case 2: SG_LAUNCH(AK, BK, 2); break;                                                            \
        case 4: SG_LAUNCH(AK, BK, 4); break;                                                            \
        case 8: SG_LAUNCH(AK, BK, 8); break;                                                            \
        default: SG_LAUNCH(AK, BK, 16); break;                                                          \
    }
    if (akc && bkc) { SG_NS(true, true) }
    else if (akc) { SG_NS(true, false) }
    else if (bkc) { SG_NS(false, true) }
    else { SG_NS(false, false) }
#undef SG_NS
#undef SG_LAUNCH
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---- C-ABI (include/cetpick_hip.h): the 2 x 2 x 2 convolutions, final in one launch ----
void mi_note_conv_kernel(const char* name);          // conv_igemm.hip: mi_debug_last_conv_kernel
extern "C" size_t mi_conv3d_cube2_workspace_bytes(int N, int C) { return MI_CUBE2_TICKET_BYTES + mi_cube2_slab_bytes(N, C); }

extern "C" int mi_conv3d_cube2_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
    return mi_cube2_usable(N, Di, Hi, Wi, Ci, Co, k, k, k, stride, pad, pad, pad, 1, 1, 1) && mi_cube2_final_usable(N, Ci) ? 1 : 0;
}

extern "C" int mi_conv3d_cube2_f32(const float* a, const float* w, float* out, const float* res, const float* mask, int relu,
                                   int dgrad, int N, int C, void* ws, size_t ws_bytes, mi_stream_t stream) {
    if (!a || !w || !out) return MI_E_ARG;
    if (!mi_conv3d_cube2_usable(N, 2, 2, 2, C, C, 3, 1, 1)) return MI_E_UNSUPPORTED;
    if (!ws || ws_bytes < mi_conv3d_cube2_workspace_bytes(N, C)) return MI_E_WORKSPACE;
    Cube2Final fin = {(unsigned*)ws, out, res, mask, relu};
    mi_note_conv_kernel("cube2");
    return mi_cube2_launch(dgrad ? 1 : 0, a, w, (float*)((char*)ws + MI_CUBE2_TICKET_BYTES), N, C, (hipStream_t)stream, &fin);
}
