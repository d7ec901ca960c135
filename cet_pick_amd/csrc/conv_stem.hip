// The 7x7x7 stride-2 stem convolution (Cin = 1 -> 64 channels, models/networks/moco_encoder_3d.py:163-169) as a
// DIRECT convolution on the matrix cores.  With one input channel the implicit GEMM's A operand is just the image:
// a workgroup stages the input patch of its output tile in LDS once and every MFMA A-fragment is ONE ds_read_b32
// of that patch at (lane base + compile-time offset) - no im2col gather, no address arithmetic, no LDS stores of
// A in the reduction loop.  The generic kernel (conv_igemm.hip, STEM path) spends its time on 8 scalar gathers +
// tap-LUT reads per thread and slice; this one issues 3 LDS reads per 2 MFMAs and nothing else.
//
// FWD  tile = 8(x) x 4(y) x 4(z) output voxels x 64 channels, 4 waves = 4 z-planes, 2 accumulators per wave.
//      patch = 13 x 13 x 21 input voxels (rows padded to 24 floats: the 32 lanes of a half-wave then hit 32
//      distinct banks).  The weights stream through LDS one kz-slab (49 taps padded to 50 x 64) at a time,
//      double-buffered, one barrier per slab.  Reduction index inside a slab: k = ky*7 + kx, the lane half h owns
//      k = 2t + h, whose patch offset differs from that of 2t by 1 (same row) or by PW - 6 (row wrap) - two lane
//      base registers cover both cases, the rest is an immediate.
// (no SLP vectorisation: it pairs the cut's subtractions into v_pk_add_f32 - 14 cycles of matrix-pipe throughput next to the MFMAs
// where a scalar v_sub_f32 is hidden, tools/probes/mfma_coissue.hip)
// hipcc-flags: -fno-slp-vectorize
#include "common.h"
#include <type_traits>
#include "../../include/cetpick_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K7 = 7, S2 = 2, P3 = 3;
constexpr int TX = 8, TY = 4, TZ = 4;                    // output tile
constexpr int PX = S2 * (TX - 1) + K7;                   // 21
constexpr int PY = S2 * (TY - 1) + K7;                   // 13
constexpr int PZ = S2 * (TZ - 1) + K7;                   // 13
constexpr int PW = 24;                                   // padded patch row
constexpr int PATCH = PZ * PY * PW;                      // 4056 floats
constexpr int KS = 50;                                   // taps per kz slab (49 + 1 zero row)
constexpr int WS = 64;                                   // weight row stride in LDS; odd rows are rotated by 32 columns so
                                                         // the two half-waves (k even / odd) hit disjoint banks
constexpr int WROWS = 64;                                // rows per LDS slab (4 float4 per thread, stored unconditionally)
constexpr int CO = 64;

struct StemParams {
    const float* x;      // (N, D, H, W) image
    const float* w;      // [343][64]
    float* y;            // (N, Do, Ho, Wo, 64)
    const float* res;    // may be null
    int relu;
    int N, D, H, W, Do, Ho, Wo;
    unsigned x_bytes;
    double* stats;       // bf16x3 kernel only, may be null: per-workgroup column sums of y and y^2, [2][64][gridDim.x]
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
    constexpr int NP = (PATCH + PW + 255) / 256;         // 16 patch elements per thread
    __shared__ __attribute__((aligned(16))) float patch[NP * 256];   // PATCH + one zeroed row (pad tap of the last slab)
    __shared__ __attribute__((aligned(16))) float wl[2][WROWS * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    // tile coordinates
    const int txn = p.Wo / TX, tyn = p.Ho / TY, tzn = p.Do / TZ;
    int b = blockIdx.x;
    const int bx = b % txn; b /= txn;
    const int by = b % tyn; b /= tyn;
    const int bz = b % tzn;
    const int n = b / tzn;
    const int ox0 = bx * TX, oy0 = by * TY, oz0 = bz * TZ;

    // ---- input patch: all loads in flight before the first LDS store (zero padding = buffer range check) ----
    {
        const __amdgpu_buffer_rsrc_t xr = rsrc(p.x, p.x_bytes);
        const int iz0 = oz0 * S2 - P3, iy0 = oy0 * S2 - P3, ix0 = ox0 * S2 - P3;
        float pv[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            const int px = i % PW, t = i / PW, py = t % PY, pz = t / PY;
            const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
            const bool ok = (px < PX) & (pz < PZ) & ((unsigned)iz < (unsigned)p.D) & ((unsigned)iy < (unsigned)p.H) &
                            ((unsigned)ix < (unsigned)p.W);
            const unsigned off = ok ? 4u * (unsigned)((((long)n * p.D + iz) * p.H + iy) * p.W + ix) : 0x80000000u;
            pv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) patch[tid + u * 256] = pv[u];
    }
    // ---- weight slabs: thread tid moves float4 #tid, #tid+256, ... of the slab (rows >= 49 read as zeros) ----
    constexpr int WLD = WROWS * (CO / 4) / 256;          // 4 float4 per thread
    const __amdgpu_buffer_rsrc_t wr = rsrc(p.w, (unsigned)(sizeof(float) * K7 * K7 * K7 * CO));
    float4 wreg[WLD];
    auto wload = [&](int kz) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int q = tid + i * 256;
            const int row = q / (CO / 4);
            const unsigned off = (row < K7 * K7 && kz < K7)
                ? 4u * (unsigned)((kz * K7 * K7 + row) * CO + (q % (CO / 4)) * 4) : 0x80000000u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)off, 0, 0);
            wreg[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int q = tid + i * 256;
            const int row = q / (CO / 4), c = (q % (CO / 4)) * 4;
            *reinterpret_cast<float4*>(&wl[buf][row * WS + ((c + 32 * (row & 1)) & 63)]) = wreg[i];
        }
    };
    wload(0);
    wstore(0);
    wload(1);

    // lane bases into the patch: output (ox, oy) = (l32 & 7, l32 >> 3) of z-plane `wave`
    const int pb = ((S2 * wave) * PY + S2 * (l32 >> 3)) * PW + S2 * (l32 & 7);
    const float* bA = patch + pb + h;                    // k and k+1 in the same patch row
    const float* bB = patch + pb + h * (PW - (K7 - 1));  // k = (ky, 6): k+1 wraps to (ky+1, 0)
    // B fragments: row k = 2t + h; odd rows are stored rotated by 32 columns
    const int wl0 = h ? WS + 32 + l32 : l32;              // columns  0..31
    const int wl1 = h ? WS + l32 : 32 + l32;              // columns 32..63

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    __syncthreads();
    for (int kz = 0; kz < K7; ++kz) {
        const float* wb0 = wl[kz & 1] + wl0;
        const float* wb1 = wl[kz & 1] + wl1;
        const float* a0 = bA + kz * (PY * PW);
        const float* a1 = bB + kz * (PY * PW);
#pragma unroll
        for (int t = 0; t < KS / 2; ++t) {
            const int k0 = 2 * t, ky = k0 / K7, kx = k0 % K7;
            const float a = (kx == K7 - 1 ? a1 : a0)[ky * PW + kx];
            const float b0 = wb0[k0 * WS], b1 = wb1[k0 * WS];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        // slab kz+1 (in registers since the previous iteration) -> the other buffer; then fetch slab kz+2
        wstore((kz + 1) & 1);
        wload(kz + 2);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31 (channel), row = (r&3) + 8*(r>>2) + 4*h (output voxel) ----
    const int oz = oz0 + wave;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = j * 32 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ox = ox0 + (m & 7), oy = oy0 + (m >> 3);
            const long o = ((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * CO + co;
            float v = j == 0 ? acc0[r] : acc1[r];
            if (p.res) v += p.res[o];
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// FWD on the bf16 matrix pipe with f32-equivalent arithmetic (the "bf16x3" cut of conv_igemm.hip: a = a0 + a1 + a2
// exactly, six products of weight <= 2 per f32 product, f32 accumulate).  The patch is cut ONCE while it is staged
// (three bf16 planes of 48-byte rows); the weights are cut once per call by stem_wprep_kernel into the LDS image
// [slab][plane][row 0..7][co][kx 0..7] (rows r = 7 kz + ky, eight to a slab; rows 49..55 and kx = 7 are zero padding), streamed one
// slab (24 KB) at a time.  One v_mfma_f32_32x32x16_bf16 takes K = 16 = two rows x 8 kx: lane-half h owns row 8 slab + 2u + h, whose eight
// kx taps are eight CONSECUTIVE patch elements (stride-2 convolution: x = 2 ox + kx) - four ds_read_b32 per plane,
// conflict-free on 48-byte rows.  25 k-steps (round 4; 28 with the rows padded per kz) x 6 products x 2 column tiles = 300 MFMAs
// of 32 cycles per wave and z-plane against 350 of 64 cycles in the f32 kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PYP = PY + 1;                              // + one zero row: the padded tap ky = 7 of the last output row
constexpr int PROW = 2 * PW;                             // bytes per bf16 patch row
constexpr int WPL = 8 * CO * 16;                         // bytes per plane of a kz slab: [ky8][co][kx8] bf16
constexpr int WSLAB = 3 * WPL;                           // 24576 bytes
constexpr size_t WPREP_BYTES = (size_t)K7 * WSLAB;       // 172032 bytes

// exact three-way bf16 cut of one f32 (truncation; see conv_igemm.hip): the three bf16 bit patterns
__device__ __forceinline__ void cut3(float a, unsigned& h0, unsigned& h1, unsigned& h2) {
    const unsigned u0 = __float_as_uint(a);
    const float r1 = a - __uint_as_float(u0 & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    h0 = u0 >> 16; h1 = u1 >> 16; h2 = __float_as_uint(r2) >> 16;
}

__global__ __launch_bounds__(256) void stem_wprep_kernel(const float* w, unsigned short* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // (slab, row j of the slab, co, kx8)
    if (i >= K7 * 8 * CO * 8) return;
    // round 4: the 49 (kz, ky) rows of the window are numbered r = 7 kz + ky and packed eight to a slab (rows 49 .. 55 are zero
    // padding): 25 k-steps of two rows instead of 7 x 4 with ky = 7 padded in every kz
    const int kx = i & 7, co = (i >> 3) & (CO - 1), j = (i >> 9) & 7, slab = i >> 12;
    const int r = 8 * slab + j, kz = r / K7, ky = r % K7;
    const float v = (r < K7 * K7 && kx < K7) ? w[((kz * K7 + ky) * K7 + kx) * CO + co] : 0.f;
    unsigned h0, h1, h2;
    cut3(v, h0, h1, h2);
    const int o = slab * (WSLAB / 2) + (j * CO + co) * 8 + kx;
    out[o] = (unsigned short)h0; out[o + WPL / 2] = (unsigned short)h1; out[o + WPL] = (unsigned short)h2;
}

// NBUF = 2: the kz slabs of the weights alternate between two LDS buffers (75 KB, two workgroups per CU); NBUF = 1: one buffer
// (51 KB, THREE workgroups per CU), the next slab stored between two barriers under the last k-step's MFMAs.
// ZPW = 2 (round 4): a wave owns TWO z-planes of the tile (tile 8 x 4 x 8 voxels, patch 21 planes: 66 KB with one weight
// buffer, two workgroups per CU): every weight fragment read from LDS feeds two A fragments (LDS bytes per MFMA 0.75 -> 0.5 KB;
// the reads ran at 73 % of the LDS rate), the patch is 2.6 instead of 3.25 input planes per output plane, and 64 samples of
// 16^3 outputs are 1,024 workgroups = exactly two rounds of 512 resident ones (2,048 on 768 slots left a third round 2/3 empty).
template <int NBUF, int ZPW>
__global__ __launch_bounds__(256, ZPW == 2 ? 2 : (NBUF == 1 ? 3 : 2)) void stem_fwd_bf3_kernel(StemParams p, const unsigned char* wprep) {
    constexpr int TZ = 4 * ZPW;                          // output z-planes per workgroup (shadows the file constant)
    constexpr int PZ = S2 * (TZ - 1) + K7;               // 13 / 21 patch planes
    constexpr int PPLANE = PZ * PYP * PROW;              // 8,736 / 14,112 bytes per bf16 plane
    __shared__ __attribute__((aligned(16))) unsigned char patchb[3 * PPLANE];
    __shared__ __attribute__((aligned(16))) unsigned char wl[NBUF][WSLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int txn = p.Wo / TX, tyn = p.Ho / TY, tzn = p.Do / TZ;
    int b = blockIdx.x;
    const int bx = b % txn; b /= txn;
    const int by = b % tyn; b /= tyn;
    const int bz = b % tzn;
    const int n = b / tzn;
    const int ox0 = bx * TX, oy0 = by * TY, oz0 = bz * TZ;

    // ---- weight slab 0 in flight first, then the patch (pairs of x-neighbours -> one packed dword per plane) ----
    constexpr int WLD = WSLAB / 16 / 256;                // 6 x 16 bytes per thread and slab
    u32x4 wreg[WLD];
    auto wload = [&](int kz) {
        if (kz >= K7) return;
#pragma unroll
        for (int i = 0; i < WLD; ++i)
            wreg[i] = *reinterpret_cast<const u32x4*>(wprep + (size_t)kz * WSLAB + 16 * (tid + i * 256));
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) *reinterpret_cast<u32x4*>(&wl[buf][16 * (tid + i * 256)]) = wreg[i];
    };
    wload(0);
    {
        constexpr int PAIRS = PZ * PYP * (PW / 2);       // 2184
        constexpr int NPP = (PAIRS + 255) / 256;         // 9 pairs per thread
        const __amdgpu_buffer_rsrc_t xr = rsrc(p.x, p.x_bytes);
        const int iz0 = oz0 * S2 - P3, iy0 = oy0 * S2 - P3, ix0 = ox0 * S2 - P3;
        float pv[NPP][2];
#pragma unroll
        for (int u = 0; u < NPP; ++u) {
            const int i = tid + u * 256;
            const int pp = i % (PW / 2), t = i / (PW / 2), py = t % PYP, pz = t / PYP;
            const int iz = iz0 + pz, iy = iy0 + py;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int px = 2 * pp + e, ix = ix0 + px;
                const bool ok = (px < PX) & (py < PY) & (pz < PZ) & ((unsigned)iz < (unsigned)p.D) &
                                ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = ok ? 4u * (unsigned)((((long)n * p.D + iz) * p.H + iy) * p.W + ix) : 0x80000000u;
                pv[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
            }
        }
#pragma unroll
        for (int u = 0; u < NPP; ++u) {
            const int i = tid + u * 256;
            if (i < PAIRS) {
                unsigned a0, a1, a2, b0, b1, b2;
                cut3(pv[u][0], a0, a1, a2);
                cut3(pv[u][1], b0, b1, b2);
                unsigned char* d = patchb + 4 * i;       // pair i of plane 0 (rows are 12 pairs = 48 bytes)
                *reinterpret_cast<unsigned*>(d) = a0 | (b0 << 16);
                *reinterpret_cast<unsigned*>(d + PPLANE) = a1 | (b1 << 16);
                *reinterpret_cast<unsigned*>(d + 2 * PPLANE) = a2 | (b2 << 16);
            }
        }
    }
    wstore(0);
    wload(1);

    // lane base into a patch plane: output (ox, oy) = (l32 & 7, l32 >> 3) of z-plane ZPW * wave (+ zi), row + h
    const int a_base = ((S2 * ZPW * wave) * PYP + S2 * (l32 >> 3)) * PROW + 4 * (l32 & 7);       // (the row of half h: frags)
    constexpr int ZSTEP = S2 * PYP * PROW;               // patch bytes between two output z-planes
    const int b_base = (h * CO + l32) * 16;              // row ky = 2u + h, column l32 (+32 for the second tile)

    f32x16 acc0[ZPW], acc1[ZPW];
#pragma unroll
    for (int zi = 0; zi < ZPW; ++zi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[zi][r] = 0.f; acc1[zi][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first

    // fragments of k-step (kz, u) -> register set u & 1; the reads of step g+1 are issued before the MFMAs of step g
    bf16x8 af[2][ZPW][3], bf0[2][3], bf1[2][3];
    // k-step (slab sl, u): the lane's window row r = 8 sl + 2 u + h = (kz, ky) -> patch row kz PYP + ky; TAILc: the last
    // k-step (sl = 6, u = 0), whose second row (r = 49) does not exist
    auto frags = [&](int sl, int u, auto SETc, auto TAILc) {
        constexpr int SET = decltype(SETc)::value;
        constexpr bool TAIL = decltype(TAILc)::value;
        const unsigned char* wb = wl[sl % NBUF] + b_base + u * (2 * CO * 16);
        const int rr = 8 * sl + 2 * u + h;
        const unsigned char* ab = patchb + a_base + ((rr / K7) * PYP + rr % K7) * PROW;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int zi = 0; zi < ZPW; ++zi) {
                const unsigned char* ap = ab + pl * PPLANE + zi * ZSTEP;
                u32x4 v;
                v.x = *reinterpret_cast<const unsigned*>(ap);
                v.y = *reinterpret_cast<const unsigned*>(ap + 4);
                v.z = *reinterpret_cast<const unsigned*>(ap + 8);
                v.w = *reinterpret_cast<const unsigned*>(ap + 12) & 0x0000ffffu;     // kx = 7 is padding: exact zero
                if (TAIL && h) v = u32x4{0u, 0u, 0u, 0u};                             // row 49 is padding
                af[SET][zi][pl] = __builtin_bit_cast(bf16x8, v);
            }
            bf0[SET][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * WPL);
            bf1[SET][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * WPL + 32 * 16);
        }
    };
    auto mfmas = [&](auto SETc) {
        constexpr int SET = decltype(SETc)::value;
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int zi = 0; zi < ZPW; ++zi) {
                acc0[zi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET][zi][PA[pr]], bf0[SET][PB[pr]], acc0[zi], 0, 0, 0);
                acc1[zi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET][zi][PA[pr]], bf1[SET][PB[pr]], acc1[zi], 0, 0, 0);
            }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    using NoTail = std::false_type;
    using Tail = std::true_type;
    // one slab = four k-steps; `nextTail`: the k-step read at the end of this slab is the window's last one
    auto slab_body = [&](int sl, auto nextTail) {
        if constexpr (NBUF == 2) {
            // slab sl+1 (in registers since the previous iteration) -> the other buffer (last read during sl-1, before the
            // barrier that ended it); then fetch slab sl+2
            wstore((sl + 1) & 1);
            wload(sl + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        frags(sl, 1, S1{}, NoTail{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(S0{});
        __builtin_amdgcn_sched_barrier(0);
        frags(sl, 2, S0{}, NoTail{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(S1{});
        __builtin_amdgcn_sched_barrier(0);
        frags(sl, 3, S1{}, NoTail{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(S0{});
        __syncthreads();                                   // NBUF 2: slab sl+1 visible; both: slab sl fully read
        if constexpr (NBUF == 1) {
            // slab sl+1 (in registers) over slab sl, under the last k-step's MFMAs; then fetch slab sl+2
            wstore(0);
            wload(sl + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(S1{});
            __syncthreads();                               // slab sl+1 visible
            frags(sl + 1, 0, S0{}, nextTail);
        } else {
            frags(sl + 1, 0, S0{}, nextTail);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(S1{});
        }
    };
    __syncthreads();
    frags(0, 0, S0{}, NoTail{});
    for (int sl = 0; sl < K7 - 2; ++sl) slab_body(sl, NoTail{});        // slabs 0 .. 4 (rows 0 .. 39)
    slab_body(K7 - 2, Tail{});                                          // slab 5, and the read of (slab 6, u = 0)
    __builtin_amdgcn_sched_barrier(0);
    mfmas(S0{});                                                        // row 48 (+ the padding row): the 25th k-step

    // ---- epilogue: C/D layout col = lane & 31 (channel), row = (r&3) + 8*(r>>2) + 4*h (output voxel) ----
    float cs[2][2];                                      // [column tile][sum, sum of squares] of what is stored
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = j * 32 + l32;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int zi = 0; zi < ZPW; ++zi) {
            const int oz = oz0 + ZPW * wave + zi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ox = ox0 + (m & 7), oy = oy0 + (m >> 3);
                const long o = ((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * CO + co;
                float v = j == 0 ? acc0[zi][r] : acc1[zi][r];
                if (p.res) v += p.res[o];
                if (p.relu) v = fmaxf(v, 0.f);
                p.y[o] = v;
                t0 += v; t1 = fmaf(v, v, t1);
            }
        }
        cs[j][0] = t0; cs[j][1] = t1;
    }
    // BatchNorm statistics of the following layer from the tile that is still in registers: the 128 voxels of the
    // workgroup per channel (16 ZPW per lane in f32, then doubles in a fixed order), one partial per workgroup stored
    // [stat][channel][workgroup] so that the finalize reads a channel's partials as one contiguous run
    if (p.stats) {
        float* red = reinterpret_cast<float*>(patchb);   // (no LDS read is left behind the loop's last barrier)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float o = cs[j][q] + __shfl_xor(cs[j][q], 32);
                if (h == 0) red[(wave * 2 + q) * CO + j * 32 + l32] = o;
            }
        __syncthreads();
        if (tid < 2 * CO) {
            const int q = tid >> 6, c = tid & (CO - 1);
            double a = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += (double)red[(w * 2 + q) * CO + c];
            p.stats[(long)(q * CO + c) * gridDim.x + blockIdx.x] = a;
        }
    }
}

// per-channel sums from the per-workgroup partials of stem_fwd_bf3_kernel ([2*64][n_part], one workgroup per entry):
// every thread a fixed stride of the run, then a fixed-shape tree - deterministic
__global__ __launch_bounds__(256) void stem_stats_finalize_kernel(const double* part, int n_part, double* sums) {
    __shared__ double red[256];
    const double* src = part + (long)blockIdx.x * n_part;
    double a = 0;
    for (int i = threadIdx.x; i < n_part; i += 256) a += src[i];
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------------
// WGRAD: dW[tap][co] = sum over output voxels of patch(voxel, tap) * dy[voxel][co].  GEMM rows = taps (343 -> 11
// tiles of 32), columns = 64 channels, reduction = voxels.  A workgroup walks `tiles_per_block` output tiles
// (8 x 4 x 4 voxels each): the tile's input patch and its dy rows go to LDS, the A fragment of (tap, voxel) is
// patch[off(tap) + base(voxel)] = lane register + immediate, again one ds_read_b32.  Wave w owns tap tiles
// w, w+4, w+8 (x 2 column tiles = 6 accumulators), so every A / B fragment read feeds 2 / 3 MFMAs.  The next
// tile's global loads are in flight while the current one is contracted.  Each workgroup writes its partial dW as
// a slab; a second kernel sums the slabs in slab order (deterministic).
constexpr int NTAP = K7 * K7 * K7;                       // 343
constexpr int TAPT = (NTAP + 31) / 32;                   // 11 tap tiles
constexpr int VOX = TX * TY * TZ;                        // 128 voxels per tile

struct StemWgradParams {
    const float* x;      // (N, D, H, W)
    const float* dy;     // (N, Do, Ho, Wo, 64)
    float* slabs;        // [gridDim.x][352][64]
    int N, D, H, W, Do, Ho, Wo;
    int n_tiles, tiles_per_block;
    unsigned x_bytes, dy_bytes;
};

__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgradParams p) {
    constexpr int NP = (PATCH + PW + 255) / 256;         // 16
    constexpr int ND = VOX * (CO / 4) / 256;             // 8 float4 of dy per thread
    __shared__ __attribute__((aligned(16))) float patch[NP * 256];
    __shared__ __attribute__((aligned(16))) float dyl[VOX * CO];      // [voxel][64], odd voxels rotated by 32 columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const __amdgpu_buffer_rsrc_t xr = rsrc(p.x, p.x_bytes), dr = rsrc(p.dy, p.dy_bytes);
    const int txn = p.Wo / TX, tyn = p.Ho / TY, tzn = p.Do / TZ;

    float pv[NP];
    float4 dv[ND];
    auto gload = [&](int tile) {                          // tile >= n_tiles: everything reads as zero
        int b = tile;
        const bool live = tile < p.n_tiles;
        const int bx = b % txn; b /= txn;
        const int by = b % tyn; b /= tyn;
        const int bz = b % tzn;
        const int n = b / tzn;
        const int ox0 = bx * TX, oy0 = by * TY, oz0 = bz * TZ;
        const int iz0 = oz0 * S2 - P3, iy0 = oy0 * S2 - P3, ix0 = ox0 * S2 - P3;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            const int px = i % PW, t = i / PW, py = t % PY, pz = t / PY;
            const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
            const bool ok = live & (px < PX) & (pz < PZ) & ((unsigned)iz < (unsigned)p.D) &
                            ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = ok ? 4u * (unsigned)((((long)n * p.D + iz) * p.H + iy) * p.W + ix) : 0x80000000u;
            pv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            const int v = q / (CO / 4), c = (q % (CO / 4)) * 4;
            const int ox = ox0 + (v & 7), oy = oy0 + ((v >> 3) & 3), oz = oz0 + (v >> 5);
            const unsigned off = live ? 4u * (unsigned)(((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * CO + c)
                                      : 0x80000000u;
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(dr, (int)off, 0, 0);
            dv[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < NP; ++u) patch[tid + u * 256] = pv[u];
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            const int v = q / (CO / 4), c = (q % (CO / 4)) * 4;
            *reinterpret_cast<float4*>(&dyl[v * CO + ((c + 32 * (v & 1)) & 63)]) = dv[u];
        }
    };

    // lane constants: tap offsets of this wave's tap tiles (+ 2h: voxel 2t+h sits 2 floats further along x)
    constexpr int MYT = 3;
    int aoff[MYT];
    bool avalid[MYT];
#pragma unroll
    for (int i = 0; i < MYT; ++i) {
        const int tt = wave + 4 * i;
        const int tap = tt * 32 + l32;
        avalid[i] = (tt < TAPT) && (tap < NTAP);
        const int tc = avalid[i] ? tap : 0;
        const int kz = tc / (K7 * K7), ky = (tc / K7) % K7, kx = tc % K7;
        aoff[i] = (kz * PY + ky) * PW + kx + S2 * h;
    }
    const int bl0 = h ? CO + 32 + l32 : l32;              // dy columns  0..31 of voxel 2t + h
    const int bl1 = h ? CO + l32 : 32 + l32;              // dy columns 32..63

    f32x16 acc[MYT][2];
#pragma unroll
    for (int i = 0; i < MYT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int tile0 = blockIdx.x * p.tiles_per_block;
    gload(tile0);
    for (int it = 0; it < p.tiles_per_block; ++it) {
        __syncthreads();                                  // every wave is done with the previous tile's LDS
        lstore();
        __syncthreads();
        gload(tile0 + it + 1 < tile0 + p.tiles_per_block ? tile0 + it + 1 : p.n_tiles);   // prefetch (zeros past the end)
#pragma unroll
        for (int t = 0; t < VOX / 2; ++t) {
            const int v0 = 2 * t;
            const int imm = ((S2 * (v0 >> 5)) * PY + S2 * ((v0 >> 3) & 3)) * PW + S2 * (v0 & 7);
            const float b0 = dyl[v0 * CO + bl0], b1 = dyl[v0 * CO + bl1];
#pragma unroll
            for (int i = 0; i < MYT; ++i) {
                float a = patch[aoff[i] + imm];
                // third tap tile: tile 10 holds taps 320..351 (343.. are padding), wave 3's would be tile 11
                // (all padding - kept as a zero contribution so that the loop has no wave-dependent branch)
                if (i == MYT - 1) a = avalid[i] ? a : 0.f;
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[i][1], 0, 0, 0);
            }
        }
    }
    // ---- slab: rows = taps (C/D layout row = (r&3) + 8*(r>>2) + 4*h), cols = channels ----
    float* slab = p.slabs + (long)blockIdx.x * (TAPT * 32) * CO;
#pragma unroll
    for (int i = 0; i < MYT; ++i) {
        const int tt = wave + 4 * i;
        if (tt >= TAPT) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tap = tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(long)tap * CO + j * 32 + l32] = acc[i][j][r];
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// WGRAD on the bf16 matrix pipe (round 3), f32-equivalent arithmetic as everywhere (three-way exact bf16 cut of both
// operands, six products, f32 accumulate).  Same decomposition as stem_wgrad_kernel - rows = taps, columns = channels,
// reduction = the 128 output voxels of a tile, three tap tiles x two column tiles per wave - but one
// v_mfma_f32_32x32x16_bf16 contracts 16 voxels: k-step ks = the 8 x-neighbours of the two output rows oy = 2 (ks & 1) + h.
//   * A fragment (lane = tap, 8 voxels along x): x[2 ox + kx], ox = 0..7 - a stride-2 run of the patch row.  The patch rows
//     are stored DE-INTERLEAVED (12 even-x entries, then 12 odd-x entries, bf16), so the run is 8 consecutive entries
//     starting at kx >> 1 of the half kx & 1: five dword reads at lane base + immediate and, for odd starts, four
//     v_alignbit shifts by 16 (the start differs per lane: the lane IS the tap).
//   * B fragment (lane = channel, 8 voxels): dy is staged [voxel][32 channels] in bf16 rows and read through the transposing
//     LDS read, as in direct3_wgrad_kernel.
// Both operands are cut once per tile while they are staged (the patch by x quads: an even and an odd pair per thread).
// 288 MFMAs of 32 cycles per tile and wave against 384 of 64 cycles; 72 KB of LDS (two workgroups per CU).
constexpr int WG_PPLANE = PZ * PY * PROW;                // 8112 bytes: one bf16 plane of the de-interleaved patch
constexpr int WG_QROW = PW / 4;                          // 6 x-quads per patch row
constexpr int WG_NQ = PZ * PY * WG_QROW;                 // 1014 quads
constexpr int WG_NQT = (WG_NQ + 255) / 256;              // 4 per thread
constexpr int WG_DROW = 64;                              // bytes of a (voxel, 32 channels) bf16 row
constexpr int WG_DHALF = VOX * WG_DROW;                  // 8192
constexpr int WG_DPLANE = 2 * WG_DHALF;                  // 16384
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4s lds_bf16x4s;

template <int OCC>
__global__ __launch_bounds__(256, OCC) void stem_wgrad_bf3_kernel(StemWgradParams p) {
    constexpr int ND = VOX * (CO / 8) / 256;             // 4 (voxel, 8 channels) units of dy per thread
    __shared__ __attribute__((aligned(16))) unsigned char patchb[3 * WG_PPLANE];
    __shared__ __attribute__((aligned(16))) unsigned char dyb[3 * WG_DPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
    const __amdgpu_buffer_rsrc_t xr = rsrc(p.x, p.x_bytes), dr = rsrc(p.dy, p.dy_bytes);
    const int txn = p.Wo / TX, tyn = p.Ho / TY, tzn = p.Do / TZ;

    float pv[WG_NQT][4];
    u32x4 dv[ND][2];
    // (the patch of the next tile is fetched in front of the current tile's MFMAs, its dy rows behind them: with both in
    // registers across the k-steps the 256-register budget of two workgroups per CU spills)
    auto gload_patch = [&](int tile) {                    // tile >= n_tiles: everything reads as zero
        int b = tile;
        const bool live = tile < p.n_tiles;
        const int bx = b % txn; b /= txn;
        const int by = b % tyn; b /= tyn;
        const int bz = b % tzn;
        const int n = b / tzn;
        const int ox0 = bx * TX, oy0 = by * TY, oz0 = bz * TZ;
        const int iz0 = oz0 * S2 - P3, iy0 = oy0 * S2 - P3, ix0 = ox0 * S2 - P3;
#pragma unroll
        for (int u = 0; u < WG_NQT; ++u) {
            const int q = tid + u * 256;
            const int qx = q % WG_QROW, t = q / WG_QROW, py = t % PY, pz = t / PY;
            const int iz = iz0 + pz, iy = iy0 + py;
            const bool rok = live & (pz < PZ) & ((unsigned)iz < (unsigned)p.D) & ((unsigned)iy < (unsigned)p.H);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int px = 4 * qx + e, ix = ix0 + px;
                const bool ok = rok & (px < PX) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = ok ? 4u * (unsigned)((((long)n * p.D + iz) * p.H + iy) * p.W + ix) : 0x80000000u;
                pv[u][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)off, 0, 0));
            }
        }
    };
    auto gload_dy = [&](int tile) {
        int b = tile;
        const bool live = tile < p.n_tiles;
        const int bx = b % txn; b /= txn;
        const int by = b % tyn; b /= tyn;
        const int bz = b % tzn;
        const int n = b / tzn;
        const int ox0 = bx * TX, oy0 = by * TY, oz0 = bz * TZ;
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            const int v = q >> 3, c = (q & 7) * 8;
            const int ox = ox0 + (v & 7), oy = oy0 + ((v >> 3) & 3), oz = oz0 + (v >> 5);
            const unsigned off = live ? 4u * (unsigned)(((((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * CO + c)
                                      : 0x80000000u;
            dv[u][0] = __builtin_amdgcn_raw_buffer_load_b128(dr, (int)off, 0, 0);
            dv[u][1] = __builtin_amdgcn_raw_buffer_load_b128(dr, (int)(off + 16u), 0, 0);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < WG_NQT; ++u) {
            const int q = tid + u * 256;
            if (q < WG_NQ) {
                unsigned c0[4], c1[4], c2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cut3(pv[u][e], c0[e], c1[e], c2[e]);
                // quad q of its row: even pair -> entries 2 qx, 2 qx + 1 of the even half, odd pair -> of the odd half
                unsigned char* d = patchb + (q / WG_QROW) * PROW + 4 * (q % WG_QROW);
                *reinterpret_cast<unsigned*>(d) = c0[0] | (c0[2] << 16);
                *reinterpret_cast<unsigned*>(d + 24) = c0[1] | (c0[3] << 16);
                *reinterpret_cast<unsigned*>(d + WG_PPLANE) = c1[0] | (c1[2] << 16);
                *reinterpret_cast<unsigned*>(d + WG_PPLANE + 24) = c1[1] | (c1[3] << 16);
                *reinterpret_cast<unsigned*>(d + 2 * WG_PPLANE) = c2[0] | (c2[2] << 16);
                *reinterpret_cast<unsigned*>(d + 2 * WG_PPLANE + 24) = c2[1] | (c2[3] << 16);
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int q = tid + u * 256;
            const int v = q >> 3, cg = q & 7;
            unsigned o0[4], o1[4], o2[4];
            unsigned a0[8], a1[8], a2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cut3(__uint_as_float(dv[u][e >> 2][e & 3]), a0[e], a1[e], a2[e]);
#pragma unroll
            for (int d2 = 0; d2 < 4; ++d2) {
                o0[d2] = a0[2 * d2] | (a0[2 * d2 + 1] << 16);
                o1[d2] = a1[2 * d2] | (a1[2 * d2 + 1] << 16);
                o2[d2] = a2[2 * d2] | (a2[2 * d2 + 1] << 16);
            }
            unsigned char* d = dyb + (cg >> 2) * WG_DHALF + v * WG_DROW + (cg & 3) * 16;
            *reinterpret_cast<u32x4*>(d) = u32x4{o0[0], o0[1], o0[2], o0[3]};
            *reinterpret_cast<u32x4*>(d + WG_DPLANE) = u32x4{o1[0], o1[1], o1[2], o1[3]};
            *reinterpret_cast<u32x4*>(d + 2 * WG_DPLANE) = u32x4{o2[0], o2[1], o2[2], o2[3]};
        }
    };

    // lane constants of this wave's tap tiles: dword-aligned start of the 8-entry run of output row oy = h, and its shift
    constexpr int MYT = 3;
    int abase[MYT];
    unsigned ash[MYT];
    bool avalid[MYT];
#pragma unroll
    for (int i = 0; i < MYT; ++i) {
        const int tt = wave + 4 * i;
        const int tap = tt * 32 + l32;
        avalid[i] = (tt < TAPT) && (tap < NTAP);
        const int tc = avalid[i] ? tap : 0;
        const int kz = tc / (K7 * K7), ky = (tc / K7) % K7, kx = tc % K7;
        const int st = kx >> 1;
        abase[i] = (kz * PY + ky + S2 * h) * PROW + (kx & 1) * 24 + (st >> 1) * 4;
        ash[i] = (unsigned)(st & 1) * 16u;
    }
    // B fragments: transposing read (this lane names row q4 of its 16-lane group's 4-row block; direct3_wgrad_kernel)
    const int q4 = i16 >> 2;
    const int b_base = (8 * h + q4) * WG_DROW + (16 * g16 + 4 * (i16 & 3)) * 2;

    f32x16 acc[MYT][2];
#pragma unroll
    for (int i = 0; i < MYT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    const int tile0 = blockIdx.x * p.tiles_per_block;
    gload_patch(tile0);
    gload_dy(tile0);
    for (int it = 0; it < p.tiles_per_block; ++it) {
        __syncthreads();                                  // every wave is done with the previous tile's LDS
        lstore();
        __syncthreads();
        const int nxt = it + 1 < p.tiles_per_block ? tile0 + it + 1 : p.n_tiles;     // (zeros past the end)
        gload_patch(nxt);
        // software pipeline: the fragments of k-step ks + 1 are read (and shifted) while the MFMAs of k-step ks run
        bf16x8 bf[2][2][3], af[2][MYT][3];
        auto frags = [&](int ks, int SET) {
            const int imm = ((S2 * (ks >> 1)) * PY + 2 * S2 * (ks & 1)) * PROW;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned char* bp = dyb + pl * WG_DPLANE + j * WG_DHALF + b_base + ks * 16 * WG_DROW;
                    const bf16x4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4s*)(bp));
                    const bf16x4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4s*)(bp + 4 * WG_DROW));
                    bf[SET][j][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
            for (int i = 0; i < MYT; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned char* ap = patchb + pl * WG_PPLANE + abase[i] + imm;
                    unsigned d[5];
#pragma unroll
                    for (int e = 0; e < 5; ++e) d[e] = *reinterpret_cast<const unsigned*>(ap + 4 * e);
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_alignbit(d[e + 1], d[e], ash[i]);
                    // third tap tile: taps 343.. of tile 10 (and wave 3's tile 11) are padding
                    if (i == MYT - 1 && !avalid[i]) v = u32x4{0u, 0u, 0u, 0u};
                    af[SET][i][pl] = __builtin_bit_cast(bf16x8, v);
                }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < VOX / 16; ++ks) {
            if (ks + 1 < VOX / 16) frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < MYT; ++i)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i][PA[pr]], bf[ks & 1][0][PB[pr]], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i][PA[pr]], bf[ks & 1][1][PB[pr]], acc[i][1], 0, 0, 0);
                }
        }
        gload_dy(nxt);
    }
    // ---- slab: rows = taps (C/D layout row = (r&3) + 8*(r>>2) + 4*h), cols = channels ----
    float* slab = p.slabs + (long)blockIdx.x * (TAPT * 32) * CO;
#pragma unroll
    for (int i = 0; i < MYT; ++i) {
        const int tt = wave + 4 * i;
        if (tt >= TAPT) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tap = tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(long)tap * CO + j * 32 + l32] = acc[i][j][r];
            }
    }
}

// out[g] = sum of slabs [g*group, (g+1)*group) (slab order); two levels keep every pass wide:
// level 1: grid (22, G) over the workgroup slabs -> G partial slabs, level 2: grid (22, 1) over those -> dW
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* slabs, int n_slabs, int group, float* out,
                                                               long out_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;         // float4 index into [343][64]
    if (i >= NTAP * CO / 4) return;
    const int z0 = blockIdx.y * group, z1 = min(z0 + group, n_slabs);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight loads in flight, added in slab order (the sums are the rolled loop's bit for bit; rolled, every load waited for the one before:
    // 11 us per level for 16-32 slabs, and both levels sit at the very end of the step's backward chain)
    int z = z0;
    for (; z + 8 <= z1; z += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(slabs + (long)(z + u) * (TAPT * 32) * CO + 4 * i);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < z1; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (long)z * (TAPT * 32) * CO + 4 * i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long)blockIdx.y * out_stride + 4 * i) = s;
}

}  // namespace

// Called by conv_igemm.hip's dispatcher.  Returns MI_E_UNSUPPORTED when the shape is not the stem this kernel is
// specialised for (the caller then takes the generic path).
size_t mi_stem7_fwd_workspace_bytes() { return WPREP_BYTES; }
// with the BatchNorm statistics of the output from the epilogue: + one partial per workgroup and statistic
size_t mi_stem7_fwd_stats_workspace_bytes(int N, int D, int H, int W) {
    const int Do = (D + 2 * P3 - K7) / S2 + 1, Ho = (H + 2 * P3 - K7) / S2 + 1, Wo = (W + 2 * P3 - K7) / S2 + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0 || Do % TZ || Ho % TY || Wo % TX) return 0;
    const size_t blocks = (size_t)N * (Do / TZ) * (Ho / TY) * (Wo / TX);
    return ((WPREP_BYTES + 255) & ~(size_t)255) + sizeof(double) * 2 * CO * blocks;
}

// bf16x3 != 0 with a workspace of mi_stem7_fwd_workspace_bytes(): the bf16-pipe kernel (f32-equivalent); otherwise the
// f32 MFMA kernel
int mi_stem7_fwd(const float* x, const float* w, float* y, const float* res, int relu, int N, int D, int H, int W,
                 int Co, int bf16x3, void* ws, size_t ws_bytes, hipStream_t s, double* sums) {
    if (Co != CO) return MI_E_UNSUPPORTED;
    if (sums && (!bf16x3 || !ws || ws_bytes < mi_stem7_fwd_stats_workspace_bytes(N, D, H, W) || ws_bytes == 0))
        return MI_E_UNSUPPORTED;                       // (statistics: the bf16-pipe kernel's epilogue only)
    const int Do = (D + 2 * P3 - K7) / S2 + 1, Ho = (H + 2 * P3 - K7) / S2 + 1, Wo = (W + 2 * P3 - K7) / S2 + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0 || Do % TZ || Ho % TY || Wo % TX) return MI_E_UNSUPPORTED;
    const long xb = 4l * N * D * H * W;
    if (xb >= 0x7fff0000l) return MI_E_UNSUPPORTED;
    const long blocks = (long)N * (Do / TZ) * (Ho / TY) * (Wo / TX);
    if (blocks > 0x7fffffffl) return MI_E_UNSUPPORTED;
    StemParams p = {x, w, y, res, relu, N, D, H, W, Do, Ho, Wo, (unsigned)xb, nullptr};
    if (bf16x3 && ws && ws_bytes >= WPREP_BYTES) {
        if (sums) p.stats = reinterpret_cast<double*>((unsigned char*)ws + ((WPREP_BYTES + 255) & ~(size_t)255));
        hipLaunchKernelGGL(stem_wprep_kernel, dim3((K7 * 8 * CO * 8 + 255) / 256), dim3(256), 0, s, w, (unsigned short*)ws);
        MI_RETURN_IF_LAUNCH_FAILED();
        // one weight buffer / three workgroups per CU by default (captured step 1.632 against 1.646 ms, r03_experiments.txt item 23)
        // round 4: two z-planes per wave (tile 8 x 4 x 8) where the depth allows; MI_STEM_FWD_Z4=1 keeps the 8 x 4 x 4 tile
        const bool z8 = Do % (2 * TZ) == 0 && !getenv("MI_STEM_FWD_Z4") && !getenv("MI_STEM_FWD_NBUF2");
        const long nb = z8 ? blocks / 2 : blocks;
        if (z8) hipLaunchKernelGGL((stem_fwd_bf3_kernel<1, 2>), dim3((unsigned)nb), dim3(256), 0, s, p, (const unsigned char*)ws);
        else if (getenv("MI_STEM_FWD_NBUF2")) hipLaunchKernelGGL((stem_fwd_bf3_kernel<2, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p, (const unsigned char*)ws);
        else hipLaunchKernelGGL((stem_fwd_bf3_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p, (const unsigned char*)ws);
        MI_RETURN_IF_LAUNCH_FAILED();
        if (sums) {
            hipLaunchKernelGGL(stem_stats_finalize_kernel, dim3(2 * CO), dim3(256), 0, s, (const double*)p.stats, (int)nb,
                               sums);
            MI_RETURN_IF_LAUNCH_FAILED();
        }
        return MI_OK;
    }
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

constexpr int RED_GROUPS = 32;          // partial slabs of the two-level reduction

static int stem7_wgrad_blocks(long tiles) {
    // 2 workgroups per CU; each walks >= 1 tile
    long b = tiles < 512 ? tiles : 512;
    return (int)b;
}

size_t mi_stem7_wgrad_workspace_bytes(int N, int D, int H, int W, int Co) {
    if (Co != CO) return 0;
    const int Do = (D + 2 * P3 - K7) / S2 + 1, Ho = (H + 2 * P3 - K7) / S2 + 1, Wo = (W + 2 * P3 - K7) / S2 + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0 || Do % TZ || Ho % TY || Wo % TX) return 0;
    const long tiles = (long)N * (Do / TZ) * (Ho / TY) * (Wo / TX);
    return sizeof(float) * (size_t)(stem7_wgrad_blocks(tiles) + RED_GROUPS) * (TAPT * 32) * CO;
}

int mi_stem7_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Co, int bf16x3, void* ws,
                   size_t ws_bytes, hipStream_t s) {
    if (Co != CO) return MI_E_UNSUPPORTED;
    const int Do = (D + 2 * P3 - K7) / S2 + 1, Ho = (H + 2 * P3 - K7) / S2 + 1, Wo = (W + 2 * P3 - K7) / S2 + 1;
    if (Do <= 0 || Ho <= 0 || Wo <= 0 || Do % TZ || Ho % TY || Wo % TX) return MI_E_UNSUPPORTED;
    const long xb = 4l * N * D * H * W, yb = 4l * N * Do * Ho * Wo * CO;
    if (xb >= 0x7fff0000l || yb >= 0x7fff0000l) return MI_E_UNSUPPORTED;
    const long tiles = (long)N * (Do / TZ) * (Ho / TY) * (Wo / TX);
    if (tiles > 0x3fffffffl) return MI_E_UNSUPPORTED;
    const int blocks = stem7_wgrad_blocks(tiles);
    if (!ws || ws_bytes < mi_stem7_wgrad_workspace_bytes(N, D, H, W, Co)) return MI_E_UNSUPPORTED;   // generic path
    StemWgradParams p = {x, dy, (float*)ws, N, D, H, W, Do, Ho, Wo, (int)tiles, (int)((tiles + blocks - 1) / blocks),
                         (unsigned)xb, (unsigned)yb};
    const char* occ = getenv("MI_STEM_WGRAD_OCC");
    if (bf16x3 && occ && atoi(occ) == 1) hipLaunchKernelGGL(stem_wgrad_bf3_kernel<1>, dim3(blocks), dim3(256), 0, s, p);
    else if (bf16x3) hipLaunchKernelGGL(stem_wgrad_bf3_kernel<2>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(stem_wgrad_kernel, dim3(blocks), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    const int rb = (NTAP * CO / 4 + 255) / 256;
    const long slab = (long)(TAPT * 32) * CO;
    float* part = (float*)ws + (long)blocks * slab;
    const int group = (blocks + RED_GROUPS - 1) / RED_GROUPS, groups = (blocks + group - 1) / group;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(rb, groups), dim3(256), 0, s, (const float*)ws, blocks, group, part,
                       slab);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(rb, 1), dim3(256), 0, s, (const float*)part, groups, groups, dw, 0l);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
