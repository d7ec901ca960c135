// Direct 3x3x3 / stride 1 / padding 1 convolution for 64 -> 64 channels on (N, D, 8, 8, 64) channels-last activations,
// forward and data-gradient form, in the bf16x3 arithmetic of conv_igemm.hip (three-way exact bf16 cut of both operands,
// six bf16 MFMA products per f32 product, f32 accumulate).  This is layer1 of the MoCo-3D encoder
// (cet_pick/models/networks/moco_encoder_3d.py:55-84,170 - four such convolutions per encoder pass): 12 of the 75 conv
// launches of a training step and the largest share of its time.
//
// Why not the implicit GEMM: with 64 output channels a 64 x 64 tile re-gathers, re-cuts and re-stores its im2col rows
// for every one of the 27 taps (the matrix pipe is busy 14 % of the time, profiles/r02_mfma_busy.json: the waves
// spend their issue slots on the cut, the gather addresses and LDS traffic).  Here the INPUT PATCH of a tile is cut once
// and stays in LDS for all 27 taps:
//   * one 256-thread workgroup owns two z-planes of one sample (128 output voxels x 64 channels); its patch is the four
//     z-planes around them (256 voxels x 64 channels, three bf16 planes = 96 KB + zero regions = 115 KB of LDS with every
//     16-channel chunk resident; round 3: a ring of TWO chunks, 59 KB, so that two workgroups share a CU), planes
//     outside the volume zero-filled.  The 8 x 8 plane is the whole image, so the x / y halo is pure padding: a lane
//     whose neighbour falls outside reads a zero record instead (one address select per tap and row block, no data
//     select).
//   * patch layout: per (k-step of 16 channels, bf16 plane, k-half) an array of 16-byte records, one per voxel - an
//     MFMA A-fragment (row = voxel, 8 consecutive channels) is ONE ds_read_b128 at lane base + immediate; consecutive
//     voxels are consecutive records, so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots.
//   * the weights never pass through LDS: a prep kernel cuts them once into an image in MFMA B-fragment order
//     ([channel chunk][tap][column half][bf16 plane][lane] x 16 bytes), and every wave streams its 1-KB fragments straight
//     from L2 into registers, six k-steps ahead of their use (the two waves that share a column half hit in the CU's L1).
//   * the reduction runs channel chunk (16 channels) outermost: chunk 0 of the patch is staged before the loop, chunk
//     c + 1 is loaded, cut and stored in the shadow of chunk c's 27 x 12 MFMAs (three workgroup barriers in all).  Per
//     k-step and wave: 6 ds_read_b128, 3 buffer loads and 12 MFMAs (wave tile 64 voxels x 32 channels) - no gather, and
//     a cut of 4 elements per thread and k-step instead of 16.
// DGRAD is the same kernel on dY with the weight image built transposed and tap-flipped:
//   dX[i] = sum_t dY[i + 1 - t] W[t]^T = sum_t' dY[i + t' - 1] W[2 - t']^T.
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 64;                       // channels, in = out
constexpr int PLANE = 64;                   // voxels of an 8 x 8 z-plane
constexpr int TZ = 2;                       // output z-planes per workgroup
constexpr int PZ = TZ + 2;                  // patch z-planes
constexpr int NV = PZ * PLANE;              // 256 patch voxels
constexpr int LEAD = 16;                    // records in front of voxel 0 (lane bases carry the -9 of the tap offsets)
constexpr int ZREC = 34;                    // zero records behind the voxels: (record mod 16) + tap offset (0..18) stays inside
constexpr int NREC = LEAD + NV + ZREC;      // 306
constexpr int ARR = NREC * 16;              // bytes of one (k-step, plane, k-half) array
constexpr int KS = C / 16;                  // bf16 MFMA k-steps per tap = channel chunks of the patch
constexpr int PL_BYTES = 2 * ARR;
constexpr int KS_BYTES = 3 * PL_BYTES;
constexpr int LDS_BYTES = KS * KS_BYTES;    // 117,504
constexpr int ZBASE = (LEAD + NV) * 16;     // byte offset of the zero region in an array: record 272 = 0 (mod 16)
constexpr int WBLK = 1024;                  // one B fragment: 64 lanes x 16 bytes
constexpr int WSTEP = 2 * 3 * WBLK;         // bytes per k-step: [column half][plane]
constexpr int NTAP = 27;
constexpr int NSTEP = KS * NTAP;            // 108 k-steps, chunk-major: g = chunk * 27 + tap
constexpr int WIMG_BYTES = NSTEP * WSTEP;   // 663,552
constexpr int RB = 6;                       // weight fragments in flight: k-steps ahead (9 and 12 measured: no faster)
static_assert(NSTEP % RB == 0, "ring slots");
static_assert(LDS_BYTES <= 160 * 1024, "patch must fit the CU's LDS");
static_assert(((LEAD + NV) & 15) == 0, "zero region must start at a record = 0 (mod 16)");

struct Direct3Params {
    const float* a;           // X (forward) or dY (data gradient): (N, D, 8, 8, 64)
    const unsigned char* wimg;
    float* out;
    const float* res;         // out = act(acc + res)          (may be null)
    const float* mask;        // out *= (mask > 0)             (may be null)
    int relu;
    int N, D;
    unsigned a_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// exact three-way bf16 cut of 8 consecutive f32 (truncation keeps every step exact, conv_igemm.hip): planes o[0..2],
// element e in half (e & 1) of dword e / 2
__device__ __forceinline__ void cut8(const float (&v)[8], u32x4 (&o)[3]) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        u0[t] = __float_as_uint(v[t]);
        const float r1 = v[t] - __uint_as_float(u0[t] & 0xffff0000u);
        u1[t] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[t] & 0xffff0000u);
        u2[t] = __float_as_uint(r2);
    }
    constexpr unsigned HI2 = 0x07060302u;   // v_perm_b32: high halves of two dwords -> one dword (first element low)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        o[0][d] = __builtin_amdgcn_perm(u0[2 * d + 1], u0[2 * d], HI2);
        o[1][d] = __builtin_amdgcn_perm(u1[2 * d + 1], u1[2 * d], HI2);
        o[2][d] = __builtin_amdgcn_perm(u2[2 * d + 1], u2[2 * d], HI2);
    }
}

// (Measured and rejected: eight waves per workgroup - two quartets splitting the reduction over the same tile, two waves
// per SIMD, partial sums exchanged through LDS at the end - 36.9 us against 32.3 us for this form.)
// TWO: only two channel chunks of the patch are resident (a ring of two slots: 59 KB), the residual / mask operands of the
// epilogue are fetched IN the epilogue instead of during the last chunk (64 registers less), so that two workgroups share a CU
// (256 registers each) and one's prologue / epilogue runs under the other's MFMA loop.
// CT: channels, in = out: 64 (layer1 at 32^3 crops) or - round 4 - 128 (layer2 at 64^3 crops: the same 8 x 8 planes, eight
// 16-channel chunks of the patch through the ring, a workgroup per 64-channel block of the output: blockIdx.y).
template <bool TWO, int CT = C>
__global__ __launch_bounds__(256, TWO ? 2 : 1) void direct3_kernel(Direct3Params p) {
    constexpr int KS = CT / 16, NSTEP = KS * NTAP;       // (shadow the 64-channel file constants)
    constexpr int NSLOT = TWO ? 2 : KS;
    static_assert(NSTEP % RB == 0, "ring slots");
    __shared__ __attribute__((aligned(16))) unsigned char patch[NSLOT * KS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int tz = wave >> 1, cw = wave & 1;             // wave tile: z-plane tz of the pair x column half cw
    const int zb = blockIdx.x % (p.D / TZ), n = blockIdx.x / (p.D / TZ);
    const int z0 = zb * TZ;

    const int cb = blockIdx.y;                           // 64-channel block of the output channels
    const __amdgpu_buffer_rsrc_t wrs = rsrc_of(p.wimg, (CT / 64) * NSTEP * WSTEP);
    const int w_voff = cw * (3 * WBLK) + lane * 16;
    const int w_cb = cb * (NSTEP * WSTEP);
    // weight fragments of k-step g (0..107; behind the image: zeros) -> ring slot g % RB
    bf16x8 bfr[RB][3];
    auto wload = [&](int g, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bfr[SLOT][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, g < NSTEP ? w_voff + pl * WBLK : (int)0x80000000u, g < NSTEP ? w_cb + g * WSTEP : 0, 0));
    };
    auto wload_dyn = [&](int g) {                        // (g is a constant after unrolling)
        switch (g % RB) {
            case 0: wload(g, std::integral_constant<int, 0>{}); break;
            case 1: wload(g, std::integral_constant<int, 1>{}); break;
            case 2: wload(g, std::integral_constant<int, 2>{}); break;
            case 3: wload(g, std::integral_constant<int, 3>{}); break;
            case 4: wload(g, std::integral_constant<int, 4>{}); break;
            default: wload(g, std::integral_constant<int, 5>{}); break;
        }
    };
    // the first RB - 1 k-steps of weights go out before anything else
#pragma unroll
    for (int g = 0; g < RB - 1; ++g) wload_dyn(g);

    // ---- patch staging, one 16-channel chunk at a time: unit q = (voxel, k-half); 2 units per thread and chunk.
    // Chunk 0 is staged here; chunk c + 1 is loaded and cut in the shadow of chunk c's MFMAs.
    const __amdgpu_buffer_rsrc_t ars = rsrc_of(p.a, p.a_bytes);
    unsigned st_off[2];
    int st_lds[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = tid + 256 * u, vox = q >> 1, hh = q & 1;
        const int z = z0 - 1 + (vox >> 6);
        const bool ok = (unsigned)z < (unsigned)p.D;      // planes outside the volume: zeros (offset out of range)
        st_off[u] = ok ? 4u * (unsigned)((((long)n * p.D + z) * PLANE + (vox & 63)) * CT + hh * 8) : 0x80000000u;
        st_lds[u] = hh * ARR + (LEAD + vox) * 16;
    }
    u32x4 ld[2][2];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(st_off[u] + 64u * c), 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(st_off[u] + 64u * c + 16u), 0, 0);
        }
    };
    auto stage_store = [&](int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            cut8(v, o);
            unsigned char* dst = patch + (c % NSLOT) * KS_BYTES + st_lds[u];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * PL_BYTES) = o[pl];
        }
    };
    stage_load(0);
    // zero regions of the 24 arrays (the leading records are never read)
    for (int i = tid; i < NSLOT * 3 * 2 * ZREC; i += 256) {
        const int arr = i / ZREC, r = i % ZREC;
        *reinterpret_cast<u32x4*>(patch + arr * ARR + ZBASE + r * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- per-lane geometry: row block i (rows 32 i .. 32 i + 31 of the wave's z-plane), MFMA row l32 ----
    // vbase = record (LEAD + voxel - 9) of the lane's output voxel in patch plane tz; a tap adds (dz 64 + dy 8 + dx) records.
    // zbase = the zero record with the same (record mod 16): a lane whose neighbour is padding keeps its LDS slot, so the
    // 16 lanes of a ds_read_b128 group stay on 16 distinct slots (a shared zero record cost 46 % extra LDS cycles).
    int vbase[2], zbase[2];
    unsigned vmask[2];                  // bit dy * 3 + dx: the (dy, dx) neighbour is inside the 8 x 8 plane
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int y = 4 * i + (l32 >> 3), x = l32 & 7;
        const int rec = LEAD - 9 + tz * PLANE + y * 8 + x;
        vbase[i] = rec * 16 + h * ARR;
        zbase[i] = ZBASE + (rec & 15) * 16 + h * ARR;
        unsigned m = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if ((unsigned)(y + dy - 1) < 8u && (unsigned)(x + dx - 1) < 8u) m |= 1u << (dy * 3 + dx);
        vmask[i] = m;
    }

    f32x16 acc[2];                      // (two chains per row block measured: no faster)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first

    // epilogue operands (residual, mask), fetched during the last chunk; a null pointer reads zeros (empty descriptor)
    const long m0 = ((long)n * p.D + z0 + tz) * PLANE;
    const int col = cb * 64 + cw * 32 + l32;
    unsigned eoff[2][16];
    float rv[2][16], mv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) eoff[i][r] = 4u * (unsigned)((m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * CT + col);
    const __amdgpu_buffer_rsrc_t rrs = rsrc_of(p.res, p.res ? p.a_bytes : 0u), mrs = rsrc_of(p.mask, p.mask ? p.a_bytes : 0u);

    stage_store(0);
    stage_load(1);
    __syncthreads();                    // chunk 0 and the zero regions are in place

    bf16x8 af[2][2][3];
    int sel[9][2];
    auto make_sel = [&](int dz) {       // address of the (dy, dx) neighbour's record for this dz, or the lane's zero record
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int i = 0; i < 2; ++i) sel[t9][i] = ((vmask[i] >> t9) & 1u) ? vbase[i] + dz * (PLANE * 16) : zbase[i];
    };
    auto frags = [&](int c, int t9, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int imm = ((t9 / 3) * 8 + (t9 % 3)) * 16 + (c % NSLOT) * KS_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[SET][i][pl] = *reinterpret_cast<const bf16x8*>(patch + sel[t9][i] + imm + pl * PL_BYTES);
    };
    auto frags_dyn = [&](int g, int c, int t9) {
        if (g & 1) frags(c, t9, std::integral_constant<int, 1>{});
        else frags(c, t9, std::integral_constant<int, 0>{});
    };

    make_sel(0);
    frags_dyn(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < KS; ++c) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int g = c * NTAP + dz * 9 + t9;
                // weights of k-step g + RB - 1 go into the slot k-step g - 1 used
                wload_dyn(g + RB - 1);
                // the next chunk of the patch: cut + stored a third into this chunk's taps (its loads have been in flight
                // since the previous chunk), the loads of the chunk after it right behind
                if (dz == 1 && t9 == 0 && c + 1 < KS) {
                    stage_store(c + 1);
                    if (c + 2 < KS) stage_load(c + 2);
                }
                if (!TWO && c == KS - 1 && dz == 0 && t9 == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            rv[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, (int)eoff[i][r], 0, 0));
                            mv[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)eoff[i][r], 0, 0));
                        }
                }
                // A fragments of the next k-step into the other register set
                if (g + 1 < NSTEP) {
                    const int c2 = (g + 1) / NTAP, r2 = (g + 1) % NTAP;
                    if (r2 == 0) __syncthreads();            // next chunk of the patch visible (stored >= 17 k-steps ago)
                    if (r2 % 9 == 0) make_sel(r2 / 9);
                    frags_dyn(g + 1, c2, r2 % 9);
                }
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][0][PA[pr]], bfr[g % RB][PB[pr]], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][1][PA[pr]], bfr[g % RB][PB[pr]], acc[1], 0, 0, 0);
                }
                // issue order inside the k-step: every load behind an MFMA (the matrix pipe keeps running while the
                // LDS / buffer instructions issue); all nine in front of the first MFMA cost 3 us per launch
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one buffer load
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h ----
    const bool has_mask = p.mask != nullptr;
    if constexpr (TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float rr[16], mm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned eo = 4u * (unsigned)((m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * CT + col);
                rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, (int)eo, 0, 0));
                mm[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)eo, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned eo = 4u * (unsigned)((m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * CT + col);
                float v = acc[i][r] + rr[r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (has_mask) v = (mm[r] > 0.f) ? v : 0.f;
                p.out[eo / 4] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[i][r] + rv[i][r];
            if (p.relu) v = fmaxf(v, 0.f);
            if (has_mask) v = (mv[i][r] > 0.f) ? v : 0.f;
            p.out[eoff[i][r] / 4] = v;
        }
}

// ---- the same convolution on planes LARGER than 8 x 8 (round 4: layer1 of 64^3 crops, 64 -> 64 channels on 16 x 16 x 16) ---------
// A workgroup owns an 8 x 8 (y, x) tile of two z-planes; its patch is the tile with a REAL halo - 4 planes x 10 x 10 voxels, one
// record per voxel at pitch 10, voxels outside the volume zero-filled while the patch is staged - so a tap is a plain record
// offset (dz 100 + dy 10 + dx): no zero records, no address select.  Everything else is direct3_kernel<true>: two resident
// 16-channel chunks (77 KB of LDS, two workgroups per CU), weight fragments streamed from the SAME pre-cut image (format of
// kind 1), 6 ds_read_b128 + 3 buffer loads + 12 MFMAs per k-step and wave.  (Pitch 10: the 16 lanes of a ds_read_b128 group
// cover records r .. r + 7 and r + 10 .. r + 17 - two of the sixteen 16-byte slots are hit twice; every other pitch >= 10 is worse.)
constexpr int HP = 10;                      // patch pitch: 8 + 2 halo voxels
constexpr int H_NV = PZ * HP * HP;          // 400 patch voxels
constexpr int H_ARR = H_NV * 16;            // bytes of one (chunk, plane, k-half) array
constexpr int H_PL = 2 * H_ARR;
constexpr int H_KS = 3 * H_PL;              // one chunk: 38,400 bytes
constexpr int H_UNITS = (2 * H_NV + 255) / 256;      // staging units (voxel, k-half) per thread and chunk: 4

struct Direct3hParams {
    const float* a;           // X (forward) or dY (data gradient): (N, D, H, W, 64)
    const unsigned char* wimg;
    float* out;
    const float* res;
    const float* mask;
    int relu;
    int N, D, H, W;
    unsigned a_bytes;
};

__global__ __launch_bounds__(256, 2) void direct3h_kernel(Direct3hParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[2 * H_KS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int tz = wave >> 1, cw = wave & 1;             // wave tile: z-plane tz of the pair x column half cw
    const int txn = (p.W + 7) / 8, tyn = (p.H + 7) / 8;       // (round 5: ragged planes - the last tile of a row / column hangs over)
    int bi = blockIdx.x;
    const int bx = bi % txn; bi /= txn;
    const int by = bi % tyn; bi /= tyn;
    const int zb = bi % (p.D / TZ), n = bi / (p.D / TZ);
    const int z0 = zb * TZ, y0 = by * 8, x0 = bx * 8;

    const __amdgpu_buffer_rsrc_t wrs = rsrc_of(p.wimg, WIMG_BYTES);
    const int w_voff = cw * (3 * WBLK) + lane * 16;
    bf16x8 bfr[RB][3];
    auto wload = [&](int g, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bfr[SLOT][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, g < NSTEP ? w_voff + pl * WBLK : (int)0x80000000u, g < NSTEP ? g * WSTEP : 0, 0));
    };
    auto wload_dyn = [&](int g) {
        switch (g % RB) {
            case 0: wload(g, std::integral_constant<int, 0>{}); break;
            case 1: wload(g, std::integral_constant<int, 1>{}); break;
            case 2: wload(g, std::integral_constant<int, 2>{}); break;
            case 3: wload(g, std::integral_constant<int, 3>{}); break;
            case 4: wload(g, std::integral_constant<int, 4>{}); break;
            default: wload(g, std::integral_constant<int, 5>{}); break;
        }
    };
#pragma unroll
    for (int g = 0; g < RB - 1; ++g) wload_dyn(g);

    // ---- patch staging, one 16-channel chunk at a time: unit q = (patch voxel, k-half); voxels outside the volume read zeros ----
    const __amdgpu_buffer_rsrc_t ars = rsrc_of(p.a, p.a_bytes);
    unsigned st_off[H_UNITS];
    int st_lds[H_UNITS];
#pragma unroll
    for (int u = 0; u < H_UNITS; ++u) {
        const int q = tid + 256 * u, vox = q >> 1, hh = q & 1;
        const int pz = vox / (HP * HP), py = (vox / HP) % HP, px = vox % HP;
        const int z = z0 - 1 + pz, y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = q < 2 * H_NV && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        st_off[u] = ok ? 4u * (unsigned)(((((long)n * p.D + z) * p.H + y) * p.W + x) * C + hh * 8) : 0x80000000u;
        st_lds[u] = q < 2 * H_NV ? hh * H_ARR + vox * 16 : -1;
    }
    u32x4 ld[H_UNITS][2];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int u = 0; u < H_UNITS; ++u) {
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(st_off[u] + 64u * c), 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(st_off[u] + 64u * c + 16u), 0, 0);
        }
    };
    auto stage_store = [&](int c) {
#pragma unroll
        for (int u = 0; u < H_UNITS; ++u) {
            if (st_lds[u] < 0) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            cut8(v, o);
            unsigned char* dst = patch + (c & 1) * H_KS + st_lds[u];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * H_PL) = o[pl];
        }
    };
    stage_load(0);

    // ---- per-lane geometry: row block i = tile rows 4 i .. 4 i + 3, MFMA row l32 = (y & 3, x); record of the (dz, dy, dx) = 0 corner ----
    int vbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int y = 4 * i + (l32 >> 3), x = l32 & 7;
        vbase[i] = ((tz * HP + y) * HP + x) * 16 + h * H_ARR;
    }

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    const int col = cw * 32 + l32;
    const __amdgpu_buffer_rsrc_t rrs = rsrc_of(p.res, p.res ? p.a_bytes : 0u), mrs = rsrc_of(p.mask, p.mask ? p.a_bytes : 0u);

    stage_store(0);
    stage_load(1);
    __syncthreads();

    bf16x8 af[2][2][3];
    auto frags = [&](int c, int tap, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int imm = (((tap / 9) * HP + (tap / 3) % 3) * HP + tap % 3) * 16 + (c & 1) * H_KS;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[SET][i][pl] = *reinterpret_cast<const bf16x8*>(patch + vbase[i] + imm + pl * H_PL);
    };
    auto frags_dyn = [&](int g, int c, int tap) {
        if (g & 1) frags(c, tap, std::integral_constant<int, 1>{});
        else frags(c, tap, std::integral_constant<int, 0>{});
    };

    frags_dyn(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < KS; ++c) {
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int g = c * NTAP + tap;
            wload_dyn(g + RB - 1);
            if (tap == 9 && c + 1 < KS) {
                stage_store(c + 1);
                if (c + 2 < KS) stage_load(c + 2);
            }
            if (g + 1 < NSTEP) {
                const int c2 = (g + 1) / NTAP, r2 = (g + 1) % NTAP;
                if (r2 == 0) __syncthreads();            // next chunk of the patch visible (stored 17 k-steps ago)
                frags_dyn(g + 1, c2, r2);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][0][PA[pr]], bfr[g % RB][PB[pr]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][1][PA[pr]], bfr[g % RB][PB[pr]], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h = (y & 3, x) of row block i ----
    const bool has_mask = p.mask != nullptr;
    const long zrow = ((long)n * p.D + z0 + tz) * p.H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float rr[16], mm[16];
        unsigned eo[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h, y = 4 * i + (m >> 3), x = m & 7;
            const bool in = y0 + y < p.H && x0 + x < p.W;          // (a ragged tile's rows / columns outside the plane: nothing read, nothing written)
            eo[r] = in ? 4u * (unsigned)((((zrow + y0 + y) * p.W) + x0 + x) * C + col) : 0x80000000u;
            rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, (int)eo[r], 0, 0));
            mm[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)eo[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[i][r] + rr[r];
            if (p.relu) v = fmaxf(v, 0.f);
            if (has_mask) v = (mm[r] > 0.f) ? v : 0.f;
            if (eo[r] != 0x80000000u) p.out[eo[r] / 4] = v;
        }
    }
}

// ---- weight image: W[tap][ci][co] f32 -> bf16x3 B fragments, [k-step (channel chunk)][tap][column half][plane][lane] x 16 bytes ----
constexpr int PREP_MAX = 16;
struct PrepBatch {
    const float* w[PREP_MAX];
    unsigned char* img[PREP_MAX];
    int dgrad[PREP_MAX];
    int wide[PREP_MAX];       // image format: 0 = kind 1, 1 = kind 2, 2 = kind 3 (direct3_prep_kernel)
};
__device__ void direct3s_prep_body(const float* w, unsigned char* img, int dgrad, int idx);
__device__ void direct3s256_prep_body(const float* w, unsigned char* img, int dgrad, int idx);
constexpr int PREP_BLOCKS_256 = 8 * NTAP * 2 * 8 * 64 / 256;              // 864 (direct3s_kernel<256>)
constexpr int PREP_BLOCKS = NTAP * KS * 2 * 64 / 256;                     // 54
constexpr int PREP_BLOCKS_WIDE = 4 * NTAP * 2 * 4 * 64 / 256;             // 216 (both 128-channel formats)
// image of direct3_kernel<., CT>: [output block of 64][channel chunk][tap][column half][plane][lane] x 16 bytes;
// idx = (cb, tap, ks, cw, lane): (CT / 64) * 27 * (CT / 16) * 2 * 64 entries
template <int CT>
__device__ void direct3_prep_body(const float* w, unsigned char* img, int dgrad, int idx) {
    constexpr int KSn = CT / 16;
    const int lane = idx & 63, cw = (idx >> 6) & 1, rest = idx >> 7, ks = rest % KSn, tap = (rest / KSn) % NTAP, cb = rest / (KSn * NTAP);
    if (cb >= CT / 64) return;
    const int nn = cb * 64 + cw * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    float v[8];
    if (dgrad) {        // B'[tap][k = co][n = ci] = W[26 - tap][ci = n][co = k]
        const float* src = w + ((long)(NTAP - 1 - tap) * CT + nn) * CT + k0;
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {            // B[tap][k = ci][n = co] = W[tap][ci = k][co = n]
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[((long)tap * CT + k0 + e) * CT + nn];
    }
    u32x4 o[3];
    cut8(v, o);
    unsigned char* dst = img + (long)(((cb * KSn + ks) * NTAP + tap) * 2 + cw) * (3 * WBLK) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * WBLK) = o[pl];
}
// one launch cuts every image of a batch, all formats: blockIdx.y = image, blockIdx.x = its blocks (the 64-channel format
// uses the first 54 of them).  wide: 0 = direct3_kernel 64 channels, 1 = direct3s_kernel 128, 2 = direct3_kernel 128, 3 = direct3s_kernel 256
__global__ __launch_bounds__(256) void direct3_prep_kernel(PrepBatch b) {
    const int fmt = b.wide[blockIdx.y], idx = blockIdx.x * 256 + threadIdx.x;
    if (fmt == 1) { if (blockIdx.x < PREP_BLOCKS_WIDE) direct3s_prep_body(b.w[blockIdx.y], b.img[blockIdx.y], b.dgrad[blockIdx.y], idx); }
    else if (fmt == 2) { if (blockIdx.x < PREP_BLOCKS_WIDE) direct3_prep_body<128>(b.w[blockIdx.y], b.img[blockIdx.y], b.dgrad[blockIdx.y], idx); }
    else if (fmt == 3) direct3s256_prep_body(b.w[blockIdx.y], b.img[blockIdx.y], b.dgrad[blockIdx.y], idx);
    else if (blockIdx.x < PREP_BLOCKS) direct3_prep_body<C>(b.w[blockIdx.y], b.img[blockIdx.y], b.dgrad[blockIdx.y], idx);
}


// ---- weight gradient of the same convolutions -----------------------------------------------------------------------
//   dW[tap][ci][co] = sum over output voxels m of X[m + tap - 1][ci] * dY[m][co]
// The reduction runs over voxels, so both operands are needed k-major (k = voxel) while memory is channel-major: both are
// staged as [voxel][32 channels] bf16 rows of 64 bytes per channel half and bf16 plane, and the fragments come out of
// LDS through the transposing read (ds_read_b64_tr_b16: a lane names its own row, so a tap's x shift and the padding -
// a zero row - cost one address select, made once per kernel).
// Work split: a workgroup owns ONE (dz, dy) pair - the three dx taps, all 64 x 64 channels: its four waves are the four
// 32 x 32 (ci half, co half) blocks, three accumulators each - and every 28th z-plane of the batch (a plane of dY with
// the plane of X that dz pairs it with; planes whose partner lies outside the volume are skipped).  9 x 28 = 252
// workgroups; each writes its 3 taps of split-K slab s, 28 slabs in all (12 MB), summed by the caller's reduce launch.
// A z-plane (64 voxels = 4 k-steps of 16) is the staging unit: plane t + 1 is cut and stored into the other LDS buffer
// and plane t + 2 fetched in the shadow of plane t's 72 MFMAs per wave; one barrier per plane.
constexpr int WG_SPLITS = 28;
constexpr int WROW = 64;                    // bytes of a (voxel, 32 channels) row
constexpr int WHALF = (PLANE + 1) * WROW;   // one channel half of a plane: 64 voxel rows + a zero row
constexpr int WPL = 2 * WHALF;              // one bf16 plane
constexpr int WOP = 3 * WPL;                // one operand (X or dY) of one z-plane
constexpr int WBUF = 2 * WOP;               // one staging buffer
constexpr int WZERO = PLANE * WROW;         // byte offset of the zero row in a half

// Round 5: up to WG_MAXPROB convolutions of ONE geometry per launch (layer1's four weight gradients exist together when the data-gradient
// chain leaves the stage): nprob x splits workgroups per (dz, dy) pair, `splits` chains per problem (mi_direct3_wgrad_batch_splits).
constexpr int WG_MAXPROB = 4;
struct Direct3WgradParams {
    const float* x[WG_MAXPROB];       // (N, D, 8, 8, 64) each
    const float* dy[WG_MAXPROB];      // (N, D, 8, 8, 64)
    float* slabs[WG_MAXPROB];         // `splits` slabs of [27][64][64] floats per problem
    int N, D;
    unsigned bytes;           // extent of x / dy
    int nprob, splits;        // workgroups per (dz, dy) pair = nprob * splits
    int H, W;                 // TILED: the plane (multiples of 8); a unit is an 8 x 8 tile of it
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// Round 4: EIGHT waves per workgroup, two per SIMD: waves 0-3 take k-steps 0-1 of every plane and waves 4-7 k-steps 2-3 (same
// four 32 x 32 blocks, same three taps); the staging of a plane is shared by 512 threads (one unit of X and one of dY each), and
// the two halves of the reduction are added through LDS once, before the slab is written.  Measured (profiles/r04_experiments.txt
// item 12): 40.0 -> 38.7 us; of those 8.8 us are launch + prologue + the 12 MB of slabs, and 19 planes x 72 MFMAs per SIMD at the
// 2.05 GHz the chip sustains under MFMA load are 21.4 us - the loop runs at 0.72 of the matrix pipe.
// Round 5, V4 = true: the same kernel for layer2's shape - 128 -> 128 channels on 4 x 4 x 4 volumes (moco_encoder_3d.py:171; the implicit
// GEMM took 3 x 30-50 us for its three equal convolutions inside the step).  The staging unit is a SAMPLE's whole volume (64 voxels =
// 4 k-steps, like an 8 x 8 plane), so the z offset of a tap is a row select like its y offset (no partner plane, nothing skipped); a
// lane's eight k values are two image rows of four: the x-shifted fragments get a zero at BOTH row edges; a workgroup owns one
// (dz, dy) pair and one 64 x 64 block of the 128 x 128 channels (36 groups instead of 9).
// CT = channels of a voxel in memory (64 / 128 / 256): a workgroup owns one 64 x 64 block of the CT x CT channels, 9 (CT / 64)^2 groups.
// <false, 64> layer1 and <true, 128> layer2 of 32^3 crops; <false, 128> layer2 of 64^3 crops.
// TILED (round 5, <false, 64, true>: layer1 of 64^3 crops, 16 x 16 planes): the unit is an 8 x 8 TILE of a plane.  The tile of X is
// staged already shifted by the workgroup's dy (rows of the neighbouring tile, zeros outside the plane), so the fragment rows need no
// select; the x-shifted fragments take their edge element from ONE halo column per side (X[.][x0 - 1], X[.][x0 + 8]: 8 rows x 64
// channels x 3 bf16 planes x 2 sides = 6 KB per buffer, staged by all 512 threads - two floats each - and read back as 2-byte LDS reads,
// six per k-step).
constexpr int W_HB = 3 * 2 * 8 * 128;       // halo bytes of one staging buffer: [plane][side][row][64 channels] bf16
template <bool V4, int CT, bool TILED = false>
__global__ __launch_bounds__(512, 2) void direct3_wgrad_kernel(Direct3WgradParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WBUF + (TILED ? 2 * W_HB : 16)];
    unsigned char* const halo = lds + 2 * WBUF;
    static_assert(!TILED || (!V4 && CT == 64), "tiled planes: the 64-channel geometry");
    constexpr int NBLK = CT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;      // k-half of the plane; wave block: ci half wm x co half wn
    const int per_grp = p.nprob * p.splits;
    const int grp = blockIdx.x / per_grp, rem = blockIdx.x % per_grp;
    const int pb = rem / p.splits, split = rem % p.splits, stride = p.splits;
    const int pair = grp % 9, chb = grp / 9;
    const int ci0 = (chb / NBLK) * 64, co0 = (chb % NBLK) * 64;        // this workgroup's 64 x 64 channel block
    const int dz = pair / 3, dy = pair % 3;
    const int tiles_x = TILED ? p.W >> 3 : 1, tiles = TILED ? (p.H >> 3) * tiles_x : 1;
    const int n_planes = p.N * p.D * tiles;            // units: V4: D = 1, a "plane" is a sample; TILED: the 8 x 8 tiles of every plane

    // ---- staging: a thread's unit = (voxel, 8 channels) of X and the same unit of dY: 64 voxels x 8 channel groups ----
    const __amdgpu_buffer_rsrc_t xrs = rsrc_of(p.x[pb], p.bytes), yrs = rsrc_of(p.dy[pb], p.bytes);
    const int st_vox = tid >> 3, st_cg = tid & 7;
    const unsigned st_src = TILED ? 4u * (unsigned)(((st_vox >> 3) * p.W + (st_vox & 7)) * C + st_cg * 8)
                                  : 4u * (unsigned)(st_vox * CT + st_cg * 8);
    // TILED: this thread's halo unit = (side, row, channel pair)
    const int h_side = tid >> 8, h_row = (tid >> 5) & 7, h_c2 = tid & 31;
    const unsigned x_ch = 4u * (unsigned)ci0, y_ch = 4u * (unsigned)co0;
    const int zsh = V4 ? 0 : dz - 1;                   // partner plane of X (V4: the sample itself)
    const int st_lds = (st_cg >> 2) * WHALF + st_vox * WROW + (st_cg & 3) * 16;
    // plane index of this workgroup's i-th tile, or -1 behind the last: every WG_SPLITS-th (n, z) whose partner plane
    // z + dz - 1 is inside the volume
    auto next_plane = [&](int from) {
        if (V4) return from < n_planes ? from : -1;
        for (int pi = from; pi < n_planes; pi += stride) {
            const int zi = (pi / tiles) % p.D + dz - 1;
            if ((unsigned)zi < (unsigned)p.D) return pi;
        }
        return -1;
    };
    u32x4 ldx[2], ldy[2];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 ldh = {0u, 0u};
    // byte offsets of unit pi's operands: dY tile, X tile (TILED: + this thread's halo element), 0x80000000 = nothing to fetch (zeros)
    unsigned u_y, u_x, u_h;
    auto unit_offsets = [&](int pi) {
        if (pi < 0) { u_y = u_x = u_h = 0x80000000u; return; }
        if (!TILED) {
            u_y = 4u * (unsigned)((long)pi * PLANE * CT) + y_ch + st_src;
            u_x = 4u * (unsigned)((long)(pi + zsh) * PLANE * CT) + x_ch + st_src;
            u_h = 0x80000000u;
            return;
        }
        const int pl = pi / tiles, t = pi % tiles, y0 = (t / tiles_x) * 8, x0 = (t % tiles_x) * 8;
        u_y = 4u * (unsigned)((((long)pl * p.H + y0) * p.W + x0) * C) + st_src;
        const int xr = y0 + (st_vox >> 3) + dy - 1;                       // the row of X this thread stages: shifted by the tap's dy
        u_x = (unsigned)xr < (unsigned)p.H
                  ? 4u * (unsigned)((((long)(pl + dz - 1) * p.H + xr) * p.W + x0 + (st_vox & 7)) * C + st_cg * 8) : 0x80000000u;
        const int hr = y0 + h_row + dy - 1, hc = h_side ? x0 + 8 : x0 - 1;
        u_h = ((unsigned)hr < (unsigned)p.H && (unsigned)hc < (unsigned)p.W)
                  ? 4u * (unsigned)((((long)(pl + dz - 1) * p.H + hr) * p.W + hc) * C + 2 * h_c2) : 0x80000000u;
    };
    auto stage_load = [&](int pi) {                     // pi < 0: nothing to fetch (offsets out of range: zeros)
        unit_offsets(pi);
        ldx[0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)u_x, 0, 0);
        ldx[1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(u_x + 16u), 0, 0);
        ldy[0] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)u_y, 0, 0);
        ldy[1] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)(u_y + 16u), 0, 0);
        if (TILED) ldh = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)u_h, 0, 0);
    };
    // TILED: the halo unit's two floats cut into three bf16 planes, one 4-byte store per plane
    unsigned hcu[3];
    auto halo_cut = [&]() {
        const float a = __uint_as_float(ldh[0]), b = __uint_as_float(ldh[1]);
        const float a1 = a - __uint_as_float(ldh[0] & 0xffff0000u), b1 = b - __uint_as_float(ldh[1] & 0xffff0000u);
        const float a2 = a1 - __uint_as_float(__float_as_uint(a1) & 0xffff0000u), b2 = b1 - __uint_as_float(__float_as_uint(b1) & 0xffff0000u);
        constexpr unsigned HI2 = 0x07060302u;
        hcu[0] = __builtin_amdgcn_perm(ldh[1], ldh[0], HI2);
        hcu[1] = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(a1), HI2);
        hcu[2] = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), HI2);
    };
    auto halo_store = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<unsigned*>(halo + buf * W_HB + ((pl * 2 + h_side) * 8 + h_row) * 128 + h_c2 * 4) = hcu[pl];
    };
    auto stage_store_unit = [&](int buf, int op) {          // op 0: X, 1: dY
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = __uint_as_float(op ? ldy[0][e] : ldx[0][e]);
            v[4 + e] = __uint_as_float(op ? ldy[1][e] : ldx[1][e]);
        }
        u32x4 o[3];
        cut8(v, o);
        unsigned char* dst = lds + buf * WBUF + op * WOP + st_lds;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * WPL) = o[pl];
        if (TILED && op == 0) { halo_cut(); halo_store(buf); }
    };

    // ---- fragment addresses (transposing read: this lane names row q of its 16-lane group's 4-row block) ----
    // k-step ks covers output voxels 16 ks .. 16 ks + 15; MFMA k = 8 h + e: the lane's "lo" read fetches rows
    // 16 ks + 8 h + q (e = 0..3 after the transpose), the "hi" read rows + 4.  Output voxel v = (y, x) = (v >> 3, v & 7)
    // pairs with input voxel (y + dy - 1, x + dx - 1) of the partner plane.  This wave's k-steps: 2 kh + lk, lk = 0, 1.
    const int q4 = i16 >> 2;
    const int coloff = (16 * g16 + 4 * (i16 & 3)) * 2;
    const int b_base = wn * WHALF + (2 * kh * 16 + 8 * h + q4) * WROW + coloff;         // + lk * 16 rows, + 4 rows for "hi"
    // Only the dx = 1 tap is READ: a lane's eight k values are the eight x positions of ONE image row (y = 2 ks + h), so the
    // fragments of dx = 0 / 2 are the same registers moved by one bf16 element with a zero shifted in at the row's edge
    // (4 v_alignbit per fragment instead of 2 transposing reads: the LDS pipe, not the matrix pipe, was the busy one).
    int a_sel[2][2];                  // byte offset inside an X plane-half image, or the zero row
#pragma unroll
    for (int lk = 0; lk < 2; ++lk)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int v = 16 * (2 * kh + lk) + 8 * h + 4 * hi + q4;
            if (V4) {      // v = (z, y, x) of a 4 x 4 x 4 volume: the input voxel (z + dz - 1, y + dy - 1, x), x shifted in registers
                const int zi = (v >> 4) + dz - 1, yi = ((v >> 2) & 3) + dy - 1;
                a_sel[lk][hi] = wm * WHALF + (((unsigned)zi < 4u && (unsigned)yi < 4u) ? ((zi * 4 + yi) * 4 + (v & 3)) * WROW : WZERO) + coloff;
            } else if (TILED) {                               // (the tile of X was staged shifted by dy: row v is the partner of output row v)
                a_sel[lk][hi] = wm * WHALF + v * WROW + coloff;
            } else {
                const int yi = (v >> 3) + dy - 1;
                a_sel[lk][hi] = wm * WHALF + ((unsigned)yi < 8u ? (yi * 8 + (v & 7)) * WROW : WZERO) + coloff;
            }
        }
    // TILED: this lane's halo elements of local k-step lk: image row 2 (2 kh + lk) + h, channel 32 wm + l32
    int h_off[2];
#pragma unroll
    for (int lk = 0; lk < 2; ++lk) h_off[lk] = (2 * (2 * kh + lk) + h) * 128 + (32 * wm + l32) * 2;
    unsigned hl[2][2][3];             // [set][side][plane]

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    // zero rows of both buffers, both operands, all planes and halves
    for (int i = tid; i < 2 * 2 * 3 * 2 * (WROW / 16); i += 512) {
        const int img = i / (WROW / 16), c16 = i % (WROW / 16);
        *reinterpret_cast<u32x4*>(lds + img * WHALF + WZERO + c16 * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    int cur = next_plane(split);
    int nxt = cur >= 0 ? next_plane(cur + stride) : -1;
    stage_load(cur);
    if (cur >= 0) { stage_store_unit(0, 0); stage_store_unit(0, 1); }
    stage_load(nxt);
    __syncthreads();

    // Fragment reads and staging are cut into pieces, one behind each MFMA of a k-step (pinned by sched_barrier): left to
    // the scheduler the 50 instructions of a unit's cut went out in one block in front of the MFMAs and the matrix pipe sat
    // idle for a third of the k-step.
    bf16x8 af[2][3][3], bfg[2][3];
    // piece j (0..5) of the 12 transposing reads of local k-step lk: j < 3: dY plane j; else the dx = 1 tap of X, plane j - 3
    auto read_piece = [&](int buf, int lk, int j, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const unsigned char* xb = lds + buf * WBUF;
        if (j < 3) {
            const unsigned char* yb = xb + WOP + b_base + lk * 16 * WROW + j * WPL;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(yb));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(yb + 4 * WROW));
            bfg[SET][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int pl = j - 3;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xb + pl * WPL + a_sel[lk][0]));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xb + pl * WPL + a_sel[lk][1]));
            af[SET][1][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            if (TILED) {
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    hl[SET][sd][pl] = *reinterpret_cast<const unsigned short*>(halo + buf * W_HB + ((pl * 2 + sd) * 8) * 128 + h_off[lk]);
            }
        }
    };
    auto read_piece_dyn = [&](int buf, int lk, int j) {
        if (lk & 1) read_piece(buf, lk, j, std::integral_constant<int, 1>{});
        else read_piece(buf, lk, j, std::integral_constant<int, 0>{});
    };
    // the order the first MFMAs of a k-step want their operands in: (dY 0, X 2), (dY 2, X 0), (dY 1, X 1)
    auto read_pair = [&](int buf, int lk, int i) {
        constexpr int BP[3] = {0, 2, 1}, AP[3] = {2, 0, 1};
        read_piece_dyn(buf, lk, BP[i]);
        read_piece_dyn(buf, lk, 3 + AP[i]);
    };
    // fragment of tap dxv (0 or 2), plane pl, from the dx = 1 fragment: element e <- element e + dxv - 1, zero at the edge
    auto derive = [&](int set, int dxv, int pl) {
        const u32x4 c = __builtin_bit_cast(u32x4, af[set][1][pl]);
        u32x4 o;
        if (dxv == 0) {
            o[0] = c[0] << 16;
#pragma unroll
            for (int j = 1; j < 4; ++j) o[j] = __builtin_amdgcn_alignbit(c[j], c[j - 1], 16);
            if (V4) o[2] &= 0xffff0000u;               // element 4 = x 0 of the second row: nothing to its left
            if (TILED) o[0] |= hl[set][0][pl];         // element 0 <- X[.][x0 - 1]
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) o[j] = __builtin_amdgcn_alignbit(c[j + 1], c[j], 16);
            o[3] = c[3] >> 16;
            if (V4) o[1] &= 0x0000ffffu;               // element 3 = x 3 of the first row: nothing to its right
            if (TILED) o[3] |= hl[set][1][pl] << 16;   // element 7 <- X[.][x0 + 8]
        }
        af[set][dxv][pl] = __builtin_bit_cast(bf16x8, o);
    };
    // staging pieces of operand op: the cut of element e in two halves, one MFMA slot apart (first residual, then second:
    // the four instructions of an element depend on each other, and a chain of four dependent VALU operations behind
    // every MFMA held the next MFMA back - the two halves of neighbouring elements in one slot are independent);
    // then pack + store of plane pl
    unsigned cu[3][8];
    auto cut_a = [&](int op, int e) {
        const float x = __uint_as_float(op ? ldy[e >> 2][e & 3] : ldx[e >> 2][e & 3]);
        cu[0][e] = __float_as_uint(x);
        cu[1][e] = __float_as_uint(x - __uint_as_float(cu[0][e] & 0xffff0000u));
    };
    auto cut_b = [&](int e) {
        const float r1 = __uint_as_float(cu[1][e]);
        cu[2][e] = __float_as_uint(r1 - __uint_as_float(cu[1][e] & 0xffff0000u));
    };
    auto store_piece = [&](int buf, int op, int pl) {
        constexpr unsigned HI2 = 0x07060302u;
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_perm(cu[pl][2 * d + 1], cu[pl][2 * d], HI2);
        *reinterpret_cast<u32x4*>(lds + buf * WBUF + op * WOP + st_lds + pl * WPL) = o;
    };

    int buf = 0;
    if (cur >= 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) read_pair(0, 0, i);
    }
    constexpr int DXO[3] = {1, 0, 2};                     // MFMA order of the taps: the one that was read, then the derived ones
    while (cur >= 0) {
        const int nn = nxt >= 0 ? next_plane(nxt + stride) : -1;      // the plane after next: fetched during this one
        unit_offsets(nn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int lk = 0; lk < 2; ++lk) {
            const int op = lk;                            // the unit of the NEXT plane that is cut + stored behind this k-step
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const int dxm = DXO[m / 6];
                acc[dxm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[lk][dxm][PA[m % 6]], bfg[lk][PB[m % 6]], acc[dxm], 0, 0, 0);
                // behind the MFMA: the derived fragments (tap 0 behind MFMAs 0-2, tap 2 behind 6-8), one piece of the staging,
                if (m < 3) derive(lk, 0, m);
                if (m >= 6 && m < 9) derive(lk, 2, m - 6);
                if (m >= 2 && m < 10) cut_a(op, m - 2);
                if (m >= 3 && m < 11) cut_b(m - 3);
                if (m >= 11 && m < 14) store_piece(buf ^ 1, op, m - 11);
                // the fetch of the plane after next (its staging registers were consumed by the cuts above) ...
                if (lk == 1 && m >= 14) {
                    const int i = m - 14, hf = i & 1;
                    if (i < 2) ldx[hf] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(u_x + 16u * hf), 0, 0);
                    else ldy[hf] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)(u_y + 16u * hf), 0, 0);
                    if (TILED && i == 3) ldh = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)u_h, 0, 0);
                }
                // TILED: the next unit's halo column, behind the X unit's pieces (its registers were fetched with the unit's)
                if (TILED && lk == 0 && m == 14) halo_cut();
                if (TILED && lk == 0 && m == 15) halo_store(buf ^ 1);
                // ... and the fragment reads of the next k-step.  The last k-step of a plane reads the NEXT plane's first
                // k-step, which is complete once every wave has stored its last unit: the barrier sits behind MFMA 13
                if (lk == 0) {
                    if (m < 3) read_pair(buf, 1, m);
                } else {
                    if (m == 13) __syncthreads();
                    if (m >= 14 && m < 17 && nxt >= 0) read_pair(buf ^ 1, 0, m - 14);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf ^= 1;
        cur = nxt;
        nxt = nn;
    }

    // ---- the two k-halves added through LDS (waves 4-7 hand their tiles to waves 0-3), then the slab: this workgroup's
    //      three taps of slab `split`; C/D layout col = lane & 31 (co), row = ci ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);          // [block 4][48 registers][64 lanes]: 48 KB
    if (kh == 1) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave & 3) * 3 + dx) * 16 + r) * 64 + lane] = acc[dx][r];
    }
    __syncthreads();
    if (kh == 1) return;
    float* out = p.slabs[pb] + (long)split * (NTAP * CT * CT);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[((long)tap * CT + ci) * CT + co0 + 32 * wn + l32] = acc[dx][r] + red[((wave * 3 + dx) * 16 + r) * 64 + lane];
        }
    }
}


// ---- layer2-shaped convolutions: 3^3 / stride 1 / padding 1, 128 -> 128 channels on 4 x 4 x 4 volumes --------------------
// (moco_encoder_3d.py:55-84,171: three of them per encoder pass, 9 forward / data-gradient launches per step.)  M is small
// (64 voxels per sample), so the parallelism comes from the reduction - and since round 4 the reduction is split INSIDE the
// workgroup: a workgroup owns ONE sample (64 rows) x 32 output channels, its four waves take one chunk of 32 input
// channels each (64 samples x 4 column blocks = 256 workgroups), the four partial tiles are added through LDS in chunk
// order and the convolution's epilogue (residual, ReLU, mask) runs on the sum: the launch is final - no split-K slabs, no
// reduce launch behind it (rounds 2-3: 4 slabs of 2 MB and a 4.9-us launch per convolution, nine per step), and the values
// are the ones the slab form produced (same products, same order of the four partial sums).
// Same structure as direct3_kernel otherwise: the sample's 64 x 128 patch is cut once into LDS (4 chunks x 30 KB), weight
// fragments stream from the pre-cut image (a wave reads its chunk's stream), every halo voxel is padding (the volume IS
// the tile) and reads the lane's zero record.
constexpr int CS = 128;                     // channels, in = out
constexpr int VS = 64;                      // voxels of a 4 x 4 x 4 sample
constexpr int S_LEAD = 32;                  // records in front of voxel 0 (>= 21, the -21 of the tap offsets; 32 + 64 = 0 mod 16)
constexpr int S_NV = VS;                    // one sample
constexpr int S_ZREC = 64;                  // (record mod 16) + tap offset (0..42) stays inside
constexpr int S_NREC = S_LEAD + S_NV + S_ZREC;
constexpr int S_ARR = S_NREC * 16;          // 2,560
constexpr int S_PL = 2 * S_ARR;
constexpr int S_KS = 3 * S_PL;
constexpr int S_CH = 2 * S_KS;              // one 32-channel chunk of the patch: 30,720
constexpr int S_ZBASE = (S_LEAD + S_NV) * 16;
constexpr int S_CHUNKS = 4;                 // 32-channel chunks = waves of the workgroup
constexpr int S_LDS = S_CHUNKS * S_CH;      // 122,880
constexpr int S_STEPS = 2 * NTAP;           // k-steps per chunk: g = tap * 2 + ks
constexpr int S_WSTEP = 4 * 3 * WBLK;       // bytes per k-step of the image: [column block 0..3][plane]
constexpr int S_WIMG_BYTES = S_CHUNKS * S_STEPS * S_WSTEP;       // 2,654,208
static_assert(S_STEPS % RB == 0, "ring slots");
static_assert(((S_LEAD + S_NV) & 15) == 0, "zero region must start at a record = 0 (mod 16)");
static_assert(S_LDS <= 160 * 1024 && 4 * 32 * 64 * 4 <= S_LDS, "patch (and the four partial tiles behind it) fit the LDS");

struct Direct3sParams {
    const float* a;           // X or dY: (N, 4, 4, 4, 128)
    const unsigned char* wimg;
    float* out;               // (N, 4, 4, 4, 128), final
    const float* res;         // out = act(acc + res)          (may be null)
    const float* mask;        // out *= (mask > 0)             (may be null)
    int relu;
    int N;
    unsigned a_bytes;
};

// CT: channels, in = out: 128 (layer2 at 32^3 crops) or - round 4 - 256 (layer3 at 64^3 crops): CT / 32 column blocks (workgroups per
// sample) and CT / 128 passes of four 32-channel chunks through the same 120 KB patch, the accumulators kept across the passes.
template <int CT>
__global__ __launch_bounds__(256, 1) void direct3s_kernel(Direct3sParams p) {
    constexpr int NCB = CT / 32;                         // column blocks = 32-channel chunks of the reduction
    constexpr int NPASS = NCB / S_CHUNKS;
    constexpr int WST = NCB * 3 * WBLK;                  // bytes per k-step of the image: [column block][plane]
    __shared__ __attribute__((aligned(16))) unsigned char patch[S_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int cc = __builtin_amdgcn_readfirstlane(tid >> 6);          // the wave's chunk (of the pass) of 32 input channels
    const int h = lane >> 5, l32 = lane & 31;
    const int ct = blockIdx.x % NCB, n0 = blockIdx.x / NCB;           // column block of 32 output channels, sample

    const __amdgpu_buffer_rsrc_t wrs = rsrc_of(p.wimg, NCB * S_STEPS * WST);
    const int w_voff = ct * (3 * WBLK) + lane * 16;
    int w_soff = cc * (S_STEPS * WST);
    bf16x8 bfr[RB][3];
    auto wload = [&](int g, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value;
        // behind the chunk: zeros - the out-of-range offset goes into the CHECKED voffset (soffset is not range-checked)
        const int vo = g < S_STEPS ? w_voff : (int)0x80000000u;
        const int so = g < S_STEPS ? w_soff + g * WST : 0;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bfr[SLOT][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo + pl * WBLK, so, 0));
    };
    auto wload_dyn = [&](int g) {
        switch (g % RB) {
            case 0: wload(g, std::integral_constant<int, 0>{}); break;
            case 1: wload(g, std::integral_constant<int, 1>{}); break;
            case 2: wload(g, std::integral_constant<int, 2>{}); break;
            case 3: wload(g, std::integral_constant<int, 3>{}); break;
            case 4: wload(g, std::integral_constant<int, 4>{}); break;
            default: wload(g, std::integral_constant<int, 5>{}); break;
        }
    };
    const __amdgpu_buffer_rsrc_t ars = rsrc_of(p.a, p.a_bytes);
    for (int i = tid; i < 12 * S_CHUNKS * S_ZREC; i += 256) {
        const int arr = i / S_ZREC, r = i % S_ZREC;
        *reinterpret_cast<u32x4*>(patch + arr * S_ARR + S_ZBASE + r * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // ---- stage 128 channels of the patch: unit q = (voxel, group of 8 channels): 64 x 16 units, 4 per thread ----
    auto stage = [&](int pass) {
        u32x4 ld[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u, vox = q >> 4, cg = q & 15;
            const unsigned off = 4u * (unsigned)(((long)n0 * VS + vox) * CT + pass * 128 + cg * 8);
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)off, 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(off + 16u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u, vox = q >> 4, cg = q & 15;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            cut8(v, o);
            unsigned char* dst = patch + (cg >> 2) * S_CH + ((cg >> 1) & 1) * S_KS + (cg & 1) * S_ARR + (S_LEAD + vox) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * S_PL) = o[pl];
        }
    };

    // ---- per-lane geometry: row block i = voxels 32 i .. 32 i + 31 of the sample ----
    int vbase[2], zbase[2];
    unsigned vmask[2];                  // bit tap: the tap's neighbour is inside the 4 x 4 x 4 volume
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = 32 * i + l32, z = v >> 4, y = (v >> 2) & 3, x = v & 3;
        const int rec = S_LEAD - 21 + v;
        vbase[i] = cc * S_CH + rec * 16 + h * S_ARR;
        zbase[i] = cc * S_CH + S_ZBASE + (rec & 15) * 16 + h * S_ARR;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            if ((unsigned)(z + dz - 1) < 4u && (unsigned)(y + dy - 1) < 4u && (unsigned)(x + dx - 1) < 4u) m |= 1u << t;
        }
        vmask[i] = m;
    }

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    bf16x8 af[2][2][3];
    auto frags = [&](int g, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int tap = g >> 1, ksx = g & 1;
        const int imm = ((tap / 9) * 16 + ((tap / 3) % 3) * 4 + tap % 3) * 16 + ksx * S_KS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sel = ((vmask[i] >> tap) & 1u) ? vbase[i] : zbase[i];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[SET][i][pl] = *reinterpret_cast<const bf16x8*>(patch + sel + imm + pl * S_PL);
        }
    };
    auto frags_dyn = [&](int g) {
        if (g & 1) frags(g, std::integral_constant<int, 1>{});
        else frags(g, std::integral_constant<int, 0>{});
    };

#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        w_soff = (pass * S_CHUNKS + cc) * (S_STEPS * WST);
        if (pass > 0) __syncthreads();                  // every wave is done with the previous 128 channels of the patch
#pragma unroll
        for (int g = 0; g < RB - 1; ++g) wload_dyn(g);
        stage(pass);
        __syncthreads();
        frags_dyn(0);
        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
        for (int g = 0; g < S_STEPS; ++g) {
            wload_dyn(g + RB - 1);
            if (g + 1 < S_STEPS) frags_dyn(g + 1);
    #pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][0][PA[pr]], bfr[g % RB][PB[pr]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][1][PA[pr]], bfr[g % RB][PB[pr]], acc[1], 0, 0, 0);
            }
    #pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
    #pragma unroll
            for (int k = 0; k < 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the four chunks' tiles added through LDS in chunk order ((0 + 1) + 2) + 3 - the order the slab reduce used -,
    //      wave w finishing rows block (w >> 1), registers 8 (w & 1) .. + 7; C/D layout col = lane & 31,
    //      row = (r & 3) + 8 (r >> 2) + 4 h ----
    __syncthreads();                                    // every wave is done with the patch
    float* red = reinterpret_cast<float*>(patch);       // [chunk][32 registers][64 lanes]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(cc * 32 + i * 16 + r) * 64 + lane] = acc[i][r];
    __syncthreads();
    const long m0 = (long)n0 * VS;
    const int col = ct * 32 + l32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = cc * 8 + j, i = idx >> 4, r = idx & 15;
        float t = ((red[idx * 64 + lane] + red[(32 + idx) * 64 + lane]) + red[(64 + idx) * 64 + lane]) + red[(96 + idx) * 64 + lane];
        const long o = (m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * CT + col;
        if (p.res) t += p.res[o];
        if (p.relu) t = fmaxf(t, 0.f);
        if (p.mask) t = (p.mask[o] > 0.f) ? t : 0.f;
        p.out[o] = t;
    }
}

// weight image of direct3s_kernel<CT>: [chunk CT/32][tap 27][ks 2][column block CT/32][plane 3][lane 64] x 16 bytes
template <int CT>
__device__ void direct3s_prep_body_t(const float* w, unsigned char* img, int dgrad, int idx) {
    constexpr int NCB = CT / 32;
    // idx = (chunk, tap, ks, cb, lane): NCB * 27 * 2 * NCB * 64 entries (216 * 256 for 128 channels, 864 * 256 for 256)
    const int lane = idx & 63, cb = (idx >> 6) % NCB, r1 = (idx >> 6) / NCB, ksx = r1 & 1, rest = r1 >> 1, tap = rest % NTAP, cc = rest / NTAP;
    if (cc >= NCB) return;
    const int nn = cb * 32 + (lane & 31), k0 = cc * 32 + ksx * 16 + 8 * (lane >> 5);
    float v[8];
    if (dgrad) {
        const float* src = w + ((long)(NTAP - 1 - tap) * CT + nn) * CT + k0;
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[((long)tap * CT + k0 + e) * CT + nn];
    }
    u32x4 o[3];
    cut8(v, o);
    unsigned char* dst = img + (long)(((cc * NTAP + tap) * 2 + ksx) * NCB + cb) * (3 * WBLK) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * WBLK) = o[pl];
}
__device__ void direct3s_prep_body(const float* w, unsigned char* img, int dgrad, int idx) { direct3s_prep_body_t<CS>(w, img, dgrad, idx); }
__device__ void direct3s256_prep_body(const float* w, unsigned char* img, int dgrad, int idx) { direct3s_prep_body_t<256>(w, img, dgrad, idx); }

}  // namespace

// ---- host side (internal: conv_igemm.hip's run_conv dispatches here; extern "C" wrappers at the end) ----
// 0: not a direct shape; 1: 64 -> 64 channels on 8 x 8 planes (direct3_kernel); 2: 128 -> 128 on 4 x 4 x 4 (direct3s_kernel);
// 3 (round 4): 128 -> 128 on 8 x 8 planes (direct3_kernel<., 128>: layer2 of a 64^3 crop) - reached through mi_conv3d_* /
// mi_convnd_* only (image cut per call), not through the caller-kept images of mi_conv3d_direct_*
int mi_direct3_kind(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph,
                    int pw, int dd, int dh, int dw) {
    const char* off = getenv("MI_CONV_NO_DIRECT");      // A/B switch: keep the implicit GEMM
    if (off && atoi(off) != 0) return 0;
    if (kd != 3 || kh != 3 || kw != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1 || dd != 1 || dh != 1 || dw != 1) return 0;
    if (N < 1 || 4l * N * Di * Hi * Wi * Ci >= 0x7fff0000l) return 0;
    if (Ci == C && Co == C && Hi == 8 && Wi == 8 && Di >= TZ && Di % TZ == 0) return 1;
    if (Ci == CS && Co == CS && Di == 4 && Hi == 4 && Wi == 4) return 2;
    // (from 128 workgroups on: 64 of them - batch 8 of 64^3 crops - take 47.6 us where the implicit GEMM takes 39.5; batch 16: 51 against
    // 62 us, batch 32: 71 against 105 us = 204 TFLOP/s, image cut included)
    if (Ci == 128 && Co == 128 && Hi == 8 && Wi == 8 && Di >= TZ && Di % TZ == 0 && (long)N * Di >= 128 && (long)N * (Di / TZ) <= 0x7fffffffl) return 3;
    // 4 (round 4): 256 -> 256 on 4 x 4 x 4 (direct3s_kernel<256>: layer3 of a 64^3 crop), from 128 workgroups on (batch >= 16)
    if (Ci == 256 && Co == 256 && Di == 4 && Hi == 4 && Wi == 4 && N >= 16) return 4;
    // 5 (round 4): 64 -> 64 on planes of 8 x 8 tiles with a halo (direct3h_kernel: layer1 of a 64^3 crop), from 128 workgroups on
    // (round 5: ragged planes too - 12 x 12 for 48^3 crops: 8 x 8 tiles hanging over the edge; MI_DIRECT3H_RAGGED=0: multiples of 8 only)
    if (Ci == C && Co == C && (Hi > 8 || Wi > 8) && Hi >= 8 && Wi >= 8 && Di >= TZ && Di % TZ == 0) {
        // measured at 12 x 12 (batch 16): 57.5 us against the implicit GEMM's 49.0 - the plane fills 56 % of its four tiles; the ragged
        // form is taken from 75 % on (14 x 14 .. 15 x 15, 22 x 22 ..), MI_DIRECT3H_RAGGED=1 forces it, =0 forbids it
        const bool ragged = Hi % 8 != 0 || Wi % 8 != 0;
        const char* rg = getenv("MI_DIRECT3H_RAGGED");
        const long ty = (Hi + 7) / 8, tx = (Wi + 7) / 8, wgs = (long)N * (Di / TZ) * ty * tx;
        const bool fill_ok = 4l * Hi * Wi >= 3l * 64 * ty * tx;
        const bool take = !ragged || (rg ? atoi(rg) != 0 : fill_ok);
        if (wgs >= 128 && wgs <= 0x7fffffffl && take) return 5;
    }
    return 0;
}
bool mi_direct3_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph,
                       int pw, int dd, int dh, int dw) {
    return mi_direct3_kind(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw, dd, dh, dw) != 0;
}

size_t mi_direct3_wimg_bytes(int channels) {
    return channels == C ? (size_t)WIMG_BYTES : channels == CS ? (size_t)S_WIMG_BYTES : 0;
}
size_t mi_direct3_wimg_bytes_kind(int kind) {
    return (kind == 1 || kind == 5) ? (size_t)WIMG_BYTES : kind == 2 ? (size_t)S_WIMG_BYTES : kind == 3 ? (size_t)(2 * 8 * NTAP * WSTEP)
         : kind == 4 ? (size_t)8 * S_STEPS * (8 * 3 * WBLK) : 0;
}
// (rounds 2-3: split-K slabs of the 128-channel kernel; since round 4 both direct kernels are final in one launch)
size_t mi_direct3_slab_bytes(int, int) { return 0; }

// kinds[i] = 1 .. 5 (mi_direct3_kind) selects the image format of weight i (5 shares the format of 1)
int mi_direct3_prep_kind(const float* const* w, void* const* img, const int* dgrad, const int* kinds, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += PREP_MAX) {
        PrepBatch b = {};
        const int m = n - i0 < PREP_MAX ? n - i0 : PREP_MAX;
        int blocks = PREP_BLOCKS;
        for (int i = 0; i < m; ++i) {
            const int kd = kinds[i0 + i];
            if (kd < 1 || kd > 5 || !w[i0 + i] || !img[i0 + i]) return MI_E_ARG;
            b.w[i] = w[i0 + i]; b.img[i] = (unsigned char*)img[i0 + i]; b.dgrad[i] = dgrad[i0 + i];
            b.wide[i] = kd == 5 ? 0 : kd - 1;
            blocks = std::max(blocks, kd == 4 ? PREP_BLOCKS_256 : (kd == 2 || kd == 3) ? PREP_BLOCKS_WIDE : PREP_BLOCKS);
        }
        hipLaunchKernelGGL(direct3_prep_kernel, dim3(blocks, m), dim3(256), 0, s, b);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}
// channels[i] = 64 or 128 selects the image format of weight i (kinds 1 / 2: the formats of the caller-kept images)
int mi_direct3_prep(const float* const* w, void* const* img, const int* dgrad, const int* channels, int n, hipStream_t s) {
    if (n > 4096) return MI_E_ARG;
    int kinds[4096];
    for (int i = 0; i < n; ++i) {
        if (channels[i] != C && channels[i] != CS) return MI_E_ARG;
        kinds[i] = channels[i] == C ? 1 : 2;
    }
    return mi_direct3_prep_kind(w, img, dgrad, kinds, n, s);
}

// `a` = X (forward) / dY (data gradient); wimg from mi_direct3_prep with the matching `dgrad` flag
int mi_direct3_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                      int D, hipStream_t s) {
    Direct3Params p = {a, (const unsigned char*)wimg, out, res, mask, relu, N, D, (unsigned)(4l * N * D * PLANE * C)};
    // two resident chunks / two workgroups per CU by default (captured step 1.628 against 1.649 ms, r03_experiments.txt item 20);
    // MI_DIRECT3_FOUR_SLOTS=1: the whole patch resident, one workgroup per CU
    if (getenv("MI_DIRECT3_FOUR_SLOTS")) hipLaunchKernelGGL(direct3_kernel<false>, dim3((unsigned)(N * (D / TZ))), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(direct3_kernel<true>, dim3((unsigned)(N * (D / TZ))), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// 64 channels on planes of 8 x 8 tiles (kind 5)
int mi_direct3h_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                       int D, int H, int W, hipStream_t s) {
    Direct3hParams p = {a, (const unsigned char*)wimg, out, res, mask, relu, N, D, H, W, (unsigned)(4l * N * D * H * W * C)};
    hipLaunchKernelGGL(direct3h_kernel, dim3((unsigned)((long)N * (D / TZ) * ((H + 7) / 8) * ((W + 7) / 8))), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// 128 channels on 8 x 8 planes (kind 3): the same kernel, eight chunks, two 64-channel output blocks (blockIdx.y)
int mi_direct3_launch128(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                         int D, hipStream_t s) {
    Direct3Params p = {a, (const unsigned char*)wimg, out, res, mask, relu, N, D, (unsigned)(4l * N * D * PLANE * 128)};
    hipLaunchKernelGGL((direct3_kernel<true, 128>), dim3((unsigned)(N * (D / TZ)), 2), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// 128-channel kernel: one workgroup per (sample, 32 output channels), final with the epilogue
int mi_direct3s_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                       hipStream_t s) {
    Direct3sParams p = {a, (const unsigned char*)wimg, out, res, mask, relu, N, (unsigned)(4l * N * VS * CS)};
    hipLaunchKernelGGL(direct3s_kernel<CS>, dim3((unsigned)(4 * N)), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// 256 channels on 4 x 4 x 4 (kind 4): eight column blocks per sample, two passes of 128 channels
int mi_direct3s_launch256(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                          hipStream_t s) {
    Direct3sParams p = {a, (const unsigned char*)wimg, out, res, mask, relu, N, (unsigned)(4l * N * VS * 256)};
    hipLaunchKernelGGL(direct3s_kernel<256>, dim3((unsigned)(8 * N)), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

size_t mi_direct3_wgrad_slab_bytes() { return sizeof(float) * (size_t)WG_SPLITS * NTAP * C * C; }
int mi_direct3_wgrad_splits() { return WG_SPLITS; }

// writes WG_SPLITS full-size slabs ([27][64][64] floats each) into `slabs`; the caller sums them
int mi_direct3_wgrad_launch(const float* x, const float* dy, float* slabs, int N, int D, hipStream_t s) {
    Direct3WgradParams p = {};
    p.x[0] = x; p.dy[0] = dy; p.slabs[0] = slabs; p.N = N; p.D = D; p.bytes = (unsigned)(4l * N * D * PLANE * C);
    p.nprob = 1; p.splits = WG_SPLITS;
    hipLaunchKernelGGL((direct3_wgrad_kernel<false, 64>), dim3(9 * WG_SPLITS), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// nb (2..WG_MAXPROB) problems of one geometry in one launch: problem i leaves mi_direct3_wgrad_batch_splits(nb) slabs in slabs[i]
int mi_direct3_wgrad_batch_max() { return WG_MAXPROB; }
// A batched problem is cut into WG_BATCH_SPLITS chains per (dz, dy) pair WHATEVER nb is (so a problem's gradient does not depend on
// its neighbours in the launch): four problems x 7 x 9 = 252 workgroups, each walking four times the planes of a single launch's
// workgroup - launch, prologue, pipeline fill, the LDS hand-over and the slab store are paid once per 73 planes instead of once per
// 19.  Measured in the captured step (r05_experiments.txt item 7): 28 chains per problem (1,008 workgroups, four rounds) 1.561 ms =
// the four single launches; 14: 1.535; 7: 1.524.  (MI_D3W_BATCH_SPLITS: tuning.)
constexpr int WG_BATCH_SPLITS = 7;
int mi_direct3_wgrad_batch_splits(int nb) {
    if (nb < 1 || nb > WG_MAXPROB) return 0;
    if (const char* v = getenv("MI_D3W_BATCH_SPLITS")) { const int sv = atoi(v); if (sv >= 1 && sv <= WG_SPLITS) return sv; }
    return WG_BATCH_SPLITS;
}
int mi_direct3_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* slabs, int nb, int N, int D,
                                  hipStream_t s) {
    if (nb < 1 || nb > WG_MAXPROB) return MI_E_ARG;
    Direct3WgradParams p = {};
    for (int i = 0; i < nb; ++i) { p.x[i] = xs[i]; p.dy[i] = dys[i]; p.slabs[i] = slabs[i]; }
    p.N = N; p.D = D; p.bytes = (unsigned)(4l * N * D * PLANE * C);
    p.nprob = nb; p.splits = mi_direct3_wgrad_batch_splits(nb);
    hipLaunchKernelGGL((direct3_wgrad_kernel<false, 64>), dim3(9 * nb * p.splits), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---- layer2's shape (128 -> 128 on 4 x 4 x 4, kind 2): direct3_wgrad_kernel<true> ----
// single launch: 7 chains of samples per group (36 groups: 252 workgroups); batched (2..4 problems): 2 chains per problem whatever nb
// (3 problems: 216 workgroups, each walking half the batch) - MI_D3SW_BATCH_SPLITS: tuning
constexpr int WS_SPLITS = 7, WS_BATCH_SPLITS = 2;
size_t mi_direct3s_wgrad_slab_bytes() { return sizeof(float) * (size_t)WS_SPLITS * NTAP * CS * CS; }
int mi_direct3s_wgrad_splits(int nb) {
    if (nb < 1 || nb > WG_MAXPROB) return 0;
    if (nb == 1) return WS_SPLITS;
    if (const char* v = getenv("MI_D3SW_BATCH_SPLITS")) { const int sv = atoi(v); if (sv >= 1 && sv <= WS_SPLITS) return sv; }
    return WS_BATCH_SPLITS;
}
// problem i leaves mi_direct3s_wgrad_splits(nb) slabs of [27][128][128] floats in slabs[i]; the caller sums them
int mi_direct3s_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* slabs, int nb, int N, hipStream_t s) {
    if (nb < 1 || nb > WG_MAXPROB || N < 1) return MI_E_ARG;
    Direct3WgradParams p = {};
    for (int i = 0; i < nb; ++i) { p.x[i] = xs[i]; p.dy[i] = dys[i]; p.slabs[i] = slabs[i]; }
    p.N = N; p.D = 1; p.bytes = (unsigned)(4l * N * PLANE * CS);
    p.nprob = nb; p.splits = mi_direct3s_wgrad_splits(nb);
    hipLaunchKernelGGL((direct3_wgrad_kernel<true, 128>), dim3(36 * nb * p.splits), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---- 64 channels on planes of 8 x 8 tiles (kind 5 with H, W multiples of 8: layer1 of 64^3 crops): direct3_wgrad_kernel<false, 64, true> ----
// 28 chains of tiles per (dz, dy) pair like the 8 x 8 form (same slab layout, mi_direct3_wgrad_slab_bytes / _splits)
int mi_direct3t_wgrad_launch(const float* x, const float* dy, float* slabs, int N, int D, int H, int W, hipStream_t s) {
    if (H % 8 || W % 8 || H < 8 || W < 8) return MI_E_UNSUPPORTED;
    Direct3WgradParams p = {};
    p.x[0] = x; p.dy[0] = dy; p.slabs[0] = slabs; p.N = N; p.D = D; p.H = H; p.W = W;
    p.bytes = (unsigned)(4l * N * D * H * W * C);
    p.nprob = 1; p.splits = WG_SPLITS;
    hipLaunchKernelGGL((direct3_wgrad_kernel<false, 64, true>), dim3(9 * WG_SPLITS), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---- 64^3 crops: layer2 (128 -> 128 on 8 x 8 planes, kind 3), single launches: 36 groups x 7 chains of planes; slabs of [27][CT][CT] floats ----
int mi_direct3x_wgrad_splits(int kind) { return kind == 3 ? 7 : 0; }
size_t mi_direct3x_wgrad_slab_bytes(int kind) {
    const size_t ct = kind == 3 ? 128 : 0;
    return sizeof(float) * (size_t)mi_direct3x_wgrad_splits(kind) * NTAP * ct * ct;
}
int mi_direct3x_wgrad_launch(int kind, const float* x, const float* dy, float* slabs, int N, int D, hipStream_t s) {
    Direct3WgradParams p = {};
    p.x[0] = x; p.dy[0] = dy; p.slabs[0] = slabs; p.nprob = 1; p.splits = mi_direct3x_wgrad_splits(kind);
    if (kind == 3) {
        p.N = N; p.D = D; p.bytes = (unsigned)(4l * N * D * PLANE * 128);
        hipLaunchKernelGGL((direct3_wgrad_kernel<false, 128>), dim3(36 * p.splits), dim3(512), 0, s, p);
    } else return MI_E_UNSUPPORTED;
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// ---- C-ABI (include/cetpick_hip.h): the image kept by the caller across calls ----
int mi_direct3_finish_slabs(const float* slabs, int n_slabs, long out_elems, float* out, const float* res, const float* mask,
                            int relu, hipStream_t s);       // conv_igemm.hip: split-K reduce + epilogue

extern "C" size_t mi_conv3d_direct_wimg_bytes(int channels) { return mi_direct3_wimg_bytes(channels); }
extern "C" size_t mi_conv3d_direct_workspace_bytes(int N, int channels) { return mi_direct3_slab_bytes(N, channels); }

extern "C" int mi_conv3d_direct_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
    const int kind = mi_direct3_kind(N, Di, Hi, Wi, Ci, Co, k, k, k, stride, pad, pad, pad, 1, 1, 1);
    return kind <= 2 ? kind : 0;           // (kinds 3 and 5 cut their image per call inside mi_conv3d_* / mi_convnd_*)
}

extern "C" int mi_conv3d_direct_prep(const void* const* w, void* const* img, const int* dgrad, const int* channels, int n,
                                     mi_stream_t stream) {
    if (!w || !img || !dgrad || !channels || n < 0) return MI_E_ARG;
    if (n == 0) return MI_OK;
    return mi_direct3_prep(reinterpret_cast<const float* const*>(w), img, dgrad, channels, n, (hipStream_t)stream);
}

extern "C" int mi_conv3d_direct_f32(const float* a, const void* wimg, float* out, const float* res, const float* mask,
                                    int relu, int N, int Di, int Hi, int Wi, int channels, void* ws, size_t ws_bytes,
                                    mi_stream_t stream) {
    if (!a || !wimg || !out) return MI_E_ARG;
    const int kind = mi_direct3_kind(N, Di, Hi, Wi, channels, channels, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1);
    if (kind == 1) return mi_direct3_launch(a, wimg, out, res, mask, relu, N, Di, (hipStream_t)stream);
    if (kind == 2) return mi_direct3s_launch(a, wimg, out, res, mask, relu, N, (hipStream_t)stream);
    return MI_E_UNSUPPORTED;
}
