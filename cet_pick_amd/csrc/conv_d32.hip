// Patch-resident direct convolution for the detector's 32-output-channel layers, forward / inference (round 5).
//
// Replaces (reference, cet_pick/...): the Conv2d(32|64, 32, 3, padding=1) + BatchNorm2d + ReLU layers of the 256 x 256 level of
// models/networks/unet.py:198-249,319-399 (BatchNorm folded into weights + bias at inference, hipops.conv_bn) and the two
// Conv3d(32, 32, 3, padding=(1,4,4), dilation=(1,4,4)) + ReLU of the feature head, models/networks/unet_small.py:52-60.
// These are 11.7 ms of the 30.6 ms the convolutions of a 128 x 512 x 512 forward take, at 77 - 120 TFLOP/s on the implicit
// GEMM's 128 x 32 tile (profiles/r04_unet_layers.txt): nine or 27 one-tap slices per tile, each re-gathering, re-cutting and
// re-staging its 128 im2col rows.  Here (the scheme of conv_direct3.hip's direct3h_kernel, for 32 output channels):
//   * a workgroup owns a tile of outputs and stages its input PATCH once - tile + halo, every input channel, cut exactly into
//     three bf16 planes while it is staged (16-byte records per voxel and 8-channel half: an MFMA A fragment of any tap is ONE
//     ds_read_b128 at lane base + immediate) - and then runs all taps from LDS;
//   * the dilated head (dilation 4 in x and y) is 16 interleaved sub-images that never mix: a tile is 8 x 8 outputs of ONE
//     (x mod 4, y mod 4) class on two z-planes, its patch 10 x 10 x 4 voxels of that class - so a tap is again a plain record
//     offset.  The 2-D layers (D = 1, dilation 1) take 16 x 16 tiles with an 18 x 18 patch;
//   * weights come pre-cut as B fragments ([channel chunk][tap][plane][lane] x 16 bytes, mi_conv_d32_prep; 32 output columns =
//     one fragment per k-step) and stream from L2 through a six-deep register ring, as in the direct3 kernels;
//   * epilogue: + bias (folded BatchNorm), ReLU, 128-byte rows of 32 channels.
// f32-equivalent bf16x3 arithmetic (six products of the exact three-way cut, f32 accumulation), as every convolution here.
#include "common.h"
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DCO = 32;                     // output channels of the 32-column form (CO = 64: two column halves per wave)
constexpr int DW_BLK = 1024;                // one B fragment plane: 64 lanes x 16 bytes
constexpr int DW_STEP = 3 * DW_BLK;         // bytes per k-step and 32-column half of the weight image
constexpr int DRB = 6;                      // weight k-steps in flight (32 columns; 64 columns: 3 - the ring is 2 x 3 fragments per k-step)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t d_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void d_cut8(const float (&v)[8], u32x4 (&o)[3]) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        u0[t] = __float_as_uint(v[t]);
        const float r1 = v[t] - __uint_as_float(u0[t] & 0xffff0000u);
        u1[t] = __float_as_uint(r1);
        u2[t] = __float_as_uint(r1 - __uint_as_float(u1[t] & 0xffff0000u));
    }
    constexpr unsigned HI2 = 0x07060302u;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        o[0][d] = __builtin_amdgcn_perm(u0[2 * d + 1], u0[2 * d], HI2);
        o[1][d] = __builtin_amdgcn_perm(u1[2 * d + 1], u1[2 * d], HI2);
        o[2][d] = __builtin_amdgcn_perm(u2[2 * d + 1], u2[2 * d], HI2);
    }
}

struct D32Params {
    const float* x;           // (N, D, H, W, CIN) channels-last
    const unsigned char* wimg;
    const float* bias;        // [32] or null
    float* out;               // (N, D, H, W, 32)
    int relu;
    int N, D, H, W;
    unsigned x_bytes, out_bytes, w_bytes;
    int co_total;             // channels of an output voxel in memory (CO, or a multiple of 64: blockIdx.y = 64-column block, one image each)
    // up-convolution epilogue (1 x 1 products only; up_co > 0): column j = (a 2 + b) up_co + co of input voxel (y, x) goes to
    // out[n][2 y + a][2 x + b][co] of an (N, up_ho, up_wo, up_stride)-channel tensor as relu(scale[co] acc + shift[co]) - the pixel shuffle,
    // the folded BatchNorm + bias and the ReLU of mi_upconv_tail_fwd, written straight into the concatenation's first up_co channels
    int up_co, up_ho, up_wo, up_stride;
    const float* up_scale;
    const float* up_shift;
    // 2 x 2 max-pool of the output as a by-product (2-D 3 x 3 forms; H, W even): pool (N, D, H / 2, W / 2, co_total).  A lane's sixteen
    // accumulators of a row block are a 4 (y) x 4 (x) patch of ONE column - the four windows of the patch never leave the lane.
    float* pool;
    unsigned pool_bytes;
    int pool_stride;          // channels of a pooled voxel in memory (co_total may be the stride of a wider buffer `out` is a channel slice of)
};

// CIN: 16, 32 or 64 input channels.  NZT: 1 (2-D layers: D planes are independent images) or 3 z taps.  DIL: xy dilation (1 or 4).
// Tile of a workgroup (4 waves), in voxels of ONE dilation class: TX x TY x TZ = 16 x 16 x 1 (two 4 x 8 row blocks per wave) for
// NZT = 1, 8 x 8 x 2 (one row block per wave) for NZT = 3.
// Round 5, CO = 64 (the 128 x 128 level of the U-Net: 32 / 64 / 128 -> 64, 2-D): every A fragment feeds TWO column halves (12 MFMAs per
// row block and k-step); TYT = 8 halves the tile (16 x 8, one row block per wave) where the patch of 128 input channels would not fit.
// KSZ = 1: the 1 x 1 products (the transposed convolutions' 4 Co columns, the last layer): the patch is the tile, one tap per chunk.
template <int CIN, int NZT, int DIL, int CO = 32, int TYT = 16, int KSZ = 3>
struct D32Cfg {
    static constexpr int TX = NZT == 1 ? 16 : 8, TY = NZT == 1 ? TYT : 8, TZ = NZT == 1 ? 1 : 2;
    static constexpr int BPW = NZT == 1 ? TYT / 8 : 1;        // row blocks (4 y x 8 x) per wave
    static constexpr int NCH = CO / 32;                        // 32-column halves
    static constexpr int RB = CO == 32 ? DRB : 3;              // weight k-steps in flight
    static constexpr int WSTEP = NCH * DW_STEP;                // bytes per k-step of the weight image: [column half][plane][lane] x 16
    static constexpr int PX = TX + KSZ - 1, PY = TY + KSZ - 1, PZ = TZ + NZT - 1;
    static constexpr int NV = PX * PY * PZ;                    // patch voxels: 324 / 400
    static constexpr int ARR = NV * 16;                        // one (chunk, plane, k-half) array
    static constexpr int PL = 2 * ARR, CH = 3 * PL;            // plane, chunk
    static constexpr int KS = CIN / 16;                        // 16-channel chunks of the reduction
    static constexpr int NPH = CIN > 128 ? CIN / 128 : 1;      // phases: the patch of 128 channels at a time (256 input channels: two)
    static constexpr int KSP = KS / NPH;                       // chunks resident per phase
    static constexpr int LDS = KSP * CH;                        // 62,208 (2-D, 32 ch) / 124,416 (2-D, 64 ch) / 76,800 (head)
    static constexpr int NTAP = KSZ * KSZ * NZT;
    static constexpr int NSTEP = KS * NTAP;
    static constexpr int UNITS = (NV * KSP * 2 + 255) / 256;   // staging units (voxel, chunk, k-half) per thread and phase
};

template <int CIN, int NZT, int DIL, int CO = 32, int TYT = 16, int KSZ = 3>
__global__ __launch_bounds__(256, (D32Cfg<CIN, NZT, DIL, CO, TYT, KSZ>::LDS <= 80 * 1024) ? 2 : 1) void conv_d32_kernel(D32Params p) {
    typedef D32Cfg<CIN, NZT, DIL, CO, TYT, KSZ> G;
    constexpr int HALO = (KSZ - 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char patch[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    // tile index -> (image n, z tile, dilation class, tile y, tile x)
    const int cw = p.W / DIL, chh = p.H / DIL;                   // extents of a class sub-image
    const int txn = cw / G::TX, tyn = chh / G::TY, tzn = p.D / G::TZ;
    int bi = blockIdx.x;
    const int tx = bi % txn; bi /= txn;
    const int ty = bi % tyn; bi /= tyn;
    const int cls = bi % (DIL * DIL); bi /= (DIL * DIL);
    const int tz = bi % tzn, n = bi / tzn;
    const int cx = cls % DIL, cy = cls / DIL;
    const int x0 = tx * G::TX, y0 = ty * G::TY, z0 = tz * G::TZ; // tile origin in class coordinates (x, y) / planes (z)

    // 64-column blocks of a wider layer, an image each: a workgroup per block (blockIdx.y) - or, for the 1 x 1 products (LOOPCB), every
    // block in turn over the ONE staged tile (the tile is all a 1 x 1 product reads: staging it once per block made the wide ones slower)
    constexpr bool LOOPCB = KSZ == 1 && G::NPH == 1 && CO == 64;
    const int ncb = LOOPCB ? p.co_total / CO : 1;
    int co0 = blockIdx.y * CO;
    __amdgpu_buffer_rsrc_t wrs = d_rsrc(p.wimg + (size_t)blockIdx.y * p.w_bytes, p.w_bytes);
    bf16x8 bfr[G::RB][G::NCH][3];
    auto wload = [&](int g, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value % G::RB;
#pragma unroll
        for (int ch = 0; ch < G::NCH; ++ch)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bfr[SLOT][ch][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    wrs, g < G::NSTEP ? lane * 16 + (ch * 3 + pl) * DW_BLK : (int)0x80000000u, g < G::NSTEP ? g * G::WSTEP : 0, 0));
    };
    auto wload_dyn = [&](int g) {
        switch (g % G::RB) {
            case 0: wload(g, std::integral_constant<int, 0>{}); break;
            case 1: wload(g, std::integral_constant<int, 1>{}); break;
            case 2: wload(g, std::integral_constant<int, 2>{}); break;
            case 3: wload(g, std::integral_constant<int, 3>{}); break;
            case 4: wload(g, std::integral_constant<int, 4>{}); break;
            default: wload(g, std::integral_constant<int, 5>{}); break;
        }
    };
#pragma unroll
    for (int g = 0; g < G::RB - 1; ++g) wload_dyn(g);

    // ---- the patch, every chunk of a phase at once: unit q = (voxel, chunk, k-half); voxels outside the volume read zeros (the padding).
    //      (256 input channels: two phases of 128 - the second patch overwrites the first behind a barrier, the accumulators stay) ----
    auto stage = [&](int ph) {
        const __amdgpu_buffer_rsrc_t xrs = d_rsrc(p.x, p.x_bytes);
        u32x4 ld[G::UNITS][2];
        int dst[G::UNITS];
#pragma unroll
        for (int u = 0; u < G::UNITS; ++u) {
            const int q = tid + 256 * u;
            const int hh = q & 1, c = (q >> 1) % G::KSP, vox = (q >> 1) / G::KSP;
            const int pz = vox / (G::PX * G::PY), py = (vox / G::PX) % G::PY, px = vox % G::PX;
            const int z = z0 + pz - (NZT - 1) / 2, y = (y0 + py - HALO) * DIL + cy, x = (x0 + px - HALO) * DIL + cx;
            const bool ok = vox < G::NV && (unsigned)z < (unsigned)p.D && y0 + py - HALO >= 0 && y0 + py - HALO < chh && x0 + px - HALO >= 0 &&
                            x0 + px - HALO < cw;
            const unsigned off = ok ? 4u * (unsigned)(((((long)n * p.D + z) * p.H + y) * p.W + x) * CIN + (ph * G::KSP + c) * 16 + hh * 8) : 0x80000000u;
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 16, 0);
            dst[u] = vox < G::NV ? c * G::CH + hh * G::ARR + vox * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < G::UNITS; ++u) {
            if (dst[u] < 0) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            d_cut8(v, o);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(patch + dst[u] + pl * G::PL) = o[pl];
        }
    };
    stage(0);
    // ---- per-lane geometry: row block b of this wave = 4 y x 8 x outputs; MFMA row l32 = (y & 3, x); record of the tap (0, 0, 0) ----
    int vbase[G::BPW];
    int by_[G::BPW], bx_[G::BPW], bz_[G::BPW];
#pragma unroll
    for (int i = 0; i < G::BPW; ++i) {
        const int b = wave * G::BPW + i;
        if (NZT == 1) { bz_[i] = 0; by_[i] = 4 * (b >> 1); bx_[i] = 8 * (b & 1); }
        else { bz_[i] = b >> 1; by_[i] = 4 * (b & 1); bx_[i] = 0; }
        vbase[i] = ((bz_[i] * G::PY + by_[i] + (l32 >> 3)) * G::PX + bx_[i] + (l32 & 7)) * 16 + h * G::ARR;
    }
    f32x16 acc[G::BPW][G::NCH];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    __syncthreads();

    bf16x8 af[2][G::BPW][3];
    auto frags = [&](int c, int tap, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int tzz = tap / (KSZ * KSZ), tyy = (tap / KSZ) % KSZ, txx = tap % KSZ;
        const int imm = ((tzz * G::PY + tyy) * G::PX + txx) * 16 + c * G::CH;
#pragma unroll
        for (int i = 0; i < G::BPW; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[SET][i][pl] = *reinterpret_cast<const bf16x8*>(patch + vbase[i] + imm + pl * G::PL);
    };
    auto frags_dyn = [&](int g, int c, int tap) {
        if (g & 1) frags(c, tap, std::integral_constant<int, 1>{});
        else frags(c, tap, std::integral_constant<int, 0>{});
    };
    const __amdgpu_buffer_rsrc_t ors = d_rsrc(p.out, p.out_bytes);
  for (int cb = 0; cb < ncb; ++cb) {
    if (LOOPCB && cb > 0) {                                      // next 64-column block: its image, its ring
        co0 = cb * CO;
        wrs = d_rsrc(p.wimg + (size_t)cb * p.w_bytes, p.w_bytes);
#pragma unroll
        for (int g = 0; g < G::RB - 1; ++g) wload_dyn(g);
    }
#pragma unroll
    for (int i = 0; i < G::BPW; ++i)
#pragma unroll
        for (int ch = 0; ch < G::NCH; ++ch)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][ch][r] = 0.f;
    frags_dyn(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < G::KS; ++c) {
        if (G::NPH > 1 && c > 0 && c % G::KSP == 0) {             // next phase: every wave is done with the resident patch
            __syncthreads();
            stage(c / G::KSP);
            __syncthreads();
            frags_dyn(c * G::NTAP, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tap = 0; tap < G::NTAP; ++tap) {
            const int g = c * G::NTAP + tap;
            wload_dyn(g + G::RB - 1);
            // (fragments of the next k-step - unless it belongs to the next phase's patch)
            if (g + 1 < G::NSTEP && ((g + 1) / G::NTAP) / G::KSP == c / G::KSP) frags_dyn(g + 1, ((g + 1) / G::NTAP) % G::KSP, (g + 1) % G::NTAP);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int i = 0; i < G::BPW; ++i)
#pragma unroll
                    for (int ch = 0; ch < G::NCH; ++ch)
                        acc[i][ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][i][PA[pr]], bfr[g % G::RB][ch][PB[pr]], acc[i][ch], 0, 0, 0);
            // issue order inside the k-step: every load behind an MFMA
#pragma unroll
            for (int k = 0; k < 3 * G::BPW; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 3 * G::NCH; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            constexpr int REST = 6 * G::BPW * G::NCH - 3 * G::BPW - 3 * G::NCH;
            if constexpr (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h = (y & 3, x) of the row block ----
#pragma unroll
    for (int ch = 0; ch < G::NCH; ++ch) {
        if (KSZ == 1 && p.up_co > 0) {                               // (uniform) the up-convolution's shuffled, normalised output
            const int col = co0 + ch * 32 + l32, ab = col / p.up_co, co = col - ab * p.up_co;
            const float sc = p.up_scale[co], sh = p.up_shift[co];
#pragma unroll
            for (int i = 0; i < G::BPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int yo = 2 * (y0 + by_[i] + (m >> 3)) + (ab >> 1), xo = 2 * (x0 + bx_[i] + (m & 7)) + (ab & 1);
                    const float v = fmaxf(fmaf(acc[i][ch][r], sc, sh), 0.f);
                    const unsigned off = (yo < p.up_ho && xo < p.up_wo)
                                             ? 4u * (unsigned)(((((long)n * p.D + z0) * p.up_ho + yo) * p.up_wo + xo) * p.up_stride + co) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, (int)off, 0, 0);
                }
            continue;
        }
        const float bv = p.bias ? p.bias[co0 + ch * 32 + l32] : 0.f;
#pragma unroll
        for (int i = 0; i < G::BPW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int y = (y0 + by_[i] + (m >> 3)) * DIL + cy, x = (x0 + bx_[i] + (m & 7)) * DIL + cx, z = z0 + bz_[i];
                float v = acc[i][ch][r] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                acc[i][ch][r] = v;
                const unsigned off = 4u * (unsigned)(((((long)n * p.D + z) * p.H + y) * p.W + x) * p.co_total + co0 + ch * 32 + l32);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, (int)off, 0, 0);
            }
            if (NZT == 1 && DIL == 1 && KSZ == 3 && p.pool) {          // (uniform) the 2 x 2 windows of this lane's 4 x 4 patch
                const __amdgpu_buffer_rsrc_t prs = d_rsrc(p.pool, p.pool_bytes);
                const int hp = p.H >> 1, wp = p.W >> 1;
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int r0 = 8 * py + 2 * px;              // registers r = 4 y_local + x_local: (2 py, 2 px) .. (2 py + 1, 2 px + 1)
                        const float v = fmaxf(fmaxf(acc[i][ch][r0], acc[i][ch][r0 + 1]), fmaxf(acc[i][ch][r0 + 4], acc[i][ch][r0 + 5]));
                        const int yo = ((y0 + by_[i]) >> 1) + py, xo = ((x0 + bx_[i] + 4 * h) >> 1) + px;
                        const unsigned off = 4u * (unsigned)(((((long)n * p.D + z0) * hp + yo) * wp + xo) * p.pool_stride + co0 + ch * 32 + l32);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), prs, (int)off, 0, 0);
                    }
            }
        }
    }
  }
}

// weight image: W[tap][ci][co = 32] f32 -> bf16x3 B fragments [chunk][tap][plane][lane] x 16 bytes; idx = (chunk, tap, lane)
// (co = 64: [chunk][tap][column half][plane][lane]; idx = (chunk, tap, column half, lane))
// (co = 64 b: b blocks of 64 columns (blockIdx.y), each [chunk][tap][column half][plane][lane]; idx = (chunk, tap, column half, lane))
__global__ __launch_bounds__(256) void conv_d32_prep_kernel(const float* w, unsigned char* img, int cin, int ntap, int co) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int cob = co > 64 ? 64 : co;                   // columns of one image
    const int ks = cin / 16, nch = cob / 32;
    if (idx >= ks * ntap * nch * 64) return;
    const int lane = idx & 63, ch = (idx >> 6) % nch, tap = ((idx >> 6) / nch) % ntap, c = (idx >> 6) / (nch * ntap);
    const int nn = blockIdx.y * cob + ch * 32 + (lane & 31), k0 = c * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[((long)tap * cin + k0 + e) * co + nn];
    u32x4 o[3];
    d_cut8(v, o);
    unsigned char* dst = img + (size_t)blockIdx.y * ((size_t)ks * ntap * nch * DW_STEP) + (size_t)((c * ntap + tap) * nch + ch) * DW_STEP + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * DW_BLK) = o[pl];
}

}  // namespace

// 1: 2-D 3 x 3 (kd = 1, any D: independent planes), dilation 1; 2: 3 x 3 x 3 with dilation (1, 4, 4); 0: not a shape of this kernel
extern "C" int mi_conv_d32_kind(int N, int D, int H, int W, int Ci, int Co, int kd, int kh, int kw, int dd, int dh, int dw) {
    if (getenv("MI_NO_D32")) return 0;
    // 4 (round 5): 1 x 1 products, 32 or 64 / 128 / 256 / 512 output columns from 32 / 64 / 128 / 256 input channels (MI_NO_D32_1X1=1: off)
    if (kd == 1 && kh == 1 && kw == 1 && dd == 1 && dh == 1 && dw == 1 && N >= 1 && (Co == 32 || (Co % 64 == 0 && Co >= 64 && Co <= 512)) &&
        (Ci == 32 || Ci == 64 || Ci == 128 || (Ci == 256 && getenv("MI_D32_1X1_256"))) && H % 8 == 0 && W % 16 == 0 &&
        4l * N * D * H * W * (Ci > Co ? Ci : Co) < 0x7fff0000l && !getenv("MI_NO_D32_1X1"))
        return 4;       // (256 input channels: two phases and a workgroup per column block - 0.405 against 0.370 ms on the implicit GEMM: opt-in)
    // 3 (round 5): 2-D 3 x 3 to 64 output channels from 32 / 64 / 128 (the 128 x 128 level); MI_NO_D64=1: the implicit GEMM
    // (Co = 128 / 256: 64-column blocks, a workgroup each, the patch staged once per block)
    if ((Co == 64 || Co == 128 || Co == 256) && N >= 1 && kd == 1 && kh == 3 && kw == 3 && dd == 1 && dh == 1 && dw == 1 &&
        (Ci == 32 || Ci == 64 || Ci == 128 || Ci == 256) && H % 16 == 0 && W % 16 == 0 && 4l * N * D * H * W * (Ci > Co ? Ci : Co) < 0x7fff0000l &&
        !getenv("MI_NO_D64") && (Co == 64 || !getenv("MI_NO_D64_WIDE")))
        return 3;
    if (Co != DCO || N < 1 || kh != 3 || kw != 3 || dd != 1) return 0;
    if (4l * N * D * H * W * (Ci > DCO ? Ci : DCO) >= 0x7fff0000l) return 0;
    if (kd == 1 && dh == 1 && dw == 1 && (Ci == 16 || Ci == 32 || Ci == 64) && H % 16 == 0 && W % 16 == 0) return 1;
    if (kd == 3 && dh == 4 && dw == 4 && Ci == 32 && H % 32 == 0 && W % 32 == 0 && D % 2 == 0) return 2;
    return 0;
}
extern "C" size_t mi_conv_d32_image_bytes(int Ci, int ntap) { return (size_t)(Ci / 16) * ntap * DW_STEP; }

// w: [tap][Ci][32] f32 (kernel layout); img: mi_conv_d32_image_bytes(Ci, ntap) bytes
extern "C" int mi_conv_d32_prep(const float* w, void* img, int Ci, int ntap, mi_stream_t stream) {
    if (!w || !img || (Ci != 16 && Ci != 32 && Ci != 64 && Ci != 128 && Ci != 256) || (ntap != 9 && ntap != 27 && ntap != 1)) return MI_E_ARG;
    const int n = (Ci / 16) * ntap * 64;
    hipLaunchKernelGGL(conv_d32_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)img, Ci, ntap, DCO);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// the 64-output-channel form (kind 3): image of [chunk][tap][column half][plane][lane] x 16 bytes
extern "C" size_t mi_conv_d64_image_bytes(int Ci, int ntap) { return (size_t)(Ci / 16) * ntap * 2 * DW_STEP; }
// w: [tap][Ci][Co] with Co = 64, 128 or 256; img: (Co / 64) x mi_conv_d64_image_bytes(Ci, ntap) bytes (one image per 64-column block)
extern "C" int mi_conv_d64_prep_co(const float* w, void* img, int Ci, int Co, int ntap, mi_stream_t stream) {
    if (!w || !img || (Ci != 32 && Ci != 64 && Ci != 128 && Ci != 256) || Co % 64 || Co < 64 || Co > 512 || (ntap != 9 && ntap != 1)) return MI_E_ARG;
    const int n = (Ci / 16) * ntap * 2 * 64;
    hipLaunchKernelGGL(conv_d32_prep_kernel, dim3((n + 255) / 256, Co / 64), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)img, Ci, ntap, Co);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_conv_d64_prep(const float* w, void* img, int Ci, int ntap, mi_stream_t stream) {
    if (!w || !img || (Ci != 32 && Ci != 64 && Ci != 128) || ntap != 9) return MI_E_ARG;
    const int n = (Ci / 16) * ntap * 2 * 64;
    hipLaunchKernelGGL(conv_d32_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)img, Ci, ntap, 64);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_conv_d64_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                                   int Ci, int Co, mi_stream_t stream);
// y = act(conv(x, W) + bias): x (N, D, H, W, Ci) channels-last, y (N, D, H, W, 32); `kind` as mi_conv_d32_kind returns it
extern "C" int mi_conv_d32_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                                   int Ci, int kind, mi_stream_t stream) {
    if (!x || !wimg || !y || N < 1) return MI_E_ARG;
    D32Params p = {};
    p.x = x; p.wimg = (const unsigned char*)wimg; p.bias = bias; p.out = y; p.relu = relu;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x_bytes = (unsigned)(4l * N * D * H * W * Ci);
    p.out_bytes = (unsigned)(4l * N * D * H * W * DCO);
    p.co_total = DCO;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 1 && (Ci == 16 || Ci == 32 || Ci == 64) && H % 16 == 0 && W % 16 == 0) {
        p.w_bytes = (unsigned)mi_conv_d32_image_bytes(Ci, 9);
        const long grid = (long)N * D * (H / 16) * (W / 16);
        if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
        if (Ci == 16) hipLaunchKernelGGL((conv_d32_kernel<16, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);      // (the 16 -> 32 layer: one chunk, 9 k-steps)
        else if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);
    } else if (kind == 2 && Ci == 32 && H % 32 == 0 && W % 32 == 0 && D % 2 == 0) {
        p.w_bytes = (unsigned)mi_conv_d32_image_bytes(Ci, 27);
        const long grid = (long)N * (D / 2) * 16 * (H / 32) * (W / 32);
        if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
        hipLaunchKernelGGL((conv_d32_kernel<32, 3, 4>), dim3((unsigned)grid), dim3(256), 0, s, p);
    } else if (kind == 3) {
        return mi_conv_d64_fwd_f32(x, wimg, bias, y, relu, N, D, H, W, Ci, 64, stream);
    } else return MI_E_UNSUPPORTED;
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// kind 4: y = act(x W + bias) per voxel (1 x 1): Co = 32 (image: mi_conv_d32_prep(w, img, Ci, 1)) or a multiple of 64 up to 512
// (mi_conv_d64_prep_co(w, img, Ci, Co, 1): Co / 64 images); 16 x 8 tiles, the patch is the tile
extern "C" int mi_conv_d32_1x1_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                                       int Ci, int Co, mi_stream_t stream) {
    if (!x || !wimg || !y || N < 1) return MI_E_ARG;
    if ((Ci != 32 && Ci != 64 && Ci != 128 && Ci != 256) || !(Co == 32 || (Co % 64 == 0 && Co >= 64 && Co <= 512)) || H % 8 || W % 16)
        return MI_E_UNSUPPORTED;
    D32Params p = {};
    p.x = x; p.wimg = (const unsigned char*)wimg; p.bias = bias; p.out = y; p.relu = relu;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x_bytes = (unsigned)(4l * N * D * H * W * Ci);
    p.out_bytes = (unsigned)(4l * N * D * H * W * Co);
    p.co_total = Co;
    const long grid = (long)N * D * (H / 8) * (W / 16);
    if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (Co == 32) {
        p.w_bytes = (unsigned)mi_conv_d32_image_bytes(Ci, 1);
        const dim3 g((unsigned)grid);
        if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1, 32, 8, 1>), g, dim3(256), 0, s, p);
        else if (Ci == 64) hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1, 32, 8, 1>), g, dim3(256), 0, s, p);
        else if (Ci == 128) hipLaunchKernelGGL((conv_d32_kernel<128, 1, 1, 32, 8, 1>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_d32_kernel<256, 1, 1, 32, 8, 1>), g, dim3(256), 0, s, p);
    } else {
        p.w_bytes = (unsigned)mi_conv_d64_image_bytes(Ci, 1);
        const dim3 g((unsigned)grid, (unsigned)(Ci <= 128 ? 1 : Co / 64));       // (<= 128 input channels: the workgroup loops over the blocks)
        if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
        else if (Ci == 64) hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
        else if (Ci == 128) hipLaunchKernelGGL((conv_d32_kernel<128, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_d32_kernel<256, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// The 2 x 2 / stride-2 transposed convolution of an up-convolution block at inference (unet.py:251-317,319-399) in ONE launch: the 1 x 1
// product to 4 Co columns (kind 4; wimg = mi_conv_d64_prep_co(w, img, Ci, 4 Co, 1)) with the pixel shuffle, scale / shift (evaluation-mode
// BatchNorm with the bias folded in) and ReLU in its epilogue, written into out (N, Ho, Wo, cstride)[..., 0:Co] - the concatenation buffer
// whose channels Co.. the caller fills with the encoder feature.  x (N, H, W, Ci); Ho in {2 H - 1, 2 H}, Wo likewise.
extern "C" int mi_conv_d32_upconv_fwd_f32(const float* x, const void* wimg, const float* scale, const float* shift, float* out, int N,
                                          int H, int W, int Ci, int Co, int Ho, int Wo, int cstride, mi_stream_t stream) {
    if (!x || !wimg || !scale || !shift || !out || N < 1) return MI_E_ARG;
    const int C4 = 4 * Co;
    if ((Ci != 32 && Ci != 64 && Ci != 128) || C4 % 64 || C4 < 64 || C4 > 512 || Co % 32 || H % 8 || W % 16) return MI_E_UNSUPPORTED;
    if (Ho > 2 * H || Ho < 2 * H - 1 || Wo > 2 * W || Wo < 2 * W - 1 || cstride < Co) return MI_E_ARG;
    if (4l * N * Ho * Wo * cstride >= 0x7fff0000l || 4l * N * H * W * Ci >= 0x7fff0000l) return MI_E_UNSUPPORTED;
    D32Params p = {};
    p.x = x; p.wimg = (const unsigned char*)wimg; p.bias = nullptr; p.out = out; p.relu = 1;
    p.N = N; p.D = 1; p.H = H; p.W = W;
    p.x_bytes = (unsigned)(4l * N * H * W * Ci);
    p.out_bytes = (unsigned)(4l * N * Ho * Wo * cstride);
    p.co_total = C4;
    p.up_co = Co; p.up_ho = Ho; p.up_wo = Wo; p.up_stride = cstride; p.up_scale = scale; p.up_shift = shift;
    p.w_bytes = (unsigned)mi_conv_d64_image_bytes(Ci, 1);
    const long grid = (long)N * (H / 8) * (W / 16);
    if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
    const dim3 g((unsigned)grid, 1);
    hipStream_t s = (hipStream_t)stream;
    if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
    else if (Ci == 64) hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_d32_kernel<128, 1, 1, 64, 8, 1>), g, dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// kinds 1 and 3 (2-D 3 x 3) with the 2 x 2 max-pool of the result as a second output (unet.py:198-249: conv -> BatchNorm -> ReLU -> MaxPool2d(2),
// the un-pooled tensor is the skip connection): y (N, D, H, W, Co) and y_pool (N, D, H / 2, W / 2, Co); Co = 32 (kind 1) or 64 / 128 / 256
// (kind 3); H, W multiples of 16.  The images are those of mi_conv_d32_prep / mi_conv_d64_prep_co.
extern "C" int mi_conv_d32_fwd_pool_strided_f32(const float* x, const void* wimg, const float* bias, float* y, int y_cstride, float* y_pool,
                                                int relu, int N, int D, int H, int W, int Ci, int Co, mi_stream_t stream);
extern "C" int mi_conv_d32_fwd_pool_f32(const float* x, const void* wimg, const float* bias, float* y, float* y_pool, int relu, int N, int D,
                                        int H, int W, int Ci, int Co, mi_stream_t stream) {
    return mi_conv_d32_fwd_pool_strided_f32(x, wimg, bias, y, Co, y_pool, relu, N, D, H, W, Ci, Co, stream);
}
// ... with y a CHANNEL SLICE of a wider tensor: voxel v's Co channels at y + v * y_cstride (y_cstride >= Co, a multiple of 4) - the skip
// connection written straight into the concatenation buffer of the up-convolution block that consumes it (torch.cat((up, enc), 1),
// unet.py:392): y = cat + Co_up, y_cstride = Co_up + Co.  y_pool stays dense (N, D, H / 2, W / 2, Co).
extern "C" int mi_conv_d32_fwd_pool_strided_f32(const float* x, const void* wimg, const float* bias, float* y, int y_cstride, float* y_pool,
                                                int relu, int N, int D, int H, int W, int Ci, int Co, mi_stream_t stream) {
    if (!x || !wimg || !y || !y_pool || N < 1 || y_cstride < Co || (y_cstride & 3)) return MI_E_ARG;
    if (H % 16 || W % 16) return MI_E_UNSUPPORTED;
    if (4l * N * D * H * W * y_cstride >= 0x7fff0000l) return MI_E_UNSUPPORTED;
    D32Params p = {};
    p.x = x; p.wimg = (const unsigned char*)wimg; p.bias = bias; p.out = y; p.relu = relu;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x_bytes = (unsigned)(4l * N * D * H * W * Ci);
    p.out_bytes = (unsigned)(4l * (((long)N * D * H * W - 1) * y_cstride + Co));
    p.co_total = y_cstride;
    p.pool = y_pool; p.pool_bytes = (unsigned)(4l * N * D * (H / 2) * (W / 2) * Co); p.pool_stride = Co;
    hipStream_t s = (hipStream_t)stream;
    if (Co == 32 && (Ci == 16 || Ci == 32 || Ci == 64)) {
        p.w_bytes = (unsigned)mi_conv_d32_image_bytes(Ci, 9);
        const long grid = (long)N * D * (H / 16) * (W / 16);
        if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
        if (Ci == 16) hipLaunchKernelGGL((conv_d32_kernel<16, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);
        else if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p);
    } else if ((Co == 64 || Co == 128 || Co == 256) && (Ci == 32 || Ci == 64 || Ci == 128 || Ci == 256)) {
        p.w_bytes = (unsigned)mi_conv_d64_image_bytes(Ci, 9);
        const long grid = (long)N * D * (H / (Ci >= 128 ? 8 : 16)) * (W / 16);
        if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
        const dim3 g((unsigned)grid, (unsigned)(Co / 64));
        if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1, 64, 16>), g, dim3(256), 0, s, p);
        else if (Ci == 64) hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1, 64, 16>), g, dim3(256), 0, s, p);
        else if (Ci == 128) hipLaunchKernelGGL((conv_d32_kernel<128, 1, 1, 64, 8>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_d32_kernel<256, 1, 1, 64, 8>), g, dim3(256), 0, s, p);
    } else return MI_E_UNSUPPORTED;
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// kind 3 with Co = 64, 128 or 256: y (N, D, H, W, Co); wimg from mi_conv_d64_prep_co (Co / 64 images); a workgroup per tile and 64-column block
extern "C" int mi_conv_d64_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                                   int Ci, int Co, mi_stream_t stream) {
    if (!x || !wimg || !y || N < 1) return MI_E_ARG;
    if ((Ci != 32 && Ci != 64 && Ci != 128 && Ci != 256) || (Co != 64 && Co != 128 && Co != 256) || H % 16 || W % 16) return MI_E_UNSUPPORTED;
    D32Params p = {};
    p.x = x; p.wimg = (const unsigned char*)wimg; p.bias = bias; p.out = y; p.relu = relu;
    p.N = N; p.D = D; p.H = H; p.W = W;
    p.x_bytes = (unsigned)(4l * N * D * H * W * Ci);
    p.out_bytes = (unsigned)(4l * N * D * H * W * Co);
    p.co_total = Co;
    p.w_bytes = (unsigned)mi_conv_d64_image_bytes(Ci, 9);
    const long grid = (long)N * D * (H / (Ci >= 128 ? 8 : 16)) * (W / 16);
    if (grid > 0x7fffffffl) return MI_E_UNSUPPORTED;
    const dim3 g((unsigned)grid, (unsigned)(Co / 64));
    hipStream_t s = (hipStream_t)stream;
    if (Ci == 32) hipLaunchKernelGGL((conv_d32_kernel<32, 1, 1, 64, 16>), g, dim3(256), 0, s, p);
    else if (Ci == 64) hipLaunchKernelGGL((conv_d32_kernel<64, 1, 1, 64, 16>), g, dim3(256), 0, s, p);
    else if (Ci == 128) hipLaunchKernelGGL((conv_d32_kernel<128, 1, 1, 64, 8>), g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_d32_kernel<256, 1, 1, 64, 8>), g, dim3(256), 0, s, p);       // (two phases of 128 input channels)
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
