// Patch-resident direct 3 x 3 convolution on small 2-D planes, forward and data gradient, training (round 6).
//
// Replaces (reference, cet_pick/...): the Conv2d(C, C, 3, padding=1) layers of the SimSiam 2-D encoder's BasicBlocks,
// models/networks/simsiam_model_2d.py:473-502 (conv1 / conv2 of every stride-1 block) inside TomoResClassifier2D.forward :776-819 -
// at --bbox 36 (docs/explore.md:67): 64 -> 64 channels on 36 x 36, 128 -> 128 on 18 x 18, 256 -> 256 on 9 x 9, batch 256 per view.
// These ran on the implicit GEMM at 135 - 155 TFLOP/s: nine one-tap slices per tile, each re-gathering, re-cutting and re-staging
// its im2col rows.  Here (the scheme of conv_direct3.hip's direct3h_kernel, in 2-D):
//   * the batch is ONE tall image of N H rows in which a zero row separates consecutive planes (padded row P = n (H + 1) + 1 + y):
//     a workgroup owns 128 consecutive output voxels of the FLAT (n, y, x) order x 64 output channels - any H, W, N tile without a
//     ragged edge (36 x 36 = 40.5 MFMA row blocks per plane) - and its patch is the padded rows those voxels touch, W + 2 wide;
//   * the patch is staged one 16-channel chunk at a time, cut exactly into three bf16 planes on the way (16-byte records per voxel
//     and 8-channel half: an MFMA A fragment of any tap is ONE ds_read_b128 at lane base + immediate); two chunks are resident
//     (chunk c + 1 is fetched, cut and stored in the shadow of chunk c's 9 x 12 MFMAs per wave), so the LDS footprint does not
//     depend on the channel count and two workgroups share a CU;
//   * weights come pre-cut as B fragments ([64-column block][chunk][tap][column half][plane][lane] x 16 bytes) and stream from L2
//     through a six-deep register ring; the data gradient is the same kernel on dY with the image built transposed and tap-flipped;
//   * epilogue: (+ residual) (ReLU) (x (mask > 0)), rows of the flat voxel order: out[o C + col].
// f32-equivalent bf16x3 arithmetic (six products of the exact three-way cut, f32 accumulation), as every convolution here.
#include "common.h"
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PW_BLK = 1024;                // one B fragment plane: 64 lanes x 16 bytes
constexpr int PW_STEP = 2 * 3 * PW_BLK;     // bytes per k-step: [column half][plane]
#ifndef P2D_RB
#define P2D_RB 6
#endif
#ifndef P2D_STAGE_TAP
#define P2D_STAGE_TAP 2
#endif
constexpr int P_RB = P2D_RB;                // weight k-steps in flight (tuning builds: -DP2D_RB=3|6|9)
constexpr int P_TM = 128;                   // output voxels per workgroup (wave tile: 64 voxels x 32 columns)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void p_cut8(const float (&v)[8], u32x4 (&o)[3]) {
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

struct P2DParams {
    const float* a;           // X (forward) or dY (data gradient): (N, H, W, CT) channels-last
    const unsigned char* wimg;
    float* out;               // (N, H, W, CT)
    const float* res;         // out = act(acc + res)          (may be null)
    const float* mask;        // out *= (mask > 0)             (may be null)
    int relu;
    int N, H;
    long total;               // N H W output voxels
    unsigned a_bytes;
    int W, nv;                // generic instance (W_ = 0): plane width and patch voxels (PR (W + 2)), from the host
};

// W_: plane width (the patch pitch W_ + 2 is a compile-time constant: tap offsets are ds_read immediates); HMIN: smallest plane
// height this instance is launched on (bounds the zero rows a tile can contain); CT: channels, in = out.
// W_ = 0 (round 6, the generic instance): width, height and patch extent are run-time values - a tap offset costs an address add instead
// of being an immediate, LDS is sized at launch - so that ANY --bbox takes the kernel (the default is 32: 32 / 16 / 8 planes), not only 36.
template <int W_, int HMIN, int CT>
struct P2DCfg {
    static constexpr bool GEN = W_ == 0;
    static constexpr int WD = GEN ? 1 : W_, HD = GEN ? 1 : HMIN;       // (divisors of the compile-time geometry)
    static constexpr int PX = W_ + 2;
    static constexpr int R = (P_TM - 1 + WD - 1) / WD + 1;              // image rows 128 consecutive voxels can touch
    static constexpr int NB = (R - 1 + HD - 1) / HD;                   // plane boundaries among them
    static constexpr int PR = R + NB + 2;                               // patch rows: + separators + halo
    static constexpr int NV = GEN ? 16 : PR * PX;
    static constexpr int ARR = NV * 16;                                 // one (chunk, plane, k-half) array
    static constexpr int PL = 2 * ARR, KSB = 3 * PL;                    // plane, chunk
    static constexpr int LDS = 2 * KSB;                                 // two resident chunks
    static constexpr int UNITS = GEN ? 4 : (2 * NV + 255) / 256;        // staging units (voxel, k-half) per thread and chunk (generic: <= 512 patch voxels)
    static constexpr int KS = CT / 16, NSTEP = KS * 9;
    static_assert(KS % 2 == 0 && 18 % P_RB == 0, "chunk pairs keep ring and LDS slots static");
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};
extern __shared__ __attribute__((aligned(16))) unsigned char p2d_dyn_lds[];

template <int W_, int HMIN, int CT>
__global__ __launch_bounds__(256, 2) void p2d_kernel(P2DParams p) {
    typedef P2DCfg<W_, HMIN, CT> G;
    constexpr bool GEN = G::GEN;
    __shared__ __attribute__((aligned(16))) unsigned char patch_static[GEN ? 16 : G::LDS];
    unsigned char* const patch = GEN ? p2d_dyn_lds : patch_static;
    // geometry: compile-time constants, or (generic instance) the launch's values
    const int Wd = GEN ? p.W : W_, PXd = Wd + 2, NVd = GEN ? p.nv : G::NV;
    const int ARRd = NVd * 16, PLd = 2 * ARRd, KSBd = 3 * PLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int mh = wave >> 1, cw = wave & 1;             // wave tile: voxel half mh of the 128 x column half cw
    const int cb = blockIdx.y;                           // 64-channel block of the output channels
    const long o0 = (long)blockIdx.x * P_TM;             // first output voxel (flat)
    const int HW = p.H * Wd;
    // padded row of the first voxel; the patch starts one row above it
    const int n0 = (int)(o0 / HW), y0 = (int)(o0 - (long)n0 * HW) / Wd;
    const int pstart = n0 * (p.H + 1) + y0;              // = P(o0) - 1

    const __amdgpu_buffer_rsrc_t wrs = p_rsrc(p.wimg, (unsigned)((CT / 64) * G::NSTEP * PW_STEP));
    const int w_voff = cw * (3 * PW_BLK) + lane * 16;
    const int w_cb = cb * (G::NSTEP * PW_STEP);
    bf16x8 bfr[P_RB][3];
    auto wload = [&](int g, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bfr[SLOT][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, g < G::NSTEP ? w_voff + pl * PW_BLK : (int)0x80000000u, g < G::NSTEP ? w_cb + g * PW_STEP : 0, 0));
    };
    // (slot = static position of the k-step inside its chunk pair; g itself is a run-time value)
    auto wload_at = [&](int g, int slot) {
        switch (slot % P_RB) {
            case 0: wload(g, std::integral_constant<int, 0>{}); break;
            case 1: wload(g, std::integral_constant<int, 1>{}); break;
            case 2: wload(g, std::integral_constant<int, 2>{}); break;
            case 3: wload(g, std::integral_constant<int, 3>{}); break;
            case 4: wload(g, std::integral_constant<int, 4>{}); break;
            default: wload(g, std::integral_constant<int, 5>{}); break;
        }
    };
#pragma unroll
    for (int g = 0; g < P_RB - 1; ++g) wload_at(g, g);

    // ---- patch staging, one 16-channel chunk at a time: unit q = (patch voxel, k-half); separator rows, halo columns and rows
    //      outside the batch read zeros (offset out of range) ----
    const __amdgpu_buffer_rsrc_t ars = p_rsrc(p.a, p.a_bytes);
    unsigned st_off[G::UNITS];
    int st_lds[G::UNITS];
#pragma unroll
    for (int u = 0; u < G::UNITS; ++u) {
        const int q = tid + 256 * u, vox = q >> 1, hh = q & 1;
        const int pr = vox / PXd, pc = vox - pr * PXd;
        const int P = pstart + pr;                       // padded row: n (H + 1) + 1 + y; 0 (mod H + 1) = separator
        const int n = P / (p.H + 1), y = P - n * (p.H + 1) - 1, x = pc - 1;
        const bool ok = vox < NVd && y >= 0 && n < p.N && (unsigned)x < (unsigned)Wd;
        st_off[u] = ok ? 4u * (unsigned)((((long)n * p.H + y) * Wd + x) * CT + hh * 8) : 0x80000000u;
        st_lds[u] = vox < NVd ? hh * ARRd + vox * 16 : -1;
    }
    u32x4 ld[G::UNITS][2];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int u = 0; u < G::UNITS; ++u) {
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)st_off[u], 64 * c, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)st_off[u], 64 * c + 16, 0);
        }
    };
    auto stage_store = [&](int slot) {
#pragma unroll
        for (int u = 0; u < G::UNITS; ++u) {
            if (st_lds[u] < 0) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            p_cut8(v, o);
            unsigned char* dst = patch + slot * KSBd + st_lds[u];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * PLd) = o[pl];
        }
    };
    stage_load(0);

    // ---- per-lane geometry: row block i = voxels o0 + 64 mh + 32 i .. + 31, MFMA row l32; record of the tap (0, 0) = (y - 1, x - 1) ----
    int vbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long o = o0 + 64 * mh + 32 * i + l32;
        if (o >= p.total) o = p.total - 1;               // (rows behind the batch: computed on a valid voxel, never stored)
        const int n = (int)(o / HW), rem = (int)(o - (long)n * HW), y = rem / Wd, x = rem - y * Wd;
        const int prow = n * (p.H + 1) + 1 + y - pstart; // >= 1
        vbase[i] = ((prow - 1) * PXd + x) * 16 + h * ARRd;
    }

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first

    stage_store(0);
    stage_load(1);
    __syncthreads();

    bf16x8 af[2][2][3];
    auto frags = [&](int slot, int tap, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const int imm = ((tap / 3) * PXd + tap % 3) * 16 + slot * KSBd;        // (an immediate in the compile-time instances)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[SET][i][pl] = *reinterpret_cast<const bf16x8*>(patch + vbase[i] + imm + pl * PLd);
    };
    auto frags_at = [&](int s, int slot, int tap) {      // s: static position inside the chunk pair (its parity picks the set)
        if (s & 1) frags(slot, tap, std::integral_constant<int, 1>{});
        else frags(slot, tap, std::integral_constant<int, 0>{});
    };

    frags_at(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int cp = 0; cp < G::KS / 2; ++cp) {             // chunk pairs: 18 k-steps, every register-ring / LDS slot static inside
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int cc = s / 9, tap = s % 9, c = 2 * cp + cc, g = cp * 18 + s;
            wload_at(g + P_RB - 1, s + P_RB - 1);
            // the next chunk of the patch: cut + stored a few taps into this chunk (its loads have been in flight since the
            // previous chunk), the loads of the chunk after it right behind
            if (tap == P2D_STAGE_TAP && c + 1 < G::KS) {
                stage_store((cc + 1) & 1);
                if (c + 2 < G::KS) stage_load(c + 2);
            }
            if (g + 1 < G::NSTEP) {
                if (tap == 8) __syncthreads();           // next chunk of the patch visible (stored six k-steps ago)
                frags_at(s + 1, ((s + 1) / 9) & 1, (s + 1) % 9);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][0][PA[pr]], bfr[s % P_RB][PB[pr]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][1][PA[pr]], bfr[s % P_RB][PB[pr]], acc[1], 0, 0, 0);
            }
            // issue order inside the k-step: every load behind an MFMA
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

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h of row block i; a row IS a flat voxel ----
    const bool has_mask = p.mask != nullptr;
    const int col = cb * 64 + cw * 32 + l32;
    const __amdgpu_buffer_rsrc_t rrs = p_rsrc(p.res, p.res ? p.a_bytes : 0u), mrs = p_rsrc(p.mask, p.mask ? p.a_bytes : 0u);
    const __amdgpu_buffer_rsrc_t ors = p_rsrc(p.out, p.a_bytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float rr[16], mm[16];
        unsigned eo[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long o = o0 + 64 * mh + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
            eo[r] = o < p.total ? 4u * (unsigned)(o * CT + col) : 0x80000000u;
            rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, (int)eo[r], 0, 0));
            mm[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)eo[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[i][r] + rr[r];
            if (p.relu) v = fmaxf(v, 0.f);
            if (has_mask) v = (mm[r] > 0.f) ? v : 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, (int)eo[r], 0, 0);
        }
    }
}

// ---- weight image: W[tap][ci][co] f32 -> bf16x3 B fragments [64-column block][chunk][tap][column half][plane][lane] x 16 bytes ----
constexpr int PP_MAX = 16;
struct P2DPrepBatch {
    const float* w[PP_MAX];
    unsigned char* img[PP_MAX];
    int dgrad[PP_MAX];
    int ct[PP_MAX];
};
// idx = (cb, chunk, tap, cw, lane): (ct / 64) (ct / 16) 9 x 2 x 64 entries per image
__global__ __launch_bounds__(256) void p2d_prep_kernel(P2DPrepBatch b) {
    const int ct = b.ct[blockIdx.y], ks = ct / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (ct / 64) * ks * 9 * 128) return;
    const int lane = idx & 63, cw = (idx >> 6) & 1, rest = idx >> 7, tap = rest % 9, c = (rest / 9) % ks, cb = rest / (9 * ks);
    const float* w = b.w[blockIdx.y];
    const int nn = cb * 64 + cw * 32 + (lane & 31), k0 = c * 16 + 8 * (lane >> 5);
    float v[8];
    if (b.dgrad[blockIdx.y]) {      // B'[tap][k = co][n = ci] = W[8 - tap][ci = n][co = k]
        const float* src = w + ((long)(8 - tap) * ct + nn) * ct + k0;
        const float4 a = *reinterpret_cast<const float4*>(src), d = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
    } else {                        // B[tap][k = ci][n = co] = W[tap][ci = k][co = n]
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[((long)tap * ct + k0 + e) * ct + nn];
    }
    u32x4 o[3];
    p_cut8(v, o);
    unsigned char* dst = b.img[blockIdx.y] + (size_t)(((cb * ks + c) * 9 + tap) * 2 + cw) * (3 * PW_BLK) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * PW_BLK) = o[pl];
}

// ---- the encoder's first layer: Conv2d(1, Co, 3, padding=1) (simsiam_model_2d.py:634) - nine taps of ONE input channel ----
// Not matrix work: 9 multiply-adds per output element against 4 bytes written (forward) or read (weight gradient) - HBM-bound at
// 85 MB per view and direction; on the implicit GEMM (K = 9 padded to a 32-deep slice) it took 56 / 147 us.  f32 FMA chains.
// forward: a thread owns (voxel, 4 output channels); weights in registers.
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const float* x, const float* w, float* y, int H, int W, int Co, long total) {
    const int cg = Co >> 2;                              // channel groups of 4 per voxel
    const int g = threadIdx.x % cg;
    float wv[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(w + t * Co + 4 * g);
        wv[t][0] = q.x; wv[t][1] = q.y; wv[t][2] = q.z; wv[t][3] = q.w;
    }
    const int vpb = 256 / cg;                            // voxels per block pass
    for (long v = (long)blockIdx.x * vpb + threadIdx.x / cg; v < total; v += (long)gridDim.x * vpb) {
        const int xx = (int)(v % W), yy = (int)((v / W) % H);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int y2 = yy + ty - 1, x2 = xx + tx - 1;
                const float xv = ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) ? x[v + (ty - 1) * W + (tx - 1)] : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = fmaf(xv, wv[ty * 3 + tx][k], a[k]);
            }
        *reinterpret_cast<float4*>(y + v * Co + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
    }
}
// weight gradient: dW[t][co] = sum_v x[v + t] dy[v][co].  A thread owns 4 channels and every (256 / cg)-th voxel of its block's
// share; the block's partial [9][Co] goes to the workspace and stem3_wgrad_final_kernel adds the blocks in block order (fp64).
constexpr int ST3_BLOCKS = 1024;
__global__ __launch_bounds__(256) void stem3_wgrad_kernel(const float* x, const float* dy, float* part, int H, int W, int Co, long total) {
    __shared__ float red[256][37];                       // (+1: bank spread)
    const int cg = Co >> 2, g = threadIdx.x % cg, vl = threadIdx.x / cg, vpb = 256 / cg;
    float a[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[t][k] = 0.f;
    const int per = (int)((total + gridDim.x - 1) / gridDim.x);
    const int v0 = blockIdx.x * per, v1 = (long)v0 + per < total ? v0 + per : (int)total;
    int xx = (v0 + vl) % W, yy = ((v0 + vl) / W) % H;   // (walked, not divided, from here on)
    for (int v = v0 + vl; v < v1; v += vpb, xx += vpb) {
        while (xx >= W) { xx -= W; yy = yy + 1 == H ? 0 : yy + 1; }
        const float4 d = *reinterpret_cast<const float4*>(dy + (long)v * Co + 4 * g);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int y2 = yy + ty - 1, x2 = xx + tx - 1;
                const float xv = ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W) ? x[v + (ty - 1) * W + (tx - 1)] : 0.f;
                a[ty * 3 + tx][0] = fmaf(xv, d.x, a[ty * 3 + tx][0]); a[ty * 3 + tx][1] = fmaf(xv, d.y, a[ty * 3 + tx][1]);
                a[ty * 3 + tx][2] = fmaf(xv, d.z, a[ty * 3 + tx][2]); a[ty * 3 + tx][3] = fmaf(xv, d.w, a[ty * 3 + tx][3]);
            }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x][t * 4 + k] = a[t][k];
    __syncthreads();
    // thread j < 9 Co: element (t, co) = sum over the block's voxel lanes, in lane order
    for (int j = threadIdx.x; j < 9 * Co; j += 256) {
        const int t = j / Co, co = j % Co;
        float sacc = 0.f;
        for (int l = 0; l < vpb; ++l) sacc += red[l * cg + (co >> 2)][t * 4 + (co & 3)];
        part[(long)blockIdx.x * 9 * Co + j] = sacc;
    }
}
// one WAVE per gradient element: lane l adds the partials of blocks l, l + 64, ... (in that order), then the lanes in a fixed tree
__global__ __launch_bounds__(256) void stem3_wgrad_final_kernel(const float* part, float* dw, int n, int nblocks) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    double sacc = 0.0;
    for (int b = lane; b < nblocks; b += 64) sacc += (double)part[(long)b * n + j];
    sacc = wave_sum(sacc);
    if (lane == 0) dw[j] = (float)sacc;
}

// ---- weight gradient of the same layers: dW[tap][ci][co] = sum over output voxels o of X[o + tap][ci] dY[o][co] ----------------------
// The reduction runs over voxels, so both operands are staged voxel-major ([voxel][32 channels] bf16 rows of 64 bytes per channel half
// and bf16 plane) and the MFMA fragments come out of LDS through the transposing read (ds_read_b64_tr_b16: a lane names its own row), as
// in conv_direct3.hip's direct3_wgrad_kernel.  What is new is the geometry: a K-block is 64 consecutive output voxels of the FLAT
// (n, y, x) order; X is staged in the PADDED flat order f = (n (H + 1) + 1 + y) (W + 1) + x + 1 - one zero column per row, one zero row
// between planes - so that a tap is ONE row offset ((ty - 1) (W + 1) + tx - 1) for every lane and every out-of-plane neighbour is a
// staged zero: no select, no mask, any H = W.  A workgroup owns (32 input channels, 64 output channels, one chain of K-blocks): its four
// waves are (output-channel half) x (half of a K-block's four k-steps), nine accumulators each - all taps from one staged window
// (per k-step and wave: 6 + 54 transposing reads, 54 MFMAs); the two k-halves are added through LDS at the end and the chain's
// [9][32][64] tile goes into its split-K slab; p2d_wgrad_reduce_kernel adds the slabs in slab order.
constexpr int PW_KB = 64;                   // output voxels per K-block (4 k-steps)
constexpr int PW_ROW = 64;                  // bytes of a (voxel, 32 channels) bf16 row
typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4p lds_bf16x4p;

// W_ = 0: the generic instance (any H, W whose window has <= 192 rows: run-time geometry, divisions through a float reciprocal)
template <int W_>
struct P2WCfg {
    static constexpr bool GEN = W_ == 0;
    static constexpr int WD = GEN ? 1 : W_;
    static constexpr int H_ = W_;                                       // square planes (compile-time instances)
    static constexpr int PW = W_ + 1;                                   // padded-flat pitch
    static constexpr int R = (PW_KB - 1 + WD - 1) / WD + 1;             // image rows a K-block can touch
    static constexpr int NB = (R - 1 + WD - 1) / WD;                    // plane boundaries among them
    static constexpr int SPAN = PW_KB + (R - 1) + NB * PW;              // padded-flat positions from its first to its last voxel
    static constexpr int XR = GEN ? 8 : SPAN + 2 * (PW + 1);            // + the taps' reach on both sides
    static constexpr int XPL = XR * PW_ROW;                             // one bf16 plane of the X window (32 channels)
    static constexpr int YH = PW_KB * PW_ROW, YPL = 2 * YH;             // dY: channel half, plane
    static constexpr int YB = 3 * XPL;                                  // byte offset of dY
    static constexpr int LDS = 3 * XPL + 3 * YPL;
    static constexpr int UX = GEN ? 3 : (4 * XR + 255) / 256;           // X staging units (row, 8 channels) per thread
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
    static_assert(LDS >= 2 * 12288, "the k-half exchange (two channel halves x three taps) reuses the staging memory");
};
// n / d and n % d for 0 <= n < 2^24 through the reciprocal (the generic instances' plane decode: a handful per K-block, where an
// emulated 32-bit division is ~40 instructions)
__device__ __forceinline__ void p_divmod(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    else if (r >= d) { q += 1; r -= d; }
}

struct P2WParams {
    const float* x;           // (N, H, W, CT)
    const float* dy;          // (N, H, W, CT)
    float* slabs;             // [splits][9][CT][CT]
    int N;
    long total;               // N H W
    int nkb;                  // K-blocks in all
    int splits;
    unsigned bytes;
    int H, W, xr;             // generic instance: plane extents and window rows
};

template <int W_, int CT>
__global__ __launch_bounds__(256, 2) void p2d_wgrad_kernel(P2WParams p) {
    typedef P2WCfg<W_> G;
    constexpr bool GEN = G::GEN;
    __shared__ __attribute__((aligned(16))) unsigned char lds_static[GEN ? 16 : G::LDS];
    unsigned char* const lds = GEN ? p2d_dyn_lds : lds_static;
    const int Wd = GEN ? p.W : W_, H_ = GEN ? p.H : G::H_, PW = Wd + 1, HW = H_ * Wd, XRd = GEN ? p.xr : G::XR;
    const int XPLd = XRd * PW_ROW, YBd = 3 * XPLd;
    const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)Wd, inv_pw = 1.0f / (float)PW, inv_h1 = 1.0f / (float)(H_ + 1);
    // (o / HW, o % HW / W ...: the compile-time instances divide by constants, the generic one through reciprocals)
    auto voxel_of = [&](long o, int& n, int& y, int& x) {
        if (GEN) { int rr; p_divmod((int)o, HW, inv_hw, n, rr); p_divmod(rr, Wd, inv_w, y, x); }
        else { n = (int)(o / HW); const int rr = (int)(o - (long)n * HW); y = rr / Wd; x = rr - y * Wd; }
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int wn = wave & 1, kh = wave >> 1;             // output-channel half; k-steps 2 kh, 2 kh + 1 of every K-block
    constexpr int NCI = CT / 32, NCO = CT / 64;
    const int tile = blockIdx.x % (NCI * NCO), split = blockIdx.x / (NCI * NCO);
    const int ci0 = (tile / NCO) * 32, co0 = (tile % NCO) * 64;
    // this chain's K-blocks: [kb0, kb1)
    const int per = (p.nkb + p.splits - 1) / p.splits;
    const int kb0 = split * per, kb1 = kb0 + per < p.nkb ? kb0 + per : p.nkb;

    const __amdgpu_buffer_rsrc_t xrs = p_rsrc(p.x, p.bytes), yrs = p_rsrc(p.dy, p.bytes);
    // fragment addressing (transposing read): this lane names row q4 of its 16-lane group's 4-row block, columns 16 g16 + 4 (i16 & 3)
    const int q4 = i16 >> 2;
    const int coloff = (16 * g16 + 4 * (i16 & 3)) * 2;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    // staging registers: the NEXT K-block's operands are fetched while this one's products run
    u32x4 ldy[2][2], ldx[G::UX][2];
    auto window_start = [&](int kb) {                    // padded-flat position of window row 0 of K-block kb (may be < 0)
        const long o0 = (long)kb * PW_KB;
        int n0, y0, x0;
        voxel_of(o0, n0, y0, x0);
        return (n0 * (H_ + 1) + 1 + y0) * PW + x0 + 1 - (PW + 1);
    };
    auto fetch = [&](int kb) {
        const long o0 = (long)kb * PW_KB;
        const int fstart = window_start(kb);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u, v = q >> 3, cg = q & 7;
            const long o = o0 + v;
            const unsigned off = (kb < kb1 && o < p.total) ? 4u * (unsigned)(o * CT + co0 + 8 * cg) : 0x80000000u;
            ldy[u][0] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)off, 0, 0);
            ldy[u][1] = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)off, 16, 0);
        }
#pragma unroll
        for (int u = 0; u < G::UX; ++u) {
            const int q = tid + 256 * u, r = q >> 2, cg = q & 3;
            const int f = fstart + r;
            int P, xc, n, yy;
            if (GEN) { p_divmod(f < 0 ? 0 : f, PW, inv_pw, P, xc); p_divmod(P, H_ + 1, inv_h1, n, yy); yy -= 1; }
            else { P = f / PW; xc = f - P * PW; n = P / (H_ + 1); yy = P - n * (H_ + 1) - 1; }
            const bool ok = kb < kb1 && r < XRd && f >= 0 && xc >= 1 && yy >= 0 && n < p.N;
            const unsigned off = ok ? 4u * (unsigned)((((long)n * H_ + yy) * Wd + (xc - 1)) * CT + ci0 + 8 * cg) : 0x80000000u;
            ldx[u][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0);
            ldx[u][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 16, 0);
        }
    };
    auto store = [&]() {                                 // cut + store what `fetch` brought: dY units, then the X window's units
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u, v = q >> 3, cg = q & 7;
            float vv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { vv[e] = __uint_as_float(ldy[u][0][e]); vv[4 + e] = __uint_as_float(ldy[u][1][e]); }
            u32x4 o3[3];
            p_cut8(vv, o3);
            unsigned char* dst = lds + YBd + (cg >> 2) * G::YH + v * PW_ROW + (cg & 3) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * G::YPL) = o3[pl];
        }
#pragma unroll
        for (int u = 0; u < G::UX; ++u) {
            const int q = tid + 256 * u, r = q >> 2, cg = q & 3;
            if (r < XRd) {
                float vv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { vv[e] = __uint_as_float(ldx[u][0][e]); vv[4 + e] = __uint_as_float(ldx[u][1][e]); }
                u32x4 o3[3];
                p_cut8(vv, o3);
                unsigned char* dst = lds + r * PW_ROW + cg * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * XPLd) = o3[pl];
            }
        }
    };
    fetch(kb0);
    for (int kb = kb0; kb < kb1; ++kb) {
        const long o0 = (long)kb * PW_KB;
        const int fstart = window_start(kb);
        __syncthreads();                                 // every wave is done with the previous window
        store();
        // ---- this lane's rows of the window for its two k-steps: voxel o0 + 16 (2 kh + lk) + 8 h + q4 (+ 4) ----
        int xrow[2][2];
#pragma unroll
        for (int lk = 0; lk < 2; ++lk)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                long o = o0 + 16 * (2 * kh + lk) + 8 * h + q4 + 4 * hi;
                if (o >= p.total) o = p.total - 1;       // (its dY row is zero: any staged row will do)
                int n, y, x;
                voxel_of(o, n, y, x);
                // (biased by the taps' reach: the offset of tap (ty, tx) is then (ty PW + tx) rows >= 0 - a ds_read immediate)
                xrow[lk][hi] = ((n * (H_ + 1) + 1 + y) * PW + x + 1 - fstart - (PW + 1)) * PW_ROW + coloff;
            }
        fetch(kb + 1);                                   // (behind the last K-block: nothing - offsets out of range)
        __syncthreads();
        // ---- 2 k-steps x (dY fragment + 9 taps x X fragment) ----
#pragma unroll
        for (int lk = 0; lk < 2; ++lk) {
            bf16x8 bfg[3];
            const unsigned char* yb = lds + YBd + wn * G::YH + (16 * (2 * kh + lk) + 8 * h + q4) * PW_ROW + coloff;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const bf16x4p lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4p*)(yb + pl * G::YPL));
                const bf16x4p hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4p*)(yb + pl * G::YPL + 4 * PW_ROW));
                bfg[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            bf16x8 af[2][3];
            auto read_a = [&](int t, int set) {
                const int sh = ((t / 3) * PW + t % 3) * PW_ROW;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x4p lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4p*)(lds + pl * XPLd + xrow[lk][0] + sh));
                    const bf16x4p hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4p*)(lds + pl * XPLd + xrow[lk][1] + sh));
                    af[set][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            read_a(0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) read_a(t + 1, (t + 1) & 1);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][PA[pr]], bfg[PB[pr]], acc[t], 0, 0, 0);
                // issue order inside a tap: its first MFMA, then the next tap's six fragment reads (five MFMAs = 160 cycles to land)
                // (the 36-wide window has three staging units per thread in flight: pinned, its allocation spills 16 registers and the
                // kernel is no faster - left to the scheduler there)
                if constexpr (G::UX <= 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                }
            }
            if constexpr (G::UX <= 2) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the two k-halves added through LDS (three taps at a time), then the slab: C/D layout col = lane & 31 (co), row = ci ----
    float* red = reinterpret_cast<float*>(lds);          // [co half 2][tap 3][register 16][lane 64]
    float* out = p.slabs + (long)split * (9l * CT * CT);
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        __syncthreads();
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wn * 3 + t) * 16 + r) * 64 + lane] = acc[3 * g3 + t][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    out[((long)(3 * g3 + t) * CT + ci) * CT + co0 + 32 * wn + l32] = acc[3 * g3 + t][r] + red[((wn * 3 + t) * 16 + r) * 64 + lane];
                }
        }
    }
}

// dW = sum of the slabs, in slab order: a block owns 16 float4 of the gradient x 16 slab lanes (lane l adds slabs l, l + 16, ...), the
// sixteen partial sums of an element are then added in lane order
__global__ __launch_bounds__(256) void p2d_wgrad_reduce_kernel(const float* slabs, int n_slabs, long n4, float* out) {
    __shared__ float4 part[16][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i4 = (long)blockIdx.x * 16 + e;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4)
        for (int sidx = sl; sidx < n_slabs; sidx += 16) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + ((long)sidx * n4 + i4) * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    part[sl][e] = a;
    __syncthreads();
    if (sl == 0 && i4 < n4) {
        float4 t = part[0][e];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = part[k][e]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + i4 * 4) = t;
    }
}

// window rows of the generic instance for (H, W) planes: 0 = does not fit (three staging units per thread, reciprocal divisions < 2^24)
int p2dg_wgrad_rows(int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1) return 0;
    const int PW = W + 1, R = (PW_KB - 1 + W - 1) / W + 1, NB = (R - 1 + H - 1) / H;
    const int xr = PW_KB + (R - 1) + NB * PW + 2 * (PW + 1);
    if (xr > 192 || (long)(N + 1) * (H + 1) * PW >= (1l << 24) || (long)N * H * W >= (1l << 24)) return 0;
    return xr;
}
template <int W_, int CT>
int p2d_wgrad_launch(const float* x, const float* dy, float* dw, float* slabs, int N, int splits, hipStream_t s, int H = W_, int W = W_) {
    P2WParams p = {};
    p.x = x; p.dy = dy; p.slabs = slabs; p.N = N; p.total = (long)N * H * W;
    p.nkb = (int)((p.total + PW_KB - 1) / PW_KB); p.splits = splits; p.bytes = (unsigned)(4l * p.total * CT);
    p.H = H; p.W = W; p.xr = W_ ? 0 : p2dg_wgrad_rows(N, H, W);
    const size_t dyn = W_ ? 0 : (size_t)3 * p.xr * PW_ROW + 3 * 2 * PW_KB * PW_ROW;
    hipLaunchKernelGGL((p2d_wgrad_kernel<W_, CT>), dim3((unsigned)((CT / 32) * (CT / 64) * splits)), dim3(256), dyn, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    const long n4 = 9l * CT * CT / 4;
    hipLaunchKernelGGL(p2d_wgrad_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, (const float*)slabs, splits, n4, dw);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

template <int W_, int HMIN, int CT>
int p2d_launch(const P2DParams& p, hipStream_t s) {
    const long tiles = (p.total + P_TM - 1) / P_TM;
    if (tiles > 0x7fffffffl) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL((p2d_kernel<W_, HMIN, CT>), dim3((unsigned)tiles, CT / 64), dim3(256), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
// the generic instance's patch: rows 128 consecutive voxels can touch + separators + halo, W + 2 wide; 0 = does not fit
int p2dg_patch_voxels(int H, int W) {
    if (H < 1 || W < 1) return 0;
    const int R = (P_TM - 1 + W - 1) / W + 1, NB = (R - 1 + H - 1) / H, nv = (R + NB + 2) * (W + 2);
    return (nv <= 426 && 2 * nv <= 1024) ? nv : 0;       // 2 chunks x 6 arrays x 16 bytes per voxel <= 80 KB; <= 4 staging units per thread
}
template <int CT>
int p2dg_launch(P2DParams p, int W, hipStream_t s) {
    const long tiles = (p.total + P_TM - 1) / P_TM;
    if (tiles > 0x7fffffffl) return MI_E_UNSUPPORTED;
    p.W = W; p.nv = p2dg_patch_voxels(p.H, W);
    if (!p.nv) return MI_E_UNSUPPORTED;
    hipLaunchKernelGGL((p2d_kernel<0, 0, CT>), dim3((unsigned)tiles, CT / 64), dim3(256), (size_t)(2 * 6 * 16) * p.nv, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

}  // namespace

// 1: a shape of p2d_kernel - 3 x 3, stride 1, padding 1, C -> C channels on (N, H, W) planes with (W, C) one of (36, 64), (18, 128),
// (9, 256) (the SimSiam 2-D encoder at --bbox 36) and H >= W; MI_NO_P2D=1: off (the implicit GEMM takes the layer)
extern "C" int mi_conv2d_p2d_usable(int N, int H, int W, int C) {
    if (getenv("MI_NO_P2D") || N < 1 || H < 1 || W < 1) return 0;
    if (C != 64 && C != 128 && C != 256) return 0;
    if (4l * N * H * W * C >= 0x7fff0000l) return 0;
    if (H >= W && ((W == 36 && C == 64) || (W == 18 && C == 128) || (W == 9 && C == 256))) return 1;      // compile-time instances
    // 2: the generic instance (any plane whose patch fits; MI_NO_P2D_GENERIC=1: off)
    return (!getenv("MI_NO_P2D_GENERIC") && p2dg_patch_voxels(H, W) > 0) ? 2 : 0;
}
extern "C" size_t mi_conv2d_p2d_wimg_bytes(int C) { return (size_t)(C / 64) * (C / 16) * 9 * PW_STEP; }

// One launch cuts n weight images (n <= 16 per launch, more in several): w[i] = (3, 3, C_i, C_i) kernel-layout weights, img[i] =
// mi_conv2d_p2d_wimg_bytes(C_i) bytes, dgrad[i] != 0: the transposed, tap-flipped image of the data gradient.
extern "C" int mi_conv2d_p2d_prep(const void* const* w, void* const* img, const int* dgrad, const int* channels, int n, mi_stream_t stream) {
    if (!w || !img || !dgrad || !channels || n < 0) return MI_E_ARG;
    for (int i0 = 0; i0 < n; i0 += PP_MAX) {
        P2DPrepBatch b = {};
        const int m = n - i0 < PP_MAX ? n - i0 : PP_MAX;
        int cmax = 0;
        for (int i = 0; i < m; ++i) {
            const int c = channels[i0 + i];
            if (!w[i0 + i] || !img[i0 + i] || (c != 64 && c != 128 && c != 256)) return MI_E_ARG;
            b.w[i] = (const float*)w[i0 + i]; b.img[i] = (unsigned char*)img[i0 + i]; b.dgrad[i] = dgrad[i0 + i]; b.ct[i] = c;
            cmax = c > cmax ? c : cmax;
        }
        const int entries = (cmax / 64) * (cmax / 16) * 9 * 128;
        hipLaunchKernelGGL(p2d_prep_kernel, dim3((entries + 255) / 256, m), dim3(256), 0, (hipStream_t)stream, b);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}

// out = act(conv3x3(a; image) + res) * (mask > 0): a, out, res, mask (N, H, W, C) channels-last f32; res / mask may be null
extern "C" int mi_conv2d_p2d_f32(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                                 int H, int W, int C, mi_stream_t stream) {
    if (!a || !wimg || !out) return MI_E_ARG;
    const int kind = mi_conv2d_p2d_usable(N, H, W, C);
    if (!kind) return MI_E_UNSUPPORTED;
    P2DParams p = {};
    p.a = a; p.wimg = (const unsigned char*)wimg; p.out = out; p.res = res; p.mask = mask; p.relu = relu;
    p.N = N; p.H = H; p.total = (long)N * H * W; p.a_bytes = (unsigned)(4l * N * H * W * C);
    hipStream_t s = (hipStream_t)stream;
    if (kind == 2) return C == 64 ? p2dg_launch<64>(p, W, s) : C == 128 ? p2dg_launch<128>(p, W, s) : p2dg_launch<256>(p, W, s);
    if (W == 36) return p2d_launch<36, 36, 64>(p, s);
    if (W == 18) return p2d_launch<18, 18, 128>(p, s);
    return p2d_launch<9, 9, 256>(p, s);
}

// The 2-D encoder's first layer, Conv2d(1, Co, 3, padding=1) (models/networks/simsiam_model_2d.py:634; Co a multiple of 4, <= 64):
// x (N, H, W) one channel, w (3, 3, 1, Co) kernel layout, y / dy (N, H, W, Co).  ws: mi_conv2d_stem3_workspace_bytes(Co) bytes.
extern "C" size_t mi_conv2d_stem3_workspace_bytes(int Co) { return sizeof(float) * (size_t)ST3_BLOCKS * 9 * (size_t)(Co > 0 ? Co : 0); }
extern "C" int mi_conv2d_stem3_fwd_f32(const float* x, const float* w, float* y, int N, int H, int W, int Co, mi_stream_t stream) {
    if (!x || !w || !y || N < 1 || H < 1 || W < 1) return MI_E_ARG;
    if (Co < 4 || Co > 64 || (Co & 3) || 256 % (Co >> 2)) return MI_E_UNSUPPORTED;
    const long total = (long)N * H * W;
    const long blocks = (total + 256 / (Co >> 2) - 1) / (256 / (Co >> 2));
    hipLaunchKernelGGL(stem3_fwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, w, y, H, W, Co,
                       total);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_conv2d_stem3_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int Co, void* ws, size_t ws_bytes,
                                         mi_stream_t stream) {
    if (!x || !dy || !dw || !ws || N < 1 || H < 1 || W < 1) return MI_E_ARG;
    if (Co < 4 || Co > 64 || (Co & 3) || 256 % (Co >> 2)) return MI_E_UNSUPPORTED;
    if (ws_bytes < mi_conv2d_stem3_workspace_bytes(Co)) return MI_E_ARG;
    const long total = (long)N * H * W;
    if (total >= 0x7fffffffl) return MI_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem3_wgrad_kernel, dim3(ST3_BLOCKS), dim3(256), 0, s, x, dy, (float*)ws, H, W, Co, total);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(stem3_wgrad_final_kernel, dim3((9 * Co + 3) / 4), dim3(256), 0, s, (const float*)ws, dw, 9 * Co, ST3_BLOCKS);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

// Weight gradient of the same layers (mi_conv2d_p2d_usable shapes with H == W): dw (3, 3, C, C) kernel layout, written (not accumulated).
// ws: mi_conv2d_p2d_wgrad_workspace_bytes(N, H, W, C) bytes of split-K slabs (512 chains of K-blocks over the chip).
static int p2d_wgrad_splits(int C) { return 512 / ((C / 32) * (C / 64)); }
extern "C" size_t mi_conv2d_p2d_wgrad_workspace_bytes(int N, int H, int W, int C) {
    const int kind = mi_conv2d_p2d_usable(N, H, W, C);
    if (!kind) return 0;
    const bool fixed = kind == 1 && H == W;              // (a taller plane of a compile-time width takes the generic instance)
    if (!fixed && (getenv("MI_NO_P2D_GENERIC") || !p2dg_wgrad_rows(N, H, W))) return 0;
    return sizeof(float) * (size_t)p2d_wgrad_splits(C) * 9 * (size_t)C * C;
}
extern "C" int mi_conv2d_p2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int C, void* ws, size_t ws_bytes,
                                       mi_stream_t stream) {
    if (!x || !dy || !dw || !ws) return MI_E_ARG;
    const size_t need = getenv("MI_NO_P2D_WGRAD") ? 0 : mi_conv2d_p2d_wgrad_workspace_bytes(N, H, W, C);
    if (!need) return MI_E_UNSUPPORTED;
    if (ws_bytes < need) return MI_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int splits = p2d_wgrad_splits(C);
    if (mi_conv2d_p2d_usable(N, H, W, C) == 1 && H == W) {
        if (W == 36) return p2d_wgrad_launch<36, 64>(x, dy, dw, (float*)ws, N, splits, s);
        if (W == 18) return p2d_wgrad_launch<18, 128>(x, dy, dw, (float*)ws, N, splits, s);
        return p2d_wgrad_launch<9, 256>(x, dy, dw, (float*)ws, N, splits, s);
    }
    if (C == 64) return p2d_wgrad_launch<0, 64>(x, dy, dw, (float*)ws, N, splits, s, H, W);
    if (C == 128) return p2d_wgrad_launch<0, 128>(x, dy, dw, (float*)ws, N, splits, s, H, W);
    return p2d_wgrad_launch<0, 256>(x, dy, dw, (float*)ws, N, splits, s, H, W);
}
