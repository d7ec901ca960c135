// Data gradient of the two STRIDE-2 3x3x3 convolutions of the MoCo-3D encoder (layer2.0.conv1: 64 -> 128 channels, 8^3 -> 4^3;
// layer3.0.conv1: 128 -> 256, 4^3 -> 2^3; cet_pick/models/networks/moco_encoder_3d.py:55-84,257-272) with the gradient of the
// block's 1x1 stride-2 shortcut folded in, bf16x3 arithmetic (exact three-way bf16 cut, six products, f32 accumulate).
//
//   dX[i] = sum over taps t with (i + 1 - t) even of dH[(i + 1 - t) / 2] . W[t]^T      (+ dOut[i / 2] . Wds^T  where i is all-even)
//
// Per axis an even input coordinate i = 2a meets ONE tap (t = 1, output voxel a), an odd one i = 2a + 1 meets TWO (t = 2 -> a,
// t = 0 -> a + 1): the 8 parity classes of the input grid have 1, 2, 4 or 8 taps (27 in all).  The implicit GEMM runs the
// classes as ragged tiles of one launch and ends with its heaviest tile - 32 reduction slices at ~1 us for the all-odd class of
// layer2.0.conv1: 35 us + 11 us for the shortcut's own launch (profiles/r03_step_timeline.txt).  Here:
//   * a workgroup owns ONE parity class of one sample (layer2: 64 rows) or of four samples (layer3: 32 rows) x all output
//     channels; its patch is the dH block itself - G^3 voxels per sample x all reduction channels, cut ONCE into LDS as 16-byte
//     records per (16-channel k-step, bf16 plane, k-half), one record per voxel: the A fragment of a tap is one ds_read_b128 at
//     (voxel record + a tap offset of 0 / 1 per axis), and a voxel whose neighbour a + 1 falls outside the grid reads a zero
//     record in the SAME 16-byte slot (conv_direct3.hip);
//   * weights never pass through LDS: a prep kernel cuts W[t]^T into an image in MFMA B-fragment order, class by class, and
//     every wave streams its 1-KB fragments from L2, four k-steps ahead;
//   * the all-even class then re-stages the patch from dOut (the gradient of the block's output) and adds the shortcut's
//     single "tap" from the shortcut's image: the two data-gradient launches (+ their split-K reduces) of a block become one;
//   * epilogue: (acc + res) * (mask > 0) scattered to the class's voxels of dX; the 8 classes tile dX exactly: no slabs, no
//     reduce launch.  Heaviest classes are launched first.
// 60 - 72 KB of LDS: two workgroups (of eight waves) per CU.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int S2_WBLK = 1024;                // one B fragment: 64 lanes x 16 bytes

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_s2(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void cut8s(const float (&v)[8], u32x4 (&o)[3]) {
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

// class c = (pz << 2) | (py << 1) | px, parities of the input voxel; local tap j enumerates the odd axes (z, y, x order):
// bit = 1 -> delta = 1 (tap 0), bit = 0 -> delta = 0 (tap 2); an even axis has delta = 0 (tap 1)
__host__ __device__ inline void s2_tap_of(int c, int j, int* tap, int* dz, int* dy, int* dx) {
    int d[3], t[3], bit = 0;
    // bits of j are consumed from the LAST odd axis upwards: x is the fastest
    for (int ax = 2; ax >= 0; --ax) {
        const int p = (c >> (2 - ax)) & 1;
        if (p) { d[ax] = (j >> bit) & 1; t[ax] = d[ax] ? 0 : 2; ++bit; }
        else { d[ax] = 0; t[ax] = 1; }
    }
    *tap = (t[0] * 3 + t[1]) * 3 + t[2];
    *dz = d[0]; *dy = d[1]; *dx = d[2];
}

struct S2DgradParams {
    const float* dh;          // (N, G, G, G, CR): gradient of the strided convolution's output
    const float* dout;        // (N, G, G, G, CR): gradient of the block's output (the shortcut's input gradient source); may be null
    const unsigned char* wimg;     // [27 taps, class order][KS][CB][plane][lane] x 16 bytes
    const unsigned char* dsimg;    // [KS][CB][plane][lane] x 16 bytes (with dout)
    float* dx;                // (N, 2G, 2G, 2G, CN)
    const float* res;         // may be null
    const float* mask;        // may be null: dx *= (mask > 0)
    int N, n_groups;
    unsigned a_bytes, o_bytes, wimg_bytes, dsimg_bytes;
    unsigned char order[8];   // classes, heaviest first
    unsigned char tap_off[8]; // first image tap of a class
};

// Eight waves: RBK row blocks x CBW column blocks x KW parts of a tap's k-steps (k-step ks belongs to part ks % KW; the parts
// meet in LDS at the end).  A workgroup's time is its chain of MFMAs - the all-odd class has 8 taps - so the chain is cut KW
// ways, and layer3 (32 rows per workgroup) spreads its 128 output channels over two workgroups to reach 256 of them.
template <int G, int CR, int CN, int CBW, int KW, int RING, int OCC>
__global__ __launch_bounds__(512, OCC) void s2_dgrad_kernel(S2DgradParams p) {
    constexpr int VO = G * G * G;                       // voxels of a dH sample
    constexpr int RBK = VO >= 32 ? VO / 32 : 1;         // row blocks of a workgroup
    constexpr int NV = 32 * RBK;                        // patch records: one sample (layer2) / 32 / VO samples (layer3)
    constexpr int SPW = NV / VO;                        // samples per workgroup
    constexpr int CB = CN / 32;                         // column blocks of the image
    static_assert(RBK * CBW * KW == 8, "eight waves");
    constexpr int KS = CR / 16;                         // k-steps per tap
    constexpr int SPT = KS / KW;                        // ... of one wave
    static_assert(SPT == 4 && (RING == 4 || RING == 8), "one or two taps per ring round");
    constexpr int TPR = RING / SPT;                     // taps per ring round (RING weight fragments in flight per wave)
    constexpr int ARR = (NV + 18) * 16;                 // one (k-step, plane, k-half) array: records + 16 zero records (+ 2: the k-half arrays 32 bytes apart mod 128, no bank shared by the two lane halves)
    constexpr int LDS_BYTES = KS * 6 * ARR;
    constexpr int WSTEP = CB * 3 * S2_WBLK;             // bytes per k-step of the image
    static_assert(LDS_BYTES >= 4 * 16 * 64 * 4 * (KW - 1), "the reduction buffer fits the patch");
    __shared__ __attribute__((aligned(16))) unsigned char patch[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int kw = wave / (RBK * CBW), tile = wave % (RBK * CBW);
    const int rb = tile / CBW, cbk = blockIdx.y * CBW + tile % CBW;
    const int cls = p.order[blockIdx.x / p.n_groups], grp = blockIdx.x % p.n_groups;      // class-major: heaviest classes first
    const int n0 = grp * SPW;
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    const int ntaps = 1 << (pz + py + px);

    const __amdgpu_buffer_rsrc_t wrs = rsrc_s2(p.wimg, p.wimg_bytes), drs = rsrc_s2(p.dsimg, p.dsimg ? p.dsimg_bytes : 0u);
    const int w_voff = cbk * (3 * S2_WBLK) + lane * 16;
    bf16x8 bfr[RING][3];

    // ---- patch staging: unit q = (voxel record, 8 channels): NV * CR / 8 units, UPT per thread (2; 4 for the 64^3 crops' layer3.0) ----
    constexpr int UPT = NV * CR / 8 / 512;
    static_assert(UPT * 512 * 8 == NV * CR, "staging divides");
    auto stage = [&](const float* src) {
        const __amdgpu_buffer_rsrc_t ars = rsrc_s2(src, p.a_bytes);
        u32x4 ld[UPT][2];
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int q = tid + 512 * u, vox = q / (CR / 8), cg = q % (CR / 8);
            const bool ok = n0 + vox / VO < p.N;
            const unsigned off = ok ? 4u * (unsigned)(((long)n0 * VO + vox) * CR + cg * 8) : 0x80000000u;
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)off, 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(off + 16u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int q = tid + 512 * u, vox = q / (CR / 8), cg = q % (CR / 8);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            cut8s(v, o);
            unsigned char* dst = patch + ((cg >> 1) * 6 + (cg & 1)) * ARR + vox * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 2 * ARR) = o[pl];
        }
    };
    for (int i = tid; i < KS * 6 * 16; i += 512)            // zero records of every array
        *reinterpret_cast<u32x4*>(patch + (i >> 4) * ARR + (NV + (i & 15)) * 16) = u32x4{0u, 0u, 0u, 0u};
    stage(p.dh);

    // ---- per-lane geometry: record v of the patch = (sample, a, b, c) ----
    const int v = rb * 32 + l32, vv = v % VO;
    const int a = vv / (G * G), b = (vv / G) % G, c = vv % G;
    const int vaddr = v * 16 + h * ARR, zaddr = (NV + (v & 15)) * 16 + h * ARR;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    auto wload = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned so, auto SLOTc) {
        constexpr int SLOT = decltype(SLOTc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            // the offset (and its out-of-range sentinel) rides in the CHECKED voffset: soffset is not part of the range check
            bfr[SLOT][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(so + (unsigned)(w_voff + pl * S2_WBLK)), 0, 0));
    };
    auto wload_dyn = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned so, int slot) {
        switch (slot) {
            case 0: wload(rs, so, std::integral_constant<int, 0>{}); break;
            case 1: wload(rs, so, std::integral_constant<int, 1>{}); break;
            case 2: wload(rs, so, std::integral_constant<int, 2>{}); break;
            case 3: wload(rs, so, std::integral_constant<int, 3 % RING>{}); break;
            case 4: wload(rs, so, std::integral_constant<int, 4 % RING>{}); break;
            case 5: wload(rs, so, std::integral_constant<int, 5 % RING>{}); break;
            case 6: wload(rs, so, std::integral_constant<int, 6 % RING>{}); break;
            default: wload(rs, so, std::integral_constant<int, 7 % RING>{}); break;
        }
    };
    const unsigned tap_bytes = (unsigned)KS * WSTEP;
    // this wave's step s of a tap sequence (SPT steps per tap): image byte offset, or out of range behind the last tap
    auto woff = [&](unsigned base, int s, int n_t) {
        const int j = s / SPT, i = s % SPT;
        return j < n_t ? base + (unsigned)j * tap_bytes + (unsigned)(i * KW + kw) * WSTEP : 0x80000000u;
    };
    auto sel_of = [&](int j, int n_t) {
        if (j >= n_t) return zaddr;
        int tap, dz, dy, dxx;
        s2_tap_of(cls, j, &tap, &dz, &dy, &dxx);
        const bool ok = (a + dz < G) && (b + dy < G) && (c + dxx < G);
        return ok ? vaddr + ((dz * G + dy) * G + dxx) * 16 : zaddr;
    };
    // a sequence of n_t taps: rounds of RING = 8 steps (two taps), fragments RING - 1 steps ahead
    auto run = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned base, int n_t, bool plain) {
#pragma unroll
        for (int g = 0; g < RING - 1; ++g) wload_dyn(rs, woff(base, g, n_t), g);
        for (int s0 = 0; s0 < n_t * SPT; s0 += RING) {
            const int j0 = s0 / SPT;
            const int sel0 = plain ? vaddr : sel_of(j0, n_t), sel1 = (plain || TPR == 1) ? zaddr : sel_of(j0 + 1, n_t);
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                wload_dyn(rs, woff(base, s0 + u + RING - 1, n_t), (u + RING - 1) % RING);
                const int ks = (u % SPT) * KW + kw;
                const int sel = u < SPT ? sel0 : sel1;
                bf16x8 af[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    af[pl] = *reinterpret_cast<const bf16x8*>(patch + sel + (ks * 6 + pl * 2) * ARR);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bfr[u][PB[pr]], acc, 0, 0, 0);
            }
        }
    };

    __syncthreads();
    run(wrs, (unsigned)p.tap_off[cls] * tap_bytes, ntaps, false);
    if (cls == 0 && p.dout) {
        // the 1x1 stride-2 shortcut: dX[2a, 2b, 2c] += dOut[a, b, c] . Wds^T - the patch is re-staged from dOut
        __syncthreads();
        stage(p.dout);
        __syncthreads();
        run(drs, 0u, 1, true);
    }

    // ---- the KW parts meet in LDS (the patch is no longer read): part 0 adds parts 1.. in order, then the epilogue ----
    __syncthreads();
    float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(patch);
    if (kw > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[kw - 1][tile][r][lane] = acc[r];
    }
    __syncthreads();
    if (kw == 0) {
        constexpr int G2 = 2 * G;
        const int col = cbk * 32 + l32;
        const bool has_mask = p.mask != nullptr, has_res = p.res != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = acc[r];
#pragma unroll
            for (int k = 1; k < KW; ++k) t += red[k - 1][tile][r][lane];
            const int rv = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int n = n0 + rv / VO, w = rv % VO;
            if (n >= p.N) continue;
            const int z = 2 * (w / (G * G)) + pz, y = 2 * ((w / G) % G) + py, x = 2 * (w % G) + px;
            const long o = ((((long)n * G2 + z) * G2 + y) * G2 + x) * CN + col;
            if (has_res) t += p.res[o];
            if (has_mask) t = p.mask[o] > 0.f ? t : 0.f;
            p.dx[o] = t;
        }
    }
}

// (weight images: s2_prep_batch_kernel below cuts W[t]^T, class by class, into B-fragment order)
bool s2_shape(int Gi, int Ci, int Co, int* G) {
    // input grid Gi (= 2 G), Ci input channels of the convolution (the gradient's output channels), Co its output channels
    if (Gi == 8 && Ci == 64 && Co == 128) { *G = 4; return true; }
    if (Gi == 4 && Ci == 128 && Co == 256) { *G = 2; return true; }
    if (Gi == 8 && Ci == 128 && Co == 256) { *G = 4; return true; }        // layer3.0 of the 64^3 crops (round 6)
    return false;
}
size_t s2_img_bytes(int Ci, int Co) { return (size_t)27 * (Co / 16) * (Ci / 32) * 3 * S2_WBLK; }
size_t s2_dsimg_bytes(int Ci, int Co) { return (size_t)(Co / 16) * (Ci / 32) * 3 * S2_WBLK; }


// ===========================================================================================================================
// FORWARD of a stride-2 block front: hmid = relu(conv3x3x3 stride 2 (x; W1)) and the 1x1 stride-2 shortcut r = x[2o] . Wds in ONE
// launch (the implicit GEMM took a launch + a split-K reduce for the first and a launch for the second: 32 + 30 us per encoder).
//
//   hmid[o] = sum over taps t of x[2 o + t - 1] . W[t]:   per axis t = 1 reads the EVEN input 2a, t = 0 / 2 the ODD inputs
//   2(a - 1) + 1 / 2a + 1 - so with the input split into its 8 parity sub-grids X_p[a] = x[2a + p] (each G^3 voxels) a tap reads
//   sub-grid p(t) at voxel offset -1 or 0 per axis, and an offset that leaves the grid (a - 1 < 0) is the padding: a zero record.
//
// A workgroup owns the output rows of a sample group (layer2.0: one sample = 64 rows; layer3.0: four samples = 32 rows) x CBW
// blocks of 32 output channels; the input patch passes through LDS one 16-channel k-step at a time (two slots), cut ONCE into
// 16-byte bf16x3 records per voxel and (plane, k-half), records ordered [sample][parity class][a][b][c]: the A fragment of a
// tap is one ds_read_b128 per plane at a per-lane record address that does not depend on the k-step.  The 27 taps + the
// shortcut are 28 equal entries, split evenly over the KW wave groups (14 or 7 each: one B-fragment ring round of 7); the
// shortcut entry reads the centre tap's A fragment and accumulates into a second tile that leaves as `r`.  Weights come from a
// pre-cut image in B-fragment order, streamed from L2 through a 7-deep register ring.  The KW partial tiles meet in LDS.
struct S2FwdParams {
    const float* x;           // (N, 2G, 2G, 2G, CI)
    const unsigned char* wimg;      // [28 entries][KS][CB][plane][lane] x 16 bytes; entry 27 = the shortcut
    float* hmid;              // (N, G, G, G, CO) = relu(conv)
    float* r;                 // (N, G, G, G, CO) = shortcut
    int N;
    unsigned x_bytes, wimg_bytes;
};

template <int G, int CI, int CO, int RBK, int CBW, int KW, int OCC>
__global__ __launch_bounds__(512, OCC) void s2_fwd_kernel(S2FwdParams p) {
    constexpr int VO = G * G * G;                       // output voxels of a sample
    constexpr int VI = 8 * VO;                          // input voxels of a sample
    constexpr int SPW = RBK * 32 / VO;                  // samples per workgroup
    static_assert(SPW >= 1 && SPW * VO == RBK * 32 && RBK * CBW * KW == 8, "rows of whole samples, eight waves");
    constexpr int NV = SPW * VI;                        // patch records per array
    constexpr int ARR = (NV + 18) * 16;                 // + 16 zero records (+ 2: the two k-half arrays 32 bytes apart mod 128)
    constexpr int SLOT = 6 * ARR;                       // one k-step of the patch: (plane, k-half) arrays
    constexpr int KS = CI / 16, CB = CO / 32;
    // 28 entries (27 taps + the shortcut) in equal ranges over the KW wave groups; KW = 8 pads to 32 (four idle steps)
    constexpr int NREAL = 28, NE = (NREAL + KW - 1) / KW * KW, NTW = NE / KW, RING = NTW < 7 ? NTW : 7;
    static_assert(NTW % RING == 0 && RING >= 2, "whole ring rounds");
    constexpr int ROUNDS = NTW / RING;
    constexpr int WSTEP = CB * 3 * S2_WBLK;             // image bytes per (entry, k-step)
    constexpr int UNITS = NV * 2;                       // staging units: (voxel, 8 channels)
    constexpr int UPT = UNITS / 512;                    // ... per thread
    constexpr int TILES = RBK * CBW;
    static_assert(UNITS % 512 == 0 && 2 * SLOT >= TILES * 16 * 64 * 4 * (KW - 1), "staging divides, reduction buffer fits");
    __shared__ __attribute__((aligned(16))) unsigned char patch[2 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int kw = wave / (RBK * CBW), tile = wave % (RBK * CBW);
    const int rb = tile / CBW, cbk = blockIdx.y * CBW + tile % CBW;
    const int n0 = blockIdx.x * SPW;

    // ---- staging: unit q = (input voxel of the group, 8 of the k-step's 16 channels) ----
    const __amdgpu_buffer_rsrc_t xrs = rsrc_s2(p.x, p.x_bytes);
    u32x4 ld[UPT][2];
    auto gload = [&](int ks) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int q = tid + 512 * u, vox = q >> 1, half = q & 1;
            const bool ok = n0 + vox / VI < p.N;
            const unsigned off = ok ? 4u * (unsigned)(((long)n0 * VI + vox) * CI + ks * 16 + half * 8) : 0x80000000u;
            ld[u][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0);
            ld[u][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(off + 16u), 0, 0);
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int q = tid + 512 * u, vox = q >> 1, half = q & 1;
            const int sm = vox / VI, vi = vox % VI;
            const int z = vi / (4 * G * G), y = (vi / (2 * G)) % (2 * G), xx = vi % (2 * G);
            const int cls = ((z & 1) << 2) | ((y & 1) << 1) | (xx & 1);
            const int rec = sm * VI + cls * VO + ((z >> 1) * G + (y >> 1)) * G + (xx >> 1);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ld[u][0][e]); v[4 + e] = __uint_as_float(ld[u][1][e]); }
            u32x4 o[3];
            cut8s(v, o);
            unsigned char* dst = patch + slot * SLOT + half * ARR + rec * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * 2 * ARR) = o[pl];
        }
    };
    for (int i = tid; i < 2 * 6 * 16; i += 512)             // zero records of every array of both slots
        *reinterpret_cast<u32x4*>(patch + (i >> 4) * ARR + (NV + (i & 15)) * 16) = u32x4{0u, 0u, 0u, 0u};
    gload(0);

    // ---- per-lane record address of each of this wave's entries (independent of the k-step) ----
    const int row = rb * 32 + l32, sm = row / VO, o = row % VO;
    const int oa = o / (G * G), ob = (o / G) % G, oc = o % G;
    const int zaddr = (NV + (row & 15)) * 16 + h * ARR;
    int sel[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int e = kw * NTW + j, t = e < 27 ? e : 13;    // the shortcut reads the centre tap's fragment (padding: unused)
        const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
        const int a = oa - (tz == 0), b = ob - (ty == 0), c = oc - (tx == 0);
        const int cls = ((tz != 1) << 2) | ((ty != 1) << 1) | (tx != 1);
        const bool ok = (a >= 0) & (b >= 0) & (c >= 0);
        sel[j] = ok ? (sm * VI + cls * VO + (a * G + b) * G + c) * 16 + h * ARR : zaddr;
    }

    const __amdgpu_buffer_rsrc_t wrs = rsrc_s2(p.wimg, p.wimg_bytes);
    const int w_voff = cbk * (3 * S2_WBLK) + lane * 16;
    bf16x8 bfr[RING][3];
    // flat step s = ks * NTW + j of this wave -> image offset of entry kw * NTW + j at k-step ks (out of range: zeros)
    auto woff = [&](int s) {
        const int ks = s / NTW, j = s % NTW;
        return (ks < KS && kw * NTW + j < NREAL) ? (unsigned)(((kw * NTW + j) * KS + ks) * WSTEP) : 0x80000000u;
    };
    auto wload = [&](unsigned so, auto SLOTc) {
        constexpr int SL = decltype(SLOTc)::value;
        const int vo = w_voff + (int)so;                      // (out of range rides in the CHECKED offset: zeros)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bfr[SL][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo + pl * S2_WBLK, 0, 0));
    };

    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};

    // first RING - 1 fragments in flight, then the patch
    wload(woff(0), std::integral_constant<int, 0>{});
    if constexpr (RING > 2) wload(woff(1), std::integral_constant<int, 1 % RING>{});
    if constexpr (RING > 3) wload(woff(2), std::integral_constant<int, 2 % RING>{});
    if constexpr (RING > 4) wload(woff(3), std::integral_constant<int, 3 % RING>{});
    if constexpr (RING > 5) wload(woff(4), std::integral_constant<int, 4 % RING>{});
    if constexpr (RING > 6) wload(woff(5), std::integral_constant<int, 5 % RING>{});
    lstore(0);
    if (KS > 1) gload(1);
    __syncthreads();

    for (int ks = 0; ks < KS; ++ks) {
        // k-step ks + 1 (in registers since the previous iteration) -> the other slot (last read during ks - 1, before the
        // barrier that ended it); then fetch k-step ks + 2
        if (ks + 1 < KS) lstore((ks + 1) & 1);
        if (ks + 2 < KS) gload(ks + 2);
        const unsigned char* pb = patch + (ks & 1) * SLOT;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int j = rd * RING + u;
                // fragment of step s + RING - 1 -> the slot the previous step freed
                const unsigned so = woff(ks * NTW + j + RING - 1);
                switch ((u + RING - 1) % RING) {
                    case 0: wload(so, std::integral_constant<int, 0>{}); break;
                    case 1: wload(so, std::integral_constant<int, 1 % RING>{}); break;
                    case 2: wload(so, std::integral_constant<int, 2 % RING>{}); break;
                    case 3: wload(so, std::integral_constant<int, 3 % RING>{}); break;
                    case 4: wload(so, std::integral_constant<int, 4 % RING>{}); break;
                    case 5: wload(so, std::integral_constant<int, 5 % RING>{}); break;
                    default: wload(so, std::integral_constant<int, 6 % RING>{}); break;
                }
                bf16x8 af[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(pb + sel[j] + pl * 2 * ARR);
                if (kw * NTW + j >= NREAL) continue;          // (wave-uniform) padding entry: nothing to add
                if (kw * NTW + j == NREAL - 1) {              // (wave-uniform) the shortcut entry
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bfr[u][PB[pr]], acc2, 0, 0, 0);
                } else {
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bfr[u][PB[pr]], acc, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- the KW parts meet in LDS (the patch is no longer read): part 0 adds parts 1.. in order, then the epilogue ----
    float (*red)[TILES][16][64] = reinterpret_cast<float (*)[TILES][16][64]>(patch);
    if (kw > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[kw - 1][tile][r][lane] = acc[r];
    }
    __syncthreads();
    const int col = cbk * 32 + l32;
    if (kw == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = acc[r];
#pragma unroll
            for (int k = 1; k < KW; ++k) t += red[k - 1][tile][r][lane];
            const int rv = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int n = n0 + rv / VO;
            if (n < p.N) p.hmid[((long)n * VO + rv % VO) * CO + col] = fmaxf(t, 0.f);
        }
    }
    if (kw == (NREAL - 1) / NTW) {                            // the wave group that holds the shortcut entry
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rv = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int n = n0 + rv / VO;
            if (n < p.N) p.r[((long)n * VO + rv % VO) * CO + col] = acc2[r];
        }
    }
}

// ---- forward image: entry e < 27: B[k = ci][n = co] = W[e][ci][co]; entry 27: Wds[ci][co] (8 consecutive ci: stride CO) ----
size_t s2_fwd_img_bytes(int Ci, int Co) { return (size_t)28 * (Ci / 16) * (Co / 32) * 3 * S2_WBLK; }

// several images in ONE launch (blockIdx.y = job): the engine re-cuts the images of an encoder's stride-2 fronts behind its
// SGD / momentum kernel instead of one prep launch in front of every call
constexpr int S2_PREP_JOBS = 8;
struct S2PrepJob { const float* w; const float* wds; unsigned char* img; unsigned char* dsimg; int Ci, Co, dgrad, pad; };
struct S2PrepBatch { S2PrepJob j[S2_PREP_JOBS]; unsigned char slot_tap[27]; };

__global__ __launch_bounds__(256) void s2_prep_batch_kernel(S2PrepBatch b) {
    const S2PrepJob& jb = b.j[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (!jb.dgrad) {
        const int KS = jb.Ci / 16, CB = jb.Co / 32;
        const long per_e = (long)KS * CB * 64;
        if (idx >= 28 * per_e) return;
        const int lane = (int)(idx & 63), cbk = (int)((idx >> 6) % CB), ks = (int)((idx / (64 * CB)) % KS), e = (int)(idx / per_e);
        const int co = cbk * 32 + (lane & 31), ci0 = ks * 16 + 8 * (lane >> 5);
        const float* src = (e < 27 ? jb.w + (long)e * jb.Ci * jb.Co : jb.wds) + (long)ci0 * jb.Co + co;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = src[(long)t * jb.Co];
        u32x4 o[3];
        cut8s(v, o);
        unsigned char* dst = jb.img + (((long)e * KS + ks) * CB + cbk) * (3 * S2_WBLK) + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * S2_WBLK) = o[pl];
        return;
    }
    // data-gradient image (s2_dgrad_prep_kernel): reduction = the convolution's output channels, columns = its input channels
    const int CR = jb.Co, CN = jb.Ci;
    const int KS = CR / 16, CB = CN / 32;
    const long per_tap = (long)KS * CB * 64;
    const long total = 27 * per_tap, total_ds = jb.wds ? per_tap : 0;
    if (idx >= total + total_ds) return;
    const bool ds = idx >= total;
    const long i = ds ? idx - total : idx;
    const int lane = (int)(i & 63), cbk = (int)((i >> 6) % CB), ks = (int)((i / (64 * CB)) % KS), slot = (int)(i / per_tap);
    const int ci = cbk * 32 + (lane & 31), co0 = ks * 16 + 8 * (lane >> 5);
    const float* src = ds ? jb.wds + (long)ci * CR + co0 : jb.w + ((long)b.slot_tap[slot] * CN + ci) * CR + co0;
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32x4 o[3];
    cut8s(v, o);
    unsigned char* dst = (ds ? jb.dsimg : jb.img) + (((long)(ds ? 0 : slot) * KS + ks) * CB + cbk) * (3 * S2_WBLK) + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * S2_WBLK) = o[pl];
}

// class order of the data-gradient image (heaviest first) and the image slot -> convolution tap table
void s2_dgrad_tables(unsigned char order[8], unsigned char tap_off[8], unsigned char slot_tap[27]) {
    const int ord[8] = {7, 3, 5, 6, 1, 2, 4, 0};
    int slot = 0;
    for (int oi = 0; oi < 8; ++oi) {
        const int c = ord[oi];
        order[oi] = (unsigned char)c;
        tap_off[c] = (unsigned char)slot;
        const int nt = 1 << (((c >> 2) & 1) + ((c >> 1) & 1) + (c & 1));
        for (int j = 0; j < nt; ++j) {
            int tap, dz, dy, dxx;
            s2_tap_of(c, j, &tap, &dz, &dy, &dxx);
            slot_tap[slot++] = (unsigned char)tap;
        }
    }
}

}  // namespace

extern "C" int mi_conv3d_s2_dgrad_usable(int N, int Gi, int Ci, int Co) {
    const char* off = getenv("MI_CONV_NO_DIRECT");
    if (off && atoi(off) != 0) return 0;
    const char* ar = getenv("MI_CONV_ARITH");
    if (ar && ar[0] == 'f') return 0;
    const char* no = getenv("MI_CONV_NO_S2");          // A/B switch: the two generic launches
    if (no && atoi(no) != 0) return 0;
    int G;
    return N >= 1 && s2_shape(Gi, Ci, Co, &G) && 4l * N * Gi * Gi * Gi * Ci < 0x7fff0000l ? 1 : 0;
}

extern "C" size_t mi_conv3d_s2_dgrad_workspace_bytes(int Ci, int Co) {
    return mi_align_up(s2_img_bytes(Ci, Co), 256) + mi_align_up(s2_dsimg_bytes(Ci, Co), 256);
}

/* dx (N, Gi, Gi, Gi, Ci) = data gradient of conv3d(k 3, stride 2, pad 1; w [27][Ci][Co]) w.r.t. its input from dh (N, Gi/2.., Co),
 * + the data gradient of the 1x1 stride-2 shortcut (w_ds [Ci][Co]) from dout when both are given; epilogue (.. + res) * (mask > 0).
 * The weight images are cut into `ws` by this call. */
static int s2_dgrad_launch(const float* dh, const float* dout, const unsigned char* img, const unsigned char* dsimg, float* dx,
                           const float* res, const float* mask, int N, int Gi, int Ci, int Co, hipStream_t s) {
    int G;
    if (!s2_shape(Gi, Ci, Co, &G)) return MI_E_UNSUPPORTED;
    S2DgradParams p = {};
    p.dh = dh; p.dout = dout; p.wimg = img; p.dsimg = dout ? dsimg : nullptr; p.dx = dx; p.res = res; p.mask = mask; p.N = N;
    p.a_bytes = (unsigned)(4l * N * G * G * G * Co); p.o_bytes = (unsigned)(4l * N * Gi * Gi * Gi * Ci);
    p.wimg_bytes = (unsigned)s2_img_bytes(Ci, Co); p.dsimg_bytes = (unsigned)s2_dsimg_bytes(Ci, Co);
    unsigned char slot_tap[27];
    s2_dgrad_tables(p.order, p.tap_off, slot_tap);
    if (G == 4 && Ci == 128) {
        // 64^3 crops, layer3.0: 8^3 -> 4^3 at 128 -> 256 channels - one sample per workgroup (64 rows), its 256 reduction channels cut four
        // ways over the wave groups, one column block of the 128 per workgroup (126 KB of LDS: one workgroup per CU)
        p.n_groups = N;
        hipLaunchKernelGGL((s2_dgrad_kernel<4, 256, 128, 1, 4, 8, 1>), dim3((unsigned)(N * 8), 4), dim3(512), 0, s, p);
    } else if (G == 4) {
        p.n_groups = N;
        const char* rg = getenv("MI_S2_RING");
        if (rg && atoi(rg) == 8) hipLaunchKernelGGL((s2_dgrad_kernel<4, 128, 64, 2, 2, 8, 2>), dim3((unsigned)(N * 8), 1), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((s2_dgrad_kernel<4, 128, 64, 2, 2, 4, 4>), dim3((unsigned)(N * 8), 1), dim3(512), 0, s, p);
    } else {
        p.n_groups = (N + 3) / 4;
        const char* rg = getenv("MI_S2_RING");
        if (rg && atoi(rg) == 4) hipLaunchKernelGGL((s2_dgrad_kernel<2, 256, 128, 2, 4, 4, 4>), dim3((unsigned)(p.n_groups * 8), 2), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((s2_dgrad_kernel<2, 256, 128, 2, 4, 8, 2>), dim3((unsigned)(p.n_groups * 8), 2), dim3(512), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

static int s2_prep_batch(const S2PrepJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += S2_PREP_JOBS) {
        S2PrepBatch b = {};
        unsigned char order[8], tap_off[8];
        s2_dgrad_tables(order, tap_off, b.slot_tap);
        const int m = std::min(S2_PREP_JOBS, n - i0);
        long most = 0;
        for (int i = 0; i < m; ++i) {
            b.j[i] = jobs[i0 + i];
            const S2PrepJob& j = b.j[i];
            const long thr = j.dgrad ? (27l + (j.wds ? 1 : 0)) * (j.Co / 16) * (j.Ci / 32) * 64 : 28l * (j.Ci / 16) * (j.Co / 32) * 64;
            most = std::max(most, thr);
        }
        hipLaunchKernelGGL(s2_prep_batch_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)m), dim3(256), 0, s, b);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}

extern "C" int mi_conv3d_s2_dgrad_f32(const float* dh, const float* dout, const float* w, const float* w_ds, float* dx,
                                      const float* res, const float* mask, int N, int Gi, int Ci, int Co, void* ws, size_t ws_bytes,
                                      mi_stream_t stream) {
    if (!dh || !w || !dx || !ws || ((dout == nullptr) != (w_ds == nullptr))) return MI_E_ARG;
    int G;
    if (!mi_conv3d_s2_dgrad_usable(N, Gi, Ci, Co) || !s2_shape(Gi, Ci, Co, &G)) return MI_E_UNSUPPORTED;
    if (ws_bytes < mi_conv3d_s2_dgrad_workspace_bytes(Ci, Co)) return MI_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* img = (unsigned char*)ws;
    unsigned char* dsimg = img + mi_align_up(s2_img_bytes(Ci, Co), 256);
    S2PrepJob jb = {w, w_ds, img, dsimg, Ci, Co, 1, 0};
    int rc = s2_prep_batch(&jb, 1, s);
    if (rc) return rc;
    return s2_dgrad_launch(dh, dout, img, dsimg, dx, res, mask, N, Gi, Ci, Co, s);
}

/* The same with images the caller keeps (mi_conv3d_s2_prep cut them: `img` holds mi_conv3d_s2_dgrad_workspace_bytes(Ci, Co) bytes,
 * the shortcut's image behind the convolution's). */
extern "C" int mi_conv3d_s2_dgrad_img_f32(const float* dh, const float* dout, const void* img, float* dx, const float* res,
                                          const float* mask, int N, int Gi, int Ci, int Co, mi_stream_t stream) {
    if (!dh || !img || !dx) return MI_E_ARG;
    if (!mi_conv3d_s2_dgrad_usable(N, Gi, Ci, Co)) return MI_E_UNSUPPORTED;
    const unsigned char* im = (const unsigned char*)img;
    return s2_dgrad_launch(dh, dout, im, im + mi_align_up(s2_img_bytes(Ci, Co), 256), dx, res, mask, N, Gi, Ci, Co, (hipStream_t)stream);
}

/* Cut the images of n stride-2 fronts in one launch per 8: w[i] ([27][Ci][Co]), w_ds[i] ([Ci][Co]; may be NULL for a data-gradient
 * image without shortcut), img[i] (forward: mi_conv3d_s2_fwd_workspace_bytes, data gradient: mi_conv3d_s2_dgrad_workspace_bytes),
 * dgrad[i] 0 / 1.  Host arrays of device pointers. */
extern "C" int mi_conv3d_s2_prep(const float* const* w, const float* const* w_ds, void* const* img, const int* ci, const int* co,
                                 const int* dgrad, int n, mi_stream_t stream) {
    if (!w || !w_ds || !img || !ci || !co || !dgrad || n < 0 || n > 64) return MI_E_ARG;
    S2PrepJob jobs[64];
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !img[i] || (!dgrad[i] && !w_ds[i])) return MI_E_ARG;
        unsigned char* im = (unsigned char*)img[i];
        jobs[i] = {w[i], w_ds[i], im, dgrad[i] ? im + mi_align_up(s2_img_bytes(ci[i], co[i]), 256) : nullptr, ci[i], co[i], dgrad[i], 0};
    }
    return s2_prep_batch(jobs, n, (hipStream_t)stream);
}

/* Forward of a stride-2 block front of the MoCo-3D encoder (models/networks/moco_encoder_3d.py:55-84, 257-272): hmid = relu(conv3d(x;
 * w [27][Ci][Co], k 3, stride 2, pad 1)) and the shortcut r = conv3d(x; w_ds [Ci][Co], k 1, stride 2) in one launch.  Shapes: the
 * encoder's two at 32^3 crops (Gi 8, 64 -> 128 and Gi 4, 128 -> 256) and layer3.0 of the 64^3 crops (Gi 8, 128 -> 256), bf16x3 arithmetic; otherwise MI_E_UNSUPPORTED (run mi_conv3d_fwd_f32
 * twice).  The weight image is cut into `ws` (mi_conv3d_s2_fwd_workspace_bytes) by this call. */
extern "C" int mi_conv3d_s2_fwd_usable(int N, int Gi, int Ci, int Co) {
    const char* no = getenv("MI_CONV_NO_S2FWD");       // A/B switch: the generic launches
    if (no && atoi(no) != 0) return 0;
    return mi_conv3d_s2_dgrad_usable(N, Gi, Ci, Co);
}

extern "C" size_t mi_conv3d_s2_fwd_workspace_bytes(int Ci, int Co) { return mi_align_up(s2_fwd_img_bytes(Ci, Co), 256); }

static int s2_fwd_launch(const float* x, const unsigned char* img, float* hmid, float* r, int N, int Gi, int Ci, int Co, hipStream_t s) {
    int G;
    if (!s2_shape(Gi, Ci, Co, &G)) return MI_E_UNSUPPORTED;
    S2FwdParams p = {x, img, hmid, r, N, (unsigned)(4l * N * Gi * Gi * Gi * Ci), (unsigned)s2_fwd_img_bytes(Ci, Co)};
    const char* nar = getenv("MI_S2FWD_NARROW");        // tuning: two column blocks per workgroup (half the workgroups)
    const bool narrow = nar && atoi(nar);
    if (G == 4 && Ci == 128) {
        // 64^3 crops, layer3.0 (8^3 -> 4^3, 128 -> 256): a sample's 64 rows x one block of 32 output channels per workgroup
        hipLaunchKernelGGL((s2_fwd_kernel<4, 128, 256, 2, 1, 4, 1>), dim3((unsigned)N, 8), dim3(512), 0, s, p);
    } else if (G == 4) {
        if (narrow) hipLaunchKernelGGL((s2_fwd_kernel<4, 64, 128, 2, 2, 2, 2>), dim3((unsigned)N, 2), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((s2_fwd_kernel<4, 64, 128, 2, 1, 4, 2>), dim3((unsigned)N, 4), dim3(512), 0, s, p);
    } else {
        if (narrow) hipLaunchKernelGGL((s2_fwd_kernel<2, 128, 256, 1, 2, 4, 2>), dim3((unsigned)((N + 3) / 4), 4), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((s2_fwd_kernel<2, 128, 256, 1, 1, 8, 2>), dim3((unsigned)((N + 3) / 4), 8), dim3(512), 0, s, p);
    }
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" int mi_conv3d_s2_fwd_f32(const float* x, const float* w, const float* w_ds, float* hmid, float* r, int N, int Gi, int Ci,
                                    int Co, void* ws, size_t ws_bytes, mi_stream_t stream) {
    if (!x || !w || !w_ds || !hmid || !r || !ws) return MI_E_ARG;
    if (!mi_conv3d_s2_fwd_usable(N, Gi, Ci, Co)) return MI_E_UNSUPPORTED;
    if (ws_bytes < mi_conv3d_s2_fwd_workspace_bytes(Ci, Co)) return MI_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    S2PrepJob jb = {w, w_ds, (unsigned char*)ws, nullptr, Ci, Co, 0, 0};
    int rc = s2_prep_batch(&jb, 1, s);
    if (rc) return rc;
    return s2_fwd_launch(x, (const unsigned char*)ws, hmid, r, N, Gi, Ci, Co, s);
}

/* The same with an image the caller keeps (mi_conv3d_s2_prep cut it). */
extern "C" int mi_conv3d_s2_fwd_img_f32(const float* x, const void* img, float* hmid, float* r, int N, int Gi, int Ci, int Co,
                                        mi_stream_t stream) {
    if (!x || !img || !hmid || !r) return MI_E_ARG;
    if (!mi_conv3d_s2_fwd_usable(N, Gi, Ci, Co)) return MI_E_UNSUPPORTED;
    return s2_fwd_launch(x, (const unsigned char*)img, hmid, r, N, Gi, Ci, Co, (hipStream_t)stream);
}
