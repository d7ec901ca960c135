// Weight gradients of the small-volume convolutions from PRE-CUT operand images (round 4, OPT-IN: MI_PAIRW=1 with
// MI_PAIR_WGRAD_MAXOUT=4): layer2 (128 -> 128 on 4^3), layer2.0's stride-2 convolution and 1 x 1 x 1 shortcut and the 2^3 outputs of
// layer3 (cet_pick/models/networks/moco_encoder_3d.py:55-84,170-178).
//
//   dW[tap][ci][co] = sum over samples n and over the (input voxel vi, output voxel vo) pairs the tap connects of X[n][vi][ci] dY[n][vo][co]
//
// pair_wgrad_kernel (conv_cube2.hip) fetches the two 64-sample x 64-channel blocks of a pair as f32, cuts them into bf16x3 planes in
// registers (100 VALU operations per thread), writes the planes to LDS and reads fragments back through the transposing LDS read - for
// EVERY pair a block takes part in (27 on a stride-1 volume).  Here the cut and the transposition happen ONCE per tensor, in
// pairw_prep_kernel: X and dY are rewritten as images
//   [voxel][64-channel block][64-sample chunk][half chunk 2][plane 3][k-step 2][k half 2][channel 64][8 samples] bf16   (24 KB per block)
// whose blocks are exactly the LDS image the MFMA fragments want (a lane's 8 consecutive samples are 16 contiguous bytes; a wave's
// ds_read_b128 covers 1 KB without a bank conflict).  The gradient kernel then only MOVES: a workgroup owns a (tap, tile, segment of the
// tap's chain of pairs), copies the blocks of a pair global -> LDS with LDS-DMA (global_load_lds_dwordx4: no registers, no VALU) two
// iterations ahead of the products (three stages of 48 KB, waits counted by hand), one barrier per iteration.  Segments write slabs
// that the caller's reduce sums (as the implicit GEMM's split-K does).
//
// MEASURED (r04_experiments.txt item 26; layer2 at batch 64): 23 - 25 us + 6 - 8 us for the images, against ~30 us for the implicit GEMM
// it would replace - no gain per convolution, so the dispatch keeps the implicit GEMM and this file stays an opt-in (tested) form.
// Where its time goes (elimination): an empty launch of its 432 workgroups (144 KB of LDS: one per CU, two rounds) 7.9 us, the copies
// +3, the products +9 (5.9 at the sustained MFMA rate), the 28 MB of slab stores +3.5.  Neither the depth of the copy pipeline (two or
// three stages) nor the bytes copied per product (64 x 64 against 128 x 128 tiles) moved it.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int IMG_PLANE = 4 * 2 * 64 * 16;      // [k-step][k half][channel] x 16 bytes
constexpr int IMG_BLOCK = 3 * IMG_PLANE;        // 24,576: one (voxel, 64 channels, 64 samples) block, three planes
constexpr int STAGE = 2 * IMG_BLOCK;            // X block + dY block
constexpr int PW_MAXTAP = 27;

// exact three-way bf16 cut of 8 f32 (truncation, as conv_igemm.hip / conv_direct3.hip / conv_cube2.hip)
__device__ __forceinline__ void cut8p(const float (&v)[8], u32x4 (&o)[3]) {
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

struct PairwPrepParams {
    const float* t[2];        // X (N, V0, C0), dY (N, V1, C1)
    unsigned char* img[2];
    int V[2], C[2];
    int N, chunks;
    int nblk0;                // blocks of tensor 0: V0 * (C0 / 64) * chunks
};

// one workgroup per image block: 64 samples x 64 channels f32 in (rows of 256 bytes), transposed through LDS, cut, 24 KB out
__global__ __launch_bounds__(256) void pairw_prep_kernel(PairwPrepParams p) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    int b = blockIdx.x, op = 0;
    if (b >= p.nblk0) { op = 1; b -= p.nblk0; }
    const int V = p.V[op], C = p.C[op], cbs = C >> 6;
    const int chunk = b % p.chunks, cb = (b / p.chunks) % cbs, v = b / (p.chunks * cbs);
    const float* src = p.t[op];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int nl = pass * 16 + (tid >> 4), c4 = (tid & 15) * 4, n = chunk * 64 + nl;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.N) val = *reinterpret_cast<const float4*>(src + ((long)n * V + v) * C + cb * 64 + c4);
        tile[nl][c4] = val.x; tile[nl][c4 + 1] = val.y; tile[nl][c4 + 2] = val.z; tile[nl][c4 + 3] = val.w;
    }
    __syncthreads();
    unsigned char* dst = p.img[op] + (long)b * IMG_BLOCK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = tid + 256 * j;                     // unit = (k-step, k half, channel): 8 consecutive samples of one channel
        const int ch = u & 63, s0 = (u >> 6) * 8;        // (u >> 6) = 2 * k-step + k half: samples s0 .. s0 + 7
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = tile[s0 + e][ch];
        u32x4 o[3];
        cut8p(vv, o);
        // [half of the chunk 2][plane 3][k-step 2][k half 2][channel 64] x 16 bytes
        const int off = (u >> 8) * (IMG_BLOCK / 2) + (u & 255) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + off + pl * 4096) = o[pl];
    }
}

struct PairwParams {
    const unsigned char* ximg;    // [VI][CI / 64][chunks] blocks
    const unsigned char* yimg;    // [VO][CO / 64][chunks] blocks
    float* dw;                    // [ntaps][CI][CO]: final (S == 1) or slab 0 of S slabs
    int chunks, CI, CO, ntaps;
    int Di, Do, stride, pad, ks;
    int S;
    int dbg;                      // measurement aid (MI_PAIRW_DEBUG): 1 no tile stores, 2 no products, 4 no copies (results are wrong)
    long slab_stride;             // floats between slabs
    unsigned char order[PW_MAXTAP];       // taps, heaviest first
    unsigned short cnt[PW_MAXTAP];        // pairs of a tap
    unsigned char lo[PW_MAXTAP][3], len[PW_MAXTAP][3];   // [tap][z, y, x]: the box of output voxels the tap connects
};

// six LDS-DMA instructions of 1 KB each: bytes 0 .. 6143 behind `src` (this lane's 16 bytes of each KB) to the LDS byte address `dst`
// (wave-uniform).  Written as asm so that the waits are OURS: a DMA is counted on vmcnt, and the compiler's bookkeeping would drain
// every one of them (vmcnt(0)) at the next barrier or LDS read - here two stages stay in flight behind the one being multiplied.
// The instruction offset moves both addresses; M0 = the LDS base (written in the statement that uses it; s_nop: M0 hazard).
__device__ __forceinline__ void glds6(const unsigned char* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, off\n\t"
                 "global_load_lds_dwordx4 %4, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst), "s"(dst + 4096u), "v"(src + 4096) : "memory");
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// B2 = 1: a 64 ci x 64 co tile, a stage = the two 24 KB blocks of a pair (64 samples: four k-steps); eight waves = 2 halves of co x
//         4 k-steps, the k-parts meet in LDS at the end.
// B2 = 2: a 128 x 128 tile (layer2: all of it), a stage = HALF a chunk (32 samples: two k-steps) of the two X and two dY blocks of a
//         pair - 48 KB as well, for four times the products: the L2 -> LDS path (~34 bytes a clock per CU) is what bounds this kernel.
//         Eight waves = the eight 64 x 32 sub-tiles, both k-steps each; tiles are written from the accumulators.
// Three stages: the copies run two iterations ahead of the products.
template <int B2>
__global__ __launch_bounds__(512, 1) void pairw_kernel(PairwParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int ncb = p.CO >> 6, nib = p.CI >> 6, tiles = (nib / B2) * (ncb / B2);
    const int rem = blockIdx.x % tiles, ts = blockIdx.x / tiles;
    const int tap = p.order[ts / p.S], seg = ts % p.S;
    const int ib = (rem / (ncb / B2)) * B2, cb = (rem % (ncb / B2)) * B2;           // first 64-channel block of the tile
    constexpr int HALVES = B2;                                                        // stages per (pair, chunk)
    const int all = p.cnt[tap] * p.chunks * HALVES, per = (all + p.S - 1) / p.S;
    const int it0 = seg * per, total = min(all, it0 + per);
    const int tz = tap / (p.ks * p.ks), ty = (tap / p.ks) % p.ks, tx = tap % p.ks;
    const int lz = p.lo[tap][0], ly = p.lo[tap][1], lx = p.lo[tap][2], ny = p.len[tap][1], nx = p.len[tap][2];

    // LDS-DMA: wave w moves bytes 6144 w .. 6144 w + 6143 of a stage.  B2 = 1: waves 0-3 the X block, 4-7 the dY block.
    // B2 = 2: pieces of 12 KB (half a chunk of one block): X ib, X ib + 1, dY cb, dY cb + 1; wave w the half w & 1 of piece w >> 1.
    int f_half = it0 % HALVES, f_chunk = (it0 / HALVES) % p.chunks, f_px, f_py, f_pz;
    { const int pr = it0 / (HALVES * p.chunks); f_px = pr % nx; f_py = (pr / nx) % ny; f_pz = pr / (nx * ny); }
    const unsigned lds0 = (unsigned)(size_t)(lds_u8*)lds;
    auto dma = [&](int stage_off) {
        const int oz = lz + f_pz, oy = ly + f_py, ox = lx + f_px;
        const int vo = (oz * p.Do + oy) * p.Do + ox;
        const int vi = ((p.stride * oz + tz - p.pad) * p.Di + p.stride * oy + ty - p.pad) * p.Di + p.stride * ox + tx - p.pad;
        const unsigned char* src;
        if (B2 == 1) {
            const unsigned char* blk = wave < 4 ? p.ximg + ((long)(vi * nib + ib) * p.chunks + f_chunk) * IMG_BLOCK
                                                : p.yimg + ((long)(vo * ncb + cb) * p.chunks + f_chunk) * IMG_BLOCK;
            src = blk + (wave & 3) * 6144;
        } else {
            const int piece = wave >> 1;
            const unsigned char* blk = piece < 2 ? p.ximg + ((long)(vi * nib + ib + piece) * p.chunks + f_chunk) * IMG_BLOCK
                                                 : p.yimg + ((long)(vo * ncb + cb + piece - 2) * p.chunks + f_chunk) * IMG_BLOCK;
            src = blk + f_half * (IMG_BLOCK / 2) + (wave & 1) * 6144;
        }
        if (!(p.dbg & 4)) glds6(src + lane * 16, lds0 + stage_off + wave * 6144);
        if (++f_half == HALVES) { f_half = 0;
            if (++f_chunk == p.chunks) { f_chunk = 0; if (++f_px == nx) { f_px = 0; if (++f_py == ny) { f_py = 0; ++f_pz; } } } }
    };

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    // image of half a chunk: [plane 3][k-step 2][k half 2][channel 64] x 16 bytes = 12 KB; a block = two of them
    constexpr int HPL = 4096, HBLK = 12288;
    // B2 = 1: wave = (wn = wave & 1: co half, kq = wave >> 1: k-step of the chunk).  B2 = 2: wave = (sm = wave >> 2: ci block, sn = wave & 3:
    // 32-column block of the 128 co), k-steps 0 and 1 of the half
    const int wn = B2 == 1 ? (wave & 1) : (wave & 3), kq = wave >> 1, sm = wave >> 2;
    const int a_off = B2 == 1 ? (kq >> 1) * HBLK + (kq & 1) * 2048 + h * 1024 + l32 * 16
                              : sm * HBLK + h * 1024 + l32 * 16;
    const int b_off = B2 == 1 ? IMG_BLOCK + (kq >> 1) * HBLK + (kq & 1) * 2048 + h * 1024 + (wn * 32 + l32) * 16
                              : (2 + (wn >> 1)) * HBLK + h * 1024 + ((wn & 1) * 32 + l32) * 16;
    auto multiply = [&](int stage_off) {
#pragma unroll
        for (int ks = 0; ks < B2; ++ks) {
            bf16x8 a0[3], a1[3], bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned char* ap = lds + stage_off + a_off + pl * HPL + ks * 2048;
                a0[pl] = *reinterpret_cast<const bf16x8*>(ap);
                a1[pl] = *reinterpret_cast<const bf16x8*>(ap + 512);
                bf[pl] = *reinterpret_cast<const bf16x8*>(lds + stage_off + b_off + pl * HPL + ks * 2048);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[pr]], bf[PB[pr]], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[pr]], bf[PB[pr]], acc[1], 0, 0, 0);
            }
        }
    };

    // stage of iteration it = (it - it0) % 3.  At the head of iteration it the copies of it and it + 1 are in flight (six instructions
    // each, per wave): wait for the older six, meet - every wave's part has landed, every wave is done with the stage of it - 1 -, send
    // the copies of it + 2 into that stage, multiply.
    int st = 0;                                          // byte offset of iteration it's stage
    if (it0 < total) dma(0);
    if (it0 + 1 < total) dma(STAGE);
    for (int it = it0; it < total; ++it) {
        if (it + 1 < total) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int st2 = st >= STAGE ? st - STAGE : st + 2 * STAGE;       // (st + 2 stages) mod 3 stages
        if (it + 2 < total) dma(st2);
        if (!(p.dbg & 2)) multiply(st);
        st = st == 2 * STAGE ? 0 : st + STAGE;
    }
    if (p.dbg & 1) return;
    if (B2 == 2) {
        // the wave's 64 x 32 sub-tile, from the accumulators: C/D layout col = lane & 31 (co), row = ci
        float* out = p.dw + seg * p.slab_stride + ((long)tap * p.CI + (ib + sm) * 64) * p.CO + cb * 64 + 32 * wn + l32;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(long)(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * p.CO] = acc[rb][r];
        return;
    }
    // B2 = 1: the four k-parts of a tile meet in LDS (fixed order 0 + 1 + 2 + 3); wave (wn, kq) finishes eight of the 32 rows of its half
    __syncthreads();
    float (*red)[32][64] = reinterpret_cast<float (*)[32][64]>(lds);          // [wave][register][lane]: 64 KB
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][rb * 16 + r][lane] = acc[rb][r];
    __syncthreads();
    float* out = p.dw + seg * p.slab_stride + ((long)tap * p.CI + ib * 64) * p.CO + cb * 64 + 32 * wn + l32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = kq * 8 + j, rb = idx >> 4, r = idx & 15;
        const float t = ((red[wn][idx][lane] + red[2 + wn][idx][lane]) + red[4 + wn][idx][lane]) + red[6 + wn][idx][lane];
        out[(long)(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * p.CO] = t;       // C/D layout: col = lane & 31 (co), row = ci
    }
}

struct PwGeom { int ntaps, maxcnt, Do; };
PwGeom pw_geom(int Di, int k, int stride, PairwParams* p) {
    const int pad = k == 3 ? 1 : 0, Do = (Di + 2 * pad - k) / stride + 1;
    PwGeom r = {k * k * k, 0, Do};
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
    }
    return r;
}

}  // namespace

// 128 x 128 tiles (B2 = 2) where both channel counts allow it; MI_PAIRW_B2=1 keeps 64 x 64
int mi_pairw_b2(int Ci, int Co) {
    const char* e = getenv("MI_PAIRW_B2");
    if (e && atoi(e) == 1) return 1;
    return (Ci % 128 == 0 && Co % 128 == 0) ? 2 : 1;
}

// segments of a tap's chain of (pair, sample chunk) iterations: about eight iterations per workgroup, at most 16 segments
int mi_pairw_splits(int N, int Di, int Ci, int Co, int k, int stride) {
    const PwGeom g = pw_geom(Di, k, stride, nullptr);
    const char* e = getenv("MI_PAIRW_SPLITS");
    if (e && atoi(e) > 0) return std::min(atoi(e), 64);
    const int b2 = mi_pairw_b2(Ci, Co);
    const int iters = g.maxcnt * ((N + 63) / 64) * b2, tiles = (Ci / 64 / b2) * (Co / 64 / b2);
    int S = (iters + 7) / 8;
    if (g.ntaps * tiles * S < 256) S = (iters + 3) / 4;
    return std::max(1, std::min(S, 16));
}

// workspace: [S slabs (S > 1)] [X image] [dY image]
size_t mi_pairw_workspace_bytes(int N, int Di, int Ci, int Co, int k, int stride) {
    const PwGeom g = pw_geom(Di, k, stride, nullptr);
    const int S = mi_pairw_splits(N, Di, Ci, Co, k, stride), chunks = (N + 63) / 64;
    const size_t slabs = S > 1 ? mi_align_up(sizeof(float) * (size_t)S * g.ntaps * Ci * Co, 256) : 0;
    return slabs + (size_t)IMG_BLOCK * chunks * ((size_t)Di * Di * Di * (Ci / 64) + (size_t)g.Do * g.Do * g.Do * (Co / 64));
}

// dW through the pre-cut images.  S == 1: `dwt` is final; S > 1: S slabs at the head of `ws`, the caller sums them.
int mi_pairw_launch(const float* x, const float* dy, float* dwt, void* ws, int N, int Di, int Ci, int Co, int k, int stride, hipStream_t s) {
    PairwParams p = {};
    const PwGeom g = pw_geom(Di, k, stride, &p);
    const int S = mi_pairw_splits(N, Di, Ci, Co, k, stride), chunks = (N + 63) / 64;
    const int VI = Di * Di * Di, VO = g.Do * g.Do * g.Do;
    const size_t slabs = S > 1 ? mi_align_up(sizeof(float) * (size_t)S * g.ntaps * Ci * Co, 256) : 0;
    unsigned char* ximg = (unsigned char*)ws + slabs;
    unsigned char* yimg = ximg + (size_t)IMG_BLOCK * chunks * VI * (Ci / 64);
    PairwPrepParams q = {};
    q.t[0] = x; q.t[1] = dy; q.img[0] = ximg; q.img[1] = yimg; q.V[0] = VI; q.V[1] = VO; q.C[0] = Ci; q.C[1] = Co;
    q.N = N; q.chunks = chunks; q.nblk0 = VI * (Ci / 64) * chunks;
    hipLaunchKernelGGL(pairw_prep_kernel, dim3((unsigned)(q.nblk0 + VO * (Co / 64) * chunks)), dim3(256), 0, s, q);
    MI_RETURN_IF_LAUNCH_FAILED();
    p.ximg = ximg; p.yimg = yimg; p.dw = S > 1 ? (float*)ws : dwt;
    p.chunks = chunks; p.CI = Ci; p.CO = Co; p.ntaps = g.ntaps;
    p.Di = Di; p.Do = g.Do; p.stride = stride; p.pad = k == 3 ? 1 : 0; p.ks = k; p.S = S; p.slab_stride = (long)g.ntaps * Ci * Co;
    int n = 0;
    for (int want = g.maxcnt; want >= 0; --want)
        for (int t = 0; t < g.ntaps; ++t)
            if (p.cnt[t] == want) p.order[n++] = (unsigned char)t;
    const int b2 = mi_pairw_b2(Ci, Co);
    const int tiles = (Ci / 64 / b2) * (Co / 64 / b2), grid = g.ntaps * tiles * S;
    { const char* d = getenv("MI_PAIRW_DEBUG"); p.dbg = d ? atoi(d) : 0; }
    if (b2 == 2) hipLaunchKernelGGL(pairw_kernel<2>, dim3((unsigned)grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(pairw_kernel<1>, dim3((unsigned)grid), dim3(512), 0, s, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
