// 3-D convolution as implicit GEMM on the gfx950 matrix cores, f32 in / f32 accumulate, in two arithmetics:
// BF3 = false: v_mfma_f32_32x32x2_f32 - bit-for-bit an fmaf chain; BF3 = true (the default, MI_CONV_ARITH): the same
// products formed on the bf16 pipe from an exact three-way bf16 cut of both operands (see the kernel's header below).
// Either way parity with the fp32 reference holds.
//
// Replaces the nn.Conv3d / nn.Linear calls of the reference encoders
// (cet_pick/models/networks/moco_encoder_3d.py:40-84,156-236) in forward, data-gradient and
// weight-gradient form.  Activations are channels-last (N,D,H,W,C); weights live as
// [tap][Cin][Cout] (tap = (kd,kh,kw) flattened) - the Python side exposes that storage to
// state_dict() as a permuted (Cout,Cin,kd,kh,kw) view, so checkpoints keep the reference layout.
//
//   FWD   Y[m,co]  = sum_{tap,ci} X[src(m,tap),ci] * W[tap][ci][co]          m = output voxel
//   DGRAD dX[m,ci] = sum_{tap,co} dY[srcT(m,tap),co] * W[tap][ci][co]        m = input voxel
//   WGRAD dW[tap][ci][co] = sum_m X[src(m,tap),ci] * dY[m,co]                m = output voxel
//
// One 256-thread workgroup (4 waves, 2 x 2) owns a BM x 64 tile of the GEMM output and walks the
// reduction in BK-deep slices (16 or 32), double-buffered in LDS with register-staged prefetch
// (one barrier per slice).  Each operand is kept in LDS in the orientation its global layout is
// contiguous in:
//   "RowK" [row][k] (+4 pad): fragments by ds_read_b128            (im2col rows, W in DGRAD)
//   "KRow" [k][row]:          fragments by ds_read_b32             (W in FWD, dY/X in WGRAD)
// Within a slice lane-half h of the wave owns k = h*BK/2 .. for BOTH operands, which turns the
// 32x32x2 MFMA's (k = lane>>5) operand map into contiguous LDS reads; the reduction order inside a
// slice is therefore permuted (irrelevant beyond fp32 rounding).
//
// Gather cost is kept off the matrix pipe's critical path: every im2col row keeps ONE base pointer
// and a packed per-axis validity mask, a tap adds a wave-uniform offset, loads are branchless
// (invalid lanes read a safe address and select 0), and the (tap, channel) cursor advances
// incrementally.  A strided DGRAD is decomposed by output-coordinate parity class so that only the
// taps that can reach a row are multiplied (k=3, s=2: 27 tap-rows instead of 216).
// Small-M layers are split along the reduction (grid.z) into fp32 slabs, summed by a second
// kernel that also applies the epilogue (deterministic, no atomics).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

// conv_stem.hip: direct kernels for the 7^3 stride-2 stem; MI_E_UNSUPPORTED = shape declined, take the generic path
int mi_stem7_fwd(const float* x, const float* w, float* y, const float* res, int relu, int N, int D, int H, int W,
                 int Co, int bf16x3, void* ws, size_t ws_bytes, hipStream_t s, double* sums = nullptr);
size_t mi_stem7_fwd_workspace_bytes();
size_t mi_stem7_fwd_stats_workspace_bytes(int N, int D, int H, int W);
size_t mi_stem7_wgrad_workspace_bytes(int N, int D, int H, int W, int Co);
int mi_stem7_wgrad(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Co, int bf16x3, void* ws,
                   size_t ws_bytes, hipStream_t s);

// conv_direct3.hip: patch-resident direct kernels for the 3^3 / stride 1 convolutions of layer1 (64 -> 64 channels on 8 x 8
// planes: forward, data and weight gradient) and layer2 (128 -> 128 on 4 x 4 x 4: forward, data gradient) of the MoCo-3D
// encoder, bf16x3 arithmetic only
int mi_direct3_kind(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph,
                    int pw, int dd, int dh, int dw);
size_t mi_direct3_wimg_bytes(int channels);
size_t mi_direct3_slab_bytes(int N, int channels);
int mi_direct3_prep(const float* const* w, void* const* img, const int* dgrad, const int* channels, int n, hipStream_t s);
int mi_direct3_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                      int D, hipStream_t s);
size_t mi_direct3_wimg_bytes_kind(int kind);
int mi_direct3_prep_kind(const float* const* w, void* const* img, const int* dgrad, const int* kinds, int n, hipStream_t s);
int mi_direct3s_launch256(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                          hipStream_t s);
int mi_direct3h_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                       int D, int H, int W, hipStream_t s);
int mi_direct3_launch128(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                         int D, hipStream_t s);
int mi_direct3s_launch(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N,
                       hipStream_t s);
int mi_direct3_finish_slabs(const float* slabs, int n_slabs, long out_elems, float* out, const float* res, const float* mask,
                            int relu, hipStream_t s);
size_t mi_direct3_wgrad_slab_bytes();
int mi_direct3_wgrad_splits();
int mi_direct3_wgrad_launch(const float* x, const float* dy, float* slabs, int N, int D, hipStream_t s);
int mi_direct3_wgrad_batch_max();
int mi_direct3_wgrad_batch_splits(int nb);
int mi_direct3_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* slabs, int nb, int N, int D, hipStream_t s);
size_t mi_direct3s_wgrad_slab_bytes();
int mi_direct3s_wgrad_splits(int nb);
int mi_direct3s_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* slabs, int nb, int N, hipStream_t s);
int mi_direct3t_wgrad_launch(const float* x, const float* dy, float* slabs, int N, int D, int H, int W, hipStream_t s);
int mi_direct3x_wgrad_splits(int kind);
size_t mi_direct3x_wgrad_slab_bytes(int kind);
int mi_direct3x_wgrad_launch(int kind, const float* x, const float* dy, float* slabs, int N, int D, hipStream_t s);
int mi_pair_wgrad_batch_max();
int mi_pair_wgrad_launch_batch(const float* const* xs, const float* const* dys, float* const* dws, float* const* slabs, int nb, int N, int Di,
                               int Ci, int Co, int k, int stride, hipStream_t s);
// conv_cube2.hip: 3^3 convolutions on 2 x 2 x 2 volumes (layer3 / feature_3d) as a dense GEMM with register-staged operands
bool mi_cube2_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw,
                     int dd, int dh, int dw);
size_t mi_cube2_slab_bytes(int N, int C);
int mi_cube2_splits();
int mi_cube2_launch(int dgrad, const float* a, const float* w, float* slabs, int N, int C, hipStream_t s, const Cube2Final* fin = nullptr);
// ... and the small dense products of the Linear layers (register-staged, final in one launch)
bool mi_pair_wgrad_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw,
                          int dd, int dh, int dw);
int mi_pair_wgrad_splits(int N, int Di, int Ci, int Co, int k, int stride);
size_t mi_pair_wgrad_slab_bytes(int N, int Di, int Ci, int Co, int k, int stride);
int mi_pair_wgrad_launch(const float* x, const float* dy, float* dwt, float* slabs, int N, int Di, int Ci, int Co, int k, int stride,
                         hipStream_t s);
bool mi_small_gemm_usable(long M, long N, long K);
int mi_small_gemm_launch(const float* a, long lda_m, long lda_k, long a_elems, const float* b, long ldb_k, long ldb_n,
                         long b_elems, const float* bias, float* c, int M, int N, int K, hipStream_t s,
                         const MiSmallGemmBN* bn = nullptr);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int NTHREADS = 256;
constexpr int MAX_CLASSES = 27;    // parity classes (stride 2: 8) or border classes (3 runs per axis: 27)

// Border class: a box of row positions (pos0 .. pos0+cnt-1 per axis) that all see the same valid tap range
// (tlo .. tlo+tcnt-1 per axis).  Rows next to the zero padding multiply zeros for the taps that fall outside the
// tensor (a 3^3 window on a 2^3 volume: 19 of 27 taps); grouping rows by border class removes those taps from the
// reduction altogether instead of gathering zeros for them.
struct BorderClass { short pos0[3], cnt[3], tlo[3], tcnt[3]; };
constexpr int LUT_TAPS = 352;      // 7^3 = 343 rounded up to a multiple of 32
constexpr int LUT_INVALID = 0x070707;   // bit 7 of a per-axis mask byte is never set (k <= 7)

constexpr int MI_WGRAD_BATCH_MAX = 4;
struct ConvParams {
    const float* a_src;   // gathered tensor: X (FWD, WGRAD) or dY (DGRAD)
    const float* b_src;   // W (FWD, DGRAD) or dY (WGRAD)
    float* out;           // Y / dX / dW, or the split-K slabs
    const float* res;     // epilogue: out = act(acc + res)         (may be null)
    int res_bcast;        // 1: res is one row of Ncols values added to every output row (a Linear layer's bias)
    const float* mask;    // epilogue: out *= (mask > 0)            (may be null)
    int relu;
    int N, Dg, Hg, Wg, Cg;         // gathered tensor grid / channels
    int Dr, Hr, Wr;                // row grid: FWD/WGRAD output voxels, DGRAD input voxels
    unsigned mgW, mgH, mgD;        // branch-free division of a row index (< 2^31) by Wr, Hr, Dr:
    int shW, shH, shD;             //   q = (uint64(n) * mg) >> sh   (Granlund-Montgomery, N = 31)
    int kd, kh, kw, stride, pd, ph, pw;   // window / zero padding per axis (2-D convs: kd = 1, pd = 0, D = 1)
    int dd, dh, dw;                // dilation per axis (1 unless stride == 1 and Ci > 1)
    int Ci, Co;                    // conv channels (weights are [tap][Ci][Co])
    long M;                        // GEMM rows (all classes)
    int Ncols;                     // GEMM cols
    unsigned a_bytes, b_bytes;     // extents of a_src / b_src (< 2 GiB each): buffer-load range check
    int splits;
    long slab_stride;              // elements between split-K slabs (0: direct epilogue)
    long n_red_vox;                // WGRAD: reduction length in voxels
    int n_classes;                 // row classes (1: none): DGRAD stride^3 parity classes, or border classes
    int border;                    // 1: the classes are border classes (FWD, stride-1 DGRAD), table below
    int cls_tile_start[MAX_CLASSES + 1];
    BorderClass bcls[MAX_CLASSES];
    int ny_tiles;                  // column tiles; the grid is 1-D: (row tile, column tile, split) linearised
    int tiles_x;                   // row tiles
    int fold;                      // pair the two ends of the work list on a CU (class launches)
    int wbox;                      // WGRAD: walk only the voxel box that is valid for the tile's tap
    // Round 5: nbatch > 0 = that many problems of this one geometry in the launch, problem = blockIdx.y (weight gradients of a stage:
    // the launch's fixed costs once instead of per convolution); a_src / b_src / out are then taken from the tables
    int nbatch;
    const float* a_tab[MI_WGRAD_BATCH_MAX];
    const float* b_tab[MI_WGRAD_BATCH_MAX];
    float* out_tab[MI_WGRAD_BATCH_MAX];
};

// All gathers are raw buffer loads: 32-bit byte offset against a descriptor of the whole tensor.  An offset
// outside [0, bytes) returns zeros without touching memory, so zero padding, ragged tiles and the prefetch
// behind the last slice need no select on loaded data and no branch: an invalid element just gets the
// offset OOR (tensors are < 2 GiB, checked on the host, so OOR plus any in-tensor displacement stays outside).
constexpr unsigned OOR = 0x80000000u;
__device__ __forceinline__ float4 ld4(const float* ptr) { return *reinterpret_cast<const float4*>(ptr); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

// bit t of the result: 0 <= base + step*t < n, for t in [0, count)
__device__ __forceinline__ unsigned axis_mask(int base, int step, int count, int n) {
    unsigned m = 0;
    for (int t = 0; t < count; ++t)
        if ((unsigned)(base + step * t) < (unsigned)n) m |= 1u << t;
    return m;
}

// flat voxel index -> (n, z, y, x) of a (N, D, H, W) grid, 32-bit and branch-free: division by the
// (run-time) extents through precomputed multiply-shift constants, exact for every index < 2^31
struct GridDec {
    int D, H, W;
    unsigned mD, mH, mW;
    int sD, sH, sW;
    __device__ __forceinline__ static unsigned divm(unsigned n, unsigned m, int s) {
        return (unsigned)(((unsigned long long)n * m) >> s);
    }
    __device__ __forceinline__ void operator()(unsigned m, int& n, int& z, int& y, int& x) const {
        unsigned q = divm(m, mW, sW); x = (int)(m - q * (unsigned)W); m = q;
        q = divm(m, mH, sH); y = (int)(m - q * (unsigned)H); m = q;
        q = divm(m, mD, sD); z = (int)(m - q * (unsigned)D); n = (int)q;
    }
};
__host__ __device__ inline void magic31(int d, unsigned* m, int* s) {
    // M = floor(2^(31+L)/d) + 1 with L = ceil(log2 d): floor(n*M / 2^(31+L)) == n/d for 0 <= n < 2^31
    int L = 0;
    while ((1ll << L) < (long long)d) ++L;
    *s = 31 + L;
    *m = (unsigned)(((1ull << (31 + L)) / (unsigned long long)d) + 1ull);
}
__device__ __forceinline__ GridDec make_dec(int D, int H, int W) {      // prologue-only (DGRAD class grid)
    GridDec g;
    g.D = D > 0 ? D : 1; g.H = H > 0 ? H : 1; g.W = W > 0 ? W : 1;
    magic31(g.D, &g.mD, &g.sD); magic31(g.H, &g.mH, &g.sH); magic31(g.W, &g.mW, &g.sW);
    return g;
}

// (tap, channel-offset) cursor of the reduction, advanced one slice at a time without divisions
struct Cursor {
    int c0, ia, ib, ic;     // channel offset inside the tap, tap coordinates inside the class list
};

// STEM: Cin == 1 (the 7x7x7 stride-2 stem, moco_encoder_3d.py:163-169): the reduction index is
// the tap itself and each of a chunk's 4 taps is gathered separately through a tap LUT in LDS.
//
// BF3 = true: the same GEMM on the bf16 matrix pipe with f32-equivalent arithmetic.  Every f32 operand element is cut
// (exactly: a = a0 + a1 + a2, three bf16 values of 8 significant bits each) while it is staged into LDS, and the six
// products of weight <= 2 (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0) are accumulated in f32 by
// v_mfma_f32_32x32x16_bf16: each product is exact in f32, the dropped terms (a1 b2, a2 b1, a2 b2) are <= 2^-23 |a b|,
// i.e. of the size of ONE f32 rounding - the result differs from the f32 fmaf chain by rounding-order noise only
// (tests/test_train_gpu.py::test_conv_bf16x3_is_f32_equivalent measures both against float64; an Inf operand
// becomes NaN - Inf - Inf in the cut - where an f32 multiply would keep Inf).  Six bf16 MFMAs of K = 16 take 192 cycles against 512 for
// the eight f32 MFMAs they replace.  LDS holds three bf16 planes per operand:
//   "RowK" plane [row][BK] bf16, 16-byte chunks XOR-swizzled by row : fragments by ds_read_b128 (8 k's)
//   "KRow" plane, one [BK][32 columns] sub-tile per 32 columns (64-byte rows) : fragments by ds_read_b64_tr_b16
//          (the transposing LDS read of gfx950: 4 k's x 16 columns per 16-lane group, conflict-free on 64-byte rows)
template <int MODE, int BM, int BN, int BK, bool STEM, bool BF3 = false>
// (occupancy asked for: 3 workgroups per SIMD-quad for the bf16x3 tiles, 2 for the 128-row x 32-column inference tile, whose
// 8 accumulator blocks per wave do not fit the 168 registers of occupancy 3 - the compiler said so on every build)
__global__ __launch_bounds__(NTHREADS, BF3 ? ((BM == 128 && BN == 32) ? 2 : 3) : 1) void conv_igemm_kernel(ConvParams p) {
    static_assert(!BF3 || (!STEM && (BK == 16 || BK == 32)), "bf16x3 path: generic layers, 16- or 32-deep slices");
    static_assert(BK == 16 || BK == 32 || BK == 64, "slice depth");
    static_assert(BN == 32 || BN == 64 || BN == 128, "tile width");
    static_assert(BN != 32 || (BF3 && BM == 128 && BK == 32), "32-wide tiles: 128 rows, bf16x3, 32-deep slices");
    // four waves: 2 x 2 wave tiles; a 32-column tile (round 4: the 32-channel layers of the detector's U-Net, where half of a
    // 64-wide tile's MFMAs multiplied padding) stacks them 4 x 1
    constexpr int WN = BN == 32 ? 1 : 2, WM = 4 / WN;
    constexpr int LDK = BK + 4;
    constexpr int KH = BK / 2;                       // k's owned by one lane-half per slice
    constexpr int KC = BK / 4;                       // 16-B chunks along k per RowK row
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    static_assert(MT >= 1 && NT >= 1, "wave tile");
    constexpr bool A_ROWK = (MODE != MODE_WGRAD);
    constexpr bool B_ROWK = (MODE == MODE_DGRAD);
    constexpr int A_ELEMS = A_ROWK ? BM * LDK : BK * BM;
    constexpr int B_ELEMS = B_ROWK ? BN * LDK : BK * BN;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int A_CH = BM * KC / NTHREADS;         // 16-B chunks per thread per slice
    constexpr int B_CH = BN * KC / NTHREADS;
    constexpr int TPV = NTHREADS / BK;               // WGRAD: threads sharing one reduction voxel
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for 256 threads");
    // bf16x3 images (bytes)
    constexpr int RS = 2 * BK;                       // RowK plane: bytes per row; 16-byte chunk c of row r sits at chunk
                                                     // c ^ rowk_swz(r) (conflict-free ds_read_b128 without padding)
    constexpr int KSUB = 64 * BK + 64;               // KRow plane: bytes per 32-column sub-tile
    constexpr int A_PLANE = A_ROWK ? BM * RS : (BM / 32) * KSUB;
    constexpr int B_PLANE = B_ROWK ? BN * RS : (BN / 32) * KSUB;
    constexpr int STAGE_B = 3 * (A_PLANE + B_PLANE);
    constexpr int KS = BK / 16;                      // bf16 MFMA k-steps per slice

    auto rowk_swz = [](int r) { return BK == 32 ? (r >> 2) & 3 : (r >> 3) & 1; };
    __shared__ __attribute__((aligned(16))) float lds[BF3 ? STAGE_B / 2 : 2 * STAGE];
    unsigned char* const ldsb = reinterpret_cast<unsigned char*>(lds);
    __shared__ long rowmap[BM];                       // DGRAD classes: tile row -> output row
    __shared__ int2 taplut[STEM ? LUT_TAPS : 1];      // STEM: tap -> (voxel delta, a | b<<8 | c<<16)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l32 = lane & 31;
    // 1-D grid -> (row tile, column tile, split).  Class launches carry tiles of very different reduction lengths
    // and every workgroup is co-resident (two per CU: block L shares its CU with block L + 256), so consecutive
    // groups of 256 blocks walk the work list (heaviest class first) from opposite ends: heavy tiles pair with light.
    int blk_x, blk_y, blk_z;
    {
        const int total = gridDim.x, L = blockIdx.x;
        int q = L;
        if (p.fold) {                                      // (tools/cu_map_probe.hip: L and L + 256 do share a CU)
            const int g = L >> 8, r = L & 255, m = g >> 1;
            q = (g & 1) ? total - 1 - (m * 256 + r) : m * 256 + r;
        }
        if (p.fold) {                                      // work list = row tiles in class order, (column, split) minor
            const int per = p.ny_tiles * p.splits;
            blk_x = q / per;
            const int rem = q - blk_x * per;
            blk_y = rem % p.ny_tiles;
            blk_z = rem / p.ny_tiles;
        } else {                                           // row tiles fastest (neighbours share the weight tile)
            blk_x = q % p.tiles_x;
            const int rem = q / p.tiles_x;
            blk_y = rem % p.ny_tiles;
            blk_z = rem / p.ny_tiles;
        }
    }
    const int n0 = blk_y * BN;
    const int Kz = p.kd, Ky = p.kh, Kx = p.kw, S = p.stride, Pz = p.pd, Py = p.ph, Px = p.pw;
    const int taps = Kz * Ky * Kx;
    const int Lz = p.dd, Ly = p.dh, Lx = p.dw;        // dilation

    // ---- which rows does this workgroup own? ---------------------------------------------------
    int cls = 0;
    if (MODE != MODE_WGRAD && p.n_classes > 1) {
        while (cls + 1 < p.n_classes && blk_x >= p.cls_tile_start[cls + 1]) ++cls;
    }
    const long tile_in_cls = (long)blk_x - ((MODE != MODE_WGRAD) ? p.cls_tile_start[cls] : 0);
    const long m0 = tile_in_cls * BM;                 // first row (within the class for DGRAD)
    // DGRAD class geometry (stride 1: a single class with cz = cy = cx = 0)
    int cz = 0, cy = 0, cx = 0, zf = 0, yf = 0, xf = 0, Dz = p.Dr, Dy = p.Hr, Dx = p.Wr;
    int nz = Kz, ny = Ky, nx = Kx;                    // taps of the class per axis
    int ta0 = 0, tb0 = 0, tc0 = 0, tstep = 1;         // tap index of the class's i-th tap: t0 + tstep * i
    int pz0 = 0, py0 = 0, px0 = 0;                    // border class: first row position per axis
    const bool border = (MODE != MODE_WGRAD) && p.border;
    if (border) {
        const BorderClass bc = p.bcls[cls];
        pz0 = bc.pos0[0]; py0 = bc.pos0[1]; px0 = bc.pos0[2];
        Dz = bc.cnt[0]; Dy = bc.cnt[1]; Dx = bc.cnt[2];
        ta0 = bc.tlo[0]; tb0 = bc.tlo[1]; tc0 = bc.tlo[2];
        nz = bc.tcnt[0]; ny = bc.tcnt[1]; nx = bc.tcnt[2];
    } else if (MODE == MODE_DGRAD) {
        cz = cls / (S * S); cy = (cls / S) % S; cx = cls % S;
        zf = ((cz - Pz) % S + S) % S; yf = ((cy - Py) % S + S) % S; xf = ((cx - Px) % S + S) % S;
        Dz = zf < p.Dr ? (p.Dr - zf + S - 1) / S : 0;
        Dy = yf < p.Hr ? (p.Hr - yf + S - 1) / S : 0;
        Dx = xf < p.Wr ? (p.Wr - xf + S - 1) / S : 0;
        nz = cz < Kz ? (Kz - cz + S - 1) / S : 0;
        ny = cy < Ky ? (Ky - cy + S - 1) / S : 0;
        nx = cx < Kx ? (Kx - cx + S - 1) / S : 0;
        ta0 = cz; tb0 = cy; tc0 = cx; tstep = S;
    }
    const long M_here = (MODE == MODE_DGRAD || border) ? (long)p.N * Dz * Dy * Dx : p.M;

    // ---- reduction extent of this workgroup ----------------------------------------------------
    // WGRAD (not the stem): a row tile that lies inside ONE tap only needs the output voxels whose window position
    // for that tap is inside the input - a sub-box of the output grid.  The reduction walks that box (the rest would
    // multiply padding zeros), each thread decoding one voxel per slice for both operands.
    int wz0 = 0, wy0 = 0, wx0 = 0, wbz = p.Dr, wby = p.Hr, wbx = p.Wr;
    if (MODE == MODE_WGRAD && !STEM && p.wbox && (p.Ci % BM) == 0) {
        const int tap = (int)(m0 / p.Ci);
        const int a = tap / (Ky * Kx), b = (tap / Kx) % Ky, c = tap % Kx;
        auto range = [&](int t, int L_, int P_, int n_in, int n_out, int& lo, int& cnt) {
            // 0 <= o*S - P + t*L < n_in
            const int num = P_ - t * L_;
            int l = num > 0 ? (num + S - 1) / S : 0;
            int hsrc = n_in - 1 + P_ - t * L_;
            int hgh = hsrc < 0 ? -1 : hsrc / S;
            if (hgh > n_out - 1) hgh = n_out - 1;
            lo = l; cnt = hgh >= l ? hgh - l + 1 : 0;
        };
        range(a, Lz, Pz, p.Dg, p.Dr, wz0, wbz);
        range(b, Ly, Py, p.Hg, p.Hr, wy0, wby);
        range(c, Lx, Px, p.Wg, p.Wr, wx0, wbx);
    }
    const long w_red = (MODE == MODE_WGRAD && !STEM) ? (long)p.N * wbz * wby * wbx : p.n_red_vox;
    int nk;
    if (MODE == MODE_WGRAD) nk = (int)((w_red + BK - 1) / BK);
    else if (STEM) nk = (taps + BK - 1) / BK;
    else nk = nz * ny * nx * (((MODE == MODE_FWD) ? p.Ci : p.Co) / BK);
    const int nk_per_split = (nk + p.splits - 1) / p.splits;
    const int kt0 = min(blk_z * nk_per_split, nk);
    const int kt1 = min(kt0 + nk_per_split, nk);

    if (STEM) {
        for (int t = tid; t < LUT_TAPS; t += NTHREADS) {
            int a = t / (Ky * Kx), b = (t / Kx) % Ky, c = t % Kx;
            taplut[t] = (t < taps) ? make_int2((a * p.Hg + b) * p.Wg + c, a | (b << 8) | (c << 16))
                                   : make_int2(0, LUT_INVALID);
        }
    }

    const GridDec rdec = {p.Dr, p.Hr, p.Wr, p.mgD, p.mgH, p.mgW, p.shD, p.shH, p.shW};   // row grid
    const GridDec cdec = (MODE == MODE_WGRAD) ? make_dec(wbz, wby, wbx)       // WGRAD: the tile's voxel box
                                              : make_dec(Dz, Dy, Dx);          // class grid

    // ---- per-thread staging state --------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(p.nbatch ? p.a_tab[blockIdx.y] : p.a_src, p.a_bytes),
                                 b_rs = make_rsrc(p.nbatch ? p.b_tab[blockIdx.y] : p.b_src, p.b_bytes);
    unsigned a_off[A_CH];            // byte offset of the row base (+ chunk) in a_src, mod 2^32 (padding rows
                                     // start "before" the tensor); WGRAD: tap + ci offset
    unsigned a_msk[A_CH];            // RowK: per-axis validity bits (z | y<<8 | x<<16), 0 = row off
                                     // WGRAD: the chunk's tap coordinates a | b<<8 | c<<16
    bool w_ok[A_CH];
    int a_lds[A_CH];
    const int w_kk = tid / TPV;

    if (A_ROWK) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int q = tid + i * NTHREADS;
            const int row = q / KC, c = q % KC;
            const long m = m0 + row;
            a_lds[i] = BF3 ? row * RS + 16 * ((c >> 1) ^ rowk_swz(row)) + 8 * (c & 1) : row * LDK + 4 * c;
            a_msk[i] = 0;
            a_off[i] = 0;
            w_ok[i] = false;
            if (m < M_here) {
                if (MODE == MODE_FWD) {
                    int n, z, y, x;
                    if (border) {
                        cdec((unsigned)m, n, z, y, x);
                        z += pz0; y += py0; x += px0;
                        if (c == 0) rowmap[row] = (((long)n * p.Dr + z) * p.Hr + y) * p.Wr + x;
                    } else {
                        rdec((unsigned)m, n, z, y, x);
                    }
                    // (border classes: the window starts at the class's first valid tap)
                    const int zb = z * S - Pz + ta0 * Lz, yb = y * S - Py + tb0 * Ly, xb = x * S - Px + tc0 * Lx;
                    a_msk[i] = axis_mask(zb, Lz, nz, p.Dg) | (axis_mask(yb, Ly, ny, p.Hg) << 8) |
                               (axis_mask(xb, Lx, nx, p.Wg) << 16);
                    a_off[i] = 4u * (unsigned)(((((long)n * p.Dg + zb) * p.Hg + yb) * p.Wg + xb) * p.Cg + (STEM ? 0 : 4 * c));
                } else if (border) {
                    // stride-1 DGRAD: input voxel z receives dy[z + P - a*L] * w[a] for the class's taps a = ta0 + i
                    int n, z, y, x;
                    cdec((unsigned)m, n, z, y, x);
                    z += pz0; y += py0; x += px0;
                    const int zb = z + Pz - ta0 * Lz, yb = y + Py - tb0 * Ly, xb = x + Px - tc0 * Lx;
                    a_msk[i] = axis_mask(zb, -Lz, nz, p.Dg) | (axis_mask(yb, -Ly, ny, p.Hg) << 8) |
                               (axis_mask(xb, -Lx, nx, p.Wg) << 16);
                    a_off[i] = 4u * (unsigned)(((((long)n * p.Dg + zb) * p.Hg + yb) * p.Wg + xb) * p.Cg + 4 * c);
                    if (c == 0) rowmap[row] = (((long)n * p.Dr + z) * p.Hr + y) * p.Wr + x;
                } else {
                    int n, jz, jy, jx;
                    cdec((unsigned)m, n, jz, jy, jx);
                    const int z = zf + S * jz, y = yf + S * jy, x = xf + S * jx;
                    const int zb = (z + Pz - cz) / S, yb = (y + Py - cy) / S, xb = (x + Px - cx) / S;
                    a_msk[i] = axis_mask(zb, -Lz, nz, p.Dg) | (axis_mask(yb, -Ly, ny, p.Hg) << 8) |
                               (axis_mask(xb, -Lx, nx, p.Wg) << 16);
                    a_off[i] = 4u * (unsigned)(((((long)n * p.Dg + zb) * p.Hg + yb) * p.Wg + xb) * p.Cg + 4 * c);
                    if (c == 0 && p.n_classes > 1) rowmap[row] = (((long)n * p.Dr + z) * p.Hr + y) * p.Wr + x;
                }
            } else if (MODE != MODE_WGRAD && c == 0 && p.n_classes > 1) {
                rowmap[row] = -1;
            }
        }
    } else {
        // WGRAD: GEMM rows are (tap, ci).  TPV threads share one reduction voxel kk (one voxel
        // decode per thread and slice); chunk i of a thread covers rows 4*(tid%TPV + TPV*i) ..+3,
        // which share a tap because Ci % 4 == 0.
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int j = (tid % TPV) + TPV * i;
            a_lds[i] = BF3 ? ((4 * j) >> 5) * KSUB + w_kk * 64 + ((4 * j) & 31) * 2 : w_kk * BM + 4 * j;
            const long row = m0 + 4 * j;
            w_ok[i] = row < p.M;
            a_msk[i] = 0;
            a_off[i] = 0;
            if (w_ok[i] && !STEM) {
                const int tap = (int)(row / p.Ci);
                const int ci = (int)(row % p.Ci);
                const int a = tap / (Ky * Kx), b = (tap / Kx) % Ky, c = tap % Kx;
                a_off[i] = 4u * (unsigned)(((long)(a * Lz * p.Hg + b * Ly) * p.Wg + c * Lx) * p.Cg + ci);
                a_msk[i] = (unsigned)((a * Lz) | ((b * Ly) << 8) | ((c * Lx) << 16));   // dilated tap coordinates
            }
        }
    }
    // B operand: per-thread byte offset (OOR when the column is out of range) plus a wave-uniform
    // offset per slice
    int b_lds[B_CH], b_row[B_CH], b_col[B_CH];
    unsigned b_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int q = tid + i * NTHREADS;
        if (B_ROWK) {            // DGRAD: LDS [n][k]; global W[tap][ci = n][co = k]
            b_row[i] = q / KC; b_col[i] = (q % KC) * 4;
            b_lds[i] = BF3 ? b_row[i] * RS + 16 * ((b_col[i] >> 3) ^ rowk_swz(b_row[i])) + 2 * (b_col[i] & 4)
                           : b_row[i] * LDK + b_col[i];
            const bool ok = n0 + b_row[i] < p.Ncols;
            b_off[i] = ok ? 4u * (unsigned)((long)(n0 + b_row[i]) * p.Co + b_col[i]) : OOR;
        } else if (MODE == MODE_WGRAD && !STEM) {
            // dY row of the voxel this thread decodes (w_kk), chunk (tid % TPV) + TPV*i of its BN columns
            b_row[i] = w_kk; b_col[i] = 4 * ((tid % TPV) + TPV * i);
            b_lds[i] = BF3 ? (b_col[i] >> 5) * KSUB + b_row[i] * 64 + (b_col[i] & 31) * 2 : b_row[i] * BN + b_col[i];
            const bool ok = n0 + b_col[i] < p.Ncols;
            b_off[i] = ok ? 4u * (unsigned)(n0 + b_col[i]) : OOR;
        } else {                 // LDS [k][n]; global rows k, cols n contiguous
            b_row[i] = q / (BN / 4); b_col[i] = (q % (BN / 4)) * 4;
            b_lds[i] = BF3 ? (b_col[i] >> 5) * KSUB + b_row[i] * 64 + (b_col[i] & 31) * 2 : b_row[i] * BN + b_col[i];
            const bool ok = n0 + b_col[i] < p.Ncols;
            b_off[i] = ok ? 4u * (unsigned)((long)b_row[i] * p.Co + n0 + b_col[i]) : OOR;
        }
    }

    // reduction cursor at kt0 (divisions here, none in the loop)
    Cursor cur = {0, 0, 0, 0};
    if (MODE != MODE_WGRAD && !STEM && nk > 0) {
        const int per_tap = ((MODE == MODE_FWD) ? p.Ci : p.Co) / BK;
        int t = kt0 / per_tap;
        cur.c0 = (kt0 - t * per_tap) * BK;
        cur.ic = t % nx; t /= nx; cur.ib = t % ny; cur.ia = t / ny;
    }
    if (STEM) __syncthreads();          // tap LUT visible

    // two staging register sets (statically indexed): while set P is being filled with slice s+2,
    // set 1-P (slice s+1, loaded one whole iteration earlier) goes to LDS - a prefetch distance of
    // two slices, enough to cover the loaded global-memory latency (a distance of one was not)
    float4 a_reg[2][A_CH], b_reg[2][B_CH];

    // ---- gather of one slice, cut into NPARTS pieces so that the main loop can issue one piece in
    // the shadow of each MFMA (64 cycles): part 0 = wave-uniform preparation, parts 1..A_CH = the A
    // chunks, then the B chunks, last = cursor advance.  live == false (wave-uniform): the prefetch
    // behind the last slice - every load then reads the zero block.
    constexpr int NPARTS = 2 + A_CH + B_CH;
    unsigned pt_aoff = 0, pt_boff = 0, pt_voff = 0;       // byte offsets (mod 2^32)
    int pt_zb = 0, pt_yb = 0, pt_xb = 0;
    bool pt_vok = false;
    // (SETc / CURc are std::integral_constant: the staging sets must be indexed by compile-time
    // constants or the register arrays spill to scratch)
    auto load_part = [&](int kt, bool live, int part, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        if (part == 0) {
            if (MODE == MODE_WGRAD) {
                // reduction index = output voxel mv = kt*BK + kk (one decode per thread); a voxel behind
                // the last one decodes to n >= N: its offsets fall behind the tensors (zeros)
                const long mv = (long)kt * BK + w_kk;
                pt_vok = live & (mv < w_red);
                int n, z, y, x;
                if (STEM) {
                    rdec(pt_vok ? (unsigned)mv : 0u, n, z, y, x);
                } else {
                    cdec(pt_vok ? (unsigned)mv : 0u, n, z, y, x);
                    z += wz0; y += wy0; x += wx0;
                    pt_boff = pt_vok ? 4u * (unsigned)(((((long)n * p.Dr + z) * p.Hr + y) * p.Wr + x) * p.Co) : OOR;
                }
                pt_zb = z * S - Pz; pt_yb = y * S - Py; pt_xb = x * S - Px;
                pt_voff = 4u * (unsigned)(((((long)n * p.Dg + pt_zb) * p.Hg + pt_yb) * p.Wg + pt_xb) * p.Cg);
            } else if (!STEM) {
                const int ia = cur.ia, ib = cur.ib, ic = cur.ic, c0 = cur.c0;
                int wtap;
                if (MODE == MODE_FWD) {
                    pt_aoff = 4u * (unsigned)(((long)(ia * Lz * p.Hg + ib * Ly) * p.Wg + ic * Lx) * p.Cg + c0);
                    wtap = ((ta0 + ia) * Ky + (tb0 + ib)) * Kx + (tc0 + ic);
                } else {
                    pt_aoff = 4u * (unsigned)(-((long)(ia * Lz * p.Hg + ib * Ly) * p.Wg + ic * Lx) * p.Cg + c0);
                    wtap = ((ta0 + tstep * ia) * Ky + (tb0 + tstep * ib)) * Kx + (tc0 + tstep * ic);
                }
                // weights: FWD rows (wtap*Ci + c0 + k) of [.][Co]; DGRAD row (wtap*Ci + ci), cols c0 + k
                const long bo = (MODE == MODE_FWD) ? ((long)wtap * p.Ci + c0) * p.Co : (long)wtap * p.Ci * p.Co + c0;
                pt_boff = live ? 4u * (unsigned)bo : OOR;
            }
        } else if (part <= A_CH) {
            const int i = part - 1;
            if (MODE == MODE_WGRAD) {
                if (STEM) {
                    float e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int tap = (int)m0 + 4 * ((tid % TPV) + TPV * i) + u;
                        const int2 tl = taplut[min(tap, LUT_TAPS - 1)];
                        const int a = tl.y & 0xff, b = (tl.y >> 8) & 0xff, c = (tl.y >> 16) & 0xff;
                        const bool ok = pt_vok & (tap < taps) & ((unsigned)(pt_zb + a) < (unsigned)p.Dg) &
                                        ((unsigned)(pt_yb + b) < (unsigned)p.Hg) & ((unsigned)(pt_xb + c) < (unsigned)p.Wg);
                        e[u] = bld1(a_rs, ok ? pt_voff + 4u * (unsigned)tl.x : OOR);
                    }
                    a_reg[SET][i] = make_float4(e[0], e[1], e[2], e[3]);
                } else {
                    const int a = a_msk[i] & 0xff, b = (a_msk[i] >> 8) & 0xff, c = (a_msk[i] >> 16) & 0xff;
                    const bool ok = pt_vok & w_ok[i] & ((unsigned)(pt_zb + a) < (unsigned)p.Dg) &
                                    ((unsigned)(pt_yb + b) < (unsigned)p.Hg) & ((unsigned)(pt_xb + c) < (unsigned)p.Wg);
                    a_reg[SET][i] = bld4(a_rs, ok ? a_off[i] + pt_voff : OOR);
                }
            } else if (STEM) {
                // FWD stem: slice kt covers taps kt*BK .. kt*BK+BK-1
                const int q = tid + i * NTHREADS;
                const unsigned m = a_msk[i];
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tap = kt * BK + 4 * (q % KC) + u;
                    const int2 tl = taplut[min(tap, LUT_TAPS - 1)];
                    const bool ok = live & (((m >> (tl.y & 0xff)) & (m >> (8 + ((tl.y >> 8) & 0xff))) &
                                             (m >> (16 + ((tl.y >> 16) & 0xff))) & 1u) != 0);
                    e[u] = bld1(a_rs, ok ? a_off[i] + 4u * (unsigned)tl.x : OOR);
                }
                a_reg[SET][i] = make_float4(e[0], e[1], e[2], e[3]);
            } else {
                const unsigned m = a_msk[i];
                const bool ok = live & (((m >> cur.ia) & (m >> (8 + cur.ib)) & (m >> (16 + cur.ic)) & 1u) != 0);
                a_reg[SET][i] = bld4(a_rs, ok ? a_off[i] + pt_aoff : OOR);
            }
        } else if (part <= A_CH + B_CH) {
            const int i = part - 1 - A_CH;
            if (MODE == MODE_WGRAD && !STEM) {
                // the dY row of the voxel decoded in part 0 (OOR behind the box / the last slice)
                b_reg[SET][i] = bld4(b_rs, b_off[i] + pt_boff);
            } else if (MODE == MODE_WGRAD || STEM) {
                // rows kt*BK + b_row of dY (stem WGRAD) / of the stem weights: a row behind the last one is
                // behind the tensor (zeros)
                const unsigned so = live ? 4u * (unsigned)((long)kt * BK * p.Co) : OOR;
                b_reg[SET][i] = bld4(b_rs, b_off[i] + so);
            } else {
                // (an out-of-range column has b_off == OOR: the sum stays out of range; both OOR only
                // happens behind the last slice, whose tile is never consumed)
                b_reg[SET][i] = bld4(b_rs, b_off[i] + pt_boff);
            }
        } else if (MODE != MODE_WGRAD && !STEM) {
            // advance the (tap, channel) cursor, branch-free
            const int cred = (MODE == MODE_FWD) ? p.Ci : p.Co;
            const int c0n = cur.c0 + BK;
            const int w0 = c0n >= cred ? 1 : 0;
            cur.c0 = w0 ? 0 : c0n;
            const int icn = cur.ic + w0;
            const int w1 = icn >= nx ? 1 : 0;
            cur.ic = w1 ? 0 : icn;
            const int ibn = cur.ib + w1;
            const int w2 = ibn >= ny ? 1 : 0;
            cur.ib = w2 ? 0 : ibn;
            cur.ia += w2;
        }
    };
    auto load_tile = [&](int kt, bool live, auto SETc) {
#pragma unroll
        for (int part = 0; part < NPARTS; ++part) load_part(kt, live, part, SETc);
    };
    // bf16x3: exact three-way cut of 4 consecutive f32 (one 16-byte chunk) into 3 x 4 bf16, one 8-byte store per plane.
    // Truncation keeps every step exact: a0 = top 16 bits of a, r1 = a - a0 (<= 16 significant bits), a1 = top 16 bits
    // of r1, a2 = r1 - a1 (<= 8 significant bits: a bf16 value).
    auto split_cut = [&](const float4 v, uint2 (&o)[3]) {
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned u0[4], u1[4], u2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u0[t] = __float_as_uint(e[t]);
            const float r1 = e[t] - __uint_as_float(u0[t] & 0xffff0000u);
            u1[t] = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(u1[t] & 0xffff0000u);
            u2[t] = __float_as_uint(r2);
        }
        // v_perm_b32: the high halves of two dwords -> one dword of two bf16 (element t in the low half)
        constexpr unsigned HI2 = 0x07060302u;
        o[0] = make_uint2(__builtin_amdgcn_perm(u0[1], u0[0], HI2), __builtin_amdgcn_perm(u0[3], u0[2], HI2));
        o[1] = make_uint2(__builtin_amdgcn_perm(u1[1], u1[0], HI2), __builtin_amdgcn_perm(u1[3], u1[2], HI2));
        o[2] = make_uint2(__builtin_amdgcn_perm(u2[1], u2[0], HI2), __builtin_amdgcn_perm(u2[3], u2[2], HI2));
    };
    auto split_store = [&](unsigned char* dst, int plane_bytes, const float4 v) {
        uint2 o[3];
        split_cut(v, o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + pl * plane_bytes) = o[pl];
    };
    auto store_tile = [&](int buf, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        if constexpr (BF3) {
            unsigned char* const Ab = ldsb + buf * STAGE_B;
            unsigned char* const Bb = Ab + 3 * A_PLANE;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) split_store(Ab + a_lds[i], A_PLANE, a_reg[SET][i]);
#pragma unroll
            for (int i = 0; i < B_CH; ++i) split_store(Bb + b_lds[i], B_PLANE, b_reg[SET][i]);
        } else {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) *reinterpret_cast<float4*>(lds + buf * STAGE + a_lds[i]) = a_reg[SET][i];
#pragma unroll
            for (int i = 0; i < B_CH; ++i) *reinterpret_cast<float4*>(lds + buf * STAGE + A_ELEMS + b_lds[i]) = b_reg[SET][i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software pipeline over the reduction slices (prefetch distance 2) ----------------------
    //   LDS buffer (s & 1)      : slice s, read into fragment registers at the top of iteration s
    //   staging set (s+1) & 1   : slice s+1, loaded during iteration s-1, stored to LDS at the end of s
    //   staging set s & 1       : slice s+2, its global loads are issued during iteration s
    // One iteration, parametrised by the (compile-time) parity of the staging sets:
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    auto iteration = [&](int kt, int buf, auto CURc) {
        constexpr int CUR = decltype(CURc)::value;
        using SetCur = std::integral_constant<int, CUR>;
        using SetOther = std::integral_constant<int, CUR ^ 1>;
        const bool more2 = (kt + 2 < kt1);
        if constexpr (BF3) {
            // Phase 1: per operand 3 planes x KS k-steps of 8 bf16 (k = 16 s + 8 h + e for both operands)
            const unsigned char* Ab = ldsb + buf * STAGE_B;
            const unsigned char* Bb = Ab + 3 * A_PLANE;
            const int i16 = lane & 15, g16 = (lane >> 4) & 1;
            // transposing read: lane 4q+pp of a 16-lane group addresses row k0+q, columns 4pp..4pp+3 of the group's
            // 4 x 16 block and receives column (lane & 15), rows k0 .. k0+3
            const int tr_off = (i16 >> 2) * 64 + (16 * g16 + 4 * (i16 & 3)) * 2;
            bf16x8 af[MT][3][KS], bf[NT][3][KS];
            auto frag = [&](const unsigned char* plane0, int plane_bytes, bool rowk, int r0, bf16x8 (&f)[3][KS]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2) {
                        if (rowk) {
                            f[pl][s2] = *reinterpret_cast<const bf16x8*>(plane0 + pl * plane_bytes + (r0 + l32) * RS +
                                                                         16 * ((2 * s2 + h) ^ rowk_swz(r0 + l32)));
                        } else {
                            const unsigned char* sb = plane0 + pl * plane_bytes + (r0 >> 5) * KSUB + (16 * s2 + 8 * h) * 64 + tr_off;
                            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb));
                            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + 4 * 64));
                            f[pl][s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    }
            };
#pragma unroll
            for (int i = 0; i < MT; ++i) frag(Ab, A_PLANE, A_ROWK, wm * WTM + i * 32, af[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) frag(Bb, B_PLANE, B_ROWK, wn * WTN + j * 32, bf[j]);
            // ... and in the shadow of those LDS reads the cut of slice kt+1 (set CUR^1, landed during the previous
            // iteration) into its bf16 planes: VALU only
            constexpr int NCH = A_CH + B_CH;
            uint2 cv[NCH][3];
#pragma unroll
            for (int i = 0; i < A_CH; ++i) split_cut(a_reg[CUR ^ 1][i], cv[i]);
#pragma unroll
            for (int i = 0; i < B_CH; ++i) split_cut(b_reg[CUR ^ 1][i], cv[A_CH + i]);
            unsigned char* const An = ldsb + (buf ^ 1) * STAGE_B;
            unsigned char* const Bn = An + 3 * A_PLANE;
            __builtin_amdgcn_sched_barrier(0);
            // Phase 2: six products per k-step, smallest first; the gather pieces of slice kt+2 and the LDS stores of
            // slice kt+1 (other buffer: nobody reads it before the barrier) ride behind the MFMAs
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            constexpr int NMF3 = 6 * KS * MT * NT;
            static_assert(NPARTS <= NMF3, "more gather pieces than MFMAs");
#pragma unroll
            for (int g = 0; g < NMF3; ++g) {
                const int ij = g % (MT * NT), pr = (g / (MT * NT)) % 6, s2 = g / (MT * NT * 6);
                const int i = ij / NT, j = ij % NT;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[pr]][s2], bf[j][PB[pr]][s2], acc[i][j], 0, 0, 0);
                if (g < NPARTS) load_part(kt + 2, more2, g, SetCur{});
                // the 3 * NCH plane stores, spread over the MFMAs
#pragma unroll
                for (int w = g * 3 * NCH / NMF3; w < (g + 1) * 3 * NCH / NMF3; ++w) {
                    const int ch = w / 3, pl = w % 3;
                    unsigned char* dst = ch < A_CH ? An + a_lds[ch < A_CH ? ch : 0] + pl * A_PLANE
                                                   : Bn + b_lds[ch < A_CH ? 0 : ch - A_CH] + pl * B_PLANE;
                    *reinterpret_cast<uint2*>(dst) = cv[ch][pl];
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            return;
        }
        // Phase 1: fragments of the current slice from LDS
        const float* Ab = lds + buf * STAGE;
        const float* Bb = Ab + A_ELEMS;
        float af[MT][KH], bf[NT][KH];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * WTM + i * 32 + l32;
            if (A_ROWK) {
#pragma unroll
                for (int u = 0; u < KH / 4; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(Ab + r * LDK + h * KH + 4 * u);
                    af[i][4 * u] = v.x; af[i][4 * u + 1] = v.y; af[i][4 * u + 2] = v.z; af[i][4 * u + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KH; ++t) af[i][t] = Ab[(h * KH + t) * BM + r];
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = wn * WTN + j * 32 + l32;
            if (B_ROWK) {
#pragma unroll
                for (int u = 0; u < KH / 4; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(Bb + c * LDK + h * KH + 4 * u);
                    bf[j][4 * u] = v.x; bf[j][4 * u + 1] = v.y; bf[j][4 * u + 2] = v.z; bf[j][4 * u + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KH; ++t) bf[j][t] = Bb[(h * KH + t) * BN + c];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // Phase 2: the MFMAs of slice kt; behind each of the first NPARTS of them one piece of the
        // gather of slice kt+2 (address arithmetic + one global load into set CUR^1... see above) is
        // pinned, so it issues while that MFMA occupies the matrix pipe (64 cycles).  The gather is
        // unconditional (zeros behind the last slice): the body has no branch, and out-of-range
        // elements are LOADED from a zero block, so nothing depends on the loaded data until the
        // LDS stores of the NEXT iteration.
        constexpr int NMF = KH * MT * NT;
        static_assert(NPARTS <= NMF, "more gather pieces than MFMAs");
#pragma unroll
        for (int g = 0; g < NMF; ++g) {
            const int t = g / (MT * NT), i = (g / NT) % MT, j = g % NT;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
            if (g < NPARTS) {
                load_part(kt + 2, more2, g, SetCur{});
                // compiler-level fence (no instruction): keeps this piece's global load from being
                // sunk towards its use; the sched_barrier pins the machine schedule
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // Phase 3: slice kt+1 (set CUR^1, in flight since the previous iteration) goes to the other
        // LDS buffer; the wait only covers those older loads (counted vmcnt)
        store_tile(buf ^ 1, SetOther{});
        __syncthreads();
    };

    if (kt0 < kt1) {
        load_tile(kt0, true, Set0{});
        store_tile(0, Set0{});
        load_tile(kt0 + 1, kt0 + 1 < kt1, Set1{});       // slice kt0+1 -> set 1, stored at the end of iteration kt0
    }
    __syncthreads();

    // pairs in the loop, the odd slice after it: a skip of the second half inside the loop would make the
    // compiler wait at the loop head for gathers that are only in flight on that (exiting) path
    int kt = kt0;
    for (; kt + 1 < kt1; kt += 2) {
        iteration(kt, 0, Set0{});                          // loads slice kt+2 into set 0, stores set 1
        iteration(kt + 1, 1, Set1{});                      // loads slice kt+3 into set 1, stores set 0
    }
    if (kt < kt1) iteration(kt, 0, Set0{});

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* outp = (p.nbatch ? p.out_tab[blockIdx.y] : p.out) + (long)blk_z * p.slab_stride;
    const bool direct = (p.slab_stride == 0);
    const bool mapped = (MODE != MODE_WGRAD) && p.n_classes > 1;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trow = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            long row = m0 + trow;
            bool ok = row < M_here && col < p.Ncols;
            if (mapped) { row = rowmap[trow]; ok = ok && row >= 0; }
            if (ok) {
                const long o = row * p.Ncols + col;
                float v = acc[i][j][r];
                if (direct) {
                    if (p.res) v += p.res[p.res_bcast ? (long)col : o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mask) v = (p.mask[o] > 0.f) ? v : 0.f;
                }
                outp[o] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int n_slabs,
                                                           long slab_stride, float* out,
                                                           const float* res, const float* mask,
                                                           int relu, long n4, int res_mod = 0) {
    // res_mod > 0: res is one row of res_mod values (a bias), broadcast over the rows
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 s = ld4(slabs + 4 * i);
        for (int z = 1; z < n_slabs; ++z) {
            float4 v = ld4(slabs + (long)z * slab_stride + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (res) { float4 v = ld4(res + (res_mod ? (4 * i) % res_mod : 4 * i)); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        if (mask) {
            float4 v = ld4(mask + 4 * i);
            s.x = v.x > 0.f ? s.x : 0.f; s.y = v.y > 0.f ? s.y : 0.f;
            s.z = v.z > 0.f ? s.z : 0.f; s.w = v.w > 0.f ? s.w : 0.f;
        }
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
}

// the split-K slabs of up to MI_REDUCE_BATCH launches summed in ONE launch (blockIdx.y = launch): the weight gradients of
// a whole backward pass are reduced together instead of one tiny launch behind every wgrad
constexpr int MI_REDUCE_BATCH = 24;
struct SplitDesc { const float* slabs; float* out; long slab_stride; long n4; int n_slabs; int pad; };
struct SplitBatch { SplitDesc d[MI_REDUCE_BATCH]; };
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(SplitBatch bt) {
    const SplitDesc& d = bt.d[blockIdx.y];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.n4; i += (long)gridDim.x * 256) {
        float4 s = ld4(d.slabs + 4 * i);
        for (int z = 1; z < d.n_slabs; ++z) {
            float4 v = ld4(d.slabs + (long)z * d.slab_stride + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(d.out + 4 * i) = s;
    }
}

struct Plan { int bm, bn, bk, splits; long tiles_x; };

int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

// Arithmetic of the generic kernel: "bf16x3" (default; f32-equivalent three-way bf16 cut on the bf16 matrix pipe, see
// the kernel's header) or "f32" (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain) - MI_CONV_ARITH selects.
bool conv_arith_bf16x3() {
    const char* v = getenv("MI_CONV_ARITH");
    return !(v && v[0] == 'f');
}

// red_ch: reduction channels (FWD: Ci, DGRAD: Co; 0 for WGRAD / stem where any depth works)
// red_len: longest reduction in elements; tiles_x_of(bm): row tiles for a given BM
// Tile choice measured on MI355X (tools/bench_conv.py, profiles/r01_conv_tuning.txt): with the
// prefetch-distance-2 pipeline 64-row tiles (two workgroups per CU) win or tie on every layer and
// mode; 128-row tiles stay available through MI_CONV_BM for tuning.
template <class F>
Plan make_plan(int mode, bool stem, long M, int Ncols, long red_len, int red_ch, F tiles_x_of) {
    Plan pl;
    (void)mode; (void)stem;
    pl.bm = 64;
    pl.bk = (red_ch == 0 || red_ch % 32 == 0) ? 32 : 16;
    pl.bn = 64;                     // 128-wide tiles measured no faster on any layer (profiles/r01_conv_tuning.txt)
    if (int v = env_int("MI_CONV_BM")) pl.bm = (v == 128) ? 128 : 64;            // tuning overrides
    if (int v = env_int("MI_CONV_BN")) pl.bn = (v == 128 && Ncols % 128 == 0) ? 128 : 64;
    if (int v = env_int("MI_CONV_BK")) if ((v == 16 || v == 32 || v == 64) && (red_ch == 0 || red_ch % v == 0)) pl.bk = v;
    if (pl.bk != 32 || stem) pl.bn = 64;
    if (pl.bk == 64 && (stem || pl.bm != 128)) pl.bk = 32;     // 64-deep slices: 128x64 tiles only
    // forward convolutions with at most 32 output channels (the detector's 32-channel layers and head): 128 x 32 tiles, where
    // there are rows for two workgroups per CU (MI_CONV_NARROW=0: the 64 x 64 tile with half its columns empty)
    {
        const char* nv = getenv("MI_CONV_NARROW");
        if (mode == MODE_FWD && !stem && Ncols <= 32 && pl.bk == 32 && pl.bm == 64 && pl.bn == 64 && conv_arith_bf16x3() &&
            !(nv && atoi(nv) == 0) && tiles_x_of(128) >= 512) { pl.bm = 128; pl.bn = 32; }
    }
    pl.tiles_x = tiles_x_of(pl.bm);
    const long tiles = pl.tiles_x * ((Ncols + pl.bn - 1) / pl.bn);
    const long nk = (red_len + pl.bk - 1) / pl.bk;
    // Split count from a load model of the 256 CUs. Workgroups of one launch are all co-resident (<= 5 per CU
    // by LDS), so the launch ends with the most loaded CU: time ~ ceil(blocks/256) * (slices per block +
    // prologue/epilogue), a lone workgroup per CU running at ~0.7 of the paired rate. 513 blocks cost 3/2 of 512
    // (measured: l1 wgrad with 19 splits 122 us, see profiles/r01_conv_tuning.txt).
    int splits = 1;
    {
        const long min_slices = 128 / pl.bk;             // at least 128 reduction elements per split
        long max_splits = nk / min_slices > 0 ? nk / min_slices : 1;
        if (max_splits > 128) max_splits = 128;
        double best = 1e30;
        for (long sp = 1; sp <= max_splits; ++sp) {
            const long per = (nk + sp - 1) / sp;
            if ((sp - 1) * per >= nk) continue;           // an empty last split: same schedule as a smaller sp
            const long blocks = tiles * sp;
            const long per_cu = (blocks + 255) / 256;
            const double rate = per_cu == 1 ? 1.4 : (double)per_cu;
            double cost = rate * (double)(per + 6);
            if (sp > 1) cost += 2.0 + 0.05 * (double)sp;  // slab write + reduce pass
            if (cost < best - 1e-9) { best = cost; splits = (int)sp; }
        }
    }
    if (int v = env_int("MI_CONV_SPLITS")) splits = v;
    pl.splits = splits < 1 ? 1 : splits;
    return pl;
}

template <int MODE, bool STEM>
int launch_mode(const ConvParams& p, const Plan& pl, hipStream_t s) {
    dim3 grid((unsigned)(pl.tiles_x * ((p.Ncols + pl.bn - 1) / pl.bn) * pl.splits), (unsigned)std::max(1, p.nbatch));    // x: decoded in the kernel
    if constexpr (!STEM) {
        // (a 128 x 64 bf16x3 tile was built and measured in round 2: layer-1 forward 100.6 us against 52.8 us, layer-2
        // 43.4 against 33.4, layer-3 25.0 against 23.7 - a batch-64 step does not have the rows for it: 32,768 im2col
        // rows are 256 tiles of 128, one per CU)
        if constexpr (MODE == MODE_FWD) {
            if (conv_arith_bf16x3() && pl.bm == 128 && pl.bn == 32 && pl.bk == 32) {
                hipLaunchKernelGGL((conv_igemm_kernel<MODE, 128, 32, 32, false, true>), grid, dim3(NTHREADS), 0, s, p);
                MI_RETURN_IF_LAUNCH_FAILED();
                return MI_OK;
            }
        }
        if (conv_arith_bf16x3() && pl.bm == 64 && pl.bn == 64 && (pl.bk == 32 || pl.bk == 16)) {
            if (pl.bk == 32) hipLaunchKernelGGL((conv_igemm_kernel<MODE, 64, 64, 32, false, true>), grid, dim3(NTHREADS), 0, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<MODE, 64, 64, 16, false, true>), grid, dim3(NTHREADS), 0, s, p);
            MI_RETURN_IF_LAUNCH_FAILED();
            return MI_OK;
        }
    }
#define MI_LAUNCH(BM_, BN_, BK_) \
    hipLaunchKernelGGL((conv_igemm_kernel<MODE, BM_, BN_, BK_, STEM>), grid, dim3(NTHREADS), 0, s, p)
    if (pl.bn == 128 && !STEM) {           // 128-wide tiles only with 32-deep slices
        if (pl.bm == 128) MI_LAUNCH(128, 128, 32); else MI_LAUNCH(64, 128, 32);
    } else if (pl.bm == 128 && pl.bk == 64 && !STEM) MI_LAUNCH(128, 64, 64);
    else if (pl.bm == 128 && pl.bk == 32) MI_LAUNCH(128, 64, 32);
    else if (pl.bm == 128) MI_LAUNCH(128, 64, 16);
    else if (pl.bk == 32) MI_LAUNCH(64, 64, 32);
    else MI_LAUNCH(64, 64, 16);
#undef MI_LAUNCH
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

struct Geom {
    int N, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, kd, kh, kw, stride, pd, ph, pw;
    int dd = 1, dh = 1, dw = 1;
};

bool geom_ok(const Geom& g) {
    if (g.N <= 0 || g.Di <= 0 || g.Hi <= 0 || g.Wi <= 0 || g.Ci <= 0 || g.Co <= 0) return false;
    if (g.kd <= 0 || g.kd > 7 || g.kh <= 0 || g.kh > 7 || g.kw <= 0 || g.kw > 7) return false;
    if (g.stride <= 0 || g.stride > 2 || g.pd < 0 || g.ph < 0 || g.pw < 0) return false;
    if ((g.Ci % 16 && g.Ci != 1) || g.Co % 16) return false;
    if ((long)g.N * g.Di * g.Hi * g.Wi >= (1l << 31)) return false;     // 32-bit voxel indices
    if (g.dd < 1 || g.dh < 1 || g.dw < 1 || g.dd > 36 || g.dh > 36 || g.dw > 36) return false;
    if ((g.dd > 1 || g.dh > 1 || g.dw > 1) && (g.stride != 1 || g.Ci == 1)) return false;   // dilation: stride 1, no stem
    return true;
}
Geom make_geom_nd(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride,
                  int pd, int ph, int pw, int dd = 1, int dh = 1, int dw = 1) {
    Geom g{N, Di, Hi, Wi, Ci, 0, 0, 0, Co, kd, kh, kw, stride, pd, ph, pw, dd, dh, dw};
    g.Do = (Di + 2 * pd - dd * (kd - 1) - 1) / stride + 1;
    g.Ho = (Hi + 2 * ph - dh * (kh - 1) - 1) / stride + 1;
    g.Wo = (Wi + 2 * pw - dw * (kw - 1) - 1) / stride + 1;
    return g;
}
Geom make_geom(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
    return make_geom_nd(N, Di, Hi, Wi, Ci, Co, k, k, k, stride, pad, pad, pad);
}

// rows of DGRAD parity class c and its reduction length in taps
void dgrad_class(const Geom& g, int c, long* rows, int* ntaps) {
    const int S = g.stride;
    const int Ps[3] = {g.pd, g.ph, g.pw}, Ks[3] = {g.kd, g.kh, g.kw};
    const int cc[3] = {c / (S * S), (c / S) % S, c % S};
    const int dims[3] = {g.Di, g.Hi, g.Wi};
    long r = g.N;
    int t = 1;
    for (int a = 0; a < 3; ++a) {
        const int f = ((cc[a] - Ps[a]) % S + S) % S;
        r *= f < dims[a] ? (dims[a] - f + S - 1) / S : 0;
        t *= cc[a] < Ks[a] ? (Ks[a] - cc[a] + S - 1) / S : 0;
    }
    *rows = r; *ntaps = t;
}

// ---- border classes --------------------------------------------------------------------------------------------
// Per axis: runs of consecutive row positions with the same valid tap range.  FWD: output o sees tap t iff
// 0 <= o*S - P + t*L < in;  stride-1 DGRAD: input z gets tap a iff 0 <= z + P - a*L < out.
struct AxisRun { int pos0, cnt, tlo, tcnt; };

int axis_runs(int mode, int n_pos, int n_src, int K, int S, int P, int L, AxisRun* runs, int max_runs) {
    int nr = 0;
    for (int o = 0; o < n_pos; ++o) {
        int lo = K, hi = -1;
        for (int t = 0; t < K; ++t) {
            const int src = (mode == MODE_FWD) ? o * S - P + t * L : o + P - t * L;
            if (src >= 0 && src < n_src) { lo = std::min(lo, t); hi = std::max(hi, t); }
        }
        // a range with holes (possible with dilation) is kept whole: the per-row masks still zero the holes
        const int tlo = hi < lo ? 0 : lo, tcnt = hi < lo ? 0 : hi - lo + 1;
        if (nr && runs[nr - 1].tlo == tlo && runs[nr - 1].tcnt == tcnt) { ++runs[nr - 1].cnt; continue; }
        if (nr == max_runs) return -1;
        runs[nr++] = AxisRun{o, 1, tlo, tcnt};
    }
    return nr;
}

// fills p.bcls / p.n_classes / p.border; returns false when the layer has no padding to skip (or too many classes)
bool build_border_classes(int mode, const Geom& g, ConvParams* p) {
    if (g.Ci == 1 || (mode == MODE_DGRAD && g.stride != 1) || mode == MODE_WGRAD) return false;
    if (env_int("MI_CONV_NO_BORDER")) return false;
    AxisRun rz[3], ry[3], rx[3];
    const bool f = (mode == MODE_FWD);
    const int nz = axis_runs(mode, f ? g.Do : g.Di, f ? g.Di : g.Do, g.kd, g.stride, g.pd, g.dd, rz, 3);
    const int ny = axis_runs(mode, f ? g.Ho : g.Hi, f ? g.Hi : g.Ho, g.kh, g.stride, g.ph, g.dh, ry, 3);
    const int nx = axis_runs(mode, f ? g.Wo : g.Wi, f ? g.Wi : g.Wo, g.kw, g.stride, g.pw, g.dw, rx, 3);
    if (nz < 1 || ny < 1 || nx < 1 || nz * ny * nx < 2) return false;       // -1: more than 3 runs on an axis
    int n = 0;
    for (int a = 0; a < nz; ++a)
        for (int b = 0; b < ny; ++b)
            for (int c = 0; c < nx; ++c) {
                BorderClass& bc = p->bcls[n++];
                bc.pos0[0] = (short)rz[a].pos0; bc.pos0[1] = (short)ry[b].pos0; bc.pos0[2] = (short)rx[c].pos0;
                bc.cnt[0] = (short)rz[a].cnt; bc.cnt[1] = (short)ry[b].cnt; bc.cnt[2] = (short)rx[c].cnt;
                bc.tlo[0] = (short)rz[a].tlo; bc.tlo[1] = (short)ry[b].tlo; bc.tlo[2] = (short)rx[c].tlo;
                bc.tcnt[0] = (short)rz[a].tcnt; bc.tcnt[1] = (short)ry[b].tcnt; bc.tcnt[2] = (short)rx[c].tcnt;
            }
    // worth it only when a real share of the multiply-adds falls on padding (small volumes); on large images the class
    // bookkeeping (row map, mapped epilogue) costs more than the few border rows save (unet_4 forward: -7 %)
    {
        double useful = 0, full = 0;
        for (int c = 0; c < n; ++c) {
            const BorderClass& b = p->bcls[c];
            const double rows = (double)b.cnt[0] * b.cnt[1] * b.cnt[2];
            useful += rows * b.tcnt[0] * b.tcnt[1] * b.tcnt[2];
            full += rows * g.kd * g.kh * g.kw;
        }
        if (useful > 0.92 * full) return false;
    }
    // heaviest reduction first (the kernel pairs the two ends of the list on a CU)
    std::stable_sort(p->bcls, p->bcls + n, [](const BorderClass& u, const BorderClass& v) {
        return u.tcnt[0] * u.tcnt[1] * u.tcnt[2] > v.tcnt[0] * v.tcnt[1] * v.tcnt[2];
    });
    p->n_classes = n;
    p->border = 1;
    return true;
}

struct Setup { ConvParams p; Plan pl; };

// plan + tile table of a border-class launch; red_ch = channels of the reduction (FWD: Ci, DGRAD: Co)
void plan_border(Setup* st, int N, int red_ch, int mode, bool stem) {
    ConvParams& p = st->p;
    const int nc = p.n_classes;
    long rows[MAX_CLASSES], taps_c[MAX_CLASSES];
    for (int c = 0; c < nc; ++c) {
        const BorderClass& b = p.bcls[c];
        rows[c] = (long)N * b.cnt[0] * b.cnt[1] * b.cnt[2];
        taps_c[c] = (long)b.tcnt[0] * b.tcnt[1] * b.tcnt[2];
    }
    auto tiles_of = [&](int bm) { long t = 0; for (int c = 0; c < nc; ++c) t += (rows[c] + bm - 1) / bm; return t; };
    // reduction length seen by the split model: the tile-weighted mean over the classes
    long tw = 0, tt = 0;
    for (int c = 0; c < nc; ++c) { const long t = (rows[c] + 63) / 64; tw += t * taps_c[c]; tt += t; }
    const long mean_taps = std::max<long>(1, (tw + tt - 1) / std::max<long>(tt, 1));
    st->pl = make_plan(mode, stem, p.M, p.Ncols, mean_taps * red_ch, red_ch, tiles_of);
    long acc = 0;
    for (int c = 0; c < nc; ++c) { p.cls_tile_start[c] = (int)acc; acc += (rows[c] + st->pl.bm - 1) / st->pl.bm; }
    p.cls_tile_start[nc] = (int)acc;
}

int setup_conv(int mode, const Geom& g, Setup* st) {
    ConvParams& p = st->p;
    p = ConvParams{};
    p.N = g.N; p.kd = g.kd; p.kh = g.kh; p.kw = g.kw; p.stride = g.stride;
    p.dd = g.dd; p.dh = g.dh; p.dw = g.dw;
    p.pd = g.pd; p.ph = g.ph; p.pw = g.pw; p.Ci = g.Ci; p.Co = g.Co;
    const int taps = g.kd * g.kh * g.kw;
    const bool stem = (g.Ci == 1);
    const long Mout = (long)g.N * g.Do * g.Ho * g.Wo, Min = (long)g.N * g.Di * g.Hi * g.Wi;
    p.n_classes = 1;
    if (mode == MODE_FWD) {
        p.Dg = g.Di; p.Hg = g.Hi; p.Wg = g.Wi; p.Cg = g.Ci;
        p.Dr = g.Do; p.Hr = g.Ho; p.Wr = g.Wo;
        p.M = Mout; p.Ncols = g.Co;
        const long M = p.M;
        if (build_border_classes(mode, g, &p)) {
            plan_border(st, g.N, g.Ci, mode, stem);
        } else {
            st->pl = make_plan(mode, stem, p.M, p.Ncols, (long)taps * g.Ci, stem ? 0 : g.Ci, [M](int bm) { return (M + bm - 1) / bm; });
        }
    } else if (mode == MODE_DGRAD && build_border_classes(mode, g, &p)) {
        p.Dg = g.Do; p.Hg = g.Ho; p.Wg = g.Wo; p.Cg = g.Co;
        p.Dr = g.Di; p.Hr = g.Hi; p.Wr = g.Wi;
        p.M = Min; p.Ncols = g.Ci;
        plan_border(st, g.N, g.Co, mode, stem);
    } else if (mode == MODE_DGRAD) {
        if (stem) return MI_E_UNSUPPORTED;    // the stem's input is the image: no data gradient
        p.Dg = g.Do; p.Hg = g.Ho; p.Wg = g.Wo; p.Cg = g.Co;
        p.Dr = g.Di; p.Hr = g.Hi; p.Wr = g.Wi;
        p.M = Min; p.Ncols = g.Ci;
        const int nc = g.stride * g.stride * g.stride;
        p.n_classes = nc;
        long rows[MAX_CLASSES]; int nt[MAX_CLASSES]; int max_t = 0;
        for (int c = 0; c < nc; ++c) { dgrad_class(g, c, &rows[c], &nt[c]); max_t = std::max(max_t, nt[c]); }
        auto tiles_of = [&](int bm) { long t = 0; for (int c = 0; c < nc; ++c) t += (rows[c] + bm - 1) / bm; return t; };
        st->pl = make_plan(mode, stem, p.M, p.Ncols, (long)std::max(max_t, 1) * g.Co, g.Co, tiles_of);
        long acc = 0;
        for (int c = 0; c < nc; ++c) { p.cls_tile_start[c] = (int)acc; acc += (rows[c] + st->pl.bm - 1) / st->pl.bm; }
        p.cls_tile_start[nc] = (int)acc;
    } else {
        p.Dg = g.Di; p.Hg = g.Hi; p.Wg = g.Wi; p.Cg = g.Ci;
        p.Dr = g.Do; p.Hr = g.Ho; p.Wr = g.Wo;
        p.M = (long)taps * g.Ci; p.Ncols = g.Co; p.n_red_vox = Mout;
        const long M = p.M;
        st->pl = make_plan(mode, stem, p.M, p.Ncols, Mout, 0, [M](int bm) { return (M + bm - 1) / bm; });
    }
    magic31(p.Dr, &p.mgD, &p.shD); magic31(p.Hr, &p.mgH, &p.shH); magic31(p.Wr, &p.mgW, &p.shW);
    // operand extents for the buffer descriptors; 32-bit byte offsets with the top bit as "out of range"
    const long x_bytes = 4 * Min * g.Ci, y_bytes = 4 * Mout * g.Co, w_bytes = 4l * taps * g.Ci * g.Co;
    const long ab = (mode == MODE_DGRAD) ? y_bytes : x_bytes, bb = (mode == MODE_WGRAD) ? y_bytes : w_bytes;
    if (ab >= 0x7fff0000l || bb >= 0x7fff0000l) return MI_E_UNSUPPORTED;   // split the batch on the host side
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.splits = st->pl.splits;
    p.ny_tiles = (p.Ncols + st->pl.bn - 1) / st->pl.bn;
    p.tiles_x = (int)st->pl.tiles_x;
    p.fold = (p.n_classes > 1 && !env_int("MI_CONV_NO_FOLD")) ? 1 : 0;
    p.wbox = (mode == MODE_WGRAD && !stem && !env_int("MI_CONV_NO_BORDER")) ? 1 : 0;
    if (p.wbox) {        // same criterion as the border classes: only when padding is a real share of the work
        ConvParams tmp = ConvParams{};
        p.wbox = build_border_classes(MODE_FWD, g, &tmp) ? 1 : 0;
    }
    return MI_OK;
}

bool is_stem7(const Geom& g) {
    return g.Ci == 1 && g.kd == 7 && g.kh == 7 && g.kw == 7 && g.stride == 2 && g.pd == 3 && g.ph == 3 && g.pw == 3 &&
           g.dd == 1 && g.dh == 1 && g.dw == 1;
}

int direct3_kind(const Geom& g) {
    return mi_direct3_kind(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw);
}
bool is_cube2(const Geom& g) {
    return mi_cube2_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw);
}
// workspace of the direct kernels for this geometry: weight image (+ split-K slabs), or the weight-gradient slabs
size_t direct3_ws_bytes(const Geom& g) {
    if (is_cube2(g)) return mi_cube2_slab_bytes(g.N, g.Ci);
    const int kind = direct3_kind(g);
    if (!kind) return 0;
    size_t b = mi_align_up(mi_direct3_wimg_bytes_kind(kind), 256);
    if (kind == 1 || kind == 5) b = std::max(b, mi_direct3_wgrad_slab_bytes());
    if (kind == 2) b = std::max(b, mi_direct3s_wgrad_slab_bytes());
    if (kind == 3 || kind == 4) b = std::max(b, mi_direct3x_wgrad_slab_bytes(kind));
    return b;
}

// defer_splits != null (weight gradients only): a split launch leaves its slabs in `ws` un-reduced and reports the split
// count - the caller sums many layers' slabs in one mi_splitk_reduce_batch launch; an unsplit launch (or the stem,
// which reduces by itself) reports 1 and `out` is final.
// measurement aid (tools/bench_conv.py --kernels): the kernel family the last convolution call of this thread dispatched to
thread_local const char* g_last_conv_kernel = "none";

int run_conv(int mode, const Geom& g, const float* a_src, const float* b_src, float* out,
             const float* res, const float* mask, int relu, void* ws, size_t ws_bytes, hipStream_t s,
             int* defer_splits = nullptr, int res_bcast = 0) {
    if (defer_splits) *defer_splits = 1;
    // A gathered operand of 2 GiB or more (32-bit buffer offsets): FWD / DGRAD rows are independent per sample,
    // so the batch is cut in halves until every piece fits
    if (mode != MODE_WGRAD && g.N > 1) {
        const long in_per = (long)g.Di * g.Hi * g.Wi * g.Ci, out_per = (long)g.Do * g.Ho * g.Wo * g.Co;
        const long a_per = (mode == MODE_FWD) ? in_per : out_per, o_per = (mode == MODE_FWD) ? out_per : in_per;
        if (4 * a_per * g.N >= 0x7fff0000l) {
            Geom lo = g, hi = g;
            lo.N = g.N / 2; hi.N = g.N - lo.N;
            int rc = run_conv(mode, lo, a_src, b_src, out, res, mask, relu, ws, ws_bytes, s, nullptr, res_bcast);
            if (rc) return rc;
            const long ao = a_per * lo.N, oo = o_per * lo.N;
            return run_conv(mode, hi, a_src + ao, b_src, out + oo, (res && !res_bcast) ? res + oo : res,
                            mask ? mask + oo : nullptr, relu, ws, ws_bytes, s, nullptr, res_bcast);
        }
    }
    // the 7^3 stride-2 stem has its own direct kernels (conv_stem.hip); anything they decline runs below
    if (is_stem7(g) && !env_int("MI_CONV_NO_STEM") && !res_bcast) {
        int rc = MI_E_UNSUPPORTED;
        if (mode == MODE_FWD)
            rc = mi_stem7_fwd(a_src, b_src, out, res, relu, g.N, g.Di, g.Hi, g.Wi, g.Co, conv_arith_bf16x3() ? 1 : 0, ws,
                              ws_bytes, s);
        else if (mode == MODE_WGRAD) rc = mi_stem7_wgrad(a_src, b_src, out, g.N, g.Di, g.Hi, g.Wi, g.Co,
                                                         (conv_arith_bf16x3() && !env_int("MI_STEM_WGRAD_F32")) ? 1 : 0, ws, ws_bytes, s);
        if (rc != MI_E_UNSUPPORTED) { g_last_conv_kernel = mode == MODE_FWD ? "stem_fwd" : "stem_wgrad"; return rc; }
    }
    // layer1-shaped convolutions (3^3, stride 1, 64 -> 64 channels, 8 x 8 planes): patch-resident direct kernel; the
    // weight image goes into `ws` (a short ws keeps the implicit GEMM)
    // Linear layers (1x1x1 window on 1x1x1 "volumes", few rows): one register-staged launch, bias included
    if (conv_arith_bf16x3() && g.Di == 1 && g.Hi == 1 && g.Wi == 1 && g.kd == 1 && g.kh == 1 && g.kw == 1 && g.stride == 1 &&
        g.pd == 0 && g.ph == 0 && g.pw == 0 && !mask && !relu && (!res || res_bcast)) {
        const long xe = (long)g.N * g.Ci, ye = (long)g.N * g.Co, we = (long)g.Ci * g.Co;
        g_last_conv_kernel = "small_gemm";
        if (mode == MODE_FWD && mi_small_gemm_usable(g.N, g.Co, g.Ci))
            return mi_small_gemm_launch(a_src, g.Ci, 1, xe, b_src, g.Co, 1, we, res, out, g.N, g.Co, g.Ci, s);
        if (mode == MODE_DGRAD && !res && mi_small_gemm_usable(g.N, g.Ci, g.Co))
            return mi_small_gemm_launch(a_src, g.Co, 1, ye, b_src, 1, g.Co, we, nullptr, out, g.N, g.Ci, g.Co, s);
        if (mode == MODE_WGRAD && mi_small_gemm_usable(g.Ci, g.Co, g.N))
            return mi_small_gemm_launch(a_src, 1, g.Ci, xe, b_src, g.Co, 1, ye, nullptr, out, g.Ci, g.Co, g.N, s);
    }
    const int dkind = (conv_arith_bf16x3() && !res_bcast) ? direct3_kind(g) : 0;      // (a bias row: the generic kernel's epilogue)
    const size_t dimg = mi_align_up(mi_direct3_wimg_bytes_kind(dkind), 256);
    if (mode != MODE_WGRAD && dkind && ws && ws_bytes >= dimg) {
        const float* wl[1] = {b_src};
        void* il[1] = {ws};
        const int dg[1] = {mode == MODE_DGRAD ? 1 : 0}, kd[1] = {dkind};
        int rc = mi_direct3_prep_kind(wl, il, dg, kd, 1, s);
        if (rc) return rc;
        g_last_conv_kernel = dkind == 1 ? "direct3" : dkind == 2 ? "direct3s" : dkind == 3 ? "direct3 (128 channels)" : dkind == 4 ? "direct3s (256 channels)" : "direct3h (8 x 8 tiles)";
        if (dkind == 1) return mi_direct3_launch(a_src, ws, out, res, mask, relu, g.N, g.Di, s);
        if (dkind == 4) return mi_direct3s_launch256(a_src, ws, out, res, mask, relu, g.N, s);
        if (dkind == 5) return mi_direct3h_launch(a_src, ws, out, res, mask, relu, g.N, g.Di, g.Hi, g.Wi, s);
        if (dkind == 3) return mi_direct3_launch128(a_src, ws, out, res, mask, relu, g.N, g.Di, s);
        return mi_direct3s_launch(a_src, ws, out, res, mask, relu, g.N, s);       // 128-channel kernel on 4^3: final as well
    }
    if (mode != MODE_WGRAD && conv_arith_bf16x3() && !res_bcast && is_cube2(g) && ws && ws_bytes >= mi_cube2_slab_bytes(g.N, g.Ci)) {
        g_last_conv_kernel = "cube2 + reduce";
        int rc = mi_cube2_launch(mode == MODE_DGRAD ? 1 : 0, a_src, b_src, (float*)ws, g.N, g.Ci, s);
        if (rc) return rc;
        return mi_direct3_finish_slabs((const float*)ws, mi_cube2_splits(), (long)g.N * 8 * g.Ci, out, res, mask, relu, s);
    }
    // weight gradients of the convolutions with a 2 x 2 x 2 output (layer3, feature_3d, layer3.0.conv1): final in one launch
    if (mode == MODE_WGRAD && conv_arith_bf16x3() &&
        mi_pair_wgrad_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw))
    {
        // (round 4 also built this from pre-cut operand images + LDS-DMA - conv_pairw.hip, measured not faster: r04_experiments.txt
        // item 26 - it left the library in round 6)
        const int splits = mi_pair_wgrad_splits(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride);
        if (splits == 1 || (ws && ws_bytes >= mi_pair_wgrad_slab_bytes(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride))) {
            g_last_conv_kernel = splits > 1 ? "pair_wgrad + reduce" : "pair_wgrad";
            int rc = mi_pair_wgrad_launch(a_src, b_src, out, (float*)ws, g.N, g.Di, g.Ci, g.Co, g.kd, g.stride, s);
            if (rc || splits == 1) return rc;
            if (defer_splits) { *defer_splits = splits; return MI_OK; }
            return mi_direct3_finish_slabs((const float*)ws, splits, (long)g.kd * g.kh * g.kw * g.Ci * g.Co, out, nullptr, nullptr, 0, s);
        }
    }
    if (mode == MODE_WGRAD && dkind == 1 && ws && ws_bytes >= mi_direct3_wgrad_slab_bytes()) {
        g_last_conv_kernel = "direct3_wgrad + reduce";
        int rc = mi_direct3_wgrad_launch(a_src, b_src, (float*)ws, g.N, g.Di, s);
        if (rc) return rc;
        const int splits = mi_direct3_wgrad_splits();
        if (defer_splits) { *defer_splits = splits; return MI_OK; }
        const long out_elems = 27l * g.Ci * g.Co, n4 = out_elems / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, s,
                           (const float*)ws, splits, out_elems, out, (const float*)nullptr, (const float*)nullptr, 0, n4);
        MI_RETURN_IF_LAUNCH_FAILED();
        return MI_OK;
    }
    // (opt-in, MI_D3S_WGRAD=1: 68 against 87 us for layer2's three gradients alone, but the captured step is 10-30 us SLOWER with it:
    // its workgroups keep 100 KB of LDS each, and next to layer1's data-gradient chain that costs more than the implicit GEMM's
    // small workgroups do - r05_experiments.txt item 9)
    if (mode == MODE_WGRAD && dkind == 2 && ws && ws_bytes >= mi_direct3s_wgrad_slab_bytes() && env_int("MI_D3S_WGRAD")) {
        g_last_conv_kernel = "direct3s_wgrad + reduce";
        const float* xs[1] = {a_src};
        const float* dys[1] = {b_src};
        float* sl[1] = {(float*)ws};
        int rc = mi_direct3s_wgrad_launch_batch(xs, dys, sl, 1, g.N, s);
        if (rc) return rc;
        const int splits = mi_direct3s_wgrad_splits(1);
        if (defer_splits) { *defer_splits = splits; return MI_OK; }
        return mi_direct3_finish_slabs((const float*)ws, splits, 27l * g.Ci * g.Co, out, nullptr, nullptr, 0, s);
    }
    // 64 channels on planes of whole 8 x 8 tiles (layer1 of 64^3 crops; MI_NO_D3T_WGRAD=1: the implicit GEMM)
    if (mode == MODE_WGRAD && dkind == 5 && g.Hi % 8 == 0 && g.Wi % 8 == 0 && ws && ws_bytes >= mi_direct3_wgrad_slab_bytes() &&
        !env_int("MI_NO_D3T_WGRAD")) {
        g_last_conv_kernel = "direct3_wgrad (8 x 8 tiles) + reduce";
        int rc = mi_direct3t_wgrad_launch(a_src, b_src, (float*)ws, g.N, g.Di, g.Hi, g.Wi, s);
        if (rc) return rc;
        const int splits = mi_direct3_wgrad_splits();
        if (defer_splits) { *defer_splits = splits; return MI_OK; }
        return mi_direct3_finish_slabs((const float*)ws, splits, 27l * g.Ci * g.Co, out, nullptr, nullptr, 0, s);
    }
    // 64^3 crops: layer2 (kind 3: 71.3 against 100.5 us at batch 32, 41.8 against 57.7 at batch 16; MI_NO_D3X_WGRAD=1: the implicit GEMM)
    // (layer3, kind 4, on the same kernel: 59.3 against 60.4 / 38.0 against 35.6 us, no gain - r05_experiments.txt; removed in round 6)
    if (mode == MODE_WGRAD && dkind == 3 && !env_int("MI_NO_D3X_WGRAD") && ws &&
        ws_bytes >= mi_direct3x_wgrad_slab_bytes(dkind)) {
        g_last_conv_kernel = "direct3_wgrad (128 channels) + reduce";
        int rc = mi_direct3x_wgrad_launch(dkind, a_src, b_src, (float*)ws, g.N, g.Di, s);
        if (rc) return rc;
        const int splits = mi_direct3x_wgrad_splits(dkind);
        if (defer_splits) { *defer_splits = splits; return MI_OK; }
        return mi_direct3_finish_slabs((const float*)ws, splits, 27l * g.Ci * g.Co, out, nullptr, nullptr, 0, s);
    }
    Setup st;
    int rc = setup_conv(mode, g, &st);
    if (rc) return rc;
    g_last_conv_kernel = st.pl.splits > 1 ? "implicit GEMM + reduce" : "implicit GEMM";
    ConvParams& p = st.p;
    Plan& pl = st.pl;
    p.a_src = a_src; p.b_src = b_src; p.res = res; p.mask = mask; p.relu = relu;
    p.res_bcast = res_bcast;
    const bool stem = (g.Ci == 1);
    const long out_elems = p.M * p.Ncols;
    if (pl.splits > 1) {
        size_t need = sizeof(float) * (size_t)out_elems * pl.splits;
        if (!ws || ws_bytes < need) pl.splits = 1;     // unsplit schedule: same values up to summation order
    }
    p.splits = pl.splits;
    if (pl.splits > 1) { p.out = (float*)ws; p.slab_stride = out_elems; }
    else { p.out = out; p.slab_stride = 0; }
    if (mode == MODE_FWD) rc = stem ? launch_mode<MODE_FWD, true>(p, pl, s) : launch_mode<MODE_FWD, false>(p, pl, s);
    else if (mode == MODE_DGRAD) rc = launch_mode<MODE_DGRAD, false>(p, pl, s);
    else rc = stem ? launch_mode<MODE_WGRAD, true>(p, pl, s) : launch_mode<MODE_WGRAD, false>(p, pl, s);
    if (rc) return rc;
    if (pl.splits > 1 && defer_splits) { *defer_splits = pl.splits; return MI_OK; }
    if (pl.splits > 1) {
        long n4 = out_elems / 4;
        int blocks = (int)std::min<long>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws,
                           pl.splits, out_elems, out, res, mask, relu, n4, res_bcast ? p.Ncols : 0);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}

}  // namespace

// split-K slabs of a direct kernel -> out, with the convolution epilogue (conv_direct3.hip)
int mi_direct3_finish_slabs(const float* slabs, int n_slabs, long out_elems, float* out, const float* res, const float* mask,
                            int relu, hipStream_t s) {
    const long n4 = out_elems / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, slabs,
                       n_slabs, out_elems, out, res, mask, relu, n4, 0);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

extern "C" size_t mi_conv3d_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int k,
                                            int stride, int pad) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return 0;
    size_t best = is_stem7(g) ? std::max(mi_stem7_wgrad_workspace_bytes(g.N, g.Di, g.Hi, g.Wi, g.Co), mi_stem7_fwd_workspace_bytes()) : 0;
    best = std::max(best, direct3_ws_bytes(g));
    if (mi_pair_wgrad_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw))
        best = std::max(best, mi_pair_wgrad_slab_bytes(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride));
    for (int mode = 0; mode < 3; ++mode) {
        Setup st;
        if (setup_conv(mode, g, &st)) continue;
        if (st.pl.splits > 1) best = std::max(best, sizeof(float) * (size_t)st.p.M * st.p.Ncols * st.pl.splits);
    }
    return best + 256;
}

extern "C" int mi_conv3d_fwd_f32(const float* x, const float* w, float* y, const float* res,
                                 int relu, int N, int Di, int Hi, int Wi, int Ci, int Co, int k,
                                 int stride, int pad, void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!x || !w || !y || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, res, nullptr, relu, ws, ws_bytes, (hipStream_t)stream);
}

/* measurement aid: the kernel family the last mi_conv* call of the calling thread ran ("direct3", "cube2 + reduce", ...) */
extern "C" const char* mi_debug_last_conv_kernel(void) { return g_last_conv_kernel; }
void mi_note_conv_kernel(const char* name) { g_last_conv_kernel = name; }       // (entry points outside run_conv: conv_cube2.hip)

/* nn.Linear (+ bias) followed by training-mode nn.BatchNorm1d (+ ReLU) in ONE launch - the projection MLP of the MoCo-3D
 * encoder (models/networks/moco_encoder_3d.py:238-255: Linear, BatchNorm1d, ReLU three times over a batch of <= 64 rows).
 * xlin = x W + bias is written as well (the BatchNorm backward reads it); y = act(bn(xlin)); save = mean[Co], invstd[Co];
 * running statistics / num_batches_tracked updated when given.  Same arithmetic as mi_linear_fwd_f32 + mi_bn_small_fwd.
 * MI_E_UNSUPPORTED for M > 64 (a workgroup's tile must hold every row) or shapes the small product declines. */
extern "C" int mi_linear_bn_fwd_f32(const float* x, const float* w, const float* bias, float* xlin, float* y, int M, int Ci,
                                    int Co, const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    float* save_mean_invstd, int relu, mi_stream_t stream) {
    if (!x || !w || !xlin || !y || !save_mean_invstd || M <= 0 || Ci <= 0 || Co <= 0) return MI_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MI_E_ARG;
    if (M > 64 || !conv_arith_bf16x3() || !mi_small_gemm_usable(M, Co, Ci) || env_int("MI_NO_LINEAR_BN")) return MI_E_UNSUPPORTED;
    MiSmallGemmBN bn = {y, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean_invstd, relu,
                        nullptr};
    return mi_small_gemm_launch(x, Ci, 1, (long)M * Ci, w, Co, 1, (long)Ci * Co, bias, xlin, M, Co, Ci, (hipStream_t)stream, &bn);
}

/* nn.Linear (+ bias) with the column sums a SyncBatchNorm behind it needs (sums[0..Co) = sum of xlin, sums[Co..2Co) = sum of
 * xlin^2 over this rank's M <= 64 rows, doubles: what mi_bn_stats(xlin) would produce) from the product's epilogue - the
 * all-reduce of `sums` and mi_bn_apply_fwd follow (models/networks/moco_encoder_3d.py:238-255 under
 * SyncBatchNorm.convert_sync_batchnorm, moco_main.py:65-66). */
extern "C" int mi_linear_stats_fwd_f32(const float* x, const float* w, const float* bias, float* xlin, double* sums, int M, int Ci,
                                       int Co, mi_stream_t stream) {
    if (!x || !w || !xlin || !sums || M <= 0 || Ci <= 0 || Co <= 0) return MI_E_ARG;
    if (M > 64 || !conv_arith_bf16x3() || !mi_small_gemm_usable(M, Co, Ci) || env_int("MI_NO_LINEAR_BN")) return MI_E_UNSUPPORTED;
    MiSmallGemmBN bn = {};
    bn.sums_only = sums;
    return mi_small_gemm_launch(x, Ci, 1, (long)M * Ci, w, Co, 1, (long)Ci * Co, bias, xlin, M, Co, Ci, (hipStream_t)stream, &bn);
}

/* The 7^3 stride-2 single-channel stem convolution (models/networks/moco_encoder_3d.py:170-176 `conv1`) together with
 * the batch statistics its BatchNorm3d needs (`bn1`, :326-328): sums[0..Co) = column sums of y, sums[Co..2Co) = column sums
 * of y^2 (doubles, what mi_bn_stats would produce from y), taken from the output tiles while they are still in registers.
 * MI_E_UNSUPPORTED where the stem kernel does not apply (shape, Co != 64, MI_CONV_ARITH=f32): the caller then runs
 * mi_conv3d_fwd_f32 + mi_bn_stats. */
extern "C" size_t mi_conv3d_stem_stats_workspace_bytes(int N, int Di, int Hi, int Wi, int Co) {
    Geom g = make_geom(N, Di, Hi, Wi, 1, Co, 7, 2, 3);
    if (!geom_ok(g) || !is_stem7(g) || Co != 64) return 0;
    return mi_stem7_fwd_stats_workspace_bytes(N, Di, Hi, Wi);
}

extern "C" int mi_conv3d_stem_stats_f32(const float* x, const float* w, float* y, int N, int Di, int Hi, int Wi, int Co,
                                        double* sums, void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, 1, Co, 7, 2, 3);
    if (!x || !w || !y || !sums || !geom_ok(g)) return MI_E_ARG;
    if (!is_stem7(g) || !conv_arith_bf16x3() || env_int("MI_CONV_NO_STEM") || env_int("MI_STEM_NO_STATS")) return MI_E_UNSUPPORTED;
    return mi_stem7_fwd(x, w, y, nullptr, 0, N, Di, Hi, Wi, Co, 1, ws, ws_bytes, (hipStream_t)stream, sums);
}

/* nn.Linear forward in one pass: y[M][Co] = x[M][Ci] . W + bias (W in kernel layout [Ci][Co], bias may be NULL) - the
 * 1 x 1 x 1 convolution with the bias added in the epilogue of the kernel (or of its split-K reduce). */
extern "C" int mi_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int M, int Ci, int Co,
                                 void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(M, 1, 1, 1, Ci, Co, 1, 1, 0);
    if (!x || !w || !y || !geom_ok(g)) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, bias, nullptr, 0, ws, ws_bytes, (hipStream_t)stream, nullptr, bias ? 1 : 0);
}

extern "C" int mi_conv3d_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                                   const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co,
                                   int k, int stride, int pad, void* ws, size_t ws_bytes,
                                   mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!dy || !w || !dx || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_DGRAD, g, dy, w, dx, res, mask, 0, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_conv3d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int Di,
                                   int Hi, int Wi, int Ci, int Co, int k, int stride, int pad,
                                   void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom(N, Di, Hi, Wi, Ci, Co, k, stride, pad);
    if (!x || !dy || !dw || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_WGRAD, g, x, dy, dw, nullptr, nullptr, 0, ws, ws_bytes, (hipStream_t)stream);
}

// ---- per-axis window / padding (2-D convolutions of the SimSiam 2-D encoder: kd = 1, pd = 0, D = 1)
extern "C" size_t mi_convnd_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh,
                                            int kw, int stride, int pd, int ph, int pw) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return 0;
    size_t best = is_stem7(g) ? std::max(mi_stem7_wgrad_workspace_bytes(g.N, g.Di, g.Hi, g.Wi, g.Co), mi_stem7_fwd_workspace_bytes()) : 0;
    best = std::max(best, direct3_ws_bytes(g));
    if (mi_pair_wgrad_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw))
        best = std::max(best, mi_pair_wgrad_slab_bytes(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride));
    for (int mode = 0; mode < 3; ++mode) {
        Setup st;
        if (setup_conv(mode, g, &st)) continue;
        if (st.pl.splits > 1) best = std::max(best, sizeof(float) * (size_t)st.p.M * st.p.Ncols * st.pl.splits);
    }
    return best + 256;
}

extern "C" int mi_convnd_fwd_f32(const float* x, const float* w, float* y, const float* res, int relu,
                                 int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                 int stride, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                                 mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!x || !w || !y || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, res, nullptr, relu, ws, ws_bytes, (hipStream_t)stream);
}

/* y = act(conv(x, w) + bias[co]) - the same launch with the residual read as ONE row of Co values.  Evaluation-mode BatchNorm
 * folds into it (w' = w * gamma / sqrt(var + eps) per output channel, bias = beta - mean * gamma / sqrt(var + eps)): the detector's
 * U-Net at inference (models/networks/unet_small.py:30-97, unet.py:198-399: conv -> BatchNorm -> ReLU) drops its BatchNorm passes. */
extern "C" int mi_convnd_fwd_bias_f32(const float* x, const float* w, float* y, const float* bias, int relu,
                                      int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                      int stride, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                                      mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!x || !w || !y || !bias || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, bias, nullptr, relu, ws, ws_bytes, (hipStream_t)stream, nullptr, 1);
}

extern "C" int mi_convnd_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                                   const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co,
                                   int kd, int kh, int kw, int stride, int pd, int ph, int pw, void* ws,
                                   size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!dy || !w || !dx || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_DGRAD, g, dy, w, dx, res, mask, 0, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_convnd_wgrad_f32(const float* x, const float* dy, float* dw, int N, int Di, int Hi,
                                   int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd,
                                   int ph, int pw, void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!x || !dy || !dw || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_WGRAD, g, x, dy, dw, nullptr, nullptr, 0, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_convnd_wgrad_slabs_f32(const float* x, const float* dy, float* dw, int N, int Di, int Hi, int Wi, int Ci,
                                         int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw, void* ws,
                                         size_t ws_bytes, int* splits_out, mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!x || !dy || !dw || !splits_out || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_WGRAD, g, x, dy, dw, nullptr, nullptr, 0, ws, ws_bytes, (hipStream_t)stream, splits_out);
}

/* nb weight gradients of ONE geometry in one launch (the engine issues a stage's weight gradients together, behind the stage's
 * data-gradient chain): problem i reads xs[i] / dys[i] and leaves *splits_out slabs in wss[i] (> 1: the caller's
 * mi_splitk_reduce_batch sums them into dws[i]) or, *splits_out == 1, the final gradient in dws[i].  MI_E_UNSUPPORTED: this
 * geometry has no batched kernel - the caller issues nb single calls.  ws_bytes: size of EACH workspace. */
extern "C" int mi_convnd_wgrad_slabs_batch_f32(const float* const* xs, const float* const* dys, float* const* dws, void* const* wss,
                                               int nb, int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                               int stride, int pd, int ph, int pw, size_t ws_bytes, int* splits_out,
                                               mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, stride, pd, ph, pw);
    if (!xs || !dys || !dws || !wss || !splits_out || nb < 1 || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    for (int i = 0; i < nb; ++i) if (!xs[i] || !dys[i] || !dws[i]) return MI_E_ARG;
    if (nb < 2 || !conv_arith_bf16x3() || env_int("MI_NO_WGRAD_BATCH")) return MI_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* slabs[MI_WGRAD_BATCH_MAX];
    if (nb > MI_WGRAD_BATCH_MAX) return MI_E_UNSUPPORTED;
    for (int i = 0; i < nb; ++i) slabs[i] = (float*)wss[i];
    if (is_stem7(g)) return MI_E_UNSUPPORTED;
    // same order of preference as run_conv(MODE_WGRAD, ...)
    if (mi_pair_wgrad_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw)) {
        if (nb > mi_pair_wgrad_batch_max()) return MI_E_UNSUPPORTED;
        const int splits = mi_pair_wgrad_splits(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride);
        if (splits > 1) {
            if (ws_bytes < mi_pair_wgrad_slab_bytes(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride)) return MI_E_UNSUPPORTED;
            for (int i = 0; i < nb; ++i) if (!wss[i]) return MI_E_ARG;
        }
        g_last_conv_kernel = splits > 1 ? "pair_wgrad x nb + reduce" : "pair_wgrad x nb";
        *splits_out = splits;
        return mi_pair_wgrad_launch_batch(xs, dys, dws, slabs, nb, g.N, g.Di, g.Ci, g.Co, g.kd, g.stride, s);
    }
    if (direct3_kind(g) == 1) {
        if (nb > mi_direct3_wgrad_batch_max() || ws_bytes < mi_direct3_wgrad_slab_bytes()) return MI_E_UNSUPPORTED;
        for (int i = 0; i < nb; ++i) if (!wss[i]) return MI_E_ARG;
        g_last_conv_kernel = "direct3_wgrad x nb + reduce";
        *splits_out = mi_direct3_wgrad_batch_splits(nb);
        return mi_direct3_wgrad_launch_batch(xs, dys, slabs, nb, g.N, g.Di, s);
    }
    if (direct3_kind(g) == 2 && env_int("MI_D3S_WGRAD")) {
        if (nb > mi_direct3_wgrad_batch_max() || ws_bytes < mi_direct3s_wgrad_slab_bytes()) return MI_E_UNSUPPORTED;
        for (int i = 0; i < nb; ++i) if (!wss[i]) return MI_E_ARG;
        g_last_conv_kernel = "direct3s_wgrad x nb + reduce";
        *splits_out = mi_direct3s_wgrad_splits(nb);
        return mi_direct3s_wgrad_launch_batch(xs, dys, slabs, nb, g.N, s);
    }
    //  64^3 crops: layer2 (kind 3) and layer1 on 8 x 8 tiles (kind 5) have dedicated single-launch kernels in run_conv and no
    // batched form: the caller issues single launches, which reach them - the batched implicit GEMM below would bypass them
    {
        const int dk = direct3_kind(g);
        if (dk == 3 && !env_int("MI_NO_D3X_WGRAD")) return MI_E_UNSUPPORTED;
        if (dk == 5 && g.Hi % 8 == 0 && g.Wi % 8 == 0 && !env_int("MI_NO_D3T_WGRAD")) return MI_E_UNSUPPORTED;
    }
    Setup st;
    int rc = setup_conv(MODE_WGRAD, g, &st);
    if (rc) return rc;
    if (g.Ci == 1) return MI_E_UNSUPPORTED;
    ConvParams& p = st.p;
    Plan& pl = st.pl;
    // the single launch's split count: the slabs (and their sum) are then the single launches' bit for bit.  (MI_WGRAD_BATCH_SPLITS:
    // tuning - a problem's share of the chip is 1 / nb of what the plan assumed, fewer and longer splits would do)
    int splits = pl.splits;
    if (const char* v = getenv("MI_WGRAD_BATCH_SPLITS")) { const int sv = atoi(v); if (sv >= 1 && sv <= pl.splits) splits = sv; }
    const long out_elems = p.M * p.Ncols;
    if (splits > 1) {
        if (ws_bytes < sizeof(float) * (size_t)out_elems * splits) return MI_E_UNSUPPORTED;
        for (int i = 0; i < nb; ++i) if (!wss[i]) return MI_E_ARG;
    }
    pl.splits = splits;
    p.splits = splits;
    p.res = nullptr; p.mask = nullptr; p.relu = 0; p.res_bcast = 0;
    p.a_src = xs[0]; p.b_src = dys[0];
    p.nbatch = nb;
    for (int i = 0; i < nb; ++i) { p.a_tab[i] = xs[i]; p.b_tab[i] = dys[i]; p.out_tab[i] = splits > 1 ? slabs[i] : dws[i]; }
    p.out = p.out_tab[0];
    p.slab_stride = splits > 1 ? out_elems : 0;
    g_last_conv_kernel = splits > 1 ? "implicit GEMM x nb + reduce" : "implicit GEMM x nb";
    rc = launch_mode<MODE_WGRAD, false>(p, pl, s);
    if (rc) return rc;
    *splits_out = splits;
    return MI_OK;
}

extern "C" int mi_splitk_reduce_batch(const void* const* slabs, void* const* outs, const int* n_slabs, const long* out_elems,
                                      int n, mi_stream_t stream) {
    if (n <= 0) return MI_OK;
    if (!slabs || !outs || !n_slabs || !out_elems) return MI_E_ARG;
    for (int i0 = 0; i0 < n; i0 += MI_REDUCE_BATCH) {
        SplitBatch bt = {};
        const int m = std::min(MI_REDUCE_BATCH, n - i0);
        long max4 = 1;
        for (int i = 0; i < m; ++i) {
            if (!slabs[i0 + i] || !outs[i0 + i] || n_slabs[i0 + i] < 1 || (out_elems[i0 + i] & 3)) return MI_E_ARG;
            bt.d[i] = SplitDesc{(const float*)slabs[i0 + i], (float*)outs[i0 + i], out_elems[i0 + i], out_elems[i0 + i] / 4,
                                n_slabs[i0 + i], 0};
            max4 = std::max(max4, bt.d[i].n4);
        }
        const int bx = (int)std::min<long>((max4 + 255) / 256, 256);
        hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, bt);
        MI_RETURN_IF_LAUNCH_FAILED();
    }
    return MI_OK;
}

// ---- dilated windows (stride 1): the 3-D head of the detector, unet_small.py:38-41 (kernel 3x3x3,
// dilation (1,4,4), padding (1,4,4))
extern "C" size_t mi_convnd_dil_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh,
                                                int kw, int pd, int ph, int pw, int dd, int dh, int dw) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, 1, pd, ph, pw, dd, dh, dw);
    if (!geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return 0;
    size_t best = is_stem7(g) ? std::max(mi_stem7_wgrad_workspace_bytes(g.N, g.Di, g.Hi, g.Wi, g.Co), mi_stem7_fwd_workspace_bytes()) : 0;
    best = std::max(best, direct3_ws_bytes(g));
    if (mi_pair_wgrad_usable(g.N, g.Di, g.Hi, g.Wi, g.Ci, g.Co, g.kd, g.kh, g.kw, g.stride, g.pd, g.ph, g.pw, g.dd, g.dh, g.dw))
        best = std::max(best, mi_pair_wgrad_slab_bytes(g.N, g.Di, g.Ci, g.Co, g.kd, g.stride));
    for (int mode = 0; mode < 3; ++mode) {
        Setup st;
        if (setup_conv(mode, g, &st)) continue;
        if (st.pl.splits > 1) best = std::max(best, sizeof(float) * (size_t)st.p.M * st.p.Ncols * st.pl.splits);
    }
    return best + 256;
}

extern "C" int mi_convnd_dil_fwd_f32(const float* x, const float* w, float* y, const float* res, int relu,
                                     int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                     int pd, int ph, int pw, int dd, int dh, int dw, void* ws,
                                     size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, 1, pd, ph, pw, dd, dh, dw);
    if (!x || !w || !y || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_FWD, g, x, w, y, res, nullptr, relu, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_convnd_dil_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                                       const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co,
                                       int kd, int kh, int kw, int pd, int ph, int pw, int dd, int dh,
                                       int dw, void* ws, size_t ws_bytes, mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, 1, pd, ph, pw, dd, dh, dw);
    if (!dy || !w || !dx || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_DGRAD, g, dy, w, dx, res, mask, 0, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mi_convnd_dil_wgrad_f32(const float* x, const float* dy, float* dwt, int N, int Di, int Hi,
                                       int Wi, int Ci, int Co, int kd, int kh, int kw, int pd, int ph,
                                       int pw, int dd, int dh, int dw, void* ws, size_t ws_bytes,
                                       mi_stream_t stream) {
    Geom g = make_geom_nd(N, Di, Hi, Wi, Ci, Co, kd, kh, kw, 1, pd, ph, pw, dd, dh, dw);
    if (!x || !dy || !dwt || !geom_ok(g) || g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return MI_E_ARG;
    return run_conv(MODE_WGRAD, g, x, dy, dwt, nullptr, nullptr, 0, ws, ws_bytes, (hipStream_t)stream);
}
