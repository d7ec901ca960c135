// DoG particle picker, filter stage on the MATRIX cores (round 5): the three one-dimensional Gaussian passes of both sigmas
// as banded-Toeplitz products, f32-equivalent (bf16x3: both operands cut exactly into three bf16 pieces, six products,
// f32 accumulation - the arithmetic of the convolution kernels).
//   dogm_xz_kernel : x pass and z pass of BOTH Gaussians, one read of the tomogram, nothing but registers in between
//   dogm_y_kernel  : y pass of both + DoG + border + `_nms_xy` (3x3) + fp64 statistics + candidate compaction
// Replaces (reference, cet_pick/...): the two scipy.ndimage.gaussian_filter calls, `rec_i = gaussian(s2) - gaussian(s1)`,
// the border zeroing, `_nms_xy(.., kernel=3)` and the `mean + 0.5 std` statistics of utils/image.py:152-179 - and the
// vector-unit kernels of round 4 (infer_dogf.hip: 54 multiply-adds + adds per voxel and pass, 203 + 152 us on
// 256x512x512; both bound by vector issue).  Opt-in with `MI_DOGM=1` (see mi_dogm_usable).
//
// A 1-D filter of radius R over 32 consecutive outputs reads 32 + 2R <= 72 inputs: out[i] = sum_p T[i][p] in[p] with
// T[i][p] = w[|p - i - 20|] over an 80-wide window that starts 20 before the first output - five k-steps of
// v_mfma_f32_32x32x16_bf16 for R <= 20, four (p in [8, 64)) for R <= 12.  The 32 x 16 blocks of T are constants: built
// once per workgroup into LDS (27 KB, already cut), one ds_read_b128 per block and plane.  What makes the chain cheap is
// that NO transposition is ever needed between passes:
//   * the k axis of an MFMA is a free permutation.  With k slot (h, j) of a 16-block standing for position
//     pi(h, j) = (j & 3) + 8 (j >> 2) + 4 h, the eight values a lane holds of a 32 x 32 accumulator half (rows
//     (r & 3) + 8 (r >> 2) + 4 h, r = 8 s' .. 8 s' + 7) ARE the k-step s' of an operand whose k axis is the accumulator's
//     row axis.  T is built in the same slot order and serves as A or B operand alike;
//   * x pass: A = data (lane = z row, 8 x of the window per k-step: two 16-byte loads), B = T  ->  C[z][x'] (lane = x');
//     z pass: A = T, B = C of the x pass (its rows are z: the k axis)  ->  C'[z'][x'] (lane = x': 128-byte row stores).
//     A wave owns 32 x' of one y row and marches along z: every 32 planes one x-pass tile, whose two k-steps go into
//     the three output tiles that need them (accumulators of tiles t, t-1, t-2: the "window" never exists);
//   * y pass (second kernel): A = T, B = data (lane = x column, 8 rows per k-step: 4-byte loads, 128 bytes per row) ->
//     C[y'][x] with the lane still the column: DoG = difference of two accumulators, x neighbours of the NMS by DPP,
//     y neighbours inside the lane or from the other lane half.
// Borders: z reflects (scipy 'reflect', one reflection - the host admits no shallower volume), x and y never do - the
// picker zeroes a border of 30 / 60 >= 20 voxels, so the window of a live output lies inside the image; rows / columns
// past the image read as zeros through the buffer range check (0 x weight stays finite).
// Per 32 x 32 outputs and pass: 54 MFMAs (30 + 24) = 1,728 matrix cycles against ~2,900 vector cycles before.
// hipcc-flags: -fno-slp-vectorize
#include "common.h"
#include "infer_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MOFF = 20;              // outputs start MOFF inside their 80-wide window
constexpr int NF5 = 5, NF3 = 4;       // k-steps of the wide / narrow filter
constexpr int NFRAG = NF5 + NF3;
constexpr int MT = 256;               // threads of both kernels: four waves, one per SIMD
constexpr int MY_OWN = 30;            // columns a strip of the y march owns (32 computed)
constexpr int MRING = 512;

template <int R>
struct MTaps { float w[R + 1]; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t m_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
// exact three-way bf16 cut of 8 f32 (truncation; conv_cube2.hip cut8r)
__device__ __forceinline__ void m_cut8(const float (&v)[8], bf16x8 (&o)[3]) {
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
// the six products of weight <= 2 of (a0 + a1 + a2)(b0 + b1 + b2), smallest first
__device__ __forceinline__ f32x16 m_mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pr]], b[PB[pr]], acc, 0, 0, 0);
    return acc;
}

// The Toeplitz blocks of both filters, cut, in LDS: block f (0..4: wide filter k-steps, 5..8: narrow), plane pl, lane l.
// Value of slot j of lane (i = l & 31, h = l >> 5) of k-step s: w[|16 s + pi(h, j) - i - MOFF|].
struct MToep {
    u32x4 t[NFRAG * 3 * 64];          // 27,648 bytes
    float tw[2][32];                  // taps, zero beyond the radius
};
template <int RA, int RB>
__device__ __forceinline__ void m_build_toeplitz(MToep& L, const MTaps<RA>& wa, const MTaps<RB>& wb, int tid) {
    if (tid < 32) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int d = 0; d <= RB; ++d) {                    // (a select chain: the taps are kernel arguments)
            if (d == tid) b = wb.w[d];
            if (d <= RA && d == tid) a = wa.w[d];
        }
        L.tw[0][tid] = a;
        L.tw[1][tid] = b;
    }
    __syncthreads();
    for (int idx = tid; idx < NFRAG * 64; idx += MT) {
        const int f = idx >> 6, l = idx & 63, i = l & 31, hh = l >> 5;
        const int wide = f < NF5 ? 1 : 0, s = f < NF5 ? f : f - NF5;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pp = 16 * s + (j & 3) + 8 * (j >> 2) + 4 * hh;
            int d = pp - i - MOFF;
            d = d < 0 ? -d : d;
            v[j] = d < 32 ? L.tw[wide][d] : 0.f;
        }
        bf16x8 o[3];
        m_cut8(v, o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) L.t[(f * 3 + pl) * 64 + l] = __builtin_bit_cast(u32x4, o[pl]);
    }
    __syncthreads();
}
// (the lane index goes through an empty asm: every call reads its block where it is used - merged by the compiler, the 27
// blocks stay in 108 registers; a volatile pointer loses the LDS address space and becomes a flat load)
__device__ __forceinline__ void m_toep(const MToep& L, int f, int lane, bf16x8 (&o)[3]) {
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o[pl] = __builtin_bit_cast(bf16x8, L.t[(f * 3 + pl) * 64 + l]);
}

// ---------------------------------------------------------------------------------------------------------------------
// x + z
// ---------------------------------------------------------------------------------------------------------------------
struct DogmXZ {
    const float* rec;
    float* g1;               // narrow sigma after the x and z passes: planes [bz, D - bz), rows [ylo, yhi), columns [xa, xa + 32 n_xstrips)
    float* g2;               // wide sigma
    int D, H, W, bz;
    int ylo, yhi, xa, n_xstrips, n_quads, rows_per_iter, n_iter;
    unsigned vol_bytes;
    unsigned* clr[2];
    unsigned clr_n[2];
};

template <int RA, int RB>
__global__ __launch_bounds__(MT, 1) void dogm_xz_kernel(DogmXZ p, MTaps<RA> wa, MTaps<RB> wb) {
    __shared__ MToep L;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    {   // the picker's header (and candidate bitmap) are zeroed here: no clearing pass
        const long gt = (long)blockIdx.x * MT + tid, nt = (long)gridDim.x * MT;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (p.clr[k])
                for (long i = gt; i < (long)p.clr_n[k]; i += nt) p.clr[k][i] = 0u;
    }
    m_build_toeplitz<RA, RB>(L, wa, wb, tid);
    const int D = p.D, H = p.H, W = p.W;
    const __amdgpu_buffer_rsrc_t rrs = m_rsrc(p.rec, p.vol_bytes), g1rs = m_rsrc(p.g1, p.vol_bytes), g2rs = m_rsrc(p.g2, p.vol_bytes);
    const int G0 = p.bz - MOFF;                            // z of k-step 0, slot position 0
    const int NJ = (D - 2 * p.bz + 31) / 32;               // output tiles along z
    const int NTX = NJ + 2;                                // x-pass tiles (the last one feeds k-step 4 of the last output tile)
    // blocks b, b + 8, b + 16, b + 24 - one XCD under round-robin placement - take the quads of ONE row (their windows overlap)
    const int b = blockIdx.x;
    const int quad = (b >> 3) % p.n_quads, r_local = (b / (8 * p.n_quads)) * 8 + (b & 7);
    const int strip = quad * 4 + wv;
    if (r_local >= p.rows_per_iter || strip >= p.n_xstrips) return;     // (wave-uniform; no barrier below)
    const int x0 = p.xa + 32 * strip, xw = x0 - MOFF;
    // One step = 18 groups of six MFMAs (9 of the x pass, 9 of the z pass).  A wave has its SIMD to itself (~330 registers),
    // so nothing fills the matrix pipe's shadow but its own vector work - the exact cuts of the next operand (11 instructions
    // per pair of values), the Toeplitz block reads, the stores of the finished tile, the next tile's loads.  Each piece is
    // placed behind ONE MFMA and fenced there (sched_barrier): left to the scheduler - and with sched_group_barrier masks too -
    // the cuts went out as blocks between runs of MFMAs and the two pipes took turns (305 - 345 us).
    // the Toeplitz blocks of both filters in REGISTERS (27 x 4: a wave alone on its SIMD has 512): read from LDS where they are used
    // they cost an LDS instruction + a wait per MFMA pair out of an issue budget of six
    u32x4 T5r[NF5][3], T3r[NF3][3];
#pragma unroll
    for (int f = 0; f < NF5; ++f)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) T5r[f][pl] = L.t[(f * 3 + pl) * 64 + lane];
#pragma unroll
    for (int f = 0; f < NF3; ++f)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) T3r[f][pl] = L.t[((NF5 + f) * 3 + pl) * 64 + lane];
    const bool x_inside = xw + 16 * NF5 <= W;               // (uniform) the window does not leave the row: immediate offsets
    const bool x_full = x0 + 32 <= W;                       // (uniform)
    const unsigned zstride = 4u * (unsigned)(H * W);
    for (int it = 0; it < p.n_iter; ++it) {
        const int y = p.ylo + it * p.rows_per_iter + r_local;
        if (y >= p.yhi) break;
        auto issue = [&](int t, u32x4 (&raw)[2 * NF5]) {
            int z = G0 + 32 * t + n;
            z = z < 0 ? -1 - z : (z >= D ? 2 * D - 1 - z : z);          // scipy 'reflect', once
            z = min(max(z, 0), D - 1);                                  // (rows past the reflection feed no live output)
            const unsigned rowoff = 4u * (unsigned)((z * H + y) * W + xw + 4 * h);
            if (x_inside) {
#pragma unroll
                for (int q = 0; q < 2 * NF5; ++q) raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(rowoff + 32u * (unsigned)q), 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 2 * NF5; ++q) {
                    const int xc = xw + 4 * h + 8 * q;
                    const unsigned off = xc < W ? rowoff + 32u * (unsigned)q : 0x80000000u;        // past the row: zeros
                    raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off, 0, 0);
                }
            }
        };
        // The exact cut of a pair of values (elements 2 d, 2 d + 1 of a fragment) in two halves of 5 / 6 vector instructions, so that
        // no MFMA has more than six behind it (11 behind one MFMA stretch that gap from 32 to 52 cycles): half A = the top plane and
        // the two remainders, half B = the middle and bottom planes.  `HC` carries the remainders from A to B.
        struct HC { float ra[4], rb[4]; };
        auto hcA = [&](float va, float vb, int d, HC& st, u32x4 (&o)[3]) {
            const unsigned a0 = __float_as_uint(va), b0 = __float_as_uint(vb);
            st.ra[d] = va - __uint_as_float(a0 & 0xffff0000u);
            st.rb[d] = vb - __uint_as_float(b0 & 0xffff0000u);
            o[0][d] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
        };
        auto hcB = [&](int d, HC& st, u32x4 (&o)[3]) {
            const unsigned a1 = __float_as_uint(st.ra[d]), b1 = __float_as_uint(st.rb[d]);
            const unsigned a2 = __float_as_uint(st.ra[d] - __uint_as_float(a1 & 0xffff0000u));
            const unsigned b2 = __float_as_uint(st.rb[d] - __uint_as_float(b1 & 0xffff0000u));
            o[1][d] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
            o[2][d] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
        };
        // half-op q = 0 .. 7 of the cut of k-step s of `raw` / of half sp of an accumulator
        auto hc_raw = [&](const u32x4 (&raw)[2 * NF5], int s, int q, HC& st, u32x4 (&o)[3]) {
            const int d = q >> 1;
            if (q & 1) { hcB(d, st, o); return; }
            const u32x4& w = raw[2 * s + (d >> 1)];
            hcA(__uint_as_float(w[2 * (d & 1)]), __uint_as_float(w[2 * (d & 1) + 1]), d, st, o);
        };
        auto hc_acc = [&](const f32x16& a, int sp, int q, HC& st, u32x4 (&o)[3]) {
            const int d = q >> 1;
            if (q & 1) hcB(d, st, o); else hcA(a[8 * sp + 2 * d], a[8 * sp + 2 * d + 1], d, st, o);
        };
        // six MFMAs, `side(k)` fenced behind the k-th
        auto grp = [&](f32x16& acc, const u32x4 (&a)[3], const u32x4 (&b)[3], auto&& side) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[k]]), __builtin_bit_cast(bf16x8, b[PB[k]]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                side(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f32x16 z5[3], z3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) { z5[k][e] = 0.f; z3[k][e] = 0.f; }
        u32x4 rawA[2 * NF5], rawB[2 * NF5];
        u32x4 afc[3];                                        // A fragment of k-step 0 of the coming tile
        HC h0;
        issue(0, rawA);
        issue(1, rawB);
#pragma unroll
        for (int q = 0; q < 8; ++q) hc_raw(rawA, 0, q, h0, afc);
        // z pass with the operand roles swapped: A = the x pass's result (its accumulator layout IS an A fragment: lane = x', k = z
        // along the registers), B = T  ->  C[x'][z']: lane = z', registers = x' in runs of four consecutive (r & 3)  ->  16-byte
        // stores, 4 per tile and sigma instead of 16 four-byte ones (the address unit takes ~16 cycles per store instruction
        // whatever its width: 32 of them per tile and wave were 2,000 cycles of every CU's 4 waves)
        const int zi = n;                                    // this lane's output plane inside a tile
        auto step = [&](int t, f32x16& z5_t, f32x16& z3_t, f32x16& z5_p, f32x16& z3_p, f32x16& z5_pp, f32x16& z3_pp,
                        u32x4 (&raw)[2 * NF5], u32x4 (&rawn)[2 * NF5]) {
            // ---- x pass of tile t (rows z = G0 + 32 t + (lane & 31)); on entry afc = cut k-step 0
            // (no branch inside a step: a lone wave issues one instruction per 4 cycles, an MFMA leaves room for six - scalar ones
            // included; what the last steps need not do - loads past the volume, the cut of a tile that does not exist - they do
            // on clamped or stale data that nothing reads)
            f32x16 x5, x3;
#pragma unroll
            for (int e = 0; e < 16; ++e) { x5[e] = 0.f; x3[e] = 0.f; }
            u32x4 afn[3], b5[2][3], b3[2][3];
            HC hn, h30, h50, h31, h51;
#pragma unroll
            for (int s = 0; s < NF5; ++s) {
                // x5 += af[s] T5[s]     behind the MFMAs: six half-cuts of k-step s + 1 - or, in the last group, the whole cut of the
                //                       first half of x3 (complete since the group before)
                grp(x5, afc, T5r[s], [&](int k) {
                    if (s + 1 < NF5) hc_raw(raw, s + 1, k, hn, afn);
                    else { hc_acc(x3, 0, k, h30, b3[0]); if (k < 2) hc_acc(x3, 0, 6 + k, h30, b3[0]); }
                });
                if (s < NF3) {
                    // x3 += af[s] T3[s]  behind: the last two half-cuts of k-step s + 1
                    grp(x3, afc, T3r[s], [&](int k) { if (k < 2) hc_raw(raw, s + 1, 6 + k, hn, afn); });
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afc[pl] = afn[pl];
            }
            // `raw` is free (every k-step of tile t has been cut): tile t + 2 goes into it now and has a whole step to land - tile
            // t + 1 was issued a step ago into `rawn` (with ONE wave per SIMD a load that is waited for is pure dead time: issued in
            // the z pass of the step that ends with its first cut, the loads cost ~15 us of the kernel)
            issue(min(t + 2, NTX - 1), raw);
            __builtin_amdgcn_sched_barrier(0);
            // ---- z pass: the two k-steps of this tile go into output tiles t (steps 0, 1), t - 1 (2, 3), t - 2 (4: wide only)
#pragma unroll
            for (int e = 0; e < 16; ++e) { z5_t[e] = 0.f; z3_t[e] = 0.f; }
            const int j = t - 2;
            const bool st_ok = j >= 0 && j < NJ;             // (uniform)
            const bool z_ok = p.bz + 32 * j + zi < D - p.bz;
            // this lane's plane and row, columns x0 + 4 h ..; the tile's 32 j planes ride in the scalar offset, the run's 8 (r >> 2)
            // columns in the immediate
            const unsigned vbase = (st_ok && z_ok) ? 4u * (unsigned)(((p.bz + zi) * H + y) * W + x0 + 4 * h) : 0x80000000u;
            const unsigned soff = st_ok ? zstride * (unsigned)(32 * j) : 0u;
            auto store4 = [&](const f32x16& acc, const __amdgpu_buffer_rsrc_t& rs, int gq) {
                const u32x4 v = {__float_as_uint(acc[4 * gq]), __float_as_uint(acc[4 * gq + 1]), __float_as_uint(acc[4 * gq + 2]),
                                 __float_as_uint(acc[4 * gq + 3])};
                const unsigned off = (x_full || x0 + 8 * gq + 4 * h < W) ? vbase + 32u * (unsigned)gq : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, (int)soff, 0);
            };
            // z3_t += x3[0] T3[0]     behind: six half-cuts of the first half of x5 (complete with the last x group)
            grp(z3_t, b3[0], T3r[0], [&](int k) { hc_acc(x5, 0, k, h50, b5[0]); });
            // z3_p += x3[0] T3[2]     behind: the rest of that cut, four half-cuts of the second half of x3
            grp(z3_p, b3[0], T3r[2], [&](int k) { if (k < 2) hc_acc(x5, 0, 6 + k, h50, b5[0]); else hc_acc(x3, 1, k - 2, h31, b3[1]); });
            // z5_t += x5[0] T5[0]     behind: the rest of x3's second half, two half-cuts of x5's
            grp(z5_t, b5[0], T5r[0], [&](int k) { if (k < 4) hc_acc(x3, 1, 4 + k, h31, b3[1]); else hc_acc(x5, 1, k - 4, h51, b5[1]); });
            // z5_p += x5[0] T5[2]     behind: the rest of x5's second half
            grp(z5_p, b5[0], T5r[2], [&](int k) { hc_acc(x5, 1, 2 + k, h51, b5[1]); });
            // z5_pp += x5[0] T5[4]    behind: the narrow half of the finished tile t - 2 leaves (it was complete a step ago)
            grp(z5_pp, b5[0], T5r[4], [&](int k) { if (k >= 2) store4(z3_pp, g1rs, k - 2); });
            // z3_t += x3[1] T3[1]
            grp(z3_t, b3[1], T3r[1], [&](int) {});
            // z3_p += x3[1] T3[3]     behind: the wide half of tile t - 2 leaves
            grp(z3_p, b3[1], T3r[3], [&](int k) { if (k >= 2) store4(z5_pp, g2rs, k - 2); });
            // z5_t += x5[1] T5[1]     behind: two half-cuts of k-step 0 of the next tile (loaded during the previous step)
            grp(z5_t, b5[1], T5r[1], [&](int k) { if (k >= 4) hc_raw(rawn, 0, k - 4, h0, afc); });
            // z5_p += x5[1] T5[3]     behind: the rest of that cut
            grp(z5_p, b5[1], T5r[3], [&](int k) { hc_raw(rawn, 0, 2 + k, h0, afc); });
        };
        for (int t0 = 0; t0 < NTX; t0 += 6) {              // six steps per turn: accumulator roles (3) and load buffers (2) are static
            step(t0, z5[0], z3[0], z5[2], z3[2], z5[1], z3[1], rawA, rawB);
            if (t0 + 1 < NTX) step(t0 + 1, z5[1], z3[1], z5[0], z3[0], z5[2], z3[2], rawB, rawA);
            if (t0 + 2 < NTX) step(t0 + 2, z5[2], z3[2], z5[1], z3[1], z5[0], z3[0], rawA, rawB);
            if (t0 + 3 < NTX) step(t0 + 3, z5[0], z3[0], z5[2], z3[2], z5[1], z3[1], rawB, rawA);
            if (t0 + 4 < NTX) step(t0 + 4, z5[1], z3[1], z5[0], z3[0], z5[2], z3[2], rawA, rawB);
            if (t0 + 5 < NTX) step(t0 + 5, z5[2], z3[2], z5[1], z3[1], z5[0], z3[0], rawB, rawA);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// y + DoG + NMS
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float m_from_lower(float v) {        // lane l gets lane l - 1's value (lane 0: 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float m_from_upper(float v) {        // lane l gets lane l + 1's value (lane 63: 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float m_other_half(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float m_mx3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int RA, int RB>
__global__ __launch_bounds__(MT, 1) void dogm_y_kernel(DogfParams p, MTaps<RA> wa, MTaps<RB> wb) {
    __shared__ MToep L;
    __shared__ uint2 ring_all[4][MRING];
    __shared__ double s_acc[4][3][64];
    __shared__ double s_st[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    m_build_toeplitz<RA, RB>(L, wa, wb, tid);
    uint2* ring = ring_all[wv];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_acc[wv][k][lane] = 0.0;
    const int H = p.H, W = p.W;
    const int Gy = p.by - MOFF;                            // row of k-step 0, slot position 0
    const int NJ = (H - 2 * p.by + 31) / 32;               // output tiles along y
    const int NTY = NJ + 2;
    const unsigned plane_bytes = 4u * (unsigned)(H * W);
    for (long g = (long)blockIdx.x * 4 + wv; g < (long)p.n_seg; g += (long)gridDim.x * 4) {
        const int strip = (int)(g % p.n_strips), z = p.bz + (int)(g / p.n_strips);
        const int x = p.bx - 1 + MY_OWN * strip + n;
        const bool live_col = x >= p.bx && x < W - p.bx;
        const bool owned = live_col && n >= 1 && n <= MY_OWN;
        const unsigned plane = (unsigned)z * (unsigned)(H * W);
        // a descriptor per plane: rows past the image read as zeros (their k-steps feed masked outputs only)
        const __amdgpu_buffer_rsrc_t r1 = m_rsrc(p.g1 + plane, plane_bytes), r2 = m_rsrc(p.g2 + plane, plane_bytes);
        const int voff = 4 * min(x, W - 1) + 16 * W * h;   // this lane's column, and its half's four-row offset
        uint2* seg_base = p.cands + (size_t)g * p.seg_cap;
        unsigned cnt = 0, flushed = 0;
        auto flush64 = [&](unsigned n_valid) {             // entries [flushed, flushed + 64) of the ring leave; lane < n_valid hold one
            const uint2 e = ring[(flushed + lane) & (MRING - 1)];
            if ((unsigned)lane < n_valid) {
                if (flushed + lane < p.seg_cap) seg_base[flushed + lane] = e;
                const double dv = (double)__uint_as_float(e.x);
                s_acc[wv][0][lane] += 1.0; s_acc[wv][1][lane] += dv; s_acc[wv][2][lane] += dv * dv;
            }
        };
        auto emit = [&](bool is, float out, unsigned oidx) {
            const unsigned long long mask = __ballot(is);
            if (mask) {
                if (is) {
                    const unsigned pos = cnt + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                    ring[pos & (MRING - 1)] = make_uint2(__float_as_uint(out), oidx);
                }
                cnt += (unsigned)__popcll(mask);
            }
        };
        float raw1[16], raw2[16];
        const int xoff = 4 * min(x, W - 1);
        auto issue = [&](int t) {                          // rows Gy + 32 t + 16 sp + pi(h, j)
            const int rbase = Gy + 32 * t;
            if (rbase + 32 <= p.yhi) {                     // (uniform) every row was written by the first kernel
                // the row in the SCALAR offset (not range-checked: it stays inside the plane), no vector address arithmetic
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned soff = 4u * (unsigned)W * (unsigned)(rbase + 16 * sp + (j & 3) + 8 * (j >> 2));
                        raw2[8 * sp + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, voff, (int)soff, 0));
                        raw1[8 * sp + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, voff, (int)soff, 0));
                    }
            } else {
                // the last tiles reach past the rows the first kernel wrote (stale workspace there: a NaN times a zero tap
                // would still poison a live output): those rows read as zeros through the range check of the vector offset
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = rbase + 16 * sp + (j & 3) + 8 * (j >> 2) + 4 * h;
                        const unsigned off = row < p.yhi ? (unsigned)xoff + 4u * (unsigned)(row * W) : 0x80000000u;
                        raw2[8 * sp + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, (int)off, 0, 0));
                        raw1[8 * sp + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, (int)off, 0, 0));
                    }
            }
        };
        f32x16 y5[3], y3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) { y5[k][e] = 0.f; y3[k][e] = 0.f; }
        // state across tiles: the x-pooled last row of the previous tile as seen from the other lane half (the row above
        // row 0), and the deferred row 31 (it needs row 0 of the next tile): its value and max(xm 30, xm 31), in the
        // upper lane half
        float carry_up = 0.f, pend_c = 0.f, pend_up = 0.f;
        int pend_row = -1;                                 // (uniform) image row of the deferred row, -1: none
        issue(0);
        auto step = [&](int t, f32x16& y5_t, f32x16& y3_t, f32x16& y5_p, f32x16& y3_p, f32x16& y5_pp, f32x16& y3_pp) {
            bf16x8 b2[2][3], b1[2][3];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw2[8 * sp + j];
                m_cut8(v, b2[sp]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw1[8 * sp + j];
                m_cut8(v, b1[sp]);
            }
            if (t + 1 < NTY) issue(t + 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) { y5_t[e] = 0.f; y3_t[e] = 0.f; }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 ta[3];
                m_toep(L, sp, lane, ta);           y5_t = m_mfma6(ta, b2[sp], y5_t);
                m_toep(L, NF5 + sp, lane, ta);     y3_t = m_mfma6(ta, b1[sp], y3_t);
                m_toep(L, 2 + sp, lane, ta);       y5_p = m_mfma6(ta, b2[sp], y5_p);
                m_toep(L, NF5 + 2 + sp, lane, ta); y3_p = m_mfma6(ta, b1[sp], y3_p);
                if (sp == 0) { m_toep(L, 4, lane, ta); y5_pp = m_mfma6(ta, b2[sp], y5_pp); }
            }
            const int j = t - 2;
            if (j < 0 || j >= NJ) return;                  // (uniform)
            // ---- output tile j: rows yr(r) = by + 32 j + pi(h, r)
            const int ybase = p.by + 32 * j;
            float dog[16], xm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int yr = ybase + (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool live = live_col && yr < H - p.by;
                dog[r] = live ? y5_pp[r] - y3_pp[r] : 0.f;                      // zeroed border (inside the image)
                xm[r] = m_mx3(m_from_lower(dog[r]), dog[r], m_from_upper(dog[r]));
            }
            float ptop[4], pbot[4];                        // the other half's rows 4 g + 3 / 4 g of each group of eight rows
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) { ptop[gq] = m_other_half(xm[4 * gq + 3]); pbot[gq] = m_other_half(xm[4 * gq]); }
            // the deferred row of the previous tile (image row ybase - 1, upper half): its lower neighbour is row 0 of this tile
            if (pend_row >= 0) {                           // (uniform)
                const float hm = fmaxf(pend_up, pbot[0]);
                const float out = (hm == pend_c) ? pend_c : 0.f;
                const unsigned oidx = plane + (unsigned)(pend_row * W + x);
                const bool mine = h == 1 && owned;
                if (p.nms_out && mine) p.nms_out[oidx] = out;
                emit(mine && out > 0.f, out, oidx);
            }
            const float up_first = carry_up;               // row ybase - 1 as the lower half sees it
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gq = r >> 2, rr = r & 3;
                const int yr = ybase + rr + 8 * gq + 4 * h;
                float up, down;
                if (rr > 0) up = xm[r - 1];
                else up = h ? ptop[gq] : (gq ? ptop[gq - 1] : up_first);
                if (rr < 3) down = xm[r + 1];
                else down = h ? (gq < 3 ? pbot[gq + 1] : 0.f) : pbot[gq];
                const float hm = m_mx3(up, xm[r], down);
                const float out = (hm == dog[r]) ? dog[r] : 0.f;
                const bool deferred = (r == 15) && h == 1;                      // row 31: decided with the next tile
                const bool row_ok = yr < H - p.by;
                const unsigned oidx = plane + (unsigned)(yr * W + x);
                const bool mine = owned && row_ok && !deferred;
                if (p.nms_out && mine) p.nms_out[oidx] = out;
                emit(mine && out > 0.f, out, oidx);
                if ((r & 3) == 3) {                        // (at most 63 + 4 x 64 entries wait in the 512-entry ring)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        if (cnt - flushed >= 64u) { flush64(64u); flushed += 64; }
                }
            }
            pend_c = dog[15];
            pend_up = fmaxf(xm[14], xm[15]);
            pend_row = (ybase + 31 < H - p.by) ? ybase + 31 : -1;
            carry_up = m_other_half(xm[15]);
        };
        for (int t0 = 0; t0 < NTY; t0 += 3) {
            step(t0, y5[0], y3[0], y5[2], y3[2], y5[1], y3[1]);
            if (t0 + 1 < NTY) step(t0 + 1, y5[1], y3[1], y5[0], y3[0], y5[2], y3[2]);
            if (t0 + 2 < NTY) step(t0 + 2, y5[2], y3[2], y5[1], y3[1], y5[0], y3[0]);
        }
        if (pend_row >= 0) {                               // the last live row ends a tile: the row below it is the zeroed border
            const float hm = fmaxf(pend_up, 0.f);
            const float out = (hm == pend_c) ? pend_c : 0.f;
            const unsigned oidx = plane + (unsigned)(pend_row * W + x);
            const bool mine = h == 1 && owned;
            if (p.nms_out && mine) p.nms_out[oidx] = out;
            emit(mine && out > 0.f, out, oidx);
        }
        while (cnt - flushed >= 64u) { flush64(64u); flushed += 64; }
        if (cnt > flushed) flush64(cnt - flushed);
        if (lane == 0) {
            p.seg_count[g] = min(cnt, p.seg_cap);
            if (cnt > p.seg_cap) atomicOr(p.overflow, 1u);
        }
    }
    // statistics: one (count, sum, sum of squares) per workgroup, waves added in order (deterministic)
    const double a = wave_sum(s_acc[wv][0][lane]), s = wave_sum(s_acc[wv][1][lane]), ss = wave_sum(s_acc[wv][2][lane]);
    if (lane == 0) { s_st[wv][0] = a; s_st[wv][1] = s; s_st[wv][2] = ss; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v += s_st[k][threadIdx.x];
        p.stats[3 * (size_t)blockIdx.x + threadIdx.x] = v;
    }
}

inline int dogm_radius(float sigma) { return (int)(4.0f * sigma + 0.5f); }

template <int R>
void fill_mtaps(float sigma, MTaps<R>& t) {
    const int rs = dogm_radius(sigma);
    double tmp[R + 1], sum = 0;
    const double c = -0.5 / ((double)sigma * (double)sigma);
    for (int d = 0; d <= R; ++d) {
        tmp[d] = d <= rs ? exp(c * (double)d * (double)d) : 0.0;
        sum += d == 0 ? tmp[d] : 2.0 * tmp[d];
    }
    for (int d = 0; d <= R; ++d) t.w[d] = (float)(tmp[d] / sum);
}

}  // namespace

// The matrix-core chain takes what the fused vector chain takes (two sigmas with radii <= 12 / <= 20, a 3x3 window, rows of
// 64..512 voxels in multiples of 4) with a zeroed xy border of at least 20 voxels (a window starts 20 before its outputs and
// never reflects in x / y) and planes deep enough for ONE z reflection of the 80-wide window.
bool mi_dogm_usable(const float* rec, const float* g1, const float* g2, const float* nms_out, int D, int H, int W,
                    float s1, float s2, int k, int bz, int bxy) {
    // Opt-in (MI_DOGM=1) as of round 5: results identical to the vector chain's to 1e-7, picks equal - but 0.63 ms against
    // 0.48 ms on 256x512x512 (profiles/r05_experiments.txt): the x + z kernel is bound by its address-unit time (32
    // four-byte stores per tile and wave: 16 cycles each whatever the width) and by vector work bunched behind a third of
    // its MFMAs, the y kernel by 64 four-byte loads per tile.  What is left to do is listed in DESIGN.md.
    if (!getenv("MI_DOGM")) return false;
    auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const int r1 = dogm_radius(s1), r2 = dogm_radius(s2);
    if ((size_t)D * H * W >= ((size_t)1 << 29)) return false;          // 32-bit byte offsets inside the volume
    if (D < 2 * MOFF + 4 || D - 2 * bz < 1) return false;
    return k == 3 && s1 <= s2 && (W & 3) == 0 && W >= 64 && W <= 512 && r1 >= 1 && r1 <= 12 && r2 >= r1 && r2 <= 20 &&
           bxy >= MOFF && 2 * bxy < H && 2 * bxy < W && bz >= 0 && bz <= MOFF && al(rec) && al(g1) && al(g2) && al(nms_out);
}

int mi_dogm_own() { return MY_OWN; }

DogfGrid mi_dogm_grid(int D, int H, int W, int bz, int bxy) {
    DogfGrid g = {};
    if (2 * bxy >= H || 2 * bxy >= W || 2 * bz >= D) return g;
    const int rows = H - 2 * bxy, planes = D - 2 * bz;
    g.n_strips = mi_cdiv(W - 2 * bxy, MY_OWN);
    g.ychunk = rows;                                       // one chunk: a wave marches the whole live height of its strip
    g.n_ychunks = 1;
    g.n_seg = (unsigned)((long)planes * g.n_strips);
    g.n_wg = (unsigned)std::min<long>(mi_cdiv(g.n_seg, 4), 256);        // persistent: one workgroup per CU
    g.seg_cap = (unsigned)(g.ychunk * 8 + 64);             // xy-NMS survivors: at most one per 2x2 patch of 30 columns
    return g;
}

int mi_launch_dogm(DogfParams p, const DogfGrid& g, float s1, float s2, hipStream_t st) {
    p.vol_bytes = 4u * (unsigned)((size_t)p.D * p.H * p.W);
    p.ychunk = g.ychunk; p.n_ychunks = 1; p.n_strips = g.n_strips; p.n_seg = g.n_seg; p.n_wg = g.n_wg;
    p.seg_cap = g.seg_cap;
    p.ylo = std::max(p.by - MOFF, 0); p.yhi = std::min(p.H - p.by + MOFF, p.H);
    DogmXZ q = {};
    q.rec = p.rec; q.g1 = p.g1; q.g2 = p.g2; q.D = p.D; q.H = p.H; q.W = p.W; q.bz = p.bz;
    q.ylo = p.ylo; q.yhi = p.yhi;
    q.xa = p.bx & ~3;                                       // 16-byte aligned windows (xa - 20 is a multiple of 4)
    q.n_xstrips = mi_cdiv(p.W - p.bx - q.xa, 32);
    q.n_quads = mi_cdiv(q.n_xstrips, 4);
    q.rows_per_iter = (256 / (8 * q.n_quads)) * 8;
    q.n_iter = mi_cdiv(q.yhi - q.ylo, q.rows_per_iter);
    q.vol_bytes = p.vol_bytes;
    q.clr[0] = p.clr[0]; q.clr[1] = p.clr[1]; q.clr_n[0] = p.clr_n[0]; q.clr_n[1] = p.clr_n[1];
    MTaps<12> wa;
    MTaps<20> wb;
    fill_mtaps<12>(s1, wa);
    fill_mtaps<20>(s2, wb);
    hipLaunchKernelGGL((dogm_xz_kernel<12, 20>), dim3(256), dim3(MT), 0, st, q, wa, wb);
    MI_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL((dogm_y_kernel<12, 20>), dim3(p.n_wg), dim3(MT), 0, st, p, wa, wb);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
