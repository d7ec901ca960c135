// Forward convolutions with a SHORT reduction at inference (round 4): the 1 x 1 convolutions of the detector's U-Net (the last layer
// 32 -> 32, the transposed convolutions' 1 x 1 products to 4 Co columns) and the (3, 1, 1) head (models/networks/unet.py:319-399,
// unet_small.py:30-97).  K = taps x Ci is 32 - 256: the implicit GEMM's tile pipeline (row decode, tap cursor, two staging sets, LDS
// images, a prologue and an epilogue per 64-row tile) runs ONE to EIGHT slices per tile and spends its time outside them - 16 - 100
// TFLOP/s on layers that only have to stream their activations (profiles/r04_unet_layers.txt).
// Here nothing is staged: rows of the activation tensor are contiguous along K, so a wave loads the A fragments of its 32 rows straight
// from memory (two 16-byte loads per lane and k-step), cuts them in registers (bf16x3: three planes, six products, f32 accumulate - the
// arithmetic of every other convolution here) and multiplies them with B fragments that come, 16 bytes per lane, from a weight IMAGE
// in fragment order (smallk_prep_kernel, built once per set of weights by the caller).  A workgroup = four waves = 128 rows x (32 NB)
// columns; bias and ReLU in the epilogue.  (3, 1, 1): tap t reads the row one z-plane below / at / above - zeros outside the volume.
#include "common.h"
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_sk(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// exact three-way bf16 cut of 8 f32 (truncation, as conv_igemm.hip / conv_direct3.hip / conv_cube2.hip)
__device__ __forceinline__ void cut8k(const float (&v)[8], bf16x8 (&o)[3]) {
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

// weight image: [k-step][32-column block][plane 3][lane 64] x 16 bytes; lane (j = lane & 31, h = lane >> 5) holds W[16 ks + 8 h + e][32 cb + j]
__global__ __launch_bounds__(256) void smallk_prep_kernel(const float* w, unsigned char* img, int K, int Co) {
    const int ncb = Co >> 5, KS = K >> 4;
    const long total = (long)KS * ncb * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long q = i >> 6;
        const int cb = (int)(q % ncb), ks = (int)(q / ncb);
        const int j = lane & 31, h = lane >> 5;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[(long)(16 * ks + 8 * h + e) * Co + 32 * cb + j];
        bf16x8 o[3];
        cut8k(v, o);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<u32x4*>(img + ((q * 3 + pl) * 64 + lane) * 16) = __builtin_bit_cast(u32x4, o[pl]);
    }
}

struct SmallKParams {
    const float* x;           // (M, Ci) rows
    const unsigned char* img;
    const float* bias;        // may be null
    float* y;                 // (M, Co)
    long M;
    int Ci, Co, ntaps, relu;
    long plane;               // (3, 1, 1): rows per z-plane (H * W); tap t reads row m + (t - 1) * plane
    int D;                    //            planes per sample
    unsigned x_bytes, img_bytes;
};

template <int NB>             // 32-column blocks per wave
__global__ __launch_bounds__(256) void smallk_fwd_kernel(SmallKParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const long m0 = (long)blockIdx.x * 128 + wave * 32;
    const int cb0 = blockIdx.y * NB, ncb = p.Co >> 5;
    const __amdgpu_buffer_rsrc_t xrs = rsrc_sk(p.x, p.x_bytes), irs = rsrc_sk(p.img, p.img_bytes);
    const long m = m0 + l32;
    const bool row_ok = m < p.M;
    const int cpt = p.Ci >> 4, KS = p.ntaps * cpt;         // k-steps per tap, in all
    // (3, 1, 1): this row's z inside its sample decides which taps exist
    int z = 0;
    if (p.ntaps == 3) z = (int)((m / p.plane) % p.D);
    auto a_off = [&](int ks) -> unsigned {
        const int t = ks / cpt, c0 = (ks - t * cpt) * 16 + 8 * h;
        bool ok = row_ok;
        long r = m;
        if (p.ntaps == 3) { const int zz = z + t - 1; ok = ok && zz >= 0 && zz < p.D; r = m + (long)(t - 1) * p.plane; }
        return ok ? 4u * (unsigned)(r * p.Ci + c0) : 0x80000000u;
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    u32x4 araw[2][2], braw[2][NB][3];
    auto fetch = [&](int ks, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const unsigned ao = ks < KS ? a_off(ks) : 0x80000000u;
        araw[SET][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)ao, 0, 0);
        araw[SET][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ao + 16u), 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned bo = (ks < KS && cb0 + j < ncb) ? (unsigned)((((long)ks * ncb + cb0 + j) * 3 + pl) * 64 + lane) * 16u : 0x80000000u;
                braw[SET][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(irs, (int)bo, 0, 0);
            }
    };
    auto step = [&](auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(araw[SET][0][e]); v[4 + e] = __uint_as_float(araw[SET][1][e]); }
        bf16x8 af[3];
        cut8k(v, af);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bf16x8 bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[pl] = __builtin_bit_cast(bf16x8, braw[SET][j][pl]);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bf[PB[pr]], acc[j], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    fetch(0, S0{});
    for (int ks = 0; ks < KS; ks += 2) {
        fetch(ks + 1, S1{});
        step(S0{});
        fetch(ks + 2, S0{});
        if (ks + 1 < KS) step(S1{});
    }
    // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = 32 * (cb0 + j) + l32;
        if (cb0 + j >= ncb) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < p.M) {
                float t = acc[j][r] + bv;
                if (p.relu) t = fmaxf(t, 0.f);
                p.y[row * p.Co + col] = t;
            }
        }
    }
}

// ---- the detector's two heads over the feature volume in ONE pass (round 5; unet_small.py:86-97 + models/utils.py: `proj` =
// normalize(Conv3d(C, 32, (3,1,1))(v)), `hm` = Conv3d(C, K <= 4, (3,1,1))(v)) ------------------------------------------------------------
// smallk_fwd_kernel<1> for the 32 `proj` columns with (a) the L2 normalisation of a row in the epilogue - a row's 32 columns sit in the
// 32 lanes of a half-wave: four DPP butterflies and one swizzle per row, no second pass over the 1 GB volume - and (b) the `hm` outputs
// as a by-product of the A fragments: a lane holds 8 of its row's K values per k-step, so K_hm dot products cost 8 K_hm FMAs per k-step
// on the vector unit (f32, as mi_zhead_fwd), the two k-halves of a row meet by one swizzle - no third pass over the volume either.
struct SmallKHeadParams {
    SmallKParams s;
    const float* w_hm;        // [3][Ci][K_hm] (HipZHead's storage)
    float* y_hm;              // (M, K_hm)
    int k_hm;
};

__device__ __forceinline__ float half_sum32(float v) {          // sum over the 32 lanes of this lane's half-wave, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));                      // lane ^ 16
    return v;
}

template <int KH>
__global__ __launch_bounds__(256) void smallk_head_kernel(SmallKHeadParams hp) {
    const SmallKParams& p = hp.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const long m0 = (long)blockIdx.x * 128 + wave * 32;
    const __amdgpu_buffer_rsrc_t xrs = rsrc_sk(p.x, p.x_bytes), irs = rsrc_sk(p.img, p.img_bytes);
    const long m = m0 + l32;
    const bool row_ok = m < p.M;
    const int cpt = p.Ci >> 4, KS = 3 * cpt;
    const int z = (int)((m / p.plane) % p.D);
    auto a_off = [&](int ks) -> unsigned {
        const int t = ks / cpt, c0 = (ks - t * cpt) * 16 + 8 * h;
        const int zz = z + t - 1;
        const bool ok = row_ok && zz >= 0 && zz < p.D;
        return ok ? 4u * (unsigned)((m + (long)(t - 1) * p.plane) * p.Ci + c0) : 0x80000000u;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float hm[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) hm[k] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    u32x4 araw[2][2], braw[2][3];
    auto fetch = [&](int ks, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        const unsigned ao = ks < KS ? a_off(ks) : 0x80000000u;
        araw[SET][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)ao, 0, 0);
        araw[SET][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ao + 16u), 0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const unsigned bo = ks < KS ? (unsigned)(((long)ks * 3 + pl) * 64 + lane) * 16u : 0x80000000u;
            braw[SET][pl] = __builtin_amdgcn_raw_buffer_load_b128(irs, (int)bo, 0, 0);
        }
    };
    auto step = [&](int ks, auto SETc) {
        constexpr int SET = decltype(SETc)::value;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(araw[SET][0][e]); v[4 + e] = __uint_as_float(araw[SET][1][e]); }
        // hm: this lane's 8 K values of its row against w_hm[tap][c0 + e][k] (rows outside the volume were loaded as zeros)
        const float* wr = hp.w_hm + (long)(ks / cpt) * p.Ci * KH + (long)((ks % cpt) * 16 + 8 * h) * KH;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < KH; ++k) hm[k] = fmaf(v[e], wr[e * KH + k], hm[k]);
        bf16x8 af[3], bf[3];
        cut8k(v, af);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[pl] = __builtin_bit_cast(bf16x8, braw[SET][pl]);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]], bf[PB[pr]], acc, 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    fetch(0, S0{});
    for (int ks = 0; ks < KS; ks += 2) {
        fetch(ks + 1, S1{});
        step(ks, S0{});
        fetch(ks + 2, S0{});
        if (ks + 1 < KS) step(ks + 1, S1{});
    }
    // hm: the two k-halves of row l32 sit in lanes l32 and l32 + 32
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        const float o = __shfl_xor(hm[k], 32, 64);
        if (h == 0 && row_ok) hp.y_hm[m * KH + k] = hm[k] + o;
    }
    // proj: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h: normalise every row over its 32 columns (F.normalize: eps 1e-12)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float t = acc[r];
        const float ss = half_sum32(t * t);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        if (row < p.M) p.y[row * 32 + l32] = t * inv;
    }
}

}  // namespace

/* The detector's heads in one pass over the feature volume x (M = N D H W rows of Ci channels; `plane` = H W, D planes per sample):
 *   y_proj (M, 32) = normalize(conv(3,1,1)(x; img))       img: mi_smallk_prep of the (3, 1, 1) weights [3][Ci][32] (K = 3 Ci)
 *   y_hm   (M, K)  = conv(3,1,1)(x; w_hm), K <= 4         w_hm: [3][Ci][K] f32 (no bias), f32 FMAs as mi_zhead_fwd
 * (unet_small.py:86-97: `proj` + F.normalize, `hm`).  Ci a multiple of 16 with 3 Ci <= 512; inference only. */
extern "C" int mi_smallk_heads_fwd_f32(const float* x, const void* img, float* y_proj, const float* w_hm, float* y_hm, int k_hm, long M,
                                       int Ci, long plane, int D, mi_stream_t stream) {
    if (!x || !img || !y_proj || !w_hm || !y_hm || M <= 0 || k_hm < 1 || k_hm > 4 || !mi_smallk_image_bytes(3 * Ci, 32)) return MI_E_ARG;
    if (plane <= 0 || D <= 0 || M % (plane * D) != 0 || 3 * Ci > 512) return MI_E_ARG;
    if (4l * M * Ci >= 0x7fff0000l || (M + 127) / 128 > 0x7fffffffl) return MI_E_UNSUPPORTED;
    SmallKHeadParams hp = {};
    hp.s = SmallKParams{x, (const unsigned char*)img, nullptr, y_proj, M, Ci, 32, 3, 0, plane, D, (unsigned)(4l * M * Ci),
                        (unsigned)mi_smallk_image_bytes(3 * Ci, 32)};
    hp.w_hm = w_hm; hp.y_hm = y_hm; hp.k_hm = k_hm;
    const dim3 grid((unsigned)((M + 127) / 128));
    hipStream_t s = (hipStream_t)stream;
    if (k_hm == 1) hipLaunchKernelGGL(smallk_head_kernel<1>, grid, dim3(256), 0, s, hp);
    else if (k_hm == 2) hipLaunchKernelGGL(smallk_head_kernel<2>, grid, dim3(256), 0, s, hp);
    else if (k_hm == 3) hipLaunchKernelGGL(smallk_head_kernel<3>, grid, dim3(256), 0, s, hp);
    else hipLaunchKernelGGL(smallk_head_kernel<4>, grid, dim3(256), 0, s, hp);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}

/* Forward convolution with a short reduction, inference path (no gradient form): 1 x 1 (ntaps = 1) or (3, 1, 1) with padding (1, 0, 0)
 * (ntaps = 3: `plane` = H * W rows per z-plane, `D` planes per sample) over channels-last rows x (M, Ci) -> y (M, Co) = act(x . W + bias).
 * Ci a multiple of 16, Co a multiple of 32, the operands below 2 GiB.  `img`: mi_smallk_image_bytes(ntaps * Ci, Co) bytes written by
 * mi_smallk_prep from the kernel-layout weights [tap][Ci][Co] (= a (ntaps * Ci, Co) matrix) - once per set of weights. */
extern "C" size_t mi_smallk_image_bytes(int K, int Co) {
    if (K <= 0 || Co <= 0 || (K & 15) || (Co & 31)) return 0;
    return (size_t)(K / 16) * (Co / 32) * 3 * 64 * 16;
}
extern "C" int mi_smallk_prep(const float* w, void* img, int K, int Co, mi_stream_t stream) {
    if (!w || !img || !mi_smallk_image_bytes(K, Co)) return MI_E_ARG;
    const long total = (long)(K / 16) * (Co / 32) * 64;
    hipLaunchKernelGGL(smallk_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned char*)img, K, Co);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
extern "C" int mi_smallk_fwd_f32(const float* x, const void* img, const float* bias, float* y, int relu, long M, int Ci, int Co, int ntaps,
                                 long plane, int D, mi_stream_t stream) {
    if (!x || !img || !y || M <= 0 || (ntaps != 1 && ntaps != 3) || !mi_smallk_image_bytes(ntaps * Ci, Co)) return MI_E_ARG;
    if (ntaps == 3 && (plane <= 0 || D <= 0 || M % (plane * D) != 0)) return MI_E_ARG;
    if (4l * M * Ci >= 0x7fff0000l || (M + 127) / 128 > 0x7fffffffl) return MI_E_UNSUPPORTED;
    SmallKParams p = {x, (const unsigned char*)img, bias, y, M, Ci, Co, ntaps, relu, plane, D, (unsigned)(4l * M * Ci),
                      (unsigned)mi_smallk_image_bytes(ntaps * Ci, Co)};
    const int ncb = Co / 32;
    const int nb = ncb % 4 == 0 ? 4 : (ncb % 2 == 0 ? 2 : 1);
    const dim3 grid((unsigned)((M + 127) / 128), (unsigned)(ncb / nb));
    if (nb == 4) hipLaunchKernelGGL(smallk_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (nb == 2) hipLaunchKernelGGL(smallk_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(smallk_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    MI_RETURN_IF_LAUNCH_FAILED();
    return MI_OK;
}
