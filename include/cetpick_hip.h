/*
 * cetpick_hip.h - C-ABI of the MI355X (gfx950) hot path for nextpyp/cet_pick (MiLoPYP).
 *
 * The reference is pure Python on PyTorch and has no FFI of its own; its
 * boundary for this path is the Python API.  The Python mirror in cet_pick_amd/ keeps that API
 * (same names / arguments) and reaches the device only through the entry points declared here,
 * via ctypes.  Each entry point cites the reference code it replaces (paths relative to the
 * reference root, cet_pick/...).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors); the caller keeps it
 *    alive until the stream has passed the call;
 *  - every function enqueues work on `stream` and returns without synchronising the device;
 *  - return 0 = ok, negative = argument error (MI_E_*), positive = hipError_t;
 *  - no function allocates device memory: scratch comes from the caller, sized by the matching
 *    *_workspace_bytes() query;  launches are hipGraph-capturable;
 *  - volumes are (D,H,W) row-major fp32 ("Z,H,W" in the reference's in-memory order);
 *    activations of the training path are channels-last (N,D,H,W,C) fp32.
 */
#ifndef CETPICK_HIP_H
#define CETPICK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi_stream_t; /* hipStream_t */

#define MI_OK 0
#define MI_E_ARG (-1)       /* bad shape / null pointer / unsupported window */
#define MI_E_WORKSPACE (-2) /* workspace too small */
#define MI_E_UNSUPPORTED (-3)

/* Library identity: returns the ABI version (bumped on any signature change). */
int mi_abi_version(void);
/* measurement aid: nodes of a captured hipGraph (hipGraph_t) by type: counts[4] = kernel, memcpy, memset, other */
int mi_graph_node_counts(void* graph, int* counts);
/* measurement aid: a one-thread launch that writes the 100 MHz wall clock into *slot (device uint64) */
int mi_debug_stamp(void* slot, mi_stream_t stream);
/* measurement aid: the kernel family the last mi_conv* call of the calling thread dispatched to (tools/bench_conv.py) */
const char* mi_debug_last_conv_kernel(void);
/* gfx target the code objects were built for, e.g. "gfx950". */
const char* mi_build_arch(void);

/* ------------------------------------------------------------------------------------------
 * Inference path
 * ------------------------------------------------------------------------------------------ */

/* models/utils.py:167-169  `_sigmoid`: x <- sigmoid(x) IN PLACE; y <- clamp(x, 1e-4, 1-1e-4).
 * y may alias x. */
int mi_sigmoid_clamp(float* x, float* y, size_t n, mi_stream_t stream);

/* models/decode.py:11-33 `_nms_xy` (1,k,k) / `_nms_z` (k,1,1) / `_nms` (3,k,k) and
 * utils/image.py:97-105 `_nms` (k,k,k):  out = heat * (max_pool3d(heat, (kd,kh,kh), stride 1,
 * pad (k-1)/2, -inf padding) == heat).   kd, kh odd, in {1,3,5,7}.  out must not alias heat. */
int mi_nms3d(const float* heat, float* out, int D, int H, int W, int kd, int kh,
             mi_stream_t stream);

/* Fused detector decode = `_sigmoid` + `tomo_decode` (models/decode.py:123-155, reg=None) for one
 * (1,1,D,H,W) volume:
 *   heat_out = clamp(sigmoid(logits))            (if apply_sigmoid; heat_out may be NULL)
 *   nms      = `_nms` window (3,k,k)             (fiber != 0: `_nms_xy` (1,k,k) then `_nms_z` (k,1,1))
 *   top-K of nms by (score desc, flat index asc) -> dets[K][5] = {x+.25, y+.25, z, score, score}
 * x,y,z follow `_convert_1d_to_3d` (decode.py:35-41).  Rows past the number of positive local
 * maxima are {0.25,0.25,0,0,0} (torch.topk leaves their order unspecified).
 * n_valid_out (device int32, may be NULL) receives min(K, #positive maxima).
 * logits must not alias heat_out.
 * apply_sigmoid: bit 0 = apply `_sigmoid`; bit 1 = "the workspace header is clean": the caller zeroed it once with
 * mi_decode_workspace_init and has since used the workspace only through calls that passed this bit - such a call
 * skips its own clearing pass and leaves the header clean again (one launch less per decode).  Without bit 1 the
 * workspace needs no initialisation. */
size_t mi_decode_workspace_bytes(int D, int H, int W, int K);
int mi_decode_workspace_init(void* workspace, size_t workspace_bytes, mi_stream_t stream);
int mi_sigmoid_nms_topk(const float* logits, float* heat_out, int D, int H, int W, int k,
                        int fiber, int apply_sigmoid, int K, float* dets, int32_t* n_valid_out,
                        void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* scipy.ndimage.gaussian_filter as called by utils/image.py:152-156 and utils/loader.py:102
 * (mode='reflect', truncate=4 -> radius=int(4*sigma+.5)), separable, fp32 storage,
 * fp32 accumulation.  `tmp` is a scratch volume of D*H*W floats; out may alias in. */
int mi_gauss3d_sep(const float* in, float* out, float* tmp, int D, int H, int W, float sigma,
                   mi_stream_t stream);

/* Per-slice 2-D variant (axes 1 and 2 only): `gaussian_filter(sli, sigma)` in the tilt-series branch of
 * utils/loader.py:94-96. */
int mi_gauss2d_slices(const float* in, float* out, float* tmp, int D, int H, int W, float sigma,
                      mi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tomogram loading: utils/loader.py `load_rec` :27-88, `quantize` :16-25,
 * `preprocess` :90-121.  Values are evaluated in fp64 registers and stored as fp32.
 * ------------------------------------------------------------------------------------------ */
enum { MI_ORDER_XYZ = 0, MI_ORDER_XZY = 1, MI_ORDER_YXZ = 2, MI_ORDER_ZXY = 3 };   /* `--order` */

/* Axis reorder of an MRC data block `src` of shape (d0,d1,d2) (= mrcfile's .data.shape), element type by
 * MRC mode (0 int8, 1 int16, 2 float32, 6 uint16), into dst (Z',X,Y) fp32 as load_rec does
 * (loader.py:32-36,:62), with `compress`: Z' = ceil(Z/2), slice j = max(slice 2j, slice 2j+1)
 * (loader.py:45-47,:70-71; zxy + compress + odd Z is the reference's IndexError -> MI_E_ARG). */
int mi_rec_reorder(const void* src, int mrc_mode, int d0, int d1, int d2, int order, int compress,
                   float* dst, mi_stream_t stream);
/* stats[s] = {mean, std (ddof 0), min, max} of slice s of x [n_slices][slice_elems] (n_slices = 1: the whole
 * volume, loader.py:59,:87; per slice: the is_tilt branches :48-49,:97). */
size_t mi_vol_stats_workspace_bytes(long n_slices, long slice_elems);
int mi_vol_stats(const float* x, long n_slices, long slice_elems, double* stats, void* ws, size_t ws_bytes,
                 mi_stream_t stream);
/* y = (x - mean) / std per slice (y may alias x). */
int mi_zscore(const float* x, float* y, long n_slices, long slice_elems, const double* stats,
              mi_stream_t stream);
/* preprocess (loader.py:103-106,:118-120): z-score -> q = round(clip(255 (z - mi)/(ma - mi), 0, 255)) ->
 * (q - min q)/(max q - min q).  flat_zero != 0: a constant slice gives 0 (cv2.normalize, :98,:114) instead
 * of NaN.  y may alias x. */
int mi_zscore_quantize_minmax(const float* x, float* y, long n_slices, long slice_elems, const double* stats,
                              double mi, double ma, int flat_zero, mi_stream_t stream);

/* utils/image.py:138-183 `get_potential_coords_pyramid`: DoG pyramid -> border zero (z: border_z
 * slices each end, x/y: 30, or 60 when H>512 and W>512) -> `_nms_xy`(k) -> max over levels ->
 * cutoff = mean(pos)+0.5*std(pos) -> greedy 3-D NMS (`non_maximum_suppression_3d`, distance d).
 * Outputs (device): scores[max_out] fp32, coords[max_out][3] int32 as (x,y,z), n_out int32,
 * cutoff_out fp32 (may be NULL).  Picks are emitted in greedy order (score descending).
 * heat_out (may be NULL) receives the dense NMS'd DoG map (D*H*W). */
size_t mi_dog_pick_workspace_bytes(int D, int H, int W, int n_sigmas);
int mi_dog_pick(const float* rec, int D, int H, int W, const float* sigmas_host, int n_sigmas,
                int k, int border_z, int nms_d, float* heat_out, float* scores, int32_t* coords,
                int32_t* n_out, int max_out, float* cutoff_out, void* workspace,
                size_t workspace_bytes, mi_stream_t stream);

/* models/decode.py:42-79 == utils/image.py:42-79 `non_maximum_suppression_3d` on a dense (D,H,W)
 * volume: voxels with value > threshold are visited in (value desc, flat index desc) order; a
 * visited, unsuppressed voxel is emitted and suppresses every flat offset of the radius
 * scale*d/2 ball (flat offsets, no bounds check - wraps exactly like the reference). */
size_t mi_greedy_nms3d_workspace_bytes(int D, int H, int W);
int mi_greedy_nms3d(const float* vol, int D, int H, int W, float d, float scale, float threshold,
                    float* scores, int32_t* coords, int32_t* n_out, int max_out, void* workspace,
                    size_t workspace_bytes, mi_stream_t stream);


/* Sub-tomogram crops, one per centre (x,y,z int32); window along an axis is
 * [c - s/2, c - s/2 + s) as the reference slices it; out-of-volume voxels repeat the edge.
 *   mode 0: raw crop (n, cz, cy, cx)         datasets/tomo_pre_proj_angle_select_new3d_vol.py:130-138
 *   mode 1: sum over z, min-max -> (n, cy, cx)                                          ...:117-128
 *   mode 2: z-normalised 3-D crop (mean 0, unbiased std 1) -> (n, cz, cy, cx)   (MoCo-3D input, §8d C2)
 *   mode 3: ZNormalization -> RescaleIntensity(-3, 3) -> ZNormalization of the crop -> (n, cz, cy, cx)
 *           (datasets/tomo_pre.py:57-60; the `Crop` of that chain is the window the centre and size describe)
 * flip_x mirrors the crop along x (second contrastive view). */
int mi_crop_normalize(const float* vol, int D, int H, int W, const int32_t* centres_xyz, int n,
                      int cz, int cy, int cx, int mode, int flip_x, float* out, mi_stream_t stream);

/* The same crops for a batch whose bookkeeping already lives on the device (replaces the per-batch host work of the
 * reference's DataLoader path, datasets/particle_pre_3d_vol.py:70-85 under moco_main.py:122-130): crop i of the launch
 * is dataset sample s = order[first + i] (order == NULL: s = first + i); it is cut from tomogram vols[owner[s]]
 * (owner == NULL: vols[0]) around centres_xyz[s] (+ shift_xyz[s] when shift_xyz != NULL).  All arrays are device
 * memory.  Modes 0..2 of mi_crop_normalize. */
typedef struct mi_vol_desc {
    const float* vol;
    int32_t D, H, W, reserved;
} mi_vol_desc;
int mi_crop_normalize_table(const mi_vol_desc* vols, const int32_t* owner, const int32_t* centres_xyz,
                            const int32_t* shift_xyz, const int64_t* order, int64_t first, int n, int cz, int cy,
                            int cx, int mode, int flip_x, float* out, mi_stream_t stream);
/* simsiam_test_hm_3d.py:45-51 on min-max'ed crops: y = (floor(255 x) / 255 - mean) / std  (ToPILImage -> ToTensor ->
 * Normalize; mean / std = the dataset statistics of tomo_pre_proj_angle_select_new3d_vol.py:238-239).  y may alias x. */
int mi_u8_roundtrip_normalize(const float* x, float* y, size_t n, float mean, float std, mi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training path: channels-last fp32 activations (N,D,H,W,C),
 * weights [tap][Cin][Cout] with tap = (kd*k + kh)*k + kw.
 * ------------------------------------------------------------------------------------------ */

/* nn.Conv3d (cubic kernel k, stride, pad, no bias) / nn.Linear (k=1, D=H=W=1) as implicit GEMM
 * on the matrix cores: f32 operands, f32 accumulation.  The environment variable MI_CONV_ARITH selects how the f32
 * products are formed: "bf16x3" (default) - every operand element is cut exactly into three bf16 values and the six
 * products of weight <= 2 are accumulated by v_mfma_f32_32x32x16_bf16 (f32-equivalent: the dropped terms are below one
 * f32 rounding; non-finite inputs give NaN where an f32 multiply would give Inf) - or "f32" - v_mfma_f32_32x32x2_f32,
 * bit-for-bit an fmaf chain.  Replaces models/networks/moco_encoder_3d.py:40-84 (conv3x3x3,
 * BasicBlock), :163-169 (7x7x7 stem, Ci == 1), :183, :189, :198-205 (feature conv, fc, proj).
 * Ci % 16 == 0 (or Ci == 1), Co % 16 == 0.
 *   fwd:   y  = act(conv(x, w) + res)                 res may be NULL, relu 0/1
 *   dgrad: dx = (conv_transpose(dy, w) + res) * (mask > 0)     res, mask may be NULL
 *   wgrad: dw = sum over output voxels
 * `ws` holds split-K slabs (mi_conv3d_workspace_bytes covers all three); a NULL / short ws
 * silently selects the unsplit schedule (same values up to fp32 summation order). */
size_t mi_conv3d_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride,
                                 int pad);
int mi_conv3d_fwd_f32(const float* x, const float* w, float* y, const float* res, int relu, int N,
                      int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, void* ws,
                      size_t ws_bytes, mi_stream_t stream);
int mi_conv3d_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                        const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co, int k,
                        int stride, int pad, void* ws, size_t ws_bytes, mi_stream_t stream);
int mi_conv3d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int Di, int Hi, int Wi,
                        int Ci, int Co, int k, int stride, int pad, void* ws, size_t ws_bytes,
                        mi_stream_t stream);

/* nn.Linear (+ bias) followed by training-mode nn.BatchNorm1d (+ ReLU) in one launch (the projection MLP,
 * models/networks/moco_encoder_3d.py:238-255): xlin (M, Co) = x W + bias (kept for the backward), y = act(bn(xlin)),
 * save = mean[Co], invstd[Co]; running statistics and num_batches_tracked (int64) are updated when given.  Arithmetic of
 * mi_linear_fwd_f32 followed by mi_bn_small_fwd.  MI_E_UNSUPPORTED for M > 64 or MI_CONV_ARITH=f32: run those two. */
int mi_linear_bn_fwd_f32(const float* x, const float* w, const float* bias, float* xlin, float* y, int M, int Ci, int Co,
                         const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* save_mean_invstd, int relu,
                         mi_stream_t stream);

/* The SyncBatchNorm form of the same fusion (moco_main.py:65-66 converts the encoders): xlin = x W + bias and this rank's column
 * sums of xlin and xlin^2 (2 Co doubles, as mi_bn_stats) from the product's epilogue; the all-reduce of `sums` and
 * mi_bn_apply_fwd follow.  M <= 64 rows. */
int mi_linear_stats_fwd_f32(const float* x, const float* w, const float* bias, float* xlin, double* sums, int M, int Ci, int Co,
                            mi_stream_t stream);

/* conv1 + the batch statistics of bn1 in one pass (models/networks/moco_encoder_3d.py:170-176, 326-328): the 7^3
 * stride-2 single-channel stem convolution, with sums[0..Co) = column sums of y and sums[Co..2Co) = column sums of y^2
 * (device doubles; what mi_bn_stats(y) would produce) taken from the output tiles while they are in registers.
 * MI_E_UNSUPPORTED where the stem kernel does not apply (shape, Co != 64, MI_CONV_ARITH=f32): run mi_conv3d_fwd_f32 and
 * mi_bn_stats instead.  Workspace: mi_conv3d_stem_stats_workspace_bytes (0 = unsupported shape). */
size_t mi_conv3d_stem_stats_workspace_bytes(int N, int Di, int Hi, int Wi, int Co);
int mi_conv3d_stem_stats_f32(const float* x, const float* w, float* y, int N, int Di, int Hi, int Wi, int Co,
                             double* sums, void* ws, size_t ws_bytes, mi_stream_t stream);

/* nn.Linear (models/networks/moco_encoder_3d.py:183-236: fc and the projection head): y[M][Co] = x[M][Ci] . W + bias with
 * W in kernel layout [Ci][Co]; bias (Co values, may be NULL) is added in the epilogue of the 1x1x1 convolution launch.
 * Workspace as mi_conv3d_workspace_bytes(M, 1, 1, 1, Ci, Co, 1, 1, 0). */
int mi_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int M, int Ci, int Co, void* ws,
                      size_t ws_bytes, mi_stream_t stream);

/* Same kernels with a per-axis window (kd,kh,kw) and zero padding (pd,ph,pw); weights
 * [kd][kh][kw][Cin][Cout].  nn.Conv2d of the 2-D encoder (models/networks/simsiam_model_2d.py:25-28,
 * 473-502, 617-661) is the D = 1, kd = 1, pd = 0 case on (N,1,H,W,C) activations. */
size_t mi_convnd_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                 int stride, int pd, int ph, int pw);
int mi_convnd_fwd_f32(const float* x, const float* w, float* y, const float* res, int relu, int N,
                      int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd,
                      int ph, int pw, void* ws, size_t ws_bytes, mi_stream_t stream);
/* y = act(conv(x, w) + bias[Co]): mi_convnd_fwd_f32 with the residual read as one row of Co values (a bias).  Replaces the
 * reference's conv -> BatchNorm(eval) -> ReLU triple at inference (models/networks/unet.py:198-249,319-399) once the caller has
 * folded the BatchNorm's scale into w and its shift into bias. */
int mi_convnd_fwd_bias_f32(const float* x, const float* w, float* y, const float* bias, int relu, int N, int Di,
                           int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph,
                           int pw, void* ws, size_t ws_bytes, mi_stream_t stream);
/* y = act(conv2d(x, w, 7 x 7, stride 2, padding 3) + bias): ONE input channel, 16 output channels, x (N, H, W), w in kernel layout
 * [7][7][1][16], bias[16] or NULL, y (N, Ho, Wo, 16) channels-last with Ho = (H - 1) / 2 + 1 (models/networks/unet_small.py:35: the
 * detector's first layer, with its evaluation-mode BatchNorm folded in by the caller). */
int mi_stem2d_fwd_bias_f32(const float* x, const float* w, const float* bias, float* y, int relu, int N, int H, int W,
                           mi_stream_t stream);
/* Forward convolutions with a short reduction, inference path (round 4; no gradient form - training keeps mi_convnd_*): 1 x 1
 * (ntaps 1) or (3, 1, 1) with padding (1, 0, 0) (ntaps 3: `plane` = H * W rows per z-plane, `D` planes per sample) on channels-last rows
 * x (M, Ci) -> y (M, Co) = act(x . W + bias), bias may be NULL.  Ci a multiple of 16, Co of 32.  `img` (mi_smallk_image_bytes(ntaps * Ci, Co)
 * bytes) is written by mi_smallk_prep from the kernel-layout weights [tap][Ci][Co] once per set of weights.  Replaces the implicit GEMM
 * for the detector's 1 x 1 layers and (3, 1, 1) head (models/networks/unet.py:319-399,880-886, unet_small.py:86-97). */
size_t mi_smallk_image_bytes(int K, int Co);
int mi_smallk_prep(const float* w, void* img, int K, int Co, mi_stream_t stream);
int mi_smallk_fwd_f32(const float* x, const void* img, const float* bias, float* y, int relu, long M, int Ci, int Co,
                      int ntaps, long plane, int D, mi_stream_t stream);
/* Round 5: the detector's two heads in ONE pass over the feature volume (unet_small.py:86-97: `proj` = F.normalize(Conv3d(C, 32,
 * (3,1,1), padding (1,0,0))(v)), `hm` = Conv3d(C, K <= 4, (3,1,1))(v)): y_proj (M, 32) normalised in the epilogue, y_hm (M, K) as a
 * by-product of the fragments the product loads (f32 FMAs, as mi_zhead_fwd).  img: mi_smallk_prep(w_proj, img, 3 Ci, 32);
 * w_hm: [3][Ci][K] f32.  M = N D plane rows.  Inference only. */
int mi_smallk_heads_fwd_f32(const float* x, const void* img, float* y_proj, const float* w_hm, float* y_hm, int k_hm, long M, int Ci,
                            long plane, int D, mi_stream_t stream);
int mi_convnd_dgrad_f32(const float* dy, const float* w, float* dx, const float* res,
                        const float* mask, int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh,
                        int kw, int stride, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                        mi_stream_t stream);
int mi_convnd_wgrad_f32(const float* x, const float* dy, float* dw, int N, int Di, int Hi, int Wi,
                        int Ci, int Co, int kd, int kh, int kw, int stride, int pd, int ph, int pw,
                        void* ws, size_t ws_bytes, mi_stream_t stream);
/* The weight gradient with its split-K reduction left to the caller: *splits_out = 1 -> dw is final; > 1 -> `ws` holds
 * that many slabs of Co*Ci*kd*kh*kw floats (slab s at ws + s * that many floats) and dw is untouched.
 * mi_splitk_reduce_batch sums the slabs of n such launches (outs[i] <- sum of n_slabs[i] slabs at slabs[i], out_elems[i]
 * floats each, a multiple of 4) in one launch per 24 of them: one reduce for the weight gradients of a whole backward
 * pass.  The pointer / count arrays are HOST arrays (read during the call). */
int mi_convnd_wgrad_slabs_f32(const float* x, const float* dy, float* dw, int N, int Di, int Hi, int Wi, int Ci, int Co,
                              int kd, int kh, int kw, int stride, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                              int* splits_out, mi_stream_t stream);
int mi_splitk_reduce_batch(const void* const* slabs, void* const* outs, const int* n_slabs, const long* out_elems, int n,
                           mi_stream_t stream);
/* Round 5: nb (2..4) weight gradients of ONE geometry in ONE launch - the weight gradients of a residual stage exist together
 * once loss.backward() (trains/base_trainer.py:497) has left the stage (moco_encoder_3d.py:55-84: layer1's four, layer2's and
 * layer3's three equal convolutions), and a launch's fixed costs are a third of each when they run one by one.  Problem i reads
 * xs[i] / dys[i] and leaves *splits_out slabs in wss[i] (each workspace ws_bytes long; > 1: mi_splitk_reduce_batch sums them into
 * dws[i]) or, *splits_out == 1, the final gradient in dws[i].  The pointer arrays are HOST arrays.  MI_E_UNSUPPORTED: no batched
 * kernel for this geometry / nb (or MI_NO_WGRAD_BATCH=1) - issue nb single calls. */
int mi_convnd_wgrad_slabs_batch_f32(const float* const* xs, const float* const* dys, float* const* dws, void* const* wss, int nb,
                                    int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int stride, int pd,
                                    int ph, int pw, size_t ws_bytes, int* splits_out, mi_stream_t stream);

/* Patch-resident direct kernels for the 3^3 / stride 1 / padding 1 convolutions of the MoCo-3D encoder's residual layers
 * (models/networks/moco_encoder_3d.py:55-84,170-171), forward and data gradient, bf16x3 arithmetic:
 *   channels 64 : nn.Conv3d(64, 64, 3, 1, 1) on (N, D, 8, 8, 64), D even                  (layer1)
 *   channels 128: nn.Conv3d(128, 128, 3, 1, 1) on (N, 4, 4, 4, 128)                        (layer2)
 * mi_conv3d_fwd_f32 / mi_conv3d_dgrad_f32 (and the mi_convnd_* forms) take them by themselves for such shapes and build
 * the weight image in `ws` on every call (MI_CONV_NO_DIRECT=1 keeps the implicit GEMM).  A caller that knows when the
 * weights change keeps the images instead: mi_conv3d_direct_prep cuts n weight tensors ([tap][Cin][Cout] f32, channels[i]
 * = 64 or 128) into n images of mi_conv3d_direct_wimg_bytes(channels[i]) bytes in one launch per channel count (dgrad[i]
 * != 0: the transposed, tap-flipped image the data gradient needs; the four arrays are HOST arrays), and
 * mi_conv3d_direct_f32 runs
 *   forward  (image with dgrad = 0): out = act(conv(a, w) + res)                       a = x,  relu as given, mask NULL
 *   dgrad    (image with dgrad = 1): out = (conv_transpose(a, w) + res) * (mask > 0)   a = dy, relu 0
 * `ws`: mi_conv3d_direct_workspace_bytes(N, channels) bytes (0 since round 4: both direct kernels are final in one launch; rounds 2-3 kept split-K slabs of the 128-channel kernel there).
 * MI_E_UNSUPPORTED for any other shape (mi_conv3d_direct_usable: 0 / 1 (64 channels) / 2 (128 channels)). */
size_t mi_conv3d_direct_wimg_bytes(int channels);
size_t mi_conv3d_direct_workspace_bytes(int N, int channels);
int mi_conv3d_direct_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad);
int mi_conv3d_direct_prep(const void* const* w, void* const* img, const int* dgrad, const int* channels, int n,
                          mi_stream_t stream);
int mi_conv3d_direct_f32(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu,
                         int N, int Di, int Hi, int Wi, int channels, void* ws, size_t ws_bytes, mi_stream_t stream);

/* Round 6: patch-resident direct kernel for the 3 x 3 / stride 1 / padding 1 convolutions of the SimSiam 2-D encoder's BasicBlocks
 * (models/networks/simsiam_model_2d.py:473-502; TomoResClassifier2D.forward :776-819) at --bbox 36 (docs/explore.md:67), forward and
 * data gradient, bf16x3 arithmetic: C -> C channels (64 / 128 / 256) on (N, H, W, C) channels-last planes (csrc/conv_p2d.hip: the batch
 * tiled as ONE flat run of voxels, 128 per workgroup, a zero row between planes).
 *   mi_conv2d_p2d_usable: 1 = a compile-time instance ((W, C) = (36, 64), (18, 128), (9, 256), H >= W: tap offsets are immediates), 2 = the
 *     generic instance (any H, W up to ~64 whose patch fits: run-time geometry - the default --bbox 32 and every other), 0 = neither
 *     (MI_NO_P2D=1: always 0; MI_NO_P2D_GENERIC=1: never 2);  mi_conv2d_p2d_wimg_bytes(C): bytes of one weight image;
 *   mi_conv2d_p2d_prep: n images in one launch from (3, 3, C_i, C_i) kernel-layout weights (dgrad[i] != 0: the transposed,
 *     tap-flipped image of the data gradient; the four arrays are HOST arrays);
 *   mi_conv2d_p2d_f32: out = act(conv(a; image) + res) * (mask > 0); res / mask may be NULL (forward: a = x; dgrad: a = dy, relu 0). */
int mi_conv2d_p2d_usable(int N, int H, int W, int C);
size_t mi_conv2d_p2d_wimg_bytes(int C);
int mi_conv2d_p2d_prep(const void* const* w, void* const* img, const int* dgrad, const int* channels, int n, mi_stream_t stream);
int mi_conv2d_p2d_f32(const float* a, const void* wimg, float* out, const float* res, const float* mask, int relu, int N, int H, int W,
                      int C, mi_stream_t stream);
/* ... and their weight gradient (_workspace_bytes = 0: this shape has none - windows of more than 192 rows, W > ~40): dw (3, 3, C, C) kernel layout = sum over the batch's voxels of x[o + tap] (x) dy[o], written (not
 * accumulated); both operands staged voxel-major and read through the transposing LDS read, X in a padded flat order in which a tap is one
 * row offset; split-K slabs in ws (mi_conv2d_p2d_wgrad_workspace_bytes), added in slab order.  MI_NO_P2D_WGRAD=1: MI_E_UNSUPPORTED. */
size_t mi_conv2d_p2d_wgrad_workspace_bytes(int N, int H, int W, int C);
int mi_conv2d_p2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int C, void* ws, size_t ws_bytes,
                            mi_stream_t stream);

/* The 2-D encoder's first layer, Conv2d(1, Co, 3, padding=1) (models/networks/simsiam_model_2d.py:634): nine taps of one input channel -
 * HBM-bound f32 FMA chains, not matrix work.  x (N, H, W), w (3, 3, 1, Co) kernel layout, y / dy (N, H, W, Co); Co a multiple of 4, <= 64
 * (else MI_E_UNSUPPORTED).  The weight gradient sums per-block partials in block order (deterministic); ws: _workspace_bytes(Co). */
size_t mi_conv2d_stem3_workspace_bytes(int Co);
int mi_conv2d_stem3_fwd_f32(const float* x, const float* w, float* y, int N, int H, int W, int Co, mi_stream_t stream);
int mi_conv2d_stem3_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int Co, void* ws, size_t ws_bytes,
                              mi_stream_t stream);

/* 3^3 / stride 1 / padding 1 convolutions on 2 x 2 x 2 volumes, C -> C channels, C = 128 / 256 / 512 (layer3 and feature_3d
 * of the MoCo-3D encoder, models/networks/moco_encoder_3d.py:55-84,172,178), bf16x3 arithmetic, FINAL IN ONE LAUNCH (round 4:
 * the last of the four reduction quarters of an output tile to arrive sums them - in a fixed order - and applies the epilogue):
 *   dgrad = 0: out = act(conv(a, w) + res)                      a = x (N, 2, 2, 2, C),  w = [27][Cin][Cout] f32, mask NULL
 *   dgrad = 1: out = (conv_transpose(a, w) + res) * (mask > 0)  a = dy, relu 0
 * mi_conv3d_fwd_f32 / mi_conv3d_dgrad_f32 take such shapes too (same kernel, partial sums + a reduce launch: their `ws` has no
 * state).  `ws` here: mi_conv3d_cube2_workspace_bytes(N, C) bytes OWNED BY ONE STREAM whose first 16 KiB (arrival counters) are
 * ZERO before the first call; every completed call leaves them zero.  MI_E_UNSUPPORTED for other shapes
 * (mi_conv3d_cube2_usable). */
size_t mi_conv3d_cube2_workspace_bytes(int N, int C);
int mi_conv3d_cube2_usable(int N, int Di, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad);
int mi_conv3d_cube2_f32(const float* a, const float* w, float* out, const float* res, const float* mask, int relu, int dgrad,
                        int N, int C, void* ws, size_t ws_bytes, mi_stream_t stream);

/* Dilated windows, stride 1 (kernel (3,3,3), dilation (1,4,4), padding (1,4,4): the 3-D head of the detector
 * network, models/networks/unet_small.py:38-41).  Same contract as mi_convnd_*; output extent per axis
 * = in + 2*pad - dil*(k-1). */
size_t mi_convnd_dil_workspace_bytes(int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw,
                                     int pd, int ph, int pw, int dd, int dh, int dw);
int mi_convnd_dil_fwd_f32(const float* x, const float* w, float* y, const float* res, int relu, int N, int Di,
                          int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int pd, int ph, int pw,
                          int dd, int dh, int dw, void* ws, size_t ws_bytes, mi_stream_t stream);

/* Patch-resident direct convolution to 32 output channels, forward / inference (conv_d32.hip): the detector's
 * Conv2d(32|64, 32, 3, padding=1) layers with their BatchNorm folded into weights + bias
 * (models/networks/unet.py:198-249) and the Conv3d(32, 32, 3, padding=(1,4,4), dilation=(1,4,4)) + ReLU of the
 * feature head (models/networks/unet_small.py:52-60).
 *   mi_conv_d32_kind: 1 = 3x3 on every plane (kd = 1, dilation 1, Ci 32 or 64, H and W multiples of 16); 2 = 3x3x3 with
 *     dilation (1,4,4) (Ci 32, H and W multiples of 32, D even); 0 = not taken.  stride 1, padding = dilation * (k - 1) / 2.
 *   mi_conv_d32_prep: w in kernel layout [tap][Ci][32] -> the pre-cut bf16x3 image (mi_conv_d32_image_bytes(Ci, 9 | 27)).
 *   mi_conv_d32_fwd_f32: y = act(conv(x) + bias); x (N, D, H, W, Ci), y (N, D, H, W, 32) channels-last; bias may be NULL. */
int mi_conv_d32_kind(int N, int D, int H, int W, int Ci, int Co, int kd, int kh, int kw, int dd, int dh, int dw);
size_t mi_conv_d32_image_bytes(int Ci, int ntap);
int mi_conv_d32_prep(const float* w, void* img, int Ci, int ntap, mi_stream_t stream);
/* Round 5: the same kernel to 64 output channels (mi_conv_d32_kind returns 3: 2-D 3 x 3, Ci 32 / 64 / 128, H and W multiples of 16 -
 * the 128 x 128 level of the U-Net, unet.py:198-249): image [chunk][tap][column half][plane][lane]; forward through
 * mi_conv_d32_fwd_f32(..., kind = 3) with y (N, D, H, W, 64).  MI_NO_D64=1 keeps the implicit GEMM. */
size_t mi_conv_d64_image_bytes(int Ci, int ntap);
int mi_conv_d64_prep(const float* w, void* img, int Ci, int ntap, mi_stream_t stream);
/* Co = 64, 128 or 256 (kind 3 too): one image per 64-column block - img holds (Co / 64) x mi_conv_d64_image_bytes(Ci, 9) bytes -,
 * a workgroup per tile and block; y (N, D, H, W, Co).  MI_NO_D64_WIDE=1 keeps Co > 64 on the implicit GEMM. */
int mi_conv_d64_prep_co(const float* w, void* img, int Ci, int Co, int ntap, mi_stream_t stream);
/* mi_conv_d32_kind = 4: 1 x 1 products (the 2 x 2 transposed convolutions' 4 Co columns, unet.py:251-317, and the last 1 x 1 layer):
 * y = act(x W + bias) per voxel, the tile resident in LDS.  Co = 32 (image: mi_conv_d32_prep(w, img, Ci, 1)) or a multiple of 64
 * up to 512 (mi_conv_d64_prep_co(w, img, Ci, Co, 1)); Ci 32 / 64 / 128 / 256; H % 8 == 0, W % 16 == 0.  MI_NO_D32_1X1=1: off. */
int mi_conv_d32_1x1_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                            int Ci, int Co, mi_stream_t stream);
/* The 2 x 2 / stride-2 transposed convolution of an up-convolution block at inference (unet.py:251-317,319-399) in ONE launch: the
 * 1 x 1 product to 4 Co columns with the pixel shuffle, scale / shift (evaluation-mode BatchNorm, bias folded in) and ReLU in its
 * epilogue, written into out (N, Ho, Wo, cstride)[..., 0:Co] - the first Co channels of the concatenation, whose channels Co.. the
 * caller fills with the encoder feature.  wimg: mi_conv_d64_prep_co(w, img, Ci, 4 Co, 1).  Ci 32 / 64 / 128, Co a multiple of 32 with
 * 64 <= 4 Co <= 512, H % 8 == 0, W % 16 == 0; MI_E_UNSUPPORTED otherwise (then: the product + mi_upconv_tail_fwd). */
/* mi_conv_d32_kind 1 / 3 (2-D 3 x 3) with the 2 x 2 max-pool of the result as a SECOND output (unet.py:198-249: conv -> BatchNorm ->
 * ReLU -> MaxPool2d(2, ceil_mode) - the un-pooled tensor is the skip connection): y (N, D, H, W, Co), y_pool (N, D, H / 2, W / 2, Co);
 * H, W multiples of 16; images as for mi_conv_d32_fwd_f32 / mi_conv_d64_fwd_f32.  A lane's accumulators of a row block are a 4 x 4
 * patch of one channel: the windows never leave the lane. */
int mi_conv_d32_fwd_pool_f32(const float* x, const void* wimg, const float* bias, float* y, float* y_pool, int relu, int N, int D, int H,
                             int W, int Ci, int Co, mi_stream_t stream);
/* ... with y a channel slice of a wider tensor (voxel v's Co channels at y + v * y_cstride): the skip connection written straight into
 * the concatenation buffer of the up-convolution block that consumes it (y = cat + Co_up, y_cstride = Co_up + Co), whose first Co_up
 * channels mi_conv_d32_upconv_fwd_f32 writes later - no concatenation pass at all.  y_pool stays dense. */
int mi_conv_d32_fwd_pool_strided_f32(const float* x, const void* wimg, const float* bias, float* y, int y_cstride, float* y_pool, int relu,
                                     int N, int D, int H, int W, int Ci, int Co, mi_stream_t stream);
/* dst[m][c0 : c0 + Cs] = src[m][:] over the M rows of a (M, Ct) tensor (channel counts and c0 multiples of 4): the encoder feature
 * into the concatenation buffer of an up-convolution block (torch.cat((up, enc), 1), unet.py:392) behind mi_conv_d32_upconv_fwd_f32. */
int mi_copy_channels_into(const float* src, int Cs, float* dst, int Ct, int c0, long M, mi_stream_t stream);
int mi_conv_d32_upconv_fwd_f32(const float* x, const void* wimg, const float* scale, const float* shift, float* out, int N, int H,
                               int W, int Ci, int Co, int Ho, int Wo, int cstride, mi_stream_t stream);
int mi_conv_d64_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W, int Ci,
                        int Co, mi_stream_t stream);
int mi_conv_d32_fwd_f32(const float* x, const void* wimg, const float* bias, float* y, int relu, int N, int D, int H, int W,
                        int Ci, int kind, mi_stream_t stream);
int mi_convnd_dil_dgrad_f32(const float* dy, const float* w, float* dx, const float* res, const float* mask,
                            int N, int Di, int Hi, int Wi, int Ci, int Co, int kd, int kh, int kw, int pd,
                            int ph, int pw, int dd, int dh, int dw, void* ws, size_t ws_bytes,
                            mi_stream_t stream);
int mi_convnd_dil_wgrad_f32(const float* x, const float* dy, float* dw_out, int N, int Di, int Hi, int Wi,
                            int Ci, int Co, int kd, int kh, int kw, int pd, int ph, int pw, int dd, int dh,
                            int dw, void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Detector network glue (models/networks/unet.py, unet_small.py), channels-last.
 * ------------------------------------------------------------------------------------------ */
/* nn.MaxPool2d(k, ceil_mode=True) per image (unet.py:231-233): Ho = ceil(Hi/k); argmax (tap in the window,
 * may be NULL) feeds the backward. */
int mi_maxpool2d_ceil_fwd(const float* x, float* y, uint8_t* argmax, int N, int Hi, int Wi, int C, int k,
                          mi_stream_t stream);
int mi_maxpool2d_ceil_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int Hi, int Wi, int C,
                          int k, mi_stream_t stream);
/* nn.ConvTranspose2d(Ci, Co, 2, stride=2) (unet.py:155-160) = 1x1 conv to 4*Co columns [(a*2+b)*Co + co]
 * (mi_convnd_fwd_f32) + this shuffle: y[n][2h+a][2w+b][co] = t[n][h][w][(a*2+b)*Co+co] + bias[co], cropped to
 * (Ho, Wo) <= (2H, 2W) (`autocrop`, unet.py:253-266).  bwd: the inverse scatter (zeros in the cropped rim). */
int mi_shuffle2x2_fwd(const float* t, const float* bias, float* y, int N, int H, int W, int Co, int Ho, int Wo,
                      mi_stream_t stream);
/* Inference tail of an up-convolution block in one pass (models/networks/unet.py:319-399: ConvTranspose2d(2, stride 2) -> BatchNorm2d
 * (eval) -> ReLU -> torch.cat with the encoder feature): out (N, Ho, Wo, Co + Ce) = cat(relu(scale[co] * shuffle(t) + shift[co]), enc);
 * t (N, H, W, 4*Co) is the 1 x 1 product of the transposed convolution, scale / shift the BatchNorm with the bias folded in. */
int mi_upconv_tail_fwd(const float* t, const float* scale, const float* shift, const float* enc, float* out, int N, int H,
                       int W, int Co, int Ce, int Ho, int Wo, mi_stream_t stream);
int mi_shuffle2x2_bwd(const float* dy, float* dt, int N, int H, int W, int Co, int Ho, int Wo,
                      mi_stream_t stream);
/* torch.cat((a, b), 1) (unet.py:385) over M rows, and its backward. */
int mi_concat_channels(const float* a, int Ca, const float* b, int Cb, float* out, long M, mi_stream_t stream);
int mi_split_channels(const float* dout, float* da, int Ca, float* db, int Cb, long M, mi_stream_t stream);
/* nn.Conv3d(C, K, (3,1,1), padding=(1,0,0), bias=False) with K <= 4 outputs: the `hm` head
 * (unet_small.py:55-62).  x (N,D,P,C), w [3][C][K], y (N,D,P,K), P = H*W. */
int mi_zhead_fwd(const float* x, const float* w, float* y, int N, int D, long P, int C, int K,
                 mi_stream_t stream);
/* its backward: dx (N,D,P,C) and dw [3][C][K], either may be NULL; 256 % C == 0 */
size_t mi_zhead_bwd_workspace_bytes(int N, int D, long P, int C, int K);
int mi_zhead_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int N, int D, long P, int C,
                 int K, void* ws, size_t ws_bytes, mi_stream_t stream);

/* nn.BatchNorm3d / BatchNorm1d over rows [M][C] (moco_encoder_3d.py:170,184,199-205), split so a
 * SyncBN all-reduce of `sums` (2*C doubles: sum x, sum x^2) fits between stats and apply.
 * count = rows behind `sums` (global M under SyncBN).  save = mean[C], invstd[C].
 * gamma/beta NULL = affine=False.  running stats NULL = not tracked; num_batches_tracked (int64, may be
 * NULL) is incremented on the device by the same launch.  C % 4 == 0, 256 % (C/4) == 0.
 * y = act(bn(x) + res): res (may be NULL) is the residual branch of the 2-D BasicBlock
 * (simsiam_model_2d.py:485-502). */
size_t mi_colreduce_workspace_bytes(long M, int C);
int mi_bn_stats(const float* x, long M, int C, double* sums, void* ws, size_t ws_bytes,
                mi_stream_t stream);
int mi_bn_apply_fwd(const float* x, float* y, long M, int C, const double* sums, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* save_mean_invstd, const float* res, int relu, mi_stream_t stream);
/* The stem's BatchNorm3d + ReLU + MaxPool3d(3, 2, 1) (moco_encoder_3d.py:170-172) fused: relu(bn(x)) is never
 * written.  fwd: after mi_bn_stats (+ SyncBN all-reduce); sums == NULL = eval mode.  x (N,Di,Hi,Wi,C), pooled
 * (N,Do,Ho,Wo,C), argmax uint8.  bwd: mi_maxpool3d_bwd, then mi_bn_relu_bwd_{reduce,apply}_x. */
int mi_bn_relu_maxpool3d_fwd(const float* x, float* y, uint8_t* argmax, int N, int Di, int Hi, int Wi, int C, int k,
                             int stride, int pad, const double* sums, double count, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float* save_mean_invstd, mi_stream_t stream);
/* Backward of relu(bn(x)) without the stored activation (mask recomputed from x); dy = mi_maxpool3d_bwd's output. */
int mi_bn_relu_bwd_reduce_x(const float* dy, const float* x, long M, int C, const float* save_mean_invstd,
                            const float* gamma, const float* beta, double* sums, void* ws, size_t ws_bytes,
                            mi_stream_t stream);
int mi_bn_relu_bwd_apply_x(const float* dy, const float* x, float* dx, long M, int C, const float* save_mean_invstd,
                           const float* gamma, const float* beta, const double* sums, double count, float* dgamma,
                           float* dbeta, mi_stream_t stream);
/* Small-M BatchNorm (M <= MI_BN_SMALL_MAX_ROWS: the BatchNorm1d layers of the projection MLPs,
 * moco_encoder_3d.py:199-205, and feature_3d's BatchNorm3d) in one launch each way; same results as the split
 * calls, which remain the path under SyncBN (the all-reduce sits between their halves). */
#define MI_BN_SMALL_MAX_ROWS 4096
int mi_bn_small_fwd(const float* x, float* y, long M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* save_mean_invstd, const float* res, int relu, mi_stream_t stream);
int mi_bn_small_bwd(const float* dy, const float* x, const float* y, float* dx, long M, int C,
                    const float* save_mean_invstd, const float* gamma, int relu, float* dgamma, float* dbeta,
                    mi_stream_t stream);
/* global_avgpool(relu(bn(x))) in the launch of the small-M BatchNorm (nn.BatchNorm3d + ReLU + AdaptiveAvgPool3d(1) behind
 * feature_3d, models/networks/moco_encoder_3d.py:178-181,385-388): `pooled` (M / V, C) is the mean over the V consecutive
 * rows of each sample (V a power of two <= 64 dividing M); y = relu(bn(x)) is written as well - the backward reads its sign.
 * mi_bn_small_pool_bwd takes the gradient of `pooled`. */
int mi_bn_small_pool_fwd(const float* x, float* y, float* pooled, long M, int C, int V, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* save_mean_invstd, mi_stream_t stream);
int mi_bn_small_pool_bwd(const float* dpooled, const float* x, const float* y, float* dx, long M, int C, int V,
                         const float* save_mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                         mi_stream_t stream);
int mi_bn_eval_fwd(const float* x, float* y, long M, int C, const float* running_mean,
                   const float* running_var, const float* gamma, const float* beta, float eps,
                   float* save_mean_invstd, const float* res, int relu, mi_stream_t stream);
/* backward: sums = {sum dy', sum dy'*xhat} with dy' = dy*(y>0) when relu; then
 * dx = gamma*invstd*(dy' - sums[0]/count - xhat*sums[1]/count), dgamma = sums[1], dbeta = sums[0]. */
int mi_bn_bwd_reduce(const float* dy, const float* x, const float* y, long M, int C,
                     const float* save_mean_invstd, int relu, double* sums, void* ws,
                     size_t ws_bytes, mi_stream_t stream);
int mi_bn_bwd_apply(const float* dy, const float* x, const float* y, float* dx, long M, int C,
                    const float* save_mean_invstd, const float* gamma, const double* sums,
                    double count, int relu, float* dgamma, float* dbeta, mi_stream_t stream);

/* mi_bn_bwd_apply for y = relu(bn(x) + residual) (cet_pick/models/networks/simsiam_model_2d.py:473-502, the end of a BasicBlock): the
 * gradient behind the ReLU, dres = dy * (y > 0), is what the residual branch receives - written by the same pass (no masking launch in
 * front of it); the statistics in `sums` are mi_bn_bwd_reduce's with relu = 1. */
int mi_bn_bwd_apply_res(const float* dy, const float* x, const float* y, float* dx, float* dres, long M, int C,
                        const float* save_mean_invstd, const float* gamma, const double* sums, double count,
                        float* dgamma, float* dbeta, mi_stream_t stream);
/* dgamma = sums[C..2C), dbeta = sums[0..C) from the LOCAL (pre all-reduce) backward sums: under
 * SyncBN the affine gradients stay per-rank and are averaged with the other gradients, exactly as
 * torch.nn.SyncBatchNorm does. */
int mi_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, mi_stream_t stream);
/* column sum (bias gradient of nn.Linear, moco_encoder_3d.py:189): out[c] = sum_m dy[m][c] */
int mi_colsum(const float* dy, long M, int C, float* out, double* sums_scratch, void* ws,
              size_t ws_bytes, mi_stream_t stream);

/* nn.MaxPool3d(k, stride, pad) (moco_encoder_3d.py:172), argmax tap kept per element (uint8,
 * first maximum in (kd,kh,kw) scan order, as torch); backward routes dy to that element. */
int mi_maxpool3d_fwd(const float* x, float* y, uint8_t* argmax, int N, int Di, int Hi, int Wi,
                     int C, int k, int stride, int pad, mi_stream_t stream);
int mi_maxpool3d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int Di, int Hi,
                     int Wi, int C, int k, int stride, int pad, mi_stream_t stream);
/* nn.AdaptiveAvgPool3d(1) (moco_encoder_3d.py:188): x [B][S][C] -> y [B][C] */
int mi_avgpool_fwd(const float* x, float* y, int B, int S, int C, mi_stream_t stream);
int mi_avgpool_bwd(const float* dy, float* dx, int B, int S, int C, mi_stream_t stream);
int mi_bias_add(float* y, const float* bias, long M, int C, mi_stream_t stream);
/* dst0 <- src0 and dst1 <- src1 (n floats each, n % 4 == 0, 16-byte aligned) in one launch: the two views of a batch
 * (trains/base_trainer.py:486-491: batch['input'], batch['input_aug']) into the step engine's static input buffers. */
int mi_copy_pair_f32(float* dst0, const float* src0, float* dst1, const float* src1, long n, mi_stream_t stream);
/* out = (dy + add) * (y > 0)   (ReLU backward; add may be NULL; n % 4 == 0) */
int mi_relu_mask(const float* dy, const float* y, const float* add, float* out, long n,
                 mi_stream_t stream);

/* F.normalize(x, dim=1) and the MoCo logits (models/moco.py:111-138):
 * logits[b] = [q_b.k_b, q_b.queue] / T, queue is [C][R]. */
int mi_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int C, mi_stream_t stream);
int mi_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int C,
                  mi_stream_t stream);
int mi_moco_logits_fwd(const float* q, const float* k, const float* queue, float* logits, int B,
                       int C, int R, float T, mi_stream_t stream);
/* The same logits from the un-normalised projections (models/moco.py:113-138: q = normalize(encoder_q(im_q)), k =
 * normalize(encoder_k(im_k)), logits): q_hat (B, C), q_inv (B) = 1/|q| and k_hat (B, C) leave as by-products - the inputs
 * of mi_l2norm_bwd / mi_moco_logits_bwd and of the enqueue.  Rows bit-identical to mi_l2norm_fwd + mi_moco_logits_fwd. */
int mi_moco_logits_norm_fwd(const float* q_raw, const float* k_raw, const float* queue, float* logits, float* q_hat,
                            float* q_inv, float* k_hat, int B, int C, int R, float T, mi_stream_t stream);
int mi_moco_logits_bwd(const float* dlogits, const float* k, const float* queue, float* dq, int B,
                       int C, int R, float T, mi_stream_t stream);
/* SimSiam loss pieces (trains/tomo_simsiam_trainer.py:28-40) on L2-normalised rows:
 * out = mean_b(a_b . b_b) (nn.CosineSimilarity(dim=1)(p, z).mean() once p, z are normalised), its
 * gradient da = grad_out/B * b, and output_std = torch.std(x, 0).mean(). */
int mi_rowdot_mean_fwd(const float* a, const float* b, float* out, int B, int C, mi_stream_t stream);
int mi_rowdot_mean_bwd(const float* b, const float* grad_out, float* da, int B, int C, mi_stream_t stream);
int mi_column_std_mean(const float* x, float* out, int B, int C, mi_stream_t stream);
/* nn.CrossEntropyLoss against label 0 (trains/tomo_moco_trainer.py:52,73; models/moco.py:141):
 * loss = mean_b(logsumexp(l_b) - l_b[0]); dlogits = grad_scale*(softmax - onehot0)/B (may be NULL). */
int mi_ce_label0(const float* logits, float* loss, float* row_loss, float* dlogits, int B, int n,
                 float grad_scale, mi_stream_t stream);
/* The same loss in ONE launch (a workgroup per row; the last one to finish takes the mean, in row order), row_loss[B] and
 * row_lse[B] scratch / kept for the backward pass; the backward reads the upstream gradient from the device (no host
 * value, no extra scaling launch): dlogits = grad_loss * (softmax - onehot0) / B.
 * counter: the arrival counter of the "last workgroup" - ONE 4-byte word owned by the caller, zero before the first call and
 * left zero by every call; give each stream (and each model) its own.  NULL: a process-wide word - one call at a time per device. */
int mi_ce_label0_fwd(const float* logits, float* loss, float* loss_copy /* may be NULL: a second place for the value */,
                     float* row_loss, float* row_lse, int B, int n, unsigned* counter, mi_stream_t stream);
int mi_ce_label0_bwd(const float* logits, const float* row_lse, const float* grad_loss, float* dlogits, int B, int n,
                     mi_stream_t stream);

/* Data gradient of the encoder's stride-2 3x3x3 convolutions (moco_encoder_3d.py:55-84,257-272: layer2.0.conv1 64 -> 128 on 8^3,
 * layer3.0.conv1 128 -> 256 on 4^3 - and on 8^3, the 64^3 crops' layer3.0; padding 1) with the block's 1x1 stride-2 shortcut folded in - one launch per block instead of
 * two data-gradient launches and their split-K reduces (csrc/conv_s2.hip):
 *   dx (N, Gi, Gi, Gi, Ci) = (conv_dgrad(dh; w) [+ conv1x1_dgrad(dout; w_ds)] + res) * (mask > 0)
 * dh / dout: (N, Gi/2, Gi/2, Gi/2, Co); w: [27][Ci][Co], w_ds: [Ci][Co] (kernel layouts); dout and w_ds are given together or
 * both NULL; res / mask (shape of dx) may be NULL.  `ws`: mi_conv3d_s2_dgrad_workspace_bytes(Ci, Co) bytes - the call cuts the
 * weight images into it.  mi_conv3d_s2_dgrad_usable: 1 for the three shapes above (bf16x3 arithmetic), else 0 (the caller keeps
 * mi_convnd_dgrad_f32). */
int mi_conv3d_s2_dgrad_usable(int N, int Gi, int Ci, int Co);
size_t mi_conv3d_s2_dgrad_workspace_bytes(int Ci, int Co);
int mi_conv3d_s2_dgrad_f32(const float* dh, const float* dout, const float* w, const float* w_ds, float* dx, const float* res,
                           const float* mask, int N, int Gi, int Ci, int Co, void* ws, size_t ws_bytes, mi_stream_t stream);

/* Forward of the same block front in one launch (csrc/conv_s2.hip): hmid (N, Gi/2.., Co) = relu(conv3d(x; w [27][Ci][Co], k 3,
 * stride 2, pad 1)) and the 1x1 stride-2 shortcut r (N, Gi/2.., Co) = conv3d(x; w_ds [Ci][Co]) - replaces two mi_conv3d_fwd_f32
 * calls (models/networks/moco_encoder_3d.py:66-69,78-79: conv1 + relu, downsample).  `ws`: mi_conv3d_s2_fwd_workspace_bytes(Ci, Co)
 * bytes - the call cuts the weight image into it.  mi_conv3d_s2_fwd_usable: 1 for the two encoder shapes, else 0. */
int mi_conv3d_s2_fwd_usable(int N, int Gi, int Ci, int Co);
size_t mi_conv3d_s2_fwd_workspace_bytes(int Ci, int Co);
int mi_conv3d_s2_fwd_f32(const float* x, const float* w, const float* w_ds, float* hmid, float* r, int N, int Gi, int Ci, int Co,
                         void* ws, size_t ws_bytes, mi_stream_t stream);

/* Caller-kept images of the stride-2 fronts: mi_conv3d_s2_prep cuts the images of n fronts in one launch per 8 (host arrays of
 * device pointers; w[i] [27][Ci][Co], w_ds[i] [Ci][Co] - may be NULL only for a data-gradient image of a block without
 * shortcut -, img[i] of mi_conv3d_s2_fwd_workspace_bytes / mi_conv3d_s2_dgrad_workspace_bytes bytes, dgrad[i] 0 / 1); the _img_
 * calls are mi_conv3d_s2_fwd_f32 / mi_conv3d_s2_dgrad_f32 without the image build (a training loop cuts the images once per
 * weight update, behind its SGD / momentum kernel, instead of in front of every call). */
int mi_conv3d_s2_prep(const float* const* w, const float* const* w_ds, void* const* img, const int* ci, const int* co,
                      const int* dgrad, int n, mi_stream_t stream);
int mi_conv3d_s2_fwd_img_f32(const float* x, const void* img, float* hmid, float* r, int N, int Gi, int Ci, int Co,
                             mi_stream_t stream);
int mi_conv3d_s2_dgrad_img_f32(const float* dh, const float* dout, const void* img, float* dx, const float* res,
                               const float* mask, int N, int Gi, int Ci, int Co, mi_stream_t stream);

/* models/moco.py:31-39: k <- m*k + (1-m)*q over a flat parameter arena (16-B aligned). */
int mi_ema_update(float* k, const float* q, float m, long n, mi_stream_t stream);
/* torch.optim.SGD (moco_main.py:79, no momentum): p <- p - lr*(grad_scale*g + wd*p).  lr_dev (device float,
 * may be NULL -> lr) lets a captured graph follow the schedule; grad_scale = 1 / world size when g holds the sum of the
 * data-parallel ranks' gradients (DistributedDataParallel's averaging, moco_main.py:44-66), 1 otherwise. */
int mi_sgd_step(float* p, const float* g, const float* lr_dev, float lr, float weight_decay, float grad_scale, long n,
                mi_stream_t stream);
/* Round 6: p <- p - lr (grad_scale (g + g2) + wd p): the step of a model whose two views wrote their parameter gradients into two
 * arenas (trains/simsiam_engine.py; simsiam_main.py:65 SGD over both views' accumulated gradients). */
int mi_sgd_step2(float* p, const float* g, const float* g2, const float* lr_dev, float lr, float weight_decay, float grad_scale, long n,
                 mi_stream_t stream);
/* sums[i] += *src_i for the non-NULL device scalars a, b, c, d (the per-step loss meters of trains/base_trainer.py:514-530, kept on
 * the device and read once per print interval). */
int mi_scalar_accumulate(float* sums, const float* a, const float* b, const float* c, const float* d, mi_stream_t stream);
/* models/moco.py:41-52: queue[:, ptr:ptr+B] = keys.T; ptr = (ptr+B) % R, ptr read and advanced on
 * the device (no host sync).  R % B == 0 as the reference asserts. */
int mi_queue_enqueue(float* queue, int64_t* queue_ptr, const float* keys, int B, int C, int R,
                     mi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Detector-training losses, cet_pick/models/loss.py.  `pred` is the clamped sigmoid
 * heat-map, `gt` the label map (1 positive, (-1,1) soft, -1 unlabeled).  Each forward writes the scalar loss
 * (device float) and `sums` (11 doubles: the 8 partial sums, loss, PU branch flag, n) that the backward
 * reads; nothing synchronises.  ws: mi_voxel_loss_workspace_bytes(n).
 * ------------------------------------------------------------------------------------------ */
size_t mi_voxel_loss_workspace_bytes(long n);
/* `_pu_neg_loss(pred, gt, tau, beta, gamma)` loss.py:255-308 (the `< -beta` branch is taken on the device). */
int mi_pu_focal_loss_fwd(const float* pred, const float* gt, long n, double tau, double beta, double* sums,
                         float* loss, void* ws, size_t ws_bytes, mi_stream_t stream);
int mi_pu_focal_loss_bwd(const float* pred, const float* gt, long n, double tau, const double* sums,
                         const float* dloss, float* dpred, mi_stream_t stream);
/* `_neg_loss(pred, gt)` loss.py:378-411 (FocalLoss). */
int mi_focal_loss_fwd(const float* pred, const float* gt, long n, double* sums, float* loss, void* ws,
                      size_t ws_bytes, mi_stream_t stream);
int mi_focal_loss_bwd(const float* pred, const float* gt, long n, const double* sums, const float* dloss,
                      float* dpred, mi_stream_t stream);
/* `ConsistencyLoss` loss.py:701-712: mean((a-b)^2); db may be NULL. */
int mi_mse_loss_fwd(const float* a, const float* b, long n, double* sums, float* loss, void* ws,
                    size_t ws_bytes, mi_stream_t stream);
int mi_mse_loss_bwd(const float* a, const float* b, long n, const double* sums, const float* dloss, float* da,
                    float* db, mi_stream_t stream);
/* `UnbiasedConLoss.forward` loss.py:594-699 without the (2N)^2 matrix.  feat [n2][dim] (both views stacked,
 * n2 = 2N, row i and row i +- N are the two views of a voxel; dim 32 or 64), cls[n2]: bit 0 positive label,
 * bit 1 "other" (label < thresh).  With S = feat feat^T * inv_T, m_i = max_j S_ij and
 * E_ij = exp((S_ij - m_i) * [i != j]) (loss.py:615-624) the forward returns per row
 *   rowmax = m_i, s_all = sum_j E_ij, s_pos = sum_j E_ij [pos j], s_other = sum_j E_ij [other j], e_pair = E_i,pair(i)
 * and the backward, given the gradients of those four sums, d(loss)/d(feat) (m_i is detached like
 * `logits_max_all.detach()`). */
int mi_ucl_rowsums_fwd(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T, float* rowmax,
                       float* s_all, float* s_pos, float* s_other, float* e_pair, mi_stream_t stream);
int mi_ucl_rowsums_bwd(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T, const float* rowmax,
                       const float* g_all, const float* g_pos, const float* g_other, const float* g_pair,
                       float* dfeat, mi_stream_t stream);
/* Round 6: the same gradient with both terms of a similarity tile formed at once (S is symmetric: one product, one contraction - what
 * mi_ucl_rowsums_bwd does too, MI_UCL_BWD_SPLIT=1 for the two launches of rounds 2-5) and, where the row maxima lie within 2^16 of each
 * other - always for L2-normalised features - ONE exponential per similarity; decided on the device from `range` (2 floats of scratch). */
int mi_ucl_rowsums_bwd_ranged(const float* feat, const uint8_t* cls, int n2, int dim, float inv_T, const float* rowmax,
                              const float* g_all, const float* g_pos, const float* g_other, const float* g_pair, float* dfeat,
                              float* range, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CETPICK_HIP_H */
