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
 * logits must not alias heat_out. */
size_t mi_decode_workspace_bytes(int D, int H, int W, int K);
int mi_sigmoid_nms_topk(const float* logits, float* heat_out, int D, int H, int W, int k,
                        int fiber, int apply_sigmoid, int K, float* dets, int32_t* n_valid_out,
                        void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* scipy.ndimage.gaussian_filter as called by utils/image.py:152-156 and utils/loader.py:102
 * (mode='reflect', truncate=4 -> radius=int(4*sigma+.5)), separable, fp32 storage,
 * fp32 accumulation.  `tmp` is a scratch volume of D*H*W floats; out may alias in. */
int mi_gauss3d_sep(const float* in, float* out, float* tmp, int D, int H, int W, float sigma,
                   mi_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* CETPICK_HIP_H */
