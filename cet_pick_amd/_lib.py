"""ctypes binding of the C-ABI in include/cetpick_hip.h.

The product path has no CPU fallback: if libcetpick_hip.so is missing or a call fails this module
raises.  PyTorch is only the owner of device memory and streams here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CETPICK_HIP_LIB") or os.path.join(_HERE, "libcetpick_hip.so")   # env: tuning builds

_c = ctypes
_P, _I, _F, _Z, _D, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_double, _c.c_long

# name -> (restype, argtypes); kept in step with include/cetpick_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "mi_abi_version": (_I, []),
    "mi_graph_node_counts": (_I, [_P, _P]),
    "mi_debug_stamp": (_I, [_P, _P]),
    "mi_linear_stats_fwd_f32": (_I, [_P] * 5 + [_I] * 3 + [_P]),
    "mi_debug_last_conv_kernel": (_c.c_char_p, []),
    "mi_linear_bn_fwd_f32": (_I, [_P] * 5 + [_I] * 3 + [_P, _P, _F, _F] + [_P] * 4 + [_I, _P]),
    "mi_conv3d_stem_stats_workspace_bytes": (_Z, [_I] * 5),
    "mi_conv3d_stem_stats_f32": (_I, [_P, _P, _P] + [_I] * 5 + [_P, _P, _Z, _P]),
    "mi_build_arch": (_c.c_char_p, []),
    "mi_sigmoid_clamp": (_I, [_P, _P, _Z, _P]),
    "mi_nms3d": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mi_decode_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "mi_decode_workspace_init": (_I, [_P, _Z, _P]),
    "mi_sigmoid_nms_topk": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "mi_gauss3d_sep": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "mi_dog_pick_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "mi_dog_pick": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _Z, _P]),
    "mi_greedy_nms3d_workspace_bytes": (_Z, [_I, _I, _I]),
    "mi_greedy_nms3d": (_I, [_P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _I, _P, _Z, _P]),
    "mi_crop_normalize": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mi_crop_normalize_table": (_I, [_P, _P, _P, _P, _P, _c.c_int64, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mi_u8_roundtrip_normalize": (_I, [_P, _P, _Z, _F, _F, _P]),
    # training path
    "mi_gauss2d_slices": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "mi_rec_reorder": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mi_vol_stats_workspace_bytes": (_Z, [_L, _L]),
    "mi_vol_stats": (_I, [_P, _L, _L, _P, _P, _Z, _P]),
    "mi_zscore": (_I, [_P, _P, _L, _L, _P, _P]),
    "mi_zscore_quantize_minmax": (_I, [_P, _P, _L, _L, _P, _D, _D, _I, _P]),
    "mi_conv3d_workspace_bytes": (_Z, [_I] * 9),
    "mi_conv3d_fwd_f32": (_I, [_P, _P, _P, _P, _I] + [_I] * 9 + [_P, _Z, _P]),
    "mi_conv3d_dgrad_f32": (_I, [_P, _P, _P, _P, _P] + [_I] * 9 + [_P, _Z, _P]),
    "mi_conv3d_wgrad_f32": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _Z, _P]),
    "mi_convnd_workspace_bytes": (_Z, [_I] * 13),
    "mi_convnd_fwd_f32": (_I, [_P, _P, _P, _P, _I] + [_I] * 13 + [_P, _Z, _P]),
    "mi_convnd_dgrad_f32": (_I, [_P, _P, _P, _P, _P] + [_I] * 13 + [_P, _Z, _P]),
    "mi_convnd_wgrad_f32": (_I, [_P, _P, _P] + [_I] * 13 + [_P, _Z, _P]),
    "mi_convnd_fwd_bias_f32": (_I, [_P, _P, _P, _P] + [_I] * 14 + [_P, _Z, _P]),
    "mi_stem2d_fwd_bias_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mi_upconv_tail_fwd": (_I, [_P, _P, _P, _P, _P] + [_I] * 7 + [_P]),
    "mi_smallk_image_bytes": (_Z, [_I, _I]),
    "mi_smallk_prep": (_I, [_P, _P, _I, _I, _P]),
    "mi_smallk_fwd_f32": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _I, _L, _I, _P]),
    "mi_smallk_heads_fwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _L, _I, _P]),
    "mi_convnd_wgrad_slabs_batch_f32": (_I, [_P, _P, _P, _P] + [_I] * 14 + [_Z, _P, _P]),
    "mi_convnd_wgrad_slabs_f32": (_I, [_P, _P, _P] + [_I] * 13 + [_P, _Z, _P, _P]),
    "mi_splitk_reduce_batch": (_I, [_P, _P, _P, _P, _I, _P]),
    "mi_linear_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "mi_conv3d_direct_wimg_bytes": (_Z, [_I]),
    "mi_conv3d_direct_workspace_bytes": (_Z, [_I, _I]),
    "mi_conv3d_direct_usable": (_I, [_I] * 9),
    "mi_conv3d_direct_prep": (_I, [_P, _P, _P, _P, _I, _P]),
    "mi_conv3d_direct_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "mi_conv2d_p2d_usable": (_I, [_I] * 4),
    "mi_conv2d_p2d_wimg_bytes": (_Z, [_I]),
    "mi_conv2d_p2d_prep": (_I, [_P, _P, _P, _P, _I, _P]),
    "mi_conv2d_p2d_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mi_conv2d_p2d_wgrad_workspace_bytes": (_Z, [_I] * 4),
    "mi_conv2d_p2d_wgrad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "mi_conv2d_stem3_workspace_bytes": (_Z, [_I]),
    "mi_conv2d_stem3_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mi_conv2d_stem3_wgrad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "mi_conv3d_cube2_workspace_bytes": (_Z, [_I, _I]),
    "mi_conv3d_cube2_usable": (_I, [_I] * 9),
    "mi_conv3d_cube2_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "mi_convnd_dil_workspace_bytes": (_Z, [_I] * 15),
    "mi_conv_d32_kind": (_I, [_I] * 12),
    "mi_conv_d32_image_bytes": (_Z, [_I, _I]),
    "mi_conv_d64_image_bytes": (_Z, [_I, _I]),
    "mi_conv_d32_prep": (_I, [_P, _P, _I, _I, _P]),
    "mi_conv_d64_prep": (_I, [_P, _P, _I, _I, _P]),
    "mi_conv_d64_prep_co": (_I, [_P, _P, _I, _I, _I, _P]),
    "mi_conv_d32_1x1_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mi_conv_d32_fwd_pool_f32": (_I, [_P, _P, _P, _P, _P] + [_I] * 7 + [_P]),
    "mi_conv_d32_fwd_pool_strided_f32": (_I, [_P, _P, _P, _P, _I, _P] + [_I] * 7 + [_P]),
    "mi_copy_channels_into": (_I, [_P, _I, _P, _I, _I, _L, _P]),
    "mi_conv_d32_upconv_fwd_f32": (_I, [_P, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "mi_conv_d64_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mi_conv_d32_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mi_convnd_dil_fwd_f32": (_I, [_P, _P, _P, _P, _I] + [_I] * 15 + [_P, _Z, _P]),
    "mi_convnd_dil_dgrad_f32": (_I, [_P, _P, _P, _P, _P] + [_I] * 15 + [_P, _Z, _P]),
    "mi_convnd_dil_wgrad_f32": (_I, [_P, _P, _P] + [_I] * 15 + [_P, _Z, _P]),
    "mi_maxpool2d_ceil_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mi_maxpool2d_ceil_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mi_shuffle2x2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mi_shuffle2x2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mi_concat_channels": (_I, [_P, _I, _P, _I, _P, _L, _P]),
    "mi_split_channels": (_I, [_P, _P, _I, _P, _I, _L, _P]),
    "mi_zhead_fwd": (_I, [_P, _P, _P, _I, _I, _L, _I, _I, _P]),
    "mi_zhead_bwd_workspace_bytes": (_Z, [_I, _I, _L, _I, _I]),
    "mi_zhead_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _P, _Z, _P]),
    "mi_voxel_loss_workspace_bytes": (_Z, [_L]),
    "mi_pu_focal_loss_fwd": (_I, [_P, _P, _L, _D, _D, _P, _P, _P, _Z, _P]),
    "mi_pu_focal_loss_bwd": (_I, [_P, _P, _L, _D, _P, _P, _P, _P]),
    "mi_focal_loss_fwd": (_I, [_P, _P, _L, _P, _P, _P, _Z, _P]),
    "mi_focal_loss_bwd": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "mi_mse_loss_fwd": (_I, [_P, _P, _L, _P, _P, _P, _Z, _P]),
    "mi_mse_loss_bwd": (_I, [_P, _P, _L, _P, _P, _P, _P, _P]),
    "mi_ucl_rowsums_fwd": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "mi_ucl_rowsums_bwd": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "mi_ucl_rowsums_bwd_ranged": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_colreduce_workspace_bytes": (_Z, [_L, _I]),
    "mi_bn_stats": (_I, [_P, _L, _I, _P, _P, _Z, _P]),
    "mi_bn_apply_fwd": (_I, [_P, _P, _L, _I, _P, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _P]),
    "mi_bn_relu_maxpool3d_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mi_bn_relu_bwd_reduce_x": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "mi_bn_relu_bwd_apply_x": (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _P, _D, _P, _P, _P]),
    "mi_bn_small_fwd": (_I, [_P, _P, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _P]),
    "mi_bn_small_bwd": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _P, _P]),
    "mi_bn_small_pool_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mi_bn_small_pool_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "mi_bn_eval_fwd": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "mi_bn_bwd_reduce": (_I, [_P, _P, _P, _L, _I, _P, _I, _P, _P, _Z, _P]),
    "mi_bn_bwd_apply": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _D, _I, _P, _P, _P]),
    "mi_bn_bwd_apply_res": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _D, _P, _P, _P]),
    "mi_bn_param_grads": (_I, [_P, _I, _P, _P, _P]),
    "mi_colsum": (_I, [_P, _L, _I, _P, _P, _P, _Z, _P]),
    "mi_maxpool3d_fwd": (_I, [_P, _P, _P] + [_I] * 8 + [_P]),
    "mi_maxpool3d_bwd": (_I, [_P, _P, _P] + [_I] * 8 + [_P]),
    "mi_avgpool_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mi_avgpool_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mi_bias_add": (_I, [_P, _P, _L, _I, _P]),
    "mi_copy_pair_f32": (_I, [_P, _P, _P, _P, _L, _P]),
    "mi_relu_mask": (_I, [_P, _P, _P, _P, _L, _P]),
    "mi_l2norm_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "mi_l2norm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "mi_moco_logits_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "mi_moco_logits_norm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "mi_moco_logits_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "mi_rowdot_mean_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "mi_rowdot_mean_bwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "mi_column_std_mean": (_I, [_P, _P, _I, _I, _P]),
    "mi_ce_label0": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "mi_ce_label0_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "mi_ce_label0_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "mi_conv3d_s2_dgrad_usable": (_I, [_I, _I, _I, _I]),
    "mi_conv3d_s2_dgrad_workspace_bytes": (_Z, [_I, _I]),
    "mi_conv3d_s2_dgrad_f32": (_I, [_P] * 7 + [_I, _I, _I, _I, _P, _Z, _P]),
    "mi_conv3d_s2_fwd_usable": (_I, [_I, _I, _I, _I]),
    "mi_conv3d_s2_fwd_workspace_bytes": (_Z, [_I, _I]),
    "mi_conv3d_s2_fwd_f32": (_I, [_P] * 5 + [_I, _I, _I, _I, _P, _Z, _P]),
    "mi_conv3d_s2_prep": (_I, [_P] * 6 + [_I, _P]),
    "mi_conv3d_s2_fwd_img_f32": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "mi_conv3d_s2_dgrad_img_f32": (_I, [_P] * 6 + [_I] * 4 + [_P]),
    "mi_ema_update": (_I, [_P, _P, _F, _L, _P]),
    "mi_sgd_step": (_I, [_P, _P, _P, _F, _F, _F, _L, _P]),
    "mi_sgd_step2": (_I, [_P, _P, _P, _P, _F, _F, _F, _L, _P]),
    "mi_scalar_accumulate": (_I, [_P, _P, _P, _P, _P, _P]),
    "mi_queue_enqueue": (_I, [_P, _P, _P, _I, _I, _I, _P]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def lib():
    """The loaded shared library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                "%s is missing: run `python -m cet_pick_amd.build` (there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "workspace too small", -3: "unsupported"}.get(rc, "hipError %d" % rc)
        raise HipExtensionError("%s failed: %s" % (what, kind))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name="tensor", dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipExtensionError("%s must be a tensor on the MI355X (cuda) device; there is no CPU path" % name)
    if dtype is not None and t.dtype != dtype:
        raise HipExtensionError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


_workspaces = {}


def _ws_key(device, tag):
    return (str(device), tag, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)


def workspace(nbytes, device, tag="default", init=None):
    """A cached device scratch buffer of at least nbytes (grown geometrically, never shrunk).
    `init(buf)` runs once per (re)allocation - for workspaces whose header is kept clean from call to call."""
    # one buffer per (device, purpose, stream): launches on different streams must not share scratch
    key = _ws_key(device, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        if init is not None:
            init(buf)
            _ws_inits[key] = init
        _workspaces[key] = buf
    return buf


_ws_inits = {}


def prime_workspaces_for_stream(src_stream, dst_stream):
    """Give `dst_stream` its own copy of every kept-clean (init) workspace `src_stream` has, initialised NOW: a step engine calls this
    in front of a hipGraph capture on `dst_stream` - workspaces are keyed by stream, and one that is first allocated inside the capture
    has its initialising fill recorded as a node that replays with every step (2.6 MB of zeros for cube2_kernel's arrival counters)."""
    src, dst = src_stream.cuda_stream, dst_stream.cuda_stream
    if src == dst:
        return
    with torch.cuda.stream(dst_stream):
        for key in [k for k in _workspaces if k[2] == src and k in _ws_inits]:
            nk = (key[0], key[1], dst)
            if nk in _workspaces and _workspaces[nk].numel() >= _workspaces[key].numel():
                continue
            buf = torch.empty(_workspaces[key].numel(), dtype=torch.uint8, device=_workspaces[key].device)
            _ws_inits[key](buf)
            _workspaces[nk], _ws_inits[nk] = buf, _ws_inits[key]
    dst_stream.synchronize()


def drop_workspace(device, tag):
    """Forget a cached workspace (after a failed call its kept-clean header can no longer be trusted)."""
    _workspaces.pop(_ws_key(device, tag), None)
    _ws_inits.pop(_ws_key(device, tag), None)
