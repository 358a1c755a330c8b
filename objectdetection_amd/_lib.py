"""ctypes binding of libfrcnn_hip.so (the C ABI declared in include/frcnn_hip.h).

There is no CPU fallback: if the shared library is missing the import of any op raises, and
every op refuses tensors that are not on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfrcnn_hip.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_D = _c.c_double
_L = _c.c_int64
_Z = _c.c_size_t

# name -> (restype, argtypes); mirrors include/frcnn_hip.h one to one
SIGNATURES = {
    "frcnn_abi_version": (_I, []),
    "frcnn_last_error_string": (_c.c_char_p, []),
    "frcnn_base_anchors_f32": (_I, [_P, _P, _I, _P]),
    "frcnn_anchor_grid_f32": (_I, [_P, _I, _I, _I, _F, _F, _P, _P]),
    "frcnn_rpn_flatten_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "frcnn_topk_workspace_bytes": (_Z, [_I, _I, _I]),
    "frcnn_topk_desc_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "frcnn_topk_desc_ex_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _Z, _I, _P]),
    "frcnn_box_regression_f32": (_I, [_P, _P, _I, _I, _P, _P]),
    "frcnn_box_regression_backward_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "frcnn_box_clip_f32": (_I, [_P, _I, _F, _F, _P, _P]),
    "frcnn_remove_small_box_f32": (_I, [_P, _I, _F, _P, _P, _P, _P]),
    "frcnn_select_gt_f32": (_I, [_P, _I, _F, _P, _P, _P]),
    "frcnn_decode_clip_filter_f32": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "frcnn_topk_desc_nchw_supported": (_I, [_I, _I, _I, _I]),
    "frcnn_topk_desc_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _Z, _I, _P]),
    "frcnn_decode_clip_filter_nchw_f32": (_I, [_P, _L, _P, _P, _I, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P,
                                               _P]),
    "frcnn_rpn_proposals_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "frcnn_rpn_proposals_nchw_f32": (_I, [_P, _P, _P, _I, _I, _P, _L, _I, _I, _F, _F, _F, _D, _I, _P, _P, _P, _Z, _I, _I,
                                          _P, _P]),
    "frcnn_nms_workspace_bytes": (_Z, [_I, _I]),
    "frcnn_nms_sorted_f32": (_I, [_P, _P, _I, _I, _D, _I, _P, _P, _P, _Z, _P]),
    "frcnn_nms_sorted_rounds_f32": (_I, [_P, _P, _I, _I, _D, _I, _P, _P, _P, _Z, _I, _P, _P]),
    "frcnn_nms_static_scratch_bytes": (_Z, []),
    "frcnn_nms_sorted_static_f32": (_I, [_P, _P, _I, _I, _D, _I, _P, _P, _P]),
    "frcnn_gather_boxes_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "frcnn_box_iou_f32": (_I, [_P, _I, _P, _I, _P, _P]),
    "frcnn_iou_label_encode_f32": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "frcnn_build_rois_f32": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "frcnn_level_map_f32": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "frcnn_roi_align_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _F, _I, _P, _I, _P, _P]),
    "frcnn_roi_align_batched_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _F, _I, _I, _I, _I, _I, _P, _P, _P]),
    "frcnn_roi_align_levels_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "frcnn_roi_align_levels_cl_workspace_bytes": (_Z, [_P, _I, _I, _I, _I, _I]),
    "frcnn_roi_align_levels_cl_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z,
                                           _P]),
    "frcnn_roi_align_batched_backward_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _F, _I, _I, _I, _I, _I, _P, _P]),
    "frcnn_roi_pool_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _P, _P, _P]),
    "frcnn_roi_pool_batched_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "frcnn_roi_detect_filter_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P]),
    "frcnn_rpn_sampled_loss_f32": (_I, [_P, _P, _P, _P, _P, _I, _L, _F, _P, _P, _P, _P]),
    "frcnn_roi_loss_f32": (_I, [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "frcnn_resize_pad_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "frcnn_resize_pad_batch_f32": (_I, [_P, _P, _I, _I, _P, _I, _I, _P]),
}


class FrcnnError(RuntimeError):
    pass


_lib = None


def load():
    """Load libfrcnn_hip.so and bind every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FrcnnError(
            "libfrcnn_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries its own copy of the HIP runtime (torch/lib/libamdhip64.so, soname
    # libamdhip64.so.7, the name this library needs).  Loaded after torch, libfrcnn_hip.so binds to that copy
    # and shares torch's device state and streams; loaded before it, the system copy under /opt/rocm comes in
    # as a SECOND runtime in the process, and whichever initialises second finds "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().frcnn_last_error_string()
        raise FrcnnError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
