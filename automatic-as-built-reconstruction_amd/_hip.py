"""ctypes binding of lib/libaabr_hip.so (C ABI: include/aabr_hip.h).

PyTorch is used only for device memory and streams; every compute call goes through this
module.  Loading fails loudly when the library is missing -- there is no fallback path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaabr_hip.so")

_lib = None
META_WORDS = 16
ABI_VERSION = 600      # include/aabr_hip.h AABR_ABI_VERSION this binding (_SIGS) was written for

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

_SIGS = {
    "aabr_version": (C.c_int, []),
    "aabr_build_flags": (C.c_int, []),
    "aabr_last_error": (C.c_char_p, []),
    "aabr_set_knob": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "aabr_quantize_points": (C.c_int, [_vp, _i32, _i64, C.c_double, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "aabr_input_layer_status_words": (C.c_int64, [_i64]),
    "aabr_input_layer_sites": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aabr_input_layer_pack_bits": (C.c_int, [_i64, _i32p]),
    "aabr_input_layer_sites_packed": (C.c_int, [_vp, _i64, _i32, _i32p, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp, _vp, _vp, _vp, _vp]),
    "aabr_input_layer_forward": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "aabr_input_layer_backward": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "aabr_input_layer_rule_table": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "aabr_submanifold_table": (C.c_int, [_vp, _i64, _vp, _i64, _i32p, _vp, _vp, _vp]),
    "aabr_convolution_sites": (C.c_int, [_vp, _i64, _i32p, _i32p, _i32p, _vp, _i64, _vp, _vp, _vp, _vp]),
    "aabr_convolution_tables": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32p, _i32p,
                                          _i32p, _vp, _vp, _vp, _vp]),
    "aabr_convolution_tables2": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32p, _i32p,
                                           _i32p, _vp, _vp, _vp, _vp, _vp]),
    "aabr_sample_offsets": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "aabr_brick_scratch_words": (C.c_int64, [_i64, _i64]),
    "aabr_brick_build": (C.c_int, [_vp, _i64, _vp, _i32p, _i32p, _i32p, _i32p, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp,
                                   _i32, _vp]),
    "aabr_brick_renumber": (C.c_int, [_vp, _i64, _i32p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                      _vp, _vp]),
    "aabr_points_prepare": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aabr_points_sites": (C.c_int, [_vp, _i64, _i32p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "aabr_brick_submanifold_table": (C.c_int, [_vp, _i64, _i32p, _vp, _vp, _i32p, _vp, _vp, _vp]),
    "aabr_brick_convolution_tables": (C.c_int, [_vp, _i64, _i32p, _vp, _vp, _vp, _i64, _i32p, _vp, _vp, _i32p, _i32p,
                                                _i32p, _vp, _vp, _vp, _vp, _vp]),
    "aabr_table_to_rulebook": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "aabr_spatial_locations": (C.c_int, [_vp, _i64, _vp, _vp]),
    "aabr_conv_last_variant": (C.c_char_p, []),
    "aabr_conv_wide_tile_rows": (C.c_int, [_i32, _i32, _i64, _i64, _i32]),
    "aabr_wide_blocks_words": (C.c_int64, [_i64, _i32, _i32]),
    "aabr_build_wide_blocks": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "aabr_conv_pack_weights": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "aabr_conv_forward_wide": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "aabr_conv_wide_split": (C.c_int, [_i32, _i32, _i64, _i64, _i32]),
    "aabr_conv_wide_split_scratch_floats": (C.c_int64, [_i64, _i32, _i32]),
    "aabr_conv_narrow_ok": (C.c_int, [_i32, _i32, _i64, _i64, _i32, _i32]),
    "aabr_conv_forward_narrow": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp]),
    "aabr_conv_forward_narrow_bf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp]),
    "aabr_conv_narrow_parts": (C.c_int, [_i64]),
    "aabr_conv_forward_narrow_bf16_stats": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "aabr_conv_forward_narrow_bf16_bwd_stats": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp,
                                                          _vp, _f32, _vp]),
    "aabr_conv_wide_split_bf16": (C.c_int, [_i32, _i32, _i64, _i64, _i32]),
    "aabr_conv_forward_wide_split_bf16": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp,
                                                    _i32, _vp, _vp]),
    "aabr_conv_forward_wide_split_bf16_res": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp,
                                                        _i32, _vp, _vp, _vp]),
    "aabr_conv_forward_wide_bf16_res": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp,
                                                  _vp, _vp, _vp, _vp, _f32, _vp]),
    "aabr_bn_backward_add_bf16": (C.c_int, [_vp] * 4 + [_i64, _i32] + [_vp] * 6 + [_f32, _vp, _i32, _vp, _vp, _vp]),
    "aabr_conv_forward_wide_split": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp,
                                               _i32, _vp, _vp]),
    "aabr_conv_wide_tile_rows_bf16": (C.c_int, [_i32, _i32, _i64, _i64, _i32]),
    "aabr_conv_forward_wide_bf16": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "aabr_conv_forward_wide_res": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp,
                                             _vp]),
    "aabr_conv_wide_stats_doubles": (C.c_int64, [_i64, _i32, _i32]),
    "aabr_conv_forward_wide_stats": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp,
                                               _vp, _vp]),
    "aabr_conv_forward_wide_bf16_stats": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp,
                                                    _vp, _vp]),
    "aabr_conv_forward_wide_bf16_bwd_stats": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp,
                                                        _vp, _vp, _vp, _vp, _f32, _vp]),
    "aabr_bn_backward_parts_bf16": (C.c_int, [_vp] * 4 + [_i64, _i32] + [_vp] * 6 + [_f32, _vp, _i32, _vp, _vp]),
    "aabr_conv_forward_wide_bwd_stats": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp,
                                                   _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp]),
    "aabr_bn_backward_parts": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32,
                                         _vp, _vp, _vp]),
    "aabr_bn_forward_parts": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp,
                                        _i32, _vp, _vp]),
    "aabr_bn_forward_parts_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp,
                                             _i32, _vp, _vp]),
    "aabr_mailbox_create": (C.c_int, [_i64, _vp]),
    "aabr_mailbox_destroy": (C.c_int, [_vp]),
    "aabr_mailbox_post": (C.c_int, [_vp, _i64, _vp, C.c_uint32, _vp]),
    "aabr_conv_wpack_floats": (C.c_int64, [_i32, _i32, _i32]),
    "aabr_conv_forward": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "aabr_conv_dw_scratch_floats": (C.c_int64, [_i64, _i32, _i32]),
    "aabr_conv_backward_weight": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp, _vp]),
    "aabr_conv_dw_chunk_pairs": (C.c_int, [_i64, _i32, _i32, _i32]),
    "aabr_conv_pack_weights2": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "aabr_conv_pack_weights2_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "aabr_plan_run": (C.c_int, [_vp, _i32, _vp]),
    "aabr_plan_submit": (C.c_int, [_vp, _i32, _vp, _i32]),
    "aabr_plan_drain": (C.c_int, []),
    "aabr_plan_launcher_stats": (None, [_vp, _vp, _vp]),
    "aabr_geom_run": (C.c_int, [_vp, _i32, _vp]),
    "aabr_add": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "aabr_cast_storage": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "aabr_sum_counts": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "aabr_conv_pack_job_blocks": (C.c_int64, [_i32, _i32, _i32]),
    "aabr_conv_pack_weights_jobs": (C.c_int, [_vp, _i32, _i64, _vp]),
    "aabr_conv_wpack_bf16_elems": (C.c_int64, [_i32, _i32, _i32]),
    "aabr_conv_forward_bf16": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "aabr_conv_backward_weight_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp,
                                                 _vp]),
    "aabr_tile_blocks_words": (C.c_int64, [_i64, _i32]),
    "aabr_build_tile_blocks": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "aabr_offset_pairs_words": (C.c_int64, [_i64, _i32]),
    "aabr_build_offset_pairs": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "aabr_bn_scratch_floats": (C.c_int64, [_i32]),
    "aabr_bn_forward": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _f32,
                                  _vp, _vp]),
    "aabr_bn_backward": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp]),
    "aabr_bn_backward_add": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp,
                                       _vp]),
    "aabr_bn_forward_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _f32,
                                       _vp, _vp]),
    "aabr_bn_backward_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp,
                                        _vp]),
    "aabr_rotate_iou_eval": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "aabr_boxes_iou_3d": (C.c_int, [_vp, _i64, _vp, _i64, _f32p, _i32, _i32, _vp, _vp]),
    "aabr_sparse_to_dense_forward": (C.c_int, [_vp, _i64, _vp, _i32, _i32p, _i64, _vp, _vp]),
    "aabr_sparse_to_dense_backward": (C.c_int, [_vp, _i64, _vp, _i32, _i32p, _vp, _vp]),
    "aabr_roi_align_rotated_3d_forward": (C.c_int, [_vp, _vp, _i64, _f32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                                    _i32, _vp, _vp]),
    "aabr_roi_align_rotated_3d_backward": (C.c_int, [_vp, _vp, _i64, _f32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                                     _i32, _i32, _vp, _vp]),
    "aabr_roi_cellmap": (C.c_int, [_vp, _i64, _i32p, _i32, _vp, _vp]),
    "aabr_roi_align_rotated_3d_sparse_forward": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _f32,
                                                           _i32, _i32, _i32, _i32, _vp, _vp]),
    "aabr_roi_align_rotated_3d_sparse_backward": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _f32,
                                                            _i32, _i32, _i32, _i32, _i64, _vp, _vp]),
    "aabr_rpn_decode": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _f32, _f32p, _f32p, _f32, _vp, _vp]),
    "aabr_rpn_decode_maps": (C.c_int, [_i32, _vp, _vp, _vp, _i32p, _i32p, _f32p, _vp, _i32, _f32, _f32p, _f32, _f32,
                                       _f32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "aabr_rpn_label_generation": (C.c_int, [_i32, _vp, _i32, _i32p, _i32p, _f32p, _vp, _i32, _f32, _vp, _i32p, _f32p,
                                            _i32, _i32, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "aabr_rpn_label_generation_targets": (C.c_int, [_i32, _vp, _i32, _i32p, _i32p, _f32p, _vp, _i32, _f32, _vp, _i32p,
                                                    _f32p, _i32, _i32, _f32, _f32, _f32, _i32, _vp, _vp, _vp, _vp,
                                                    _f32p, _vp, _vp]),
    "aabr_box_encode": (C.c_int, [_vp, _vp, _i64, _f32p, _vp, _vp]),
    "aabr_box_decode": (C.c_int, [_vp, _vp, _i64, _i32, _f32p, _f32, _vp, _vp]),
    "aabr_rpn_topk_scratch_words": (C.c_int64, [_i32]),
    "aabr_rpn_topk_maps": (C.c_int, [_i32, _vp, _i32, _i32p, _i32p, _i32, _i32p, _vp, _i64, _vp, _vp, _vp]),
    "aabr_rpn_gather_logits": (C.c_int, [_i32, _vp, _i32, _i32p, _i32p, _i32, _i64, _vp, _vp]),
    "aabr_rpn_proposals_batch": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32p, _i32p, _f32p, _vp, _i32, _f32, _f32p, _f32,
                                           _f32, _f32, _vp, _i64, _vp, _vp, _vp, _f32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "aabr_rotate_nms_sorted": (C.c_int, [_vp, _i64, _f32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "aabr_nms_sorted": (C.c_int, [_vp, _i64, _f32, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGS))


class AabrError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (no GPU needed just to load it and resolve symbols)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AabrError(
                "%s is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        if lib.aabr_version() != ABI_VERSION:     # a stale .so would take shifted arguments into kernels that write
            raise AabrError("%s reports ABI version %d, this binding was written for %d: rebuild it "
                            "(`python -c \"import __graft_entry__ as g; g.build()\"`)"
                            % (LIB_PATH, lib.aabr_version(), ABI_VERSION))
        _lib = lib
    return _lib


_gpu_ok = None


def require_gpu(t=None):
    global _gpu_ok
    if t is not None:
        if t.is_cuda:
            return
        if _gpu_ok:
            raise AabrError("expected a tensor in device memory, got %s" % t.device)
    if _gpu_ok is None:
        _gpu_ok = bool(torch.cuda.is_available())
    if not _gpu_ok:
        raise AabrError("the MI355X hot path needs a GPU (torch.cuda.is_available() is False); "
                        "there is no CPU fallback")
    if t is not None and not t.is_cuda:
        raise AabrError("expected a tensor in device memory, got %s" % t.device)


def set_knob(name, value=None):
    """tuning knob of the library (tests / tools): value None = back to "no value" (the shipped default)"""
    check(load().aabr_set_knob(name.encode(), 0 if value is None else int(value), 1 if value is None else 0))


def check(rc):
    if rc != 0:
        raise AabrError("libaabr_hip: rc=%d: %s" % (rc, load().aabr_last_error().decode()))


def ptr(t):
    """device pointer of a tensor (plain int, which ctypes converts for c_void_p), None -> NULL"""
    if t is None:
        return None
    p = t.data_ptr()
    return p if p else None


def stream():
    """raw hipStream_t of torch's CURRENT stream on the current device (honours
    `torch.cuda.stream(...)` contexts); the private fast path avoids the ~13 us
    `torch.cuda.current_stream()` spends re-checking the device count on every call"""
    return _raw_stream(torch._C._cuda_getDevice()) or None


_ws = {}


def workspace(name, numel, dtype, device):
    """Grow-only scratch buffer per (purpose, dtype, device).  Kernels on one stream run in
    order, so a scratch buffer whose contents never outlive the call that fills it can be shared
    by consecutive calls; buffers are keyed by the current stream so concurrent streams never
    share one."""
    key = (name, dtype, device, _raw_stream(torch._C._cuda_getDevice()))
    t = _ws.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(max(int(numel), 1), dtype=dtype, device=device)
        _ws[key] = t
    return t


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream


import threading as _threading
import time as _time

_mb = _threading.local()
rb_trace = None       # tools/: list collecting (label, host clock) inside read_back
_MB_BYTES = 1 << 16
_CT = {torch.int32: C.c_int32, torch.int64: C.c_int64, torch.float32: C.c_float}


def read_back(t):
    """Small device tensor (int32 / int64 / float32, <= 64 KiB) -> Python list through the library's MAILBOX
    (include/aabr_hip.h: aabr_mailbox_post): one small kernel on the current stream writes the values and then a
    sequence word into coherent host memory, and this thread spins on that word -- no stream / event call on the waiting
    side.  Measured on the bench step (tools/tools_step_timeline.py, profiles/r03_step_timeline.txt): the proposal
    stage's one read, issued on its own stream while the backward pass runs on the main one, is complete on the device
    at 8.3 ms of the step; `tensor.tolist()` (hipStreamSynchronize), hipEventSynchronize and a hipEventQuery polling
    loop on that stream ALL returned at 13.6 ms, right behind the last backward kernel of the other stream, and the
    update and the next step started late by as much."""
    n = t.numel()
    ct = _CT.get(t.dtype)
    if n == 0 or ct is None or n * t.element_size() > _MB_BYTES or not t.is_cuda:
        return t.tolist()
    t = t.contiguous()
    st = getattr(_mb, "state", None)
    if st is None:
        box = C.c_void_p()
        if load().aabr_mailbox_create(_MB_BYTES, C.byref(box)) != 0 or not box.value:
            st = _mb.state = False      # no coherent host allocation on this system: the plain synchronous read
        else:
            st = _mb.state = [box.value, 0, C.c_uint32.from_address(box.value)]
    if st is False:
        return t.tolist()
    st[1] = seq = (st[1] % 0x7fffffff) + 1
    if rb_trace is not None:
        rb_trace.append(("post", _time.perf_counter()))
    check(load().aabr_mailbox_post(ptr(t), n * t.element_size(), st[0], seq, stream()))
    if rb_trace is not None:
        rb_trace.append(("posted", _time.perf_counter()))
    word, t0, spins = st[2], None, 0
    while word.value != seq:
        spins += 1
        if spins & 0xfffff == 0:          # a launch that never completes must not hang the host silently
            t0 = t0 or _time.monotonic()
            if _time.monotonic() - t0 > 120.0:
                raise RuntimeError("aabr mailbox: no answer from the device after 120 s")
    if rb_trace is not None:
        rb_trace.append(("seen", _time.perf_counter()))
    st.append(C.c_uint32.from_address(st[0] + 4).value) if len(st) == 3 else st.__setitem__(3, C.c_uint32.from_address(st[0] + 4).value)
    flat = list((ct * n).from_address(st[0] + 8))
    if t.dim() <= 1:
        return flat if t.dim() == 1 else flat[0]
    import numpy as _np
    return _np.array(flat).reshape(tuple(t.shape)).tolist()


class Mailbox(object):
    """A mailbox of its own for a read that is POSTED now and COLLECTED later (read_back above posts and spins at once):
    `post(tensor)` enqueues the copy kernel on the current stream, `wait()` spins until it has run and returns the list.
    One post in flight per object."""

    def __init__(self, nbytes=_MB_BYTES):
        box = C.c_void_p()
        self.ok = load().aabr_mailbox_create(nbytes, C.byref(box)) == 0 and bool(box.value)
        self.box, self.seq, self.nbytes = box.value, 0, nbytes
        self.word = C.c_uint32.from_address(box.value) if self.ok else None
        self.pending = None

    def post(self, t):
        assert self.pending is None, "one post in flight per mailbox"
        ct = _CT[t.dtype]
        n = t.numel()
        if not self.ok or n * t.element_size() > self.nbytes:
            self.pending = ("sync", t)
            return
        t = t.contiguous()
        self.seq = (self.seq % 0x7fffffff) + 1
        check(load().aabr_mailbox_post(ptr(t), n * t.element_size(), self.box, self.seq, stream()))
        self.pending = ("box", t, ct, n)        # (the tensor stays alive until the kernel has read it)

    def wait(self):
        p, self.pending = self.pending, None
        if p[0] == "sync":
            return p[1].reshape(-1).tolist()
        _, t, ct, n = p
        t0, spins = None, 0
        while self.word.value != self.seq:
            spins += 1
            if spins & 0xfffff == 0:
                t0 = t0 or _time.monotonic()
                if _time.monotonic() - t0 > 120.0:
                    raise RuntimeError("aabr mailbox: no answer from the device after 120 s")
        return list((ct * n).from_address(self.box + 8))


def last_post_clock():
    """device real-time counter (100 MHz, low 32 bits) at which this thread's last read_back post ran (tools/)"""
    st = getattr(_mb, "state", None)
    return st[3] if st is not None and len(st) > 3 else None


def i32x3(v):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def i32xn(v):
    v = [int(x) for x in v]
    return (C.c_int32 * len(v))(*v)


def ptrs(ts):
    """host array of device pointers"""
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])


def f32x4(v):
    return (C.c_float * 4)(*[float(x) for x in v])


def f32xn(v):
    v = [float(x) for x in v]
    return (C.c_float * len(v))(*v)


def next_pow2(n):
    c = 64
    while c < n:
        c <<= 1
    return c
