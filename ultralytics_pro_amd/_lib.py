"""ctypes binding of libupa_hip.so (the C ABI declared in include/upa.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
`import torch` must precede the dlopen so that the library binds to the HIP runtime torch already loaded
(one runtime per process: streams and device pointers are shared with torch's allocator).
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  (loads libamdhip64.so.7 first - see module docstring)

UPA_F32, UPA_BF16, UPA_U8_BGR_HWC = 0, 1, 2
UPA_EUNSUPPORTED = -2
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("UPA_HIP_LIB", _PKG / "libupa_hip.so"))

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every prototype of include/upa.h (tests/test_abi.py checks this)
PROTOTYPES = {
    "upa_version": (_i, []),
    "upa_last_error": (C.c_char_p, []),
    "upa_conv_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "upa_pack_conv_weight": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "upa_conv2d_bias_act": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_conv_variant": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "upa_conv_big_mode": (_i, [_i]),
    "upa_bottleneck_pair": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "upa_stem_packed_weight_bytes": (_sz, [_i, _i, _i]),
    "upa_pack_stem_weight": (_i, [_vp, _i, _i, _i, _vp]),
    "upa_conv2d_stem_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_stem_conv_fused": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "upa_maxpool2d": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_sppf_pool3": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "upa_upsample2x": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_copy_view": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_add_view": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_nhwc_to_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "upa_letterbox_u8": (_i, [_vp, _i, _i, _i, C.c_long, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_detect_decode": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp]),
    "upa_detect_tail": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _vp, _i, _i, _vp]),
    "upa_conv1x1_upcat": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "upa_bottleneck_pair_cv2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "upa_c2f_fused": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "upa_tail_packed_weight_bytes": (_sz, [_i, _i]),
    "upa_pack_tail_weight": (_i, [_vp, _i, _i, _vp]),
    "upa_detect_branch_tail": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _i, _vp, _i, _vp]),
    "upa_nms_batched_hot": (_i, [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "upa_nms_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "upa_nms_batched": (_i, [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "upa_mhsa": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _i, _vp]),
    "upa_linear": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "upa_layer_norm": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _vp, _vp]),
    "upa_rows_add": (_i, [_vp, _vp, _vp, C.c_long, _i, _vp]),
    "upa_rows_scale": (_i, [_vp, _vp, _vp, C.c_long, _i, _vp]),
    "upa_rows_gather": (_i, [_vp, _vp, _vp, C.c_long, _i, _vp]),
    "upa_topk_tokens": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "upa_box_refine": (_i, [_vp, _vp, _vp, C.c_long, _vp]),
    "upa_box_add_anchors": (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp]),
    "upa_sigmoid": (_i, [_vp, _vp, C.c_long, _vp]),
    "upa_rtdetr_output": (_i, [_vp, _vp, _vp, C.c_long, _i, _vp]),
    "upa_rtdetr_postprocess": (_i, [_vp, _i, _i, _i, _f, _vp, _i, _vp, _f, _f, _vp, _vp, _vp]),
    "upa_msdeform_attn": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "upa_msdeform_attn_strided": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "upa_box_iou": (_i, [_vp, _i, _vp, _i, _f, _vp, _vp]),
    "upa_match_predictions": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "upa_scale_boxes": (_i, [_vp, C.c_long, _i, _f, _f, _f, _i, _f, _f, _vp]),
    "upa_pack_conv_weight_dev": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "upa_channel_reduce_workspace_bytes": (_sz, [_i]),
    "upa_bn_stats": (_i, [_vp, C.c_long, _i, _i, _vp, _i, _vp]),
    "upa_bn_finalize": (_i, [_vp, C.c_long, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "upa_bn_act_fwd": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _i, _i, _vp]),
    "upa_bn_act_bwd": (_i, [_vp, _vp, C.c_long, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "upa_channel_sum": (_i, [_vp, C.c_long, _i, _i, _vp, _i, _vp, _i, _vp]),
    "upa_conv2d_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "upa_conv2d_wgrad": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "upa_dilate2x": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "upa_dgrad_s2_phase_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "upa_interleave2x": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "upa_upsample2x_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "upa_maxpool2d_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "upa_maxpool2d_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "upa_pack_conv_weights_batched": (_i, [_vp, _i, _vp]),
    "upa_sumsq_workspace_bytes": (C.c_size_t, []),
    "upa_sumsq": (_i, [_vp, C.c_long, _vp, _i, _vp, _vp]),
    "upa_sgd_nesterov_ema": (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp, _f, _f, _f, _f, _i, _f, _vp, _i, _vp]),
    "upa_ema_update": (_i, [_vp, _vp, C.c_long, _f, _vp, _vp]),
    "upa_cast_view": (_i, [_vp, _i, _i, _vp, _i, _i, C.c_long, _i, _vp]),
    "upa_detection_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "upa_detection_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _sz, _vp]),
    "upa_graph_begin": (_i, [_vp]),
    "upa_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "upa_graph_launch": (_i, [_vp, _vp]),
    "upa_graph_destroy": (_i, [_vp]),
}

_lib = None


class UpaError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raises loudly when it is absent - there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.is_file():
            raise UpaError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C ultralytics_pro_amd/csrc). The HIP path has no fallback.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().upa_last_error().decode(errors="replace")
        raise UpaError(f"{what or 'upa call'} failed (rc={rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return UPA_F32
    if dt == torch.bfloat16:
        return UPA_BF16
    raise UpaError(f"unsupported activation dtype {dt}; use torch.float32 (parity) or torch.bfloat16 (perf)")


def current_stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise UpaError(f"{what}: tensor is on {t.device}; the HIP path only runs on an MI355X (cuda:N) - "
                       "there is deliberately no CPU fallback")
