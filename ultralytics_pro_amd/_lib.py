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


class Opts(C.Structure):
    """`upa_opts` of include/upa.h: dispatch / tuning overrides that travel with a call (all zero = production defaults).
    The library keeps no mode and reads no environment variable; a model carries its own `opts` (BaseModel.opts) and the
    parity tests / sweep tools wrap calls in `runtime.use_opts(Opts(...))`.  `Opts.from_env()` maps the UPA_* variable names
    of rounds 1-2 onto fields for the command-line tools (tools/bench_conv.py, tools/experiments)."""

    _fields_ = [("size", C.c_uint32), ("conv_big", _i), ("conv_big_bm", _i), ("conv_force", _i * 4), ("conv_ckt", _i),
                ("no_ws", _i), ("no_pipe", _i), ("no_1x1", _i), ("no_c16", _i), ("no_upcat", _i),
                ("pipe_all", _i), ("pipe_min_tiles", _i), ("pipe_wgs", _i), ("c16_wgs", _i),
                ("c1_mt", _i), ("c1_waves", _i), ("c1_wgs", _i),
                ("pair", _i), ("pair_tile64", _i), ("pair_tile32", _i), ("no_pair_cv2", _i),
                ("c2f", _i), ("c2f16_waves", _i), ("c2f32_th", _i),
                ("no_branch_tail", _i), ("branch_tail_bm", _i),
                ("stem_wgs", _i), ("stemf_wgs", _i), ("stemf_waves", _i), ("stem_no_mfma", _i),
                ("ablate_conv", _i), ("ablate_pipe", _i), ("ablate_c1", _i), ("ablate_stem", _i), ("c2f64_max_px", _i), ("conv_ws3", _i), ("no_group", _i), ("no_c2f32_up", _i), ("conv_mm", _i), ("no_xcd", _i), ("keys_only", _i), ("conv_p8", _i), ("c2f_stream", _i), ("c2f_stream_rows", _i), ("no_stack_first", _i), ("no_epi_stats", _i), ("nms_stages", _i), ("nms_first_prefix", _i), ("detect_stream", _i), ("detect_stream_rows", _i), ("no_sppf_front", _i), ("no_c2f16_down", _i)]

    def __init__(self, **kw):
        super().__init__()
        self.size = C.sizeof(Opts)
        for k, v in kw.items():
            if k == "conv_force":
                for j, x in enumerate(v):
                    self.conv_force[j] = int(x)
            elif k in dict(self._fields_):
                setattr(self, k, int(v))
            else:
                raise AttributeError(f"upa_opts has no field '{k}'")

    def replace(self, **kw) -> "Opts":
        o = Opts()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(Opts))
        for k, v in kw.items():
            if k == "conv_force":
                for j, x in enumerate(v):
                    o.conv_force[j] = int(x)
            else:
                setattr(o, k, int(v))
        return o

    # name of the round-1/2 environment switch -> (field, value transform)
    _ENV = {"UPA_CONV_BIG": ("conv_big", lambda v: {0: 1, 1: 0, 2: 2}[int(v)]), "UPA_CONV_BIG_BM": ("conv_big_bm", int), "UPA_CONV_WS3": ("conv_ws3", int),
            "UPA_CONV_CKT": ("conv_ckt", int), "UPA_CONV_NO_WS": ("no_ws", lambda v: 1), "UPA_CONV_NO_PIPE": ("no_pipe", lambda v: 1),
            "UPA_CONV_NO_1X1": ("no_1x1", lambda v: 1), "UPA_CONV_NO_C16": ("no_c16", lambda v: 1), "UPA_NO_UPCAT": ("no_upcat", lambda v: 1),
            "UPA_PIPE_ALL": ("pipe_all", lambda v: 1), "UPA_PIPE_MIN_TILES": ("pipe_min_tiles", int), "UPA_PIPE_WGS": ("pipe_wgs", int),
            "UPA_C16_WGS": ("c16_wgs", int), "UPA_C1_MT": ("c1_mt", int), "UPA_C1_WAVES": ("c1_waves", int), "UPA_C1_WGS": ("c1_wgs", int),
            "UPA_NO_PAIR": ("pair", lambda v: {0: 2, 1: 1, 2: 0, 3: 3}[int(v)]), "UPA_PAIR_T64": ("pair_tile64", int),
            "UPA_PAIR_T32": ("pair_tile32", int), "UPA_NO_PAIR_CV2": ("no_pair_cv2", int), "UPA_NO_C2F": ("c2f", int),
            "UPA_C2F16_WAVES": ("c2f16_waves", int), "UPA_C2F32_TH": ("c2f32_th", int), "UPA_NO_BRANCH_TAIL": ("no_branch_tail", int),
            "UPA_BRANCH_TAIL_BM": ("branch_tail_bm", int), "UPA_STEM_WGS": ("stem_wgs", int), "UPA_STEMF_WGS": ("stemf_wgs", int),
            "UPA_STEMF_WAVES": ("stemf_waves", int), "UPA_STEM_NO_MFMA": ("stem_no_mfma", lambda v: 1),
            "UPA_CONV_ABLATE": ("ablate_conv", int), "UPA_PIPE_ABLATE": ("ablate_pipe", int), "UPA_C1_ABLATE": ("ablate_c1", int),
            "UPA_STEM_ABLATE": ("ablate_stem", int)}

    @classmethod
    def from_env(cls, env=None) -> "Opts":
        """Tool-side convenience (never called by the product path): UPA_* variables -> an Opts."""
        env = os.environ if env is None else env
        o = cls()
        for name, (field, conv) in cls._ENV.items():
            if name in env:
                setattr(o, field, conv(env[name]))
        if "UPA_CONV_FORCE" in env:
            for j, x in enumerate(env["UPA_CONV_FORCE"].split(",")[:4]):
                o.conv_force[j] = int(x)
        return o


class BranchLevel(C.Structure):
    """`upa_branch_level` (upa_detect_branch_tail_group)."""
    _fields_ = [("x", C.c_void_p), ("n", _i), ("h", _i), ("w", _i), ("c", _i), ("ldx", _i), ("w3_packed", C.c_void_p), ("b3", C.c_void_p),
                ("wt_packed", C.c_void_p), ("bt", C.c_void_p), ("stride_px", C.c_float), ("a0", _i)]


class DetectBranch(C.Structure):
    """`upa_detect_branch` (upa_detect_level_stream)."""
    _fields_ = [("c", _i), ("reserved", _i), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("wt", C.c_void_p), ("bt", C.c_void_p)]


class ConvProblem(C.Structure):
    """`upa_conv_problem` (upa_conv2d_bias_act_group)."""
    _fields_ = [("x", C.c_void_p), ("n", _i), ("h", _i), ("w", _i), ("cin", _i), ("ldx", _i), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("cout", _i), ("ldy", _i), ("residual", C.c_void_p), ("ldr", _i)]


_op = C.POINTER(Opts)

# name -> (restype, argtypes); must list every prototype of include/upa.h (tests/test_abi.py checks this)
PROTOTYPES = {
    "upa_version": (_i, []),
    "upa_last_error": (C.c_char_p, []),
    "upa_opts_size": (_sz, []),
    "upa_conv_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "upa_pack_conv_weight": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "upa_conv2d_bias_act": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _op, _vp]),
    "upa_conv_variant": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _op]),
    "upa_bottleneck_pair": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_bottleneck_pair_e": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_stem_packed_weight_bytes": (_sz, [_i, _i, _i]),
    "upa_pack_stem_weight": (_i, [_vp, _i, _i, _i, _vp]),
    "upa_conv2d_stem_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _op, _vp]),
    "upa_conv2d_stem_nchw_pool2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _op, _vp]),
    "upa_conv2d_pool2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _op, _vp]),
    "upa_stem_conv_fused": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _op, _vp]),
    "upa_stem_conv_fused_k": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _op, _vp]),
    "upa_stem_conv_fused_c": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _op, _vp]),
    "upa_stem_conv_fused_s": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _op, _vp]),
    "upa_maxpool2d": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_sppf_pool3": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "upa_upsample2x": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_copy_view": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_copy_to_host": (_i, [_vp, _vp, _sz, _vp]),
    "upa_results_to_host": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "upa_add_view": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "upa_nhwc_to_nchw": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "upa_letterbox_u8": (_i, [_vp, _i, _i, _i, C.c_long, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "upa_detect_decode": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _i, _i, _vp]),
    "upa_detect_branch_tail_group": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _op, _vp]),
    "upa_detect_head_tails": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _op, _vp]),
    "upa_conv2d_bias_act_group": (_i, [_vp, _i, _i, _i, _i, _i, _i, _op, _vp]),
    "upa_detect_level_stream": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _i, _vp, _i, _op, _vp]),
    "upa_detect_tail": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _vp, _i, _vp, _i, _op, _vp]),
    "upa_conv1x1_upcat": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_bottleneck_pair_cv2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _op, _vp]),
    "upa_sppf_front": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _op, _vp]),
    "upa_c2f16_down_fused": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _op, _vp]),
    "upa_c2f_fused": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_c2f64_fused": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_c2f32_up_fused": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _op, _vp]),
    "upa_tail_packed_weight_bytes": (_sz, [_i, _i]),
    "upa_pack_tail_weight": (_i, [_vp, _i, _i, _vp]),
    "upa_detect_branch_tail": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _i, _vp, _i, _op, _vp]),
    "upa_nms_batched_hot": (_i, [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "upa_nms_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "upa_nms_batched": (_i, [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "upa_nms_batched_opts": (_i, [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "upa_mhsa": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _i, _vp]),
    "upa_linear": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "upa_linear_bf16": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "upa_linear_mixed": (_i, [_vp, _i, C.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
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
    "upa_conv2d_bn_act_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _i,
                                    _vp, _i, _vp, _vp]),
    "upa_conv2d_bn_stats": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "upa_bn_act_fwd": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _i, _i, _vp]),
    "upa_bn_act_bwd": (_i, [_vp, _vp, C.c_long, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "upa_conv_bn_act_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                  _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "upa_conv2d_dgrad_s2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
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
    "upa_detection_loss_scaled": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "upa_sgd_nesterov_ema_scaled": (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp, _f, _f, _f, _f, _i, _f, _vp, _i, _vp, _vp]),
    "upa_grad_scaler_update": (_i, [_vp, _vp, _f, _f, _i, _vp]),
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
        if handle.upa_opts_size() != C.sizeof(Opts):
            raise UpaError(f"{LIB_PATH}: upa_opts is {handle.upa_opts_size()} bytes, the binding's mirror {C.sizeof(Opts)} - rebuild the library")
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
