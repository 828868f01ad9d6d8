"""ctypes binding of libpixelpick_hip.so (the C ABI in include/pixelpick_hip.h).

The library is the product: if it is missing this module raises — there is no CPU/eager fallback.
torch is imported first so that the HIP runtime already loaded by PyTorch-ROCm (same soname,
libamdhip64.so.7) is the one our kernels, streams and pointers live in.
"""
import ctypes
import os
import struct

import torch  # noqa: F401  (loads PyTorch's libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpixelpick_hip.so")                # the product: no pp_debug_* symbol, planners on their defaults
KNOBS_LIB_PATH = os.path.join(_HERE, "libpixelpick_hip_knobs.so")    # the test build: same sources + -DPP_DEBUG_KNOBS
_use_knobs = [os.environ.get("PIXELPICK_KNOBS_BUILD", "0") not in ("", "0")]


def use_knobs_build(on: bool = True):
    """Select the test build (planner switches `pp_debug_*`, experiment kernels) for this process.  Must run before the first lib():
    one process holds ONE library (two copies would each keep their own launch epochs and exchange tags)."""
    if _lib is not None and _use_knobs[0] != bool(on):
        raise PixelPickHipError("use_knobs_build() after the library was loaded")
    _use_knobs[0] = bool(on)


def knobs_build() -> bool:
    return _use_knobs[0]

_i64, _p, _int, _sz, _f = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must list every symbol include/pixelpick_hip.h declares
SIGNATURES = {
    "pp_version": (_int, []),
    "pp_last_error": (ctypes.c_char_p, []),
    "pp_acq_workspace_bytes": (_sz, [_i64] * 5),
    "pp_acq_score_topk": (_int, [_p] + [_i64] * 8 + [_p, _int, _i64, _p, _p, _p, _p, _sz, _p]),
    "pp_acq_score_map": (_int, [_p] + [_i64] * 8 + [_p, _int, _p, _p]),
    "pp_acq_softmax_sum": (_int, [_p] + [_i64] * 8 + [_p, _p, _int, _f, _int, _p]),
    "pp_uncertainty_from_prob": (_int, [_p] + [_i64] * 8 + [_int, _p, _p]),
    "pp_topk_workspace_bytes": (_sz, [_i64] * 3),
    "pp_topk_select": (_int, [_p, _i64, _i64, _i64, _int, _p, _p, _p, _sz, _p]),
    "pp_acq_lowres_workspace_bytes": (_sz, [_i64] * 5),
    "pp_acq_lowres_score_topk": (_int, [_p] + [_i64] * 7 + [_int, _i64, _i64, _p, _int, _i64, _p, _p, _p, _p, _sz, _p]),
    "pp_acq_lowres_score_at": (_int, [_p] + [_i64] * 7 + [_int, _i64, _i64, _int, _p, _p, _i64, _p, _p]),
    "pp_conv2d_fwd_workspace_bytes": (_sz, [_int] * 10),
    "pp_conv2d_bwd_data_workspace_bytes": (_sz, [_int] * 10),
    "pp_conv2d_fwd": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _p, _sz, _p]),
    "pp_conv2d_fwd_stats_rows": (_i64, [_int] * 10),
    "pp_conv2d_fwd_stats": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _p, _sz, _p, _sz, _p]),
    "pp_bn_train_fwd_partials": (_int, [_p, _i64, _i64, _int, _p, _i64, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i64, _int, _f, _u64, _p, _p, _i64, _p]),
    "pp_conv2d_fwd_bn_act": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _int, _int, _int, _int, _p, _p, _p, _p, _f, _p, _i64, _int,
                                    _p, _i64, _int, _p, _sz, _p]),
    "pp_dwconv3x3_fwd_bn_act": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _p, _p, _p, _p, _f, _p, _i64, _int, _p, _i64, _p]),
    "pp_bn_finalize_partials": (_int, [_p, _i64, _i64, _int, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "pp_dwconv3x3_fwd_stats_rows": (_i64, [_int] * 7),
    "pp_dwconv3x3_fwd_fused": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _p, _p, _int, _p, _i64, _p, _sz, _p]),
    "pp_dwconv3x3_bwd_weight_affine_in": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _p, _i64, _int, _int, _int, _p, _p, _sz, _p]),
    "pp_conv2d_fwd_accepts_affine_in": (_int, [_int] * 10),
    "pp_conv2d_fwd_affine_in": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _p, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _p, _sz, _p]),
    "pp_conv2d_bwd_data": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _p, _sz, _p]),
    "pp_conv2d_bwd_weight_workspace_bytes": (_sz, [_int] * 10),
    "pp_conv2d_bwd_weight": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _int, _int, _p, _p, _p, _sz, _p]),
    "pp_conv2d_fwd_bn_train_ok": (_int, [_int] * 10),
    "pp_conv2d_fwd_bn_train_xchg_bytes": (_sz, [_int] * 10),
    "pp_conv2d_fwd_bn_train": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _int, _int, _p, _i64, _p, _p, _f, _f, _p, _p, _p, _p,
                                      _p, _i64, _int, _p, _i64, _int, _p, _sz, _p, _sz, _p]),
    "pp_conv2d_bwd_data_bn_bwd_ok": (_int, [_int] * 10),
    "pp_conv2d_bwd_data_bn_bwd_xchg_bytes": (_sz, [_int] * 10),
    "pp_conv2d_bwd_data_bn_bwd": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _int, _int, _int, _int, _int, _p, _i64, _p, _p, _p, _p,
                                         _int, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _sz, _p, _sz, _p]),
    "pp_x3_planes_bytes": (_sz, [_i64, _int]),
    "pp_x3_split": (_int, [_p, _i64, _i64, _int, _p, _sz, _p]),
    "pp_conv2d_x3_planes_bytes": (_sz, [_int] * 11),
    "pp_conv2d_fwd_pre": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _p, _sz, _p, _p]),
    "pp_conv2d_bwd_data_multi_workspace_bytes": (_sz, [_int] * 14),
    "pp_conv2d_bwd_data_multi": (_int, [_p, _i64, _int, _int, _int, _int, _int, _p, _int, _int, _p, _int, _int, _p, _int, _int, _p, _int, _int,
                                       _p, _i64, _int, _int, _p, _sz, _p]),
    "pp_conv2d_bwd_data_pre": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _p, _sz, _p, _p]),
    "pp_x3_weight_planes_bytes": (_sz, [_int, _int, _int, _int]),
    "pp_x3_split_weights": (_int, [_p, _int, _int, _int, _int, _p, _sz, _p]),
    "pp_conv2d_fwd_pre2": (_int, [_p, _i64, _int, _int, _int, _int, _p, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _p, _sz, _p, _p, _p]),
    "pp_conv2d_bwd_data_pre2": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _p, _sz, _p, _p, _p]),
    "pp_conv2d_bwd_weight_pre": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _int, _int, _p, _p, _p, _sz, _p, _p, _p]),
    "pp_conv2d_bwd_weight_partials": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _int, _int, _int, _p, _p, _p, _sz, _p, _p]),
    "pp_wgrad_reduce_batch": (_int, [_p, _int, _p]),
    "pp_colreduce_workspace_bytes": (_sz, [_i64, _int]),
    "pp_bn_train_fwd": (_int, [_p, _i64, _i64, _int, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pp_bn_eval_affine": (_int, [_int, _p, _p, _p, _p, _f, _p, _p, _p]),
    "pp_scale_shift_act": (_int, [_p, _i64, _i64, _int, _p, _p, _p, _i64, _int, _p, _i64, _p]),
    "pp_bn_bwd": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _i64, _int, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "pp_bn_fused_workspace_bytes": (_sz, [_i64, _int]),
    "pp_bn_fused_sync_ints": (_sz, [_int]),
    "pp_bn_fused_rows_cached": (_int, [_i64, _int]),
    "pp_bn_train_fwd_fused": (_int, [_p, _i64, _i64, _int, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i64, _int, _f, _u64, _p, _p, _i64, _p, _sz, _p, _sz, _p]),
    "pp_dwconv3x3_bn_train_fwd_fused": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _p, _i64, _p, _p, _f, _f, _p, _p, _p, _p,
                                               _p, _i64, _int, _p, _i64, _p, _sz, _p, _sz, _p]),
    "pp_bn_bwd_fused": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _i64, _int, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _f, _p, _p, _sz, _p, _sz, _p]),
    "pp_bn_bwd_fused_sparse": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _i64, _int, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _f, _p, _p, _sz, _p, _sz, _p, _p]),
    "pp_row_flags": (_int, [_p, _i64, _i64, _int, _p, _p]),
    "pp_conv1x1_bwd_weight_sparse_workspace_bytes": (_sz, [_i64, _int, _int]),
    "pp_conv1x1_bwd_weight_sparse": (_int, [_p, _i64, _i64, _int, _p, _i64, _int, _p, _p, _p, _p, _sz, _p]),
    "pp_conv1x1_bwd_data_sparse": (_int, [_p, _i64, _i64, _int, _p, _int, _p, _p, _i64, _p]),
    "pp_dwconv3x3_fwd": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _p, _i64, _p]),
    "pp_dwconv3x3_bwd_data": (_int, [_p, _i64, _int, _int, _int, _int, _p, _int, _int, _int, _p, _i64, _p]),
    "pp_dwconv3x3_bwd_weight": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _p, _p, _sz, _p]),
    "pp_dwconv3x3_bwd_weight_partials": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _p, _p, _sz, _p, _p]),
    "pp_groupnorm_workspace_bytes": (_sz, [_int, _i64, _int]),
    "pp_groupnorm_relu_fwd": (_int, [_p, _i64, _int, _i64, _int, _int, _p, _p, _f, _int, _p, _i64, _p, _p, _p, _sz, _p]),
    "pp_groupnorm_relu_bwd": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _i64, _int, _int, _p, _p, _p, _p, _p, _p, _i64, _p, _sz, _p]),
    "pp_maxpool2d_fwd": (_int, [_p, _i64, _int, _int, _int, _int, _int, _int, _int, _p, _i64, _p, _p]),
    "pp_maxpool2d_bwd": (_int, [_p, _i64, _p, _int, _int, _int, _int, _int, _int, _int, _p, _i64, _p]),
    "pp_pad2d": (_int, [_p, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _p, _i64, _p]),
    "pp_crop2d_add": (_int, [_p, _i64, _int, _int, _int, _int, _int, _int, _p, _i64, _p, _i64, _int, _int, _p]),
    "pp_bilinear_fwd": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _f, _f, _int, _p]),
    "pp_bilinear_bwd_workspace_bytes": (_sz, [_int] * 4),
    "pp_bilinear_bwd": (_int, [_p, _i64, _int, _int, _int, _int, _p, _i64, _int, _int, _int, _f, _f, _int, _p, _sz, _p]),
    "pp_image_colsum": (_int, [_p, _i64, _int, _i64, _int, _f, _p, _i64, _p]),
    "pp_image_broadcast": (_int, [_p, _i64, _int, _i64, _int, _f, _p, _i64, _p]),
    "pp_dropout": (_int, [_p, _i64, _p, _i64, _i64, _int, _f, ctypes.c_uint64, _p, _p]),
    "pp_dropout2d": (_int, [_p, _i64, _p, _i64, _int, _i64, _int, _f, ctypes.c_uint64, _p, _p]),
    "pp_aug_resample_h": (_int, [_p, _int, _int, _p, _p, _int, _int, _p, _p]),
    "pp_aug_vcrop": (_int, [_p, _p, _p] + [_int] * 11 + [_p, _p]),
    "pp_aug_labels": (_int, [_p, _p, _int, _p, _p, _p, _p] + [_int] * 8 + [_p, _p, _p]),
    "pp_aug_jitter": (_int, [_p, _i64, _int, _f, _p, _p]),
    "pp_aug_blur": (_int, [_p, _int, _int, _p, _int, _p, _p]),
    "pp_aug_blur_q8": (_int, [_p, _int, _int, _p, _int, _p, _p]),
    "pp_aug_to_tensor": (_int, [_p, _i64, ctypes.POINTER(_f), ctypes.POINTER(_f), _p, _p]),
    "pp_sparse_ce_workspace_bytes": (_sz, []),
    "pp_sparse_ce_fwd_bwd": (_int, [_p, _int, _int, _i64, _i64, _i64, _p, _int, _p, _p, _p, _p, _p, _sz, _p]),
    "pp_sparse_ce_lowres_workspace_bytes": (_sz, []),
    "pp_sparse_ce_lowres_fwd_bwd": (_int, [_p, _i64] + [_int] * 7 + [_p, _int, _p, _p, _p, _p, _i64, _p, _sz, _p]),
    "pp_confusion_matrix_update": (_int, [_p, _int, _int, _i64, _i64, _i64, _p, _p, _p]),
    "pp_adam_step_flat": (_int, [_p, _p, _p, _p, _i64, _i64, _f, _f, _f, _f, _f, _f, _i64, _f, _p, _p]),
    "pp_sgd_step_flat": (_int, [_p, _p, _p, _i64, _i64, _f, _f, _f, _f, _i64, _f, _p, _p]),
    "pp_add2d": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _int, _p]),
    "pp_nhwc_to_nchw": (_int, [_p, _i64, _int, _int, _i64, _p, _p]),
    "pp_nchw_to_nhwc": (_int, [_p, _int, _int, _i64, _p, _i64, _p]),
    "pp_bn_fused_capacity": (_int, []),
    "pp_yardstick_stream_read": (_int, [_p, _sz, _int, _p, _p]),
    "pp_yardstick_mfma_stream": (_int, [_int, _int, _p, _p]),
    "pp_set_kernel_events": (None, [_p, _p, _int]),
    "pp_set_comm_cu_reserve": (None, [_int]),
    "pp_get_comm_cu_reserve": (_int, []),
    "pp_plan_create": (_p, []),
    "pp_plan_destroy": (None, [_p]),
    "pp_plan_size": (_i64, [_p]),
    "pp_plan_entry_args": (_int, [_p]),
    "pp_plan_add_call": (_int, [_p, _p, _p, _int]),
    "pp_plan_add_event_record": (_int, [_p, _p, _p]),
    "pp_plan_add_stream_wait": (_int, [_p, _p, _p]),
    "pp_plan_add_join": (_int, [_p, _p, _p]),
    "pp_plan_add_host_break": (_int, [_p]),
    "pp_plan_replay": (_int, [_p, _i64, ctypes.POINTER(_i64)]),
}

# The test build's planner switches (include/pixelpick_hip_knobs.h): exported by libpixelpick_hip_knobs.so only
KNOB_SIGNATURES = {
    "pp_debug_set_reduce_mode": (None, [_int]),
    "pp_debug_set_exact_formula": (None, [_int]),
    "pp_debug_set_acq_tuning": (None, [_int, _int]),
    "pp_debug_set_dw_variant": (None, [_int]),
    "pp_debug_set_splitk": (None, [_int]),
    "pp_debug_set_wgrad_target": (None, [_int]),
    "pp_debug_set_bn_target": (None, [_int]),
    "pp_debug_set_bn_bytes_per_block": (None, [_int]),
    "pp_debug_set_bn_probe": (None, [_p]),
    "pp_debug_set_conv_thresholds": (None, [_int]),
    "pp_debug_conv_plan": (None, [_i64, _int, _int, _int, _p]),
    "pp_debug_set_conv_variant": (None, [_int]),
    "pp_debug_set_conv_rows": (None, [_int]),
    "pp_debug_set_conv_bn_fuse": (None, [_int]),
    "pp_debug_set_x3": (None, [_int]),
    "pp_debug_set_x3f": (None, [_int]),
    "pp_debug_set_gemm_pw": (None, [_int]),
    "pp_debug_set_x3_variant": (None, [_int]),
    "pp_debug_occupy_cus": (_int, [_int, _p, _u64, _p, _p]),
}

_lib = None
# Every pp_debug_set_* knob can change what the library plans for a shape; host-side memo tables of plan-dependent answers
# (engine._wsbytes, _conv_ws, conv_accepts_lazy_input) are keyed on this epoch and refilled after a knob moved.
knob_epoch = [0]


def _knob_setter(fn):
    def setter(*args):
        knob_epoch[0] += 1
        return fn(*args)
    setter.__name__ = getattr(fn, "__name__", "pp_debug_set")
    setter.__wrapped__ = fn
    return setter


class PixelPickHipError(RuntimeError):
    pass


def _preload_torch_hip():
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load (once) and return the shared library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        path = KNOBS_LIB_PATH if _use_knobs[0] else LIB_PATH
        if not os.path.exists(path):
            raise PixelPickHipError(
                f"{path} not found: build the HIP extension first "
                f"(python -m pixelpick_amd.build, or __graft_entry__.build()). No fallback path exists.")
        _preload_torch_hip()
        L = ctypes.CDLL(path)
        sigs = dict(SIGNATURES)
        if _use_knobs[0]:
            sigs.update(KNOB_SIGNATURES)
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
            if name.startswith(("pp_debug_set_", "pp_set_comm")):
                setattr(L, name, _knob_setter(fn))
        _lib = L
    return _recording[0] or _lib


# ------------------------------------------------------------------------------------------------ launch plans
# A train step is ~415 launches through this binding plus ~110 stream fork / join operations, each with a few microseconds of
# Python around it (shape arithmetic, tape bookkeeping, workspace look-ups): 5.4 ms of host time for a 6.85 ms step.  With
# static shapes and stable addresses (FlatTrainer.enable_replay runs the step inside a private torch memory pool) every one
# of those calls repeats with identical arguments, so the step is recorded ONCE as a flat list of (function, converted
# arguments) and re-issued from a tight loop.  Unlike a hipGraph the replay keeps the two-queue eager schedule (main stream +
# weight-gradient stream, the all-reduce under the encoder backward) - profiles/r02_graph_replay.txt shows why the graph loses.
_NOT_LAUNCHES = ("pp_version", "pp_last_error", "pp_bn_fused_capacity", "pp_bn_fused_rows_cached", "pp_set_comm_cu_reserve",
                 "pp_get_comm_cu_reserve")


def _is_launch(name: str) -> bool:
    return not (name in _NOT_LAUNCHES or name.startswith(("pp_debug_", "pp_plan_", "pp_set_", "pp_yardstick_"))
                or name.endswith(("_bytes", "_rows", "_ints", "_accepts_affine_in", "_ok")))


class ReduceJob(ctypes.Structure):
    """pp_reduce_job (include/pixelpick_hip.h): a deferred weight-gradient reduce."""
    _fields_ = [("part", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("cn", ctypes.c_int64), ("splits", ctypes.c_int32),
                ("ntaps", ctypes.c_int32), ("kind", ctypes.c_int32), ("widx", ctypes.c_uint8 * 12)]


class LaunchPlan:
    """The recorded step: `calls` = [(callable, args tuple)].  C-ABI entries return 0 on success, stream operations None."""
    __slots__ = ("calls",)

    def __init__(self):
        self.calls = []

    def replay(self):
        for fn, args in self.calls:
            if fn(*args):
                raise PixelPickHipError(f"{getattr(fn, '__name__', fn)} failed during replay: "
                                        f"{lib().pp_last_error().decode('utf-8', 'replace')}")

    def __len__(self):
        return len(self.calls)


def _slot(v) -> int:
    """One argument as the 8-byte slot pp_plan_add_call takes (see include/pixelpick_hip.h)."""
    if isinstance(v, ctypes.c_float):
        return struct.unpack("<I", struct.pack("<f", v.value))[0]
    if isinstance(v, ctypes.c_void_p):
        return (v.value or 0) & 0xFFFFFFFFFFFFFFFF
    if isinstance(v, ctypes._SimpleCData):
        return int(v.value) & 0xFFFFFFFFFFFFFFFF
    if isinstance(v, (ctypes._Pointer, ctypes.Array)):
        return (ctypes.cast(v, ctypes.c_void_p).value or 0) & 0xFFFFFFFFFFFFFFFF
    raise TypeError(f"cannot record an argument of type {type(v)}")


class NativePlan(LaunchPlan):
    """The recorded step held by the library (pp_plan_*, csrc/plan.hip): replay() is ONE foreign call per stretch between host
    breaks - none at all in a single-rank step - instead of one per launch.  The Python list of the base class is kept beside it
    (same entries, same order): it keeps the recorded arguments alive and `replay_python()` re-issues it for A/B and tests."""
    __slots__ = ("handle", "breaks", "host_notes", "_next", "_keep", "_ops")

    def __init__(self):
        super().__init__()
        L = lib_real()
        self.handle = L.pp_plan_create()
        if not self.handle:
            raise PixelPickHipError("pp_plan_create failed")
        self.breaks = {}            # op index of a host break -> (callable, args)
        self.host_notes = []        # plan_note_host(): host-only counters, run after the native loop
        self._next = ctypes.c_int64(0)
        self._keep = []
        self._ops = []              # recorded ops, handed to the library by finalize() (record_plan.__exit__)

    def add_call(self, fn, cargs):
        self._ops.append(("call", fn, cargs))

    def add_note(self, fn, args):
        self._ops.append(("note", fn, args))

    # PIXELPICK_PLAN_SIDE_DELAY=K (experiment, default 0): the launches of the weight-gradient queue are handed to the runtime K
    # main-queue calls later than they were recorded (the event their queue waits for stays where it was recorded, so the
    # dependency is the same; the join flushes everything).  Eager execution issues them "late" by construction - the host is
    # slower than the GPU - and runs ~1 % faster on the GPU than a replay that issues the whole step in 1.5 ms.
    def finalize(self, main_stream: int):
        delay = int(os.environ.get("PIXELPICK_PLAN_SIDE_DELAY", "0"))
        ops, self._ops = self._ops, []
        if delay <= 0:
            for op in ops:
                self._emit(op)
            return
        pending = []                                   # [release countdown, op]

        def flush(all_=False):
            while pending and (all_ or pending[0][0] <= 0):
                self._emit(pending.pop(0)[1])
        for op in ops:
            kind, fn, args = op
            side = False
            if kind == "call":
                st = args[-1]
                side = isinstance(st, ctypes.c_void_p) and (st.value or 0) != main_stream
            else:
                owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
                if isinstance(owner, torch.cuda.Stream) and name == "wait_event" and owner.cuda_stream != main_stream:
                    side = True
                elif not (isinstance(owner, torch.cuda.Event) and name == "record"):
                    flush(True)                         # joins, host breaks: everything recorded before them goes first
            if side:
                pending.append([delay, op])
            else:
                self._emit(op)
                for q in pending:
                    q[0] -= 1
                flush()
        flush(True)

    def _emit(self, op):
        kind, fn, args = op
        if kind == "call":
            self._emit_call(fn, args)
        else:
            self._emit_note(fn, args)

    def _emit_call(self, fn, cargs):
        L = lib_real()
        n = len(cargs)
        slots = (ctypes.c_uint64 * max(n, 1))(*[_slot(a) for a in cargs])
        check(L.pp_plan_add_call(self.handle, ctypes.cast(fn, ctypes.c_void_p), slots, n), "pp_plan_add_call")

    def _emit_note(self, fn, args):
        """A stream operation of torch (Event.record, Stream.wait_event, Stream.wait_stream) becomes a native op; anything else is
        a host break: the native loop returns there, the callable runs in Python, the loop resumes."""
        L = lib_real()
        owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
        if isinstance(owner, torch.cuda.Event) and name == "record" and len(args) == 1 and isinstance(args[0], torch.cuda.Stream):
            check(L.pp_plan_add_event_record(self.handle, owner.cuda_event, args[0].cuda_stream), "pp_plan_add_event_record")
        elif isinstance(owner, torch.cuda.Stream) and name == "wait_event" and len(args) == 1 and isinstance(args[0], torch.cuda.Event):
            check(L.pp_plan_add_stream_wait(self.handle, owner.cuda_stream, args[0].cuda_event), "pp_plan_add_stream_wait")
        elif isinstance(owner, torch.cuda.Stream) and name == "wait_stream" and len(args) == 1 and isinstance(args[0], torch.cuda.Stream):
            check(L.pp_plan_add_join(self.handle, owner.cuda_stream, args[0].cuda_stream), "pp_plan_add_join")
        else:
            check(L.pp_plan_add_host_break(self.handle), "pp_plan_add_host_break")
            self.breaks[int(L.pp_plan_size(self.handle))] = (fn, args)      # keyed by the index the loop resumes from
            return
        self._keep.append((owner, args))

    def replay(self):
        L = lib_real()
        h, nxt, n = self.handle, self._next, int(L.pp_plan_size(self.handle))
        i = 0
        while True:
            rc = L.pp_plan_replay(h, i, ctypes.byref(nxt))
            if rc:
                raise PixelPickHipError(f"launch-plan replay failed at op {nxt.value}: {L.pp_last_error().decode('utf-8', 'replace')}")
            i = nxt.value
            if i in self.breaks:
                fn, args = self.breaks[i]
                fn(*args)
            if i >= n:
                break
        for fn, args in self.host_notes:
            fn(*args)

    def replay_python(self):
        LaunchPlan.replay(self)

    def native_ops(self) -> int:
        return int(lib_real().pp_plan_size(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            lib_real().pp_plan_destroy(self.handle)
            self.handle = None
        self.calls.clear()
        self.breaks.clear()
        self.host_notes.clear()
        self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lib_real():
    """The library itself, also while a plan is being recorded."""
    lib()
    return _lib


class _RecordingLib:
    """What lib() returns while a plan is being recorded: every launch is executed AND appended to the plan."""

    def __init__(self, real, plan):
        self.__dict__["_real"] = real
        self.__dict__["_plan"] = plan

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not _is_launch(name):
            return fn
        calls, argtypes = self._plan.calls, fn.argtypes
        native = self._plan if isinstance(self._plan, NativePlan) else None

        def rec(*args):
            rc = fn(*args)
            if rc == 0:
                # converted once: ctypes passes instances of the declared types straight through
                cargs = tuple(a if isinstance(a, (ctypes._SimpleCData, ctypes._Pointer, ctypes.Array)) else t(a)
                              for t, a in zip(argtypes, args))
                calls.append((fn, cargs))
                if native is not None:
                    native.add_call(fn, cargs)
            return rc

        self.__dict__[name] = rec
        return rec


_recording = [None]


class record_plan:
    """with record_plan() as plan: ...   (not re-entrant, calling thread only).  native=True (default): a NativePlan, replayed by
    the library's own loop; native=False: the Python list only (round-2 form, kept for A/B)."""

    def __init__(self, native: bool = True):
        self.native = native

    def __enter__(self):
        if _recording[0] is not None:
            raise RuntimeError("a launch plan is already being recorded")
        plan = NativePlan() if self.native else LaunchPlan()
        _recording[0] = _RecordingLib(lib(), plan)
        return plan

    def __exit__(self, *exc):
        rec = _recording[0]
        _recording[0] = None
        if exc[0] is None and rec is not None and isinstance(rec._plan, NativePlan):
            rec._plan.finalize(current_stream_ptr())
        return False


def plan_note(fn, *args):
    """Run fn(*args) (a stream fork / join, a collective, a host-side counter: must return None) and, while a plan is being
    recorded, make it part of the replay."""
    rec = _recording[0]
    if rec is not None:
        rec._plan.calls.append((fn, args))
        if isinstance(rec._plan, NativePlan):
            out = fn(*args)             # first: torch creates an event's handle at its first record
            rec._plan.add_note(fn, args)
            return out
    return fn(*args)


def plan_note_host(fn, *args):
    """plan_note for a host-only side effect whose position among the launches does not matter (a step counter): a NativePlan
    runs these after its native loop instead of returning to Python in the middle of the step for each of them."""
    rec = _recording[0]
    if rec is not None:
        rec._plan.calls.append((fn, args))
        if isinstance(rec._plan, NativePlan):
            rec._plan.host_notes.append((fn, args))
    return fn(*args)


def recording() -> bool:
    return _recording[0] is not None


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().pp_last_error().decode("utf-8", "replace")
        if rc in (-1, -2):
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise PixelPickHipError(f"{what}: {msg} (code {rc})")


def current_stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
