"""ctypes binding of libzutis_hip.so (include/zutis_hip.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZUTIS_HIP_LIB") or os.path.join(HERE, "libzutis_hip.so")   # override: developer A/B builds only
HEADER = os.path.join(os.path.dirname(HERE), "include", "zutis_hip.h")

_lib = None
RECORDER = None   # zutis_amd.plan.Recorder while a launch plan is being recorded


class ZutisHipError(RuntimeError):
    pass


def declared_symbols(header: str = HEADER):
    """Every function name declared in include/zutis_hip.h."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zh_[a-z0-9_]+)\s*\(", txt)))


_vp, _l, _i, _f, _sz, _d = C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_size_t, C.c_double
_SIGS = {
    "zh_version": (C.c_int, []),
    "zh_arch": (C.c_char_p, []),
    "zh_last_error": (C.c_char_p, []),
    "zh_dev_set_gemm_overrides": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "zh_dev_set_gemm_persist": (C.c_int, [C.c_int]),
    "zh_gemm_f16": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "zh_gemm_f16x3": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _f, _vp, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i,
                           _i, _vp]),
    "zh_attention_f16": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _i, _f, _l, _l, _l, _l, _vp]),
    "zh_attention_splitk_workspace_size": (_sz, [_i, _i, _i, _i, _i]),
    "zh_attention_f16_splitk": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _i, _f, _l, _l, _l, _l, _i, _vp, _sz, _vp]),
    "zh_attention_causal_f16": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _f, _l, _l, _l, _l, _vp]),
    "zh_embed_tokens_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "zh_eot_rows_f32": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "zh_group_mean_l2norm": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "zh_layernorm_f32": (_i, [_vp, _l, _l, _l, _l, _l, _l, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _vp, _vp]),
    "zh_sum_layernorm_f32": (_i, [_vp, _i, _l, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _l, _l, _l, _l, _i, _vp, _vp, _f, _vp, _vp, _l, _l, _l, _l, _i, _i, _vp, _vp]),
    "zh_assemble_tokens_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "zh_l2norm_rows": (_i, [_vp, _vp, _vp, _f, _i, _i, _l, _f, _vp]),
    "zh_global_ln_l2_workspace_size": (_sz, [_i, _i, _i]),
    "zh_global_ln_l2": (_i, [_vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _sz, _l, _f, _vp, _vp]),
    "zh_im2col_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _l, _vp]),
    "zh_posembed_bicubic": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "zh_select_upsample_mask": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "zh_upsample2x_bilinear_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _vp]),
    "zh_sine_pe": (_i, [_vp, _i, _i, _i, _f, _vp]),
    "zh_add_rowperiodic_f16": (_i, [_vp, _vp, _vp, _l, _i, _i, _l, _l, _vp]),
    "zh_fill_f32": (_i, [_vp, _f, _l, _vp]),
    "zh_cast_f32_f16": (_i, [_vp, _vp, _i, _vp, _l, _i, _l, _f, _vp]),
    "zh_upsample_argmax": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "zh_upsample_bilinear_nchw": (_i, [_vp, _vp, _vp, _f, _l, _i, _i, _i, _i, _f, _f, _vp]),
    "zh_confusion_hist": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "zh_topk_rows": (_i, [_vp, _l, _i, _l, _i, _vp, C.c_longlong, _vp, _vp, _l, _vp]),
    "zh_mask_runs_workspace_size": (_sz, [_i, _i]),
    "zh_mask_runs": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "zh_mask_rle_fused_supported": (_i, [_i, _i, _i]),
    "zh_mask_rle_fused_kept": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp]),
    "zh_mask_rle_kept": (_i, [_vp, _l, _vp, _vp, _i, _i, _i, _l, _vp, _l, _vp, _vp]),
    "zh_mask_runs_kept": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _sz, _vp]),
    "zh_rle_counts_to_string_host": (C.c_long, [_vp, C.c_long, _vp, C.c_long]),
    "zh_rle_from_transitions_host": (C.c_long, [_vp, C.c_long, _i, _vp, C.c_long, C.c_long, _vp, C.c_long, _vp]),
    "zh_rle_encode_host": (C.c_long, [_vp, _i, _i, _vp, C.c_long]),
    "zh_resize_nearest_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "zh_instance_mask_stats": (_i, [_vp, _l, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "zh_masked_mean_workspace_size": (_sz, [_i, _i, _i, _i]),
    "zh_masked_mean_tokens": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "zh_instance_classify": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp]),
    "zh_denormalize_u8": (_i, [_vp, _vp, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "zh_bgrid_coords": (_i, [_vp, _i, _i, _d, _d, _d, _vp, _vp]),
    "zh_bilateral_workspace_size": (_sz, [_i, _i, _d, _d, _d]),
    "zh_bilateral_solve": (_i, [_vp, _vp, _vp, _i, _i, _d, _d, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "zh_bilateral_solve_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "zh_threshold_f64_u8": (_i, [_vp, _d, _vp, _l, _vp]),
    "zh_plan_op_name": (C.c_char_p, [_i]),
    "zh_plan_run": (_i, [_vp, _i, _vp]),
    "zh_plan_run_multi": (_i, [_vp, _vp, _vp, _i]),
    "zh_plan_run2": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "zh_mask_nms": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zh_mask_iou_workspace_size": (_sz, [_i, _l]),
    "zh_mask_iou_counts": (_i, [_vp, _i, _l, _vp, _vp, _vp, _sz, _vp]),
}


COUNTER = None    # a dict while launches are being counted (bench.py: kernels per image): entry-point name -> calls


class _CountingProxy:
    """Forwards every call and counts the plannable (= launching) entry points."""

    def __init__(self, lib, counts):
        self._lib, self._counts = lib, counts

    def __getattr__(self, name):
        from . import plan
        fn = getattr(self._lib, name)
        if name in plan.op_table():
            def _count(*args):
                self._counts[name] = self._counts.get(name, 0) + 1
                return fn(*args)
            return _count
        return fn


class _RecordingProxy:
    """Stands in for the CDLL while a plan is recorded: plannable entry points are logged, everything else passes through."""

    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        from . import plan
        fn = getattr(self._lib, name)
        if name in plan.op_table():
            def _log(*args):
                self._rec.calls.append((name, args))
                return 0
            return _log
        return fn


def header_abi_version() -> int:
    """ZH_ABI_VERSION of include/zutis_hip.h (the header travels with the package: bindings and library must agree on it)."""
    import re
    m = re.search(r"^#define\s+ZH_ABI_VERSION\s+(\d+)", open(HEADER).read(), re.M)
    if not m:
        raise ZutisHipError(f"{HEADER}: ZH_ABI_VERSION not found")
    return int(m.group(1))


def load(raw: bool = False):
    """Load the shared library (building nothing: run `python -m zutis_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        if RECORDER is not None and not raw:
            return _RecordingProxy(_lib, RECORDER)
        if COUNTER is not None and not raw:
            return _CountingProxy(_lib, COUNTER)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZutisHipError(
            f"{LIB_PATH} is missing: the HIP extension is REQUIRED (no CPU fallback). "
            "Build it with `python -m zutis_amd.build`.")
    import torch  # noqa: F401  (first: the process must use ONE HIP runtime — the one torch loads; libzutis_hip binds to it)
    lib = C.CDLL(LIB_PATH)
    lib.zh_version.restype = C.c_int
    built, want = lib.zh_version(), header_abi_version()
    if built != want:       # a stale build: ctypes would pass the new argument lists to the old entry points
        raise ZutisHipError(f"{LIB_PATH} was built for ABI {built} but include/zutis_hip.h declares {want}: "
                            "rebuild it with `python -m zutis_amd.build`.")
    for name, (res, args) in _SIGS.items():
        if not hasattr(lib, name):
            continue  # symbol check is test_capi's job; optional groups may be absent in partial builds
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return load(raw)


def register(name, restype, argtypes):
    _SIGS[name] = (restype, argtypes)
    if _lib is not None and hasattr(_lib, name):
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().zh_last_error().decode("utf-8", "replace")
        raise ZutisHipError(f"{what} failed (rc={rc}): {msg}")
