"""ctypes binding of libsavsr_hip.so (the C ABI declared in include/savsr_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing the kernels raises.  The product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAVSR_LIB_PATH: diagnostics only (the instrumented build libsavsr_hip_diag.so of `SAVSR_DIAG=1 build.sh`, or an experiment
# build of tools/ab_conv.sh under its own name: the product library is never overwritten); unset, the in-tree
# product library is the only one ever loaded.
LIB_PATH = os.environ.get("SAVSR_LIB_PATH") or os.path.join(_HERE, "csrc", "libsavsr_hip.so")

MAX_SRC = 5
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3
SATU_LRCAT = 160
SATU_TABLE = 8
ABI_VERSION = 28
CONV_DIRECT, CONV_DIRECT_THROUGHPUT, CONV_WINOGRAD_Y, CONV_WINOGRAD_Y_THROUGHPUT = 0, 2, 3, 4
CONV_WY_FORMS = (CONV_WINOGRAD_Y, CONV_WINOGRAD_Y_THROUGHPUT)
SATU_LRCAT_TAIL = 96
TAIL_PLANES = 27

fptr = C.c_void_p   # raw device pointers travel as integers


class ConvDesc(C.Structure):
    _fields_ = [
        ("src", fptr * MAX_SRC),
        ("src_pix", C.c_int32 * MAX_SRC),
        ("nsrc", C.c_int32), ("src_ch", C.c_int32),
        ("h", C.c_int32), ("w", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32),
        ("wpacked", fptr), ("bias", fptr),
        ("act", C.c_int32), ("slope", C.c_float),
        ("mul_px", fptr),
        ("res1", fptr), ("res1_pix", C.c_int32),
        ("res2", fptr), ("res2_pix", C.c_int32),
        ("res2_scale", C.c_float),
        ("out", fptr), ("out_pix", C.c_int32),
        ("pool", fptr), ("pool_stride", C.c_int32),
        ("algo", C.c_int32),
    ]


class OSConvAttnDesc(C.Structure):
    _fields_ = [
        ("cin", C.c_int32), ("cout", C.c_int32), ("hidden", C.c_int32), ("knum", C.c_int32),
        ("inv_sh", C.c_float), ("inv_sw", C.c_float),
        ("partial", fptr), ("nblk", C.c_int32), ("inv_n", C.c_float),
        ("l1_w", fptr), ("l1_b", fptr), ("l2_w", fptr), ("l2_b", fptr),
        ("fc_w", fptr), ("bn_scale", fptr), ("bn_shift", fptr),
        ("ch_w", fptr), ("ch_b", fptr), ("fl_w", fptr), ("fl_b", fptr),
        ("sp_w", fptr), ("sp_b", fptr), ("kn_w", fptr), ("kn_b", fptr),
        ("v1", fptr), ("v2", fptr),
        ("bank", fptr), ("nunits", C.c_int64), ("wimg_out", fptr),
        ("att", fptr), ("wy", C.c_int32), ("fused", C.c_int32),
    ]


class SatuWeights(C.Structure):
    _fields_ = [
        ("body0_w", fptr), ("body0_b", fptr), ("body2_w", fptr), ("body2_b", fptr),
        ("head_w", fptr), ("head_b", fptr), ("kconv_w", fptr), ("kconv_b", fptr),
        ("proj_w", fptr), ("wbe_w", fptr), ("fusion_b", fptr),
    ]


class SatuTiling(C.Structure):
    _fields_ = [("tile_rows", C.c_int32), ("tile_cols32", C.c_int32), ("lr_rows", C.c_int32), ("lr_cols", C.c_int32),
                ("off_min_x", C.c_float), ("off_min_y", C.c_float), ("table_entries", C.c_int32),
                ("step_x", C.c_float), ("step_y", C.c_float), ("variant", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/savsr_hip.h declares
SIGNATURES = {
    "savsr_version": (C.c_char_p, []),
    "savsr_conv2d_max_batch": (C.c_int, []),
    "savsr_conv_wy_tile_count": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "savsr_osconv_weights_max_batch": (C.c_int, []),
    "savsr_conv_wy_packed_elems": (C.c_int64, [C.c_int, C.c_int]),
    "savsr_conv_wy_pack_index": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "savsr_source_hash": (C.c_char_p, []),
    "savsr_source_hash_satu": (C.c_char_p, []),
    "savsr_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "savsr_last_error": (C.c_char_p, []),
    "savsr_abi_version": (C.c_int, []),
    "savsr_prepare_device": (C.c_int, []),
    "savsr_conv_packed_elems": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "savsr_conv_pack_index": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "savsr_conv_pool_blocks": (C.c_int, [C.c_int, C.c_int]),
    "savsr_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "savsr_conv2d_batch": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "savsr_channel_sums": (C.c_int, [C.POINTER(fptr), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int64, C.c_int, fptr, C.c_void_p]),
    "savsr_osconv_weights": (C.c_int, [C.POINTER(OSConvAttnDesc), C.c_void_p]),
    "savsr_osconv_weights_batch": (C.c_int, [C.POINTER(OSConvAttnDesc), C.c_int, C.c_void_p]),
    "savsr_se_gate": (C.c_int, [fptr, C.c_int, C.c_float, fptr, fptr, fptr, fptr, C.c_int, C.c_int, fptr, C.c_void_p]),
    "savsr_scale_residual": (C.c_int, [fptr, fptr, fptr, fptr, C.c_int, C.c_int64, C.c_void_p]),
    "savsr_se_scale_residual": (C.c_int, [fptr, C.c_int, C.c_float, fptr, fptr, fptr, fptr, C.c_int, C.c_int, fptr, fptr, fptr, C.c_int64, C.c_void_p]),
    "savsr_se_scale_residual_batch": (C.c_int, [fptr, C.c_int, C.c_float, fptr, fptr, fptr, fptr, C.c_int, C.c_int, fptr, fptr, fptr, C.c_int64, C.c_int,
                                                C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "savsr_avgpool2": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "savsr_upsample2x": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "savsr_pack_windows": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "savsr_satu_phase_table": (C.c_int, [C.POINTER(SatuWeights), fptr, C.c_int, fptr, C.c_int, C.c_float, C.c_float,
                                         fptr, C.c_void_p]),
    "savsr_satu_lr_stage": (C.c_int, [C.POINTER(SatuWeights), fptr, fptr, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                      fptr, C.c_void_p]),
    "savsr_satu_expand_table": (C.c_int, [fptr, C.c_int, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "savsr_satu_hr_upsample": (C.c_int, [C.POINTER(SatuWeights), fptr, C.c_int, C.c_int, fptr, C.c_int, C.c_int, fptr, fptr, fptr, fptr, fptr,
                                         C.c_int, C.c_int, C.POINTER(SatuTiling), fptr, fptr, C.c_int64, C.c_void_p]),
    "savsr_satu_lr_stage_tail": (C.c_int, [C.POINTER(SatuWeights), fptr, fptr, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                           fptr, C.c_void_p]),
    "savsr_satu_hr_tail": (C.c_int, [C.POINTER(SatuWeights), fptr, C.c_int, C.c_int, fptr, C.c_int, C.c_int, fptr, fptr, fptr, fptr, fptr,
                                     C.c_int, C.c_int, C.POINTER(SatuTiling), fptr, fptr, C.c_int64, C.c_void_p]),
    "savsr_satu_hr_occupancy_target": (C.c_int, [C.c_int]),
    "savsr_satu_hr_compute_waves": (C.c_int, [C.c_int]),
    "savsr_satu_hr_variants": (C.c_int, []),
    "savsr_satu_hr_rows_per_wave_tile": (C.c_int, [C.c_int]),
    "savsr_satu_hr_lds_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "savsr_tail_gather": (C.c_int, [fptr, C.c_int64, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "savsr_satu_hr_tail_q": (C.c_int, [C.POINTER(SatuWeights), fptr, C.c_int, C.c_int, fptr, C.c_int, C.c_int, fptr, fptr, fptr, fptr, fptr,
                                       C.c_int, C.c_int, C.POINTER(SatuTiling), fptr, fptr, C.c_int64, fptr, C.c_int64, C.c_void_p]),
    "savsr_tail_gather_q": (C.c_int, [fptr, C.c_int64, fptr, C.c_int64, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "savsr_resize_aa_axis": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr, fptr, fptr, C.c_int, fptr, C.c_void_p]),
    "savsr_metrics_blocks": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "savsr_metrics_psnr_ssim_y": (C.c_int, [fptr, C.c_int64, fptr, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "savsr_metrics_psnr_ssim": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "savsr_tail_residual": (C.c_int, [fptr, C.c_int64, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
}

# the instrumented library only (header section under SAVSR_DIAG; tools load it through SAVSR_LIB_PATH)
DIAG_SIGNATURES = {
    "savsr_debug_conv_stamps": (C.c_int, [C.c_int]),
    "savsr_debug_read_conv_stamps": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "savsr_debug_satu_stamps": (C.c_int, [C.c_int]),
    "savsr_debug_read_satu_stamps": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "savsr_debug_satu_occupancy": (C.c_int, [C.c_int, C.c_int]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libsavsr_hip.so and bind every symbol; raises HipLibraryError on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  savsr_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():      # present in libsavsr_hip_diag.so only
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.savsr_abi_version() != ABI_VERSION:
        raise HipLibraryError("libsavsr_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().savsr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
