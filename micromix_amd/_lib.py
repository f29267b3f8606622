"""ctypes binding of libmicromix_hip.so (the C ABI declared in include/micromix_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every
op raises.  A CPU path would silently void the parity claims of the GPU tests.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# MICROMIX_HIP_LIB lets a kernel developer A/B an experimental build of the same C ABI (never a fallback).
LIB_PATH = os.environ.get("MICROMIX_HIP_LIB") or os.path.join(_PKG, "lib", "libmicromix_hip.so")

# every symbol include/micromix_hip.h declares
EXPORTS = (
    "mm_version", "mm_strerror", "mm_last_error",
    "mm_sf_bytes_x", "mm_sf_bytes_w", "mm_sf_offset",
    "mm_reorder_quantize", "mm_reorder_quantize_gather", "mm_activate_quantize", "mm_downproj_quantize", "mm_matmul",
    "mm_matmul_ws", "mm_matmul_workspace_bytes", "mm_matmul_ws_reset",
    "mm_gate_up_activate", "mm_gate_up_activate_decode", "mm_rmsnorm_gate_up_activate_decode", "mm_rmsnorm_gate_up_activate_decode_supported", "mm_gate_up_activate_decode_supported", "mm_down_activate_decode", "mm_down_activate_decode_supported", "mm_down_activate_decode_supported_w", "mm_gate_up_activate_workspace_bytes", "mm_gate_up_activate_describe", "mm_rmsnorm_quantize", "mm_qlinear_decode", "mm_qlinear_decode_supported", "mm_qlinear_decode_supported_w", "mm_rmsnorm_qlinear_decode", "mm_rmsnorm_qlinear_decode_supported", "mm_rmsnorm_qlinear_decode_supported_w", "mm_matmul_grouped", "mm_reorder_quantize_grouped",
    "mm_matmul_describe", "mm_test_function", "mm_diag_set_kernel_events",
)
# every symbol include/micromix_diag.h declares (libmicromix_diag.so: hardware probes for tests/tools, never used by the ops)
DIAG_LIB_PATH = os.environ.get("MICROMIX_DIAG_LIB") or os.path.join(_PKG, "lib", "libmicromix_diag.so")
DIAG_EXPORTS = ("mm_diag_mfma", "mm_diag_hw_convert", "mm_diag_mfma_rate", "mm_diag_l2_bw", "mm_diag_stream_once")

MM_OK, MM_ERR_BAD_SPLIT, MM_ERR_BAD_ARG, MM_ERR_LAUNCH, MM_ERR_UNSUPPORTED, MM_ERR_NO_DEVICE = range(6)
MM_QUANT_MIXED, MM_QUANT_W4 = 0, 1
MM_W_MATCH, MM_W_FP4 = 0, 1
MM_ROUND_PER_SEGMENT, MM_ROUND_ONCE, MM_SPLIT_K_ALWAYS, MM_WS_TICKETS_ZEROED, MM_OUT_F32 = 0, 1, 2, 4, 8
MM_WS_TICKET_BYTES = 4096
MM_RMS_REFERENCE, MM_RMS_NO_INTEGER_ROUND = 0, 1
MM_NORM_NO_INTEGER_ROUND = 0x100

class MMGroup(ctypes.Structure):
    """mm_group of include/micromix_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("AN", "AS", "AO", "SFAN", "SFAS", "SFAO", "BN", "BS", "BO", "SFBN", "SFBS", "SFBO",
                                              "bias_bf16", "D")] + [("M", ctypes.c_int)]


class MMQuantGroup(ctypes.Structure):
    """mm_quant_group of include/micromix_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("src_bf16", "reorder_index", "oN", "oS", "oO", "sfN", "sfS", "sfO")] + [("rows", ctypes.c_int)]


_lib = None


class MicroMixLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MicroMixLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m micromix_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.mm_version.restype = i
    lib.mm_version.argtypes = []
    lib.mm_strerror.restype = ctypes.c_char_p
    lib.mm_strerror.argtypes = [i]
    lib.mm_last_error.restype = ctypes.c_char_p
    lib.mm_last_error.argtypes = []
    lib.mm_sf_bytes_x.restype = sz
    lib.mm_sf_bytes_x.argtypes = [i, i]
    lib.mm_sf_bytes_w.restype = sz
    lib.mm_sf_bytes_w.argtypes = [i, i]
    lib.mm_sf_offset.restype = sz
    lib.mm_sf_offset.argtypes = [i, i, i]
    lib.mm_reorder_quantize.restype = i
    lib.mm_reorder_quantize.argtypes = [vp, i, i, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.mm_reorder_quantize_gather.restype = i
    lib.mm_reorder_quantize_gather.argtypes = [vp, i, i, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.mm_activate_quantize.restype = i
    lib.mm_activate_quantize.argtypes = [vp, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.mm_downproj_quantize.restype = i
    lib.mm_downproj_quantize.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.mm_matmul.restype = i
    lib.mm_matmul.argtypes = [vp] * 12 + [i] * 7 + [vp, vp, vp]
    lib.mm_rmsnorm_quantize.restype = i
    lib.mm_rmsnorm_quantize.argtypes = [vp, vp, ctypes.c_float, i, i, vp, i, i, i, i] + [vp] * 7
    lib.mm_qlinear_decode.restype = i
    lib.mm_qlinear_decode.argtypes = [vp] * 8 + [i] * 7 + [vp, vp, vp]
    lib.mm_qlinear_decode_supported.restype = i
    lib.mm_qlinear_decode_supported.argtypes = [i] * 5
    for name in ("mm_qlinear_decode_supported_w", "mm_rmsnorm_qlinear_decode_supported_w", "mm_down_activate_decode_supported_w"):
        getattr(lib, name).restype = i
        getattr(lib, name).argtypes = [i] * 6
    lib.mm_rmsnorm_qlinear_decode.restype = i
    lib.mm_rmsnorm_qlinear_decode.argtypes = [vp, vp, ctypes.c_float] + [vp] * 7 + [i] * 7 + [vp, vp, vp]
    lib.mm_rmsnorm_qlinear_decode_supported.restype = i
    lib.mm_rmsnorm_qlinear_decode_supported.argtypes = [i] * 5
    lib.mm_reorder_quantize_grouped.restype = i
    lib.mm_reorder_quantize_grouped.argtypes = [ctypes.POINTER(MMQuantGroup), i, i, i, i, i, i, vp]
    lib.mm_matmul_grouped.restype = i
    lib.mm_matmul_grouped.argtypes = [ctypes.POINTER(MMGroup), i, i, i, i, i, i, i, vp]
    lib.mm_matmul_ws.restype = i
    lib.mm_matmul_ws.argtypes = [vp] * 12 + [i] * 7 + [vp, vp, vp, ctypes.c_size_t, vp]
    lib.mm_gate_up_activate.restype = i
    lib.mm_gate_up_activate.argtypes = [vp] * 12 + [i] * 9 + [vp] * 7 + [ctypes.c_size_t, vp]
    lib.mm_rmsnorm_gate_up_activate_decode_supported.restype = i
    lib.mm_rmsnorm_gate_up_activate_decode_supported.argtypes = [i] * 5
    lib.mm_gate_up_activate_decode_supported.restype = i
    lib.mm_gate_up_activate_decode_supported.argtypes = [i] * 5
    lib.mm_rmsnorm_gate_up_activate_decode.restype = i
    lib.mm_rmsnorm_gate_up_activate_decode.argtypes = [vp, vp, ctypes.c_float] + [vp] * 7 + [i] * 9 + [vp] * 7 + [ctypes.c_size_t, vp]
    lib.mm_gate_up_activate_decode.restype = i
    lib.mm_gate_up_activate_decode.argtypes = [vp] * 8 + [i] * 9 + [vp] * 7 + [ctypes.c_size_t, vp]
    lib.mm_down_activate_decode_supported.restype = i
    lib.mm_down_activate_decode_supported.argtypes = [i] * 5
    lib.mm_down_activate_decode.restype = i
    lib.mm_down_activate_decode.argtypes = [vp] * 7 + [i] * 7 + [vp, vp, vp]
    lib.mm_gate_up_activate_workspace_bytes.restype = ctypes.c_size_t
    lib.mm_gate_up_activate_workspace_bytes.argtypes = [i, i]
    lib.mm_gate_up_activate_describe.restype = ctypes.c_char_p
    lib.mm_gate_up_activate_describe.argtypes = [i, i]
    lib.mm_matmul_ws_reset.restype = i
    lib.mm_matmul_ws_reset.argtypes = [vp, ctypes.c_size_t, vp]
    lib.mm_matmul_workspace_bytes.restype = ctypes.c_size_t
    lib.mm_matmul_workspace_bytes.argtypes = [i] * 7
    lib.mm_matmul_describe.restype = ctypes.c_char_p
    lib.mm_matmul_describe.argtypes = [i] * 7 + [ctypes.c_size_t]
    lib.mm_test_function.restype = ctypes.c_char_p
    lib.mm_test_function.argtypes = []
    if hasattr(lib, "mm_diag_set_clock_buffer"):      # only the -DMM_INSTRUMENT developer variant exports it (csrc/mx_instrument.h)
        lib.mm_diag_set_clock_buffer.restype = i
        lib.mm_diag_set_clock_buffer.argtypes = [vp]
    lib.mm_diag_set_kernel_events.restype = i
    lib.mm_diag_set_kernel_events.argtypes = [vp, vp]
    _lib = lib
    return lib


_diag = None


def load_diag():
    """libmicromix_diag.so (include/micromix_diag.h): hardware probes and microbenchmarks for tests/ and tools/."""
    global _diag
    if _diag is not None:
        return _diag
    if not os.path.exists(DIAG_LIB_PATH):
        raise MicroMixLibraryError(f"{DIAG_LIB_PATH} is missing: build it with `python -m micromix_amd.build`")
    lib = ctypes.CDLL(DIAG_LIB_PATH)
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mm_diag_mfma.restype = i
    lib.mm_diag_mfma.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, vp]
    lib.mm_diag_hw_convert.restype = i
    lib.mm_diag_hw_convert.argtypes = [vp, i, ctypes.c_float, i, vp, vp]
    lib.mm_diag_mfma_rate.restype = i
    lib.mm_diag_mfma_rate.argtypes = [i, i, i, i, i, vp, vp, vp]
    lib.mm_diag_l2_bw.restype = i
    lib.mm_diag_l2_bw.argtypes = [vp, ctypes.c_uint, i, i, i, i, i, vp, vp]
    lib.mm_diag_stream_once.restype = i
    lib.mm_diag_stream_once.argtypes = [vp, i, i, i, vp, vp]
    _diag = lib
    return lib


def check(status: int, what: str):
    """Map an mm_status to the exception the reference's binding raises."""
    if status == MM_OK:
        return
    lib = load()
    msg = lib.mm_strerror(status).decode()
    if status == MM_ERR_BAD_SPLIT:
        # reference: throw std::runtime_error("Value error in run_reorder_quantize_x") (bindings.cpp:145-148)
        raise RuntimeError(f"Value error in run_{what}: {msg}")
    if status == MM_ERR_LAUNCH:
        raise RuntimeError(f"{what}: {msg}: {lib.mm_last_error().decode()}")
    raise RuntimeError(f"{what}: {msg}")
