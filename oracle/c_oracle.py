"""ctypes loader for oracle/mx_oracle.c -- TEST INFRASTRUCTURE ONLY (see mx_oracle.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmx_oracle.so")
FMT = {"fp4": 0, "fp6": 1, "fp8": 2}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.mxo_decode.restype = ctypes.c_float
        _lib.mxo_decode.argtypes = [ctypes.c_int, ctypes.c_int]
        for name in ("mxo_encode_search", "mxo_encode_fast", "mxo_scale_exponent_literal"):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_float, ctypes.c_int]
        _lib.mxo_sf_offset.restype = ctypes.c_int64
        _lib.mxo_sf_offset.argtypes = [ctypes.c_int64] * 3
        _lib.mxo_reorder_quantize.restype = ctypes.c_int
        _lib.mxo_matmul.restype = ctypes.c_int
        _lib.mxo_direct_quantize.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def reorder_quantize(x_bits, idx, kn, ks, ko, mode, sf_fill=0):
    from . import mx_oracle as o
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    idx = np.ascontiguousarray(idx, dtype=np.int16)
    rows, k = x_bits.shape
    fm = ("fp4", "fp4", "fp4") if mode == "w4" else ("fp4", "fp6", "fp8")
    outs = [np.zeros((rows, o.packed_width(f, kk)), np.uint8) for f, kk in zip(fm, (kn, ks, ko))]
    size = o.sf_size_x if mode == "x" else o.sf_size_w
    sfs = [np.full((size(rows, kk),), sf_fill, np.uint8) for kk in (kn, ks, ko)]
    rc = lib().mxo_reorder_quantize(_p(x_bits), rows, k, _p(idx), kn, ks, ko, 1 if mode == "w4" else 0,
                                    *[_p(a) for a in outs], *[_p(a) for a in sfs])
    if rc:
        raise ValueError("bad split")
    return (*outs, *sfs)


def matmul(an, bn, a_s, bs, ao, bo, sfan, sfbn, sfas, sfbs, sfao, sfbo):
    from . import mx_oracle as o
    m, n, kn, ks, ko, wmode = o.matmul_shapes(an, bn, a_s, bs, ao, bo)
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in
            (an, bn, a_s, bs, ao, bo, sfan, sfbn, sfas, sfbs, sfao, sfbo)]
    d = np.zeros((m, n), np.uint16)
    rc = lib().mxo_matmul(*[_p(a) for a in arrs], m, n, kn, ks, ko, 1 if wmode == "w4" else 0, _p(d))
    if rc:
        raise RuntimeError("mxo_matmul failed")
    return d


def direct_quantize(a_bits, b_bits, kn, ks, ko, mode, sf_fill=0):
    """mode 0: silu(a) * b (activate_quantize_x); 1: a, mixed formats (downproj_quantize_w); 2: a, all fp4 (downproj_quantize_w4)"""
    from . import mx_oracle as o
    a_bits = np.ascontiguousarray(a_bits, dtype=np.uint16)
    b_bits = np.ascontiguousarray(b_bits if b_bits is not None else a_bits, dtype=np.uint16)
    rows, k = a_bits.shape
    fm = ("fp4", "fp4", "fp4") if mode == 2 else ("fp4", "fp6", "fp8")
    outs = [np.zeros((rows, o.packed_width(f, kk)), np.uint8) for f, kk in zip(fm, (kn, ks, ko))]
    sfs = [np.full((o.sf_size_x(rows, kk),), sf_fill, np.uint8) for kk in (kn, ks, ko)]
    rc = lib().mxo_direct_quantize(_p(a_bits), _p(b_bits), rows, kn, ks, ko, mode, *[_p(x) for x in outs], *[_p(x) for x in sfs])
    if rc:
        raise ValueError("bad split")
    return (*outs, *sfs)
