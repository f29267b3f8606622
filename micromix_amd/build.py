"""Build recipe for libmicromix_hip.so (gfx950 only): a plain hipcc invocation, in-tree output.

    python -m micromix_amd.build [--force] [--keep-temps]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmicromix_hip.so")
SOURCES = ["capi.hip", "reorder_quantize.hip", "direct_quantize.hip", "rmsnorm_quantize.hip", "mx_gemm.hip", "mx_gemm256.hip", "mx_gemm_skinny.hip", "qlinear_decode.hip", "diag.hip"]
HEADERS = ["mx_common.h", "mx_kernels.h", "mx_acc_regs.h", "mx_gemm_tile.inc", "mx_gemm_tile_vgpr.inc", "mx_group_convert.h", os.path.join("..", "..", "include", "micromix_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-fgpu-rdc=0" if False else "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmicromix_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if keep_temps:
        tmp = os.path.join(LIBDIR, "temps")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj"]
    if verbose:
        print("[micromix_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=LIBDIR)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print(LIB)
