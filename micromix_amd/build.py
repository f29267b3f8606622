"""Build recipe for the gfx950 libraries: plain hipcc invocations, in-tree output.

    python -m micromix_amd.build [--force] [--keep-temps]

  lib/libmicromix_hip.so   the product library (include/micromix_hip.h)
  lib/libmicromix_diag.so  hardware probes / microbenchmarks for tests and tools (include/micromix_diag.h); not loaded by the ops

Every source is compiled to its own object (in parallel, only when it or a header changed) and linked with hipcc.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmicromix_hip.so")
DIAG_LIB = os.path.join(LIBDIR, "libmicromix_diag.so")
SOURCES = ["capi.hip", "reorder_quantize.hip", "direct_quantize.hip", "rmsnorm_quantize.hip", "mx_gemm.hip", "mx_gemm256.hip",
           "mx_gemm_tiles_small.hip", "mx_gemm_skinny.hip", "mx_gemm_stream.hip", "qlinear_decode.hip"]
DIAG_SOURCES = ["diag.hip"]
HEADERS = ["mx_common.h", "mx_kernels.h", "mx_acc_regs.h", "mx_gemm_tile.inc", "mx_group_convert.h", "mx_instrument.h", "mx_direct_convert.h", "mx_decode_quant.h", "mx_rms_convert.h", "mx_gemm_prelude.h",
           os.path.join("..", "..", "include", "micromix_hip.h"), os.path.join("..", "..", "include", "micromix_diag.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP libraries cannot be built")
    return exe


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + [os.path.abspath(__file__)] if os.path.exists(os.path.join(CSRC, h)))


def _stale(target: str, deps_time: float, src: str | None = None) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return t < deps_time or (src is not None and t < os.path.getmtime(src))


def needs_build() -> bool:
    ht = _newest_header()
    for lib, srcs in ((LIB, SOURCES), (DIAG_LIB, DIAG_SOURCES)):
        if _stale(lib, ht) or any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(lib) for s in srcs):
            return True
    return False


# sources whose kernels keep accumulators (and, in the streaming kernels, pending load destinations) in registers that only inline asm
# names: _check_acc_regs.py examines the assembly of every build of them
GUARDED = {"mx_gemm256.hip": "verify", "mx_gemm_tiles_small.hip": "verify", "mx_gemm_stream.hip": "verify_stream",
           "rmsnorm_quantize.hip": "verify_pending", "qlinear_decode.hip": "verify_pending"}


def verify_no_scratch(objdir: str, src: str) -> int:
    """EVERY kernel of EVERY product source: no scratch (no spilled register), no static LDS in the tile kernels.  Returns the number
    of kernels examined (0 for a source without kernels, capi.hip); raises RuntimeError on a violation."""
    from . import _check_acc_regs as check_acc_regs
    stem = src.replace(".hip", "")
    asm = [f for f in os.listdir(objdir) if f.startswith(stem + "-") and f.endswith(".s") and "gfx950" in f]
    if not asm:
        raise RuntimeError(f"no device assembly of {src} found (was it compiled with -save-temps=obj?)")
    return check_acc_regs.verify_scratch(open(os.path.join(objdir, asm[0])).read())


def verify_acc_regs(objdir: str = OBJDIR, src: str = "mx_gemm256.hip") -> int:
    """The assembly hipcc generated for a guarded source (kept by -save-temps=obj) must not touch an asm-owned register outside
    the inline asm; raises RuntimeError otherwise.  Part of every build of those files, whatever its flags."""
    from . import _check_acc_regs as check_acc_regs
    stem = src.replace(".hip", "")
    asm = [f for f in os.listdir(objdir) if f.startswith(stem + "-") and f.endswith(".s") and "gfx950" in f]
    if not asm:
        raise RuntimeError(f"no device assembly of {src} found (was it compiled with -save-temps=obj?)")
    text = open(os.path.join(objdir, asm[0])).read()
    if GUARDED[src] == "verify":
        return check_acc_regs.verify(text, check_acc_regs.EXPECTED_BY_FILE[src])
    return getattr(check_acc_regs, GUARDED[src])(text)


def _flags_stamp() -> str:
    return os.path.join(OBJDIR, "flags.txt")


def _flags_changed(flags) -> bool:
    """objects in OBJDIR were compiled with other flags (an -D ablation build, --keep-temps): they must not be reused.  The stamp is
    written by build() only after a successful link, so an interrupted build with new flags is forced again next time."""
    stamp = _flags_stamp()
    return not (os.path.exists(stamp) and open(stamp).read() == " ".join(flags))


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True, extra_flags=()) -> str:
    flags = [*FLAGS, *extra_flags] + (["-save-temps=obj"] if keep_temps else [])
    if _flags_changed(flags):
        force = True
        if os.path.exists(_flags_stamp()):
            os.remove(_flags_stamp())      # whatever happens below, the old stamp no longer describes the objects
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc, ht = hipcc(), _newest_header()

    def compile_one(src):
        s, obj = os.path.join(CSRC, src), os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(obj, ht, s):
            product = src in SOURCES              # every product source is examined (scratch); the probes of diag.hip are not
            guarded = product and "-save-temps=obj" not in flags
            cmd = [cc, *flags, *(["-save-temps=obj"] if guarded else []), "-c", s, "-o", obj]
            if verbose:
                print("[micromix_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=OBJDIR)
            if product:
                try:
                    ns = verify_no_scratch(OBJDIR, src)   # a spilled register fails the build
                    n = verify_acc_regs(OBJDIR, src) if src in GUARDED else 0  # a violation fails the build: the library would compute wrong GEMMs
                except RuntimeError:
                    if os.path.exists(obj):
                        os.remove(obj)            # the next build compiles -- and examines -- this source again
                    raise
                if verbose:
                    print(f"[micromix_amd.build] {src}: {ns} kernels without scratch" + (f", asm-owned register guard: {n} kernels clean" if src in GUARDED else ""), flush=True)
                if guarded:                       # the temporaries were kept for the guards only
                    stem = src.replace(".hip", "")
                    for f in os.listdir(OBJDIR):
                        if f.startswith(stem + "-") or (f.startswith(stem + ".") and f != stem + ".o"):
                            os.remove(os.path.join(OBJDIR, f))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = dict(zip(SOURCES + DIAG_SOURCES, pool.map(compile_one, SOURCES + DIAG_SOURCES)))
    for lib, srcs in ((LIB, SOURCES), (DIAG_LIB, DIAG_SOURCES)):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", *[objs[s] for s in srcs], "-o", lib]
        if verbose:
            print("[micromix_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=LIBDIR)
    with open(_flags_stamp(), "w") as f:
        f.write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print(LIB)
    print(DIAG_LIB)
