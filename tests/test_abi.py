"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/micromix_hip.h declares, its host-side helpers agree with the oracle, argument
validation happens before any device work, and the Python `mixedgemm` surface mirrors the
reference's pybind signatures (mgemm/src/bindings.cpp:682-742).  No kernel is launched here."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from micromix_amd import _lib, mixedgemm
from oracle import mx_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="micromix_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef MM_INSTRUMENT.*?#endif", "", text, flags=re.S)     # developer-variant-only declarations
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert declared and set(declared) == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert not hasattr(lib, "mm_diag_set_clock_buffer")    # in-kernel clock stamps exist only in the -DMM_INSTRUMENT variant
    assert lib.mm_version() >= 200
    assert mixedgemm.test_function() == "Hello from test_function!"        # bindings.cpp:700


def test_diag_library_is_separate_from_the_product_library():
    """the hardware probes / microbenchmarks live in libmicromix_diag.so; the product library exports none of them"""
    diag = _lib.load_diag()
    declared = header_functions("micromix_diag.h")
    assert declared and set(declared) == set(_lib.DIAG_EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(diag, name) and not hasattr(lib, name), name


def test_host_helpers_match_oracle():
    lib = _lib.load()
    for m, k in ((1, 128), (127, 4096), (128, 4096), (130, 1024), (4096, 4096)):
        assert lib.mm_sf_bytes_x(m, k) == o.sf_size_x(m, k)
        assert lib.mm_sf_bytes_w(m, k) == o.sf_size_w(m, k)
    rng = np.random.default_rng(1)
    for _ in range(500):
        r, j = int(rng.integers(0, 20000)), int(rng.integers(0, 448))
        assert lib.mm_sf_offset(r, j, 14336) == int(o.sf_offset(r, j, 14336))


def test_status_codes_without_device_work():
    lib = _lib.load()
    z = None
    # MM_OUT_F32 belongs to mm_matmul: the grouped and the fused-decode entries refuse it, and mm_matmul refuses it without
    # MM_ROUND_ONCE or with a bias (checked before any pointer is touched)
    assert lib.mm_matmul_grouped(None, 0, 128, 128, 0, 0, 1, _lib.MM_OUT_F32 | _lib.MM_ROUND_ONCE, None) == _lib.MM_ERR_UNSUPPORTED
    assert lib.mm_qlinear_decode(*([None] * 8), 1, 128, 128, 0, 0, 1, _lib.MM_OUT_F32 | _lib.MM_ROUND_ONCE, None, None, None) == _lib.MM_ERR_UNSUPPORTED
    assert lib.mm_matmul(*([1] * 12), 8, 128, 128, 0, 0, 1, _lib.MM_OUT_F32, None, 1, None) == _lib.MM_ERR_BAD_ARG
    assert lib.mm_matmul(*([1] * 12), 8, 128, 128, 0, 0, 1, _lib.MM_OUT_F32 | _lib.MM_ROUND_ONCE, 1, 1, None) == _lib.MM_ERR_BAD_ARG
    # bad splits are rejected before any pointer is touched (reference: "Value error in run_reorder_quantize_x")
    for K, kn, ks, ko in ((4096, 100, 0, 3996), (4096, 2048, 1024, 512), (0, 0, 0, 0), (256, -128, 128, 256)):
        assert lib.mm_reorder_quantize(z, 4, K, z, kn, ks, ko, 0, z, z, z, z, z, z, z) == _lib.MM_ERR_BAD_SPLIT
    assert lib.mm_reorder_quantize(z, 0, 256, z, 256, 0, 0, 0, z, z, z, z, z, z, z) == _lib.MM_OK        # empty input
    assert lib.mm_reorder_quantize(z, 4, 256, z, 256, 0, 0, 0, z, z, z, z, z, z, z) == _lib.MM_ERR_BAD_ARG  # null src
    assert lib.mm_reorder_quantize(z, 4, 256, z, 256, 0, 0, 7, z, z, z, z, z, z, z) == _lib.MM_ERR_BAD_ARG  # bad mode
    assert lib.mm_matmul(*[z] * 12, 0, 128, 128, 0, 0, 1, 0, z, z, z) == _lib.MM_OK                     # M == 0
    assert lib.mm_matmul(*[z] * 12, 4, 128, 100, 0, 0, 1, 0, z, z, z) == _lib.MM_ERR_BAD_SPLIT
    assert lib.mm_matmul(*[z] * 12, 4, 128, 128, 0, 0, 1, 0, z, z, z) == _lib.MM_ERR_BAD_ARG            # null D
    assert lib.mm_strerror(_lib.MM_ERR_BAD_SPLIT).decode().startswith("KN, KS, KO")
    with pytest.raises(RuntimeError, match="Value error in run_reorder_quantize_x"):
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "reorder_quantize_x")


def test_python_surface_matches_reference_signatures():
    # py::arg names of bindings.cpp:686-735
    assert list(inspect.signature(mixedgemm.matmul).parameters)[:12] == [
        "AN", "BN", "AS", "BS", "AO", "BO", "SFAN", "SFBN", "SFAS", "SFBS", "SFAO", "SFBO"]
    for fn, first in ((mixedgemm.reorder_quantize_x, "X"), (mixedgemm.reorder_quantize_w, "W"),
                      (mixedgemm.reorder_quantize_w4, "W")):
        assert list(inspect.signature(fn).parameters) == [first, "reorder_index", "KN", "KS", "KO"]
    assert list(inspect.signature(mixedgemm.activate_quantize_x).parameters) == ["A", "B", "KN", "KS", "KO"]
    assert list(inspect.signature(mixedgemm.downproj_quantize_w).parameters) == ["W", "KN", "KS", "KO"]
    assert list(inspect.signature(mixedgemm.downproj_quantize_w4).parameters) == ["W", "KN", "KS", "KO"]
    for name in ("batch_decode_i4", "batch_decode_f16", "init_kv_f16", "append_kv_i4"):
        with pytest.raises(NotImplementedError):
            getattr(mixedgemm, name)()


def test_no_cpu_fallback():
    x = torch.zeros((4, 256), dtype=torch.bfloat16)
    idx = torch.arange(256, dtype=torch.int16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        mixedgemm.reorder_quantize_x(x, idx, 256, 0, 0)
    with pytest.raises(TypeError):
        mixedgemm.reorder_quantize_x(x.float(), idx, 256, 0, 0)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmicromix_hip.so")
    with pytest.raises(_lib.MicroMixLibraryError, match="no CPU fallback"):
        _lib.load()


def test_dropin_module_importable_by_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mixedgemm_dropin", os.path.join(ROOT, "micromix_amd", "dropin", "mixedgemm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.matmul is mixedgemm.matmul and mod.reorder_quantize_w4 is mixedgemm.reorder_quantize_w4


def test_new_entry_points_validate_on_the_host():
    """split-K planning, the decode-kernel predicate and the rmsnorm entry are pure host logic until they launch."""
    lib = _lib.load()
    z = None
    # no split for decode sizes, for the headline shape, or for bad splits; a split (and a positive size) for a medium-M shape
    assert lib.mm_matmul_workspace_bytes(16, 4096, 0, 0, 4096, 1, 0) == 0
    assert lib.mm_matmul_workspace_bytes(4096, 4096, 0, 0, 4096, 1, 0) == 0
    assert lib.mm_matmul_workspace_bytes(128, 4096, 0, 0, 100, 1, 0) == 0
    ws = lib.mm_matmul_workspace_bytes(128, 4096, 7168, 512, 6656, 1, 0)   # down_proj: 16 tiles, 112 slabs -> split
    assert ws > 0 and ws % (128 * 1024) == 0                       # whole 128 KiB partial-sum blocks
    assert lib.mm_matmul_workspace_bytes(128, 4096, 7168, 512, 6656, 1, _lib.MM_SPLIT_K_ALWAYS) >= ws
    assert lib.mm_matmul_workspace_bytes(128, 4096, 0, 0, 4096, 1, _lib.MM_SPLIT_K_ALWAYS) > 0
    assert lib.mm_qlinear_decode_supported(1, 4096, 2048, 128, 1920) == 2
    assert lib.mm_qlinear_decode_supported(9, 4096, 2048, 128, 1920) == 0
    assert lib.mm_qlinear_decode_supported(4, 4096, 100, 0, 0) == 0
    assert lib.mm_qlinear_decode(z, z, z, z, z, z, z, z, 9, 128, 128, 0, 0, 1, 0, z, z, z) == _lib.MM_ERR_BAD_ARG or \
        lib.mm_qlinear_decode(z, z, z, z, z, z, z, z, 9, 128, 128, 0, 0, 1, 0, z, z, z) == _lib.MM_ERR_UNSUPPORTED
    assert lib.mm_qlinear_decode(z, z, z, z, z, z, z, z, 1, 128, 100, 0, 0, 1, 0, z, z, z) == _lib.MM_ERR_BAD_SPLIT
    assert lib.mm_rmsnorm_quantize(z, z, 1e-5, 4, 256, z, 128, 64, 64, 0, z, z, z, z, z, z, z) == _lib.MM_ERR_BAD_SPLIT
    assert lib.mm_rmsnorm_quantize(z, z, 1e-5, 0, 256, z, 128, 128, 0, 0, z, z, z, z, z, z, z) == _lib.MM_OK      # no rows
    assert lib.mm_rmsnorm_quantize(z, z, 1e-5, 4, 256, z, 128, 128, 0, 0, z, z, z, z, z, z, z) == _lib.MM_ERR_BAD_ARG
