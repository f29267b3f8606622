import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Test-infrastructure convenience only: the product never builds or falls back by itself (micromix_amd._lib.load() raises
    # when the library is absent).  If the built library did not travel with the checkout, build it once here.
    from micromix_amd import _lib, build
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.DIAG_LIB_PATH)):
        try:
            build.build(verbose=False)
        except Exception as e:  # the tests that need the library will fail with the loader's message
            print(f"[conftest] could not build libmicromix_hip.so: {e}", file=sys.stderr)


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- bf16 <-> numpy bit helpers shared by the tests ------------------------------------------
def t_from_bits(bits: np.ndarray, device):
    """uint16 bf16 bit patterns -> torch bf16 tensor on device."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy())
    return t.view(torch.bfloat16).to(device)


def bits_from_t(t) -> np.ndarray:
    """torch bf16 tensor -> uint16 bit patterns (numpy)."""
    import torch
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def u8(t) -> np.ndarray:
    return t.detach().cpu().numpy()


def make_inputs(rng, rows, k, kind="normal"):
    """bf16 bit patterns [rows, k] with the distribution the bench uses: N(0,1), 1% outlier columns x20."""
    from oracle import mx_oracle as o
    x = rng.standard_normal((rows, k)).astype(np.float32)
    if kind == "normal":
        cols = rng.choice(k, size=max(1, k // 100), replace=False)
        x[:, cols] *= 20.0
    elif kind == "weight":
        x *= 0.02
    return o.f32_to_bf16(x)
