"""GPU test: a stand-alone C++ program (examples/cabi_demo.cpp: hipMalloc + the C ABI, no torch, no Python) builds against
include/micromix_hip.h, links libmicromix_hip.so and gets deterministic, exactly scale-linear results on every GEMM path."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("m", (4, 40, 200, 512, 2048))     # skinny 16-feature, skinny, split-K / 128-row tiles, tiles
def test_cpp_client_of_the_c_abi(dev, m):
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box")
    ex = os.path.join(ROOT, "examples")
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    built = subprocess.run(["make", "-C", ex, "-s", "cabi_demo"], env=env, capture_output=True, text=True)
    if built.returncode != 0:
        if os.path.exists(os.path.join(ex, "cabi_demo")):
            pass                                    # a prebuilt binary travelled with the checkout: use it
        else:
            pytest.skip("could not build examples/cabi_demo here: " + built.stderr[-300:])
    res = subprocess.run([os.path.join(ex, "cabi_demo"), str(m)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "run-to-run differences 0, scale-linearity violations 0" in res.stdout
    assert "bytes differing from mm_matmul x 2 + mm_activate_quantize: 0" in res.stdout
