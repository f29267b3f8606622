"""The weight-streaming kernel (mx_gemm_stream.hip) at 32 < M <= 64 -- three and four 16-token tiles, the per-segment reduction of the
eight-tile variant, token rows in row group 1 of the activation scale atoms -- on all rows against the oracle.  plan_tiles sends only
some shapes of that range to this kernel, so the test pins it with the kernel-developer switch MICROMIX_MID_M_STREAM=1, which the
library reads once per process: the cases run in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
from conftest import *            # noqa: F401,F403  (sys.path set-up)
from gemm_check import check_gemm
from micromix_amd import _lib, mixedgemm
from oracle import mx_oracle as o
from test_matmul_gpu import gpu_matmul, quantized
dev = torch.device("cuda:0")
CASES = [(33, 272, (1024, 128, 896)), (48, 8224, (512, 128, 384)), (40, 4100, (0, 256, 2048)), (64, 8200, (2048, 128, 1920)), (49, 528, (384, 128, 640)),
         (64, 1024, (12288, 1024, 1024)), (57, 8448, (0, 1280, 0)), (64, 200, (128, 0, 0))]
for m, n, split in CASES:
    for wmode in ("w4", "w"):
        desc = _lib.load().mm_matmul_describe(m, n, *split, 1 if wmode == "w4" else 0, 0, 0).decode()
        assert "mx_gemm_stream_kernel" in desc, desc
        rng = np.random.default_rng(m * 131 + n)
        qx, qw = quantized(rng, m, n, sum(split), split, wmode)
        for rounding in ("reference", "fused"):
            got = gpu_matmul(dev, qx, qw, rounding=rounding, split_k=False)
            check_gemm(got, qx, qw, rounding, label=f"stream {m}x{n} {split} {wmode} {rounding}")
print("ok", len(CASES))
'''


@pytest.mark.gpu
def test_streaming_kernel_with_three_and_four_token_tiles():
    env = dict(os.environ, MICROMIX_MID_M_STREAM="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok 8" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
