"""GPU: the HIP quantizers of section 8f against the frozen fixture tests/golden/golden_v2.npz (no oracle code involved for
rmsnorm / downproj: digests of the kernel's bytes must equal the stored digests)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_v2 as g2  # noqa: E402
from conftest import t_from_bits, u8  # noqa: E402
from make_golden import QN, digest  # noqa: E402
from micromix_amd import mixedgemm  # noqa: E402
from oracle import mx_oracle as o  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(HERE, "golden", "golden_v2.npz"))


def test_rmsnorm_against_fixture(dev):
    import torch
    for k, split in g2.G7:
        x, w, idx = g2.g7_inputs(k)
        for ir in (True, False):
            got = mixedgemm.rmsnorm_quantize_x(t_from_bits(x, dev), t_from_bits(w, dev), g2.EPS,
                                               torch.from_numpy(idx.astype(np.int16)).to(dev), *split, integer_round=ir)
            torch.cuda.synchronize()
            for i, (n, t) in enumerate(zip(QN, got)):
                a = u8(t)
                a = a if i < 3 else a[o.sf_valid_offsets(24, split[i - 3])]
                assert np.array_equal(digest(a), GOLD[f"g7_{k}_{int(ir)}_{n}_sha"]), (k, ir, n)


def test_downproj_against_fixture(dev):
    import torch
    w = g2.g8_inputs()
    for w4, fn in ((False, mixedgemm.downproj_quantize_w), (True, mixedgemm.downproj_quantize_w4)):
        got = fn(t_from_bits(w, dev), 2048, 1024, 1024)
        torch.cuda.synchronize()
        for i, (n, t) in enumerate(zip(QN, got)):
            a = u8(t)
            a = a if i < 3 else a[o.sf_valid_offsets(64, (2048, 1024, 1024)[i - 3])]
            assert np.array_equal(digest(a), GOLD[f"g8_{int(w4)}_{n}_sha"]), (w4, n)


def test_activate_against_fixture(dev):
    """silu uses the device's expf: >= 99.9 % of the bytes equal the fixture (same budget as test_direct_quantize_gpu.py)."""
    import torch
    a, b = g2.g9_inputs()
    got = mixedgemm.activate_quantize_x(t_from_bits(a, dev), t_from_bits(b, dev), 512, 256, 256)
    torch.cuda.synchronize()
    for i, (n, t) in enumerate(zip(QN, got)):
        g, want = u8(t), GOLD[f"g9_{n}"]
        if i >= 3:
            off = o.sf_valid_offsets(16, (512, 256, 256)[i - 3])
            g, want = g[off], want[off]
        assert g.shape == want.shape
        assert (g == want).mean() >= 0.999, (n, (g != want).sum())
