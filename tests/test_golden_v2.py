"""CPU: the oracle still reproduces the frozen section-8f fixture tests/golden/golden_v2.npz byte for byte."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_v2 as g2  # noqa: E402
from make_golden import QN, digest  # noqa: E402
from oracle import mx_oracle as o  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "golden_v2.npz"))


def sf_valid(a, rows, kseg):
    return a[o.sf_valid_offsets(rows, kseg)]


def test_rmsnorm_fixture():
    for k, split in g2.G7:
        x, w, idx = g2.g7_inputs(k)
        assert np.array_equal(o.rmsnorm_rvar(x, g2.EPS), GOLD[f"g7_{k}_rvar"])
        for ir in (True, False):
            res = o.rmsnorm_quantize(x, w, g2.EPS, idx, *split, integer_round=ir)
            for i, (n, a) in enumerate(zip(QN, res)):
                got = a if i < 3 else sf_valid(a, 24, split[i - 3])
                assert np.array_equal(digest(got), GOLD[f"g7_{k}_{int(ir)}_{n}_sha"]), (k, ir, n)
                if k == 4096 and ir and i < 3:
                    assert np.array_equal(a, GOLD[f"g7_{k}_1_{n}"])


def test_downproj_fixture():
    w = g2.g8_inputs()
    for w4 in (False, True):
        res = o.downproj_quantize(w, 2048, 1024, 1024, w4=w4)
        for i, (n, a) in enumerate(zip(QN, res)):
            got = a if i < 3 else sf_valid(a, 64, (2048, 1024, 1024)[i - 3])
            assert np.array_equal(digest(got), GOLD[f"g8_{int(w4)}_{n}_sha"]), (w4, n)
        if not w4:
            assert res[3][o.sf_offset(0, 0, 2048)] == 127      # empty block -> scale 1.0 in activate.cu (reorder.cu has 126)


def test_activate_fixture():
    a, b = g2.g9_inputs()
    for i, (n, r) in enumerate(zip(QN, o.activate_quantize(a, b, 512, 256, 256))):
        want = GOLD[f"g9_{n}"]
        if i < 3:
            assert np.array_equal(r, want), n
        else:
            off = o.sf_valid_offsets(16, (512, 256, 256)[i - 3])
            assert np.array_equal(r[off], want[off]), n
