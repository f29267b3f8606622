"""Seeded random shapes / splits / weight modes / rounding modes / bias through mixedgemm.matmul against the oracle on a row sample:
the deterministic shape lists of test_matmul_gpu.py pick the dispatch boundaries, this picks everything else at random (the
randomised tools/stress.py compares paths with each other; only tests may ask the oracle)."""
import os

import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits, u8
from gemm_check import check_gemm
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu


def _cases(n=48, seed=20260):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        m = int(rng.choice([1, 3, 8, 16, 17, 31, 40, 64, 65, 96, 130, 200, 257, 300, 390, 520, 700, 1030, 1600]))
        nn = int(rng.choice([24, 72, 136, 256, 520, 1024, 2050, 4096, 4104, 8200]))
        g = int(rng.choice([1, 2, 3, 5, 8, 12]))
        a = int(rng.integers(0, g + 1)); b = int(rng.integers(0, g - a + 1))
        if m * nn > (1 << 22):
            g = min(g, 3); a = min(a, g); b = min(b, g - a)
        split = (a * 128, b * 128, (g - a - b) * 128)
        out.append((m, nn, split, "w4" if rng.integers(0, 2) else "w", "reference" if rng.integers(0, 3) else "fused", bool(rng.integers(0, 2))))
    return out


# RANDOM_ORACLE_CASES / RANDOM_ORACLE_SEED: longer one-off runs (1500 cases of another seed take about five minutes)
CASES = _cases(int(os.environ.get("RANDOM_ORACLE_CASES", "48")), int(os.environ.get("RANDOM_ORACLE_SEED", "20260")))


@pytest.mark.parametrize("m,n,split,wmode,rounding,with_bias", CASES,
                         ids=[f"{c[0]}x{c[1]}-{'_'.join(map(str, c[2]))}-{c[3]}-{c[4]}{'-bias' if c[5] else ''}" for c in CASES])
def test_random_case_matches_oracle(dev, m, n, split, wmode, rounding, with_bias):
    import torch
    K = sum(split)
    rng = np.random.default_rng(m * 977 + n * 13 + K)
    xb = make_inputs(rng, m, K)
    wb = make_inputs(rng, n, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    bias = t_from_bits(make_inputs(rng, 1, n)[0], dev) if with_bias else None
    d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], bias=bias, rounding=rounding)
    rows = np.unique(np.concatenate([rng.choice(m, min(m, 16), replace=False), [0, m - 1]]))
    qx = o.reorder_quantize(xb[rows], idx, *split, "x")
    check_gemm(bits_from_t(d)[rows], qx, [u8(t) for t in b], rounding, label=f"random {m}x{n} {split} {wmode} {rounding}",
               bias_bits=None if bias is None else bits_from_t(bias))
