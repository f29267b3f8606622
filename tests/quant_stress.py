"""Randomised check of the quantizer kernels against the oracle, byte for byte:  python tests/quant_stress.py [seconds]

reorder_quantize_x / _w / _w4, rmsnorm_quantize_x (with and without the reference's integer rounding), downproj_quantize_w / _w4
(all of them exactly specified arithmetic: identical bytes) and activate_quantize_x (hardware exp / rcp: at most 1e-3 of the code
bytes may differ) on random row counts, K, splits (empty segments, sums below K for the gathering kernels), index shapes (random
permutation, identity, reversed, permuted blocks of 32) and input distributions (normal + outlier columns, zero rows and blocks,
values around 2^-120 and 2^100).  tests/test_stress_gpu.py runs a seeded slice of it; it lives under tests/ because it imports the
oracle (test infrastructure)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import t_from_bits, u8  # noqa: E402
from micromix_amd import mixedgemm  # noqa: E402
from oracle import mx_oracle as o  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "0")))


def rand_split(k, allow_short):
    g = k // 128
    if allow_short and g > 1 and rng.integers(0, 4) == 0:
        g = int(rng.integers(1, g))                       # a K-shard: the index selects fewer columns than the row has
    a = int(rng.integers(0, g + 1))
    b = int(rng.integers(0, g - a + 1))
    return (a * 128, b * 128, (g - a - b) * 128)


def rand_index(k, n):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        idx = np.arange(k)
    elif kind == 1:
        idx = np.arange(k)[::-1]
    elif kind == 2:
        idx = (rng.permutation(k // 32)[:, None] * 32 + np.arange(32)[None, :]).reshape(-1)
    else:
        idx = rng.permutation(k)
    return np.ascontiguousarray(idx[:n]).astype(np.int16)


def rand_rows(rows, k):
    kind = int(rng.integers(0, 5))
    x = rng.standard_normal((rows, k)).astype(np.float32)
    if kind == 0:
        x[:, rng.choice(k, size=max(1, k // 100), replace=False)] *= 20.0
    elif kind == 1:
        x *= np.float32(2.0 ** -120)                       # blocks whose scale exponent bottoms out at -127
        x[:, : k // 2] *= np.float32(2.0 ** -8)
    elif kind == 2:
        x *= np.float32(2.0 ** 100)
    elif kind == 3:
        x[rng.integers(0, rows)] = 0.0
        x[:, : 64] = 0.0
    b = o.f32_to_bf16(x)
    if kind == 4:
        b[:, ::7] = 0x0001                                  # bf16 denormals among ordinary values
    return b


def same(got, want, rows, split, exact=True):
    bad = 0
    for i, (g, w) in enumerate(zip(got, want)):
        g = u8(g)
        if g.shape != w.shape:
            return False, f"shape of output {i}: {g.shape} vs {w.shape}"
        if i < 3:
            diff = int((g != w).sum())
            if exact and diff:
                return False, f"packed segment {i}: {diff} bytes differ"
            bad += diff
        else:
            offs = o.sf_valid_offsets(rows, split[i - 3])
            if not np.array_equal(g[offs], w[offs]):
                if exact:
                    return False, f"scale bytes of segment {i - 3} differ"
                bad += int((g[offs] != w[offs]).sum())
    total = sum(u8(g).size for g in got[:3])
    return (True, "") if exact or bad <= max(2, total * 1e-3) else (False, f"{bad} of {total} bytes differ")


t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    rows = int(rng.choice([1, 2, 3, 5, 17, 64, 65, 100, 129, 300]))
    k = int(rng.choice([128, 256, 384, 512, 1024, 2048, 3584, 4096, 5120, 8192, 8320]))
    if rows * k > (1 << 20):
        rows = max(1, (1 << 20) // k)
    which = int(rng.integers(0, 7))
    label, ok, why = "", True, ""
    if which <= 2:                                          # reorder_quantize_x / _w / _w4
        split = rand_split(k, allow_short=False)
        idx = rand_index(k, k)
        xb = rand_rows(rows, k)
        mode = ("x", "w", "w4")[which]
        fn = (mixedgemm.reorder_quantize_x, mixedgemm.reorder_quantize_w, mixedgemm.reorder_quantize_w4)[which]
        got = fn(t_from_bits(xb, dev), torch.from_numpy(idx).to(dev), *split)
        want = o.reorder_quantize(xb, idx, *split, mode)
        if mode != "x":                                     # weight scale tensors: rows padded to 128
            ok, why = same(got, want, rows, split)
        else:
            ok, why = same(got, want, rows, split)
        label = f"reorder_quantize_{mode} rows={rows} K={k} split={split}"
    elif which <= 4:                                        # rmsnorm_quantize_x
        split = rand_split(k, allow_short=False)
        idx = rand_index(k, k)
        xb = rand_rows(rows, k)
        wb = o.f32_to_bf16((1.0 + 0.2 * rng.standard_normal(k)).astype(np.float32))
        ir = which == 3
        eps = float(rng.choice([1e-5, 1e-6]))
        got = mixedgemm.rmsnorm_quantize_x(t_from_bits(xb, dev), t_from_bits(wb, dev), eps, torch.from_numpy(idx).to(dev), *split,
                                           integer_round=ir)
        want = o.rmsnorm_quantize(xb, wb, eps, idx, *split, integer_round=ir)
        ok, why = same(got, want, rows, split)
        label = f"rmsnorm_quantize_x rows={rows} K={k} split={split} integer_round={ir}"
    elif which == 5:                                        # downproj_quantize_w / _w4
        split = rand_split(k, allow_short=False)
        wb = rand_rows(rows, k)
        w4 = bool(rng.integers(0, 2))
        got = (mixedgemm.downproj_quantize_w4 if w4 else mixedgemm.downproj_quantize_w)(t_from_bits(wb, dev), *split)
        want = o.downproj_quantize(wb, *split, w4)
        ok, why = same(got, want, rows, split)
        label = f"downproj_quantize_w{'4' if w4 else ''} rows={rows} K={k} split={split}"
    else:                                                   # activate_quantize_x
        split = rand_split(k, allow_short=False)
        ab = o.f32_to_bf16(rng.standard_normal((rows, k)).astype(np.float32) * 2)
        bb = o.f32_to_bf16(rng.standard_normal((rows, k)).astype(np.float32))
        got = mixedgemm.activate_quantize_x(t_from_bits(ab, dev), t_from_bits(bb, dev), *split)
        want = o.activate_quantize(ab, bb, *split)
        ok, why = same(got, want, rows, split, exact=False)
        label = f"activate_quantize_x rows={rows} K={k} split={split}"
    torch.cuda.synchronize()
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH {label}: {why}", flush=True)
print(f"{cases} cases, {fails} mismatches", flush=True)
sys.exit(1 if fails else 0)
