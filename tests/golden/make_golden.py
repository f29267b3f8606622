"""Generates the committed golden fixtures under tests/golden/ with the CPU oracle.

    python tests/golden/make_golden.py

The reference holds no known-answer vectors for this path and cannot run in this image
(CUDA + CUTLASS), so these fixtures are produced by oracle/mx_oracle.py (numpy) and
cross-checked against oracle/mx_oracle.c at generation time; they freeze the oracle's
behaviour (any later change to the oracle or kernels that alters a byte is caught) and give
the GPU box inputs + expected outputs that do not depend on a numpy RNG implementation.
Inputs are stored as raw bf16 bit patterns; outputs as packed bytes / SF bytes / bf16 bits.

Cases:
  G1  quantize-x, M=130 (two SF row tiles, M % 128 != 0), K=4096, four splits, random
      permutation; inputs include an all-zero group, amax == FMAX*2^e exactly, RNE ties,
      bf16 subnormals, negative zero.
  G2  quantize-w / -w4, N=256, same K / splits.
  G3  GEMM of G1 x G2 in both weight modes, reference and fused rounding (+ fp64 sums).
  G4  K=14336 and K=5120 (M=32, N=128): non-power-of-two K.
  G5  QLinearLayer.forward analogue with bias (M=40, N=256, K=1024).
  G6  mgemm/test.py's input distribution (growing magnitude in the last columns, random
      signs, identity reorder, split (0,0,K)), reduced to M=128, N=256, K=1024.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import c_oracle as c  # noqa: E402
from oracle import mx_oracle as o  # noqa: E402


def special_rows(k, rng):
    """rows exercising scale/rounding edge cases, as float32 (all exactly bf16-representable)."""
    rows = []
    z = rng.standard_normal(k).astype(np.float32)
    z = o.bf16_to_f32(o.f32_to_bf16(z))
    r = z.copy(); r[0:32] = 0.0; rows.append(r)                               # all-zero group
    r = z.copy(); r[32:64] = 0.0; r[40] = -0.0; rows.append(r)                # zero group incl. -0
    for fmax in (6.0, 28.0, 448.0):                                           # amax == FMAX * 2^e
        for e in (-3, 0, 5):
            r = np.clip(z, -1, 1) * np.float32(fmax * 2.0 ** e) * 0.5
            r = o.bf16_to_f32(o.f32_to_bf16(r)); r[::32] = fmax * 2.0 ** e; rows.append(r)
    r = np.zeros(k, np.float32)                                               # RNE ties for every format
    ties = np.array([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 0.03125, 0.09375, 1.125, 1.375, 18, 22, 26,
                     2 ** -10, 3 * 2 ** -10, 1.0625, 1.1875, 416, 6, 28, 448], np.float32)
    for g in range(k // 32):
        r[g * 32:(g + 1) * 32] = np.resize(ties * np.float32((-1) ** g), 32)
        r[g * 32] = (6.0, 28.0, 448.0)[g % 3] * (1 if g % 2 else -1)
    rows.append(r)
    sub = o.bf16_to_f32(np.arange(1, k + 1, dtype=np.uint16) % 0x7F + 1)     # bf16 subnormals
    rows.append(sub.astype(np.float32))
    r = z.copy() * np.float32(2.0 ** 100); rows.append(r)                     # huge
    r = z.copy() * np.float32(2.0 ** -100); rows.append(r)                    # tiny normal
    return np.stack(rows)


def check_c(x, idx, split, mode, res):
    res_c = c.reorder_quantize(x, idx, *split, mode)
    for a, b in zip(res, res_c):
        assert np.array_equal(a, b), "numpy and C oracles disagree"


def main():
    rng = np.random.default_rng(20260213)
    out = {}
    K = 4096
    splits = [(2048, 1024, 1024), (0, 0, 4096), (4096, 0, 0), (3968, 128, 0)]
    sp = special_rows(K, rng)
    M = 130
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[:, rng.choice(K, 41, replace=False)] *= 20
    x[: len(sp)] = sp
    xb = o.f32_to_bf16(x)
    idx = rng.permutation(K).astype(np.int16)
    w = (rng.standard_normal((256, K)) * 0.02).astype(np.float32)
    wb = o.f32_to_bf16(w)
    out.update(g1_x=xb, g1_idx=idx, g2_w=wb, splits=np.array(splits, np.int32))
    for si, split in enumerate(splits):
        qx = o.reorder_quantize(xb, idx, *split, "x")
        check_c(xb, idx, split, "x", qx)
        for n, a in zip(("xn", "xs", "xo", "sfxn", "sfxs", "sfxo"), qx):
            out[f"g1_{si}_{n}"] = a
        for mode in ("w", "w4"):
            qw = o.reorder_quantize(wb, idx, *split, mode)
            check_c(wb, idx, split, mode, qw)
            for n, a in zip(("wn", "ws", "wo", "sfwn", "sfws", "sfwo"), qw):
                out[f"g2_{si}_{mode}_{n}"] = a
            args = (qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5])
            d_ref, d64 = o.matmul(*args, rounding="reference", return_f64=True)
            d_fused = o.matmul(*args, rounding="fused")
            d_c = c.matmul(*args)
            assert o.bf16_ulp_distance(d_ref, d_c).max() <= 1
            out[f"g3_{si}_{mode}_ref"] = d_ref
            out[f"g3_{si}_{mode}_fused"] = d_fused
            out[f"g3_{si}_{mode}_f64"] = d64.astype(np.float32)
    # G4
    for K4, split in ((14336, (7168, 512, 6656)), (5120, (4096, 512, 512))):
        x4 = o.f32_to_bf16(rng.standard_normal((32, K4)).astype(np.float32))
        w4 = o.f32_to_bf16((rng.standard_normal((128, K4)) * 0.02).astype(np.float32))
        i4 = rng.permutation(K4).astype(np.int16)
        qx = o.reorder_quantize(x4, i4, *split, "x")
        qw = o.reorder_quantize(w4, i4, *split, "w4")
        check_c(x4, i4, split, "x", qx)
        d = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5])
        out.update({f"g4_{K4}_x": x4, f"g4_{K4}_w": w4, f"g4_{K4}_idx": i4, f"g4_{K4}_split": np.array(split, np.int32),
                    f"g4_{K4}_d": d})
        for n, a in zip(("xn", "xs", "xo", "sfxn", "sfxs", "sfxo"), qx):
            out[f"g4_{K4}_{n}"] = a
    # G5
    K5, N5, M5, split5 = 1024, 256, 40, (512, 128, 384)
    x5 = o.f32_to_bf16(rng.standard_normal((M5, K5)).astype(np.float32))
    w5 = o.f32_to_bf16((rng.standard_normal((N5, K5)) * 0.05).astype(np.float32))
    b5 = o.f32_to_bf16(rng.standard_normal(N5).astype(np.float32))
    i5 = np.argsort(np.abs(o.bf16_to_f32(x5)).mean(0), kind="stable").astype(np.int16)  # reorder_indices.py:64-69
    pw = o.qlinear_pack_weight(w5, i5, *split5)
    y5 = o.qlinear_forward(x5, i5, *split5, pw, bias_bits=b5)
    out.update(g5_x=x5, g5_w=w5, g5_bias=b5, g5_idx=i5, g5_split=np.array(split5, np.int32), g5_y=y5)
    # G6 (mgemm/test.py:13-27 distribution, CPU generator, reduced size)
    M6, N6, K6 = 128, 256, 1024
    kn, ks, ko = K6 - 256, 256 - 128, 128
    signs = rng.integers(0, 2, (M6, K6)).astype(np.float32) * 2 - 1
    x6 = rng.random((M6, K6)).astype(np.float32) * 3
    x6[:, -kn:] = rng.random((M6, kn)).astype(np.float32) * 8 + 8
    x6[:, -ks:] = rng.random((M6, ks)).astype(np.float32) * 16 + 16
    x6[:, -ko:] = rng.random((M6, ko)).astype(np.float32) * 32 + 32
    x6 = o.f32_to_bf16(x6 * signs)
    w6 = o.f32_to_bf16(rng.random((N6, K6)).astype(np.float32))
    i6 = np.arange(K6, dtype=np.int16)
    qx = o.reorder_quantize(x6, i6, 0, 0, K6, "x")
    qw = o.reorder_quantize(w6, i6, 0, 0, K6, "w4")
    d6 = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5])
    out.update(g6_x=x6, g6_w=w6, g6_d=d6)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
