"""Tolerance model for comparing the GPU GEMM with the oracle.

Bit-exactness of the GEMM is impossible by construction: the reference
itself is CUTLASS + Blackwell tensor cores with an unspecified in-block summation order, and the
CDNA4 scaled MFMA sums the 32 products of a block in a fixed-point adder tree that is NOT a
correctly rounded fp32 dot product when fp6/fp8 operands are involved (measured: relative error
up to ~1e-4 of the largest term for fp8 x fp8, exact for fp4 x fp4 and fp4 x fp6; see
tests/test_hw_gpu.py).  Stated tolerance, per output element:

    |got - want| <= ULPS * 2^-7 * (sum over segments of |running D after that segment|)   (bf16 roundings)
                  + EPS_HW * S,   S = sum_k |a_k| * |b_k|  (dequantised magnitudes)        (MFMA adder tree)

with ULPS = 1 per rounding and EPS_HW = 2^-11.  The second term is what an honest mixed-precision
GEMM bound looks like; for typical data S ~ sqrt(K) * |want| so it is far below one bf16 ulp.
"""
import numpy as np

from oracle import mx_oracle as o

EPS_HW = 2.0 ** -11


def abs_dot_and_partials(qx, qw):
    """returns (S [M,N] float64, list of fp64 per-segment products)."""
    m, n, kn, ks, ko, wmode = o.matmul_shapes(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2])
    af = ("fp4", "fp6", "fp8")
    bf = ("fp4", "fp4", "fp4") if wmode == "w4" else af
    S = np.zeros((m, n))
    parts = []
    for i, kseg in enumerate((kn, ks, ko)):
        if not kseg:
            continue
        a = o.dequant_segment(qx[i], qx[3 + i], m, kseg, af[i], np.float64)
        b = o.dequant_segment(qw[i], qw[3 + i], n, kseg, bf[i], np.float64)
        S += np.abs(a) @ np.abs(b).T
        parts.append(a @ b.T)
    return S, parts


def check_gemm(got_bits, qx, qw, rounding="reference", eps=EPS_HW, label=""):
    """asserts the tolerance above; returns a dict of statistics."""
    want = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5], rounding=rounding)
    S, parts = abs_dot_and_partials(qx, qw)
    run = np.zeros_like(S)
    rounding_budget = np.zeros_like(S)
    if rounding == "reference":
        for p in parts:
            run = run + p
            rounding_budget += np.abs(run)
    else:
        rounding_budget = np.abs(sum(parts)) if parts else rounding_budget
    g = o.bf16_to_f32(got_bits).astype(np.float64)
    w = o.bf16_to_f32(want).astype(np.float64)
    finite = np.isfinite(w) & np.isfinite(S)
    tol = 2.0 ** -7 * rounding_budget + eps * S + 1e-37
    err = np.abs(g - w)
    bad = finite & ~(err <= tol)
    ulp = o.bf16_ulp_distance(got_bits, want)
    stats = dict(max_ulp=int(ulp[finite].max()) if finite.any() else 0, frac_exact=float((ulp[finite] == 0).mean()),
                 frac_gt1=float((ulp[finite] > 1).mean()), worst_ratio=float((err[finite] / tol[finite]).max()),
                 hw_eps=float((np.maximum(err - 2.0 ** -8 * np.abs(w), 0)[finite] / (S[finite] + 1e-300)).max()))
    assert not bad.any(), f"{label}: {int(bad.sum())} elements outside tolerance; stats {stats}"
    return stats
