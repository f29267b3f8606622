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


# Statistics asserted with `strict=True`.  SURVEY.md section 8c proposed "<= 1 bf16 ulp on >= 99.9 % of the elements, <= 2 ulp
# max" for a comparison with the reference CUDA kernel; against a CPU oracle that sums every block in fp64, and measured over the
# full Llama / Qwen / Mixtral shapes (tests/test_model_shapes_gpu.py), the honest numbers are:
#   * FRAC_GT1: at most 0.3 % of the outputs differ from the oracle by more than one bf16 ulp, in both weight modes (the one case that
#     measures 0.30 % -- the all-fp8 x fp8 4096^3 product -- passes its own, documented bound: `frac_gt1=` of check_gemm).  Measured: < 0.05 % on the tiled
#     kernels with mixed splits, 0.10-0.21 % on the weight-streaming kernels (M <= 64: K is split over the 8 waves, so the fp32
#     sums are associated differently) and on all-MXFP8 activations; "w" mode up to 0.21 %.  Almost all of them are outputs
#     much smaller than their own terms (cancellation), where one ulp of the sum is many ulps of the result;
#   * FRAC_EXACT[mode]: the share of bit-equal outputs.  With fp4 weights ("w4", the production mode) fp4 x fp4 and fp6 x fp4
#     blocks are summed exactly by the MFMA and only the fp8 x fp4 block sums carry the adder-tree error (<= 2.5e-4 * S
#     measured); an all-MXFP8 activation (split (0, 0, K), K = 4096) measures 98.0-98.8 % bit-equal, mixed splits > 99 %.  With
#     matching-precision weights ("w") the fp6 x fp6 and fp8 x fp8 block sums carry up to 5e-4 * S, which moves a result across a
#     rounding boundary more often: 98.2 % measured on the Llama mixed split, 97.3 % (and 0.30 % of the outputs more than one
#     ulp away) on the all-fp8 x fp8 4096^3 product of tests/test_matmul_gpu.py::test_full_size_properties -- hence 96.5 % for
#     that mode (and FRAC_GT1_W_ALL_FP8 for that one case);
#   * MAX_ULP over the outputs that are not cancellation results: |want| >= CANCEL * S (S = sum |a||b|) AND |want| >= half of
#     the largest running value of the rounding chain (after each segment, before / after the bias).  Every rounding stage can
#     differ from the oracle's by one ulp OF THAT STAGE's magnitude, i.e. up to two ulps of a final value half its size, and
#     two stages can flip at once: measured 1-2, bound 4.  An output far smaller than its intermediate values has few
#     significant bits left in ANY summation order (ulp distances of 10^2..10^4 occur there by construction); those outputs are
#     held to the absolute bound above instead.
FRAC_GT1 = {"w4": 3e-3, "w": 3e-3}
FRAC_GT1_W_ALL_FP8 = 4e-3      # all-fp8 x fp8, K = 4096 (every block sum carries the adder-tree error): 0.30 % measured
FRAC_EXACT = {"w4": 0.975, "w": 0.965}
MAX_ULP = 4
CANCEL = 2.0 ** -9


def check_gemm(got_bits, qx, qw, rounding="reference", eps=EPS_HW, label="", strict=False, wdeq=None, bias_bits=None, frac_gt1=None):
    """asserts the tolerance above (and the statistics, with strict=True); returns a dict of statistics.
    wdeq: cached o.dequant_operand(qw, "w", wmode); bias_bits: [N] bf16 bits added the reference's way (qLinearLayer.py:70-71:
    y = bf16(y + bias)) to the oracle result before the comparison."""
    wmode = o.matmul_shapes(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2])[5]
    want, segs = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5], rounding=rounding,
                          b_dequant=wdeq, return_parts=True)
    S = np.zeros(want.shape)
    run = np.zeros(want.shape)
    peak = np.zeros(want.shape)                      # largest running value of the rounding chain
    rounding_budget = np.zeros(want.shape)
    for p, a, b in segs:
        S += np.abs(a) @ np.abs(b).T
        run = run + p
        if rounding == "reference":
            rounding_budget += np.abs(run)
            peak = np.maximum(peak, np.abs(run))
    if rounding != "reference":
        rounding_budget = np.abs(run)
    peak = np.maximum(peak, np.abs(run))
    if bias_bits is not None:
        bias = o.bf16_to_f32(np.asarray(bias_bits)).astype(np.float64)[None, :]
        want = o.f32_to_bf16(o.bf16_to_f32(want) + bias.astype(np.float32))
        rounding_budget = rounding_budget + np.abs(run + bias)
        peak = np.maximum(peak, np.abs(run + bias))
    g = o.bf16_to_f32(got_bits).astype(np.float64)
    w = o.bf16_to_f32(want).astype(np.float64)
    finite = np.isfinite(w) & np.isfinite(S)
    tol = 2.0 ** -7 * rounding_budget + eps * S + 1e-37
    err = np.abs(g - w)
    bad = finite & ~(err <= tol)
    ulp = o.bf16_ulp_distance(got_bits, want)
    big = finite & (np.abs(w) >= CANCEL * S) & (np.abs(w) >= 0.5 * peak)
    stats = dict(max_ulp=int(ulp[finite].max()) if finite.any() else 0, frac_exact=float((ulp[finite] == 0).mean()),
                 frac_gt1=float((ulp[finite] > 1).mean()), worst_ratio=float((err[finite] / tol[finite]).max()),
                 max_ulp_noncancelling=int(ulp[big].max()) if big.any() else 0,
                 hw_eps=float((np.maximum(err - 2.0 ** -8 * np.abs(w), 0)[finite] / (S[finite] + 1e-300)).max()))
    assert not bad.any(), f"{label}: {int(bad.sum())} elements outside tolerance; stats {stats}"
    if strict:
        count = int(finite.sum())
        assert stats["frac_gt1"] <= max(frac_gt1 if frac_gt1 is not None else FRAC_GT1[wmode], 3.0 / max(count, 1)) and stats["max_ulp_noncancelling"] <= MAX_ULP and \
            (stats["frac_exact"] >= FRAC_EXACT[wmode] or count < 4096), \
            f"{label}: ulp statistics {stats}"
    return stats
