"""GPU parity tests of mixedgemm.matmul (through the C ABI) against the oracle, within the
tolerance stated in tests/gemm_check.py; bit-exact where the hardware arithmetic is exact."""
import os
import sys

import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits, u8
from gemm_check import FRAC_GT1_W_ALL_FP8, check_gemm
from micromix_amd import mixedgemm
from model_case import tile_positions
from oracle import mx_oracle as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def to_dev(dev, arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def gpu_matmul(dev, qx, qw, **kw):
    import torch
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
    torch.cuda.synchronize()
    return bits_from_t(d)


def quantized(rng, m, n, k, split, wmode):
    xb = make_inputs(rng, m, k)
    wb = make_inputs(rng, n, k, "weight")
    idx = rng.permutation(k).astype(np.int16)
    return o.reorder_quantize(xb, idx, *split, "x"), o.reorder_quantize(wb, idx, *split, wmode)


SHAPES = [
    (128, 128, 128, (128, 0, 0)), (128, 128, 128, (0, 128, 0)), (128, 128, 128, (0, 0, 128)),
    (1, 128, 512, (256, 128, 128)), (7, 256, 384, (128, 128, 128)), (33, 384, 1024, (512, 128, 384)),
    (130, 256, 4096, (2048, 1024, 1024)), (257, 512, 1024, (0, 0, 1024)), (64, 200, 512, (256, 0, 256)),
    (300, 1024, 2048, (1024, 0, 1024)), (96, 640, 5120, (4096, 512, 512)),
    (140, 131, 256, (128, 0, 128)), (300, 72, 128, (0, 128, 0)), (513, 257, 384, (128, 128, 128)),   # odd N: scalar store path
    (16, 200, 640, (256, 128, 256)), (12, 4096, 384, (128, 128, 128)),    # M <= 16: 16-feature skinny kernel (N <= 4096)
    (9, 4128, 256, (128, 0, 128)), (20, 384, 640, (256, 128, 256)),       # 32-feature skinny kernel, one token tile
]
# shapes the library runs as split-K when given a workspace (few output tiles, M > 64); each also runs unsplit
SPLIT_SHAPES = [
    (65, 256, 512, (0, 0, 512)), (130, 256, 4096, (2048, 1024, 1024)), (128, 1024, 4096, (0, 0, 4096)),
    (256, 512, 1792, (1024, 128, 640)), (200, 300, 2048, (128, 1792, 128)), (512, 2048, 1024, (512, 512, 0)),
    (96, 640, 5120, (4096, 512, 512)), (192, 256, 14336, (12288, 1024, 1024)), (129, 264, 768, (256, 256, 256)),
    (48, 256, 14336, (12288, 1024, 1024)),     # 32 < M <= 64 with a long K: split-K tiles when forced, the weight-streaming kernel otherwise
]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("rounding", ("reference", "fused"))
@pytest.mark.parametrize("m,n,k,split", SHAPES)
def test_matmul_matches_oracle(dev, wmode, rounding, m, n, k, split):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    qx, qw = quantized(rng, m, n, k, split, wmode)
    got = gpu_matmul(dev, qx, qw, rounding=rounding)
    check_gemm(got, qx, qw, rounding, label=f"{m}x{n}x{k} {split} {wmode} {rounding}")


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("rounding", ("reference", "fused"))
@pytest.mark.parametrize("m,n,k,split", SPLIT_SHAPES)
def test_split_k_matches_oracle(dev, wmode, rounding, m, n, k, split):
    """the K-split path (partial sums in a workspace + reduction kernel) and the unsplit kernels on the same inputs"""
    from micromix_amd import _lib
    rng = np.random.default_rng(m * 5 + n * 11 + k)
    qx, qw = quantized(rng, m, n, k, split, wmode)
    assert _lib.load().mm_matmul_workspace_bytes(m, n, *split, 1 if wmode == "w4" else 0, _lib.MM_SPLIT_K_ALWAYS) > 0
    for split_k in ("force", True, False):
        got = gpu_matmul(dev, qx, qw, rounding=rounding, split_k=split_k)
        check_gemm(got, qx, qw, rounding, label=f"{m}x{n}x{k} {split} {wmode} {rounding} split_k={split_k}")


# Every K the reference binding compiles (mgemm/src/bindings.cpp:134-148: 3072, 3584, 4096, 5120, 8192, 11008, 12288, 13824,
# 14336, 18944) reaches mm_matmul with a three-segment split, in both weight modes, on a weight-streaming launch (M = 8) and on a
# tiled launch (M = 200); all rows against the oracle.
REFERENCE_K = [(3072, (2048, 512, 512)), (3584, (2560, 512, 512)), (4096, (2048, 128, 1920)), (5120, (4096, 512, 512)),
               (8192, (4096, 1024, 3072)), (11008, (9984, 896, 128)), (12288, (8192, 2048, 2048)), (13824, (12288, 1024, 512)),
               (14336, (7168, 512, 6656)), (18944, (12544, 3200, 3200))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("k,split", REFERENCE_K, ids=[str(c[0]) for c in REFERENCE_K])
def test_reference_compiled_k_values(dev, k, split, wmode):
    rng = np.random.default_rng(k)
    for m, n in ((8, 264), (200, 264)):
        qx, qw = quantized(rng, m, n, k, split, wmode)
        got = gpu_matmul(dev, qx, qw)
        check_gemm(got, qx, qw, "reference", label=f"reference K={k} M={m} {split} {wmode}")


def test_reference_smoke_shape_through_qlinear(dev):
    """mgemm/test.py:6-27: M = 128, N = 3584, K = 11008, identity reorder index, its sign / magnitude distribution (including the
    overlapping tail assignments at :14-20) and the split overridden to (0, 0, 11008) at :27; w4 weights (:35).  Run through
    QLinearLayer and through the three ops, compared with the oracle on every row (the reference script asserts nothing: it
    prints an MSE against the bf16 product)."""
    import torch
    from micromix_amd.qlinear import QLinearLayer
    M, N, K = 128, 3584, 11008
    g = torch.Generator().manual_seed(721)
    kn, ks, ko = K - 1024, 1024 - 128, 128
    signs = torch.randint(0, 2, (M, K), generator=g).to(torch.bfloat16) * 2 - 1
    X = torch.rand((M, K), generator=g).to(torch.bfloat16) * 3
    X[:, -kn:] = torch.rand((M, kn), generator=g).to(torch.bfloat16) * 8 + 8
    X[:, -ks:] = torch.rand((M, ks), generator=g).to(torch.bfloat16) * 16 + 16
    X[:, -ko:] = torch.rand((M, ko), generator=g).to(torch.bfloat16) * 32 + 32
    X = (X * signs).to(dev)
    W = torch.rand((N, K), generator=g).to(torch.bfloat16).to(dev)
    idx = torch.arange(K, dtype=torch.int16, device=dev)
    split = (0, 0, K)
    a = mixedgemm.reorder_quantize_x(X, idx, *split)
    b = mixedgemm.reorder_quantize_w4(W, idx, *split)
    C = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    qx = o.reorder_quantize(bits_from_t(X), u8(idx), *split, "x")
    qw = o.reorder_quantize(bits_from_t(W), u8(idx), *split, "w4")
    # packed bytes and every real row's scale bytes (the SF tensors are over-allocated as bindings.cpp:120-123 does; the rows past M
    # are never written)
    from model_case import assert_rows_match_oracle
    assert_rows_match_oracle(a, np.arange(M), qx, split, "mgemm/test.py X")
    assert_rows_match_oracle(b, np.arange(N), qw, split, "mgemm/test.py W")
    check_gemm(bits_from_t(C), qx, qw, "reference", label="mgemm/test.py shape", strict=True)
    lin = torch.nn.Linear(K, N, bias=False, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        lin.weight.copy_(W)
    layer = QLinearLayer(lin, p8_num=K, p6_num=0, reorder_index=idx.long())
    assert torch.equal(layer(X.reshape(1, M, K)).reshape(M, N), C)      # forward takes [bsz, q_len, K] (qLinearLayer.py:58-66)
    # the script's own figure of merit: MSE against the unquantised product, relative to the output variance
    D = X.float() @ W.float().t()
    assert float(((C.float() - D) ** 2).mean() / D.var()) < 5e-2      # MXFP4 weights: ~1e-2 expected from the fp4 grid alone


# launches of at most 128 64x64 tiles (K <= 8192): split-K with the reduction inside the launch (split_tile_reduce: ticket per tile, the last
# workgroup sums the partial sums in split order).  The DEFAULT plan is asserted, so a change of the rule cannot drop the case.
IN_KERNEL_SPLIT = [(128, 1024, 4096, (2048, 128, 1920)), (100, 520, 2048, (1024, 256, 768)), (192, 1000, 5120, (0, 0, 5120)),
                   (65, 700, 2560, (2048, 0, 512)), (129, 1024, 4096, (4096, 0, 0))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,k,split", IN_KERNEL_SPLIT)
def test_in_kernel_split_k(dev, m, n, k, split, wmode):
    import torch
    from micromix_amd import _lib
    lib = _lib.load()
    flags = _lib.MM_WS_TICKETS_ZEROED
    need = lib.mm_matmul_workspace_bytes(m, n, *split, 1 if wmode == "w4" else 0, flags)
    desc = lib.mm_matmul_describe(m, n, *split, 1 if wmode == "w4" else 0, flags, need).decode()
    assert need > 0 and "in-kernel split-K" in desc, desc
    assert lib.mm_matmul_workspace_bytes(m, n, *split, 1 if wmode == "w4" else 0, 0) == 0 or "in-kernel" not in \
        lib.mm_matmul_describe(m, n, *split, 1 if wmode == "w4" else 0, 0, 1 << 30).decode()     # not without the caller's word on the tickets
    rng = np.random.default_rng(m + n + k)
    qx, qw = quantized(rng, m, n, k, split, wmode)
    bias = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(torch.bfloat16).to(dev)
    for rounding in ("reference", "fused"):
        runs = [gpu_matmul(dev, qx, qw, rounding=rounding) for _ in range(3)]
        assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])          # whichever workgroup reduces
        check_gemm(runs[0], qx, qw, rounding, label=f"in-kernel split {m}x{n}x{k} {split} {wmode} {rounding}")
        check_gemm(gpu_matmul(dev, qx, qw, rounding=rounding, split_k=False), qx, qw, rounding, label="unsplit twin")
    with_bias = gpu_matmul(dev, qx, qw, bias=bias)
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    plain = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    assert np.array_equal(with_bias, bits_from_t(plain + bias))
    # replayed from a hipGraph: the ticket counters are back at zero after every launch
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        mixedgemm.matmul(*args, out=out)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mixedgemm.matmul(*args, out=out)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, plain)


# MM_OUT_F32: the unrounded fp32 accumulator as output (partial products of a K-sharded tensor-parallel layer), on every kernel
# family: weight-streaming (16- and 32-feature), 64-row tiles (unsplit and in-kernel split), 128-row tiles, two-launch split-K,
# 256-row tiles; ragged M and N
F32_SHAPES = [(9, 200, 640, (256, 128, 256)), (40, 300, 512, (256, 0, 256)), (128, 1024, 4096, (2048, 128, 1920)),
              (250, 4000, 640, (256, 128, 256)), (700, 4090, 640, (256, 128, 256)), (192, 256, 14336, (12288, 1024, 1024)),
              (1000, 2050, 768, (256, 256, 256))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,k,split", F32_SHAPES)
def test_fp32_output(dev, m, n, k, split, wmode):
    import torch
    rng = np.random.default_rng(m + 3 * n + k)
    qx, qw = quantized(rng, m, n, k, split, wmode)
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    d32 = mixedgemm.matmul(*args, rounding="fused", out_dtype=torch.float32)
    assert d32.dtype == torch.float32 and tuple(d32.shape) == (m, n)
    # rounding the fp32 result once gives the bf16 result of the fused mode, bit for bit (same kernel, same accumulator)
    assert torch.equal(d32.to(torch.bfloat16), mixedgemm.matmul(*args, rounding="fused"))
    # and the fp32 values follow the oracle's fp64 sums within the hardware's block-sum error
    _, f64 = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5], rounding="fused", return_f64=True)
    _, segs = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5], rounding="fused", return_parts=True)
    S = sum(np.abs(pa) @ np.abs(pb).T for _, pa, pb in segs)
    got = d32.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - f64) <= 2.0 ** -11 * S + 2.0 ** -22 * np.abs(f64) + 1e-30)
    with pytest.raises(ValueError):
        mixedgemm.matmul(*args, out_dtype=torch.float32)                       # reference rounding: refused
    with pytest.raises(ValueError):
        mixedgemm.matmul(*args, rounding="fused", out_dtype=torch.float32, bias=torch.zeros(n, dtype=torch.bfloat16, device=dev))


def test_split_k_is_deterministic_and_keeps_bias(dev):
    import torch
    rng = np.random.default_rng(77)
    m, n, k, split = 160, 512, 2048, (1024, 256, 768)
    qx, qw = quantized(rng, m, n, k, split, "w4")
    bias = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(torch.bfloat16).to(dev)
    runs = [gpu_matmul(dev, qx, qw, bias=bias, split_k="force") for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    # the bias epilogue of the split path = the same split product plus a separate add (qLinearLayer.py:68-71)
    plain = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], split_k="force")
    assert np.array_equal(runs[0], bits_from_t(plain + bias))


@pytest.mark.parametrize("split", [(512, 0, 0), (0, 512, 0), (256, 256, 0)])
def test_exact_when_hardware_sums_exactly(dev, split):
    """fp4 x fp4 and fp6 x fp4 blocks are summed exactly by the MFMA: with a single segment the result equals the
    oracle bit for bit (production w4 weights), and also with two segments in fused rounding."""
    rng = np.random.default_rng(5)
    qx, qw = quantized(rng, 160, 256, 512, split, "w4")
    for rounding in ("reference", "fused"):
        got = gpu_matmul(dev, qx, qw, rounding=rounding)
        want = mg.mm(qx, qw, rounding=rounding)
        ulp = o.bf16_ulp_distance(got, want)
        # fp32 accumulation order differs from the oracle's fp64 sum only below one fp32 ulp
        assert ulp.max() <= 1 and (ulp > 0).mean() < 2e-3


def test_small_integer_data_is_bit_exact(dev):
    """operands whose products and sums are exactly representable: every format pair must be exact."""
    rng = np.random.default_rng(6)
    m, n, k = 96, 160, 384
    xv = rng.integers(-3, 4, (m, k)).astype(np.float32)
    wv = rng.integers(-2, 3, (n, k)).astype(np.float32) * 0.5
    xv[:, ::32] = 6.0   # pins every block scale: amax = 6 -> e = 0 (fp4), -3 (fp6: 28*2^-3 < 6 <= 28*2^-2) ...
    wv[:, ::32] = 3.0
    idx = rng.permutation(k).astype(np.int16)
    xb, wb = o.f32_to_bf16(xv), o.f32_to_bf16(wv)
    for wmode in ("w4", "w"):
        qx = o.reorder_quantize(xb, idx, 128, 128, 128, "x")
        qw = o.reorder_quantize(wb, idx, 128, 128, 128, wmode)
        got = gpu_matmul(dev, qx, qw, rounding="fused")
        assert np.array_equal(got, mg.mm(qx, qw, rounding="fused"))
        exact = o.f32_to_bf16((xv[:, idx.astype(int)] @ wv[:, idx.astype(int)].T).astype(np.float32))
        assert np.array_equal(got, exact)   # small integers survive quantisation: equals the plain product


def test_golden_gemm(dev):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    xb, idx, wb = mg.g1_inputs(g["g1_special_rows"])
    nsp = len(g["g1_special_rows"])
    for si, split in enumerate(g["splits"].tolist()):
        qx = o.reorder_quantize(xb, idx, *split, "x")
        for mode in ("w", "w4"):
            qw = o.reorder_quantize(wb, idx, *split, mode)
            for rounding, key in (("reference", "ref"), ("fused", "fused")):
                got = gpu_matmul(dev, qx, qw, rounding=rounding)
                want = g[f"g3_{si}_{mode}_{key}"]
                check_gemm(got[nsp:], _rows(qx, nsp), qw, rounding,
                           label=f"golden {split} {mode} {rounding}")
                ulp = o.bf16_ulp_distance(got[nsp:], want[nsp:])
                assert (ulp > 1).mean() < 5e-3
    for k in (14336, 5120):
        x4, w4, i4 = mg.g4_inputs(k)
        split = g[f"g4_{k}_split"].tolist()
        qx, qw = o.reorder_quantize(x4, i4, *split, "x"), o.reorder_quantize(w4, i4, *split, "w4")
        got = gpu_matmul(dev, qx, qw)
        check_gemm(got, qx, qw, "reference", label=f"golden K={k}")
        assert (o.bf16_ulp_distance(got, g[f"g4_{k}_d"]) > 1).mean() < 5e-3
    x6, w6, i6 = mg.g6_inputs()
    qx, qw = o.reorder_quantize(x6, i6, 0, 0, 1024, "x"), o.reorder_quantize(w6, i6, 0, 0, 1024, "w4")
    got = gpu_matmul(dev, qx, qw)
    check_gemm(got, qx, qw, "reference", label="golden test.py distribution")
    assert (o.bf16_ulp_distance(got, g["g6_d"]) > 1).mean() < 5e-3


def _rows(q, start):
    """drop the first `start` rows of a quantised activation (SF tensors re-laid out)."""
    m = q[0].shape[0]
    out = [q[0][start:], q[1][start:], q[2][start:]]
    widths = (q[0].shape[1] * 2, q[1].shape[1] * 4 // 3, q[2].shape[1])
    for sf, kseg in zip(q[3:], widths):
        new = np.zeros(o.sf_size_x(m - start, kseg), np.uint8)
        if kseg:
            r = np.arange(start, m)[:, None]
            j = np.arange(kseg // 32)[None, :]
            new[o.sf_offset(r - start, j, kseg)] = sf[o.sf_offset(r, j, kseg)]
        out.append(new)
    return out


def test_bias_epilogue_equals_separate_add(dev):
    import torch
    rng = np.random.default_rng(8)
    qx, qw = quantized(rng, 70, 384, 512, (256, 128, 128), "w4")
    bias = o.f32_to_bf16(rng.standard_normal(384).astype(np.float32))
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    tb = t_from_bits(bias, dev)
    fused = mixedgemm.matmul(*args, bias=tb)
    plain = mixedgemm.matmul(*args) + tb          # qLinearLayer.py:68-71
    assert torch.equal(fused, plain)


def test_full_size_properties(dev):
    """BASELINE configs[1] size (4096^3, all-MXFP8 activations, both weight modes):
    oracle on a row sample, exact power-of-two linearity, determinism, tile-independence."""
    import torch
    rng = np.random.default_rng(0)
    M = N = K = 4096
    xb = make_inputs(rng, M, K)
    wb = make_inputs(rng, N, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    for split in ((0, 0, 4096), (2048, 128, 1920)):
        a = mixedgemm.reorder_quantize_x(x, tidx, *split)
        for fn in (mixedgemm.reorder_quantize_w4, mixedgemm.reorder_quantize_w):
            b = fn(w, tidx, *split)
            args = lambda aa: (aa[0], b[0], aa[1], b[1], aa[2], b[2], aa[3], b[3], aa[4], b[4], aa[5], b[5])
            d1 = mixedgemm.matmul(*args(a))
            d2 = mixedgemm.matmul(*args(a))
            assert torch.equal(d1, d2)                                       # deterministic
            # linearity: +1 on every activation scale byte doubles the output exactly
            a2 = list(a[:3]) + [t + 1 for t in a[3:]]
            assert torch.equal(mixedgemm.matmul(*args(a2)).float(), d1.float() * 2)
            # a row block computed alone equals the same rows of the full product (tile independence)
            r0, r1 = 1000, 1100
            sub = mixedgemm.reorder_quantize_x(x[r0:r1].contiguous(), tidx, *split)
            assert torch.equal(mixedgemm.matmul(*args(sub)), d1[r0:r1])
            # oracle on a sample of rows that hits every (tile row, wave row, MFMA tile, lane half) position of the 256-row tiles
            # (model_case.tile_positions: 256 rows at M = 4096), with the asserted ulp statistics, in both rounding modes: this is
            # the exact shape, split and kernel bench.py times
            rows = tile_positions(rng, M)
            assert len(rows) == 256
            qx = o.reorder_quantize(xb[rows], idx, *split, "x")
            qw = [u8(t) for t in b]
            w4 = fn is mixedgemm.reorder_quantize_w4
            wdeq = o.dequant_operand(qw, "w", "w4" if w4 else "w")
            gt1 = FRAC_GT1_W_ALL_FP8 if (not w4 and split == (0, 0, 4096)) else None
            check_gemm(bits_from_t(d1)[rows], qx, qw, "reference", label=f"4096^3 {split} {fn.__name__}", strict=True, wdeq=wdeq, frac_gt1=gt1)
            df = mixedgemm.matmul(*args(a), rounding="fused")
            check_gemm(bits_from_t(df)[rows], qx, qw, "fused", label=f"4096^3 {split} {fn.__name__} fused", strict=True, wdeq=wdeq, frac_gt1=gt1)


def test_headline_launch_every_output(dev):
    """The bench's exact launch -- 4096^3, split (0, 0, 4096), fp4 weights, reference rounding, bench.synth_inputs -- against the
    oracle on EVERY one of its 16.7 M outputs, 512 rows at a time, each block under the strict statistics (one
    full-matrix comparison per round; the sampled tests see at most 6 % of the rows)."""
    import torch
    import bench
    M = N = K = 4096
    split = (0, 0, 4096)
    x, w, idx = bench.synth_inputs()
    xb, idxn = bits_from_t(x), idx.numpy().astype(np.int16)
    x, w, idx = x.to(dev), w.to(dev), idx.to(dev)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    got = bits_from_t(d)
    qw = [u8(t) for t in b]
    wdeq = o.dequant_operand(qw, "w", "w4")
    qa = [u8(t) for t in a]
    exact = gt1 = 0
    worst = 0
    for r0 in range(0, M, 512):
        rows = np.arange(r0, r0 + 512)
        qx = o.reorder_quantize(xb[rows], idxn, *split, "x")
        assert np.array_equal(qa[2][rows], qx[2])                      # the GPU quantizer's bytes ARE the oracle's, every row
        st = check_gemm(got[rows], qx, qw, "reference", label=f"headline rows {r0}..{r0 + 511}", strict=True, wdeq=wdeq)
        exact += st["frac_exact"] * 512
        gt1 += st["frac_gt1"] * 512
        worst = max(worst, st["max_ulp_noncancelling"])
    print(f"headline launch, all {M * N} outputs: {100 * exact / M:.3f} % bit-equal, {100 * gt1 / M:.4f} % more than one ulp off, "
          f"max {worst} ulp on non-cancelling outputs")


def test_poisoned_ticket_and_ws_reset(dev):
    """The in-kernel split-K counts arrivals per tile in the first MM_WS_TICKET_BYTES of the caller's workspace and relies on the
    caller's promise (MM_WS_TICKETS_ZEROED) that they are zero.  A launch that died mid-way can leave one non-zero; the documented
    behaviour is then: later launches on that workspace still terminate (nothing ever spins on a ticket) but may reduce a tile too
    early or never, i.e. return wrong or unwritten outputs -- until mm_matmul_ws_reset re-arms the workspace."""
    import torch
    from micromix_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    M, N, K = 128, 1024, 4096                     # k/v at M = 128: 32 tiles of 64 x 64, six splits (plan_small_split)
    split = (2048, 128, 1920)
    xb, wb = make_inputs(rng, M, K), make_inputs(rng, N, K, "weight")
    idx = torch.from_numpy(rng.permutation(K).astype(np.int16)).to(dev)
    a = mixedgemm.reorder_quantize_x(t_from_bits(xb, dev), idx, *split)
    b = mixedgemm.reorder_quantize_w4(t_from_bits(wb, dev), idx, *split)
    flags = _lib.MM_ROUND_PER_SEGMENT | _lib.MM_WS_TICKETS_ZEROED
    need = lib.mm_matmul_workspace_bytes(M, N, *split, _lib.MM_W_FP4, flags)
    assert need > _lib.MM_WS_TICKET_BYTES
    assert b"in-kernel split-K" in lib.mm_matmul_describe(M, N, *split, _lib.MM_W_FP4, flags, need)
    ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr() if t.numel() else None

    def run(out):
        st = lib.mm_matmul_ws(p(a[0]), p(b[0]), p(a[1]), p(b[1]), p(a[2]), p(b[2]), p(a[3]), p(b[3]), p(a[4]), p(b[4]), p(a[5]), p(b[5]),
                              M, N, *split, _lib.MM_W_FP4, flags, None, out.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        assert st == 0
        torch.cuda.synchronize()

    good = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    run(good)
    assert int(ws[:_lib.MM_WS_TICKET_BYTES].view(torch.int32).abs().sum()) == 0       # every launch leaves the tickets zero
    ref = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], split_k=False)
    assert torch.equal(good, ref) or float((good.float() - ref.float()).abs().max()) <= 2.0 ** -6 * float(ref.float().abs().max())
    # poison: as if a launch had died after some arrivals
    ws[:_lib.MM_WS_TICKET_BYTES].view(torch.int32)[:32] = torch.arange(32, device=dev, dtype=torch.int32) % 5 + 1
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    run(out)                                       # terminates (the assertion is that this call returns)
    assert lib.mm_matmul_ws_reset(None, 0, stream) == _lib.MM_ERR_BAD_ARG
    assert lib.mm_matmul_ws_reset(ws.data_ptr(), ws.numel(), stream) == 0
    out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    run(out2)
    assert torch.equal(out2, good)                 # re-armed: bit-identical to the clean run
    assert int(ws[:_lib.MM_WS_TICKET_BYTES].view(torch.int32).abs().sum()) == 0


# one problem per tile kernel of mx_gemm256.hip (the dispatch is asserted, so a change of plan_tiles cannot silently drop one):
# ragged M and N, all three segments, both weight modes; oracle on a row sample over every column
TILE_KERNELS = [
    # (and 32 < M <= 64 from N > 4096 on, instead of the weight-streaming kernel)
    ("g16", "32x64", 100, 4090), ("g16", "32x64", 128, 4000), ("g16", "32x64", 64, 6144), ("g16", "32x64", 40, 6144),     # 64 x 64 tiles would fill at most half of the CUs: 32 x 64 tiles
    ("g32n", "64x64", 250, 4000), ("g32n", "64x64", 50, 8230),   # 32 < M <= 64 and more than a round of skinny workgroups: tiles
    ("g32", "64x128", 500, 4090), ("g64", "128x128", 700, 4090),
    ("g128", "128x256", 1500, 4000), ("g256", "256x256", 4000, 4090),
]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("ns,tile,M,N", TILE_KERNELS, ids=[f"{t[0]}-{t[2]}" for t in TILE_KERNELS])
def test_every_tile_kernel(dev, ns, tile, M, N, wmode):
    import torch
    from micromix_amd import _lib
    K, split = 640, (256, 128, 256)
    desc = _lib.load().mm_matmul_describe(M, N, *split, 1 if wmode == "w4" else 0, 0, 0).decode()
    assert f"mm::{ns}::" in desc and f"({tile} tiles" in desc, desc
    rng = np.random.default_rng(M + N)
    xb = make_inputs(rng, M, K)
    wb = make_inputs(rng, N, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    for rounding in ("reference", "fused"):
        d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=rounding, split_k=False)
        rows = np.unique(np.concatenate([rng.choice(M, 24, replace=False), [0, 63, 64, 127, 128, M - 1]]))
        rows = rows[rows < M]
        qx = o.reorder_quantize(xb[rows], idx, *split, "x")
        check_gemm(bits_from_t(d)[rows], qx, [u8(t) for t in b], rounding, label=f"{ns} {M}x{N} {wmode} {rounding}")


# The second weight-streaming kernel (mx_gemm_stream.hip, M <= 32): every instantiation -- 16 / 32 features per workgroup x one / two
# 16-token tiles x both weight modes -- on all rows, with the cases its ring has to get right: fewer slabs than waves (idle waves),
# slab counts that need 1 .. D - 1 phantom steps, a lone fp6 slab between long segments, absent segments, N that is not a multiple of
# the workgroup's features, a long K (14 slabs per wave), bias, the fp32 output and the single-rounding mode.
STREAM_CASES = [
    # (M, N, split)                          what it exercises
    (9, 4128, (256, 128, 256)),            # narrow (16 features), 5 slabs < 8 waves
    (16, 272, (1024, 128, 896)),           # narrow, 16 slabs: two per wave, N = 17 x 16
    (12, 200, (128, 0, 0)),                # one slab, N % 16 != 0
    (3, 8200, (2048, 128, 1920)),          # wide (32 features, N / 32 > CUs), 32 slabs, ragged last workgroup (8200 = 256 x 32 + 8)
    (16, 8448, (0, 0, 1152)),              # wide, fp8 only, 9 slabs: one wave has two
    (14, 8256, (640, 0, 0)),               # wide, fp4 only
    (11, 8192, (0, 1280, 0)),              # wide, fp6 only (the 1.5-piece tiles)
    (1, 1024, (12288, 1024, 1024)),        # down_proj's K: 112 slabs, 14 per wave
    (17, 528, (384, 128, 640)),            # two token tiles, narrow, 9 slabs
    (32, 8224, (512, 128, 384)),           # two token tiles, wide, ragged N
    (25, 4100, (0, 256, 2048)),            # two token tiles, no fp4 segment
    (32, 300, (3072, 896, 128)),           # two token tiles, 32 slabs
    (5, 16640, (256, 128, 256)),           # more than one round of 32-feature workgroups: 64 features x 4 waves (F = 4), half tiles
    (16, 16424, (128, 256, 1024)),         # ... full tiles, ragged last workgroup (16424 = 256 x 64 + 40)
]
# 32 < M <= 64 (three / four token tiles; the dispatch sends only some of these shapes to the streaming kernel, so it is forced through
# MICROMIX_MID_M_STREAM in a child process: tests/test_stream_mid_m_gpu.py)


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,split", STREAM_CASES, ids=[f"{c[0]}x{c[1]}-{'_'.join(map(str, c[2]))}" for c in STREAM_CASES])
def test_weight_streaming_kernel(dev, m, n, split, wmode):
    import torch
    from micromix_amd import _lib
    k = sum(split)
    desc = _lib.load().mm_matmul_describe(m, n, *split, 1 if wmode == "w4" else 0, 0, 0).decode()
    assert "mx_gemm_stream_kernel" in desc, desc
    rng = np.random.default_rng(m * 131 + n)
    qx, qw = quantized(rng, m, n, k, split, wmode)
    for rounding in ("reference", "fused"):
        got = gpu_matmul(dev, qx, qw, rounding=rounding)
        check_gemm(got, qx, qw, rounding, label=f"stream {m}x{n}x{k} {split} {wmode} {rounding}")
    # bias: the reference's two roundings (qLinearLayer.py:70-71), as a separate add on the GPU result
    a, b = to_dev(dev, qx), to_dev(dev, qw)
    bias = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(torch.bfloat16).to(dev)
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    plain = mixedgemm.matmul(*args)
    with_bias = mixedgemm.matmul(*args, bias=bias)
    assert torch.equal(with_bias, (plain.float() + bias.float()).to(torch.bfloat16))
    # fp32 partial sums (tensor-parallel shards): rounding them once gives the single-rounding result
    f32 = mixedgemm.matmul(*args, rounding="fused", out_dtype=torch.float32)
    assert torch.equal(f32.to(torch.bfloat16), mixedgemm.matmul(*args, rounding="fused"))


# Chained segment hand-over of the 256-row tile (run_slabs_big, fp4 weights): every way a segment can pass its successor's first slabs
# on -- fp4 ring of 2 / 3 / 6 slabs into S or straight into O, successors shorter than, equal to and longer than their ring, S -> O
# through one ring, a segment shorter than the ring that owes its successor's slabs, and the cases that must NOT chain (odd fp4
# width, fewer than two 256-deep slabs).  Both weight modes (the matching-precision kernel keeps every segment's own prologue).
CHAIN_SPLITS = [(512, 128, 128), (512, 256, 384), (768, 0, 512), (1024, 384, 0), (640, 128, 256), (0, 384, 512), (0, 128, 128),
                (512, 128, 640), (1536, 128, 128), (256, 128, 256), (512, 0, 128), (512, 384, 128), (0, 128, 1024), (1536, 512, 0)]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("split", CHAIN_SPLITS, ids=["_".join(map(str, c)) for c in CHAIN_SPLITS])
def test_chained_segments_on_256_row_tiles(dev, split, wmode):
    import torch
    from micromix_amd import _lib
    M, N, K = 3900, 4090, sum(split)
    desc = _lib.load().mm_matmul_describe(M, N, *split, 1 if wmode == "w4" else 0, 0, 0).decode()
    assert "mm::g256::" in desc, desc
    rng = np.random.default_rng(K + split[0] + 7 * split[1])
    xb = make_inputs(rng, M, K)
    wb = make_inputs(rng, N, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    rows = np.unique(np.concatenate([rng.choice(M, 20, replace=False), [0, 255, 256, M - 1]]))
    qx = o.reorder_quantize(xb[rows], idx, *split, "x")
    qw = [u8(t) for t in b]
    wdeq = o.dequant_operand(qw, "w", wmode)
    for rounding in ("reference", "fused"):
        d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=rounding)
        assert torch.equal(d, mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=rounding))
        check_gemm(bits_from_t(d)[rows], qx, qw, rounding, label=f"chained {split} {wmode} {rounding}", wdeq=wdeq)


def _boundary_shapes():
    """token / feature counts on both sides of every dispatch threshold of mx_gemm.hip / plan_tiles (16 | 32 | 64 rows for the
    skinny kernels, N / 32 against the CUs, the 64x64 / 64x128 / 128x128 / 128x256 tile rounds), with ragged edges"""
    shapes = []
    for m in (15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 129, 191, 193, 255, 257):
        for n in (40, 264, 4104, 8200):
            shapes.append((m, n))
    shapes += [(256, 4096), (257, 4096), (512, 4096), (513, 4100), (385, 4096), (1025, 1000), (640, 2060)]
    # M <= 32 and more than three rounds of skinny workgroups (fused gate + up): 32 x 64 tiles; (17, 24576) stays on the skinny kernel
    shapes += [(1, 24608), (8, 24608), (16, 24608), (17, 24608), (32, 24608), (17, 24576)]
    return shapes


@pytest.mark.parametrize("m,n", _boundary_shapes(), ids=[f"{m}x{n}" for m, n in _boundary_shapes()])
def test_dispatch_boundaries(dev, m, n):
    """every kernel family at its edges: oracle on a row sample over every column, w4 weights, reference rounding, all three
    segments; the 8 or fewer features past a multiple of 8 take the scalar store path"""
    import torch
    K, split = 384, (128, 128, 128)
    rng = np.random.default_rng(m * 131 + n)
    xb = make_inputs(rng, m, K)
    wb = make_inputs(rng, n, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = mixedgemm.reorder_quantize_w4(w, tidx, *split)
    d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    rows = np.unique(np.concatenate([rng.choice(m, min(m, 20), replace=False), [0, m - 1, min(m - 1, 63), min(m - 1, 64)]]))
    qx = o.reorder_quantize(xb[rows], idx, *split, "x")
    check_gemm(bits_from_t(d)[rows], qx, [u8(t) for t in b], "reference", label=f"boundary {m}x{n}")


STRESS_REGRESSIONS = [   # found by tools/stress.py: many fp6 slabs at a token count that runs on few tiles, both weight modes
    (300, 4096, (0, 1792, 0)), (300, 4128, (1280, 2176, 640)), (300, 4096, (2176, 1792, 128)), (500, 4090, (0, 2048, 2048)),
]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,split", STRESS_REGRESSIONS, ids=[f"{c[0]}x{c[1]}-{'_'.join(map(str, c[2]))}" for c in STRESS_REGRESSIONS])
def test_many_slabs_on_few_tiles(dev, m, n, split, wmode):
    import torch
    K = sum(split)
    rng = np.random.default_rng(m + n + K)
    xb = make_inputs(rng, m, K)
    wb = make_inputs(rng, n, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    d = mixedgemm.matmul(*args, split_k=False)
    for _ in range(5):
        assert torch.equal(mixedgemm.matmul(*args, split_k=False), d)          # run-to-run determinism
    rows = np.unique(np.concatenate([rng.choice(m, 24, replace=False), [0, 63, 64, m - 1]]))
    qx = o.reorder_quantize(xb[rows], idx, *split, "x")
    check_gemm(bits_from_t(d)[rows], qx, [u8(t) for t in b], "reference", label=f"{m}x{n} {split} {wmode}")


def test_tail_balanced_launch(dev):
    """more 256x256 tiles than CUs with a small remainder: the launcher runs the last tile columns as 128-row tiles (two
    launches).  M=2048, N=8448 -> 8 x 33 = 264 tiles = 256 + one column.  Oracle on a row sample, every column."""
    import torch
    rng = np.random.default_rng(4)
    M, N, K, split = 2048, 8448, 256, (128, 0, 128)
    xb = make_inputs(rng, M, K)
    wb = make_inputs(rng, N, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = mixedgemm.reorder_quantize_w4(w, tidx, *split)
    d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    rows = np.sort(np.concatenate([rng.choice(M, 40, replace=False), [0, 127, 128, 255, 256, M - 1]]))
    qx = o.reorder_quantize(xb[rows], idx, *split, "x")
    check_gemm(bits_from_t(d)[rows], qx, [u8(t) for t in b], "reference", label="tail-balanced 2048x8448")
    # a row block computed alone (one launch, 128-row tiles) equals the same rows of the two-launch product
    sub = mixedgemm.reorder_quantize_x(x[300:500].contiguous(), tidx, *split)
    assert torch.equal(mixedgemm.matmul(sub[0], b[0], sub[1], b[1], sub[2], b[2], sub[3], b[3], sub[4], b[4], sub[5], b[5],
                                        split_k=False), d[300:500])


def test_errors(dev):
    import torch
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="BS has shape"):
        mixedgemm.matmul(z(4, 64), z(8, 64), z(4, 96), z(8, 50), z(4, 0), z(8, 0), z(512), z(512), z(512), z(512), z(0), z(0))
    with pytest.raises(RuntimeError, match="scale bytes"):
        mixedgemm.matmul(z(4, 64), z(8, 64), z(4, 0), z(8, 0), z(4, 0), z(8, 0), z(4), z(512), z(0), z(0), z(0), z(0))
    d = mixedgemm.matmul(z(4, 0), z(8, 0), z(4, 0), z(8, 0), z(4, 0), z(8, 0), z(0), z(0), z(0), z(0), z(0), z(0))
    assert d.shape == (4, 8) and float(d.abs().sum()) == 0.0                 # K == 0 -> zeros (bindings.cpp:72)


def test_minimal_scale_tensors_and_alignment(dev):
    """activation scale tensors of the minimal size ceil(M/128) row tiles (the reference allocates M/128 + 1, bindings.cpp:120)
    are enough for every kernel, also when M % 256 == 128 (second half of a 256-row tile absent); unaligned views are rejected"""
    import torch
    rng = np.random.default_rng(31)
    for m, n, k, split in ((128, 512, 512, (256, 128, 128)), (384, 2048, 256, (128, 0, 128)), (1152, 4096, 256, (0, 0, 256))):
        qx, qw = quantized(rng, m, n, k, split, "w4")
        widths = split
        small = list(qx[:3]) + [sf[: (m + 127) // 128 * 128 * (w // 32)].copy() for sf, w in zip(qx[3:], widths)]
        a, b = to_dev(dev, small), to_dev(dev, qw)
        # place each minimal scale tensor at the very end of its own allocation-sized buffer region: a read past it would
        # leave the tensor (caught by the checker below only through wrong results, so compare with the full-size run)
        full = gpu_matmul(dev, qx, qw)
        got = bits_from_t(mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5]))
        assert np.array_equal(got, full)
        check_gemm(got, qx, qw, "reference", label=f"minimal SF {m}x{n}x{k}")
    z = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
    big = z(4, 64 + 16)
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        mixedgemm.matmul(big.view(-1)[1:257].view(4, 64), z(8, 64), z(4, 0), z(8, 0), z(4, 0), z(8, 0), z(512), z(512), z(0), z(0), z(0), z(0))


def test_two_devices_in_one_process(dev):
    """the reference's --multi_gpu mode (parallel_utils.py:135-156) places layers of ONE process on several devices: every
    device gets its own dynamic-LDS attribute and CU count (the launchers cache both per device id)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    rng = np.random.default_rng(41)
    qx, qw = quantized(rng, 300, 512, 1024, (512, 128, 384), "w4")
    outs = []
    for d in (1, 0, 1):
        dd = torch.device("cuda", d)
        outs.append(gpu_matmul(dd, qx, qw))
        x = t_from_bits(make_inputs(rng, 4, 1024), dd)          # decode + quantizer launchers on that device too
        idx = torch.arange(1024, dtype=torch.int16, device=dd)
        mixedgemm.rmsnorm_quantize_x(x, x[0].contiguous(), 1e-5, idx, 512, 128, 384)
        torch.cuda.synchronize(dd)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    check_gemm(outs[0], qx, qw, "reference", label="two devices")


