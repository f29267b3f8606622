"""GPU test of the grouped GEMM (MoE expert batching): every group against the oracle, and
bit-identical to one matmul per group.  (Full Mixtral expert shapes: tests/test_model_shapes_gpu.py.)"""
import pytest

from conftest import bits_from_t, u8
from gemm_check import check_gemm
from micromix_amd import mixedgemm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("n,k,split,ms", [
    (256, 512, (256, 128, 128), (3, 0, 17, 64, 1, 40, 8, 33, 5, 12)),        # ten groups: two launches, one empty group
    (4096, 1024, (512, 0, 512), (4, 4, 9, 2)),                               # 16-feature kernel (4 x 128 workgroups)
    (1024, 384, (128, 128, 128), (64, 100, 7, 300)),                         # small and large groups mixed
    (512, 256, (128, 0, 128), (128, 200, 65, 512, 300, 96, 1000, 70, 130)),  # nine large groups: two launches of the tiled kernels
    (2048, 512, (256, 128, 128), (700, 1100)),                               # enough tiles for the 256-row tiles
    (14336, 256, (0, 0, 256), (2, 31)),                                      # many workgroups: the 32-feature kernel
])
def test_grouped_equals_per_group_matmul(dev, wmode, n, k, split, ms):
    import torch
    g = torch.Generator().manual_seed(n + k + len(ms))
    As, Bs, biases, want = [], [], [], []
    quant_w = mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w
    for i, m in enumerate(ms):
        idx = torch.randperm(k, generator=g).to(torch.int16).to(dev)           # every expert has its own reorder index
        w = (torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16).to(dev)
        x = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
        b = quant_w(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        bias = torch.randn((n,), generator=g).to(torch.bfloat16).to(dev) if i % 2 else None
        As.append(a); Bs.append(b); biases.append(bias)
        want.append(mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], bias=bias, split_k=False))
    for rounding in ("reference", "fused"):
        if rounding == "fused":
            want = [mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], bias=bi, split_k=False,
                                     rounding="fused") for a, b, bi in zip(As, Bs, biases)]
        got = mixedgemm.matmul_grouped(As, Bs, biases=biases, rounding=rounding)
        torch.cuda.synchronize()
        assert len(got) == len(ms)
        for i, (y, ref) in enumerate(zip(got, want)):
            assert y.shape == ref.shape and torch.equal(y, ref), (i, ms[i], rounding)
            if ms[i]:      # the oracle on the same packed operands (quantizer parity: tests/test_quantize_gpu.py)
                check_gemm(bits_from_t(y), [u8(t) for t in As[i]], [u8(t) for t in Bs[i]], rounding, label=f"group {i} M={ms[i]} {rounding}",
                           bias_bits=bits_from_t(biases[i]) if biases[i] is not None else None)


def test_grouped_argument_checks(dev):
    import torch
    assert mixedgemm.matmul_grouped([], []) == []
    k, split = 256, (128, 0, 128)
    idx = torch.arange(k, dtype=torch.int16, device=dev)
    w = torch.zeros((128, k), dtype=torch.bfloat16, device=dev)
    x = torch.zeros((4, k), dtype=torch.bfloat16, device=dev)
    a, b = mixedgemm.reorder_quantize_x(x, idx, *split), mixedgemm.reorder_quantize_w4(w, idx, *split)
    other = mixedgemm.reorder_quantize_x(x, idx, 256, 0, 0)
    with pytest.raises(RuntimeError):
        mixedgemm.matmul_grouped([a, other], [b, b])
    with pytest.raises(ValueError):
        mixedgemm.matmul_grouped([a], [b, b])


@pytest.mark.parametrize("k,split,rows", [(512, (256, 128, 128), (3, 0, 17, 64, 1, 40, 8, 33, 5, 130)),
                                          (4096, (3584, 256, 256), (2, 2, 1, 9)), (14336, (12544, 1024, 768), (1, 4))])
def test_grouped_quantizer_equals_separate_calls(dev, k, split, rows):
    """reorder_quantize_x_grouped: every expert's rows with its own reorder index, byte for byte the separate calls"""
    import torch
    from conftest import u8
    from oracle import mx_oracle as o
    g = torch.Generator().manual_seed(k + len(rows))
    Xs = [torch.randn((r, k), generator=g).to(torch.bfloat16).to(dev) for r in rows]
    idxs = [torch.randperm(k, generator=g).to(torch.int16).to(dev) for _ in rows]
    got = mixedgemm.reorder_quantize_x_grouped(Xs, idxs, *split)
    torch.cuda.synchronize()
    assert len(got) == len(rows)
    for X, idx, q, r in zip(Xs, idxs, got, rows):
        want = mixedgemm.reorder_quantize_x(X, idx, *split)
        for i, (a, b) in enumerate(zip(q, want)):
            assert a.shape == b.shape
            if i < 3:
                assert torch.equal(a, b)
            elif r:
                off = o.sf_valid_offsets(r, split[i - 3])
                assert (u8(a)[off] == u8(b)[off]).all()
