"""GPU test of the operator boundary: QLinearLayer (mirror of model/qLinearLayer.py) against the oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import bits_from_t, t_from_bits
from micromix_amd.qlinear import QLinearLayer, find_qlinear_layers
from oracle import mx_oracle as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu


def test_qlinear_forward_golden(dev):
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    x5, w5, b5, i5 = mg.g5_inputs()
    p4, p6, p8 = g["g5_split"].tolist()
    for with_bias, key in ((True, "g5_y"), (False, "g5_y_nobias")):
        lin = torch.nn.Linear(1024, 256, bias=with_bias, dtype=torch.bfloat16)
        lin.weight.data = t_from_bits(w5, "cpu")
        if with_bias:
            lin.bias.data = t_from_bits(b5, "cpu")
        q = QLinearLayer(lin.to(dev), p8_num=p8, p6_num=p6, reorder_index=torch.from_numpy(i5.astype(np.int64)))
        assert (q.p4_num, q.p6_num, q.p8_num) == (p4, p6, p8) and q.BS.shape == (256, p6 // 2)
        y = q(t_from_bits(x5, dev).reshape(2, 20, 1024))
        assert y.shape == (2, 20, 256) and y.dtype == torch.bfloat16
        got = bits_from_t(y.reshape(40, 256))
        ulp = o.bf16_ulp_distance(got, g[key])
        assert (ulp > 1).mean() < 5e-3 and np.abs(o.bf16_to_f32(got) - o.bf16_to_f32(g[key])).max() < 0.25
        assert list(find_qlinear_layers(torch.nn.Sequential(q))) == ["0"]


def test_prequantized_tuple_input(dev):
    """qMixtralLayer.py:289-295 calling convention: one quantization shared by several projections."""
    import torch
    g = torch.Generator().manual_seed(1)
    k, split = 512, (256, 128, 128)
    x = torch.randn((2, 9, k), generator=g).to(torch.bfloat16).to(dev)
    idx = torch.randperm(k, generator=g)
    layers = []
    for n in (256, 128):
        lin = torch.nn.Linear(k, n, bias=False, dtype=torch.bfloat16)
        layers.append(QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx))
    shared = layers[0].quantize_input(x)
    assert len(shared) == 8 and shared[6:] == (2, 9)
    from conftest import u8
    from gemm_check import check_gemm
    ref_q = o.reorder_quantize(bits_from_t(x.reshape(18, k)), u8(layers[0].reorder_index), *split, "x")
    for q in layers:
        y = q(shared)
        assert torch.equal(y, q(x))
        # against the oracle: quantize-x of the same input, the layer's packed weight, reference rounding
        check_gemm(bits_from_t(y.reshape(18, -1)), ref_q, [u8(t) for t in (q.BN, q.BS, q.BO, q.SFBN, q.SFBS, q.SFBO)], "reference",
                   label="tuple input", strict=True)
    with pytest.raises(RuntimeError, match="different"):
        QLinearLayer(torch.nn.Linear(k, 128, bias=False, dtype=torch.bfloat16).to(dev), p8_num=256, p6_num=128, reorder_index=idx)(shared)


@pytest.mark.parametrize("m", (1, 48, 200))
def test_fused_layers_equal_separate_layers(dev, m):
    """FusedQLinear([q, k, v]) = the three layers, bit for bit (bias on some, none on others)."""
    import torch
    from micromix_amd.qlinear import FusedQLinear
    g = torch.Generator().manual_seed(5)
    k, split = 1024, (512, 128, 384)
    idx = torch.randperm(k, generator=g)
    layers = []
    for n, bias in ((512, True), (128, False), (256, True)):
        lin = torch.nn.Linear(k, n, bias=bias, dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_((torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16))
            if bias:
                lin.bias.copy_(torch.randn((n,), generator=g).to(torch.bfloat16))
        layers.append(QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx))
    fused = FusedQLinear(layers)
    x = torch.randn((1, m, k), generator=g).to(torch.bfloat16).to(dev)
    outs = fused(x)
    for layer, y in zip(layers, outs):
        ref = layer(x)
        if layer.bias is None and fused.bias is not None:
            # the fused bias vector holds zeros for this layer: y = bf16(bf16(acc) + 0) = the same value
            pass
        assert torch.equal(y, ref)


def test_expert_style_tuple_with_bsz_none(dev):
    """qMixtralLayer.py:507-519: experts quantize a 2-D token batch themselves and pass (…, None, q_len)"""
    import torch
    from micromix_amd import mixedgemm
    g = torch.Generator().manual_seed(9)
    k, split = 512, (256, 128, 128)
    idx = torch.randperm(k, generator=g)
    layer = QLinearLayer(torch.nn.Linear(k, 256, bias=False, dtype=torch.bfloat16).to(dev), p8_num=split[2], p6_num=split[1],
                         reorder_index=idx)
    x = torch.randn((37, k), generator=g).to(torch.bfloat16).to(dev)
    q = mixedgemm.reorder_quantize_x(x, layer.reorder_index, *split)
    y = layer((*q, None, 37))
    assert y.shape == (37, 256) and torch.equal(y, layer(x.unsqueeze(0))[0])
    from conftest import u8
    from gemm_check import check_gemm
    check_gemm(bits_from_t(y), o.reorder_quantize(bits_from_t(x), u8(layer.reorder_index), *split, "x"),
               [u8(t) for t in (layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS, layer.SFBO)], "reference", label="expert tuple", strict=True)


def test_call_plan_follows_replaced_weights(dev):
    """the cached C-ABI call plan must not outlive the packed tensors it points at"""
    import torch
    g = torch.Generator().manual_seed(12)
    k, split = 512, (256, 128, 128)
    idx = torch.randperm(k, generator=g)
    mk = lambda: QLinearLayer(torch.nn.Linear(k, 128, bias=False, dtype=torch.bfloat16).to(dev), p8_num=split[2], p6_num=split[1],
                              reorder_index=idx)
    a, b = mk(), mk()
    x = torch.randn((1, 4, k), generator=g).to(torch.bfloat16).to(dev)
    ya, yb = a(x), b(x)
    assert not torch.equal(ya, yb)
    for name in ("BN", "BS", "BO", "SFBN", "SFBS", "SFBO"):      # a takes b's weights
        setattr(a, name, getattr(b, name))
    assert torch.equal(a(x), yb)
    x2 = torch.randn((1, 40, k), generator=g).to(torch.bfloat16).to(dev)
    assert torch.equal(a(x2), b(x2))


@pytest.mark.parametrize("m", (4, 200))
def test_fused_rounding_option(dev, m):
    """QLinearLayer(rounding="fused"): one bf16 rounding instead of the reference's per-segment chain, on every forward path
    (fused decode kernel for m <= 8, quantize + GEMM otherwise, tuple input), against the oracle's fused chain"""
    import torch
    from conftest import u8
    from gemm_check import check_gemm
    g = torch.Generator().manual_seed(21)
    k, n, split = 1024, 384, (512, 128, 384)
    idx = torch.randperm(k, generator=g)
    lin = torch.nn.Linear(k, n, bias=True, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_((torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16))
        lin.bias.copy_(torch.randn((n,), generator=g).to(torch.bfloat16))
    ref_layer = QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx)
    layer = QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx, rounding="fused")
    x = torch.randn((1, m, k), generator=g).to(torch.bfloat16).to(dev)
    y = layer(x)
    qx = o.reorder_quantize(bits_from_t(x[0]), u8(layer.reorder_index), *split, "x")
    qw = [u8(t) for t in (layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS, layer.SFBO)]
    check_gemm(bits_from_t(y[0]), qx, qw, "fused", label=f"fused rounding m={m}", bias_bits=bits_from_t(layer.bias), strict=m > 8)
    assert torch.equal(layer(layer.quantize_input(x)), y)
    assert not torch.equal(ref_layer(x), y) or m < 8          # the two chains differ somewhere on 200 x 384 outputs
    with pytest.raises(ValueError):
        QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx, rounding="nearest")
