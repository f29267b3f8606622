"""a11: the call pattern of one decoder layer on the GPU, stage by stage against the oracle.

    rmsnorm_quantize_x(x)  -> 8-tuple -> FusedQLinear([q, k, v])          (qLlamaLayer.py:265-269; Qwen: q/k/v bias)
    o_proj(attn)                                                          (qLlamaLayer.py:316)
    rmsnorm_quantize_x(h)  -> 8-tuple -> FusedQLinear([gate, up])         (qLlamaLayer.py:377)
    activate_quantize_x(gate, up) -> matmul with downproj_quantize_w4-packed down weight   (qLlamaLayer.py:387)
plus f3: the calibration files (`saved/{model}_{reorder_index,p6_num,p8_num}_wikitext2.pt`, main.py:114-124) written,
loaded and fed into QLinearLayers for `llama_keys(1)`.

Every stage takes the GPU's own previous result as input and is compared with the oracle applied to that same input
("teacher forcing": the per-op tolerances stay meaningful along the chain); the end of the chain is additionally compared
with the oracle chain run on its own intermediates, at the noise level of the MX formats.  Attention itself (RoPE, softmax,
KV cache) is outside the hot path (section 8: out of scope); the q projection stands in for its output.
Sizes: Llama-3-8B widths (hidden 4096, kv 1024, intermediate 14336), 96 token rows.
"""
import numpy as np
import pytest

from conftest import bits_from_t, t_from_bits, u8
from gemm_check import check_gemm
from micromix_amd import calib, mixedgemm
from micromix_amd.qlinear import FusedQLinear, QLinearLayer
from model_case import gen_bf16, gen_index
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu

H, KV, INTER, M = 4096, 1024, 14336, 96
ATTN_SPLIT = (2048, 128, 1920)
O_SPLIT = (3072, 896, 128)
DOWN_SPLIT = (12288, 1024, 1024)
EPS = 1e-5


def _linear(dev, n, k, seed, bias):
    import torch
    lin = torch.nn.Linear(k, n, bias=bias, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        lin.weight.copy_(gen_bf16(dev, n, k, seed, "w"))
        if bias:
            lin.bias.copy_(gen_bf16(dev, 1, n, seed + 1, "x")[0] * 0.1)
    return lin


def _host(q):
    return [u8(t) for t in q]


def _layer_host(layer):
    return [u8(t) for t in (layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS, layer.SFBO)]


_DEQ = {}


def _deq(host):
    """dequantised fp64 form of a packed weight, cached by identity of its first array (one per layer and test)"""
    key = id(host[0])
    if key not in _DEQ:
        _DEQ[key] = (host, o.dequant_operand(host, "w", "w4"))
    return _DEQ[key][1]


def _omm(a, b, **kw):
    return o.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], b_dequant=_deq(b), **kw)


def _check_quant_tuple(got, want, rows, split, label):
    for i in range(3):
        if not split[i]:
            continue
        assert np.array_equal(got[i], want[i]), f"{label}: packed segment {i}"
        offs = o.sf_valid_offsets(rows, split[i])
        assert np.array_equal(got[3 + i][offs], want[3 + i][offs]), f"{label}: scales of segment {i}"


@pytest.mark.parametrize("qkv_bias", (False, True), ids=("llama", "qwen_bias"))
def test_decoder_layer_chain(dev, qkv_bias):
    import torch
    idx_attn = gen_index(dev, H, 1)
    idx_o = gen_index(dev, H, 2)
    x = gen_bf16(dev, M, H, 3)
    norm1 = (1.0 + 0.1 * gen_bf16(dev, 1, H, 4, "w")[0].float() * 50).to(torch.bfloat16)
    norm2 = (1.0 + 0.1 * gen_bf16(dev, 1, H, 5, "w")[0].float() * 50).to(torch.bfloat16)
    mk = lambda n, k, seed, bias, idx, split: QLinearLayer(_linear(dev, n, k, seed, bias), p8_num=split[2], p6_num=split[1],
                                                          reorder_index=idx.long())
    q, k_, v = (mk(n, H, s, qkv_bias, idx_attn, ATTN_SPLIT) for n, s in ((H, 10), (KV, 20), (KV, 30)))
    o_proj = mk(H, H, 40, False, idx_o, O_SPLIT)
    gate, up = (mk(INTER, H, s, False, idx_attn, ATTN_SPLIT) for s in (50, 60))
    w_down = gen_bf16(dev, H, INTER, 70, "w")
    qkv, gate_up = FusedQLinear([q, k_, v]), FusedQLinear([gate, up])
    _DEQ.clear()
    hw = {layer: _layer_host(layer) for layer in (q, k_, v, o_proj, gate, up)}

    # ---- stage 1: RMSNorm + reorder + quantize, shared by q/k/v --------------------------------------------------------
    t1 = mixedgemm.rmsnorm_quantize_x(x, norm1, EPS, idx_attn, *ATTN_SPLIT)
    ref1 = o.rmsnorm_quantize(bits_from_t(x), bits_from_t(norm1), EPS, u8(idx_attn), *ATTN_SPLIT)
    _check_quant_tuple(_host(t1), ref1, M, ATTN_SPLIT, "rmsnorm_quantize_x (attn)")
    # ---- stage 2: q/k/v as ONE GEMM on the shared tuple ---------------------------------------------------------------
    yq, yk, yv = qkv((*t1, 1, M))
    for name, layer, y in (("q", q, yq), ("k", k_, yk), ("v", v, yv)):
        assert y.shape == (1, M, layer.out_features)
        check_gemm(bits_from_t(y[0]), ref1, hw[layer], "reference", label=f"{name}_proj", strict=True, wdeq=_deq(hw[layer]),
                   bias_bits=bits_from_t(layer.bias) if layer.bias is not None else None)
        assert torch.equal(layer((*t1, 1, M)), y)                      # the separate layer on the same tuple: bit-identical
    # ---- stage 3: o_proj on a tensor input (its own quantization) ------------------------------------------------------
    attn = yq[0]                                                       # stand-in for the attention output [M, H]
    yo = o_proj(attn.reshape(1, M, H))[0]
    ref_qo = o.reorder_quantize(bits_from_t(attn), u8(idx_o), *O_SPLIT, "x")
    check_gemm(bits_from_t(yo), ref_qo, hw[o_proj], "reference", label="o_proj", strict=True, wdeq=_deq(hw[o_proj]))
    # ---- stage 4: second RMSNorm -> gate/up --------------------------------------------------------------------------
    h = (x.float() + yo.float()).to(torch.bfloat16)                    # residual add (outside the path, plain torch)
    t2 = mixedgemm.rmsnorm_quantize_x(h, norm2, EPS, idx_attn, *ATTN_SPLIT)
    ref2 = o.rmsnorm_quantize(bits_from_t(h), bits_from_t(norm2), EPS, u8(idx_attn), *ATTN_SPLIT)
    _check_quant_tuple(_host(t2), ref2, M, ATTN_SPLIT, "rmsnorm_quantize_x (mlp)")
    yg, yu = gate_up((*t2, 1, M))
    check_gemm(bits_from_t(yg[0]), ref2, hw[gate], "reference", label="gate_proj", strict=True, wdeq=_deq(hw[gate]))
    check_gemm(bits_from_t(yu[0]), ref2, hw[up], "reference", label="up_proj", strict=True, wdeq=_deq(hw[up]))
    # ---- stage 5: silu(gate) * up + quantize (natural order), down_proj packed with downproj_quantize_w4 -------------
    qh = mixedgemm.activate_quantize_x(yg[0].contiguous(), yu[0].contiguous(), *DOWN_SPLIT)
    ref_h = o.activate_quantize(bits_from_t(yg[0]), bits_from_t(yu[0]), *DOWN_SPLIT)
    got_h = _host(qh)
    for i in range(3):                                                 # stated budget of that op (hardware exp): see test_direct_quantize_gpu.py
        offs = o.sf_valid_offsets(M, DOWN_SPLIT[i])
        assert (got_h[i] != ref_h[i]).mean() < 1e-3 and (got_h[3 + i][offs] != ref_h[3 + i][offs]).mean() < 1e-3
    qd = mixedgemm.downproj_quantize_w4(w_down, *DOWN_SPLIT)
    ref_d = o.downproj_quantize(bits_from_t(w_down), *DOWN_SPLIT, True)
    _check_quant_tuple(_host(qd), ref_d, H, DOWN_SPLIT, "downproj_quantize_w4")
    y = mixedgemm.matmul(qh[0], qd[0], qh[1], qd[1], qh[2], qd[2], qh[3], qd[3], qh[4], qd[4], qh[5], qd[5])
    ref_d = list(ref_d)
    check_gemm(bits_from_t(y), got_h, ref_d, "reference", label="down_proj", strict=True, wdeq=_deq(ref_d))
    # ---- end to end: the oracle chain on its own intermediates --------------------------------------------------------
    bias_add = lambda d, layer: d if layer.bias is None else o.f32_to_bf16(o.bf16_to_f32(d) + o.bf16_to_f32(bits_from_t(layer.bias))[None, :])
    oq = bias_add(_omm(ref1, hw[q]), q)
    oo = _omm(o.reorder_quantize(oq, u8(idx_o), *O_SPLIT, "x"), hw[o_proj])
    oh = o.f32_to_bf16(o.bf16_to_f32(bits_from_t(x)) + o.bf16_to_f32(oo))
    r2 = o.rmsnorm_quantize(oh, bits_from_t(norm2), EPS, u8(idx_attn), *ATTN_SPLIT)
    og, ou = _omm(r2, hw[gate]), _omm(r2, hw[up])
    oy = o.bf16_to_f32(_omm(o.activate_quantize(og, ou, *DOWN_SPLIT), ref_d)).astype(np.float64)
    gy = y.float().cpu().numpy().astype(np.float64)
    assert np.linalg.norm(gy - oy) / np.linalg.norm(oy) < 0.03      # re-quantization amplifies 1-ulp differences of the intermediates
    _DEQ.clear()


def test_calibration_files_feed_qlinear_layers(dev, tmp_path):
    """f3: save_calibration -> the three .pt files -> load_calibration -> QLinearLayer per key of llama_keys(1) -> forward vs oracle
    (main.py:114-124 loads the dicts, qLlamaLayer.py:212-235,336-355 index them by these keys)."""
    import torch
    hid, inter, m = 512, 1024, 40
    shapes = {"self_attn.q_proj": (hid, hid), "self_attn.k_proj": (128, hid), "self_attn.v_proj": (128, hid),
              "self_attn.o_proj": (hid, hid), "mlp.gate_proj": (inter, hid), "mlp.up_proj": (inter, hid), "mlp.down_proj": (hid, inter)}
    keys = list(calib.llama_keys(1))
    ri, p6s, p8s, acts = {}, {}, {}, {}
    for i, key in enumerate(keys):
        n, k = shapes[key[len("layers.0."):-len(".input")]]
        a = gen_bf16(dev, 256, k, 100 + i).float().cpu()
        order, p4, p6, p8 = calib.split_from_activations(a)
        ri[key], p6s[key], p8s[key], acts[key] = order, p6, p8, (n, k)
    calib.save_calibration(str(tmp_path / "saved"), "Llama-3-8B", ri, p6s, p8s)
    ri2, p6b, p8b = calib.load_calibration(str(tmp_path / "saved"), "Llama-3-8B")
    assert list(ri2) == keys
    for i, key in enumerate(keys):
        n, k = acts[key]
        lin = _linear(dev, n, k, 200 + i, bias=False)
        layer = QLinearLayer(lin, p8_num=p8b[key], p6_num=p6b[key], reorder_index=ri2[key])
        split = (layer.p4_num, layer.p6_num, layer.p8_num)
        assert split == (k - p6s[key] - p8s[key], p6s[key], p8s[key]) and sum(split) == k
        x = gen_bf16(dev, m, k, 300 + i)
        y = layer(x.reshape(2, m // 2, k)).reshape(m, n)
        idx = ri2[key].numpy().astype(np.int16)
        qw = o.reorder_quantize(bits_from_t(lin.weight.data), idx, *split, "w4")
        _check_quant_tuple(_layer_host(layer), qw, n, split, f"{key} packed weight")
        check_gemm(bits_from_t(y), o.reorder_quantize(bits_from_t(x), idx, *split, "x"), qw, "reference", label=key, strict=True)
