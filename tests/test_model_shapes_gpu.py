"""GPU parity at the full shapes of BASELINE.json configs 2-5, through the C ABI:

  config 3  Llama-3-8B projections (N, K): q/o (4096, 4096), k/v (1024, 4096), gate/up (14336, 4096), down (4096, 14336)
            with the mixed splits of section 8d, M in {1, 16, 256, 4096};
  config 4  Qwen2.5-14B: q/o (5120, 5120) and k/v (1024, 5120) with bias, gate/up (13824, 5120), down (5120, 13824), through
            QLinearLayer (model/qLinearLayer.py:21-74; bias path :35-38,70-71), plus every rank's TP=4 K-shard;
  config 5  Mixtral-8x7B experts: w1/w3 (14336, 4096) split (3584, 256, 256), w2 (4096, 14336) split (12544, 1024, 768)
            through reorder_quantize_x_grouped + matmul_grouped (the per-expert loop of qMixtralLayer.py:507-519), plus
            every rank's TP=8 K-shard of one expert.

Method: tests/model_case.py (oracle on sampled rows of operands anchored byte-for-byte to the oracle quantizer, asserted
ulp statistics), plus the size-independent properties determinism and row-block independence.
"""
import numpy as np
import pytest

from conftest import bits_from_t, u8
from gemm_check import check_gemm
from micromix_amd import mixedgemm, tp
from micromix_amd.qlinear import QLinearLayer
from model_case import PackedWeight, assert_rows_match_oracle, check_rows, gen_bf16, gen_index, sample, sample_rows
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu

LLAMA = [
    ("q_o", 4096, 4096, (0, 0, 4096)), ("q_o", 4096, 4096, (4096, 0, 0)),       # the bench headline (BASELINE configs[1]) and all-fp4
    ("q_o", 4096, 4096, (2048, 128, 1920)), ("q_o", 4096, 4096, (3072, 896, 128)),
    ("k_v", 1024, 4096, (2048, 128, 1920)), ("k_v", 1024, 4096, (3072, 896, 128)), ("k_v", 1024, 4096, (0, 0, 4096)),
    ("gate_up", 14336, 4096, (2048, 128, 1920)), ("gate_up", 14336, 4096, (3072, 896, 128)), ("gate_up", 14336, 4096, (0, 0, 4096)),
    ("down", 4096, 14336, (7168, 512, 6656)), ("down", 4096, 14336, (12288, 1024, 1024)),
]


def _mm(a, b, **kw):
    return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)


@pytest.mark.parametrize("name,n,k,split", LLAMA, ids=[f"{c[0]}-{c[1]}x{c[2]}-{'_'.join(map(str, c[3]))}" for c in LLAMA])
def test_llama_projection(dev, name, n, k, split):
    import torch
    rng = np.random.default_rng(n + k + split[0])
    pw = PackedWeight(dev, n, k, split, seed=n * 3 + k + split[1], rng=rng)
    for m in (1, 16, 256, 4096):
        x = gen_bf16(dev, m, k, seed=m + k)
        qx = mixedgemm.reorder_quantize_x(x, pw.index, *split)
        d = _mm(qx, pw.packed)
        rows = sample_rows(rng, m, 24, always=(0, 127, 128, 255, m - 1))
        check_rows(d, x, qx, pw, rows, label=f"{name} M={m} {split}")
        assert torch.equal(d, _mm(qx, pw.packed))                                    # deterministic
        if m == 4096:
            # a row block computed alone (other tile size / kernel) equals the same rows of the full product
            r0, r1 = 1000, 1100
            sub = mixedgemm.reorder_quantize_x(x[r0:r1].contiguous(), pw.index, *split)
            assert torch.equal(_mm(sub, pw.packed, split_k=False), d[r0:r1])
            # fused rounding (one bf16 rounding) against the oracle's fused chain
            df = _mm(qx, pw.packed, rounding="fused")
            check_rows(df, x, None, pw, rows[:: max(1, len(rows) // 32)], rounding="fused", label=f"{name} M={m} {split} fused")
    del pw
    torch.cuda.empty_cache()


def test_llama_w_mode_full_shape(dev):
    """matching-precision weights ("w": fp4/fp6/fp8 weights, reorder_quantize_w) at one full Llama shape"""
    rng = np.random.default_rng(11)
    n, k, split = 4096, 4096, (2048, 128, 1920)
    pw = PackedWeight(dev, n, k, split, seed=5, wmode="w", rng=rng)
    for m in (16, 4096):
        x = gen_bf16(dev, m, k, seed=m)
        qx = mixedgemm.reorder_quantize_x(x, pw.index, *split)
        d = _mm(qx, pw.packed)
        check_rows(d, x, qx, pw, sample_rows(rng, m, 24, always=(0, m - 1)), label=f"w-mode M={m}")


QWEN = [  # (name, N, K, bias, split): hidden 5120, intermediate 13824
    ("q_o", 5120, 5120, True, (2560, 128, 2432)), ("k_v", 1024, 5120, True, (4352, 512, 256)),
    ("gate_up", 13824, 5120, False, (2560, 128, 2432)), ("down", 5120, 13824, False, (11776, 1024, 1024)),
]


def _qwen_layer(dev, n, k, bias, split, seed):
    import torch
    lin = torch.nn.Linear(k, n, bias=bias, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        lin.weight.copy_(gen_bf16(dev, n, k, seed, "w"))
        if bias:
            lin.bias.copy_(gen_bf16(dev, 1, n, seed + 7, "x")[0])
    idx = gen_index(dev, k, seed + 1)
    layer = QLinearLayer(lin, p8_num=split[2], p6_num=split[1], reorder_index=idx.long())
    return lin, layer


@pytest.mark.parametrize("name,n,k,bias,split", QWEN, ids=[c[0] for c in QWEN])
def test_qwen_qlinear_layer(dev, name, n, k, bias, split):
    """QLinearLayer.__init__ + forward at the Qwen2.5-14B shapes (q/k/v bias), against the oracle chain"""
    import torch
    rng = np.random.default_rng(n + k)
    lin, layer = _qwen_layer(dev, n, k, bias, split, seed=n + 2 * k)
    assert (layer.p4_num, layer.p6_num, layer.p8_num) == split
    pw = PackedWeight.__new__(PackedWeight)       # anchor the layer's own packed tensors to the oracle
    pw.n, pw.k, pw.split, pw.wmode, pw.index = n, k, split, "w4", layer.reorder_index
    pw.packed = (layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS, layer.SFBO)
    wrows = sample(rng, n, 24, always=(0, n - 1))
    ref = o.reorder_quantize(bits_from_t(lin.weight.data[torch.from_numpy(wrows).to(dev)]), u8(layer.reorder_index), *split, "w4")
    assert_rows_match_oracle(pw.packed, wrows, ref, split, f"qwen {name} W")
    pw.host = [u8(t) for t in pw.packed]
    pw.deq = o.dequant_operand(pw.host, "w", "w4")
    for bsz, q_len in ((1, 1), (2, 8), (1, 2048)):
        m = bsz * q_len
        x = gen_bf16(dev, m, k, seed=m + 3).reshape(bsz, q_len, k)
        y = layer(x)
        assert y.shape == (bsz, q_len, n) and y.dtype == torch.bfloat16
        rows = sample_rows(rng, m, 24, always=(0, m - 1))
        check_rows(y.reshape(m, n), x.reshape(m, k), None, pw, rows, label=f"qwen {name} M={m}", bias=layer.bias)
        # the 8-tuple input (qMixtralLayer.py:292) gives the same result
        assert torch.equal(layer(layer.quantize_input(x)), y)


@pytest.mark.parametrize("name,n,k,bias,split", QWEN, ids=[c[0] for c in QWEN])
def test_qwen_tp4_shards(dev, name, n, k, bias, split):
    """every rank's TP=4 K-shard at its real width (one GPU, rank after rank): each partial against the oracle on that shard,
    and the fp32 sum of the partials against the unsharded product"""
    _tp_shards(dev, n, k, split, world=4, m=512, seed=n + k + 4)


def _tp_shards(dev, n, k, split, world, m, seed):
    import torch
    rng = np.random.default_rng(seed)
    w = gen_bf16(dev, n, k, seed, "w")
    idx = gen_index(dev, k, seed + 1)
    x = gen_bf16(dev, m, k, seed + 2)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    full = _mm(a, b, rounding="fused").float()
    total = torch.zeros((m, n), dtype=torch.float32, device=dev)
    rows = sample_rows(rng, m, 16, always=(0, m - 1))
    ridx = torch.from_numpy(rows).to(dev)
    cols = 0
    for r in range(world):
        layer = tp.TPShardedLinear(w, idx, *split, rank=r, world=world)
        cols += sum(layer.shard_widths)
        if layer.empty:
            continue
        qx = layer.quantize_x(x)
        part = layer.ops.matmul(qx, layer.packed_w)
        total += part.float()
        widths = tuple(layer.shard_widths)
        ref_x = o.reorder_quantize(bits_from_t(x[ridx]), u8(layer.index), *widths, "x", gather_subset=True)
        assert_rows_match_oracle(qx, rows, ref_x, widths, f"tp{world} rank {r} X")
        check_gemm(bits_from_t(part[ridx]), ref_x, [u8(t) for t in layer.packed_w], tp.SHARD_ROUNDING,
                   label=f"tp{world} rank {r} widths {widths}", strict=True)
    assert cols == k
    # each partial is rounded to bf16 once (fused rounding per rank): |sum of partials - full| <= world half-ulps of the largest
    err = (total - full).abs()
    assert float(err.max()) <= world * 2.0 ** -8 * float(full.abs().max()) + 1e-3
    assert float(torch.linalg.norm(err) / torch.linalg.norm(full)) < 2.0 ** -8 * world ** 0.5


MIXTRAL = [("w1_w3", 14336, 4096, (3584, 256, 256)), ("w2", 4096, 14336, (12544, 1024, 768))]


@pytest.mark.parametrize("name,n,k,split", MIXTRAL, ids=[c[0] for c in MIXTRAL])
def test_mixtral_experts_grouped(dev, name, n, k, split):
    """8 experts, each with its own weights and reorder index, token counts from decode-sized to tile-sized, through the grouped
    quantizer and the grouped GEMM; every expert's sampled rows against the oracle"""
    import torch
    rng = np.random.default_rng(n)
    ms = (1, 37, 128, 300, 0, 64, 515, 70)
    experts = [PackedWeight(dev, n, k, split, seed=100 * e + n, rng=rng) for e in range(len(ms))]
    xs = [gen_bf16(dev, m, k, seed=e + 50) for e, m in enumerate(ms)]
    qs = mixedgemm.reorder_quantize_x_grouped(xs, [p.index for p in experts], *split)
    outs = mixedgemm.matmul_grouped(qs, [p.packed for p in experts])
    torch.cuda.synchronize()
    for e, (m, pw, x, q, d) in enumerate(zip(ms, experts, xs, qs, outs)):
        assert d.shape == (m, n)
        if m == 0:
            continue
        rows = sample_rows(rng, m, 12, always=(0, m - 1))
        check_rows(d, x, q, pw, rows, label=f"mixtral {name} expert {e} M={m}")
        pw.deq = None      # release ~0.5 GB per expert


@pytest.mark.parametrize("name,n,k,split", MIXTRAL, ids=[c[0] for c in MIXTRAL])
def test_mixtral_tp8_shards(dev, name, n, k, split):
    _tp_shards(dev, n, k, split, world=8, m=512, seed=n + 8)
