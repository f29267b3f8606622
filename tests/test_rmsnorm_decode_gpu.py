"""`mm_rmsnorm_qlinear_decode` (round 5): the RMSNorm that precedes q/k/v and gate/up in the reference's decoder layers
(rmsnorm_quantize_x, mgemm/src/rmsnorm.cu:95-352, bindings.cpp:257-303) inside the decode launch.  Contract: bit-identical to
`rmsnorm_quantize_x` followed by `matmul`; and, directly, the oracle's rmsnorm quantizer + the oracle GEMM within tests/gemm_check.py."""
import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits, u8
from gemm_check import check_gemm
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu

CASES = [(1, 128, 128, (128, 0, 0)), (1, 256, 4096, (0, 0, 4096)), (3, 200, 1024, (512, 128, 384)), (8, 384, 2048, (1024, 512, 512)),
         (2, 160, 5120, (4096, 512, 512)), (5, 96, 384, (0, 384, 0)), (7, 1024, 512, (256, 0, 256)), (8, 256, 8192, (4096, 2048, 2048)),
         (2, 4128, 256, (128, 0, 128)), (3, 8230, 256, (128, 0, 128)),         # 32-feature kernel; more feature blocks than CUs
         # wide layers at M <= 4: the streaming kernel with the quantization (and now the norm) inside every workgroup
         (2, 8200, 4096, (2048, 128, 1920)), (4, 8192, 1024, (512, 128, 384)), (1, 8448, 1152, (0, 0, 1152)), (4, 8320, 5120, (4096, 512, 512)),
         (3, 8192, 2304, (0, 2304, 0)), (1, 28672, 4096, (2048, 128, 1920))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,k,split", CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in CASES])
def test_equals_rmsnorm_quantize_then_matmul(dev, wmode, m, n, k, split):
    import torch
    rng = np.random.default_rng(m * 17 + n + k)
    xb = make_inputs(rng, m, k)
    xb[0, :64] = 0
    wb = make_inputs(rng, n, k, "weight")
    nwb = o.f32_to_bf16((1.0 + 0.25 * rng.standard_normal(k)).astype(np.float32))
    idx = rng.permutation(k).astype(np.int16)
    bias = t_from_bits(o.f32_to_bf16(rng.standard_normal(n).astype(np.float32)), dev)
    x, w, nw, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), t_from_bits(nwb, dev), torch.from_numpy(idx).to(dev)
    assert mixedgemm.rmsnorm_qlinear_decode_supported(m, n, *split) >= 1
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    for ir in (True, False):
        a = mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, tidx, *split, integer_round=ir)
        for rounding, bv in (("reference", None), ("fused", bias)):
            want = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], bias=bv, rounding=rounding)
            got = mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, tidx, *b, *split, bias=bv, rounding=rounding, integer_round=ir)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (m, n, k, split, wmode, ir, rounding)
    # and against the oracle directly (small N only: the oracle dequantises the whole weight)
    if n <= 1024:
        qx = o.rmsnorm_quantize(xb, nwb, 1e-5, idx, *split, integer_round=True)
        qw = o.reorder_quantize(wb, idx, *split, wmode)
        got = mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, tidx, *b, *split)
        check_gemm(bits_from_t(got), qx, qw, "reference", label=f"rmsnorm decode {m}x{n}x{k} {split} {wmode}")


@pytest.mark.parametrize("n", (6144, 14336), ids=("q_k_v", "gate"))
def test_llama_shapes_against_the_oracle(dev, n):
    """hidden 4096 -> q | k | v (6144 features: the first fused kernel) and gate_proj (14336: the streaming kernel), M = 1 and 4"""
    import torch
    k, split = 4096, (2048, 128, 1920)
    rng = np.random.default_rng(n + 1)
    wb = make_inputs(rng, n, k, "weight")
    nwb = o.f32_to_bf16((1.0 + 0.25 * rng.standard_normal(k)).astype(np.float32))
    idx = rng.permutation(k).astype(np.int16)
    tidx = torch.from_numpy(idx).to(dev)
    qw = o.reorder_quantize(wb, idx, *split, "w4")
    b = mixedgemm.reorder_quantize_w4(t_from_bits(wb, dev), tidx, *split)
    for s in range(3):
        assert np.array_equal(u8(b[s]), qw[s])
    wdeq = o.dequant_operand(qw, "w", "w4")
    for m in (1, 4):
        assert mixedgemm.rmsnorm_qlinear_decode_supported(m, n, *split) >= 1
        xb = make_inputs(rng, m, k)
        got = mixedgemm.rmsnorm_qlinear_decode(t_from_bits(xb, dev), t_from_bits(nwb, dev), 1e-5, tidx, *b, *split)
        qx = o.rmsnorm_quantize(xb, nwb, 1e-5, idx, *split, integer_round=True)
        check_gemm(bits_from_t(got), qx, qw, "reference", label=f"rmsnorm decode M={m} N={n}", strict=True, wdeq=wdeq)


def test_supported_range_and_errors(dev):
    import torch
    assert mixedgemm.rmsnorm_qlinear_decode_supported(9, 4096, 0, 0, 4096) == 0
    assert mixedgemm.rmsnorm_qlinear_decode_supported(1, 4096, 4096, 4096, 4096) == 0        # K = 12288 > 8192: the tree covers 256 partial sums
    assert mixedgemm.rmsnorm_qlinear_decode_supported(1, 4096, 0, 0, 4096) >= 1
    k, split = 256, (128, 0, 128)
    x = torch.zeros((1, k), dtype=torch.bfloat16, device=dev)
    nw = torch.ones((k,), dtype=torch.bfloat16, device=dev)
    idx = torch.arange(k, dtype=torch.int16, device=dev)
    b = mixedgemm.reorder_quantize_w4(torch.ones((128, k), dtype=torch.bfloat16, device=dev), idx, *split)
    y = mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, idx, *b, *split)                       # an all-zero row: rvar = 1 / sqrt(eps), v = 0
    assert y.shape == (1, 128) and not y.any()
    with pytest.raises(RuntimeError):
        mixedgemm.rmsnorm_qlinear_decode(x, nw[: k - 8], 1e-5, idx, *b, *split)


def test_layers_forward_norm(dev):
    """QLinearLayer / FusedQLinear.forward_norm and FusedMLP.forward(x, norm_weight): RMSNorm -> layer, byte-identical to the reference's
    caller pattern (rmsnorm_quantize_x -> forward(tuple)) at decode and prefill sizes"""
    import torch
    from micromix_amd.qlinear import FusedMLP, FusedQLinear, QLinearLayer
    g = torch.Generator().manual_seed(5)
    k, inter, p8, p6 = 1024, 2048, 384, 128
    idx = torch.randperm(k, generator=g)
    lin = lambda n, kk=k, b=False: torch.nn.Linear(kk, n, bias=b, dtype=torch.bfloat16).to(dev)
    q, kv = QLinearLayer(lin(512, b=True), p8, p6, idx), QLinearLayer(lin(256), p8, p6, idx)
    fused = FusedQLinear([q, kv])
    gate, up = QLinearLayer(lin(inter), p8, p6, idx), QLinearLayer(lin(inter), p8, p6, idx)
    mlp = FusedMLP(gate, up, (torch.randn((k, inter), generator=g) * 0.02).to(torch.bfloat16), (1536, 256, 256))
    nw = (1.0 + 0.2 * torch.randn((k,), generator=g)).to(torch.bfloat16).to(dev)
    for m in (1, 2, 8, 70):
        x = torch.randn((1, m, k), generator=g).to(torch.bfloat16).to(dev)
        tup = (*mixedgemm.rmsnorm_quantize_x(x.reshape(m, k), nw, 1e-5, q.reorder_index, q.p4_num, q.p6_num, q.p8_num), 1, m)
        assert torch.equal(q.forward_norm(x, nw, 1e-5), q(tup))
        yq, ykv = fused.forward_norm(x, nw, 1e-5)
        assert torch.equal(yq, q(tup)) and torch.equal(ykv, kv(tup))
        qh = mixedgemm.activate_quantize_x(gate(tup).reshape(m, inter), up(tup).reshape(m, inter), *mlp.down_split)
        want = mixedgemm.matmul(qh[0], mlp.D_BN, qh[1], mlp.D_BS, qh[2], mlp.D_BO, qh[3], mlp.D_SFBN, qh[4], mlp.D_SFBS, qh[5], mlp.D_SFBO)
        assert torch.equal(mlp(x, nw, 1e-5).reshape(m, k), want)


def test_forward_norm_checks_empty_batches_and_2d_input(dev):
    """forward_norm validates like forward (dtype / device / K), answers an empty batch without a launch, accepts the 2-D
    token form of the Mixtral caller, and its fallback runs through the layer's plan (mid-M shapes get the split-K workspace)"""
    import torch
    from micromix_amd.qlinear import QLinearLayer
    g = torch.Generator().manual_seed(9)
    k, p8, p6 = 1024, 384, 128
    idx = torch.randperm(k, generator=g)
    layer = QLinearLayer(torch.nn.Linear(k, 256, bias=True, dtype=torch.bfloat16).to(dev), p8, p6, idx)
    nw = (1.0 + 0.2 * torch.randn((k,), generator=g)).to(torch.bfloat16).to(dev)
    y = layer.forward_norm(torch.empty((2, 0, k), dtype=torch.bfloat16, device=dev), nw, 1e-5)
    assert y.shape == (2, 0, 256)
    with pytest.raises(TypeError):
        layer.forward_norm(torch.zeros((1, 1, k), dtype=torch.float32, device=dev), nw, 1e-5)
    with pytest.raises(TypeError):
        layer.forward_norm(torch.zeros((1, 1, k - 128), dtype=torch.bfloat16, device=dev), nw, 1e-5)
    with pytest.raises(TypeError):
        layer.forward_norm(torch.zeros((1, 1, k), dtype=torch.bfloat16, device=dev), nw[:-1], 1e-5)
    for m in (1, 5, 200):                      # 200 rows of a 256-feature layer: the plan's split-K path
        x = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
        y2 = layer.forward_norm(x, nw, 1e-5)                                # [tokens, K] -> [tokens, N]
        y3 = layer.forward_norm(x.reshape(1, m, k), nw, 1e-5)
        tup = (*mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, layer.reorder_index, layer.p4_num, layer.p6_num, layer.p8_num), 1, m)
        assert y2.shape == (m, 256) and torch.equal(y2, y3.reshape(m, 256)) and torch.equal(y3, layer(tup))


def test_supported_queries_take_the_weight_mode():
    """the ring / reduction tail of fp4 weights is 48 KB, of matching-precision weights 64 KB.  The _w queries budget the mode
    the launch will run in; the five-argument forms stay conservative (what they accept launches in either mode)."""
    from micromix_amd import _lib
    lib = _lib.load()
    for name in ("mm_qlinear_decode_supported", "mm_rmsnorm_qlinear_decode_supported", "mm_down_activate_decode_supported"):
        old, new = getattr(lib, name), getattr(lib, name + "_w")
        for m in (1, 2, 4, 8):
            for n, split in ((14336, (2048, 128, 1920)), (28672, (3072, 896, 128)), (4096, (12288, 1024, 1024)), (14336, (14336, 1024, 1024)),
                             (14336, (28672, 2048, 2048))):
                c, w, f = old(m, n, *split), new(m, n, *split, _lib.MM_W_MATCH), new(m, n, *split, _lib.MM_W_FP4)
                assert c == w                          # the old form IS the matching-precision answer for a three-segment split
                assert (f != 0) >= (w != 0)            # fp4 weights never fit less
        assert new(1, 4096, 4096, 0, 0, 7) == 0       # not a weight mode
    # a long-K shape only the fp4 budget accepts (K = 16384 at M = 4: ADVICE's example)
    assert lib.mm_down_activate_decode_supported_w(4, 4096, 14336, 1024, 1024, _lib.MM_W_FP4) != 0
