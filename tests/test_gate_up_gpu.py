"""mm_gate_up_activate (gate_proj + up_proj + silu(gate) * up + the MX quantization for down_proj as one launch; the reference:
model/qLlamaLayer.py:377-387, mgemm/src/activate.cu:44-202, bindings.cpp:307-334):
  * byte for byte against this library's own three-op form matmul(gate), matmul(up) -> activate_quantize_x on the same operands
    (every segment format, both tile kernels, ragged M, the M <= 64 two-launch path, both roundings), at small shapes and at the
    Llama-3-8B / Qwen2.5-14B MLP shapes with M in {16, 256, 4096};
  * against the ORACLE chain (oracle GEMM on the oracle-quantized operands -> oracle activate_quantize) with the stated budget of
    tests/test_direct_quantize_gpu.py for the silu (device exp) -- on rows whose gate / up values the GPU GEMM reproduces bit for bit;
  * FusedMLP / TPMLP route through it."""
import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits, u8
from gemm_check import check_gemm
from micromix_amd import _lib, mixedgemm, tp
from micromix_amd.qlinear import FusedMLP, QLinearLayer
from model_case import gen_bf16, gen_index, tile_positions
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu


def _mm(a, b, **kw):
    return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)


def assert_same_operands(got, want, rows, split, label):
    """two 6-tuples of quantizer outputs: packed bytes equal everywhere, scale bytes equal wherever a real row owns them"""
    for i in range(3):
        assert got[i].shape == want[i].shape, (label, i, got[i].shape, want[i].shape)
        assert np.array_equal(u8(got[i]), u8(want[i])), f"{label}: packed segment {i} differs ({int((u8(got[i]) != u8(want[i])).sum())} bytes)"
        if split[i]:
            offs = o.sf_valid_offsets(rows, split[i])
            assert got[3 + i].numel() == want[3 + i].numel()
            assert np.array_equal(u8(got[3 + i])[offs], u8(want[3 + i])[offs]), f"{label}: scale bytes of segment {i} differ"


def three_op(qx, qg, qu, dsplit, rounding="reference"):
    return mixedgemm.activate_quantize_x(_mm(qx, qg, rounding=rounding), _mm(qx, qu, rounding=rounding), *dsplit)


# (M, H, I, in_split, down_split): every consumer format in the first and the last tile, both tile kernels (128- and 256-row),
# ragged M (partial tiles, a second SF atom without rows), the M <= 64 path
SMALL = [
    (200, 512, 384, (256, 128, 128), (128, 128, 128)),
    (256, 384, 768, (128, 128, 128), (512, 128, 128)),
    (300, 256, 512, (0, 0, 256), (0, 512, 0)),
    (130, 256, 256, (256, 0, 0), (0, 0, 256)),
    (65, 640, 1280, (256, 128, 256), (1024, 128, 128)),
    (3000, 256, 2304, (128, 0, 128), (1024, 768, 512)),      # > 256 tiles of 256 rows: the 256-row kernel, several rounds
    (40, 384, 512, (128, 128, 128), (256, 128, 128)),        # M <= 64: weight-streaming GEMM + the quantizer on the interleaved scratch
    (1, 256, 256, (0, 0, 256), (128, 0, 128)),
]


@pytest.mark.parametrize("m,h,i,in_split,dsplit", SMALL, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in SMALL])
def test_fused_equals_three_ops(dev, m, h, i, in_split, dsplit):
    import torch
    rng = np.random.default_rng(m + h + i)
    x = t_from_bits(make_inputs(rng, m, h), dev)
    wg = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8          # gate values of a few units: silu away from its linear range
    wu = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8
    idx = torch.from_numpy(rng.permutation(h).astype(np.int16)).to(dev)
    qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
    qg = mixedgemm.reorder_quantize_w4(wg, idx, *in_split)
    qu = mixedgemm.reorder_quantize_w4(wu, idx, *in_split)
    qgu = mixedgemm.interleave_gate_up(qg, qu)
    back = mixedgemm.deinterleave_gate_up(qgu)
    assert all(torch.equal(a, b) for a, b in zip(back[0], qg)) and all(torch.equal(a, b) for a, b in zip(back[1], qu))
    for rounding in ("reference", "fused"):
        want = three_op(qx, qg, qu, dsplit, rounding)
        got = mixedgemm.gate_up_activate(qx, qgu, *dsplit, rounding=rounding)
        assert_same_operands(got, want, m, dsplit, f"{m}x{h}x{i} {rounding}")
        again = mixedgemm.gate_up_activate(qx, qgu, *dsplit, rounding=rounding)
        assert all(torch.equal(a, b) for a, b in zip(got[:3], again[:3]))                 # deterministic
    desc = _lib.load().mm_gate_up_activate_describe(m, i).decode()
    assert ("act_kernel" in desc) == (m > 64), desc


# the decode form (bf16 activations in, two launches): the same bytes as reorder_quantize_x -> gate_up_activate.  Small shapes (first
# fused decode kernel), a wide layer at M <= 4 (the streaming kernel with the quantization inside every workgroup) and Llama's own MLP
DECODE = [(1, 256, 256, (0, 0, 256), (128, 0, 128)), (3, 384, 512, (128, 128, 128), (256, 128, 128)), (8, 640, 1280, (256, 128, 256), (1024, 128, 128)),
          (2, 1024, 4224, (512, 128, 384), (3072, 512, 640)), (1, 4096, 14336, (2048, 128, 1920), (12288, 1024, 1024))]


@pytest.mark.parametrize("m,h,i,in_split,dsplit", DECODE, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in DECODE])
def test_decode_form_equals_quantize_then_fused(dev, m, h, i, in_split, dsplit):
    import torch
    rng = np.random.default_rng(m * 7 + h + i)
    x = t_from_bits(make_inputs(rng, m, h), dev)
    wg = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8
    wu = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8
    idx = torch.from_numpy(rng.permutation(h).astype(np.int16)).to(dev)
    qgu = mixedgemm.interleave_gate_up(mixedgemm.reorder_quantize_w4(wg, idx, *in_split), mixedgemm.reorder_quantize_w4(wu, idx, *in_split))
    assert mixedgemm.qlinear_decode_supported(m, 2 * i, *in_split) >= 1
    for rounding in ("reference", "fused"):
        want = mixedgemm.gate_up_activate(mixedgemm.reorder_quantize_x(x, idx, *in_split), qgu, *dsplit, rounding=rounding)
        got = mixedgemm.gate_up_activate_decode(x, idx, qgu, *dsplit, rounding=rounding)
        assert_same_operands(got, want, m, dsplit, f"decode {m}x{h}x{i} {rounding}")
    with pytest.raises(RuntimeError):
        mixedgemm.gate_up_activate_decode(torch.cat([x] * 9)[:9].contiguous(), idx, qgu, *dsplit)      # M = 9: not a decode batch


# Round 6: on wide layers (2 I / 64 >= the CUs) the decode-sized forms run as ONE weight-streaming launch with the activation inside
# (mx_gemm_stream.hip, ACT).  Held byte for byte to the three-op twin -- matmul(gate), matmul(up) -> activate_quantize_x, which shares
# no code with that epilogue -- from the quantized activations (M <= 16), from the bf16 rows (M <= 4) and with the RMSNorm in front.
ONE_LAUNCH = [(1, 1024, 8192, (512, 128, 384), (4096, 2048, 2048)), (3, 1024, 8192, (0, 0, 1024), (0, 8192, 0)), (4, 512, 8448, (512, 0, 0), (8448, 0, 0)),
              (1, 4096, 14336, (2048, 128, 1920), (12288, 1024, 1024)), (2, 4096, 14336, (2048, 128, 1920), (7168, 512, 6656))]


@pytest.mark.parametrize("m,h,i,in_split,dsplit", ONE_LAUNCH, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in ONE_LAUNCH])
def test_one_launch_forms_equal_three_ops(dev, m, h, i, in_split, dsplit):
    import torch
    rng = np.random.default_rng(m * 13 + h + i)
    x = t_from_bits(make_inputs(rng, m, h), dev)
    wg = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8
    wu = t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8
    nw = t_from_bits(o.f32_to_bf16((1.0 + 0.2 * rng.standard_normal(h)).astype(np.float32)), dev)
    idx = torch.from_numpy(rng.permutation(h).astype(np.int16)).to(dev)
    qg = mixedgemm.reorder_quantize_w4(wg, idx, *in_split)
    qu = mixedgemm.reorder_quantize_w4(wu, idx, *in_split)
    qgu = mixedgemm.interleave_gate_up(qg, qu)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    wide = 2 * i // 64 >= cus
    assert mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, i, *in_split) == (2 if wide and m <= 2 else 1)
    assert mixedgemm.gate_up_activate_decode_supported(m, i, *in_split) == (2 if wide and m <= 2 else 1)
    qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
    qn = mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, idx, *in_split)
    for rounding in ("reference", "fused"):
        want = three_op(qx, qg, qu, dsplit, rounding)
        assert_same_operands(mixedgemm.gate_up_activate(qx, qgu, *dsplit, rounding=rounding), want, m, dsplit, f"quantized x {rounding}")
        assert_same_operands(mixedgemm.gate_up_activate_decode(x, idx, qgu, *dsplit, rounding=rounding), want, m, dsplit, f"bf16 x {rounding}")
        want_n = three_op(qn, qg, qu, dsplit, rounding)
        got_n = mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, qgu, *dsplit, rounding=rounding)
        assert_same_operands(got_n, want_n, m, dsplit, f"norm {rounding}")
    # M = 16 (the widest single-tile batch) and two token tiles (17 .. 32 tokens) from the quantized activations
    for mm_ in (16, 17, 27, 32):
        xq = t_from_bits(make_inputs(rng, mm_, h), dev)
        qq = mixedgemm.reorder_quantize_x(xq, idx, *in_split)
        for rounding in ("reference", "fused"):
            assert_same_operands(mixedgemm.gate_up_activate(qq, qgu, *dsplit, rounding=rounding), three_op(qq, qg, qu, dsplit, rounding), mm_, dsplit,
                                 f"quantized x M={mm_} {rounding}")
        if wide:
            assert "stream_act" in _lib.load().mm_gate_up_activate_describe(mm_, i).decode()
    # and the MLP's second half on those operands is a plain matmul
    wd = t_from_bits(make_inputs(rng, 256, i, "weight"), dev)
    b = mixedgemm.downproj_quantize_w4(wd, *dsplit)
    assert torch.equal(_mm(got_n, b), _mm(want_n, b))


# down_proj straight from the bf16 gate | up matrix (mm_down_activate_decode, M <= 4): bit-identical to activate_quantize_x -> matmul;
# every consumer format, N with a ragged last workgroup, two passes of the in-workgroup quantizer (M * I / 32 > 512), bias, both weight
# modes, both roundings; Llama's own shapes
DOWN_DECODE = [(1, 256, 256, (128, 0, 128)), (3, 200, 512, (256, 128, 128)), (4, 1000, 1280, (1024, 128, 128)), (2, 8200, 768, (0, 768, 0)),
               (1, 4096, 14336, (12288, 1024, 1024)), (2, 4096, 14336, (12288, 1024, 1024))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,i,dsplit", DOWN_DECODE, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in DOWN_DECODE])
def test_down_from_gate_up_matrix(dev, m, n, i, dsplit, wmode):
    import torch
    rng = np.random.default_rng(m * 11 + n + i)
    gate = t_from_bits(make_inputs(rng, m, i), dev) * 2
    up = t_from_bits(make_inputs(rng, m, i), dev)
    wd = t_from_bits(make_inputs(rng, n, i, "weight"), dev)
    bias = t_from_bits(o.f32_to_bf16(rng.standard_normal(n).astype(np.float32)), dev)
    gu = torch.stack([gate.reshape(m, i // 128, 128), up.reshape(m, i // 128, 128)], dim=2).reshape(m, 2 * i).contiguous()   # 128 gate | 128 up
    b = (mixedgemm.downproj_quantize_w4 if wmode == "w4" else mixedgemm.downproj_quantize_w)(wd, *dsplit)
    assert mixedgemm.down_activate_decode_supported(m, n, *dsplit) >= 1
    qh = mixedgemm.activate_quantize_x(gate, up, *dsplit)
    for rounding in ("reference", "fused"):
        for bv in (None, bias):
            want = mixedgemm.matmul(qh[0], b[0], qh[1], b[1], qh[2], b[2], qh[3], b[3], qh[4], b[4], qh[5], b[5], bias=bv, rounding=rounding)
            got = mixedgemm.down_activate_decode(gu, b, *dsplit, bias=bv, rounding=rounding)
            assert torch.equal(got, want), (m, n, i, dsplit, wmode, rounding, bv is not None)
    with pytest.raises(RuntimeError):
        mixedgemm.down_activate_decode(torch.cat([gu] * 5)[:5].contiguous(), b, *dsplit)      # M = 5


MODELS = [("llama3-8b", 4096, 14336, (2048, 128, 1920), (12288, 1024, 1024)), ("llama3-8b-fp8x", 4096, 14336, (0, 0, 4096), (7168, 512, 6656)),
          ("qwen2.5-14b", 5120, 13824, (3072, 1024, 1024), (11776, 1024, 1024))]


@pytest.mark.parametrize("name,h,i,in_split,dsplit", MODELS, ids=[c[0] for c in MODELS])
def test_model_mlp_shapes(dev, name, h, i, in_split, dsplit):
    """Llama-3-8B and Qwen2.5-14B gate/up at M in {16, 256, 4096}: fused == three ops, byte for byte over the whole [M, I] output;
    and sampled rows (every tile position at M = 4096) against the oracle chain"""
    import torch
    rng = np.random.default_rng(h + i)
    wg, wu = gen_bf16(dev, i, h, 11, "w") * 4, gen_bf16(dev, i, h, 12, "w") * 4
    idx = gen_index(dev, h, 13)
    qg = mixedgemm.reorder_quantize_w4(wg, idx, *in_split)
    qu = mixedgemm.reorder_quantize_w4(wu, idx, *in_split)
    qgu = mixedgemm.interleave_gate_up(qg, qu)
    hg, hu = [u8(t) for t in qg], [u8(t) for t in qu]
    dg, du = o.dequant_operand(hg, "w", "w4"), o.dequant_operand(hu, "w", "w4")
    for m in (16, 256, 4096):
        x = gen_bf16(dev, m, h, seed=m + 5)
        qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
        want = three_op(qx, qg, qu, dsplit)
        got = mixedgemm.gate_up_activate(qx, qgu, *dsplit)
        assert_same_operands(got, want, m, dsplit, f"{name} M={m}")
        # oracle chain on sampled rows
        rows = tile_positions(rng, m)[::4] if m > 64 else np.arange(m)
        ridx = torch.from_numpy(rows).to(dev)
        ref_x = o.reorder_quantize(bits_from_t(x[ridx]), u8(idx), *in_split, "x")
        # the gate / up values inside the fused launch are those of mm_matmul (asserted above through the three-op twin): hold them to
        # the oracle GEMM here, then the fused launch's codes to the oracle's activation quantizer ON those values (silu budget)
        g_gpu, u_gpu = bits_from_t(_mm(qx, qg)[ridx]), bits_from_t(_mm(qx, qu)[ridx])
        check_gemm(g_gpu, ref_x, hg, "reference", label=f"{name} gate M={m}", strict=len(rows) * i >= 4096, wdeq=dg)
        check_gemm(u_gpu, ref_x, hu, "reference", label=f"{name} up M={m}", strict=len(rows) * i >= 4096, wdeq=du)
        ref_q = o.activate_quantize(g_gpu, u_gpu, *dsplit)
        for s in range(3):
            if not dsplit[s]:
                continue
            gq = u8(got[s][ridx])
            sf = u8(got[3 + s])
            j = np.arange(dsplit[s] // 32)[None, :]
            gsf = sf[o.sf_offset(rows[:, None], j, dsplit[s])]
            wsf = ref_q[3 + s][o.sf_offset(np.arange(len(rows))[:, None], j, dsplit[s])]
            assert (gsf != wsf).mean() < 1e-3, f"{name} M={m} segment {s}: scale bytes"
            assert (gq != ref_q[s]).mean() < 1e-3, f"{name} M={m} segment {s}: packed bytes"
    del wg, wu


def test_fused_mlp_module_and_tpmlp(dev):
    """FusedMLP == gate(x), up(x) -> activate_quantize_x -> matmul(down), bit for bit; TPMLP (world = 1 and every rank of world = 2)
    runs the fused kernel and keeps its results"""
    import torch
    rng = np.random.default_rng(3)
    m, hid, inter = 192, 512, 2048
    in_split, down_split = (256, 128, 128), (1024, 512, 512)
    x = t_from_bits(make_inputs(rng, m, hid), dev)
    lin = lambda n, k: t_from_bits(make_inputs(rng, n, k, "weight"), dev)
    wg, wu, wd = lin(inter, hid), lin(inter, hid), lin(hid, inter)
    idx = torch.from_numpy(rng.permutation(hid).astype(np.int16)).to(dev)
    with torch.no_grad():
        lg = torch.nn.Linear(hid, inter, bias=False, dtype=torch.bfloat16, device=dev); lg.weight.copy_(wg)
        lu = torch.nn.Linear(hid, inter, bias=False, dtype=torch.bfloat16, device=dev); lu.weight.copy_(wu)
    gate, up = QLinearLayer(lg, in_split[2], in_split[1], idx), QLinearLayer(lu, in_split[2], in_split[1], idx)
    mlp = FusedMLP(gate, up, wd, down_split)
    y = mlp(x.reshape(2, m // 2, hid))
    qh = mixedgemm.activate_quantize_x(gate(x.reshape(1, m, hid))[0], up(x.reshape(1, m, hid))[0], *down_split)
    qd = mixedgemm.downproj_quantize_w4(wd, *down_split)
    want = _mm(qh, qd)
    assert y.shape == (2, m // 2, hid) and torch.equal(y.reshape(m, hid), want)
    t1 = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=0, world=1)
    assert t1.fused and torch.equal(t1.partial(t1.quantize_x(x)), _mm(qh, qd, rounding="fused"))
    total = torch.zeros((m, hid), dtype=torch.float32, device=dev)
    for r in range(2):
        tr = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=r, world=2)
        total += tr.partial(tr.quantize_x(x), fp32=True)
    full = _mm(qh, qd, rounding="fused").float()
    assert float((total - full).abs().max()) <= 2.0 ** -8 * float(full.abs().max()) + 1e-3


def test_errors(dev):
    import torch
    x = torch.zeros((128, 256), dtype=torch.bfloat16, device=dev)
    idx = torch.arange(256, dtype=torch.int16, device=dev)
    qx = mixedgemm.reorder_quantize_x(x, idx, 128, 0, 128)
    w = torch.zeros((256, 256), dtype=torch.bfloat16, device=dev)
    q4 = mixedgemm.reorder_quantize_w4(w, idx, 128, 0, 128)
    qgu = mixedgemm.interleave_gate_up(q4, q4)
    with pytest.raises(RuntimeError, match="Value error in run_activate_quantize_x"):
        mixedgemm.gate_up_activate(qx, qgu, 128, 0, 0)                 # does not sum to I
    with pytest.raises(RuntimeError, match="interleaved fp4 gate/up"):
        mixedgemm.gate_up_activate(qx, mixedgemm.interleave_gate_up(*[mixedgemm.reorder_quantize_w(w, idx, 128, 0, 128)] * 2), 128, 0, 128)
    out = mixedgemm.gate_up_activate(qx, qgu, 256, 0, 0)
    assert out[0].shape == (128, 128) and out[2].shape == (128, 0)
    assert int(u8(out[0]).max()) == 0                                   # silu(0) * 0 = 0 -> fp4 code 0
    assert int(u8(out[3])[o.sf_valid_offsets(128, 256)].min()) == 127  # empty block: scale 1.0 (activate.cu:117-120), not 0.5


def test_norm_decode_form_errors_and_fallback(dev):
    """mm_rmsnorm_gate_up_activate_decode: the C ABI's argument checks, the two-launch fallback on a narrow layer (the same bytes), and the
    queries' answers past the decode sizes"""
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(5)
    m, h, i, in_split, dsplit = 2, 384, 512, (128, 128, 128), (256, 128, 128)          # 2 I / 64 = 16 workgroups: not a wide layer
    x = t_from_bits(make_inputs(rng, m, h), dev)
    nw = t_from_bits(o.f32_to_bf16((1.0 + 0.2 * rng.standard_normal(h)).astype(np.float32)), dev)
    idx = torch.from_numpy(rng.permutation(h).astype(np.int16)).to(dev)
    qg = mixedgemm.reorder_quantize_w4(t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8, idx, *in_split)
    qu = mixedgemm.reorder_quantize_w4(t_from_bits(make_inputs(rng, i, h, "weight"), dev) * 8, idx, *in_split)
    qgu = mixedgemm.interleave_gate_up(qg, qu)
    assert mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, i, *in_split) == 1
    want = three_op(mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, idx, *in_split), qg, qu, dsplit)
    assert_same_operands(mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, qgu, *dsplit), want, m, dsplit, "narrow layer, two launches")
    want_nr = three_op(mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, idx, *in_split, integer_round=False), qg, qu, dsplit)
    got_nr = mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, qgu, *dsplit, integer_round=False)
    assert_same_operands(got_nr, want_nr, m, dsplit, "no integer round")
    # queries: nothing past the decode sizes, nothing for a bad split
    assert mixedgemm.rmsnorm_gate_up_activate_decode_supported(9, 14336, 2048, 128, 1920) == 0
    assert mixedgemm.gate_up_activate_decode_supported(9, 14336, 2048, 128, 1920) == 0
    assert lib.mm_rmsnorm_gate_up_activate_decode_supported(1, 14336, 2048, 100, 1948) == 0
    assert lib.mm_rmsnorm_gate_up_activate_decode_supported(1, 14300, 2048, 128, 1920) == 0
    with pytest.raises(RuntimeError, match="Value error in run_activate_quantize_x"):
        mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, qgu, 256, 128, 0)              # does not sum to I
    with pytest.raises(RuntimeError):
        mixedgemm.rmsnorm_gate_up_activate_decode(torch.cat([x] * 5)[:9].contiguous(), nw, 1e-5, idx, qgu, *dsplit)      # M = 9
    # C ABI: null pointers, misaligned rows, unknown flag bits
    p = lambda t: t.data_ptr() if t.numel() else None
    outs = [torch.empty_like(t) for t in want]
    args = lambda xp, wp, flags: lib.mm_rmsnorm_gate_up_activate_decode(xp, wp, 1e-5, p(idx), *[p(t) for t in qgu], m, i, *in_split, *dsplit, flags,
                                                                      *[p(t) for t in outs], None, 0, None)
    assert args(None, p(nw), 0) != 0 and args(p(x) + 2, p(nw), 0) != 0 and args(p(x), None, 0) != 0 and args(p(x), p(nw), 0x40) != 0
    assert args(p(x), p(nw), 0) != 0          # the narrow layer needs the workspace of the two-launch form
    assert lib.mm_rmsnorm_gate_up_activate_decode(p(x), p(nw), 1e-5, p(idx), *[p(t) for t in qgu], 0, i, *in_split, *dsplit, 0, *[p(t) for t in outs],
                                                  None, 0, None) == 0      # M = 0: nothing to do
