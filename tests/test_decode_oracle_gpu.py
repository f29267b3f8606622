"""The decode entry points at the Llama-3-8B shapes, held to the ORACLE directly: `mm_qlinear_decode`
(quantize + GEMM in one launch) and `mm_down_activate_decode` (silu(gate) * up + its quantization + down_proj in one launch) at
M = 1 and 4.  tests/test_decode_gpu.py and tests/test_gate_up_gpu.py compare them with the library's own multi-launch forms; here
nothing of the library sits on the `want` side: activations quantized by the oracle (reorder.cu:94-269), weights packed by the oracle
(and the library's packed bytes asserted equal), the three-segment product with the reference's rounding chain by the oracle
(gemm.cu:26-78), tolerance of tests/gemm_check.py."""
import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits, u8
from gemm_check import check_gemm
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu

H, I = 4096, 14336
IN_SPLIT, DOWN_SPLIT = (2048, 128, 1920), (12288, 1024, 1024)


@pytest.mark.parametrize("n", (4096, 14336), ids=("q_o", "gate_up"))
def test_qlinear_decode_llama_shapes_against_the_oracle(dev, n):
    import torch
    rng = np.random.default_rng(n)
    wb = make_inputs(rng, n, H, "weight")
    idx = rng.permutation(H).astype(np.int16)
    tidx = torch.from_numpy(idx).to(dev)
    qw = o.reorder_quantize(wb, idx, *IN_SPLIT, "w4")
    b = mixedgemm.reorder_quantize_w4(t_from_bits(wb, dev), tidx, *IN_SPLIT)
    for s in range(3):
        assert np.array_equal(u8(b[s]), qw[s]), f"packed weight segment {s}"
    wdeq = o.dequant_operand(qw, "w", "w4")
    for m in (1, 4):
        assert mixedgemm.qlinear_decode_supported(m, n, *IN_SPLIT) >= 1
        xb = make_inputs(rng, m, H)
        got = mixedgemm.qlinear_decode(t_from_bits(xb, dev), tidx, *b, *IN_SPLIT)
        qx = o.reorder_quantize(xb, idx, *IN_SPLIT, "x")
        check_gemm(bits_from_t(got), qx, qw, "reference", label=f"qlinear_decode M={m} N={n}", strict=True, wdeq=wdeq)


def test_down_activate_decode_llama_shape_against_the_oracle(dev):
    import torch
    rng = np.random.default_rng(7)
    wb = make_inputs(rng, H, I, "weight")
    qw = o.downproj_quantize(wb, *DOWN_SPLIT, True)
    b = mixedgemm.downproj_quantize_w4(t_from_bits(wb, dev), *DOWN_SPLIT)
    for s in range(3):
        assert np.array_equal(u8(b[s]), qw[s]), f"packed down_proj weight segment {s}"
    wdeq = o.dequant_operand(qw, "w", "w4")
    for m in (1, 4):
        assert mixedgemm.down_activate_decode_supported(m, H, *DOWN_SPLIT) >= 1
        gb, ub = make_inputs(rng, m, I), make_inputs(rng, m, I)
        gate, up = t_from_bits(gb, dev), t_from_bits(ub, dev)
        gu = torch.stack([gate.reshape(m, I // 128, 128), up.reshape(m, I // 128, 128)], dim=2).reshape(m, 2 * I).contiguous()
        got = bits_from_t(mixedgemm.down_activate_decode(gu, b, *DOWN_SPLIT))
        # (1) the oracle's own activation quantizer (activate.cu:44-202) -> the oracle GEMM.  The device's exp differs from libm's in
        # the last place, which moves < 1e-3 of the codes by one step (tests/test_direct_quantize_gpu.py): a moved code changes one of
        # 14336 terms of an output by one quantization step, far inside the GEMM tolerance
        qh = o.activate_quantize(gb, ub, *DOWN_SPLIT)
        check_gemm(got, qh, qw, "reference", label=f"down_activate_decode M={m} (oracle activation)", strict=False, wdeq=wdeq)
        # (2) the same product on the codes the stand-alone quantizer produced (asserted within its budget of the oracle's): strict
        qg = mixedgemm.activate_quantize_x(gate, up, *DOWN_SPLIT)
        hq = [u8(t) for t in qg]
        for s in range(3):
            assert (hq[s] != qh[s]).mean() < 1e-3, f"activation codes, segment {s}"
        check_gemm(got, hq, qw, "reference", label=f"down_activate_decode M={m}", strict=True, wdeq=wdeq)
