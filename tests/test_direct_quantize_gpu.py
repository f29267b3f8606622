"""GPU parity tests of the reorder-free quantizers against the oracle.

downproj_quantize_w / _w4 are exact integer work on bf16 inputs: byte-for-byte.  activate_quantize_x evaluates
silu(a) * b in fp32 with the device's expf; numpy's exp may differ from it in the last ulp, which can move a value across a
rounding boundary (or, rarely, a block maximum across a power of two), so that op is compared within a stated budget:
identical shapes, >= 99.9 % of packed bytes and of scale bytes identical, and dequantised values within one quantisation
step of the oracle's."""
import numpy as np
import pytest

from conftest import make_inputs, t_from_bits, u8
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu


def check_exact(got, want, rows, split, label):
    for i, (g, w) in enumerate(zip(got, want)):
        g = u8(g)
        assert g.shape == w.shape, (label, i, g.shape, w.shape)
        if i < 3:
            assert np.array_equal(g, w), f"{label}: packed segment {i} differs ({(g != w).sum()} bytes)"
        else:
            offs = o.sf_valid_offsets(rows, split[i - 3])
            assert np.array_equal(g[offs], w[offs]), f"{label}: scale bytes of segment {i - 3} differ"


@pytest.mark.parametrize("rows,k,split", [(1, 128, (0, 128, 0)), (5, 384, (128, 128, 128)), (130, 4096, (2048, 1024, 1024)),
                                           (64, 14336, (7168, 512, 6656)), (257, 1024, (0, 0, 1024)), (33, 1024, (1024, 0, 0))])
@pytest.mark.parametrize("w4", (False, True))
def test_downproj_quantize_matches_oracle(dev, rows, k, split, w4):
    import torch
    rng = np.random.default_rng(rows + k)
    wb = make_inputs(rng, rows, k, "weight")
    wb[0, :32] = 0                      # empty block -> scale byte 127 (not 126)
    fn = mixedgemm.downproj_quantize_w4 if w4 else mixedgemm.downproj_quantize_w
    got = fn(t_from_bits(wb, dev), *split)
    torch.cuda.synchronize()
    want = o.downproj_quantize(wb, *split, w4=w4)
    check_exact(got, want, rows, split, f"downproj w4={w4} {rows}x{k}")
    first = next(i for i in range(3) if split[i])
    assert u8(got[3 + first])[int(o.sf_offset(0, 0, split[first]))] == 127 if first == 0 else True


def test_downproj_extremes(dev):
    allb = np.arange(65536, dtype=np.uint16)
    fin = allb[np.isfinite(o.bf16_to_f32(allb))]
    xb = np.resize(fin, (64, 1024))
    for w4 in (False, True):
        fn = mixedgemm.downproj_quantize_w4 if w4 else mixedgemm.downproj_quantize_w
        got = fn(t_from_bits(xb, dev), 384, 256, 384)
        check_exact(got, o.downproj_quantize(xb, 384, 256, 384, w4=w4), 64, (384, 256, 384), f"extremes w4={w4}")


@pytest.mark.parametrize("rows,k,split", [(7, 384, (128, 128, 128)), (130, 4096, (2048, 1024, 1024)), (48, 14336, (12288, 1024, 1024))])
def test_activate_quantize_close_to_oracle(dev, rows, k, split):
    import torch
    rng = np.random.default_rng(k + rows)
    ab = o.f32_to_bf16((rng.standard_normal((rows, k)) * 2).astype(np.float32))
    bb = make_inputs(rng, rows, k)
    got = [u8(t) for t in mixedgemm.activate_quantize_x(t_from_bits(ab, dev), t_from_bits(bb, dev), *split)]
    torch.cuda.synchronize()
    want = o.activate_quantize(ab, bb, *split)
    fmts = ("fp4", "fp6", "fp8")
    for i in range(3):
        if not split[i]:
            continue
        assert got[i].shape == want[i].shape and got[3 + i].shape == want[3 + i].shape
        offs = o.sf_valid_offsets(rows, split[i])
        assert (got[3 + i][offs] != want[3 + i][offs]).mean() < 1e-3
        assert (got[i] != want[i]).mean() < 1e-3
        dg = o.dequant_segment(got[i], got[3 + i], rows, split[i], fmts[i])
        dw = o.dequant_segment(want[i], want[3 + i], rows, split[i], fmts[i])
        amax = np.abs(dw).reshape(rows, -1, 32).max(-1, keepdims=True)
        step = amax * 2.0 ** (-o.FORMATS[fmts[i]]["mbits"])
        assert np.all(np.abs(dg - dw).reshape(rows, -1, 32) <= step + 1e-30)


def test_activate_feeds_gemm(dev):
    """the fused activation quantizer produces operands mixedgemm.matmul consumes: compare with the unfused chain
    bf16(silu(a)*b) -> reorder_quantize_x(identity index) at the level of the GEMM result."""
    import torch
    rng = np.random.default_rng(2)
    m, n, k, split = 96, 256, 1024, (512, 128, 384)
    a = t_from_bits(o.f32_to_bf16(rng.standard_normal((m, k)).astype(np.float32)), dev)
    b = t_from_bits(make_inputs(rng, m, k), dev)
    w = t_from_bits(make_inputs(rng, n, k, "weight"), dev)
    qw = mixedgemm.downproj_quantize_w4(w, *split)
    qx = mixedgemm.activate_quantize_x(a, b, *split)
    y_fused = mixedgemm.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5]).float()
    h = (torch.nn.functional.silu(a.float()) * b.float())
    ident = torch.arange(k, dtype=torch.int16, device=dev)
    q2 = mixedgemm.reorder_quantize_x(h.to(torch.bfloat16), ident, *split)
    y_unfused = mixedgemm.matmul(q2[0], qw[0], q2[1], qw[1], q2[2], qw[2], q2[3], qw[3], q2[4], qw[4], q2[5], qw[5]).float()
    ref = h @ w.float().t()
    e_fused = torch.linalg.norm(y_fused - ref) / torch.linalg.norm(ref)
    e_unfused = torch.linalg.norm(y_unfused - ref) / torch.linalg.norm(ref)
    assert float(e_fused) < 0.2 and float(e_fused) <= float(e_unfused) * 1.05   # MX noise level; no bf16 round trip


def test_errors(dev):
    import torch
    x = torch.zeros((4, 256), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="Value error in run_activate_quantize_x"):
        mixedgemm.activate_quantize_x(x, x, 100, 28, 128)
    with pytest.raises(RuntimeError, match="equal shape"):
        mixedgemm.activate_quantize_x(x, x[:2], 128, 0, 128)
    out = mixedgemm.downproj_quantize_w(x, 0, 0, 256)
    assert out[0].shape == (4, 0) and out[2].shape == (4, 256) and out[5].numel() == 128 * 8
