"""GPU parity tests of rmsnorm_quantize_x against the oracle: byte-for-byte.

Everything in the op is exactly specified fp32 / integer arithmetic -- the kernel follows the oracle's summation order for the
sum of squares and uses the correctly rounded divide and square root -- so packed bytes and scale bytes must be identical."""
import numpy as np
import pytest

from conftest import make_inputs, t_from_bits, u8
from micromix_amd import mixedgemm
from micromix_amd.qlinear import QLinearLayer
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


CASES = [(1, 128, (0, 128, 0)), (3, 384, (128, 128, 128)), (130, 4096, (2048, 1024, 1024)), (64, 4096, (0, 0, 4096)),
         (17, 5120, (4096, 512, 512)), (9, 3072, (1024, 1024, 1024)), (5, 3584, (3584, 0, 0)), (2, 14336, (7168, 512, 6656)),
         (4, 8192, (4096, 2048, 2048)),       # the largest K of the 32-bit product row (256 threads, four partial sums per lane)
         (3, 8320, (4096, 128, 4096)),        # the first K of the 16-bit row kernel
         (3, 20480, (8192, 4096, 8192)),      # K > 16384: the 1024-thread variant
         (3, 32768, (16384, 8192, 8192))]     # the largest K the int16 reorder index allows: 68 KiB of dynamic LDS


@pytest.mark.parametrize("integer_round", (True, False))
@pytest.mark.parametrize("rows,k,split", CASES)
def test_rmsnorm_quantize_matches_oracle(dev, rows, k, split, integer_round):
    import torch
    rng = np.random.default_rng(rows * 3 + k)
    xb = make_inputs(rng, rows, k)
    xb[0, :64] = 0                                            # an all-zero group after the reorder is unlikely: force one below
    wb = o.f32_to_bf16((1.0 + 0.2 * rng.standard_normal(k)).astype(np.float32))
    idx = rng.permutation(k).astype(np.int16)
    if rows > 1:
        xb[1, :] = 0                                          # zero row: rvar = 1/sqrt(eps), every block is empty -> byte 126
    eps = 1e-5
    got = mixedgemm.rmsnorm_quantize_x(t_from_bits(xb, dev), t_from_bits(wb, dev), eps,
                                       torch.from_numpy(idx).to(dev), *split, integer_round=integer_round)
    torch.cuda.synchronize()
    want = o.rmsnorm_quantize(xb, wb, eps, idx, *split, integer_round=integer_round)
    check_exact(got, want, rows, split, f"rmsnorm {rows}x{k} {split} integer_round={integer_round}")


def test_without_integer_round_equals_norm_then_reorder_quantize(dev):
    """integer_round=False is RMSNorm (oracle) followed by the path's own reorder_quantize_x kernel."""
    import torch
    rng = np.random.default_rng(8)
    rows, k, split = 40, 4096, (2048, 128, 1920)
    xb = make_inputs(rng, rows, k)
    wb = o.f32_to_bf16((1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32))
    idx = rng.permutation(k).astype(np.int16)
    rvar = o.rmsnorm_rvar(xb, 1e-6)
    normed = o.f32_to_bf16(((o.bf16_to_f32(xb) * o.bf16_to_f32(wb)[None, :]).astype(np.float32) * rvar[:, None]).astype(np.float32))
    tidx = torch.from_numpy(idx).to(dev)
    a = mixedgemm.rmsnorm_quantize_x(t_from_bits(xb, dev), t_from_bits(wb, dev), 1e-6, tidx, *split, integer_round=False)
    b = mixedgemm.reorder_quantize_x(t_from_bits(normed, dev), tidx, *split)
    torch.cuda.synchronize()
    check_exact(a, [u8(t) for t in b], rows, split, "norm+quantize")


def test_shared_quantization_feeds_several_layers(dev):
    """the tuple returned by rmsnorm_quantize_x is what QLinearLayer.forward accepts: q/k/v share one quantization."""
    import torch
    g = torch.Generator().manual_seed(11)
    k, split = 1024, (512, 128, 384)
    idx = torch.randperm(k, generator=g)
    x = torch.randn((1, 24, k), generator=g).to(torch.bfloat16).to(dev)
    nw = (1 + 0.1 * torch.randn(k, generator=g)).to(torch.bfloat16).to(dev)
    layers = [QLinearLayer(torch.nn.Linear(k, n, bias=False, dtype=torch.bfloat16).to(dev), p8_num=split[2], p6_num=split[1],
                           reorder_index=idx) for n in (256, 128)]
    q = mixedgemm.rmsnorm_quantize_x(x.reshape(-1, k), nw, 1e-5, idx.to(torch.int16).to(dev), *split, integer_round=False)
    xf = x.float()
    normed = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.float()).to(torch.bfloat16)
    for layer in layers:
        y = layer((*q, 1, 24))
        ref = layer(normed)
        assert y.shape == ref.shape
        err = (y.float() - ref.float()).abs().max().item()
        assert err <= 0.05 * ref.float().abs().max().item() + 1e-3   # torch's rsqrt/mean order differs in the last bf16 ulp


def test_errors(dev):
    import torch
    x = torch.zeros((4, 256), dtype=torch.bfloat16, device=dev)
    w = torch.ones((256,), dtype=torch.bfloat16, device=dev)
    idx = torch.arange(256, dtype=torch.int16, device=dev)
    with pytest.raises(RuntimeError, match="Value error in run_rmsnorm_bf16_mixed"):
        mixedgemm.rmsnorm_quantize_x(x, w, 1e-5, idx, 128, 64, 64)
    with pytest.raises(TypeError):
        mixedgemm.rmsnorm_quantize_x(x.float(), w, 1e-5, idx, 128, 128, 0)
