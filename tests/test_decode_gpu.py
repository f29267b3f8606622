"""GPU parity tests of the fused decode kernel (quantize-x + GEMM in one launch, M <= 8): bit-identical to the two-op path
(mm_reorder_quantize + mm_matmul), which is itself checked against the oracle elsewhere; plus a direct oracle check."""
import numpy as np
import pytest

from conftest import bits_from_t, make_inputs, t_from_bits
from gemm_check import check_gemm
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu

CASES = [(1, 128, 128, (128, 0, 0)), (1, 256, 4096, (0, 0, 4096)), (3, 200, 1024, (512, 128, 384)), (8, 384, 2048, (1024, 512, 512)),
         (2, 160, 5120, (4096, 512, 512)), (8, 128, 14336, (7168, 512, 6656)), (5, 96, 384, (0, 384, 0)), (7, 1024, 512, (256, 0, 256)),
         (2, 4128, 256, (128, 0, 128)), (8, 4200, 384, (128, 128, 128)),   # N > 4096: the 32-feature kernel (fewer than 16-feature workgroups per CU)
         (3, 8230, 256, (128, 0, 128)), (1, 16400, 384, (128, 128, 128)),  # more feature blocks than CUs (ragged last block); M <= 4: the streaming kernel
         # wide layers at M <= 4 run on mx_gemm_stream.hip with the quantization inside every workgroup: whole rounds of its ring, every
         # format, the early-request path (all rows staged at once) and the hook path (M * K / 32 > 512 groups)
         (2, 8200, 4096, (2048, 128, 1920)), (4, 8192, 1024, (512, 128, 384)), (1, 8448, 1152, (0, 0, 1152)), (4, 8320, 5120, (4096, 512, 512)),
         (3, 8192, 2304, (0, 2304, 0))]


@pytest.mark.parametrize("wmode", ("w4", "w"))
@pytest.mark.parametrize("m,n,k,split", CASES)
def test_fused_decode_equals_two_op_path(dev, wmode, m, n, k, split):
    import torch
    rng = np.random.default_rng(m * 13 + n + k)
    xb = make_inputs(rng, m, k)
    xb[0, :32] = 0                                           # an empty block somewhere after the reorder
    wb = make_inputs(rng, n, k, "weight")
    idx = rng.permutation(k).astype(np.int16)
    bias = t_from_bits(o.f32_to_bf16(rng.standard_normal(n).astype(np.float32)), dev)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    assert mixedgemm.qlinear_decode_supported(m, n, *split) >= 1
    b = (mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w)(w, tidx, *split)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    for rounding in ("reference", "fused"):
        for bv in (None, bias):
            want = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], bias=bv, rounding=rounding)
            got = mixedgemm.qlinear_decode(x, tidx, *b, *split, bias=bv, rounding=rounding)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (m, n, k, split, wmode, rounding, bv is not None)
    # and against the oracle directly
    qx = o.reorder_quantize(xb, idx, *split, "x")
    qw = o.reorder_quantize(wb, idx, *split, wmode)
    got = mixedgemm.qlinear_decode(x, tidx, *b, *split)
    check_gemm(bits_from_t(got), qx, qw, "reference", label=f"decode {m}x{n}x{k} {split} {wmode}")


def test_supported_range_and_layer_dispatch(dev):
    import torch
    from micromix_amd.qlinear import QLinearLayer
    assert mixedgemm.qlinear_decode_supported(9, 4096, 0, 0, 4096) == 0
    assert mixedgemm.qlinear_decode_supported(0, 4096, 0, 0, 4096) == 0
    assert mixedgemm.qlinear_decode_supported(8, 4096, 0, 0, 4096) == 2     # q/o: one round of workgroups, two passes
    assert mixedgemm.qlinear_decode_supported(8, 14336, 0, 0, 4096) == 1    # gate/up at M = 8: runs, but the two-op path is faster
    assert mixedgemm.qlinear_decode_supported(8, 4096, 0, 0, 16384) == 0    # 8 fp8 rows of 16 KiB do not fit in LDS
    g = torch.Generator().manual_seed(2)
    k, split = 1024, (512, 128, 384)
    idx = torch.randperm(k, generator=g)
    lin = torch.nn.Linear(k, 256, bias=True, dtype=torch.bfloat16)
    layer = QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx)
    x = torch.randn((2, 3, k), generator=g).to(torch.bfloat16).to(dev)        # M = 6: fused kernel
    y = layer(x)
    ref = layer(layer.quantize_input(x))                                     # tuple input: the two-op path
    assert y.shape == (2, 3, 256) and torch.equal(y, ref)
    with pytest.raises(RuntimeError):
        mixedgemm.qlinear_decode(torch.zeros((9, k), dtype=torch.bfloat16, device=dev), layer.reorder_index, layer.BN, layer.BS,
                                 layer.BO, layer.SFBN, layer.SFBS, layer.SFBO, *split)
