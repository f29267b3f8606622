"""GPU test of the K-sharded tensor-parallel path on ONE device: every rank's shard is executed in turn with
the HIP kernels (gather-subset quantize + fused GEMM) and the partials are summed as the all-reduce would."""
import numpy as np
import pytest

from conftest import make_inputs, t_from_bits
from micromix_amd import mixedgemm, tp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("split", [(0, 0, 4096), (2048, 128, 1920)])
def test_sharded_sum_equals_unsharded(dev, world, split):
    import torch
    rng = np.random.default_rng(world)
    m, n, k = 200, 512, 4096
    x = t_from_bits(make_inputs(rng, m, k), dev)
    w = t_from_bits(make_inputs(rng, n, k, "weight"), dev)
    idx = torch.from_numpy(rng.permutation(k).astype(np.int16)).to(dev)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    full = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding="fused").float()
    total = torch.zeros((m, n), dtype=torch.float32, device=dev)
    total32 = torch.zeros((m, n), dtype=torch.float32, device=dev)
    cols = 0
    for r in range(world):
        layer = tp.TPShardedLinear(w, idx, *split, rank=r, world=world)
        cols += sum(layer.shard_widths)
        if layer.empty:
            continue
        qx = layer.quantize_x(x)
        # the shard's packed columns are exactly the corresponding slice of the unsharded packing
        s0, w0 = layer.shard[2]
        if w0:
            assert torch.equal(qx[2], a[2][:, s0:s0 + w0]) and torch.equal(layer.packed_w[2], b[2][:, s0 // 2:(s0 + w0) // 2])
        total += layer.ops.matmul(qx, layer.packed_w).float()
        total32 += layer.ops.matmul_f32(qx, layer.packed_w)
    assert cols == k
    # fp32 partial sums (MM_OUT_F32), one bf16 rounding after the sum: within ONE bf16 ulp of the unsharded fused product
    # whatever the world size (the fp32 additions associate differently from the single accumulator, nothing more)
    y32 = total32.to(torch.bfloat16).float()
    assert float((y32 - full).abs().max()) <= 2.0 ** -8 * float(full.abs().max())
    assert float((y32 != full).float().mean()) < 0.02
    # every partial is rounded to bf16 ONCE (tp.SHARD_ROUNDING = "fused"): |sum of partials - full| <= world half-ulps (2^-9
    # relative each) of the largest partial, plus the half-ulp of `full` itself
    err = (total - full).abs()
    assert float(err.max()) <= (world + 1) * 2.0 ** -9 * float(full.abs().max()) + 1e-3
    assert float(torch.linalg.norm(err) / torch.linalg.norm(full)) < 2.0 ** -8 * world ** 0.5


@pytest.mark.parametrize("world", [2, 4])
def test_megatron_mlp_partials_sum_to_unsharded_chain(dev, world):
    """TPMLP (gate/up column-parallel -> activate_quantize_x on the local slice -> down row-parallel) with the HIP kernels, every
    rank in turn on one GPU: the fp32 sum of the partials against the unsharded GPU chain (gate, up, activate_quantize_x over the
    full intermediate width, down), and each rank's quantized slice byte-for-byte against the corresponding columns of the
    unsharded quantization (shards are 128-aligned, so every 32-block keeps its scale)."""
    import torch
    rng = np.random.default_rng(world + 30)
    m, hid, inter = 96, 512, 2048
    in_split, down_split = (256, 128, 128), (1024, 512, 512)
    x = t_from_bits(make_inputs(rng, m, hid), dev)
    wg = t_from_bits(make_inputs(rng, inter, hid, "weight"), dev)
    wu = t_from_bits(make_inputs(rng, inter, hid, "weight"), dev)
    wd = t_from_bits(make_inputs(rng, hid, inter, "weight"), dev)
    idx = torch.from_numpy(rng.permutation(hid).astype(np.int16)).to(dev)
    mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
    qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
    g = mm(qx, mixedgemm.reorder_quantize_w4(wg, idx, *in_split))
    u = mm(qx, mixedgemm.reorder_quantize_w4(wu, idx, *in_split))
    qh = mixedgemm.activate_quantize_x(g, u, *down_split)
    full = mm(qh, mixedgemm.downproj_quantize_w4(wd, *down_split), rounding="fused").float()
    total = torch.zeros((m, hid), dtype=torch.float32, device=dev)
    covered = 0
    for r in range(world):
        mlp = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=r, world=world)
        covered += int(mlp.positions.numel())
        part = mlp.partial(mlp.quantize_x(x))
        total += part.float()
        # the local gate slice equals the corresponding columns of the unsharded gate output, bit for bit
        gl = mlp.ops.matmul(qx, mlp.packed_gate, rounding="reference")
        assert torch.equal(gl, g[:, mlp.positions.to(dev)])
    assert covered == inter
    err = (total - full).abs()
    assert float(err.max()) <= (world + 1) * 2.0 ** -9 * float(full.abs().max()) + 1e-3
    assert float(torch.linalg.norm(err) / torch.linalg.norm(full)) < 2.0 ** -8 * world ** 0.5
