"""CPU tests of the tensor-parallel path: the K-shard planner, and a world_size-2 `gloo` run of
TPShardedLinear whose per-rank compute is the oracle (the HIP kernels need a GPU; the partition, the
sub-index gather, the SF bookkeeping and the all-reduce are what is under test here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from micromix_amd import tp
from oracle import mx_oracle as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import lcg  # noqa: E402


@pytest.mark.parametrize("split", [(0, 0, 4096), (2048, 128, 1920), (4096, 0, 0), (12288, 1024, 1024), (128, 0, 0),
                                   (3584, 256, 256), (7168, 512, 6656)])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_plan_covers_every_column_once_and_balances_cost(split, world):
    plan = tp.plan_k_shards(*split, world)
    assert len(plan) == world
    for s in range(3):
        pos = 0
        for r in range(world):
            start, width = plan[r][s]
            assert start == pos and width % 128 == 0 and width >= 0
            pos += width
        assert pos == split[s]
    cost = [sum(plan[r][s][1] * tp.SEGMENT_COST[s] for s in range(3)) for r in range(world)]
    assert max(cost) - min(cost) <= 128 * max(tp.SEGMENT_COST) + 1e-9     # within one granule of the heaviest kind


def test_plan_rejects_bad_widths():
    with pytest.raises(ValueError):
        tp.plan_k_shards(100, 0, 28, 2)
    with pytest.raises(ValueError):
        tp.plan_k_shards(128, 0, 0, 0)


class OracleOps:
    """test-only compute backend: the CPU oracle behind the same three calls as the HIP backend."""

    @staticmethod
    def _bits(t):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)

    @classmethod
    def quantize_w4(cls, w, index, kn, ks, ko):
        return o.reorder_quantize(cls._bits(w), index.numpy(), kn, ks, ko, "w4", gather_subset=True)

    @classmethod
    def quantize_x(cls, x, index, kn, ks, ko):
        return o.reorder_quantize(cls._bits(x), index.numpy(), kn, ks, ko, "x", gather_subset=True)

    @staticmethod
    def matmul(a, b, out=None, rounding=tp.SHARD_ROUNDING):
        d = o.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=rounding)
        t = torch.from_numpy(d.view(np.int16).copy()).view(torch.bfloat16)
        if out is not None:
            out.copy_(t)
            return out
        return t

    @staticmethod
    def matmul_f32(a, b, out=None):
        _, t = o.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding="fused", return_f64=True)
        return torch.from_numpy(t.astype(np.float32))

    @classmethod
    def activate_quantize(cls, a, b, kn, ks, ko):
        return o.activate_quantize(cls._bits(a), cls._bits(b), kn, ks, ko)

    @classmethod
    def downproj_quantize_w4(cls, w, kn, ks, ko):
        return o.downproj_quantize(cls._bits(w), kn, ks, ko, True)


def _inputs():
    m, n, k, split = 24, 64, 1024, (512, 128, 384)
    tb = lambda bits: torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)
    x = tb(lcg.bf16_normalish(1, (m, k)))
    w = tb(lcg.bf16_normalish(2, (n, k), exp_center=122))
    idx = torch.from_numpy(lcg.permutation(3, k))
    return m, n, k, split, x, w, idx


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, k, split, x, w, idx = _inputs()
        layer = tp.TPShardedLinear(w, idx, *split, rank=rank, world=world, group=dist.group.WORLD, ops=OracleOps)
        y = layer(x.reshape(2, m // 2, k))
        assert y.shape == (2, m // 2, n)
        y32 = layer(x.reshape(2, m // 2, k), fp32_partials=True)        # fp32 partial sums, one bf16 rounding after the all-reduce
        ret[rank] = (layer.shard_widths, y.reshape(m, n).float().numpy(), y32.reshape(m, n).float().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_unsharded_oracle():
    world = 2
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    m, n, k, split, x, w, idx = _inputs()
    bits = OracleOps._bits
    qx = o.reorder_quantize(bits(x), idx.numpy(), *split, "x")
    qw = o.reorder_quantize(bits(w), idx.numpy(), *split, "w4")
    want, f64 = o.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5],
                         rounding="fused", return_f64=True)
    assert np.array_equal(res[0][1], res[1][1])                       # all-reduce: every rank holds the sum
    assert [sum(wd) for wd in zip(res[0][0], res[1][0])] == list(split)
    got = res[0][1].astype(np.float64)
    # each rank rounds its partial to bf16 once (tp.SHARD_ROUNDING), the all-reduce adds the two in bf16: three roundings of
    # values bounded by the partial magnitudes.  |partials| <= S, the absolute dot product.
    a = [o.dequant_segment(qx[i], qx[3 + i], m, split[i], f) for i, f in enumerate(("fp4", "fp6", "fp8"))]
    b = [o.dequant_segment(qw[i], qw[3 + i], n, split[i], "fp4") for i in range(3)]
    S = sum(np.abs(ai).astype(np.float64) @ np.abs(bi).astype(np.float64).T for ai, bi in zip(a, b))
    assert np.all(np.abs(got - f64) <= 2.0 ** -7 * (np.abs(f64) + 0.25 * S) + 1e-30)
    assert np.linalg.norm(got - f64) / np.linalg.norm(f64) < 4e-3
    # fp32 partials: the sum of the ranks' fp32 accumulators rounded ONCE = the unsharded fused product up to the fp32 rounding of
    # the partial sums (at most one bf16 ulp apart, and almost always equal)
    assert np.array_equal(res[0][2], res[1][2])
    ulp = o.bf16_ulp_distance(o.f32_to_bf16(res[0][2].astype(np.float32)), want)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.01


def _worker_chunked(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, k, split = 300, 32, 512, (256, 128, 128)
        tb = lambda bits: torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)
        x, w, idx = tb(lcg.bf16_normalish(11, (m, k))), tb(lcg.bf16_normalish(12, (n, k), exp_center=122)), torch.from_numpy(lcg.permutation(13, k))
        layer = tp.TPShardedLinear(w, idx, *split, rank=rank, world=world, group=dist.group.WORLD, ops=OracleOps)
        qx = layer.quantize_x(x)
        whole = layer.matmul_allreduce(qx, out=torch.empty((m, n), dtype=torch.bfloat16))                       # one all-reduce
        chunked = layer.matmul_allreduce(qx, out=torch.empty((m, n), dtype=torch.bfloat16), chunk_rows=128)   # 128 + 128 + 44 rows
        ret[rank] = bool(torch.equal(whole, chunked))
    finally:
        dist.destroy_process_group()


def test_chunked_async_allreduce_equals_single_allreduce():
    """rows processed in 128-aligned chunks with an asynchronous all-reduce per chunk (compute/communication overlap on the
    GPU) give exactly the single-all-reduce result"""
    world = 2
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_chunked, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------------------------------
# column-parallel (N-shard) layer and the Megatron MLP pairing ("cheaper alternative")
# ---------------------------------------------------------------------------------------------------------------------
def _mlp_inputs():
    m, h, inter = 20, 256, 1024
    in_split, down_split = (128, 0, 128), (512, 128, 384)
    tb = lambda bits: torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)
    x = tb(lcg.bf16_normalish(21, (m, h)))
    wg = tb(lcg.bf16_normalish(22, (inter, h), exp_center=123))
    wu = tb(lcg.bf16_normalish(23, (inter, h), exp_center=123))
    wd = tb(lcg.bf16_normalish(24, (h, inter), exp_center=122))
    idx = torch.from_numpy(lcg.permutation(25, h))
    return m, h, inter, in_split, down_split, x, wg, wu, wd, idx


def _worker_mlp(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, h, inter, in_split, down_split, x, wg, wu, wd, idx = _mlp_inputs()
        mlp = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=rank, world=world, group=dist.group.WORLD, ops=OracleOps)
        y = mlp(x.reshape(2, m // 2, h))
        y32 = mlp(x.reshape(2, m // 2, h), fp32_partials=True)
        col = tp.ColumnParallelLinear(wg, idx, *in_split, rank=rank, world=world, group=dist.group.WORLD, ops=OracleOps,
                                      gather_output=True)
        g = col(x)
        ret[rank] = (mlp.widths, y.reshape(m, h).float().numpy(), g.float().numpy(), col.features.numpy(),
                     y32.reshape(m, h).float().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_megatron_mlp_and_column_parallel():
    world = 2
    port = 33500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_mlp, args=(world, port, ret), nprocs=world, join=True)
        res = dict(ret)
    m, h, inter, in_split, down_split, x, wg, wu, wd, idx = _mlp_inputs()
    bits = OracleOps._bits
    mm = lambda a, b, r: o.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=r)
    # unsharded oracle chain: quantize -> gate, up (reference rounding) -> silu*mul + quantize -> down
    qx = o.reorder_quantize(bits(x), idx.numpy(), *in_split, "x")
    g = mm(qx, o.reorder_quantize(bits(wg), idx.numpy(), *in_split, "w4"), "reference")
    u = mm(qx, o.reorder_quantize(bits(wu), idx.numpy(), *in_split, "w4"), "reference")
    qh = o.activate_quantize(g, u, *down_split)
    qd = o.downproj_quantize(bits(wd), *down_split, True)
    want, f64 = o.matmul(qh[0], qd[0], qh[1], qd[1], qh[2], qd[2], qh[3], qd[3], qh[4], qd[4], qh[5], qd[5], rounding="fused",
                         return_f64=True)
    # column-parallel: the gathered gate output is the unsharded gate output, bit for bit (no reduction involved)
    assert np.array_equal(res[0][2], o.bf16_to_f32(g)) and np.array_equal(res[1][2], o.bf16_to_f32(g))
    assert np.array_equal(np.concatenate([res[0][3], res[1][3]]), np.arange(inter))
    # the MLP: both ranks hold the same sum; the shards cover the intermediate features exactly once
    assert np.array_equal(res[0][1], res[1][1])
    assert [a + b for a, b in zip(res[0][0], res[1][0])] == list(down_split)
    got = res[0][1].astype(np.float64)
    a = [o.dequant_segment(qh[i], qh[3 + i], m, down_split[i], f) for i, f in enumerate(("fp4", "fp6", "fp8"))]
    b = [o.dequant_segment(qd[i], qd[3 + i], h, down_split[i], "fp4") for i in range(3)]
    S = sum(np.abs(ai).astype(np.float64) @ np.abs(bi).astype(np.float64).T for ai, bi in zip(a, b))
    assert np.all(np.abs(got - f64) <= 2.0 ** -7 * (np.abs(f64) + 0.25 * S) + 1e-30)
    assert np.linalg.norm(got - f64) / np.linalg.norm(f64) < 4e-3
    # fp32 partial sums: one bf16 rounding after the all-reduce = the unsharded fused product to within one ulp
    assert np.array_equal(res[0][4], res[1][4])
    ulp = o.bf16_ulp_distance(o.f32_to_bf16(res[0][4].astype(np.float32)), want)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.01


def test_shard_positions_partition_the_intermediate_features():
    for split in ((512, 128, 384), (12288, 1024, 1024), (3584, 256, 256)):
        for world in (1, 2, 4, 8):
            plan = tp.plan_k_shards(*split, world)
            pos = torch.cat([tp.shard_positions(*split, plan[r]) for r in range(world)])
            assert torch.equal(torch.sort(pos).values, torch.arange(sum(split)))


def test_column_parallel_gather_needs_equal_shards():
    """all_gather with ranks of different widths hangs or corrupts on RCCL: the constructor refuses such a layout, and a rank
    without features returns an [M, 0] slice instead of failing while the other ranks wait in a collective"""
    tb = lambda bits: torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)
    k, split = 256, (128, 0, 128)
    idx = torch.from_numpy(np.random.default_rng(0).permutation(k).astype(np.int16))
    w = tb(lcg.bf16_normalish(3, (384, k), exp_center=121))                # 3 granules of 128 features over 2 ranks: 256 + 128
    with pytest.raises(ValueError, match="equal, non-empty shards"):
        tp.ColumnParallelLinear(w, idx, *split, rank=0, world=2, ops=OracleOps, gather_output=True)
    with pytest.raises(ValueError, match="equal, non-empty shards"):
        tp.ColumnParallelLinear(w, idx, *split, rank=1, world=2, ops=OracleOps, gather_output=True)
    tp.ColumnParallelLinear(w[:256], idx, *split, rank=1, world=2, ops=OracleOps, gather_output=True)      # 128 + 128: fine
    lone = tp.ColumnParallelLinear(w[:128], idx, *split, rank=1, world=2, ops=OracleOps)                   # rank 1 gets nothing
    assert lone.empty
    x = tb(lcg.bf16_normalish(4, (5, k)))
    y = lone.matmul(lone.quantize_x(x))
    assert tuple(y.shape) == (5, 0) and y.dtype == torch.bfloat16
