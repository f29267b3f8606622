"""Tensor-parallel (row-parallel, K-sharded) QLinear over RCCL -- new design, the reference has no
collectives at all (BASELINE.json north_star).

Rank g owns, for each of the three reordered segments (MXFP4 | MXFP6 | MXFP8), a 128-column-aligned
slice of the *reordered* columns.  It
  1. packs only its slice of the weight once (reorder_quantize_w4 over a sub-index),
  2. per forward quantizes only its slice of the activation columns (the gather kernel reads the full
     replicated x row and produces KN_g + KS_g + KO_g columns),
  3. runs the fused three-segment GEMM on its shard -> partial [M, N] bf16,
  4. sums the partials with ONE all-reduce on the bf16 output (RCCL over xGMI; `nccl` backend).
Because MX blocks are 32 columns and shards are 128-aligned, every block keeps exactly the scale it has
in the unsharded layer.  Each rank runs its shard with ONE bf16 rounding (`SHARD_ROUNDING = "fused"`: a rank's
three segments are partial sums of the same output, rounding between them would only add error) and the
all-reduce adds the `world` rounded partials in bf16, so the result differs from the unsharded fused product by
at most ~world bf16 half-ulps of the largest partial (not bit-equal: tests/test_tp_gpu.py states the bound).

Two more layouts live here because section 8e asks for them next to the K-shard:
  * `ColumnParallelLinear` (N-shard): rank g owns a 128-aligned set of OUTPUT features, full K, no reduction; the
    local [M, N_g] slice is what the next row-parallel layer consumes (or `all_gather`s into [M, N]);
  * `TPMLP` (Megatron pairing): gate/up column-parallel over the intermediate features that the rank's down_proj
    K-shard consumes -> silu(gate)*up + quantize on the LOCAL slice (`activate_quantize_x`) -> down_proj
    row-parallel -> ONE all-reduce of [M, hidden] per MLP instead of three.

Shards are balanced by COST, not width: an fp8-operand column costs ~1.67x an fp4 column on the MFMA
(measured issue rates, tools/mfma_rate.py), so each rank gets a share of every segment.

The compute backend is injectable (`ops`) so that the partition / reduction logic can be exercised on
CPU with `gloo` by the tests; the product default is the HIP `mixedgemm` module (no CPU fallback).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

# relative MFMA cost per column of (fp4, fp6, fp8) activations against fp4 weights (w4 mode)
SEGMENT_COST = (1.0, 1.07, 1.67)
# rounding mode of a rank's partial product (see the module docstring)
SHARD_ROUNDING = "fused"


def plan_k_shards(kn: int, ks: int, ko: int, world: int) -> List[Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]]]:
    """Per rank, for each segment, (start, width) in reordered-segment columns; widths are multiples of
    128, every column is owned exactly once, and each segment is split as evenly as 128-granules allow
    (the remainder granules of different segments are dealt to different ranks, heaviest segment first,
    so that the per-rank cost stays balanced)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    for k in (kn, ks, ko):
        if k < 0 or k % 128:
            raise ValueError("segment widths must be non-negative multiples of 128")
    widths = [[0] * 3 for _ in range(world)]
    load = [0.0] * world
    order = sorted(range(3), key=lambda s: -SEGMENT_COST[s])
    for s in order:
        gran = (kn, ks, ko)[s] // 128
        base, rem = divmod(gran, world)
        for r in range(world):
            widths[r][s] = base * 128
            load[r] += base * 128 * SEGMENT_COST[s]
        for _ in range(rem):                      # leftover granules go to the currently lightest ranks
            r = min(range(world), key=lambda i: (load[i], i))
            widths[r][s] += 128
            load[r] += 128 * SEGMENT_COST[s]
    plan = []
    starts = [0, 0, 0]
    for r in range(world):
        plan.append(tuple((starts[s], widths[r][s]) for s in range(3)))
        for s in range(3):
            starts[s] += widths[r][s]
    return plan


def shard_index(reorder_index: torch.Tensor, kn: int, ks: int, ko: int, shard) -> torch.Tensor:
    """the rank's sub-index: its slice of each segment of the full reorder index, concatenated."""
    seg_base = (0, kn, kn + ks)
    parts = [reorder_index[seg_base[s] + shard[s][0]: seg_base[s] + shard[s][0] + shard[s][1]] for s in range(3)]
    return torch.cat(parts).contiguous()


class _HipOps:
    """default backend: the HIP kernels (micromix_amd.mixedgemm)."""

    @staticmethod
    def quantize_w4(w, index, kn, ks, ko):
        from . import mixedgemm
        return mixedgemm._quantize(w, index, kn, ks, ko, "w4", "reorder_quantize_w4", gather_subset=True)

    @staticmethod
    def quantize_x(x, index, kn, ks, ko):
        from . import mixedgemm
        return mixedgemm._quantize(x, index, kn, ks, ko, "x", "reorder_quantize_x", gather_subset=True)

    @staticmethod
    def matmul(a, b, out=None, rounding=SHARD_ROUNDING):
        from . import mixedgemm
        return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out, rounding=rounding)

    @staticmethod
    def matmul_f32(a, b, out=None):
        """the shard's product as the unrounded fp32 accumulator (MM_OUT_F32)"""
        from . import mixedgemm
        return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out, rounding="fused",
                                out_dtype=torch.float32)

    @staticmethod
    def activate_quantize(a, b, kn, ks, ko):
        from . import mixedgemm
        return mixedgemm.activate_quantize_x(a, b, kn, ks, ko)

    @staticmethod
    def downproj_quantize_w4(w, kn, ks, ko):
        from . import mixedgemm
        return mixedgemm.downproj_quantize_w4(w, kn, ks, ko)

    # optional ops (a backend without them makes TPMLP run gate, up and activate_quantize as three steps)
    @staticmethod
    def interleave_gate_up(gate, up):
        from . import mixedgemm
        return mixedgemm.interleave_gate_up(gate, up)

    @staticmethod
    def deinterleave_gate_up(packed):
        from . import mixedgemm
        return mixedgemm.deinterleave_gate_up(packed)

    @staticmethod
    def gate_up_activate(a, b, kn, ks, ko):
        """gate, up, silu(gate) * up and the quantization for down_proj as one launch (mm_gate_up_activate)"""
        from . import mixedgemm
        return mixedgemm.gate_up_activate(a, b, kn, ks, ko, rounding="reference")


class TPShardedLinear:
    """One QLinear, K-sharded over `world` ranks.  w [N, K] bf16 and reorder_index [K] int16 are the FULL
    tensors (each rank slices its own part); x passed to forward is the full replicated [M, K] activation."""

    def __init__(self, w: torch.Tensor, reorder_index: torch.Tensor, p4: int, p6: int, p8: int, rank: int, world: int,
                 group=None, ops=None, bias: torch.Tensor | None = None):
        self.rank, self.world, self.group = rank, world, group
        self.ops = ops if ops is not None else _HipOps
        self.N, self.K = w.shape
        if p4 + p6 + p8 != self.K:
            raise ValueError("p4 + p6 + p8 must equal in_features")
        self.plan = plan_k_shards(p4, p6, p8, world)
        self.shard = self.plan[rank]
        self.shard_widths = [s[1] for s in self.shard]
        self.index = shard_index(reorder_index, p4, p6, p8, self.shard)
        self.empty = sum(self.shard_widths) == 0
        self.bias = bias
        if not self.empty:
            self.packed_w = self.ops.quantize_w4(w, self.index, *self.shard_widths)

    def quantize_x(self, x: torch.Tensor):
        return None if self.empty else self.ops.quantize_x(x, self.index, *self.shard_widths)

    @staticmethod
    def _row_chunk(qx, widths, r0, r1):
        """rows [r0, r1) of a quantized activation tuple (r0 a multiple of 128): the packed segments are row-major and the scale
        tensors are tiled by 128 rows (512 bytes per row tile and 128-column slab), so a 128-aligned row range is contiguous."""
        out = [t[r0:r1] for t in qx[:3]]
        for sf, kseg in zip(qx[3:], widths):
            out.append(sf[(r0 // 128) * (kseg // 128) * 512:])
        return tuple(out)

    def matmul_allreduce(self, qx, out: torch.Tensor | None = None, m: int | None = None, chunk_rows: int | None = None) -> torch.Tensor:
        """partial GEMM of this rank's K-shard + ONE sum over the ranks.  With `chunk_rows` (a multiple of 128) the rows are
        processed in chunks whose all-reduce is issued asynchronously, so the collective of chunk i runs on RCCL's stream while
        the GEMM of chunk i+1 runs on the compute stream.  Off by default: for one 4096-wide linear the all-reduce is many times
        longer than the sharded GEMM and a ring collective is most efficient in one large message, so chunking only pays when
        the GEMM share is large (long K per rank, small world)."""
        import torch.distributed as dist
        if self.empty:       # more ranks than 128-column granules: contribute zeros
            if out is None:
                raise ValueError("an empty shard needs `out` (or use forward)")
            out.zero_()
            if self.world > 1:
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        else:
            rows = qx[0].shape[0] if qx[0].shape[1] else (qx[1].shape[0] if qx[1].shape[1] else qx[2].shape[0])
            if self.world > 1 and chunk_rows is not None and chunk_rows % 128 == 0 and rows > chunk_rows:
                if out is None:
                    out = torch.empty((rows, self.N), dtype=torch.bfloat16, device=qx[0].device)
                works = []
                for r0 in range(0, rows, chunk_rows):
                    r1 = min(rows, r0 + chunk_rows)
                    self.ops.matmul(self._row_chunk(qx, self.shard_widths, r0, r1), self.packed_w, out=out[r0:r1])
                    works.append(dist.all_reduce(out[r0:r1], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                for w in works:
                    w.wait()
            else:
                out = self.ops.matmul(qx, self.packed_w, out=out)
                if self.world > 1:
                    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        if self.bias is not None:
            out += self.bias
        return out

    def matmul_allreduce_f32(self, qx, m: int | None = None) -> torch.Tensor:
        """the same reduction with FP32 partial sums: every rank's GEMM leaves its unrounded fp32 accumulator (MM_OUT_F32), the
        all-reduce adds fp32 values and the result is rounded to bf16 ONCE -- within one bf16 ulp of the unsharded fused product
        whatever the world size (the bf16 path accumulates up to `world` half-ulps), at twice the bytes on the wire."""
        import torch.distributed as dist
        if self.empty:
            if m is None:
                raise ValueError("an empty shard needs the row count `m`")
            part = torch.zeros((m, self.N), dtype=torch.float32, device=self.index.device)
        else:
            part = self.ops.matmul_f32(qx, self.packed_w)
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        out = part.to(torch.bfloat16)
        if self.bias is not None:
            out += self.bias
        return out

    def forward(self, x: torch.Tensor, fp32_partials: bool = False) -> torch.Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.K).contiguous()
        if fp32_partials:
            return self.matmul_allreduce_f32(self.quantize_x(x2), m=x2.shape[0]).reshape(*lead, self.N)
        out = torch.empty((x2.shape[0], self.N), dtype=torch.bfloat16, device=x2.device)
        y = self.matmul_allreduce(self.quantize_x(x2), out=out)
        return y.reshape(*lead, self.N)

    __call__ = forward


def shard_positions(kn: int, ks: int, ko: int, shard) -> torch.Tensor:
    """positions (in the reordered column order fp4 | fp6 | fp8) that a K-shard owns: its slice of each segment, concatenated"""
    seg_base = (0, kn, kn + ks)
    return torch.cat([torch.arange(seg_base[s] + shard[s][0], seg_base[s] + shard[s][0] + shard[s][1]) for s in range(3)])


class ColumnParallelLinear:
    """One QLinear, N-sharded: this rank owns the output features `features` (a LongTensor, or None for an even 128-aligned
    split of range(N)), the full K and the full reorder index.  forward(x) -> the local [M, N_g] slice in the reference's
    rounding (this is a complete product, not a partial sum); `gather_output=True` all-gathers [M, N] (equal shard sizes only)."""

    def __init__(self, w: torch.Tensor, reorder_index: torch.Tensor, p4: int, p6: int, p8: int, rank: int, world: int,
                 group=None, ops=None, bias: torch.Tensor | None = None, features: torch.Tensor | None = None,
                 gather_output: bool = False):
        self.rank, self.world, self.group = rank, world, group
        self.ops = ops if ops is not None else _HipOps
        self.N, self.K = w.shape
        self.split = (p4, p6, p8)
        if p4 + p6 + p8 != self.K:
            raise ValueError("p4 + p6 + p8 must equal in_features")
        if features is None:
            gran = (self.N + 127) // 128
            base, rem = divmod(gran, world)
            g0 = rank * base + min(rank, rem)
            g1 = g0 + base + (1 if rank < rem else 0)
            features = torch.arange(min(g0 * 128, self.N), min(g1 * 128, self.N))
        self.features = features
        self.index = reorder_index.to(torch.int16).contiguous()
        self.empty = features.numel() == 0
        self.gather_output = gather_output
        if gather_output and world > 1:
            # all_gather needs the same width on every rank (RCCL hangs or corrupts on mismatched shapes) and no empty shard
            if features is not None and self.features.numel() * world != self.N:
                raise ValueError(f"gather_output=True needs equal, non-empty shards on every rank: rank {rank} holds "
                                 f"{self.features.numel()} of {self.N} features for world size {world} (shards are 128-feature "
                                 "granules: N must be a multiple of 128 * world, or pass equal `features` slices)")
        self.bias = bias[features.to(bias.device)] if bias is not None else None
        if not self.empty:
            self.packed_w = self.ops.quantize_w4(w[features.to(w.device)].contiguous(), self.index, p4, p6, p8)

    def quantize_x(self, x: torch.Tensor):
        return self.ops.quantize_x(x, self.index, *self.split)

    def matmul(self, qx):
        if self.empty:      # a rank without features (N < 128 * world): an [M, 0] slice, so that callers need no special case
            return torch.empty((qx[0].shape[0], 0), dtype=torch.bfloat16, device=getattr(qx[0], "device", None))
        y = self.ops.matmul(qx, self.packed_w, rounding="reference")
        if self.bias is not None:
            y = y + self.bias
        return y

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        lead = x.shape[:-1]
        y = self.matmul(self.quantize_x(x.reshape(-1, self.K).contiguous()))
        if self.gather_output and self.world > 1:
            parts = [torch.empty_like(y) for _ in range(self.world)]
            dist.all_gather(parts, y, group=self.group)
            y = torch.cat(parts, dim=1)
        return y.reshape(*lead, y.shape[-1])

    __call__ = forward


class TPMLP:
    """gate/up/down of one decoder MLP (qLlamaLayer.py:336-387) with ONE all-reduce (Megatron pairing).

    Inputs are the FULL tensors: w_gate, w_up [I, H] whose rows are already in the down_proj's reordered order and w_down
    [H, I] whose columns are in that same order -- the reference folds the down_proj reorder into the gate/up output order
    (`out_reorder_index`, qLlamaLayer.py:341,354) and packs the down weight with `downproj_quantize_w4` in natural column
    order (bindings.cpp:363-387); `in_index` / `in_split` = reorder index and (p4, p6, p8) of the hidden input of gate/up,
    `down_split` = (p4, p6, p8) of the intermediate features.

    Rank g takes the K-shard plan of the down_proj (`plan_k_shards(*down_split)`), i.e. 128-aligned slices of the fp4 / fp6 / fp8
    ranges of the intermediate features; its gate/up shards are exactly those rows.  Then
        qx = quantize(x)                      full hidden input, replicated (or shared with the caller through `quantize_x`)
        g, u = gate_g(qx), up_g(qx)           column-parallel: [M, I_g], complete products, no communication
        qh = activate_quantize_x(g, u, ...)   silu(g) * u -> mixed quantize of the LOCAL slice; every 32-block and its scale are
                                              the ones the unsharded layer computes, because shards are 128-aligned
                                              (with the HIP backend these two lines are ONE launch, mm_gate_up_activate: g and u
                                              never leave the CU; bit-identical)
        part = down_g(qh)                     row-parallel partial [M, H], one bf16 rounding
        y = all_reduce(part)                  the only collective: M * H * 2 bytes
    """

    def __init__(self, w_gate: torch.Tensor, w_up: torch.Tensor, w_down: torch.Tensor, in_index: torch.Tensor, in_split,
                 down_split, rank: int, world: int, group=None, ops=None):
        self.rank, self.world, self.group = rank, world, group
        self.ops = ops if ops is not None else _HipOps
        self.I, self.H = w_gate.shape
        if tuple(w_up.shape) != (self.I, self.H) or tuple(w_down.shape) != (self.H, self.I):
            raise ValueError("expected w_gate, w_up [I, H] and w_down [H, I]")
        if sum(in_split) != self.H or sum(down_split) != self.I:
            raise ValueError("in_split must sum to the hidden size and down_split to the intermediate size")
        self.in_split = tuple(in_split)
        self.in_index = in_index.to(torch.int16).contiguous()
        self.shard = plan_k_shards(*down_split, world)[rank]
        self.widths = tuple(s[1] for s in self.shard)
        self.positions = shard_positions(*down_split, self.shard)
        self.empty = self.positions.numel() == 0
        if not self.empty:
            sel = self.positions.to(w_gate.device)
            packed_gate = self.ops.quantize_w4(w_gate[sel].contiguous(), self.in_index, *self.in_split)
            packed_up = self.ops.quantize_w4(w_up[sel].contiguous(), self.in_index, *self.in_split)
            # with a backend that has the fused kernel the rank keeps ONE weight, gate and up rows interleaved per 128 features
            # (mixedgemm.interleave_gate_up); `packed_gate` / `packed_up` are then recovered on demand (tests)
            self.fused = hasattr(self.ops, "gate_up_activate")
            if self.fused:
                self.packed_gate_up = self.ops.interleave_gate_up(packed_gate, packed_up)
            else:
                self._packed_gate, self._packed_up = packed_gate, packed_up
            self.packed_down = self.ops.downproj_quantize_w4(w_down[:, sel].contiguous(), *self.widths)

    def _split_gate_up(self):
        return self.ops.deinterleave_gate_up(self.packed_gate_up)

    @property
    def packed_gate(self):
        return self._split_gate_up()[0] if self.fused else self._packed_gate

    @property
    def packed_up(self):
        return self._split_gate_up()[1] if self.fused else self._packed_up

    def quantize_x(self, x2d: torch.Tensor):
        return self.ops.quantize_x(x2d, self.in_index, *self.in_split)

    def partial(self, qx, fp32: bool = False) -> torch.Tensor | None:
        """this rank's [M, H] partial of the MLP output (None for an empty shard); fp32 = the unrounded accumulator (MM_OUT_F32)"""
        if self.empty:
            return None
        if self.fused:
            qh = self.ops.gate_up_activate(qx, self.packed_gate_up, *self.widths)
        else:
            g = self.ops.matmul(qx, self.packed_gate, rounding="reference")
            u = self.ops.matmul(qx, self.packed_up, rounding="reference")
            qh = self.ops.activate_quantize(g, u, *self.widths)
        return self.ops.matmul_f32(qh, self.packed_down) if fp32 else self.ops.matmul(qh, self.packed_down)

    def forward(self, x: torch.Tensor, fp32_partials: bool = False) -> torch.Tensor:
        """fp32_partials: the ranks' partial sums travel and add as fp32 and are rounded to bf16 once (twice the bytes on the wire,
        a result within one bf16 ulp of the unsharded fused product whatever the world size)"""
        import torch.distributed as dist
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.H).contiguous()
        part = self.partial(self.quantize_x(x2), fp32=fp32_partials)
        if part is None:
            part = torch.zeros((x2.shape[0], self.H), dtype=torch.float32 if fp32_partials else torch.bfloat16, device=x2.device)
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        return part.to(torch.bfloat16).reshape(*lead, self.H)

    __call__ = forward
