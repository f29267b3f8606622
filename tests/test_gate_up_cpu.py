"""CPU tests of the host side of the fused gate / up path (no GPU): the interleaved weight layout that mm_gate_up_activate
consumes, and TPMLP's routing through a backend that offers the fused op."""
import os
import sys

import numpy as np
import torch

from micromix_amd import mixedgemm, tp
from oracle import mx_oracle as o

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import lcg  # noqa: E402
from test_tp_cpu import OracleOps, _mlp_inputs  # noqa: E402


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_interleaved_packed_weight_is_the_packing_of_the_interleaved_matrix():
    """interleave_gate_up(pack(gate), pack(up)) == pack(rows of gate and up interleaved per 128): packed codes byte for byte, and
    the scale tensors too (a 128-row tile of a scale tensor is one contiguous block of (Kseg / 128) * 512 bytes)"""
    i, k, split = 384, 512, (256, 128, 128)
    gb = lcg.bf16_normalish(31, (i, k), exp_center=122)
    ub = lcg.bf16_normalish(32, (i, k), exp_center=122)
    idx = lcg.permutation(33, k)
    pg = o.reorder_quantize(gb, idx, *split, "w4")
    pu = o.reorder_quantize(ub, idx, *split, "w4")
    inter = np.stack((gb.reshape(i // 128, 128, k), ub.reshape(i // 128, 128, k)), axis=1).reshape(2 * i, k)
    want = o.reorder_quantize(inter, idx, *split, "w4")
    got = mixedgemm.interleave_gate_up(tuple(_t(a) for a in pg), tuple(_t(a) for a in pu))
    for s in range(3):
        assert np.array_equal(got[s].numpy(), want[s]), s
        assert np.array_equal(got[3 + s].numpy(), want[3 + s]), s       # N = 768 is a multiple of 128: no padding rows at all
    back_g, back_u = mixedgemm.deinterleave_gate_up(got)
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(back_g, pg)) and all(np.array_equal(a.numpy(), b) for a, b in zip(back_u, pu))


class FusedOracleOps(OracleOps):
    """OracleOps + the two optional ops of the HIP backend, restated with the oracle: what TPMLP must get back from
    `gate_up_activate` is activate_quantize(matmul(qx, gate), matmul(qx, up)) on the de-interleaved weights"""
    calls = 0

    @staticmethod
    def interleave_gate_up(gate, up):
        return tuple(t.numpy() for t in mixedgemm.interleave_gate_up(tuple(_t(a) for a in gate), tuple(_t(a) for a in up)))

    @staticmethod
    def deinterleave_gate_up(packed):
        g, u = mixedgemm.deinterleave_gate_up(tuple(_t(t) for t in packed))
        return tuple(t.numpy() for t in g), tuple(t.numpy() for t in u)

    @classmethod
    def gate_up_activate(cls, a, b, kn, ks, ko):
        cls.calls += 1
        gate, up = mixedgemm.deinterleave_gate_up(tuple(_t(t) for t in b))
        mm = lambda w: o.matmul(a[0], w[0].numpy(), a[1], w[1].numpy(), a[2], w[2].numpy(), a[3], w[3].numpy(), a[4], w[4].numpy(),
                                a[5], w[5].numpy(), rounding="reference")
        return o.activate_quantize(mm(gate), mm(up), kn, ks, ko)


def test_tpmlp_routes_through_the_fused_op_and_keeps_its_results():
    m, h, inter, in_split, down_split, x, wg, wu, wd, idx = _mlp_inputs()
    for world in (1, 2):
        for rank in range(world):
            plain = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=rank, world=world, ops=OracleOps)
            fused = tp.TPMLP(wg, wu, wd, idx, in_split, down_split, rank=rank, world=world, ops=FusedOracleOps)
            assert not plain.fused and fused.fused
            before = FusedOracleOps.calls
            qx = plain.quantize_x(x)
            p0, p1 = plain.partial(qx), fused.partial(qx)
            assert FusedOracleOps.calls == before + 1
            assert torch.equal(p0, p1)
            assert torch.equal(plain.partial(qx, fp32=True), fused.partial(qx, fp32=True))
            # the packed gate / up of the fused rank are recovered on demand (tests/test_tp_gpu.py reads them)
            for a, b in zip(fused.packed_gate, plain.packed_gate):
                assert np.array_equal(a, b)
