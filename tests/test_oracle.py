"""CPU tests of the oracle itself: format definitions, packers, scale rule, SF layout, the two
independent restatements (numpy vs C) against each other, and the committed golden fixture."""
import hashlib
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as c
from oracle import mx_oracle as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import lcg  # noqa: E402
import make_golden as mg  # noqa: E402

ELS = ("fp4", "fp6", "fp8")


def test_decode_tables_match_ocp_definitions():
    assert sorted(set(np.abs(o.decode_table("fp4")).tolist())) == [0, 0.5, 1, 1.5, 2, 3, 4, 6]   # E2M1
    t6 = o.decode_table("fp6")
    assert t6.max() == 28.0 and np.abs(t6[t6 != 0]).min() == 0.0625 and len(t6) == 64            # E3M2
    t8 = o.decode_table("fp8")
    assert np.nanmax(t8) == 448.0 and np.isnan(t8[0x7F]) and np.isnan(t8[0xFF])                  # E4M3fn
    assert t8[0x38] == 1.0 and t8[0x01] == 2.0 ** -9
    for el in ELS:   # the reference's FP4_MAX / FP6_MAX / FP8_MAX (reorder.cu:17-19)
        assert np.nanmax(o.decode_table(el)) == o.FORMATS[el]["fmax"]


@pytest.mark.parametrize("el", ELS)
def test_encode_is_rne_by_definition_for_every_bf16(el):
    """numpy bit-trick encoder == C nearest-value search (ties to even code) == C bit-trick,
    for every finite bf16 value (clipped into range, as the quantizer guarantees)."""
    L = c.lib()
    x = o.bf16_to_f32(np.arange(65536, dtype=np.uint16))
    x = x[np.isfinite(x)]
    fm = o.FORMATS[el]["fmax"]
    x = np.clip(x, -2 * fm, 2 * fm)
    enc = o.encode(x, el)
    fid = c.FMT[el]
    for i in range(len(x)):
        v = float(x[i])
        assert L.mxo_encode_search(v, fid) == enc[i] == L.mxo_encode_fast(v, fid), (el, v)
    # decode(encode(x)) is the nearest representable value
    tab = np.unique(np.abs(o.decode_table(el)[~np.isnan(o.decode_table(el))]))
    err = np.abs(np.abs(o.decode(enc, el)) - np.minimum(np.abs(x), fm))
    best = np.abs(tab[None, :] - np.minimum(np.abs(x), fm)[:, None]).min(1)
    assert np.array_equal(err, best)


def test_fp8_encode_matches_torch_cast():
    """independent implementation: torch's float8_e4m3fn cast (RNE, values pre-clamped)."""
    import torch
    x = o.bf16_to_f32(np.arange(65536, dtype=np.uint16))
    x = np.clip(x[np.isfinite(x)], -448, 448)
    got = torch.from_numpy(x.copy()).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(got, o.encode(x, "fp8"))


def test_e8m0_matches_torch_cast():
    import torch
    e = np.arange(-127, 128)
    t = torch.from_numpy(np.exp2(e.astype(np.float64)).astype(np.float32))
    got = t.to(torch.float8_e8m0fnu).view(torch.uint8).numpy()
    assert np.array_equal(got[1:], (e + 127).astype(np.uint8)[1:])   # 2^-127 is an fp32 subnormal: skip


@given(st.lists(st.integers(0, 63), min_size=32, max_size=32))
@settings(max_examples=200, deadline=None)
def test_pack_roundtrip(codes):
    a = np.array(codes, dtype=np.uint8)
    assert np.array_equal(o.unpack_fp6(o.pack_fp6(a)), a)
    assert np.array_equal(o.unpack_fp4(o.pack_fp4(a & 0xF)), a & 0xF)
    assert o.pack_fp6(a).shape == (24,) and o.pack_fp4(a).shape == (16,)


def test_pack_byte_order():
    # reorder.cu:30-33: element 2i -> low nibble; reorder.cu:60-62: b0=v0|v1<<6, b1=v1>>2|v2<<4, b2=v2>>4|v3<<2
    assert o.pack_fp4(np.array([0x1, 0x2], np.uint8)).tolist() == [0x21]
    v = np.array([0b101010, 0b110011, 0b000111, 0b100001], np.uint8)
    b = o.pack_fp6(v)
    assert b.tolist() == [(v[0] | (v[1] << 6)) & 0xFF, ((v[1] >> 2) | (v[2] << 4)) & 0xFF, ((v[2] >> 4) | (v[3] << 2)) & 0xFF]
    stream = sum(int(x) << (6 * i) for i, x in enumerate(v))
    assert [stream >> (8 * i) & 0xFF for i in range(3)] == b.tolist()


@pytest.mark.parametrize("el", ELS)
def test_scale_rule_integer_form_equals_literal_float_formula(el):
    """every positive bf16: smallest e with FMAX*2^e >= amax == (int)ceil(log2f(amax/FMAX))."""
    L = c.lib()
    pos = o.bf16_to_f32(np.arange(0, 0x7F80, dtype=np.uint16))
    e = o.scale_exponent(pos, el)
    fm = o.FORMATS[el]["fmax"]
    for i in range(len(pos)):
        assert L.mxo_scale_exponent_literal(float(pos[i]), c.FMT[el]) == e[i], (el, pos[i])
    nz = pos > 0
    ee = e[nz].astype(np.float64)
    unclamped = ee > -127
    assert np.all(fm * np.exp2(ee[unclamped]) >= pos[nz][unclamped])           # never clips
    assert np.all(fm * np.exp2(ee[unclamped] - 1) < pos[nz][unclamped])         # and is the smallest such e
    assert e[0] == -1                                                          # zero block -> 0.5 -> byte 126


def test_sf_layout():
    for rows, kseg in ((130, 4096), (256, 128), (1, 1024), (128, 14336 - 7168)):
        offs = o.sf_valid_offsets(rows, kseg)
        assert len(np.unique(offs)) == rows * kseg // 32                       # injective
        assert offs.max() < o.sf_size_w(rows, kseg) <= o.sf_size_x(rows, kseg)
    # the 4 block scales of one row for one 128-K slab are 4 consecutive bytes; rows r and r+32 are 4 bytes apart
    assert [int(o.sf_offset(5, j, 512)) for j in range(4)] == [80, 81, 82, 83]
    assert int(o.sf_offset(37, 0, 512)) == 84 and int(o.sf_offset(5, 4, 512)) == 80 + 512
    assert int(o.sf_offset(128, 0, 512)) == 4 * 512
    L = c.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        r, j, kseg = int(rng.integers(0, 5000)), int(rng.integers(0, 64)), 2048
        assert L.mxo_sf_offset(r, j, kseg) == int(o.sf_offset(r, j, kseg))
    assert o.sf_size_x(128, 4096) == 256 * 128 and o.sf_size_x(130, 4096) == 256 * 128   # bindings.cpp:120
    assert o.sf_size_w(128, 4096) == 128 * 128


@pytest.mark.parametrize("mode", ("x", "w", "w4"))
@pytest.mark.parametrize("rows,k,split", [(5, 512, (128, 128, 256)), (130, 1024, (0, 1024, 0)), (33, 1152, (384, 0, 768))])
def test_quantizer_numpy_equals_c(mode, rows, k, split):
    x = lcg.bf16_normalish(rows * k, (rows, k), exp_spread=9)
    idx = lcg.permutation(k, k)
    for a, b in zip(o.reorder_quantize(x, idx, *split, mode), c.reorder_quantize(x, idx, *split, mode)):
        assert np.array_equal(a, b)


def test_quantizer_properties():
    rows, k = 16, 512
    x = lcg.bf16_normalish(5, (rows, k), exp_spread=6)
    idx = lcg.permutation(6, k)
    n, s, ob, sfn, sfs, sfo = o.reorder_quantize(x, idx, 128, 128, 256, "x")
    assert n.shape == (rows, 64) and s.shape == (rows, 96) and ob.shape == (rows, 256)
    # dequantised values reproduce the gathered input to within half a quantum of the block
    xr = o.bf16_to_f32(x)[:, idx.astype(np.int64)]
    for seg, sf, fmt, lo, kseg in ((n, sfn, "fp4", 0, 128), (s, sfs, "fp6", 128, 128), (ob, sfo, "fp8", 256, 256)):
        deq = o.dequant_segment(seg, sf, rows, kseg, fmt)
        ref = xr[:, lo:lo + kseg]
        amax = np.abs(ref).reshape(rows, -1, 32).max(-1, keepdims=True)
        mb = o.FORMATS[fmt]["mbits"]
        assert np.all(np.abs(deq - ref).reshape(rows, -1, 32) <= amax * 2.0 ** (-mb) + 1e-30)
    # gather == permute-then-identity
    xp = np.ascontiguousarray(x[:, idx.astype(np.int64)])
    for a, b in zip(o.reorder_quantize(x, idx, 128, 128, 256, "x"),
                    o.reorder_quantize(xp, np.arange(k, dtype=np.int16), 128, 128, 256, "x")):
        assert np.array_equal(a, b)
    for bad in ((100, 156, 256), (128, 128, 128), (-128, 384, 256)):
        with pytest.raises(ValueError):
            o.reorder_quantize(x, idx, *bad, "x")


def test_matmul_mode_detection_and_segment_skip():
    rows, n, k = 8, 16, 256
    x = lcg.bf16_normalish(1, (rows, k)); w = lcg.bf16_normalish(2, (n, k)); idx = np.arange(k, dtype=np.int16)
    qx = o.reorder_quantize(x, idx, 0, 128, 128, "x")
    for mode in ("w", "w4"):
        qw = o.reorder_quantize(w, idx, 0, 128, 128, mode)
        assert o.matmul_shapes(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2])[-1] == mode
        d = mg.mm(qx, qw)
        dc = c.matmul(qx[0], qw[0], qx[1], qw[1], qx[2], qw[2], qx[3], qw[3], qx[4], qw[4], qx[5], qw[5])
        assert o.bf16_ulp_distance(d, dc).max() <= 1
    # all-fp4 split: the two weight modes coincide (bindings.cpp:74 picks matmul_host; same arithmetic)
    q4 = o.reorder_quantize(x, idx, 256, 0, 0, "x")
    assert np.array_equal(mg.mm(q4, o.reorder_quantize(w, idx, 256, 0, 0, "w")),
                          mg.mm(q4, o.reorder_quantize(w, idx, 256, 0, 0, "w4")))
    # quantisation error vs the unquantised product stays at the MX noise level
    ref = o.bf16_to_f32(x).astype(np.float64) @ o.bf16_to_f32(w).astype(np.float64).T
    got = o.bf16_to_f32(mg.mm(qx, o.reorder_quantize(w, idx, 0, 128, 128, "w")))
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 0.08


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_golden_fixture_is_reproduced_by_the_oracle():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    sp = g["g1_special_rows"]
    assert np.array_equal(sp, mg.special_rows(4096))
    xb, idx, wb = mg.g1_inputs(sp)
    for si, split in enumerate(g["splits"].tolist()):
        qx = o.reorder_quantize(xb, idx, *split, "x")
        for n, a in zip(mg.QN, qx):
            assert np.array_equal(_sha(a), g[f"g1_{si}_x{n}_sha"])
            if si == 0:
                assert np.array_equal(a, g[f"g1_{si}_x{n}"])
        for mode in ("w", "w4"):
            qw = o.reorder_quantize(wb, idx, *split, mode)
            for n, a in zip(mg.QN, qw):
                assert np.array_equal(_sha(a), g[f"g2_{si}_{mode}_{n}_sha"])
            assert np.array_equal(mg.mm(qx, qw), g[f"g3_{si}_{mode}_ref"])
            assert np.array_equal(mg.mm(qx, qw, rounding="fused"), g[f"g3_{si}_{mode}_fused"])
    x5, w5, b5, i5 = mg.g5_inputs()
    assert np.array_equal(i5, g["g5_idx"])
    pw = o.qlinear_pack_weight(w5, i5, *g["g5_split"].tolist())
    assert np.array_equal(o.qlinear_forward(x5, i5, *g["g5_split"].tolist(), pw, bias_bits=b5), g["g5_y"])
    x6, w6, i6 = mg.g6_inputs()
    qx = o.reorder_quantize(x6, i6, 0, 0, 1024, "x")
    assert np.array_equal(qx[2], g["g6_xo"]) and np.array_equal(qx[5], g["g6_sfxo"])
    assert np.array_equal(mg.mm(qx, o.reorder_quantize(w6, i6, 0, 0, 1024, "w4")), g["g6_d"])


def test_golden_k14336_and_k5120():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    for k in (14336, 5120):
        x4, w4, i4 = mg.g4_inputs(k)
        split = g[f"g4_{k}_split"].tolist()
        qx = o.reorder_quantize(x4, i4, *split, "x")
        for n, a in zip(mg.QN, qx):
            assert np.array_equal(_sha(a), g[f"g4_{k}_x{n}_sha"])
        assert np.array_equal(mg.mm(qx, o.reorder_quantize(w4, i4, *split, "w4")), g[f"g4_{k}_d"])


def test_direct_quantizers():
    """reorder-free quantizers (activate.cu): empty block -> byte 127, exact scale rule on fp32 maxima, silu values."""
    rows, k = 6, 384
    w = lcg.bf16_normalish(77, (rows, k), exp_spread=5)
    w[0, :32] = 0
    n, s, ob, sfn, sfs, sfo = o.downproj_quantize(w, 128, 128, 128, w4=False)
    assert sfn[int(o.sf_offset(0, 0, 128))] == 127 and n.shape == (rows, 64) and s.shape == (rows, 96) and ob.shape == (rows, 128)
    # bf16 inputs: same bytes as the reorder quantizer with the identity index, except for the empty-block scale
    ref = o.reorder_quantize(w, np.arange(k, dtype=np.int16), 128, 128, 128, "w")
    assert np.array_equal(n, ref[0]) and np.array_equal(s, ref[1]) and np.array_equal(ob, ref[2])
    w4 = o.downproj_quantize(w, 128, 128, 128, w4=True)
    assert w4[1].shape == (rows, 64) and w4[2].shape == (rows, 64)
    a = o.f32_to_bf16(np.array([[0.0, 1.0, -1.0, 8.0] * 32], np.float32))
    b = o.f32_to_bf16(np.ones((1, 128), np.float32))
    v = o.silu_mul(a, b)
    assert abs(v[0, 1] - 0.7310586) < 1e-6 and abs(v[0, 2] + 0.26894143) < 1e-6 and v[0, 0] == 0
    e = o.scale_exponent_f32(np.array([6.0, 6.0000005, 1e-7, 448.0, 3.0], np.float32), "fp4")
    assert e.tolist() == [0, 1, 0, 7, -1]


def test_direct_quantizers_c_restatement_agrees():
    """oracle/mx_oracle.c follows activate.cu's float formulas literally (expf, ceilf(log2f(amax / FMAX))); the numpy oracle uses the
    exact exponent and numpy's exp.  Weights (no exp): byte for byte.  silu(a) * b: the two exp implementations may differ in the
    last ulp, which can move a value across a rounding boundary -- same budget as the GPU test (< 1e-3 of the bytes)."""
    from oracle import c_oracle
    rows, k, split = 40, 1024, (512, 256, 256)
    w = lcg.bf16_normalish(91, (rows, k), exp_spread=6)
    w[3, 64:96] = 0
    for mode, w4 in ((1, False), (2, True)):
        c = c_oracle.direct_quantize(w, None, *split, mode)
        n = o.downproj_quantize(w, *split, w4=w4)
        for i in range(3):
            assert np.array_equal(c[i], n[i]), (mode, i)
            offs = o.sf_valid_offsets(rows, split[i])
            assert np.array_equal(c[3 + i][offs], n[3 + i][offs]), (mode, i)
    a = lcg.bf16_normalish(92, (rows, k), exp_center=128, exp_spread=3)
    b = lcg.bf16_normalish(93, (rows, k), exp_spread=4)
    c = c_oracle.direct_quantize(a, b, *split, 0)
    n = o.activate_quantize(a, b, *split)
    for i in range(3):
        assert (c[i] != n[i]).mean() < 1e-3, i
        offs = o.sf_valid_offsets(rows, split[i])
        assert (c[3 + i][offs] != n[3 + i][offs]).mean() < 1e-3, i


def test_rmsnorm_quantize_oracle_properties():
    """rmsnorm.cu:95-312 restatement: rvar close to the fp64 value; integer_round=False equals reorder_quantize of the
    normalised row; integer_round=True only ever produces integer-valued elements."""
    rng = np.random.default_rng(21)
    rows, k, split = 6, 1024, (512, 256, 256)
    xb = o.f32_to_bf16((rng.standard_normal((rows, k)) * 2).astype(np.float32))
    wb = o.f32_to_bf16((1 + 0.1 * rng.standard_normal(k)).astype(np.float32))
    idx = rng.permutation(k)
    rvar = o.rmsnorm_rvar(xb, 1e-5)
    truth = 1.0 / np.sqrt((o.bf16_to_f32(xb).astype(np.float64) ** 2).mean(1) + 1e-5)
    assert np.all(np.abs(rvar - truth) <= 4e-7 * truth)
    normed = o.f32_to_bf16(((o.bf16_to_f32(xb) * o.bf16_to_f32(wb)[None, :]).astype(np.float32) * rvar[:, None]).astype(np.float32))
    a = o.rmsnorm_quantize(xb, wb, 1e-5, idx, *split, integer_round=False)
    b = o.reorder_quantize(normed, idx, *split, "x")
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    c = o.rmsnorm_quantize(xb, wb, 1e-5, idx, *split)
    vals = o.decode(c[2], "fp8")                     # fp8 segment: element values before the block scale
    assert np.array_equal(vals, np.round(vals))
    vals4 = o.decode(o.unpack_fp4(c[0]), "fp4")
    assert np.array_equal(vals4, np.round(vals4))
    assert all(np.array_equal(p, q) for p, q in zip(a[3:], c[3:]))    # the scales do not depend on the rounding step
