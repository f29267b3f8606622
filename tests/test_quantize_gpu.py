"""GPU parity tests of mixedgemm.reorder_quantize_{x,w,w4} (through the C ABI) against the oracle:
byte-for-byte equality of the packed outputs and of every scale byte of a real row."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import make_inputs, t_from_bits, u8
from micromix_amd import mixedgemm
from oracle import mx_oracle as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import lcg  # noqa: E402
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu
FN = {"x": mixedgemm.reorder_quantize_x, "w": mixedgemm.reorder_quantize_w, "w4": mixedgemm.reorder_quantize_w4}


def gpu_quant(dev, xb, idx, split, mode):
    import torch
    out = FN[mode](t_from_bits(xb, dev), torch.from_numpy(np.ascontiguousarray(idx)).to(dev), *split)
    torch.cuda.synchronize()
    return [u8(t) for t in out]


def assert_quant_equal(got, want, rows, split, label=""):
    for i, (g, w) in enumerate(zip(got, want)):
        if i < 3:
            assert g.shape == w.shape and np.array_equal(g, w), f"{label}: packed segment {i} differs"
        else:
            assert g.shape == w.shape, f"{label}: SF tensor {i - 3} has shape {g.shape}, want {w.shape}"
            offs = o.sf_valid_offsets(rows, split[i - 3])
            assert np.array_equal(g[offs], w[offs]), f"{label}: scale bytes of segment {i - 3} differ"


CASES = [
    (1, 128, (128, 0, 0)), (1, 128, (0, 128, 0)), (1, 128, (0, 0, 128)),        # smallest legal K
    (3, 384, (128, 128, 128)),
    (127, 4096, (2048, 128, 1920)), (128, 4096, (2048, 0, 2048)), (129, 4096, (3072, 896, 128)),
    (257, 3072, (1024, 1024, 1024)), (64, 3584, (3584, 0, 0)),
    (40, 5120, (4096, 512, 512)), (24, 8192, (0, 0, 8192)), (17, 11008 + 128, (10112, 896, 128)),
    (9, 13824, (12288, 1024, 512)), (16, 14336, (7168, 512, 6656)), (5, 18944, (12544, 3200, 3200)),
    (3, 32768, (16384, 8192, 8192)),                                            # the largest K the int16 index allows (64 KiB of LDS)
]


@pytest.mark.parametrize("mode", ("x", "w", "w4"))
@pytest.mark.parametrize("rows,k,split", CASES)
def test_quantize_matches_oracle(dev, mode, rows, k, split):
    rng = np.random.default_rng(rows * 131 + k)
    xb = make_inputs(rng, rows, k, "normal" if mode == "x" else "weight")
    idx = rng.permutation(k).astype(np.int16)
    assert_quant_equal(gpu_quant(dev, xb, idx, split, mode), o.reorder_quantize(xb, idx, *split, mode), rows, split,
                       f"{mode} {rows}x{k} {split}")


def test_golden_edge_rows_and_digests(dev):
    """committed fixture: edge-case rows (zero groups, -0, amax == FMAX*2^e, RNE ties, subnormals, 2^+-100)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    xb, idx, wb = mg.g1_inputs(g["g1_special_rows"])
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    for si, split in enumerate(g["splits"].tolist()):
        got = gpu_quant(dev, xb, idx, split, "x")
        for i, n in enumerate(mg.QN):
            if i < 3:
                assert np.array_equal(sha(got[i]), g[f"g1_{si}_x{n}_sha"]), (split, n)
            if si == 0:
                want = g[f"g1_0_x{n}"]
                if i < 3:
                    assert np.array_equal(got[i], want)
                else:
                    offs = o.sf_valid_offsets(130, split[i - 3])
                    assert np.array_equal(got[i][offs], want[offs])
        for mode in ("w", "w4"):
            gw = gpu_quant(dev, wb, idx, split, mode)
            for i, n in enumerate(mg.QN):   # N = 128: no SF padding, whole tensors comparable
                assert np.array_equal(sha(gw[i]), g[f"g2_{si}_{mode}_{n}_sha"]), (split, mode, n)
    for k in (14336, 5120):
        x4, _, i4 = mg.g4_inputs(k)
        got = gpu_quant(dev, x4, i4, g[f"g4_{k}_split"].tolist(), "x")
        for i, n in enumerate(mg.QN[:3]):
            assert np.array_equal(sha(got[i]), g[f"g4_{k}_x{n}_sha"])
    x6, _, i6 = mg.g6_inputs()
    got = gpu_quant(dev, x6, i6, (0, 0, 1024), "x")
    assert np.array_equal(got[2], g["g6_xo"])


def test_special_index_patterns(dev):
    rng = np.random.default_rng(3)
    k = 1024
    xb = make_inputs(rng, 33, k)
    for idx in (np.arange(k), np.arange(k)[::-1], np.roll(np.arange(k), 17), np.argsort(np.abs(o.bf16_to_f32(xb)).mean(0))):
        idx = np.ascontiguousarray(idx).astype(np.int16)
        assert_quant_equal(gpu_quant(dev, xb, idx, (512, 128, 384), "x"), o.reorder_quantize(xb, idx, 512, 128, 384, "x"),
                           33, (512, 128, 384))


def test_extreme_values(dev):
    """every finite bf16 bit pattern appears; zero blocks give scale byte 126; huge/tiny blocks stay in range."""
    allb = np.arange(65536, dtype=np.uint16)
    fin = allb[np.isfinite(o.bf16_to_f32(allb))]
    k = 1024
    rows = (len(fin) + k - 1) // k
    xb = np.resize(fin, (rows, k))
    xb[0, :32] = 0
    idx = lcg.permutation(11, k)
    for mode in ("x", "w4"):
        assert_quant_equal(gpu_quant(dev, xb, idx, (384, 256, 384), mode), o.reorder_quantize(xb, idx, 384, 256, 384, mode),
                           rows, (384, 256, 384), mode)


def test_full_size_properties(dev):
    """BASELINE size (4096 x 4096): oracle on a row sample + determinism + gather equivalence."""
    import torch
    rng = np.random.default_rng(0)
    M = K = 4096
    xb = make_inputs(rng, M, K)
    idx = rng.permutation(K).astype(np.int16)
    split = (2048, 128, 1920)
    x = t_from_bits(xb, dev)
    tidx = torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    b = mixedgemm.reorder_quantize_x(x, tidx, *split)
    ident = torch.arange(K, dtype=torch.int16, device=dev)
    c = mixedgemm.reorder_quantize_x(x[:, tidx.long()].contiguous(), ident, *split)
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(a[i], b[i]) and torch.equal(a[i], c[i])
    rows = np.sort(rng.choice(M, 64, replace=False))
    want = o.reorder_quantize(xb[rows], idx, *split, "x")
    for i in range(3):
        assert np.array_equal(u8(a[i])[rows], want[i])
        r = np.arange(len(rows))[:, None]
        j = np.arange(split[i] // 32)[None, :]
        assert np.array_equal(u8(a[3 + i])[o.sf_offset(rows[:, None], j, split[i])], want[3 + i][o.sf_offset(r, j, split[i])])


def test_errors_and_empty(dev):
    import torch
    x = torch.zeros((4, 256), dtype=torch.bfloat16, device=dev)
    idx = torch.arange(256, dtype=torch.int16, device=dev)
    with pytest.raises(RuntimeError, match="Value error in run_reorder_quantize_x"):
        mixedgemm.reorder_quantize_x(x, idx, 100, 28, 128)
    with pytest.raises(RuntimeError, match="Value error in run_reorder_quantize_w4"):
        mixedgemm.reorder_quantize_w4(x, idx, 128, 128, 128)
    with pytest.raises(RuntimeError, match="contiguous"):
        mixedgemm.reorder_quantize_x(torch.zeros((256, 4), dtype=torch.bfloat16, device=dev).t(), idx, 256, 0, 0)
    with pytest.raises(TypeError):
        mixedgemm.reorder_quantize_x(x, idx.int(), 256, 0, 0)
    out = mixedgemm.reorder_quantize_x(x[:0], idx, 128, 0, 128)       # zero rows
    assert out[0].shape == (0, 64) and out[2].shape == (0, 128) and out[3].numel() == 128 * 4
    out = mixedgemm.reorder_quantize_x(x, idx, 0, 0, 256)             # zero-width segments
    assert out[0].shape == (4, 0) and out[1].shape == (4, 0) and out[3].numel() == 0
