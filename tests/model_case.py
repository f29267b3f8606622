"""Shared machinery of the full-shape parity tests (BASELINE.json configs 2-5: Llama-3-8B, Qwen2.5-14B, Mixtral-8x7B).

At these sizes the oracle cannot run on every row in seconds, so a case is checked like this:
  * operands are generated on the GPU (torch generator, fixed seed); only SAMPLED rows travel to the host;
  * the GPU quantizer's bytes for the sampled rows (activations AND weights) must equal the oracle's, byte for byte --
    that anchors the packed operands the GEMM consumes to the oracle;
  * the GEMM output rows of the sample are compared with the oracle GEMM on the oracle-quantized activations and the
    (anchored) packed weights, with the tolerance and the asserted ulp statistics of tests/gemm_check.py.
"""
import numpy as np

from conftest import bits_from_t, u8
from gemm_check import check_gemm
from oracle import mx_oracle as o


def gen_bf16(dev, rows, k, seed, kind="x"):
    """activations ~ N(0,1) with 1 % outlier channels x20 (the bench distribution); weights ~ N(0, 0.02)."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.randn((rows, k), generator=g, device=dev, dtype=torch.float32)
    if kind == "x":
        cols = torch.randperm(k, generator=g, device=dev)[: max(1, k // 100)]
        t[:, cols] *= 20.0
    else:
        t *= 0.02
    return t.to(torch.bfloat16)


def gen_index(dev, k, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(k, generator=g).to(torch.int16).to(dev)


def sample(rng, total, count, always=()):
    count = min(count, total)
    rows = set(int(r) for r in rng.choice(total, count, replace=False))
    rows.update(r for r in always if 0 <= r < total)
    return np.array(sorted(rows), dtype=np.int64)


def tile_positions(rng, total):
    """One row for every position a token row can take inside the tiled kernels, in every 256-row tile of [0, total): the tile
    row (256 rows), the wave row wm (64 rows of it), the 32-row MFMA tile of that wave, and the lane half (lanes 0-31 / 32-63 hold
    rows 8g + 0..3 / 8g + 4..7 of an MFMA tile's accumulator).  16 rows per 256-row tile -- 256 at M = 4096 -- with the register
    group g and the register i drawn at random.  The narrower tiles (128 / 64 rows per workgroup, 32 per wave) cut the same 256
    rows into sub-ranges of these positions, so one sample serves every tile kernel.  A fault confined to one (tile row, wave,
    MFMA tile, lane half) cannot slip between the sampled rows, which a uniform draw of a few dozen rows allows."""
    rows = []
    for t in range((total + 255) // 256):
        for wm in range(4):
            for tm in range(2):
                for hi in range(2):
                    r = 256 * t + 64 * wm + 32 * tm + 8 * int(rng.integers(4)) + 4 * hi + int(rng.integers(4))
                    if r < total:
                        rows.append(r)
    return np.array(sorted(set(rows)), dtype=np.int64)


def sample_rows(rng, total, count, always=()):
    """the token rows a full-shape test compares with the oracle: `count` random rows, the `always` rows, and -- from the sizes
    that run on the tiled kernels on -- every tile position (tile_positions)"""
    rows = set(int(r) for r in sample(rng, total, count, always))
    if total > 64:
        rows.update(int(r) for r in tile_positions(rng, total))
    return np.array(sorted(rows), dtype=np.int64)


def assert_rows_match_oracle(q_dev, rows, q_ref, widths, label):
    """q_dev: the GPU quantizer's 6-tuple for ALL rows; q_ref: the oracle's 6-tuple for the sampled `rows` only."""
    import torch
    ridx = torch.from_numpy(rows).to(q_dev[0].device)
    for i in range(3):
        if widths[i] == 0:
            continue
        got = u8(q_dev[i][ridx])
        assert np.array_equal(got, q_ref[i]), f"{label}: packed segment {i} differs from the oracle on the sampled rows"
        sf = u8(q_dev[3 + i])
        j = np.arange(widths[i] // 32)[None, :]
        got_sf = sf[o.sf_offset(rows[:, None], j, widths[i])]
        want_sf = q_ref[3 + i][o.sf_offset(np.arange(len(rows))[:, None], j, widths[i])]
        assert np.array_equal(got_sf, want_sf), f"{label}: scale bytes of segment {i} differ from the oracle"


class PackedWeight:
    """a weight [N, K] packed on the GPU, anchored to the oracle on sampled rows, with its dequantised form cached for the
    oracle GEMMs of several activation batches"""

    def __init__(self, dev, n, k, split, seed, wmode="w4", rng=None, index=None, w=None):
        from micromix_amd import mixedgemm
        self.n, self.k, self.split, self.wmode = n, k, split, wmode
        self.w = gen_bf16(dev, n, k, seed, "w") if w is None else w
        self.index = gen_index(dev, k, seed + 1) if index is None else index
        fn = mixedgemm.reorder_quantize_w4 if wmode == "w4" else mixedgemm.reorder_quantize_w
        self.packed = fn(self.w, self.index, *split)
        rng = rng or np.random.default_rng(seed)
        rows = sample(rng, n, 24, always=(0, 127, 128, n - 1))
        import torch
        ref = o.reorder_quantize(bits_from_t(self.w[torch.from_numpy(rows).to(dev)]), u8(self.index), *split, wmode)
        assert_rows_match_oracle(self.packed, rows, ref, split, f"W {n}x{k} {split} {wmode}")
        self.host = [u8(t) for t in self.packed]
        self.deq = o.dequant_operand(self.host, "w", wmode)


def check_rows(d_dev, x_dev, qx_dev, pw, rows, rounding="reference", label="", bias=None, strict=True):
    """sampled rows of the GPU result d_dev [M, N] against the oracle chain on the same rows"""
    import torch
    ridx = torch.from_numpy(rows).to(x_dev.device)
    ref_q = o.reorder_quantize(bits_from_t(x_dev[ridx]), u8(pw.index), *pw.split, "x")
    if qx_dev is not None:
        assert_rows_match_oracle(qx_dev, rows, ref_q, pw.split, f"{label} X")
    return check_gemm(bits_from_t(d_dev[ridx]), ref_q, pw.host, rounding, label=label, strict=strict, wdeq=pw.deq,
                      bias_bits=bits_from_t(bias) if bias is not None else None)
