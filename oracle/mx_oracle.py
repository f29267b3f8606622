"""CPU oracle for the MicroMix ``mgemm`` hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy restatement of the reference algorithm.  It is the
*checker* for the HIP kernels; nothing under ``micromix_amd/`` may import it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it.

PARITY UNPINNED: the reference (``/root/reference``) ships no golden vectors or
known-answer tests for this path (``mgemm/test.py`` only prints an MSE,
``mxf4f6f8_bench.cu --validate`` compares two GPU kernels on a Blackwell GPU),
its CUDA sources need nvcc + CUTLASS (``cutlass/`` is an empty submodule) and its
Python layer needs the CUDA extension, so neither can be built or imported
here.  The oracle is therefore pinned only by (a) a second, independent
restatement in C that follows the reference's float formulas literally
(``oracle/mx_oracle.c``), (b) definitional brute-force checks of the OCP element
encodings, (c) torch's own ``float8_e4m3fn`` / ``float8_e8m0fnu`` casts and (d) on
the GPU, the CDNA4 hardware's decode of the same bytes (scaled MFMA) and the
hardware's own MX converters (``v_cvt_scalef32_*``).

What each function follows (paths relative to /root/reference):

* element formats and packers ........ mgemm/src/reorder.cu:17-19,30-33,54-63
                                        mgemm/include/reorder.cuh:37-41
* per-32-group scale (E8M0) .......... mgemm/src/reorder.cu:164-209
* quantize + pack (mixed) ............ mgemm/src/reorder.cu:94-269
* quantize + pack (all-FP4 weights) .. mgemm/src/reorder.cu:271-432
* scale-factor tensor layout ......... mgemm/include/sm120_sf_layout.h:170-173
                                        mgemm/src/reorder.cu:182-185,194-197,205-208
* output allocation shapes ........... mgemm/src/bindings.cpp:116-123,167-172,218-223
* GEMM semantics / rounding order .... mgemm/src/gemm.cu:48-50,75-77
                                        mgemm/src/w4a4.cu:176 (beta=0), w4a6.cu:178 (beta=1)
* matmul shape derivation + mode ..... mgemm/src/bindings.cpp:66-74,87
* QLinearLayer wrapper ............... model/qLinearLayer.py:21-74
* reorder-free quantizers ............ mgemm/src/activate.cu:29,44-202,208-500; bindings.cpp:307-387

bf16 tensors are carried as ``uint16`` bit patterns so the oracle does not
depend on torch.
"""
from __future__ import annotations

import numpy as np

GROUP = 32  # MX block size (reorder.cu: GROUP_SIZE == 32)

# --------------------------------------------------------------------------
# element formats (reorder.cuh:37-41; OCP MX v1.0 encodings, no inf, e4m3 "fn")
# --------------------------------------------------------------------------
FORMATS = {
    # name: exponent bits, mantissa bits, bias, max finite, max finite code (magnitude)
    "fp4": dict(ebits=2, mbits=1, bias=1, fmax=6.0, maxcode=0x7, bits=4),     # E2M1
    "fp6": dict(ebits=3, mbits=2, bias=3, fmax=28.0, maxcode=0x1F, bits=6),   # E3M2
    "fp8": dict(ebits=4, mbits=3, bias=7, fmax=448.0, maxcode=0x7E, bits=8),  # E4M3fn
}
# FMAX = f * 2**q with f in [1,2): used by the integer-exact scale rule.
_FMAX_SPLIT = {"fp4": (0x400000, 2), "fp6": (0x600000, 4), "fp8": (0x600000, 8)}


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    """bf16 bit patterns (uint16) -> float32 (exact)."""
    bits = np.ascontiguousarray(bits).astype(np.uint16, copy=False)
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even (NaN kept quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    out = np.where(nan, (u >> np.uint32(16)) | np.uint32(0x40), rounded)
    return out.astype(np.uint16)


def decode_table(fmt: str) -> np.ndarray:
    """code (incl. sign bit) -> float32 value, by the format definition."""
    f = FORMATS[fmt]
    eb, mb, bias = f["ebits"], f["mbits"], f["bias"]
    n = 1 << (1 + eb + mb)
    tab = np.zeros(n, dtype=np.float32)
    for code in range(n):
        sign = -1.0 if code >> (eb + mb) else 1.0
        e = (code >> mb) & ((1 << eb) - 1)
        m = code & ((1 << mb) - 1)
        if e == 0:
            v = m * 2.0 ** (1 - bias - mb)
        else:
            v = (1.0 + m / (1 << mb)) * 2.0 ** (e - bias)
        tab[code] = sign * v
    if fmt == "fp8":  # e4m3fn: S.1111.111 is NaN
        tab[0x7F] = np.nan
        tab[0xFF] = np.nan
    return tab


_DECODE = {k: decode_table(k) for k in FORMATS}


def encode(x: np.ndarray, fmt: str) -> np.ndarray:
    """float32 -> element code (uint8, sign included), RNE, saturating, sign kept
    on zero.  Restates cutlass::NumericConverter<fpX, float, round_to_nearest>
    as used at reorder.cu:138-140,228-246."""
    f = FORMATS[fmt]
    eb, mb, bias, maxcode = f["ebits"], f["mbits"], f["bias"], f["maxcode"]
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    sign = (u >> np.uint32(31)).astype(np.uint32)
    a = u & np.uint32(0x7FFFFFFF)
    emin = 1 - bias
    shift = 23 - mb
    # normal range of the target
    r = a.astype(np.uint64) + np.uint64((1 << (shift - 1)) - 1) + ((a >> np.uint32(shift)) & np.uint32(1))
    code_n = (r >> np.uint64(shift)).astype(np.int64) - ((127 - bias) << mb)
    # subnormal range of the target: quantum 2**(emin-mb); np.rint is RNE
    af = a.view(np.float32)
    code_s = np.rint(af.astype(np.float64) * 2.0 ** (mb - emin)).astype(np.int64)
    is_sub = a < np.uint32((127 + emin) << 23)
    code = np.where(is_sub, code_s, code_n)
    code = np.minimum(code, maxcode)  # satfinite (also catches inf/nan magnitudes)
    return (code.astype(np.uint32) | (sign << np.uint32(eb + mb))).astype(np.uint8)


def decode(codes: np.ndarray, fmt: str) -> np.ndarray:
    return _DECODE[fmt][np.asarray(codes, dtype=np.uint8)]


# --------------------------------------------------------------------------
# packers (reorder.cu:30-33 PackFp4, :54-63 pack_4_fp6_to_3_bytes)
# --------------------------------------------------------------------------
def pack_fp4(codes: np.ndarray) -> np.ndarray:
    """[..., K] 4-bit codes -> [..., K/2] bytes; element 2i in the LOW nibble."""
    c = np.asarray(codes, dtype=np.uint8)
    lo = c[..., 0::2] & 0xF
    hi = c[..., 1::2] & 0xF
    return (lo | (hi << 4)).astype(np.uint8)


def unpack_fp4(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    out = np.empty(b.shape[:-1] + (b.shape[-1] * 2,), dtype=np.uint8)
    out[..., 0::2] = b & 0xF
    out[..., 1::2] = b >> 4
    return out


def pack_fp6(codes: np.ndarray) -> np.ndarray:
    """[..., K] 6-bit codes -> [..., 3K/4] bytes; dense little-endian bit stream."""
    c = np.asarray(codes, dtype=np.uint8) & 0x3F
    v0, v1, v2, v3 = c[..., 0::4], c[..., 1::4], c[..., 2::4], c[..., 3::4]
    out = np.empty(c.shape[:-1] + (c.shape[-1] // 4 * 3,), dtype=np.uint8)
    out[..., 0::3] = v0 | ((v1 & 0x03) << 6)
    out[..., 1::3] = (v1 >> 2) | ((v2 & 0x0F) << 4)
    out[..., 2::3] = (v2 >> 4) | (v3 << 2)
    return out


def unpack_fp6(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    b0, b1, b2 = b[..., 0::3], b[..., 1::3], b[..., 2::3]
    out = np.empty(b.shape[:-1] + (b.shape[-1] // 3 * 4,), dtype=np.uint8)
    out[..., 0::4] = b0 & 0x3F
    out[..., 1::4] = (b0 >> 6) | ((b1 & 0x0F) << 2)
    out[..., 2::4] = (b1 >> 4) | ((b2 & 0x03) << 4)
    out[..., 3::4] = b2 >> 2
    return out


_PACK = {"fp4": pack_fp4, "fp6": pack_fp6, "fp8": lambda c: np.asarray(c, dtype=np.uint8)}
_UNPACK = {"fp4": unpack_fp4, "fp6": unpack_fp6, "fp8": lambda b: np.asarray(b, dtype=np.uint8)}


def packed_width(fmt: str, k: int) -> int:
    """bytes per row of a K-column segment (bindings.cpp:116-118,167-169,218-220)."""
    return {"fp4": k // 2, "fp6": k // 4 * 3, "fp8": k}[fmt]


# --------------------------------------------------------------------------
# block scale (reorder.cu:175-209)
# --------------------------------------------------------------------------
def scale_exponent(amax: np.ndarray, fmt: str) -> np.ndarray:
    """Per-block scale exponent e (scale = 2**e, SF byte = e + 127).

    Reference: ``scale = bf16(ldexpf(1, (int)ceil(log2(amax / FMAX))))``, and
    ``amax == 0 -> scale = 0.5`` (reorder.cu:179-180).  Integer-exact restatement:
    e is the smallest integer with FMAX * 2**e >= amax.  (amax is a bf16 value,
    so amax/FMAX is never within float rounding of a power of two unless it is
    one exactly; tests/test_oracle.py checks this against the literal float
    formula in oracle/mx_oracle.c for every positive bf16.)  e is clamped to
    [-127, 127] so that it always fits UE8M0; the reference is undefined there
    (its bf16 scale underflows to 0 -> inf/NaN codes), see DESIGN.md.
    """
    fm, q = _FMAX_SPLIT[fmt]
    a = np.ascontiguousarray(amax, dtype=np.float32).view(np.uint32).astype(np.int64)
    exp = a >> 23
    mant = a & 0x7FFFFF
    e = exp - 127 - q + (mant > fm)
    e = np.where(exp == 0, -127, e)          # fp32/bf16 subnormal amax: below FMAX*2**-127
    e = np.clip(e, -127, 127)
    e = np.where(a == 0, -1, e)              # zero block -> scale 0.5 (byte 126)
    return e.astype(np.int32)


# --------------------------------------------------------------------------
# scale-factor tensor layout (sm120_sf_layout.h:170-173)
# --------------------------------------------------------------------------
def sf_offset(r, j, kseg: int):
    """byte offset of the scale of (row r, 32-block j) in a segment with kseg columns.

    Atom ((32,4),(32,4)):((16,4),(0,1)) = 512 B covering 128 rows x 4 blocks,
    atoms tiled K-inner / rows-outer (Step<_2,_1>)."""
    r = np.asarray(r, dtype=np.int64)
    j = np.asarray(j, dtype=np.int64)
    return ((r // 128) * (kseg // 128) * 512 + (j // 4) * 512
            + (r % 32) * 16 + ((r // 32) % 4) * 4 + (j % 4))


def sf_size_x(m: int, kseg: int) -> int:
    """activation SF allocation (bindings.cpp:120-123): (M/128+1)*128 rows."""
    return (m // 128 + 1) * 128 * kseg // 32


def sf_size_w(n: int, kseg: int) -> int:
    """weight SF allocation (bindings.cpp:170-172).  The reference assumes
    N % 128 == 0; rows are padded up to a multiple of 128 otherwise."""
    return (n + 127) // 128 * 128 * kseg // 32


def sf_valid_offsets(rows: int, kseg: int) -> np.ndarray:
    """offsets of every SF byte the quantizer writes (the rest is padding the
    reference leaves uninitialised: torch::empty)."""
    if kseg == 0 or rows == 0:
        return np.zeros((0,), dtype=np.int64)
    r = np.arange(rows)[:, None]
    j = np.arange(kseg // 32)[None, :]
    return sf_offset(r, j, kseg).reshape(-1)


# --------------------------------------------------------------------------
# reorder + quantize (reorder.cu:94-269 mixed, :271-432 all-fp4)
# --------------------------------------------------------------------------
def check_split(k: int, kn: int, ks: int, ko: int):
    if kn < 0 or ks < 0 or ko < 0 or kn + ks + ko != k or k <= 0:
        raise ValueError(f"bad split: KN+KS+KO = {kn}+{ks}+{ko} != K = {k}")
    if kn % 128 or ks % 128 or ko % 128:
        raise ValueError("KN, KS, KO must be multiples of 128 (qLinearLayer.py:40)")


def quantize_segment(v: np.ndarray, fmt: str):
    """v [rows, kseg] float32 (already reordered) -> (packed bytes, exponents [rows, kseg/32])."""
    rows, kseg = v.shape
    g = v.reshape(rows, kseg // GROUP, GROUP)
    amax = np.abs(g).max(axis=-1) if kseg else np.zeros((rows, 0), np.float32)
    e = scale_exponent(amax, fmt)
    # v * 2**-e is exact in fp32 (power-of-two scaling of a bf16 value); computed
    # in fp64 here so that 2**127 factors cannot misbehave, then narrowed exactly.
    q = (g.astype(np.float64) * np.exp2(-e.astype(np.float64))[..., None]).astype(np.float32)
    codes = encode(q.reshape(rows, kseg), fmt)
    return _PACK[fmt](codes), e


def reorder_quantize(x_bits: np.ndarray, idx: np.ndarray, kn: int, ks: int, ko: int,
                     mode: str, sf_fill: int = 0, gather_subset: bool = False):
    """Restates run_reorder_quantize_{x,w,w4}.

    mode: "x"  activations, formats (fp4, fp6, fp8), SF sized sf_size_x
          "w"  weights,     formats (fp4, fp6, fp8), SF sized sf_size_w
          "w4" weights,     formats (fp4, fp4, fp4), SF sized sf_size_w
    gather_subset: idx selects KN+KS+KO <= K columns (a tensor-parallel K-shard; not in the reference).
    Returns (ON, OS, OO, SFN, SFS, SFO); SF padding bytes are ``sf_fill``.
    """
    x_bits = np.asarray(x_bits)
    rows, k = x_bits.shape
    check_split(kn + ks + ko if gather_subset else k, kn, ks, ko)
    idx = np.asarray(idx).astype(np.int64)
    if idx.shape != (kn + ks + ko,) or (len(idx) and (idx.min() < 0 or idx.max() >= k)):
        raise ValueError("reorder_index must have KN+KS+KO entries in [0, K)")
    fmts = ("fp4", "fp4", "fp4") if mode == "w4" else ("fp4", "fp6", "fp8")
    v = bf16_to_f32(x_bits)[:, idx]                      # gather (reorder.cu:155-158)
    outs, sfs = [], []
    col = 0
    for kseg, fmt in zip((kn, ks, ko), fmts):
        seg = v[:, col:col + kseg]
        col += kseg
        packed, e = quantize_segment(seg, fmt)
        size = sf_size_x(rows, kseg) if mode == "x" else sf_size_w(rows, kseg)
        sf = np.full((size,), sf_fill, dtype=np.uint8)
        if kseg:
            r = np.arange(rows)[:, None]
            j = np.arange(kseg // 32)[None, :]
            sf[sf_offset(r, j, kseg)] = (e + 127).astype(np.uint8)
        outs.append(np.ascontiguousarray(packed).reshape(rows, packed_width(fmt, kseg)))
        sfs.append(sf)
    return (*outs, *sfs)


# --------------------------------------------------------------------------
# RMSNorm fused with reorder + quantize (rmsnorm.cu:95-312; bindings.cpp:257-303) -- SURVEY.md section 8f rank 2
# --------------------------------------------------------------------------
def rmsnorm_rvar(x_bits: np.ndarray, eps: float) -> np.ndarray:
    """Reciprocal RMS of every row, fp32, in the reference's summation order (rmsnorm.cu:143-185).

    One partial sum per 32-group thread t of T = K/32: the squares of elements i*K/4 + 8t + j (i = 0..3, j = 0..7)
    added one after the other in fp32 (:147-156, local_sum_p2 :43-50); then the block tree s[t] += s[t + stride] for
    stride = P/2 .. 1 (:159-175).  The reference hard-codes the tree for T = 128 (K = 4096): with T = 160 (K = 5120)
    it drops the last 32 partial sums and with T = 96 it reads past its shared array.  Here the tree is the same one
    over P = next power of two >= T with zero padding, which equals the reference's for K = 4096 and is the evident
    intent for the other K.  rvar = 1 / sqrt(sum / K + eps) with correctly rounded fp32 divide and square root; the
    reference's rsqrt() is a 2-ulp hardware approximation whose bits cannot be pinned without a Blackwell GPU.
    """
    x = bf16_to_f32(np.asarray(x_bits))
    rows, k = x.shape
    t = k // GROUP
    sq = (x * x).astype(np.float32)                      # bf16 x bf16 is exact in fp32
    part = np.zeros((rows, t), np.float32)
    for i in range(4):
        blk = sq[:, i * (k // 4):(i + 1) * (k // 4)].reshape(rows, t, 8)
        for j in range(8):
            part = (part + blk[:, :, j]).astype(np.float32)
    p = 1
    while p < t:
        p *= 2
    s = np.zeros((rows, p), np.float32)
    s[:, :t] = part
    stride = p // 2
    while stride >= 1:
        s[:, :stride] = (s[:, :stride] + s[:, stride:2 * stride]).astype(np.float32)
        stride //= 2
    mean = (s[:, 0] / np.float32(k)).astype(np.float32)
    return (np.float32(1.0) / np.sqrt((mean + np.float32(eps)).astype(np.float32))).astype(np.float32)


def _round_half_away(x: np.ndarray) -> np.ndarray:
    """C roundf(): nearest integer, halves away from zero, the sign of a zero result kept (round(-0.3) = -0.0)."""
    x = np.asarray(x, np.float32)
    return np.copysign(np.floor(np.abs(x) + np.float32(0.5)), x).astype(np.float32)


def rmsnorm_quantize(x_bits: np.ndarray, w_bits: np.ndarray, eps: float, idx: np.ndarray, kn: int, ks: int, ko: int,
                     integer_round: bool = True, sf_fill: int = 0):
    """Restates rmsnorm_bf16_mixed_kernel (rmsnorm.cu:95-312).

    v[i] = bf16((float(x[idx[i]]) * float(w[idx[i]])) * rvar)   (:190-195); per 32-group amax, scale = amax == 0 ? 0.5 :
    2^ceil(log2(amax / FMAX)) (:216-245, as reorder.cu); then -- and this is what the reference does, rmsnorm.cu:262-267 --
    q = convert(bf16(clamp(round(v / scale), -FMAX, FMAX))): the value is rounded to an INTEGER (half away from zero)
    before the element conversion.  ``integer_round=False`` gives the quantizer without that step (identical to
    reorder_quantize applied to the normalised row).
    """
    x_bits = np.asarray(x_bits)
    rows, k = x_bits.shape
    check_split(k, kn, ks, ko)
    idx = np.asarray(idx).astype(np.int64)
    rvar = rmsnorm_rvar(x_bits, eps)
    xv = bf16_to_f32(x_bits)[:, idx]
    wv = bf16_to_f32(np.asarray(w_bits))[idx]
    prod = (xv * wv[None, :]).astype(np.float32)         # exact
    v = bf16_to_f32(f32_to_bf16((prod * rvar[:, None]).astype(np.float32)))
    outs, sfs = [], []
    col = 0
    for kseg, fmt in zip((kn, ks, ko), ("fp4", "fp6", "fp8")):
        seg = v[:, col:col + kseg]
        col += kseg
        g = seg.reshape(rows, kseg // GROUP, GROUP)
        amax = np.abs(g).max(axis=-1) if kseg else np.zeros((rows, 0), np.float32)
        e = scale_exponent(amax, fmt)
        q = (g.astype(np.float64) * np.exp2(-e.astype(np.float64))[..., None]).astype(np.float32)
        if integer_round:
            fmax = np.float32(FORMATS[fmt]["fmax"])
            q = bf16_to_f32(f32_to_bf16(np.clip(_round_half_away(q), -fmax, fmax)))
        packed = _PACK[fmt](encode(q.reshape(rows, kseg), fmt))
        sf = np.full((sf_size_x(rows, kseg),), sf_fill, dtype=np.uint8)
        if kseg:
            r = np.arange(rows)[:, None]
            j = np.arange(kseg // 32)[None, :]
            sf[sf_offset(r, j, kseg)] = (e + 127).astype(np.uint8)
        outs.append(np.ascontiguousarray(packed).reshape(rows, packed_width(fmt, kseg)))
        sfs.append(sf)
    return (*outs, *sfs)


# --------------------------------------------------------------------------
# reorder-free quantizers (activate.cu:44-202 silu(a)*b; :208-500 weights) -- SURVEY.md section 8f rank 1
# --------------------------------------------------------------------------
def scale_exponent_f32(amax: np.ndarray, fmt: str) -> np.ndarray:
    """activate.cu:117-120: scale = amax > 1e-6 ? 2^ceil(log2(amax/FMAX)) : 1.0, with the exponent computed exactly
    (smallest e with FMAX*2^e >= amax) and clamped to [-127, 127]; amax is an arbitrary fp32 here."""
    fm, q = _FMAX_SPLIT[fmt]
    a = np.ascontiguousarray(amax, dtype=np.float32)
    bits = a.view(np.uint32).astype(np.int64)
    exp = bits >> 23
    e = exp - 127 - q + ((bits & 0x7FFFFF) > fm)
    e = np.where(exp == 0, -127, e)
    e = np.clip(e, -127, 127)
    return np.where(a > np.float32(1e-6), e, 0).astype(np.int32)


def silu_mul(a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
    """silu(float(a)) * float(b) in fp32, silu(x) = x / (1 + expf(-x))  (activate.cu:29,101)."""
    a = bf16_to_f32(a_bits)
    b = bf16_to_f32(b_bits)
    with np.errstate(over="ignore"):
        return ((a / (np.float32(1.0) + np.exp(-a, dtype=np.float32))) * b).astype(np.float32)


def direct_quantize(v: np.ndarray, kn: int, ks: int, ko: int, w4: bool = False, sf_fill: int = 0):
    """v [rows, K] fp32 in natural column order -> (ON, OS, OO, SFN, SFS, SFO); SF tensors sized sf_size_x."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    rows, k = v.shape
    check_split(k, kn, ks, ko)
    fmts = ("fp4", "fp4", "fp4") if w4 else ("fp4", "fp6", "fp8")
    outs, sfs = [], []
    col = 0
    for kseg, fmt in zip((kn, ks, ko), fmts):
        g = v[:, col:col + kseg].reshape(rows, kseg // GROUP, GROUP)
        col += kseg
        amax = np.abs(g).max(axis=-1) if kseg else np.zeros((rows, 0), np.float32)
        e = scale_exponent_f32(amax, fmt)
        ec = np.maximum(e, -126)     # the kernel divides by a normal fp32 (2^-127 is subnormal)
        fmax = FORMATS[fmt]["fmax"]
        q = np.clip(g.astype(np.float64) * np.exp2(-ec.astype(np.float64))[..., None], -fmax, fmax).astype(np.float32)
        codes = encode(q.reshape(rows, kseg), fmt)
        sf = np.full((sf_size_x(rows, kseg),), sf_fill, dtype=np.uint8)
        if kseg:
            r = np.arange(rows)[:, None]
            j = np.arange(kseg // 32)[None, :]
            sf[sf_offset(r, j, kseg)] = (e + 127).astype(np.uint8)
        outs.append(np.ascontiguousarray(_PACK[fmt](codes)).reshape(rows, packed_width(fmt, kseg)))
        sfs.append(sf)
    return (*outs, *sfs)


def activate_quantize(a_bits, b_bits, kn, ks, ko):
    return direct_quantize(silu_mul(a_bits, b_bits), kn, ks, ko, w4=False)


def downproj_quantize(w_bits, kn, ks, ko, w4: bool):
    return direct_quantize(bf16_to_f32(w_bits), kn, ks, ko, w4=w4)


# --------------------------------------------------------------------------
# dequantize + GEMM (gemm.cu:26-78; epilogue alpha/beta w4a4.cu:176, w4a6.cu:178)
# --------------------------------------------------------------------------
def dequant_segment(packed: np.ndarray, sf: np.ndarray, rows: int, kseg: int, fmt: str,
                    dtype=np.float32) -> np.ndarray:
    if kseg == 0:
        return np.zeros((rows, 0), dtype=dtype)
    codes = _UNPACK[fmt](np.asarray(packed, dtype=np.uint8).reshape(rows, -1))
    vals = decode(codes, fmt).astype(dtype)
    r = np.arange(rows)[:, None]
    j = np.arange(kseg // 32)[None, :]
    e = np.asarray(sf, dtype=np.uint8)[sf_offset(r, j, kseg)].astype(np.int32) - 127
    scale = np.exp2(e.astype(np.float64)).astype(dtype)
    return (vals.reshape(rows, kseg // 32, 32) * scale[..., None]).reshape(rows, kseg)


def matmul_shapes(an, bn, a_s, bs, ao, bo):
    """(M, N, KN, KS, KO, wmode) exactly as bindings.cpp:66-74 derives them."""
    m, n = an.shape[0], bn.shape[0]
    kn = an.shape[1] * 2
    ks = a_s.shape[1] * 4 // 3
    ko = ao.shape[1]
    same = (a_s.shape[1] == bs.shape[1]) and (ao.shape[1] == bo.shape[1])
    return m, n, kn, ks, ko, ("w" if same else "w4")


def dequant_operand(q, kind: str, wmode: str = "w"):
    """the three segments of a packed operand (as returned by reorder_quantize) as fp64 [rows, Kseg] arrays (None
    for an empty segment).  kind "x": formats (fp4, fp6, fp8); kind "w": the same unless wmode == "w4" (all fp4)."""
    fmts = ("fp4", "fp4", "fp4") if (kind == "w" and wmode == "w4") else ("fp4", "fp6", "fp8")
    rows = q[0].shape[0]
    out = []
    for i, fmt in enumerate(fmts):
        kseg = q[i].shape[1] * 8 // FORMATS[fmt]["bits"]
        out.append(dequant_segment(q[i], q[3 + i], rows, kseg, fmt, np.float64) if kseg else None)
    return out


def matmul(an, bn, a_s, bs, ao, bo, sfan, sfbn, sfas, sfbs, sfao, sfbo,
           rounding: str = "reference", return_f64: bool = False, b_dequant=None, return_parts: bool = False):
    """Three-segment mixed-MX GEMM -> bf16 bits [M, N].

    rounding="reference": D = bf16(acc_N); D = bf16(acc_S + D); D = bf16(acc_O + D)
                          (gemm.cu:48-50 / 75-77: each segment is its own kernel
                          and D round-trips through bf16).
    rounding="fused":     D = bf16(acc_N + acc_S + acc_O), one rounding.
    b_dequant: optional `dequant_operand(B, "w", wmode)` computed earlier (tests that reuse one weight for many
    activations); return_parts: also return the per-segment fp64 products and the dequantised operands.
    """
    m, n, kn, ks, ko, wmode = matmul_shapes(an, bn, a_s, bs, ao, bo)
    bf = ("fp4", "fp4", "fp4") if wmode == "w4" else ("fp4", "fp6", "fp8")
    af = ("fp4", "fp6", "fp8")
    segs = [(an, bn, sfan, sfbn, kn), (a_s, bs, sfas, sfbs, ks), (ao, bo, sfao, sfbo, ko)]
    d = np.zeros((m, n), dtype=np.float32)              # C = torch::zeros (bindings.cpp:72)
    total64 = np.zeros((m, n), dtype=np.float64)
    parts = []
    for i, ((a, b, sfa, sfb, kseg), fa, fb) in enumerate(zip(segs, af, bf)):
        if kseg == 0:
            continue
        a64 = dequant_segment(a, sfa, m, kseg, fa, np.float64)
        b64 = b_dequant[i] if b_dequant is not None else dequant_segment(b, sfb, n, kseg, fb, np.float64)
        acc64 = a64 @ b64.T
        total64 += acc64
        if return_parts:
            parts.append((acc64, a64, b64))
        if rounding == "reference":
            acc = acc64.astype(np.float32)               # fp32 accumulator (w4a4.cu:27)
            d = bf16_to_f32(f32_to_bf16(acc + d))
    if rounding == "fused":
        d = bf16_to_f32(f32_to_bf16(total64.astype(np.float32)))
    out = f32_to_bf16(d)
    if return_parts:
        return out, parts
    return (out, total64) if return_f64 else out


def qlinear_pack_weight(w_bits, idx, p4, p6, p8, wmode="w4"):
    """QLinearLayer.__init__ (qLinearLayer.py:42-50)."""
    return reorder_quantize(w_bits, idx, p4, p6, p8, wmode)


def qlinear_forward(x_bits, idx, p4, p6, p8, packed_w, bias_bits=None, rounding="reference"):
    """QLinearLayer.forward (qLinearLayer.py:58-74) on a [M, K] activation."""
    bn, bs, bo, sfbn, sfbs, sfbo = packed_w
    an, a_s, ao, sfan, sfas, sfao = reorder_quantize(x_bits, idx, p4, p6, p8, "x")
    y = matmul(an, bn, a_s, bs, ao, bo, sfan, sfbn, sfas, sfbs, sfao, sfbo, rounding=rounding)
    if bias_bits is not None:                            # y = y + bias in bf16
        y = f32_to_bf16(bf16_to_f32(y) + bf16_to_f32(np.asarray(bias_bits))[None, :])
    return y


# --------------------------------------------------------------------------
# tolerance helpers shared by the parity tests
# --------------------------------------------------------------------------
def bf16_ulp_distance(a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
    """distance in bf16 ulps (monotone integer mapping of the bit patterns)."""
    def key(u):
        u = np.asarray(u, dtype=np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))
