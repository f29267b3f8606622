"""Deterministic input generator for the golden fixtures: a 64-bit LCG in pure integer numpy
arithmetic (identical on every numpy version / platform), shaped into bf16 bit patterns."""
import numpy as np

_A = np.uint64(6364136223846793005)
_C = np.uint64(1442695040888963407)


def lcg_u32(seed: int, n: int) -> np.ndarray:
    """n pseudo-random uint32 values (high halves of a Knuth MMIX LCG)."""
    out = np.empty(n, dtype=np.uint32)
    # vectorised by jumping: state_i = A^i * s + C * (A^i - 1)/(A - 1); do it block-wise instead
    s = np.uint64(seed)
    block = 1 << 16
    with np.errstate(over="ignore"):
        # precompute per-lane multipliers for one block
        mul = np.empty(block, dtype=np.uint64)
        add = np.empty(block, dtype=np.uint64)
        m, a = np.uint64(1), np.uint64(0)
        for i in range(block):
            m = m * _A
            a = a * _A + _C
            mul[i], add[i] = m, a
        pos = 0
        while pos < n:
            st = mul * s + add
            take = min(block, n - pos)
            out[pos:pos + take] = (st[:take] >> np.uint64(32)).astype(np.uint32)
            s = st[block - 1]
            pos += take
    return out


def bf16_normalish(seed: int, shape, exp_center=127, exp_spread=4) -> np.ndarray:
    """bf16 bit patterns with random sign, 7 random mantissa bits and an exponent field drawn
    from exp_center + triangular(-spread..spread): a bell-ish magnitude distribution, no inf/nan."""
    n = int(np.prod(shape))
    r = lcg_u32(seed, n)
    sign = (r >> np.uint32(31)) & np.uint32(1)
    mant = (r >> np.uint32(8)) & np.uint32(0x7F)
    e1 = (r >> np.uint32(16)) % np.uint32(exp_spread + 1)
    e2 = (r >> np.uint32(24)) % np.uint32(exp_spread + 1)
    exp = (np.int64(exp_center) + e1.astype(np.int64) - e2.astype(np.int64)).clip(1, 254).astype(np.uint32)
    return ((sign << np.uint32(15)) | (exp << np.uint32(7)) | mant).astype(np.uint16).reshape(shape)


def permutation(seed: int, n: int) -> np.ndarray:
    """a permutation of range(n) as int16 (stable argsort of LCG keys)."""
    return np.argsort(lcg_u32(seed, n), kind="stable").astype(np.int16)
