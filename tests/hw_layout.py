"""Builders for the single-MFMA hardware checks (used by tests/test_hw_gpu.py and tools/gpu_probe.py).

Hypothesis under test (what micromix_amd/csrc/mx_gemm.hip relies on):
  * 32x32x64: lane l holds row/col (l % 32), h = l // 32; 16x16x128: row/col (l % 16), h = l // 16;
  * fp4 / fp6: the lane's registers hold the 32 consecutive K elements of block h, densely packed
    in K order, LSB first (fp4: nibble e, even element in the low nibble; fp6: 6-bit little-endian
    stream);
  * fp8: registers 0-3 hold K = 16h + [0,16), registers 4-7 hold K = Ktot/2 + 16h + [0,16)
    (measured with tools/fp8_probe.py: the contiguous-block hypothesis fails for fp8);
  * the lane's UE8M0 scale is byte `opsel` of its scale register and applies to K block h
    (K in [32h, 32h+32)) of its row/col, for every format;
  * accumulator: 32x32 -> D[(r&3) + 8*(r>>2) + 4*(l>>5)][l & 31] in register r;
                 16x16 -> D[4*(l>>4) + r][l & 15].
"""
import numpy as np

from oracle import mx_oracle as o

ELS = ("fp4", "fp6", "fp8")


def random_codes(rng, rows, k, el):
    f = o.FORMATS[el]
    nbits = 1 + f["ebits"] + f["mbits"]
    c = rng.integers(0, 1 << nbits, size=(rows, k), dtype=np.uint16).astype(np.uint8)
    if el == "fp8":  # avoid NaN codes
        c[(c & 0x7F) == 0x7F] = 0x7E
    return c


def lane_regs(codes, el, shape):
    """codes [R, Ktot] -> int32 [64, 8] register image under the hypothesis."""
    R = shape
    nkb = 64 // R
    regs = np.zeros((64, 32), dtype=np.uint8)
    pack = {"fp4": o.pack_fp4, "fp6": o.pack_fp6, "fp8": lambda c: c}[el]
    ktot = 32 * nkb
    for l in range(64):
        row, kb = l % R, l // R
        assert kb < nkb
        if el == "fp8":
            regs[l, 0:16] = codes[row, 16 * kb:16 * kb + 16]
            regs[l, 16:32] = codes[row, ktot // 2 + 16 * kb:ktot // 2 + 16 * kb + 16]
        else:
            b = pack(codes[row, 32 * kb:32 * kb + 32])
            regs[l, :len(b)] = b
    return regs.view(np.int32).reshape(64, 8)


def scale_regs(rng, scales, shape, opsel):
    """scales [R, nkb] bytes -> int32 [64]; the byte sits at position opsel, other bytes are noise."""
    R = shape
    out = rng.integers(0, 256, size=(64, 4), dtype=np.uint16).astype(np.uint8)
    for l in range(64):
        out[l, opsel] = scales[l % R, l // R]
    return out.view(np.int32).reshape(64)


def expected(codes_a, codes_b, sa, sb, el_a, el_b):
    va = o.decode(codes_a, el_a).astype(np.float64)
    vb = o.decode(codes_b, el_b).astype(np.float64)
    R, ktot = va.shape
    nkb = ktot // 32
    d = np.zeros((R, R))
    for kb in range(nkb):
        a = va[:, 32 * kb:32 * kb + 32] * np.exp2(sa[:, kb].astype(np.float64) - 127)[:, None]
        b = vb[:, 32 * kb:32 * kb + 32] * np.exp2(sb[:, kb].astype(np.float64) - 127)[:, None]
        d += a @ b.T
    return d


def acc_to_matrix(out, shape):
    """out [64, 16|4] float -> D [R, R]."""
    R = shape
    d = np.zeros((R, R), dtype=np.float64)
    for l in range(64):
        for r in range(out.shape[1]):
            if R == 32:
                row, col = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            else:
                row, col = 4 * (l >> 4) + r, l & 15
            d[row, col] = out[l, r]
    return d


def run_case(lib, torch, dev, rng, shape, el_a, el_b, opsel):
    R = shape
    ktot = 64 if R == 32 else 128
    nkb = ktot // 32
    ca, cb = random_codes(rng, R, ktot, el_a), random_codes(rng, R, ktot, el_b)
    sa = rng.integers(120, 135, size=(R, nkb)).astype(np.uint8)
    sb = rng.integers(120, 135, size=(R, nkb)).astype(np.uint8)
    ta = torch.from_numpy(lane_regs(ca, el_a, R)).to(dev)
    tb = torch.from_numpy(lane_regs(cb, el_b, R)).to(dev)
    tsa = torch.from_numpy(scale_regs(rng, sa, R, opsel)).to(dev)
    tsb = torch.from_numpy(scale_regs(rng, sb, R, opsel)).to(dev)
    out = torch.zeros((64, 16 if R == 32 else 4), dtype=torch.float32, device=dev)
    st = lib.mm_diag_mfma(R, ELS.index(el_a), ELS.index(el_b), opsel, ta.data_ptr(), tb.data_ptr(), tsa.data_ptr(),
                          tsb.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    got = acc_to_matrix(out.cpu().numpy(), R)
    exp = expected(ca, cb, sa, sb, el_a, el_b)
    scale = np.abs(exp).max() + 1e-30
    return float(np.abs(got - exp).max() / scale), got, exp
