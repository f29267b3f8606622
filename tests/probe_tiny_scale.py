"""The MX converters' scale operand with the bit patterns 0, 2^-127 and 2^-130 (fp32 denormals) against the oracle's encoder of
2^127 * x: the hardware reads the operand as E8M0 (exponent field only).  python tests/probe_tiny_scale.py   (under tests/ because it
asks the oracle; tests/test_hw_gpu.py pins the same fact in the suite)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from micromix_amd import _lib
from oracle import mx_oracle as o
import hw_layout as hl
from conftest import t_from_bits
dev = torch.device("cuda:0")
lib = _lib.load_diag()
allb = np.arange(65536, dtype=np.uint16)
src = allb[np.isfinite(o.bf16_to_f32(allb))]
src = src[: len(src) // 32 * 32]
tsrc = t_from_bits(src, dev)
x = o.bf16_to_f32(src).astype(np.float64)
for el in hl.ELS:
    fm = o.FORMATS[el]["fmax"]
    for name, sc in (("zero bits", 0.0), ("2^-127 denormal", 2.0 ** -127), ("2^-126", 2.0 ** -126), ("2^-130 denormal", 2.0**-130)):
        out = torch.zeros(len(src), dtype=torch.uint8, device=dev)
        assert lib.mm_diag_hw_convert(tsrc.data_ptr(), len(src), float(sc), hl.ELS.index(el), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for assumed in (-127, -126):
            scaled = x * 2.0 ** (-assumed)
            inr = np.abs(scaled) <= fm
            want = o.encode(np.clip(scaled, -2 * fm, 2 * fm).astype(np.float32), el)
            bad = (got[inr] != want[inr]).sum()
            print(f"{el} scale={name:16s} assumed 2^{assumed}: in range {inr.sum()} mismatches {bad}", flush=True)
