"""Power-capped (DVFS-settled) scaled-MFMA rate of the two instruction shapes for fp8(A) x fp4(B): ~0.5 s of back-to-back launches
each, register operands, random data.  A higher sustained rate at the power cap = fewer joules per flop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib
lib = _lib.load_diag(); dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
sink = torch.zeros(4, device=dev)
seed = rng.integers(0, 256, size=(128, 32), dtype=np.uint16).astype(np.uint8)
seed[(seed & 0x7F) == 0x7F] = 0x3C
t = torch.from_numpy(seed.view(np.int32).reshape(128, 8)).to(dev)
st = torch.cuda.current_stream().cuda_stream
for ea, eb, name in ((2, 0, "fp8 x fp4"), (0, 0, "fp4 x fp4"), (2, 2, "fp8 x fp8")):
    for shape in (32, 16):
        iters, b = 4000, 512
        mnk = 32 * 32 * 64 if shape == 32 else 16 * 16 * 128
        fl = b * 4 * iters * 8 * 2 * mnk
        for _ in range(60): lib.mm_diag_mfma_rate(shape, ea, eb, b, iters, t.data_ptr(), sink.data_ptr(), st)   # settle
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.mm_diag_mfma_rate(shape, ea, eb, b, iters, t.data_ptr(), sink.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name}  {shape}x{shape}x{64 if shape == 32 else 128}: {fl / ms / 1e9:7.0f} TFLOP/s sustained ({ms:.2f} ms per launch)", flush=True)
