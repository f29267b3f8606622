"""Measures the scaled-MFMA issue rate per format pair (register operands, random data)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib
lib = _lib.load_diag(); dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ELS = ("fp4", "fp6", "fp8")
sink = torch.zeros(4, device=dev)
for shape in (32, 16):
    for ea in range(3):
        for eb in range(3):
            seed = rng.integers(0, 256, size=(128, 32), dtype=np.uint16).astype(np.uint8)
            seed[(seed & 0x7F) == 0x7F] = 0x3C
            t = torch.from_numpy(seed.view(np.int32).reshape(128, 8)).to(dev)
            blocks, iters = 256 * 2, 4000
            for wave_mult in (1, 2):
                b = 256 * wave_mult
                lib.mm_diag_mfma_rate(shape, ea, eb, b, 200, t.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.mm_diag_mfma_rate(shape, ea, eb, b, iters, t.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                mnk = 32 * 32 * 64 if shape == 32 else 16 * 16 * 128
                fl = b * 4 * iters * 8 * 2 * mnk
                print(f"shape {shape} A={ELS[ea]} B={ELS[eb]} waves/SIMD={wave_mult}: {fl / ms / 1e9:.0f} TFLOP/s ({ms:.2f} ms)", flush=True)
