"""What a launch that ONLY streams the packed weights of a small-M GEMM costs (libmicromix_diag.so, mm_diag_stream_once): every wave
issues all of its 16-byte loads before the first use, 32 rows per 512-thread workgroup, no arithmetic, nothing written.  By access
pattern: 0 = coalesced 1 KiB per wave-instruction, 1 = the weight-streaming kernel's (lane = weight row, 64-byte slab s -> wave s % 8),
2 = the same with whole 128-byte lines per wave.  Back-to-back launches between two events, like bench.py's small_m rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib
lib = _lib.load_diag(); dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
buf = torch.randint(0, 255, (512 << 20,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rows, pitch, what in ((4096, 2048, "q/o  N=4096  K=4096 fp4"), (14336, 2048, "gate N=14336 K=4096 fp4"), (28672, 2048, "gate+up N=28672"),
                          (14336, 1024, "N=14336 K=2048 fp4"), (4096, 4096, "N=4096 K=8192 fp4")):
    for pattern in (0, 1, 2):
        for rotate in (False, True):      # rotate: a different 32 MB window per launch (nothing served by the Infinity Cache)
            n, reps = rows * pitch, 200
            def go(i):
                off = ((i * n) % (448 << 20)) if rotate else 0
                assert lib.mm_diag_stream_once(buf.data_ptr() + off, rows, pitch, pattern, sink.data_ptr(), st) == 0
            for i in range(20): go(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps): go(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"{what:26s} {n / 1e6:6.1f} MB  pattern {pattern}  {'rotating' if rotate else 'same buf'}: {us:6.2f} us per launch = {n / us / 1e6:5.2f} TB/s", flush=True)
