"""GPU time of the quantizer kernel alone: direct C-ABI calls on preallocated buffers (no allocation in the loop)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
if os.environ.get("QUANT_IDX") == "identity":      # the worst case: lane g reads element 32 g + i, a 64-byte stride = 4 banks for 32 lanes
    idx = torch.arange(idx.numel(), dtype=torch.int16, device=dev)
    tag += "/identity-index"
if os.environ.get("QUANT_IDX") == "transpose":     # a conflict-free gather (group g, element i <- column 128 i + g): the bound for any fix of the bank conflicts
    n = idx.numel()
    idx = (torch.arange(32).view(1, 32) * (n // 32) + torch.arange(n // 32).view(-1, 1)).reshape(-1).to(torch.int16).to(dev)
    tag += "/transpose-index"
st = torch.cuda.current_stream().cuda_stream
def run(src, rows, K, split, mode):
    KN, KS, KO = split
    w4 = mode == 1
    u8 = dict(dtype=torch.uint8, device=dev)
    o = [torch.empty((rows, KN // 2), **u8), torch.empty((rows, KS // 2 if w4 else KS // 4 * 3), **u8), torch.empty((rows, KO // 2 if w4 else KO), **u8)]
    sf = [torch.empty(((rows // 128 + 1) * 128 * (k // 32),), **u8) for k in split]
    p = lambda t: t.data_ptr() if t.numel() else None
    f = lambda: lib.mm_reorder_quantize(src.data_ptr(), rows, K, idx.data_ptr(), KN, KS, KO, mode, p(o[0]), p(o[1]), p(o[2]), p(sf[0]), p(sf[1]), p(sf[2]), st)
    for _ in range(20): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 100 * 1000)
    return min(ts), sorted(ts)[3]
for split in ((0, 0, 4096), (4096, 0, 0), (2048, 128, 1920), (0, 4096, 0)):
    mn, med = run(x, 4096, 4096, split, 0)
    out_b = 4096 * (split[0] // 2 + split[1] * 3 // 4 + split[2]) + 4096 * 128
    print(f"{tag:12s} x split={split}: {mn:6.2f} us (median {med:6.2f})  {(2*4096*4096 + out_b)/mn/1e6:6.2f} TB/s", flush=True)
mn, med = run(w, 4096, 4096, (0, 0, 4096), 1)
print(f"{tag:12s} w4 split=(0,0,4096): {mn:6.2f} us (median {med:6.2f})", flush=True)
