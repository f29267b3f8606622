"""activate_quantize_x (silu(A)*B -> mixed quantize) bandwidth at the Llama-3-8B down_proj input shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
K = 14336
for M, split in ((256, (7168, 512, 6656)), (4096, (7168, 512, 6656)), (4096, (12288, 1024, 1024)), (4096, (0, 0, 14336)), (4096, (14336, 0, 0))):
    a = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    o = mixedgemm.activate_quantize_x(a, b, *split)
    f = lambda: lib.mm_activate_quantize(a.data_ptr(), b.data_ptr(), M, *split, *[pp(t) for t in o], st)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    byts = 2 * M * K * 2 + M * (split[0] // 2 + split[1] * 3 // 4 + split[2]) + M * K // 32
    print(f"{os.path.basename(os.environ.get('MICROMIX_HIP_LIB', 'default')):12s} activate_quantize_x M={M:5d} K={K} {split}: {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s", flush=True)
