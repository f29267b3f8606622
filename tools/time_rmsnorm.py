"""Times mm_rmsnorm_quantize against mm_reorder_quantize (direct C-ABI calls) on [M, 4096] inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K = 4096
w = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
for M in (16, 256, 4096):
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    for split in ((0, 0, 4096), (2048, 128, 1920)):
        o = mixedgemm.reorder_quantize_x(x, idx, *split)
        fns = {"rmsnorm_quantize (ref round)": lambda: lib.mm_rmsnorm_quantize(x.data_ptr(), w.data_ptr(), 1e-5, M, K, idx.data_ptr(), *split, 0, *[pp(t) for t in o], st),
               "rmsnorm_quantize (no int round)": lambda: lib.mm_rmsnorm_quantize(x.data_ptr(), w.data_ptr(), 1e-5, M, K, idx.data_ptr(), *split, 1, *[pp(t) for t in o], st),
               "reorder_quantize": lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *split, 0, *[pp(t) for t in o], st)}
        for name, f in fns.items():
            for _ in range(20): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): f()
            e1.record(); torch.cuda.synchronize()
            print(f"M={M:5d} split={split}: {name:32s} {e0.elapsed_time(e1)/200*1000:7.1f} us", flush=True)
