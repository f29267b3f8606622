import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
g = torch.Generator().manual_seed(0)
M = N = 4096
for (K1, K2) in ((4096, 8192), (4224, 8320), (4352, 8448), (5120, 10240), (3968, 8064)):
    res = []
    for K in (K1, K2):
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
        idx = torch.arange(K, dtype=torch.int16, device=dev)
        split = (0, 0, K)
        b = mixedgemm.reorder_quantize_w4(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        for _ in range(10): f()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1000)
        res.append(min(ts))
    print(f"{tag:14s} K={K1}/{K2}: {res[0]:.1f} / {res[1]:.1f} us, per-slab {(res[1]-res[0])/((K2-K1)/128):.3f} us; TFLOP/s at K1: {2*M*N*K1/res[0]/1e6:.0f}", flush=True)
