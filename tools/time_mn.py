import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
g = torch.Generator().manual_seed(0)
for (M, N) in ((256, 256), (1024, 1024), (2048, 2048), (4096, 2048), (4096, 4096), (8192, 4096)):
    res = []
    for K in (4096, 8192):
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
    print(f"{tag:14s} M={M} N={N}: K4096 {res[0]:.1f} us, K8192 {res[1]:.1f} us, per-slab {(res[1]-res[0])/32:.3f} us, tiles {((M+255)//256)*((N+255)//256)}", flush=True)
