import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((256, 256, 128), (1024, 1024, 128), (2048, 2048, 128), (4096, 4096, 128), (8192, 8192, 128), (128, 4096, 128), (128, 128, 128)):
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
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1000)
    print(f"{tag:14s} M={M} N={N} K={K}: {min(ts):7.1f} us per launch", flush=True)
