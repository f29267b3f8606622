"""Small-M GEMM (M <= 64) at large N: mm_matmul through the C ABI, events around 300 launches.  BIGN_N / BIGN_M choose the shapes;
MICROMIX_SKINNY_MAX_M=<m> sends M > m to the 64-row tile kernels instead of the weight-streaming kernel (A/B of the dispatch rule)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
K, split = 4096, (2048, 128, 1920)
idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
NS = tuple(int(v) for v in os.environ.get("BIGN_N", "14336,28672").split(","))
MS = tuple(int(v) for v in os.environ.get("BIGN_M", "16,32,48,64").split(","))
for N in NS:
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    del w
    for M in MS:
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
        f = lambda: lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st)
        assert f() == 0
        for _ in range(50): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): f()
        e1.record(); torch.cuda.synchronize()
        print(f"N={N:6d} M={M:3d}: {e0.elapsed_time(e1) / 300 * 1000:6.1f} us", flush=True)
