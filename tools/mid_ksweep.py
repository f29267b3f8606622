"""Fixed cost against per-slab cost of a tiled launch at small M: (0,0,K) and (K,0,0) for a K sweep at M (argv[1], default 128),
N = 4096.  Prints the back-to-back launch period and the kernel duration (HIP events attached to the dispatch).
Environment switches as tools/mid_m_sweep.py."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib.mm_matmul_describe.restype = ctypes.c_char_p
ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
print("# " + (" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MICROMIX_")) or "default"))
for K in (128, 512, 1024, 2048, 4096, 8192, 16384):
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    idx = torch.arange(K, dtype=torch.int16, device=dev)
    for split in ((0, 0, K), (K, 0, 0)):
        b = mixedgemm.reorder_quantize_w4(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        pp = lambda t: t.data_ptr() if t.numel() else None
        st = torch.cuda.current_stream().cuda_stream
        ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
        f = lambda: lib.mm_matmul_ws(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        for _ in range(20): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        period = e0.elapsed_time(e1) * 10
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for p, q in evs:
            p.record(); q.record()
        torch.cuda.synchronize()
        for p, q in evs:
            lib.mm_diag_set_kernel_events(p.cuda_event, q.cuda_event)
            f()
        lib.mm_diag_set_kernel_events(None, None)
        torch.cuda.synchronize()
        kern = float(np.median([p.elapsed_time(q) for p, q in evs])) * 1e3
        d = lib.mm_matmul_describe(M, N, *split, 1, 0, ws.numel()).decode()
        print(f"M={M} N={N} K={K:5d} {'fp8' if split[2] else 'fp4'}: period {period:6.1f} us  kernel {kern:6.1f} us | {d[:60]}", flush=True)
