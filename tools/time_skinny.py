"""Direct C-ABI timing of mm_matmul at small M (no Python between launches beyond the ctypes call), Llama-3-8B shapes, w4 weights:
    python tools/time_skinny.py [M ...]            (default 1 16 32 64; MICROMIX_HIP_LIB selects a variant build)
us per launch over 300 back-to-back launches, median of 5; weight bytes / time as TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
def timed(fn, n=300):
    ts = []
    for _ in range(5):
        for _ in range(30): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return sorted(ts)[2]
Ms = [int(v) for v in sys.argv[1:]] or [1, 16, 32, 64]
for name, N, K, split in (("q/o", 4096, 4096, (2048, 128, 1920)), ("k/v", 1024, 4096, (2048, 128, 1920)), ("gate/up", 14336, 4096, (2048, 128, 1920)),
                          ("gate+up", 28672, 4096, (2048, 128, 1920)), ("down", 4096, 14336, (12288, 1024, 1024)), ("q/o fp8", 4096, 4096, (0, 0, 4096))):
    if os.environ.get("CASES") and name not in os.environ["CASES"].split(","): continue
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    wbytes = sum(t.numel() for t in b)
    row = []
    for M in Ms:
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
        wsb = lib.mm_matmul_workspace_bytes(M, N, *split, 1, 4)        # the stream's split-K workspace where the plan wants one (tickets zeroed)
        ws = mixedgemm.split_workspace(dev, wsb) if wsb else None
        f = (lambda: lib.mm_matmul_ws(*ptrs, M, N, *split, 1, 4, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st)) if ws is not None else \
            (lambda: lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st))
        assert f() == 0
        us = timed(f)
        row.append(f"M={M}: {us:6.2f} us {wbytes / us / 1e6:4.2f} TB/s")
    print(f"[{tag}] {name:8s} N={N:5d} K={K:5d} {split}: " + "   ".join(row), flush=True)
