"""Kernel durations (HIP events attached to the dispatch, median of 60 after a settle phase) of shapes with few 64-row tiles under the
plan MICROMIX_SPLIT_SMALL pins (or the library's own rule): python tools/split_small_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K, split = 4096, (2048, 128, 1920)
SHAPES = ((128, 1024), (192, 1024), (256, 1024), (384, 1024), (512, 1024), (128, 2048), (256, 2048), (128, 4096), (256, 4096))
if os.environ.get("SWEEP_SPLIT"):      # e.g. SWEEP_SPLIT=12288,1024,1024 SWEEP_SHAPES=128x4096,256x4096 (down_proj)
    split = tuple(int(v) for v in os.environ["SWEEP_SPLIT"].split(","))
    K = sum(split)
if os.environ.get("SWEEP_SHAPES"):
    SHAPES = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["SWEEP_SHAPES"].split(","))
idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
tag = os.environ.get("MICROMIX_SPLIT_SMALL", "rule")
for M, N in SHAPES:
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split); a = mixedgemm.reorder_quantize_x(x, idx, *split)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(100): f()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for e0, e1 in evs: e0.record(); e1.record()
    torch.cuda.synchronize()
    for e0, e1 in evs:
        lib.mm_diag_set_kernel_events(e0.cuda_event, e1.cuda_event); f()
    lib.mm_diag_set_kernel_events(None, None); torch.cuda.synchronize()
    us = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs])) * 1e3
    need = lib.mm_matmul_workspace_bytes(M, N, *split, 1, 4)
    d = lib.mm_matmul_describe(M, N, *split, 1, 4, max(need, 1 << 28)).decode()
    print(f"{tag:8s} M={M:4d} N={N:5d}: {us:6.2f} us | {d[:95]}", flush=True)
