"""GEMM time with 256-row vs 128-row tiles forced (MICROMIX_GEMM_TILE is read once per process, so this script is run twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
tag = os.environ.get("MICROMIX_GEMM_TILE", "auto")
for M, N, K, split in ((2048, 5120, 5120, (2560, 256, 2304)), (2048, 5120, 1280, (640, 0, 640)), (1536, 4096, 4096, (0, 0, 4096)), (2560, 4096, 4096, (0, 0, 4096)),
                       (3072, 4096, 4096, (0, 0, 4096)), (2048, 13824, 1280, (640, 0, 640)), (4096, 5120, 5120, (2560, 256, 2304)), (3072, 5120, 1280, (640, 0, 640))):
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
    f = lambda: lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st)
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    t256 = -(-M // 256) * -(-N // 256)
    print(f"tile={tag:5s} M={M} N={N} K={K}: {us:6.1f} us  {2*M*N*K/us/1e6:6.0f} TFLOP/s   (tiles256={t256}, tiles128={-(-M//128) * -(-N//256)})", flush=True)
