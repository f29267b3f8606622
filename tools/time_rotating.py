"""mm_matmul on R weight sets in rotation (weights from HBM once R x bytes exceeds the 256 MiB Infinity Cache), repeated timing blocks:
    python tools/time_rotating.py M N K KN,KS,KO [R=8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4]); split = tuple(int(v) for v in sys.argv[4].split(",")); R = int(sys.argv[5]) if len(sys.argv) > 5 else 8
DECODE = len(sys.argv) > 6 and sys.argv[6] == "decode"      # mm_qlinear_decode (quantize + GEMM in one launch) instead of mm_matmul
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
sets = []
for r in range(R):
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    sets.append(mixedgemm.reorder_quantize_w4(w, idx, *split)); del w
x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
a = mixedgemm.reorder_quantize_x(x, idx, *split)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
calls = [[pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])] for b in sets]
dcalls = [[pp(t) for t in b] for b in sets]
def loop():
    if DECODE:
        for p in dcalls: lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *p, M, N, *split, 1, 0, None, out.data_ptr(), st)
    else:
        for p in calls: lib.mm_matmul(*p, M, N, *split, 1, 0, None, out.data_ptr(), st)
print("mm_qlinear_decode" if DECODE else lib.mm_matmul_describe(M, N, *split, 1, 0, 0).decode())
res = []
for blk in range(8):
    for _ in range(5): loop()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): loop()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / (50 * R) * 1000)
wb = sum(t.numel() for t in sets[0])
print(f"M={M} N={N} K={K} {split} x {R} weight sets: us per launch by block: " + " ".join(f"{v:.1f}" for v in res) + f"   ({wb / min(res) / 1e6:.2f} TB/s at best)")
