"""BASELINE.json configs[2]: Llama-3-8B projection shapes with mixed p4/p6/p8 splits, 1 MI355X.
Kernel time of mixedgemm.matmul (HIP events, preallocated output) and of the quantizer (direct C-ABI calls).
Splits are the reference's own bench constants scaled to K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [("q/o_proj", 4096, 4096), ("k/v_proj", 1024, 4096), ("gate/up_proj", 14336, 4096), ("down_proj", 4096, 14336)]
SPLITS = {4096: [(0, 0, 4096), (2048, 128, 1920), (3072, 896, 128)], 14336: [(7168, 512, 6656), (12288, 1024, 1024)]}
Ms = [int(a) for a in sys.argv[1:]] or [1, 16, 256, 2048, 4096]
def timed(f, n=50, reps=5):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return min(ts)
ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)   # split-K scratch (mm_matmul_workspace_bytes <= 32 MiB)
print(f"{'layer':13s} {'N':>6s} {'K':>6s} {'M':>5s} {'split':>20s} | {'gemm us':>8s} {'TFLOP/s':>8s} | {'quant us':>8s} | tokens/s (quant+gemm)")
for name, N, K in SHAPES:
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    for split in SPLITS[K]:
        for M in Ms:
            x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
            idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
            b = mixedgemm.reorder_quantize_w4(w, idx, *split)
            a = mixedgemm.reorder_quantize_x(x, idx, *split)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            n = 50 if M >= 256 else 200
            pp = lambda t: t.data_ptr() if t.numel() else None
            st = torch.cuda.current_stream().cuda_stream
            ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
            # direct C-ABI call (3 us of host time) so that small problems are not measured at the Python shim's ~14 us
            tg = timed(lambda: lib.mm_matmul_ws(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st), n)
            tq = timed(lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *split, 0, pp(a[0]), pp(a[1]), pp(a[2]), pp(a[3]), pp(a[4]), pp(a[5]), st), n)
            print(f"{name:13s} {N:6d} {K:6d} {M:5d} {str(split):>20s} | {tg:8.1f} {2*M*N*K/tg/1e6:8.0f} | {tq:8.1f} | {M/(tg+tq)*1e6:,.0f}", flush=True)
