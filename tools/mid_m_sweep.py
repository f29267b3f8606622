"""GEMM time against M between the skinny kernels (M <= 64) and the full 256-row tiles: which kernel plan_tiles picks
(mm_matmul_describe) and what it costs.  python tools/mid_m_sweep.py [M ...]; the environment switches of mx_gemm256.hip
(MICROMIX_SPLITK=0, MICROMIX_GEMM_TILE=64|128|256) give the alternatives, one process each."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
CASES = [("q/o", 4096, 4096, (0, 0, 4096)), ("q/o", 4096, 4096, (2048, 128, 1920)), ("k/v", 1024, 4096, (2048, 128, 1920)),
         ("gate/up", 14336, 4096, (2048, 128, 1920)), ("down", 4096, 14336, (12288, 1024, 1024))]
Ms = [int(a) for a in sys.argv[1:]] or [64, 65, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048]
def timed(f, n=100, reps=5):
    for _ in range(10): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return min(ts)
ws = torch.empty((512 << 20,), dtype=torch.uint8, device=dev)
ws[:4096].zero_()          # MM_WS_TICKETS_ZEROED (flags = 4 below): the in-kernel split-K counts arrivals there
FLAGS = 4
lib.mm_matmul_describe.restype = ctypes.c_char_p
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MICROMIX_"))
print(f"# {tag or 'default'}")
for name, N, K, split in CASES:
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    for M in Ms:
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        pp = lambda t: t.data_ptr() if t.numel() else None
        st = torch.cuda.current_stream().cuda_stream
        ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
        t = timed(lambda: lib.mm_matmul_ws(*ptrs, M, N, *split, 1, FLAGS, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
        wbytes = N * K // 2 + N * K // 32
        d = lib.mm_matmul_describe(M, N, *split, 1, FLAGS, ws.numel()).decode()
        print(f"{name:8s} N={N:5d} K={K:5d} {str(split):>20s} M={M:5d} | {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF  W-stream {wbytes/t/1e6:5.2f} TB/s | {d}", flush=True)
