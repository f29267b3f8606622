"""Direct C-ABI timing of the fused decode kernel (mm_qlinear_decode) against quantize + GEMM, Llama-3-8B shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for name, N, K, split in (("q/o", 4096, 4096, (2048, 128, 1920)), ("qkv fused", 6144, 4096, (2048, 128, 1920)), ("k/v", 1024, 4096, (2048, 128, 1920)),
                          ("gate/up", 14336, 4096, (2048, 128, 1920)), ("gate+up fused", 28672, 4096, (2048, 128, 1920)),
                          ("down", 4096, 14336, (7168, 512, 6656))) + ((("70B q/o", 8192, 8192, (4096, 256, 3840)), ("70B qkv", 10240, 8192, (4096, 256, 3840)),
                          ("70B gate+up", 57344, 8192, (4096, 256, 3840))) if os.environ.get("TD_70B") else ()):
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    for M in (1, 2, 4, 8):
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
        b = mixedgemm.reorder_quantize_w4(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
        def two():
            lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *split, 0, *[pp(t) for t in a], st)
            lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st)
        one = lambda: lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *[pp(t) for t in b], M, N, *split, 1, 0, None, out.data_ptr(), st)
        assert one() == 0
        line = f"{name:14s} N={N:6d} K={K:6d} M={M}: quantize+gemm {timed(two):6.1f} us   fused {timed(one):6.1f} us"
        if K <= 8192 and lib.mm_rmsnorm_qlinear_decode_supported(M, N, *split):
            nw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
            def two_n():
                lib.mm_rmsnorm_quantize(x.data_ptr(), nw.data_ptr(), 1e-5, M, K, idx.data_ptr(), *split, 0, *[pp(t) for t in a], st)
                lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st)
            one_n = lambda: lib.mm_rmsnorm_qlinear_decode(x.data_ptr(), nw.data_ptr(), 1e-5, idx.data_ptr(), *[pp(t) for t in b], M, N, *split, 1, 0, None, out.data_ptr(), st)
            assert one_n() == 0
            line += f"   | with RMSNorm: rmsnorm_quantize+gemm {timed(two_n):6.1f} us   fused {timed(one_n):6.1f} us"
        print(line, flush=True)
