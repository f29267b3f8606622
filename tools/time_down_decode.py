"""down_proj at decode sizes from the bf16 gate | up matrix: mm_down_activate_decode (one launch) against mm_activate_quantize-style
quantizer + mm_matmul (two launches), direct C-ABI calls; Llama-3-8B and Qwen2.5-14B shapes."""
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
    ts = []
    for _ in range(3):
        for _ in range(30): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return sorted(ts)[1]
for name, H, I, ds in (("llama3-8b", 4096, 14336, (12288, 1024, 1024)), ("qwen2.5-14b", 5120, 13824, (11776, 1024, 1024))):
    wd = (torch.randn((H, I), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b = mixedgemm.downproj_quantize_w4(wd, *ds)
    for M in (1, 2, 4):
        gu = torch.randn((M, 2 * I), generator=g).to(torch.bfloat16).to(dev)
        out = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
        one = lambda: lib.mm_down_activate_decode(gu.data_ptr(), *[pp(t) for t in b], M, H, *ds, 1, 0, None, out.data_ptr(), st)
        assert one() == 0
        qh = mixedgemm.activate_quantize_x(gu.view(M, I // 128, 2, 128)[:, :, 0].reshape(M, I).contiguous(), gu.view(M, I // 128, 2, 128)[:, :, 1].reshape(M, I).contiguous(), *ds)
        gate = gu.view(M, I // 128, 2, 128)[:, :, 0].reshape(M, I).contiguous(); up = gu.view(M, I // 128, 2, 128)[:, :, 1].reshape(M, I).contiguous()
        ptrs = [pp(t) for t in (qh[0], b[0], qh[1], b[1], qh[2], b[2], qh[3], b[3], qh[4], b[4], qh[5], b[5])]
        def two():
            lib.mm_activate_quantize(gate.data_ptr(), up.data_ptr(), M, *ds, *[pp(t) for t in qh], st)
            lib.mm_matmul(*ptrs, M, H, *ds, 1, 0, None, out.data_ptr(), st)
        print(f"{name:12s} down N={H} K={I} M={M}: activate_quantize + matmul {timed(two):6.2f} us   down_activate_decode {timed(one):6.2f} us", flush=True)
