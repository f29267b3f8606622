"""Mixtral-8x7B expert GEMMs at decode-sized token counts: one mm_matmul per expert vs mm_matmul_grouped (direct C-ABI calls)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
E = 8
for name, N, K, split in (("w1/w3 (each)", 14336, 4096, (3584, 256, 256)), ("w2", 4096, 14336, (12544, 1024, 768))):
    ws = [(torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev) for _ in range(E)]
    idxs = [torch.randperm(K, generator=g).to(torch.int16).to(dev) for _ in range(E)]
    Bs = [mixedgemm.reorder_quantize_w4(w, i, *split) for w, i in zip(ws, idxs)]
    del ws
    for ms in ((1,) * 8, (8,) * 8, (3, 0, 9, 1, 0, 20, 2, 5), (40,) * 8, (64,) * 8, (30, 64, 1, 50, 12, 0, 33, 17), (128,) * 8, (256,) * 8, (100, 300, 64, 200, 150, 90, 500, 120)):
        As = [mixedgemm.reorder_quantize_x(torch.randn((m, K), generator=g).to(torch.bfloat16).to(dev), i, *split) for m, i in zip(ms, idxs)]
        outs = [torch.empty((m, N), dtype=torch.bfloat16, device=dev) for m in ms]
        arr = (_lib.MMGroup * E)()
        for e, (a, b, o, m) in enumerate(zip(As, Bs, outs, ms)):
            arr[e].AN, arr[e].AS, arr[e].AO, arr[e].SFAN, arr[e].SFAS, arr[e].SFAO = (pp(t) for t in a)
            arr[e].BN, arr[e].BS, arr[e].BO, arr[e].SFBN, arr[e].SFBS, arr[e].SFBO = (pp(t) for t in b)
            arr[e].bias_bf16 = None; arr[e].D = o.data_ptr(); arr[e].M = m
        def loop():
            for a, b, o, m in zip(As, Bs, outs, ms):
                if m:
                    lib.mm_matmul(a[0].data_ptr(), b[0].data_ptr(), pp(a[1]), pp(b[1]), pp(a[2]), pp(b[2]), pp(a[3]), pp(b[3]), pp(a[4]), pp(b[4]),
                                  pp(a[5]), pp(b[5]), m, N, *split, 1, 0, None, o.data_ptr(), st)
        grouped = lambda: lib.mm_matmul_grouped(arr, E, N, *split, 1, 0, st)
        assert grouped() == 0
        t_loop, t_grouped = timed(loop), timed(grouped)
        again = f"   (loop again {timed(loop):6.1f}, grouped again {timed(grouped):6.1f})" if os.environ.get("TIME_GROUPED_TWICE") else ""
        print(f"{name:13s} N={N} K={K} tokens per expert {ms}: per-expert loop {t_loop:6.1f} us   grouped {t_grouped:6.1f} us{again}", flush=True)
