"""One Mixtral-8x7B MoE FFN step at decode-sized token counts (BASELINE config 5 shapes, one GPU, no TP): the reference's per-expert
loop (quantize -> w1, w3 -> silu*mul -> quantize -> w2 per expert, qMixtralLayer.py:507-519) against the grouped entries
(reorder_quantize_x_grouped + matmul_grouped with w1|w3 concatenated along N).  Outputs are compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from micromix_amd import mixedgemm as mg
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
E, H, I = 8, 4096, 14336
s1, s2 = (3584, 256, 256), (12544, 1024, 768)          # MXFP4-dominant splits (BASELINE config 5)
idx1 = [torch.randperm(H, generator=g).to(torch.int16).to(dev) for _ in range(E)]
idx2 = [torch.randperm(I, generator=g).to(torch.int16).to(dev) for _ in range(E)]
B13, B1, B3, B2 = [], [], [], []
for e in range(E):
    w1 = (torch.randn((I, H), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    w3 = (torch.randn((I, H), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    w2 = (torch.randn((H, I), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b1, b3 = mg.reorder_quantize_w4(w1, idx1[e], *s1), mg.reorder_quantize_w4(w3, idx1[e], *s1)
    B1.append(b1); B3.append(b3)
    B13.append(tuple(torch.cat([p, q]) for p, q in zip(b1, b3)))      # w1 | w3 along N: rows and SF row tiles concatenate
    B2.append(mg.reorder_quantize_w4(w2, idx2[e], *s2))
    del w1, w3, w2
mm = lambda a, b: mg.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])

def loop(xs):
    outs = []
    for e, x in enumerate(xs):
        if x.size(0) == 0:
            outs.append(torch.empty((0, H), dtype=torch.bfloat16, device=dev)); continue
        a = mg.reorder_quantize_x(x, idx1[e], *s1)
        t = F.silu(mm(a, B1[e])) * mm(a, B3[e])
        outs.append(mm(mg.reorder_quantize_x(t, idx2[e], *s2), B2[e]))
    return outs

def grouped(xs):
    a = mg.reorder_quantize_x_grouped(xs, idx1, *s1)
    h = mg.matmul_grouped(a, B13)
    t = [F.silu(y[:, :I]) * y[:, I:] for y in h]
    return mg.matmul_grouped(mg.reorder_quantize_x_grouped(t, idx2, *s2), B2)

for ms in ((1,) * 8, (2,) * 8, (3, 0, 9, 1, 0, 20, 2, 5), (16,) * 8):
    xs = [torch.randn((m, H), generator=g).to(torch.bfloat16).to(dev) for m in ms]
    ya, yb = loop(xs), grouped(xs)
    torch.cuda.synchronize()
    same = all(torch.equal(p, q) for p, q in zip(ya, yb))
    res = []
    for f in (loop, grouped):
        for _ in range(5): f(xs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): f(xs)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e6)
    print(f"tokens per expert {ms}: per-expert loop {res[0]:7.1f} us   grouped {res[1]:7.1f} us   outputs identical: {same}", flush=True)
