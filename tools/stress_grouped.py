"""Randomised stress of the grouped (MoE expert) entry points: mm_reorder_quantize_grouped and mm_matmul_grouped against the
per-group calls (bit-identical by contract), random group counts, token counts (0 included), N, K, splits, weight modes, bias.
python tools/stress_grouped.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "0")))
t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    ng = int(rng.integers(1, 11))
    ms = [int(rng.choice([0, 1, 5, 16, 17, 33, 64, 65, 100, 128, 129, 200, 300, 515])) for _ in range(ng)]
    n = int(rng.choice([72, 256, 1000, 1024, 4096, 4128, 8200]))
    k = int(rng.choice([128, 256, 384, 512, 1024, 2048]))
    g128 = k // 128
    a_ = int(rng.integers(0, g128 + 1)); b_ = int(rng.integers(0, g128 - a_ + 1))
    split = (a_ * 128, b_ * 128, (g128 - a_ - b_) * 128)
    w4 = bool(rng.integers(0, 2)); rounding = "reference" if rng.integers(0, 2) else "fused"
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    xs = [torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev) for m in ms]
    ws = [(torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16).to(dev) for _ in ms]
    idxs = [torch.randperm(k, generator=g).to(torch.int16).to(dev) for _ in ms]
    biases = [torch.randn((n,), generator=g).to(torch.bfloat16).to(dev) for _ in ms] if rng.integers(0, 2) else None
    qw = mixedgemm.reorder_quantize_w4 if w4 else mixedgemm.reorder_quantize_w
    bs = [qw(w, i, *split) for w, i in zip(ws, idxs)]
    qs = mixedgemm.reorder_quantize_x_grouped(xs, idxs, *split)
    ok = True
    for x, i, q in zip(xs, idxs, qs):
        if x.size(0) == 0:
            continue
        ref = mixedgemm.reorder_quantize_x(x, i, *split)
        ok &= all(torch.equal(p, r) for p, r in zip(q[:3], ref[:3]))
    outs = mixedgemm.matmul_grouped(qs, bs, biases=biases, rounding=rounding)
    outs2 = mixedgemm.matmul_grouped(qs, bs, biases=biases, rounding=rounding)
    for j, (m, q, b, y, y2) in enumerate(zip(ms, qs, bs, outs, outs2)):
        ok &= tuple(y.shape) == (m, n) and torch.equal(y, y2)
        if m == 0:
            continue
        ref = mixedgemm.matmul(q[0], b[0], q[1], b[1], q[2], b[2], q[3], b[3], q[4], b[4], q[5], b[5],
                               bias=None if biases is None else biases[j], rounding=rounding, split_k=False)
        ok &= torch.equal(y, ref)
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", ms, n, k, split, "w4" if w4 else "w", rounding, biases is not None, flush=True)
torch.cuda.synchronize()
print(f"{cases} random grouped cases, {fails} mismatches", flush=True)
sys.exit(1 if fails else 0)
