"""Stress of the split-K whose reduction runs inside the GEMM launch (64 x 64 tiles, ticket per tile): random shapes that take that
plan, many back-to-back launches of alternating shapes through ONE workspace (its ticket counters must be back at zero after
every launch), every result compared bit for bit with the first run of its shape and, within the GEMM tolerance, with the unsplit
kernel.  A second stream runs its own shapes at the same time (own workspace).  python tools/stress_split.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))


def make_case():
    while True:
        m = int(rng.integers(65, 400)); n = int(rng.choice([256, 384, 512, 640, 768, 1000, 1024])); k = int(rng.choice([2048, 2560, 4096, 5120, 8192]))
        g = k // 128
        a = int(rng.integers(0, g + 1)); b = int(rng.integers(0, g - a + 1))
        split = (a * 128, b * 128, (g - a - b) * 128)
        w4 = bool(rng.integers(0, 2))
        need = lib.mm_matmul_workspace_bytes(m, n, *split, 1 if w4 else 0, _lib.MM_WS_TICKETS_ZEROED)
        if need and "in-kernel" in lib.mm_matmul_describe(m, n, *split, 1 if w4 else 0, _lib.MM_WS_TICKETS_ZEROED, need).decode():
            break
    gen = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    x = torch.randn((m, k), generator=gen).to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=gen) * 0.05).to(torch.bfloat16).to(dev)
    idx = torch.randperm(k, generator=gen).to(torch.int16).to(dev)
    qa = mixedgemm.reorder_quantize_x(x, idx, *split)
    qb = (mixedgemm.reorder_quantize_w4 if w4 else mixedgemm.reorder_quantize_w)(w, idx, *split)
    args = (qa[0], qb[0], qa[1], qb[1], qa[2], qb[2], qa[3], qb[3], qa[4], qb[4], qa[5], qb[5])
    rounding = "reference" if rng.integers(0, 2) else "fused"
    return args, rounding, (m, n, k, split, w4)


t_end = time.time() + budget
launches = cases = fails = 0
side = torch.cuda.Stream()
while time.time() < t_end:
    group = [make_case() for _ in range(4)]
    first = []
    for args, rounding, tag in group:
        y = mixedgemm.matmul(*args, rounding=rounding)
        ref = mixedgemm.matmul(*args, rounding=rounding, split_k=False).float()
        tol = 2.0 ** -6 * float(ref.abs().max()) + 1e-3
        if float((y.float() - ref).abs().max()) > tol:
            fails += 1; print("split vs unsplit", tag, float((y.float() - ref).abs().max()), tol)
        first.append(y)
        cases += 1
    outs_side = []
    for rep in range(40):
        order = rng.permutation(4)
        with torch.cuda.stream(side):      # the other stream: its own workspace (mixedgemm.split_workspace is per stream)
            a2, r2, _ = group[int(order[0])]
            outs_side.append((int(order[0]), mixedgemm.matmul(*a2, rounding=r2)))
        for j in order:
            args, rounding, tag = group[int(j)]
            y = mixedgemm.matmul(*args, rounding=rounding)
            launches += 1
            if not torch.equal(y, first[int(j)]):
                fails += 1; print("nondeterministic", tag, rep)
    torch.cuda.synchronize()
    for j, y in outs_side:
        launches += 1
        if not torch.equal(y, first[j]):
            fails += 1; print("nondeterministic on the side stream", group[j][2])
print(f"{cases} shapes, {launches} in-kernel split launches, {fails} mismatches")
sys.exit(1 if fails else 0)
