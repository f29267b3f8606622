"""Randomised stress of the GEMM paths: run-to-run determinism of every path, cross-path agreement (split-K vs unsplit within
2 bf16 ulps of the running magnitude; fused decode vs two-op bit-equal, and with the RMSNorm inside), on random shapes / splits / modes.  `python tools/stress.py [seconds]`"""
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
def rand_split(k):
    g = k // 128
    a = int(rng.integers(0, g + 1)); b = int(rng.integers(0, g - a + 1))
    return (a * 128, b * 128, (g - a - b) * 128)
while time.time() < t_end:
    m = int(rng.choice([1, 2, 5, 8, 9, 16, 17, 33, 40, 48, 56, 64, 65, 100, 128, 129, 200, 256, 300, 384, 512, 700, 768, 1024, 1536, 2048, 4100]))
    n = int(rng.choice([16, 100, 128, 200, 256, 512, 1000, 1024, 2048, 4096, 4128, 8200, 14336]))
    k = int(rng.choice([128, 256, 384, 512, 1024, 2048, 4096]))
    if m * n > (1 << 23): k = min(k, 1024)      # keep the big outputs cheap
    split = rand_split(k)
    w4 = bool(rng.integers(0, 2)); rounding = "reference" if rng.integers(0, 2) else "fused"
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    x = (torch.randn((m, k), generator=g) * (1 + 10 * (torch.rand(k, generator=g) > 0.98))).to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    idx = torch.randperm(k, generator=g).to(torch.int16).to(dev)
    bias = torch.randn((n,), generator=g).to(torch.bfloat16).to(dev) if rng.integers(0, 2) else None
    b = (mixedgemm.reorder_quantize_w4 if w4 else mixedgemm.reorder_quantize_w)(w, idx, *split)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    a2 = mixedgemm.reorder_quantize_x(x, idx, *split)
    ok = all(torch.equal(p, q) for p, q in zip(a[:3], a2[:3]))
    args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    outs = {}
    for mode in (True, False, "force"):
        y1 = mixedgemm.matmul(*args, bias=bias, rounding=rounding, split_k=mode)
        y2 = mixedgemm.matmul(*args, bias=bias, rounding=rounding, split_k=mode)
        ok &= torch.equal(y1, y2)
        outs[mode] = y1.float()
    ref = outs[False]
    scale = ref.abs().amax().item() + 1e-6
    for mode in (True, "force"):
        ok &= ((outs[mode] - ref).abs().amax().item() <= scale * 2.0 ** -6)
    if m <= 8 and mixedgemm.qlinear_decode_supported(m, n, *split):
        y = mixedgemm.qlinear_decode(x, idx, *b, *split, bias=bias, rounding=rounding)
        ok &= torch.equal(y.float(), ref) if True else True
    if m <= 8 and k <= 8192 and mixedgemm.rmsnorm_qlinear_decode_supported(m, n, *split):
        # round 5: the decode launch with the RMSNorm inside against rmsnorm_quantize_x -> matmul, bit for bit
        nw = (1.0 + 0.3 * torch.randn((k,), generator=g)).to(torch.bfloat16).to(dev)
        ir = bool(rng.integers(0, 2))
        qn = mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, idx, *split, integer_round=ir)
        want = mixedgemm.matmul(qn[0], b[0], qn[1], b[1], qn[2], b[2], qn[3], b[3], qn[4], b[4], qn[5], b[5], bias=bias, rounding=rounding)
        got = mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, idx, *b, *split, bias=bias, rounding=rounding, integer_round=ir)
        ok &= torch.equal(got, want)
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", m, n, k, split, "w4" if w4 else "w", rounding, bias is not None, flush=True)
torch.cuda.synchronize()
print(f"{cases} random cases, {fails} mismatches", flush=True)
sys.exit(1 if fails else 0)
