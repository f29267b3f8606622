"""A/B timing of mixedgemm.matmul on a list of cases, one library per process (MICROMIX_HIP_LIB selects a variant build):
    python tools/time_cases.py M,N,K:KN,KS,KO [...]        e.g. 4096,4096,4096:4096,0,0  4096,4096,14336:12288,1024,1024
Per case: back-to-back launches between two events after a DVFS-settling warm-up (min and median of 7 x 20 launches), one line of
JSON per case; rocm-smi is NOT sampled here (tools/gemm_data_power.py does that)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
cases = []
for a in sys.argv[1:]:
    if ":" in a:
        mnk, sp = a.split(":")
        cases.append((tuple(int(v) for v in mnk.split(",")), tuple(int(v) for v in sp.split(","))))
cache = {}
for (M, N, K), split in cases:
    if (N, K) not in cache:
        cache.clear()
        x, w, idx = [t.to(dev) for t in bench.synth_inputs(0 if (N, K) == (4096, 4096) else 1, 4096, N, K)]
        cache[(N, K)] = (x, w, idx)
    x, w, idx = cache[(N, K)]
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    a = mixedgemm.reorder_quantize_x(x[:M].contiguous(), idx, *split)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
    for _ in range(400 if M >= 2048 else 1500): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1000)
    ts.sort()
    t_star = bench.roofline_time_s(M, N, split) * 1e6
    print(json.dumps({"lib": tag, "M": M, "N": N, "K": K, "split": split, "us_min": round(ts[0], 2), "us_median": round(ts[3], 2),
                      "frac_of_roofline": round(t_star / ts[3], 4)}), flush=True)
