"""Times mixedgemm.matmul on the bench shapes (4096^3) for a few splits; prints one line each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
M = N = K = 4096
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
tag = os.environ.get("MICROMIX_HIP_LIB", "default")
for split in ((0, 0, 4096), (4096, 0, 0), (2048, 128, 1920)):
    for name, fn in (("w4", mixedgemm.reorder_quantize_w4), ("w", mixedgemm.reorder_quantize_w)):
        b = fn(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        for _ in range(10): f()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1000)
        print(f"{os.path.basename(tag):16s} split={split} {name}: {min(ts):.1f} us (median {sorted(ts)[2]:.1f})  {2*M*N*K/min(ts)/1e6:.0f} TFLOP/s", flush=True)
