"""GEMM with fp32 accumulator output (MM_OUT_F32, the tensor-parallel partial sums) against the bf16 output: kernel time by events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
for M in (4096, 512, 128, 16):
    for split in ((0, 0, 4096), (2048, 128, 1920)):
        b = mixedgemm.reorder_quantize_w4(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x[:M].contiguous(), idx, *split)
        for name, kw in (("bf16", dict(rounding="fused")), ("fp32", dict(rounding="fused", out_dtype=torch.float32))):
            out = torch.empty((M, 4096), dtype=kw.get("out_dtype", torch.bfloat16), device=dev)
            f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out, **kw)
            for _ in range(200): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): f()
            e1.record(); torch.cuda.synchronize()
            print(f"M={M:5d} split={split}: {name} output {e0.elapsed_time(e1) * 5:7.1f} us per call", flush=True)
