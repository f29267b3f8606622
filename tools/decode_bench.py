"""Decode-sized QLinearLayer.forward (Llama-3-8B q/k/v + gate/up sharing their inputs): eager vs hipGraph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd.graph import GraphedForward
from micromix_amd.qlinear import FusedQLinear, QLinearLayer
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K, split = 4096, (2048, 128, 1920)
idx = torch.randperm(K, generator=g)
def mk(n):
    lin = torch.nn.Linear(K, n, bias=False, dtype=torch.bfloat16)
    return QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx)
qkv = [mk(4096), mk(1024), mk(1024)]
for M in (1, 16, 64):
    x = torch.randn((1, M, K), generator=g).to(torch.bfloat16).to(dev)
    def eager():
        return [l(x) for l in qkv]
    graphed = GraphedForward(qkv, x)
    fused = FusedQLinear(qkv)
    gfused = GraphedForward([fused], x)
    for name, f in (("eager", eager), ("hipGraph", lambda: graphed(x)), ("fused qkv", lambda: fused(x)), ("fused+graph", lambda: gfused(x))):
        for _ in range(20): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): f()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 300 * 1e6
        print(f"q+k+v projections, M={M:3d}, {name:11s}: {us:7.1f} us per step  ({M/us*1e6:,.0f} tokens/s)", flush=True)
