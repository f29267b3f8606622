"""Per-step time of back-to-back 4096^3 GEMM launches: plain stream launches vs one hipGraph holding the same K launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
M = N = K = 4096
split = (0, 0, 4096)
b = mixedgemm.reorder_quantize_w4(w, idx, *split)
a = mixedgemm.reorder_quantize_x(x, idx, *split)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
step = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
STEPS = 200
for _ in range(50): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
    print(f"stream launches: {(time.perf_counter() - t0) / STEPS * 1e6:.2f} us per step", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(STEPS): step()
g.replay(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    print(f"one hipGraph of {STEPS} launches: {(time.perf_counter() - t0) / STEPS * 1e6:.2f} us per step", flush=True)

# experiment: the same K launches as two independent chains (alternating streams, separate outputs) inside one graph, so that
# the head of one kernel may overlap the tail of the other
out2 = torch.empty_like(out)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def step_to(o):
    mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=o)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    for i in range(STEPS // 2):
        with torch.cuda.stream(sa): step_to(out)
        with torch.cuda.stream(sb): step_to(out2)
    cur.wait_stream(sa); cur.wait_stream(sb)
g2.replay(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    g2.replay()
    torch.cuda.synchronize()
    print(f"hipGraph, two independent chains: {(time.perf_counter() - t0) / STEPS * 1e6:.2f} us per step", flush=True)
