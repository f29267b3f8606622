"""The MLP of a Llama-3-8B layer at decode sizes, norm included, as hipGraphs of 20 repetitions:
  A  rmsnorm_qlinear_decode (norm + quantize + gate | up GEMM) -> down_activate_decode (silu * up + quantize + down GEMM)
  B  rmsnorm_gate_up_activate_decode (... + silu * up + the quantization for down_proj inside) -> matmul (down GEMM)      (round 6)
  C  rmsnorm_quantize_x -> gate_up_activate (one launch at M <= 16 since round 6) -> matmul"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
H, I = 4096, 14336
in_split, down_split = (2048, 128, 1920), (12288, 1024, 1024)
rnd = lambda r, c: (torch.randn((r, c), generator=g, device=dev) * 0.02).to(torch.bfloat16)
x8 = torch.randn((32, H), generator=g, device=dev).to(torch.bfloat16)
idx = torch.argsort(x8.float().abs().mean(0)).to(torch.int16)
nw = torch.ones((H,), dtype=torch.bfloat16, device=dev)
gu = mixedgemm.interleave_gate_up(mixedgemm.reorder_quantize_w4(rnd(I, H), idx, *in_split), mixedgemm.reorder_quantize_w4(rnd(I, H), idx, *in_split))
wd = mixedgemm.downproj_quantize_w4(rnd(H, I), *down_split)
mm = lambda a, b: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (10 * reps))
    return best * 1e6
for m in (1, 2, 3, 4, 8, 16, 24, 32):
    x = x8[:m].contiguous()
    A = lambda: mixedgemm.down_activate_decode(mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, idx, *gu, *in_split), wd, *down_split)
    B = lambda: mm(mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, gu, *down_split), wd)
    C = lambda: mm(mixedgemm.gate_up_activate(mixedgemm.rmsnorm_quantize_x(x, nw, 1e-5, idx, *in_split), gu, *down_split), wd)
    ref = C()
    line = f"M={m}: C {graph_time(C):6.2f} us"
    if m <= 4:
        line += f"   A {graph_time(A):6.2f} us   B {graph_time(B):6.2f} us   identical: {bool(torch.equal(A(), ref)) and bool(torch.equal(B(), ref))}"
    if m <= 4:      # the launches one by one
        gub = mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, idx, *gu, *in_split)
        qh = mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, gu, *down_split)
        line += (f"\n      A: gate|up+norm {graph_time(lambda: mixedgemm.rmsnorm_qlinear_decode(x, nw, 1e-5, idx, *gu, *in_split)):5.2f}"
                 f" + down_activate {graph_time(lambda: mixedgemm.down_activate_decode(gub, wd, *down_split)):5.2f}"
                 f"   B: gate|up+norm+act {graph_time(lambda: mixedgemm.rmsnorm_gate_up_activate_decode(x, nw, 1e-5, idx, gu, *down_split)):5.2f}"
                 f" + down matmul {graph_time(lambda: mm(qh, wd)):5.2f}")
    print(line + f"   (rmsnorm_gate_up_activate_decode_supported {mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, I, *in_split)})", flush=True)
