"""Small fixed workload for the rocprofv3 passes: 10 x (reorder_quantize_x + matmul) per split on the 4096^3 bench shape, w4
weights.  `python tools/pmc_target.py [KN,KS,KO[@MxNxK] ...]` (default: the bench split (0,0,4096); @MxNxK = another shape, e.g.
12288,1024,1024@4096x4096x14336 for down_proj; M <= 8 also runs the fused decode kernel, mm_qlinear_decode, ten times;
act:KN,KS,KO = ten fused gate / up launches, mm_gate_up_activate, at the Llama-3-8B MLP shape; w:KN,KS,KO = the matching-precision
weight mode, reorder_quantize_w, on the 4096^3 shape)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import _lib, mixedgemm
dev = torch.device("cuda:0")
args = sys.argv[1:] or [",".join(str(v) for v in bench.SPLIT)]
for arg in args:
    if arg.startswith("act:"):
        # the fused gate / up launch (mm_gate_up_activate) at the Llama-3-8B MLP shape: M = 4096, hidden 4096, intermediate 14336
        split = tuple(int(v) for v in arg[4:].split(","))
        m, h, inter, dsplit = 4096, 4096, 14336, (12288, 1024, 1024)
        x, w, idx = [t.to(dev) for t in bench.synth_inputs(1, m, inter, h)]
        pg = mixedgemm.reorder_quantize_w4(w, idx, *split)
        pu = mixedgemm.reorder_quantize_w4((w.float() * 0.5).to(torch.bfloat16), idx, *split)
        gu = mixedgemm.interleave_gate_up(pg, pu)
        del pg, pu
        for _ in range(10):
            a = mixedgemm.reorder_quantize_x(x, idx, *split)
            mixedgemm.gate_up_activate(a, gu, *dsplit)
        continue
    w_match = arg.startswith("w:")
    if w_match:
        arg = arg[2:]
    sp, _, shape = arg.partition("@")
    split = tuple(int(v) for v in sp.split(","))
    m, n, k = (int(v) for v in shape.split("x")) if shape else (bench.M, bench.N, bench.K)
    x, w, idx = [t.to(dev) for t in bench.synth_inputs(0 if not shape else 1, m, n, k)]
    b = (mixedgemm.reorder_quantize_w if w_match else mixedgemm.reorder_quantize_w4)(w, idx, *split)
    for _ in range(10):
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
    if m <= 8 and not w_match:
        lib = _lib.load()
        od = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        wp = [t.data_ptr() if t.numel() else None for t in b]
        for _ in range(10):
            assert lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *wp, m, n, *split, 1, 0, None, od.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
print("ok")
