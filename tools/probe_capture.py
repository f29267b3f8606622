import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from micromix_amd import mixedgemm, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
m, n, k, split = 128, 1024, 4096, (2048, 128, 1920)
g0 = torch.Generator().manual_seed(1)
x = torch.randn((m, k), generator=g0).to(torch.bfloat16).to(dev); w = (torch.randn((n, k), generator=g0) * 0.02).to(torch.bfloat16).to(dev)
idx = torch.randperm(k, generator=g0).to(torch.int16).to(dev)
a = mixedgemm.reorder_quantize_x(x, idx, *split); b = mixedgemm.reorder_quantize_w4(w, idx, *split)
args = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
plain = mixedgemm.matmul(*args)
for mode in ("memset_node", "torch_zero"):
    os.environ["MM_CAPTURE_WS"] = mode
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): mixedgemm.matmul(*args, out=out)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): mixedgemm.matmul(*args, out=out)
    res = []
    for _ in range(4):
        out.zero_(); g.replay(); torch.cuda.synchronize()
        res.append(bool(torch.equal(out, plain)))
    print(mode, res, "max diff", float((out.float() - plain.float()).abs().max()), flush=True)
