"""Small fixed workload for the rocprofv3 passes: 10 x (reorder_quantize_x + matmul) per split on the 4096^3 bench shape, w4
weights.  `python tools/pmc_target.py [KN,KS,KO ...]` (default: the bench split (0,0,4096))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
splits = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [bench.SPLIT]
for split in splits:
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    for _ in range(10):
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
torch.cuda.synchronize()
print("ok")
