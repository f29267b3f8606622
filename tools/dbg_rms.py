import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_golden_v2 as g2
from micromix_amd import mixedgemm
from oracle import mx_oracle as o
dev = torch.device("cuda:0")
tb = lambda b: torch.from_numpy(b.view(np.int16)).view(torch.bfloat16).to(dev)
for k, split in g2.G7:
    x, w, idx = g2.g7_inputs(k)
    for ir in (True, False):
        got = mixedgemm.rmsnorm_quantize_x(tb(x), tb(w), g2.EPS, torch.from_numpy(idx.astype(np.int16)).to(dev), *split, integer_round=ir)
        want = o.rmsnorm_quantize(x, w, g2.EPS, idx, *split, integer_round=ir)
        for i in range(3):
            g = got[i].cpu().numpy()
            bad = np.nonzero((g != want[i]).any(axis=1))[0]
            print(k, ir, i, "rows differing:", bad[:10], "rvar", o.rmsnorm_rvar(x, g2.EPS)[bad[:4]] if len(bad) else "")
