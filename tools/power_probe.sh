#!/bin/bash
# power / clock telemetry while the headline GEMM runs back to back: bash tools/power_probe.sh [KN,KS,KO]
cd ${GRAFT_REPO_ROOT:-/root/repo}
SPLIT=${1:-0,0,4096}
rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
python3 - "$SPLIT" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import bench
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
split = tuple(int(v) for v in sys.argv[1].split(","))
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
b = mixedgemm.reorder_quantize_w4(w, idx, *split)
a = mixedgemm.reorder_quantize_x(x, idx, *split)
out = torch.empty((4096, 4096), dtype=torch.bfloat16, device=dev)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(500):
        mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
    torch.cuda.synchronize()
PY
sleep 6
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E -i "power|sclk|mclk|fclk|socclk|junction|edge" ; sleep 1; done
wait
