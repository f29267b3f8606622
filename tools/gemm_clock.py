"""In-kernel shader clock of the large-M GEMM (main loop only, epilogue excluded) via s_memtime / s_memrealtime."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the clock stamps exist only in the instrumented variant (csrc/mx_instrument.h): tools/build_variant.sh instr -DMM_INSTRUMENT
if "MICROMIX_HIP_LIB" not in os.environ:
    os.environ["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", "lib_instr.so")
if not os.path.exists(os.environ["MICROMIX_HIP_LIB"]):
    sys.exit("build the instrumented library first: tools/build_variant.sh instr -DMM_INSTRUMENT")
import torch
import bench
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("MICROMIX_HIP_LIB", "default"))
N = 4096
K = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("K=")), 4096)   # K=<depth>: e.g. K=14336 for down_proj splits
x, w, idx = [t.to(dev) for t in bench.synth_inputs(0 if K == 4096 else 1, 4096, N, K)]
M = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("M=")), 4096)   # M=<rows>: the first rows of the bench activations
x = x[:M].contiguous()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
clk = torch.zeros((4096, 4), dtype=torch.int64, device=dev)
ROUNDING = "fused" if "fused" in sys.argv else "reference"
SPLITS = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(0, 0, 4096), (4096, 0, 0), (2048, 128, 1920)]
for split in SPLITS:
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    f = lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out, rounding=ROUNDING)
    clk.zero_()
    for _ in range(3000): f()          # ~0.2 s of back-to-back launches so that DVFS settles
    torch.cuda.synchronize()
    lib.mm_diag_set_clock_buffer(clk.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    lib.mm_diag_set_kernel_events(e0.cuda_event, e1.cuda_event)
    f(); torch.cuda.synchronize()
    lib.mm_diag_set_kernel_events(None, None)
    lib.mm_diag_set_clock_buffer(None)
    kernel_us = e0.elapsed_time(e1) * 1e3
    c = clk.cpu().double()
    c = c[c[:, 1] > 0][:256]     # the workgroups that ran (up to one round of them)
    cyc, ticks = c[:, 0], c[:, 1]
    ghz = (cyc / ticks * 0.1)
    start, end = c[:, 2], c[:, 2] + c[:, 3]
    print(f"   start spread {(start.max()-start.min())/100:.2f} us; loop end (rel. first start): median {(start+ticks-start.min()).median()/100:.1f} max {(start+ticks-start.min()).max()/100:.1f} us; "
          f"wave0 stores done: median {(end-start.min()).median()/100:.1f} max {(end-start.min()).max()/100:.1f} us; kernel (dispatch events) {kernel_us:.1f} us")
    print(f"{tag:14s} {ROUNDING} M={M} split={split} ({len(c)} workgroups): loop cycles median {cyc.median():.0f}  loop time median {ticks.median()*10/1000:.1f} us  clock median {ghz.median():.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})", flush=True)
