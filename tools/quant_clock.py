"""Timeline of reorder_quantize_kernel's workgroups from in-kernel clock stamps (instrumented library:
tools/build_variant.sh instr -DMM_INSTRUMENT).  python tools/quant_clock.py [KN,KS,KO]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "MICROMIX_HIP_LIB" not in os.environ:
    os.environ["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", "lib_instr.so")
import torch
import bench
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
split = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (0, 0, 4096)
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
q = mixedgemm.reorder_quantize_x(x, idx, *split)
pp = lambda t: t.data_ptr() if t.numel() else None
st = torch.cuda.current_stream().cuda_stream
f = lambda: lib.mm_reorder_quantize(x.data_ptr(), 4096, 4096, idx.data_ptr(), *split, 0, *[pp(t) for t in q], st)
for _ in range(300): f()
torch.cuda.synchronize()
clk = torch.zeros((8192, 4), dtype=torch.int64, device=dev)
lib.mm_diag_set_quant_clock_buffer.argtypes = [ctypes.c_void_p]
assert lib.mm_diag_set_quant_clock_buffer(clk.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
lib.mm_diag_set_quant_clock_buffer(None)
c = clk.cpu().double(); c = c[c[:, 0] > 0]
t0 = c[:, 0].min()
us = lambda v: (v - t0) / 100.0
import numpy as np
qs = lambda v: " ".join(f"{np.percentile(us(v).numpy(), p):5.2f}" for p in (0, 10, 50, 90, 100))
print(f"split {split}: events around the launch {e0.elapsed_time(e1)*1e3:.2f} us; {len(c)} workgroups; percentiles 0/10/50/90/100 in us since the first start")
print("  start         ", qs(c[:, 0]))
print("  row staged    ", qs(c[:, 1]))
print("  group stored  ", qs(c[:, 2]))
print("  end (acked)   ", qs(c[:, 3]))
d = lambda a, b: " ".join(f"{np.percentile(((c[:, b] - c[:, a]) / 100.0).numpy(), p):5.2f}" for p in (10, 50, 90))
print("  per workgroup (10/50/90 %): load", d(0, 1), "| gather+convert+store issue", d(1, 2), "| store ack", d(2, 3))
