"""Phase times of the in-kernel split-K (64-row tiles) from in-kernel clock stamps; needs the instrumented library
(tools/build_variant.sh instr -DMM_INSTRUMENT).  python tools/split_clock.py M N [KN,KS,KO]   with MICROMIX_SPLIT_SMALL=<tile>:<S>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "MICROMIX_HIP_LIB" not in os.environ:
    os.environ["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", "lib_instr.so")
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
M, N = int(sys.argv[1]), int(sys.argv[2])
split = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2048, 128, 1920)
K = sum(split)
g = torch.Generator().manual_seed(0)
w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
b = mixedgemm.reorder_quantize_w4(w, idx, *split); a = mixedgemm.reorder_quantize_x(x, idx, *split)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
ws = torch.empty((256 << 20,), dtype=torch.uint8, device=dev); ws[:4096].zero_()
pp = lambda t: t.data_ptr() if t.numel() else None
ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
st = torch.cuda.current_stream().cuda_stream
f = lambda: lib.mm_matmul_ws(*ptrs, M, N, *split, 1, 4, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
for _ in range(200): f()
torch.cuda.synchronize()
clk = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
lib.mm_diag_set_clock_buffer(clk.data_ptr()); lib.mm_diag_set_kernel_events(e0.cuda_event, e1.cuda_event)
f(); torch.cuda.synchronize()
lib.mm_diag_set_kernel_events(None, None); lib.mm_diag_set_clock_buffer(None)
c = clk.cpu().double(); c = c[c[:, 0] > 0]
last = c[:, 3] >= 2 ** 40
c[:, 3] = c[:, 3] % 2 ** 40
t0 = c[:, 0].min()
us = lambda v: v / 100.0
print(lib.mm_matmul_describe(M, N, *split, 1, 4, ws.numel()).decode())
print(f"kernel (dispatch events) {e0.elapsed_time(e1)*1e3:.2f} us; {len(c)} workgroups, {int(last.sum())} reducers; start spread {us(c[:,0].max()-t0):.2f} us")
for name, col in (("slabs done", 1), ("partials acked", 2), ("ticket drawn", 3)):
    v = c[:, col]
    print(f"  {name:16s} since own start: median {us(v.median()):.2f} max {us(v.max()):.2f} us")
if last.any():
    r = c[last]
    print(f"  reducers: ticket {us(r[:,3].median()):.2f}, partials read+summed {us(r[:,4].median()):.2f}, tile written {us(r[:,5].median()):.2f} us since own start (medians); "
          f"last tile written {us((r[:,0]+r[:,5]).max()-t0):.2f} us after the first start")
