"""Is the 4096^3 GEMM's time set by its power?  The same launch on operands that toggle fewer bits: random data (the bench's), operands
whose codes are all zero, and operands that are one constant.  A kernel that waits for data or issue slots takes the same time whatever
the values are; a kernel at the package power cap gets faster when its multipliers switch less.  Prints the time of back-to-back
launches (events around 400 of them after a 1.5 s settle phase) and rocm-smi's package power / clock under each load.
python tools/gemm_data_power.py [KN,KS,KO]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
split = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (0, 0, 4096)
x, w, idx = [t.to(dev) for t in bench.synth_inputs()]
M = N = K = 4096
cases = {"random (bench data)": (x, w), "all zero": (torch.zeros_like(x), torch.zeros_like(w)),
         "constant 1.0 / 0.02": (torch.ones_like(x), torch.full_like(w, 0.02)),
         "random activations, zero weights": (x, torch.zeros_like(w))}
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
ptr = lambda t: t.data_ptr() if t.numel() else None
st = torch.cuda.current_stream().cuda_stream
for name, (xx, ww) in cases.items():
    b = mixedgemm.reorder_quantize_w4(ww, idx, *split)
    a = mixedgemm.reorder_quantize_x(xx, idx, *split)
    args = [ptr(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
    f = lambda: lib.mm_matmul(*args, M, N, *split, 1, 0, None, out.data_ptr(), st)
    assert f() == 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(200): f()
        torch.cuda.synchronize()
    power = {}
    th = bench.sample_power(power, delay_s=0.3)
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 1.2 or th.is_alive():
        for _ in range(200): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 400
    print(f"split {split} {name:34s}: {us:6.2f} us per launch  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s   power {power.get('package_w')} W of {power.get('cap_w')}  sclk {power.get('sclk_mhz')} MHz", flush=True)
