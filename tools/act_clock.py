"""Phase times of the fused gate / up launch (mm_gate_up_activate) and of the plain GEMM over the same 2 I features, per workgroup:
main loop against everything (epilogue included), from the in-kernel stamps of the instrumented library
(tools/build_variant.sh instr -DMM_INSTRUMENT).   python tools/act_clock.py [M=4096] [in=2048,128,1920]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "MICROMIX_HIP_LIB" not in os.environ:
    os.environ["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", "lib_instr.so")
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
arg = lambda k, d: next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith(k + "=")), d)
M = int(arg("M", "4096")); H, I = 4096, 14336
in_split = tuple(int(v) for v in arg("in", "2048,128,1920").split(",")); down_split = (12288, 1024, 1024)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c, s=1.0: (torch.randn((r, c), generator=g, device=dev) * s).to(torch.bfloat16)
x = rnd(M, H); x[:, ::97] *= 20
idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
qg = mixedgemm.reorder_quantize_w4(rnd(I, H, 0.02), idx, *in_split); qu = mixedgemm.reorder_quantize_w4(rnd(I, H, 0.02), idx, *in_split)
qgu = mixedgemm.interleave_gate_up(qg, qu)
qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
gu = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=dev)
mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
clk = torch.zeros((8192, 4), dtype=torch.int64, device=dev)
for name, f in (("plain GEMM over gate | up (bf16 out)", lambda: mm(qx, qgu, out=gu)), ("fused gate_up_activate", lambda: mixedgemm.gate_up_activate(qx, qgu, *down_split))):
    for _ in range(300): f()
    torch.cuda.synchronize()
    clk.zero_()
    lib.mm_diag_set_clock_buffer(clk.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    lib.mm_diag_set_clock_buffer(None)
    c = clk.cpu().double(); c = c[c[:, 1] > 0]
    cyc, loop, start, done = c[:, 0], c[:, 1] / 100, c[:, 2] / 100, c[:, 3] / 100
    t0 = start.min()
    print(f"{name}: {len(c)} workgroups, launch {e0.elapsed_time(e1)*1e3:.1f} us; per workgroup: loop median {loop.median():.1f} us ({cyc.median():.0f} cycles, "
          f"{(cyc/ (c[:,1]*10)).median():.3f} GHz), start -> stores done median {done.median():.1f} us, i.e. {(done-loop).median():.1f} us outside the loop; "
          f"last workgroup done {(start+done-t0).max():.1f} us after the first start", flush=True)
