"""Where a workgroup of the weight-streaming kernel spends its time: python tools/stream_clock.py [M]  with
MICROMIX_HIP_LIB=micromix_amd/lib/dbg/lib_sclock.so (tools/build_one_variant.sh sclock mx_gemm_stream -DMM_STREAM_CLOCK=1).
Wave 0 of every workgroup stamps the 100 MHz clock at: start, ring primed (first D slabs requested), loop done, barrier passed, end."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
lib.mm_diag_set_stream_clock.restype = ctypes.c_int
lib.mm_diag_set_stream_clock.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# MODE (second argument): "matmul" (default: mm_matmul on pre-quantized rows), "decode" (mm_qlinear_decode: the rows are quantized inside
# every workgroup; `prime` is then that phase), "norm" (mm_rmsnorm_qlinear_decode: + the RMSNorm), "down" (mm_down_activate_decode)
MODE = sys.argv[2] if len(sys.argv) > 2 else "matmul"
for name, N, K, split in (("q/o", 4096, 4096, (2048, 128, 1920)), ("gate/up", 14336, 4096, (2048, 128, 1920)), ("gate+up", 28672, 4096, (2048, 128, 1920)), ("down", 4096, 14336, (12288, 1024, 1024))):
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
    b = mixedgemm.reorder_quantize_w4(w, idx, *split)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    a = mixedgemm.reorder_quantize_x(x, idx, *split)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
    f = lambda: lib.mm_matmul(*ptrs, M, N, *split, 1, 0, None, out.data_ptr(), st)
    if MODE == "act":        # mm_rmsnorm_gate_up_activate_decode: norm + quantize + gate | up GEMM + silu * up + the quantization for down_proj
        if name != "gate+up":
            continue
        I = N // 2
        dsp = (12288, 1024, 1024)
        nw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
        outs = [torch.empty((M, dsp[0] // 2), dtype=torch.uint8, device=dev), torch.empty((M, dsp[1] // 4 * 3), dtype=torch.uint8, device=dev),
                torch.empty((M, dsp[2]), dtype=torch.uint8, device=dev)] + [torch.empty((128 * d // 32,), dtype=torch.uint8, device=dev) for d in dsp]
        bp = [pp(t) for t in b]
        f = lambda: lib.mm_rmsnorm_gate_up_activate_decode(x.data_ptr(), nw.data_ptr(), 1e-5, idx.data_ptr(), *bp, M, I, *split, *dsp, 0,
                                                           *[t.data_ptr() for t in outs], None, 0, st)
        if f() != 0:
            print(f"{name:8s} M={M}: {MODE} not supported"); continue
    elif MODE == "down":       # mm_down_activate_decode: silu(gate) * up, quantized inside every workgroup (K = the intermediate size)
        if name != "down":
            continue
        gu = torch.randn((M, 2 * K), generator=g).to(torch.bfloat16).to(dev)
        bd = [pp(t) for t in mixedgemm.downproj_quantize_w4(w, *split)]
        f = lambda: lib.mm_down_activate_decode(gu.data_ptr(), *bd, M, N, *split, 1, 0, None, out.data_ptr(), st)
        if f() != 0:
            print(f"{name:8s} M={M}: {MODE} not supported"); continue
    elif MODE != "matmul":
        if K > 8192 and MODE == "norm":
            continue
        nw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
        bp = [pp(t) for t in b]
        if MODE == "decode":
            f = lambda: lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *bp, M, N, *split, 1, 0, None, out.data_ptr(), st)
        else:
            f = lambda: lib.mm_rmsnorm_qlinear_decode(x.data_ptr(), nw.data_ptr(), 1e-5, idx.data_ptr(), *bp, M, N, *split, 1, 0, None, out.data_ptr(), st)
        if f() != 0:
            print(f"{name:8s} M={M}: {MODE} not supported"); continue
    clock = torch.zeros((4096, 8), dtype=torch.int64, device=dev)
    assert lib.mm_diag_set_stream_clock(clock.data_ptr()) == 0
    for _ in range(50): f()
    torch.cuda.synchronize()
    clock.zero_(); torch.cuda.synchronize()
    f(); torch.cuda.synchronize()
    c = clock.cpu().numpy()
    c = c[c[:, 0] > 0][:, :5].astype(np.float64) / 100.0          # us
    if len(c) == 0:
        print(f"{name:8s} M={M}: not on the streaming kernel ({lib.mm_matmul_describe(M, N, *split, 1, 0, 0).decode()[:60]})"); continue
    t0 = c[:, 0].min()
    ph = np.diff(c, axis=1)
    starts = np.sort(c[:, 0] - t0); ends = np.sort(c[:, 4] - t0)
    print(f"   starts (us after the first): 10% {starts[len(c)//10]:.2f} 50% {starts[len(c)//2]:.2f} 90% {starts[9*len(c)//10]:.2f} max {starts[-1]:.2f};  ends: 10% {ends[len(c)//10]:.2f} 50% {ends[len(c)//2]:.2f} 90% {ends[9*len(c)//10]:.2f} max {ends[-1]:.2f}")
    print(f"{name:8s} M={M} N={N} K={K}: {len(c)} workgroups; start spread {c[:, 0].max() - t0:.2f} us; last end {c[:, 4].max() - t0:.2f} us after the first start; "
          f"per workgroup (median / max us): prime {np.median(ph[:, 0]):.2f}/{ph[:, 0].max():.2f}  loop {np.median(ph[:, 1]):.2f}/{ph[:, 1].max():.2f}  "
          f"barrier {np.median(ph[:, 2]):.2f}/{ph[:, 2].max():.2f}  reduce+store {np.median(ph[:, 3]):.2f}/{ph[:, 3].max():.2f}  total {np.median(c[:, 4] - c[:, 0]):.2f}/{(c[:, 4] - c[:, 0]).max():.2f}", flush=True)
