"""First-contact hardware probe: prints (does not assert) how the GPU compares with the oracle.
Run on the GPU box:  python tests/gpu_probe.py   (a script, not collected by pytest; it lives here because only tests use the oracle)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

t0 = time.time()
import torch  # noqa: E402

print("torch import", round(time.time() - t0, 1), "s", torch.__version__, torch.cuda.get_device_name(0), flush=True)
from micromix_amd import _lib, mixedgemm  # noqa: E402
from oracle import mx_oracle as o  # noqa: E402
import hw_layout as hl  # noqa: E402
from conftest import t_from_bits, bits_from_t, u8, make_inputs  # noqa: E402

lib = _lib.load_diag()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)

print("== single MFMA layout hypothesis ==")
for shape in (32, 16):
    for ea in hl.ELS:
        for eb in hl.ELS:
            errs = [hl.run_case(lib, torch, dev, rng, shape, ea, eb, op)[0] for op in range(4)]
            print(f"mfma{shape} A={ea} B={eb} rel.err per opsel: " + " ".join(f"{e:.2e}" for e in errs), flush=True)

print("== hardware MX converters vs oracle encode ==")
allb = np.arange(65536, dtype=np.uint16)
finite = np.isfinite(o.bf16_to_f32(allb))
src = allb[finite]
src = src[: len(src) // 32 * 32]
tsrc = t_from_bits(src, dev)
for el in hl.ELS:
    for e in (-3, 0, 2):
        out = torch.zeros(len(src), dtype=torch.uint8, device=dev)
        st = lib.mm_diag_hw_convert(tsrc.data_ptr(), len(src), float(2.0 ** e), hl.ELS.index(el), out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        x = o.bf16_to_f32(src)
        fm = o.FORMATS[el]["fmax"]
        for name, scaled in (("div", x.astype(np.float64) / 2.0 ** e), ("mul", x.astype(np.float64) * 2.0 ** e)):
            want = o.encode(np.clip(scaled, -fm, fm).astype(np.float32), el)
            inr = np.abs(scaled) <= fm
            bad = (got != want)
            print(f"cvt {el} scale=2^{e} hyp={name}: mismatches {int(bad.sum())} (in-range {int((bad & inr).sum())}) of {len(src)}",
                  flush=True)
            if 0 < (bad & inr).sum() < 40000 and name == "div":
                ii = np.nonzero(bad & inr)[0][:6]
                for i in ii:
                    print("     x=", x[i], "scaled", scaled[i], "hw", hex(got[i]), "oracle", hex(want[i]))

print("== reorder_quantize vs oracle ==")
for (M, K, split) in [(130, 4096, (2048, 1024, 1024)), (64, 4096, (0, 0, 4096)), (33, 5120, (4096, 512, 512)),
                      (16, 14336, (7168, 512, 6656))]:
    xb = make_inputs(rng, M, K)
    idx = rng.permutation(K).astype(np.int16)
    x = t_from_bits(xb, dev)
    tidx = torch.from_numpy(idx).to(dev)
    for mode, fn in (("x", mixedgemm.reorder_quantize_x), ("w", mixedgemm.reorder_quantize_w), ("w4", mixedgemm.reorder_quantize_w4)):
        got = fn(x, tidx, *split)
        torch.cuda.synchronize()
        want = o.reorder_quantize(xb, idx, *split, mode)
        res = []
        for gi, (g, w, kseg) in enumerate(zip(got, want, list(split) * 2)):
            g = u8(g)
            if gi < 3:
                res.append(int((g != w).sum()))
            else:
                offs = o.sf_valid_offsets(M, kseg)
                res.append(int((g[offs] != w[offs]).sum()))
        print(f"quant M={M} K={K} split={split} mode={mode}: mismatching bytes {res}", flush=True)

print("== matmul vs oracle ==")
for (M, N, K, split) in [(128, 128, 512, (0, 0, 512)), (128, 128, 512, (512, 0, 0)), (128, 128, 512, (0, 512, 0)),
                         (130, 256, 4096, (2048, 1024, 1024)), (64, 384, 1024, (512, 128, 384))]:
    xb = make_inputs(rng, M, K)
    wb = make_inputs(rng, N, K, "weight")
    idx = rng.permutation(K).astype(np.int16)
    x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
    a = mixedgemm.reorder_quantize_x(x, tidx, *split)
    for wmode, fn in (("w4", mixedgemm.reorder_quantize_w4), ("w", mixedgemm.reorder_quantize_w)):
        b = fn(w, tidx, *split)
        for rounding in ("reference", "fused"):
            d = mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], rounding=rounding)
            torch.cuda.synchronize()
            an = [u8(t) for t in a]
            bn = [u8(t) for t in b]
            want = o.matmul(an[0], bn[0], an[1], bn[1], an[2], bn[2], an[3], bn[3], an[4], bn[4], an[5], bn[5], rounding=rounding)
            got = bits_from_t(d)
            ulp = o.bf16_ulp_distance(got, want)
            rel = np.linalg.norm(o.bf16_to_f32(got).astype(np.float64) - o.bf16_to_f32(want)) / (np.linalg.norm(o.bf16_to_f32(want)) + 1e-30)
            print(f"matmul M={M} N={N} K={K} split={split} {wmode} {rounding}: max ulp {int(ulp.max())} frac>0 {(ulp > 0).mean():.2e} "
                  f"frac>1 {(ulp > 1).mean():.2e} rel {rel:.2e}", flush=True)

print("== timing (4096^3, all-fp8 activations) ==")
M = N = K = 4096
xb = make_inputs(rng, M, K)
wb = make_inputs(rng, N, K, "weight")
idx = rng.permutation(K).astype(np.int16)
x, w, tidx = t_from_bits(xb, dev), t_from_bits(wb, dev), torch.from_numpy(idx).to(dev)
for split in [(0, 0, 4096), (4096, 0, 0), (2048, 128, 1920)]:
    for wmode, fn in (("w4", mixedgemm.reorder_quantize_w4), ("w", mixedgemm.reorder_quantize_w)):
        b = fn(w, tidx, *split)
        a = mixedgemm.reorder_quantize_x(x, tidx, *split)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        def run_q():
            return mixedgemm.reorder_quantize_x(x, tidx, *split)
        def run_g():
            return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        for name, f in (("quant_x", run_q), ("matmul", run_g)):
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            extra = f"{2 * M * N * K / ms / 1e9:.1f} TFLOP/s" if name == "matmul" else f"{(2 * M * K) / ms / 1e6:.1f} GB/s(in)"
            print(f"time split={split} {wmode} {name}: {ms * 1000:.1f} us  {extra}", flush=True)
print("done", round(time.time() - t0, 1), "s")
