"""One decoder MLP at M token rows (Llama-3-8B shapes by default): quantize x, gate / up, silu * up + quantize, down -- as the
three-op form (one GEMM over the concatenated gate | up weights, activate_quantize_x, down GEMM) and with the fused gate / up launch
(mm_gate_up_activate).  Per step: back-to-back launches between two events (min / median of 7 x 10), DVFS settled.
    python tools/time_mlp.py [M=4096] [in=2048,128,1920] [down=12288,1024,1024]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm
dev = torch.device("cuda:0")
arg = lambda k, d: next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith(k + "=")), d)
M = int(arg("M", "4096")); H = int(arg("H", "4096")); I = int(arg("I", "14336"))
in_split = tuple(int(v) for v in arg("in", "2048,128,1920").split(","))
down_split = tuple(int(v) for v in arg("down", "12288,1024,1024").split(","))
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c, s=1.0: (torch.randn((r, c), generator=g, device=dev) * s).to(torch.bfloat16)
x = rnd(M, H); x[:, :: 97] *= 20
idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
wg, wu, wd = rnd(I, H, 0.02), rnd(I, H, 0.02), rnd(H, I, 0.02)
qg = mixedgemm.reorder_quantize_w4(wg, idx, *in_split); qu = mixedgemm.reorder_quantize_w4(wu, idx, *in_split)
qcat = tuple(torch.cat((a, b), 0).contiguous() for a, b in zip(qg, qu))
qgu = mixedgemm.interleave_gate_up(qg, qu)
qd = mixedgemm.downproj_quantize_w4(wd, *down_split)
del wg, wu, wd
mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
gu = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=dev)
y = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
gcat = mm(qx, qcat)
gate, up = gcat[:, :I].contiguous(), gcat[:, I:].contiguous()
qh = mixedgemm.activate_quantize_x(gate, up, *down_split)
steps = {
    "quantize_x": lambda: mixedgemm.reorder_quantize_x(x, idx, *in_split),
    "gate_up_gemm (one launch over 2 I features, bf16 out)": lambda: mm(qx, qcat, out=gu),
    "activate_quantize_x": lambda: mixedgemm.activate_quantize_x(gate, up, *down_split),
    "gate_up_activate (fused)": lambda: mixedgemm.gate_up_activate(qx, qgu, *down_split),
    "down_gemm": lambda: mm(qh, qd, out=y),
}
res = {}
for name, f in steps.items():
    for _ in range(100): f()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1000)
    ts.sort()
    res[name] = (round(ts[0], 1), round(ts[3], 1))
    print(f"{name:58s} min {ts[0]:7.1f} us  median {ts[3]:7.1f} us", flush=True)
three = sum(res[k][1] for k in ("gate_up_gemm (one launch over 2 I features, bf16 out)", "activate_quantize_x", "down_gemm"))
fused = res["gate_up_activate (fused)"][1] + res["down_gemm"][1]
print(json.dumps({"M": M, "H": H, "I": I, "in_split": in_split, "down_split": down_split, "three_op_us": round(three, 1), "fused_us": round(fused, 1),
                  "steps": res}))
