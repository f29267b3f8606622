"""A/B of the fused decode launch (mm_qlinear_decode / mm_rmsnorm_qlinear_decode) across library variants, one process per variant
(MICROMIX_HIP_LIB), interleaved passes: python tools/time_decode_ab.py lib_a.so lib_b.so ...  ("default" = the product library);
AB_SHAPES=wide: the streaming kernel's layers; AB_PASSES=n; AB_HBM=1: weight copies in rotation (from HBM)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
def timed(fn, n=400):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
out = []
SHAPES = (("q/o", 4096, 4096, (2048, 128, 1920)), ("qkv", 6144, 4096, (2048, 128, 1920)), ("k/v", 1024, 4096, (2048, 128, 1920)), ("q/o fp8", 4096, 4096, (0, 0, 4096)))
if os.environ.get("AB_SHAPES") == "wide":      # the streaming kernel's layers
    SHAPES = (("gate/up", 14336, 4096, (2048, 128, 1920)), ("gate+up", 28672, 4096, (2048, 128, 1920)), ("q/o 70B", 8192, 8192, (4096, 256, 3840)))
for name, N, K, split in SHAPES:
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    for M in (1, 2, 4, 8):
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
        idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
        b = mixedgemm.reorder_quantize_w4(w, idx, *split)
        o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        nw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
        one = lambda: lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *[pp(t) for t in b], M, N, *split, 1, 0, None, o.data_ptr(), st)
        one_n = lambda: lib.mm_rmsnorm_qlinear_decode(x.data_ptr(), nw.data_ptr(), 1e-5, idx.data_ptr(), *[pp(t) for t in b], M, N, *split, 1, 0, None, o.data_ptr(), st)
        if os.environ.get("AB_HBM"):      # weights from HBM: copies in rotation, more than the 256 MB Infinity Cache holds
            nb = sum(t.numel() for t in b)
            cps = [[pp(t) for t in c] for c in [[t.clone() for t in b] for _ in range(max(2, int(600e6 // nb)))]]
            it = [0]
            def nxt():
                it[0] = (it[0] + 1) %% len(cps)
                return cps[it[0]]
            one = lambda: lib.mm_qlinear_decode(x.data_ptr(), idx.data_ptr(), *nxt(), M, N, *split, 1, 0, None, o.data_ptr(), st)
            one_n = lambda: lib.mm_rmsnorm_qlinear_decode(x.data_ptr(), nw.data_ptr(), 1e-5, idx.data_ptr(), *nxt(), M, N, *split, 1, 0, None, o.data_ptr(), st)
        assert one() == 0 and one_n() == 0
        out.append("%%-8s M=%%d plain %%5.2f norm %%5.2f" %% (name, M, timed(one), timed(one_n)))
print(" | ".join(out))
''' % ROOT
libs = sys.argv[1:] or ["default"]
for p in range(int(os.environ.get('AB_PASSES', '2'))):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", l)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{l:12s}", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
