"""down_proj at decode sizes: mm_down_activate_decode (one launch, straight through the C ABI) against activate_quantize_x + matmul (Python
wrappers), one child process per pass: python tools/time_down.py [lib.so ...]  ("default" = the product library; AB_PASSES=n)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from micromix_amd import mixedgemm, _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timed(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
out = []
for name, N, I, split, wq in (("8B w4", 4096, 14336, (7168, 512, 6656), mixedgemm.downproj_quantize_w4), ("8B w", 4096, 14336, (7168, 512, 6656), mixedgemm.downproj_quantize_w),
                              ("8B fp4", 4096, 14336, (14336, 0, 0), mixedgemm.downproj_quantize_w4), ("70B w4", 8192, 28672, (14336, 1024, 13312), mixedgemm.downproj_quantize_w4)):
    w = (torch.randn((N, I), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b = wq(w, *split)
    for M in (1, 2, 4):
        if not mixedgemm.down_activate_decode_supported(M, N, *split, weight_mode="w4" if wq is mixedgemm.downproj_quantize_w4 else "w"): continue
        gu = torch.randn((M, 2 * I), generator=g).to(torch.bfloat16).to(dev)
        gate = gu.view(M, I // 128, 2, 128)[:, :, 0].reshape(M, I).contiguous(); up = gu.view(M, I // 128, 2, 128)[:, :, 1].reshape(M, I).contiguous()
        def two():
            a = mixedgemm.activate_quantize_x(gate, up, *split)
            return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
        ref = two()
        one = mixedgemm.down_activate_decode(gu, b, *split)
        same = bool(torch.equal(ref.view(torch.int16), one.view(torch.int16)))
        bp = [pp(t) for t in b]; wm = 1 if wq is mixedgemm.downproj_quantize_w4 else 0      # (straight through the C ABI: the Python wrapper costs ~10 us per call)
        assert wm == (_lib.MM_W_FP4 if wq is mixedgemm.downproj_quantize_w4 else _lib.MM_W_MATCH)
        t1 = timed(lambda: lib.mm_down_activate_decode(gu.data_ptr(), *bp, M, N, *split, wm, 0, None, one.data_ptr(), st))
        t2 = timed(two)
        # from HBM: enough copies of the weights to overflow the 256 MB Infinity Cache, one per call
        nb = sum(t.numel() for t in b)
        copies = [[t.clone() for t in b] for _ in range(max(2, int(600e6 // nb)))]
        cps = [[pp(t) for t in c] for c in copies]
        it = [0]
        def hbm():
            it[0] = (it[0] + 1) %% len(cps)
            lib.mm_down_activate_decode(gu.data_ptr(), *cps[it[0]], M, N, *split, wm, 0, None, one.data_ptr(), st)
        t3 = timed(hbm)
        del copies
        out.append("%%-7s M=%%d one %%5.2f (hbm %%5.2f) two %%5.2f %%s" %% (name, M, t1, t3, t2, "ok" if same else "MISMATCH"))
print(" | ".join(out))
''' % ROOT
libs = sys.argv[1:] or ["default"]
for p in range(int(os.environ.get("AB_PASSES", "2"))):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["MICROMIX_HIP_LIB"] = os.path.join(ROOT, "micromix_amd", "lib", "dbg", l)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{l:12s}", (r.stdout.strip().splitlines() or [r.stderr[-600:]])[-1], flush=True)
