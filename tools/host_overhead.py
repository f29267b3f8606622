"""Host-side cost per op call (no device sync inside the loop; tiny problem so the GPU is never the bottleneck)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import mixedgemm, _lib
dev = torch.device("cuda:0")
M, N, K = 16, 128, 256
x = torch.randn((M, K), device=dev).to(torch.bfloat16)
w = torch.randn((N, K), device=dev).to(torch.bfloat16)
idx = torch.arange(K, dtype=torch.int16, device=dev)
split = (128, 0, 128)
b = mixedgemm.reorder_quantize_w4(w, idx, *split)
a = mixedgemm.reorder_quantize_x(x, idx, *split)
def t(f, n=3000):
    for _ in range(200): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6
print("torch.empty((16,128),u8):        %.2f us" % t(lambda: torch.empty((16, 128), dtype=torch.uint8, device=dev)))
print("current_stream().cuda_stream:    %.2f us" % t(lambda: torch.cuda.current_stream(dev).cuda_stream))
print("with torch.cuda.device(dev):     %.2f us" % t(lambda: torch.cuda.device(dev).__enter__()))
print("reorder_quantize_x:              %.2f us" % t(lambda: mixedgemm.reorder_quantize_x(x, idx, *split)))
print("matmul:                          %.2f us" % t(lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])))
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
print("matmul(out=):                    %.2f us" % t(lambda: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)))
lib = _lib.load()
print("raw ctypes mm_version():         %.2f us" % t(lambda: lib.mm_version()))
print("torch add (reference point):     %.2f us" % t(lambda: x + x))
# the C entry points alone (preallocated outputs): what the library itself spends per launch on the host
pp = lambda t_: t_.data_ptr() if t_.numel() else None
st = torch.cuda.current_stream().cuda_stream
qo = mixedgemm.reorder_quantize_x(x, idx, *split)
print("raw mm_reorder_quantize:         %.2f us" % t(lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *split, 0, *[pp(q) for q in qo], st)))
nw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
print("raw mm_rmsnorm_quantize:         %.2f us" % t(lambda: lib.mm_rmsnorm_quantize(x.data_ptr(), nw.data_ptr(), 1e-5, M, K, idx.data_ptr(), *split, 0, *[pp(q) for q in qo], st)))
args = [pp(t_) for t_ in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
print("raw mm_matmul:                   %.2f us" % t(lambda: lib.mm_matmul(*args, M, N, *split, 1, 0, None, out.data_ptr(), st)))
