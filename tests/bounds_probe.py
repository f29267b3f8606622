"""Run by tests/test_stream_bounds_gpu.py in a child process: the weight-streaming kernel with every operand placed at the very END of a
hipMalloc allocation of its own (whole 2 MiB pages, so the bytes behind an operand are not part of any allocation of this process).
ADVICE r4: with the row-tile term in the buffer instruction's soffset a lane whose row lies inside the descriptor's range fetched rows past
M (M = 17 .. 31 with two token tiles) or past N (the last workgroup when N is not a multiple of 16 F) -- outputs that are never stored, but
reads behind the allocation.  Prints the SHA-1 of D for the operands at the end of their allocations and for the same bytes in the
middle of one; a memory fault kills this process (the parent reports it)."""
import ctypes, hashlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
PAGE = 2 << 20


def at_end(t):
    """device address of a copy of tensor t whose last byte is the last byte of a fresh hipMalloc allocation (whole pages)"""
    n = t.numel() * t.element_size()
    if n == 0:
        return None
    size = (n + PAGE - 1) // PAGE * PAGE
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), size) == 0
    dst = p.value + size - ((n + 15) // 16 * 16)        # 16-byte aligned, ends within 15 bytes of the allocation's end
    assert hip.hipMemcpy(dst, t.data_ptr(), n, 3) == 0   # hipMemcpyDeviceToDevice
    return dst


g = torch.Generator().manual_seed(1)
st = torch.cuda.current_stream().cuda_stream
for M, N, K, split in ((17, 8200, 1024, (512, 128, 384)), (31, 4096 + 24, 512, (256, 128, 128)), (49, 2056, 1024, (512, 128, 384)), (5, 8200, 512, (256, 0, 256))):
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    idx = torch.randperm(K, generator=g).to(torch.int16).to(dev)
    for wq in (mixedgemm.reorder_quantize_w4, mixedgemm.reorder_quantize_w):
        b = wq(w, idx, *split)
        a = mixedgemm.reorder_quantize_x(x, idx, *split)
        wmode = 1 if wq is mixedgemm.reorder_quantize_w4 else 0
        order = (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])
        want = mixedgemm.matmul(*order)
        out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
        ptrs = [at_end(t) for t in order]
        assert "stream" in lib.mm_matmul_describe(M, N, *split, wmode, 0, 0).decode()
        assert lib.mm_matmul(*ptrs, M, N, *split, wmode, 0, None, out.data_ptr(), st) == 0
        torch.cuda.synchronize()
        h = lambda t: hashlib.sha1(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()
        print("case", M, N, K, split, wmode, h(out), h(want), flush=True)
print("done", flush=True)
