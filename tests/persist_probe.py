"""Run by tests/test_persist_gpu.py in child processes (MICROMIX_GEMM_PERSIST is read once per process): launches with more 256 x 256 tiles
than CUs -- plain GEMM (interior and ragged edges, bias, fp32 output, every first-segment kind, tail-balanced shapes) and the fused gate / up
epilogue (fp4, fp6 and fp8 consumer segments) -- and prints a SHA-1 of every output."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda r, c, s=1.0: (torch.randn((r, c), generator=g, device=dev) * s).to(torch.bfloat16)
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
# (M, N, K, split): tiles > 256 CUs in every case
CASES = [(4096, 6144, 1024, (512, 128, 384)),      # 384 tiles: 1.5 rounds, tail balancing; fp4 segment of 4 units: chained, prefetched
         (4096, 5120, 1024, (0, 512, 512)),        # no fp4 segment: the fp6 segment is first
         (4096, 5120, 512, (0, 0, 512)),           # the fp8 segment is first
         (4096, 5120, 512, (512, 0, 0)),           # all-fp4: no successor, not prefetched
         (4096, 5120, 768, (384, 128, 256)),       # an odd number of fp4 units: unchained first segment, not prefetched
         (4096, 5120, 1024, (768, 128, 128)),      # 6 fp4 units = 3 slabs of 256: the ring would start at stage 1, not prefetched
         (4000, 5000, 1024, (512, 128, 384)),      # ragged edges in both directions (N % 8 == 0)
         (3999, 4868, 512, (256, 128, 128))]       # N % 8 != 0: scalar stores
for M, N, K, split in CASES:
    x, w = rnd(M, K), rnd(N, K, 0.05)
    idx = torch.randperm(K, generator=g, device=dev).to(torch.int16)
    a, b = mixedgemm.reorder_quantize_x(x, idx, *split), mixedgemm.reorder_quantize_w4(w, idx, *split)
    bias = rnd(1, N).reshape(N)
    print("gemm", M, N, K, split, h(mm(a, b)), h(mm(a, b, bias=bias, rounding="fused")), h(mm(a, b, rounding="fused", out_dtype=torch.float32)),
          lib.mm_matmul_describe(M, N, *split, 1, 0, 0).decode()[:48], flush=True)
# fused gate / up: I = 2560 -> N = 5120 (320 tiles at M = 4096); consumer splits with fp4 / fp6 / fp8 tiles
for M, I, K, in_split, dsplit in ((4096, 2560, 1024, (512, 128, 384), (2048, 256, 256)), (4096, 2560, 512, (0, 0, 512), (0, 1280, 1280)),
                                  (3900, 2560, 1024, (512, 128, 384), (2304, 128, 128))):
    x, wg, wu = rnd(M, K), rnd(I, K, 0.08), rnd(I, K, 0.08)
    idx = torch.randperm(K, generator=g, device=dev).to(torch.int16)
    qx = mixedgemm.reorder_quantize_x(x, idx, *in_split)
    gu = mixedgemm.interleave_gate_up(mixedgemm.reorder_quantize_w4(wg, idx, *in_split), mixedgemm.reorder_quantize_w4(wu, idx, *in_split))
    q = mixedgemm.gate_up_activate(qx, gu, *dsplit)
    # (the SF tensors hold uninitialised padding: hash the bytes of real rows only, through the reference layout)
    print("act", M, I, K, in_split, dsplit, *[h(t) for t in q[:3]], lib.mm_gate_up_activate_describe(M, I).decode()[:52], flush=True)
torch.cuda.synchronize()
print("done", flush=True)
