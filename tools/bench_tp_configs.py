"""BASELINE.json configs 4 and 5 on ONE GPU: the per-rank work of the K-sharded (row-parallel) layers.

Config 4: Qwen2.5-14B linears, TP=4 (hidden 5120, intermediate 13824, q has bias); config 5: Mixtral-8x7B expert FFN, TP=8,
MXFP4-dominant split, M = tokens routed to one expert.  For every layer the slowest rank's shard (micromix_amd.tp.plan_k_shards)
is quantized and multiplied here; the all-reduce of the [M, N] bf16 partial is NOT included (one GPU) -- its size is printed.
Direct C-ABI calls, event timing.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib, mixedgemm, tp
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
pp = lambda t: t.data_ptr() if t.numel() else None


def timed(fn, n):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


def run(tag, name, N, K, split, world, M, bias):
    plan = tp.plan_k_shards(*split, world)
    cost = [sum(w * c for (_, w), c in zip(p, tp.SEGMENT_COST)) for p in plan]
    shard = plan[max(range(world), key=lambda r: cost[r])]
    sw = tuple(w for _, w in shard)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
    sub = tp.shard_index(idx, *split, shard)
    b = mixedgemm._quantize(w, sub, *sw, "w4", "reorder_quantize_w4", gather_subset=True)
    a = mixedgemm._quantize(x, sub, *sw, "x", "reorder_quantize_x", gather_subset=True)
    bvec = torch.randn((N,), generator=g).to(torch.bfloat16).to(dev) if bias else None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ptrs = [pp(t) for t in (a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5])]
    n = 200 if M <= 256 else 50
    tg = timed(lambda: lib.mm_matmul_ws(*ptrs, M, N, *sw, 1, 0, pp(bvec) if bias else None, out.data_ptr(), ws.data_ptr(), ws.numel(), st), n)
    tq = timed(lambda: lib.mm_reorder_quantize_gather(x.data_ptr(), M, K, sub.data_ptr(), *sw, 0, *[pp(t) for t in a], st), n)
    ks = sum(sw)
    print(f"{tag:9s} {name:14s} N={N:6d} K={K:6d} split={str(split):>20s} rank shard={str(sw):>18s} M={M:5d} | gemm {tg:7.1f} us "
          f"{2*M*N*ks/tg/1e6:6.0f} TFLOP/s | quant {tq:5.1f} us | all-reduce payload {M*N*2/2**20:6.2f} MiB", flush=True)


Ms = [int(a) for a in sys.argv[1:]] or [16, 256, 2048]
# config 4: Qwen2.5-14B (hidden 5120, 40 heads / 8 KV heads x 128, intermediate 13824), TP=4; splits scaled from the reference's
# bench constants (bench_reorder_gemm.cu:28-30): half fp4, one 128-granule per 4096 of fp6, rest fp8
for M in Ms:
    for name, N, K, split, bias in (("q_proj(+bias)", 5120, 5120, (2560, 256, 2304), True), ("k/v_proj(+bias)", 1024, 5120, (2560, 256, 2304), True),
                                    ("o_proj", 5120, 5120, (2560, 256, 2304), False), ("gate/up_proj", 13824, 5120, (2560, 256, 2304), False),
                                    ("down_proj", 5120, 13824, (6912, 512, 6400), False)):
        run("qwen tp4", name, N, K, split, 4, M, bias)
# config 5: Mixtral-8x7B experts (w1/w3: 14336 x 4096, w2: 4096 x 14336), TP=8, MXFP4-dominant; M = tokens of one expert
for M in Ms:
    for name, N, K, split in (("w1/w3", 14336, 4096, (3584, 256, 256)), ("w2", 4096, 14336, (12544, 1024, 768))):
        run("mixtral tp8", name, N, K, split, 8, max(1, M // 4), False)
