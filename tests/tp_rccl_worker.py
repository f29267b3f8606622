"""One rank of the RCCL test (tests/test_tp_rccl_gpu.py launches `world` of these through torch.distributed.run, one per GPU,
BEFORE any of them has touched a GPU).  Every rank builds the same full tensors from a seeded CPU generator, runs the three
tensor-parallel layouts of micromix_amd/tp.py over the `nccl` (= RCCL) backend and compares with the unsharded product computed
on its own GPU.  Prints `RCCL-OK rank <r>` and exits 0 on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from micromix_amd import mixedgemm, tp

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    g = torch.Generator().manual_seed(1234)
    # a Llama-3-8B projection at reduced width (N = 1024 of 4096 features, the full K) and a small MLP
    m, n, k, split = 256, 1024, 4096, (2048, 128, 1920)
    x = torch.randn((m, k), generator=g)
    x[:, torch.randperm(k, generator=g)[: k // 100]] *= 20.0
    x = x.to(torch.bfloat16).to(dev)
    w = (torch.randn((n, k), generator=g) * 0.02).to(torch.bfloat16).to(dev)
    bias = torch.randn((n,), generator=g).to(torch.bfloat16).to(dev)
    idx = torch.randperm(k, generator=g).to(torch.int16).to(dev)
    mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)
    qa, qb = mixedgemm.reorder_quantize_x(x, idx, *split), mixedgemm.reorder_quantize_w4(w, idx, *split)
    full = mm(qa, qb, rounding="fused").float()
    bound = lambda ref: (world + 1) * 2.0 ** -9 * float(ref.abs().max()) + 1e-3

    # 1. K-shard + one all-reduce on the bf16 output (the north-star layout), also with fp32 partials
    layer = tp.TPShardedLinear(w, idx, *split, rank=rank, world=world, group=dist.group.WORLD)
    y = layer.matmul_allreduce(layer.quantize_x(x), out=torch.empty((m, n), dtype=torch.bfloat16, device=dev)).float()
    assert float((y - full).abs().max()) <= bound(full), ("K-shard", float((y - full).abs().max()), bound(full))
    gathered = [torch.empty_like(y) for _ in range(world)]
    dist.all_gather(gathered, y)
    assert all(torch.equal(t, gathered[0]) for t in gathered), "ranks disagree after the all-reduce"
    if hasattr(layer, "matmul_allreduce_f32"):
        y32 = layer.matmul_allreduce_f32(layer.quantize_x(x)).float()
        # fp32 partials, one final rounding: within one bf16 ulp of the unsharded fused product whatever the world size
        assert float((y32 - full).abs().max()) <= 2.0 ** -8 * float(full.abs().max()) + 1e-3, ("K-shard fp32", float((y32 - full).abs().max()))

    # 2. column-parallel with gather: bit-identical to the unsharded layer (reference rounding, bias)
    col = tp.ColumnParallelLinear(w, idx, *split, rank=rank, world=world, group=dist.group.WORLD, bias=bias, gather_output=True)
    want = mm(qa, qb) + bias
    got = col(x)
    assert torch.equal(got, want), "column-parallel gather differs from the unsharded layer"

    # 3. Megatron MLP pairing: one all-reduce per MLP
    hid, inter, in_split, down_split = 512, 2048, (256, 128, 128), (1024, 512, 512)
    xs = torch.randn((m, hid), generator=g).to(torch.bfloat16).to(dev)
    wg = (torch.randn((inter, hid), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    wu = (torch.randn((inter, hid), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    wd = (torch.randn((hid, inter), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    idh = torch.randperm(hid, generator=g).to(torch.int16).to(dev)
    qx = mixedgemm.reorder_quantize_x(xs, idh, *in_split)
    gg, uu = mm(qx, mixedgemm.reorder_quantize_w4(wg, idh, *in_split)), mm(qx, mixedgemm.reorder_quantize_w4(wu, idh, *in_split))
    ref = mm(mixedgemm.activate_quantize_x(gg, uu, *down_split), mixedgemm.downproj_quantize_w4(wd, *down_split), rounding="fused").float()
    mlp = tp.TPMLP(wg, wu, wd, idh, in_split, down_split, rank=rank, world=world, group=dist.group.WORLD)
    ym = mlp(xs).float()
    assert float((ym - ref).abs().max()) <= bound(ref), ("TPMLP", float((ym - ref).abs().max()), bound(ref))
    ym32 = mlp(xs, fp32_partials=True).float()
    assert float((ym32 - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-3, ("TPMLP fp32", float((ym32 - ref).abs().max()))
    dist.barrier()
    torch.cuda.synchronize()
    print(f"RCCL-OK rank {rank}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
