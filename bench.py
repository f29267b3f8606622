#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MicroMix mgemm hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ONE mixed-MX GEMM, M = N = K = 4096, all-MXFP8 activations
(p8_num = 4096), weights pre-packed the way QLinearLayer deploys them (MXFP4, "w4" mode:
A fp8 x B fp4 through mixedgemm.matmul).  A "step" is one `mixedgemm.matmul` call on operands
already resident in HBM; `value` = 2*M*N*K*steps / time in TFLOP/s (the reference's own TFLOPS
convention, mgemm/benchmark/mxf4f6f8_bench.cu:165-167).  The same line also carries
  * `roofline`   : the GEMM kernel's per-launch duration from HIP events inside the timed region,
                   against the dense MFMA peak for fp8 operands (MI355X_MICROARCH.md: ~5 PF);
  * `qlinear`    : tokens/s of the full QLinearLayer.forward hot path (reorder_quantize_x + matmul),
                   timed in a second loop of the same length, and the matching-precision "w" mode;
  * `cpu_baseline`: the CPU oracle (a port of the same algorithm; the reference has no CPU path,
                   see BASELINE.md) timed on a bounded row sample on this host.
With --gpus N > 1 (one process per GPU, RCCL): the north-star tensor-parallel path -- each rank holds
a 128-aligned K-shard of every reordered segment (weights AND activation columns), computes a partial
[M, N] product and the partials are summed with one RCCL all-reduce on the bf16 output.  Total work is
fixed ("scaling": "strong"); `value` is still 2*M*N*K*steps / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M = N = K = 4096
SPLIT = (0, 0, 4096)          # (p4_num, p6_num, p8_num): all-MXFP8 activations
PEAK_TFLOPS_FP8 = 5033.0      # 2048 flop/clk/SIMD * 4 SIMD * 256 CU * 2.4 GHz (MI355X_MICROARCH.md, dense)
PEAK_TFLOPS_FP4 = 10066.0


def synth_inputs(seed=0):
    """X ~ N(0,1) bf16 with 1 % outlier channels x20; W ~ N(0, 0.02); reorder index = argsort of the
    per-channel mean |x| (reorder_indices.py:64-69).  torch CPU generator, seed fixed."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((M, K), generator=g)
    cols = torch.randperm(K, generator=g)[: K // 100]
    x[:, cols] *= 20.0
    w = torch.randn((N, K), generator=g) * 0.02
    idx = torch.argsort(x.abs().mean(0)).to(torch.int16)
    return x.to(torch.bfloat16), w.to(torch.bfloat16), idx


def cpu_baseline(x, w, idx, rows=2048, budget_s=12.0):
    """The oracle (a CPU port of QLinearLayer.forward: quantize-x + dequantise + matmul with the reference
    rounding order) on a bounded sample of the same workload: `rows`-token forwards repeated until ~budget_s
    seconds of CPU work have been timed.  Weights are packed outside the timed region, exactly as on the GPU."""
    import torch
    from oracle import mx_oracle as o
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    xb, wb, ib = bits(x[:rows]), bits(w), idx.numpy()
    packed = o.qlinear_pack_weight(wb, ib, *SPLIT, "w4")
    o.qlinear_forward(xb[:8], ib, *SPLIT, packed)               # warm-up
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < budget_s and reps < 64:
        o.qlinear_forward(xb, ib, *SPLIT, packed)
        reps += 1
        dt = time.perf_counter() - t0
    tokens = rows * reps
    return {"value": round(2.0 * tokens * N * K / dt / 1e12, 4), "unit": "TFLOP/s", "cores": int(cores), "kind": "port",
            "tokens_per_s": round(tokens / dt, 1),
            "sample": f"{reps} x {rows} of {M} token rows, full N=K=4096: oracle quantize-x + dequant + fp64 matmul + bf16 "
                      f"rounding per segment, {dt:.1f} s of CPU time"}


def load_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "gemm_traffic.json")
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from micromix_amd import _lib, mixedgemm
    from micromix_amd import tp as tpmod

    _lib.load()  # fail loudly if the HIP library is missing
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MICROMIX_BENCH_BACKEND=gloo is a dry run of the multi-rank code path on a box with fewer GPUs than ranks (ranks share
    # devices, the all-reduce is staged through the host): for testing the script, not a measurement
    backend = os.environ.get("MICROMIX_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    x_cpu, w_cpu, idx_cpu = synth_inputs()
    x, w, idx = x_cpu.to(dev), w_cpu.to(dev), idx_cpu.to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if world == 1:
        b = mixedgemm.reorder_quantize_w4(w, idx, *SPLIT)
        a = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def step():
            mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        parallelism = "single GPU"
    else:
        layer = tpmod.TPShardedLinear(w, idx, *SPLIT, rank=rank, world=world, group=dist.group.WORLD)
        a = layer.quantize_x(x)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def step():
            layer.matmul_allreduce(a, out=out)
        parallelism = f"tp{world}: K-shard (row-parallel) + RCCL all-reduce(bf16 [M,N])"
        extra["tp_shard_columns"] = layer.shard_widths

    for _ in range(args.warmup):
        step()
    barrier()
    # The K timed steps are issued as ONE hipGraph holding K GEMM launches (single GPU): consecutive launches from a stream
    # leave a ~4 us gap at every kernel boundary, a graph about half of that (tools/graph_gap.py: 57-60 vs 54-56 us per step).
    # Capture happens here, outside the timed region; MICROMIX_BENCH_GRAPH=0 times plain stream launches instead.
    graph = None
    if world == 1 and os.environ.get("MICROMIX_BENCH_GRAPH", "1") != "0":
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(args.steps):
                    step()
            graph.replay()              # untimed: first replay uploads the graph
        except Exception as e:          # fall back to stream launches
            print(f"[bench] hipGraph capture failed ({e}); timing stream launches", file=sys.stderr)
            graph = None
    extra_launch = "one hipGraph of K GEMM launches" if graph is not None else "K stream launches"
    # timed region: exactly K steps, nothing else
    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    barrier()
    dt = time.perf_counter() - t0
    # Kernel duration for the roofline: a second pass of the same K steps with HIP events attached to the GEMM dispatch
    # itself (hipExtLaunchKernel start/stop events, mm_diag_set_kernel_events) on the stream the kernel runs on; they
    # bracket exactly what rocprofv3's kernel trace reports.  It is a separate pass because attaching events widens the gap
    # between consecutive launches by ~4 us (it would cost `value` 5-7 %), and events recorded AROUND the call would
    # include that gap in every sample.
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world == 1:
        from micromix_amd import _lib
        lib = _lib.load()
        for e0, e1 in evs:      # torch creates the hipEvent_t on first record
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        for e0, e1 in evs:
            lib.mm_diag_set_kernel_events(e0.cuda_event, e1.cuda_event)
            step()
        lib.mm_diag_set_kernel_events(None, None)
    else:
        for e0, e1 in evs:      # per-step device time of GEMM + all-reduce on this rank's stream
            e0.record()
            step()
            e1.record()
    barrier()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    ms_per_step = dt * 1e3 / args.steps
    flop = 2.0 * M * N * K
    value = flop / (ms_per_step * 1e-3) / 1e12

    result = {
        "metric": "mixed-MX GEMM TFLOPS (Llama-3-8B 4096x4096x4096, all-MXFP8 activations, MXFP4 weights)",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "mxfp8 x mxfp4 -> f32 acc -> bf16",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: single 4096x4096x4096 mixed-MX GEMM, (p4,p6,p8)=(0,0,4096), "
                               "w4 weights (production QLinearLayer mode)",
                   "M": M, "N": N, "K": K, "split": list(SPLIT), "weight_mode": "w4", "parallelism": parallelism,
                   "launch": extra_launch},
    }

    if rank == 0 and world == 1:
        traffic = load_traffic()
        achieved = flop / (kern_ms * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_FP8, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_TFLOPS_FP8, 4),
            "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
            "kernel": "mm::g256::mx_gemm256_kernel<true,false> (fused three-segment scaled-MFMA GEMM, 256x256 tiles)",
            "kernel_us": round(kern_ms * 1e3, 2), "kernel_us_source": "HIP events attached to each GEMM dispatch, second pass of K steps",
            "algorithmic_flop_per_launch": flop,
            "algorithmic_bytes_per_launch": M * K + N * K // 2 + (M + N) * K // 32 + 2 * M * N,
            "note": "peak = dense fp8-operand scaled-MFMA rate; A is fp8 so the fp8 rate applies to the whole launch",
        }
        # second loop of the same length: the full QLinearLayer.forward hot path, and the "w" weight mode
        def timed(fn):
            for _ in range(args.warmup):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / args.steps

        def fwd():
            q = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
            mixedgemm.matmul(q[0], b[0], q[1], b[1], q[2], b[2], q[3], b[3], q[4], b[4], q[5], b[5], out=out)
        t_fwd = timed(fwd)
        fwd_launch = "stream launches"
        if graph is not None:       # the same K forwards as one hipGraph (as the GEMM steps above)
            try:
                gq = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gq):
                    for _ in range(args.steps):
                        fwd()
                gq.replay()
                torch.cuda.synchronize()
                t = time.perf_counter()
                gq.replay()
                torch.cuda.synchronize()
                t_graph = (time.perf_counter() - t) / args.steps
                if t_graph < t_fwd:
                    t_fwd, fwd_launch = t_graph, "one hipGraph of K forwards"
                del gq
            except Exception as e:
                print(f"[bench] hipGraph capture of the forward failed ({e})", file=sys.stderr)
        # quantizer kernel alone: direct C-ABI calls on preallocated outputs (the op-level call spends ~20 us of host
        # time on six allocations, which would hide the 11 us kernel)
        lib = _lib.load()
        qo = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
        pp = lambda t: t.data_ptr() if t.numel() else None
        stream = torch.cuda.current_stream().cuda_stream
        t_q = timed(lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *SPLIT, 0, pp(qo[0]), pp(qo[1]), pp(qo[2]),
                                                    pp(qo[3]), pp(qo[4]), pp(qo[5]), stream))
        bw = mixedgemm.reorder_quantize_w(w, idx, *SPLIT)
        t_w = timed(lambda: mixedgemm.matmul(a[0], bw[0], a[1], bw[1], a[2], bw[2], a[3], bw[3], a[4], bw[4], a[5], bw[5], out=out))
        mixed = (2048, 128, 1920)   # the reference's own bench constants (bench_reorder_gemm.cu:28-30)
        am = mixedgemm.reorder_quantize_x(x, idx, *mixed)
        bm = mixedgemm.reorder_quantize_w4(w, idx, *mixed)
        t_m = timed(lambda: mixedgemm.matmul(am[0], bm[0], am[1], bm[1], am[2], bm[2], am[3], bm[3], am[4], bm[4], am[5], bm[5], out=out))
        q_bytes = 2 * M * K + M * K + M * K // 32 + 2 * K
        result["qlinear"] = {
            "tokens_per_s": round(M / t_fwd, 1), "forward_us": round(t_fwd * 1e6, 2), "forward_launch": fwd_launch,
            "quantize_x_kernel_us": round(t_q * 1e6, 2), "quantize_x_GBps": round(q_bytes / t_q / 1e9, 1),
            "quantize_x_frac_of_8TBps": round(q_bytes / t_q / 8e12, 4),
            "gemm_w_mode_tflops": round(flop / t_w / 1e12, 2),
            "gemm_mixed_2048_128_1920_w4_tflops": round(flop / t_m / 1e12, 2),
        }
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(x_cpu, w_cpu, idx_cpu)
    if world > 1:
        # the same GEMM with rows instead of K split across the GPUs (every rank multiplies its own 4096 token rows by the
        # replicated weights, no exchange): what the node delivers when the linear layer is used data-parallel
        b = mixedgemm.reorder_quantize_w4(w, idx, *SPLIT)
        a = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
        for _ in range(args.warmup):
            mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extra["row_parallel_no_exchange"] = {"value": round(world * flop / (float(t.item()) / args.steps) / 1e12, 2), "unit": "TFLOP/s",
                                             "scaling": "weak", "global_rows": world * M}
    if rank == 0:
        result.update(extra)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
