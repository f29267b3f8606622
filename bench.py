#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MicroMix mgemm hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ONE mixed-MX GEMM, M = N = K = 4096, all-MXFP8 activations
(p8_num = 4096), weights pre-packed the way QLinearLayer deploys them (MXFP4, "w4" mode:
A fp8 x B fp4 through mixedgemm.matmul).  A "step" is one `mixedgemm.matmul` call on operands
already resident in HBM; `value` = 2*M*N*K*steps / time in TFLOP/s (the reference's own TFLOPS
convention, mgemm/benchmark/mxf4f6f8_bench.cu:165-167).

Clocks: the chip ramps its clock over the first ~second of load, and a short run (--steps 20) would be measured on
the ramp.  So, independent of --warmup, an UNTIMED settle phase of back-to-back GEMM launches (>= SETTLE_S seconds,
`prewarm_s` in the line) runs right before the timed region, and every kernel time quoted below is taken in that
same regime.

The line also carries
  * `roofline`   : the GEMM kernel's per-launch duration from HIP events attached to each dispatch, against the dense
                   MFMA peak of the operand precision (MI355X_MICROARCH.md: fp8 operands ~5.03 PF, fp6/fp4 ~10.07 PF);
  * `mixed`      : the same for the mixed (p4, p6, p8) splits of SURVEY.md section 8d, each against its per-precision
                   roofline time t* = sum_seg 2*M*N*K_seg / peak(seg);
  * `qlinear`    : tokens/s of the full QLinearLayer.forward hot path (reorder_quantize_x + matmul);
  * `cpu_baseline`: the CPU port of QLinearLayer.forward (oracle quantizer + dequantise + fp32 torch.matmul with the reference's
                   rounding order; the reference has no CPU path) timed on this host at M in {1, 128, 2048}.
  * `published_config`: the reference's only published configuration (M = 32, N = K = 4096, MXFP6 x MXFP4) timed here;
  * `quantizers`  : rmsnorm_quantize_x / activate_quantize_x / reorder_quantize_x kernel times against the 8 TB/s HBM peak;
  * `power`       : package power / cap / shader clock sampled with rocm-smi DURING the settle phase (the 4096^3 GEMM runs at the
                   1400 W cap: its time is set by energy);
  * `roofline.zero_operands`: context for that -- the SAME launch on operands whose codes are all zero (same instructions, same
                   traffic, multipliers barely switching): kernel time, power and clock when the cap does not bite.  Never `value`.
`value` is always the K stream launches' figure (the same launch mode `roofline.kernel_us` is measured in); the hipGraph replay of
the same K launches is reported beside it as `graph_launch_ms_per_step`.
With --gpus N > 1 (one process per GPU, RCCL): the north-star tensor-parallel path -- each rank holds a 128-aligned K-shard of
every reordered segment (weights AND activation columns), computes a partial [M, N] product and the partials are summed with
one RCCL all-reduce on the bf16 output.  Total work is fixed ("scaling": "strong"); `value` is 2*M*N*K*steps / max-over-ranks
time; `tp` splits a step into `gemm_us` and `allreduce_us` (events on the compute stream) and lists the payload and the
per-rank roofline; `tp_mlp` times the Megatron pairing (gate/up column-parallel -> down row-parallel, ONE all-reduce per MLP).
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
PEAK_TFLOPS_FP4 = 10066.0     # fp6 / fp4 operands: 4096 flop/clk/SIMD
SUSTAINED_MFMA_FP8_FP4 = 4398.0   # measured: register-operand fp8 x fp4 32x32x64 MFMA loop, DVFS-settled (profiles/r02_mfma_shapes.txt)
SETTLE_S = 1.5
# mixed splits of SURVEY.md section 8d (name, M, N, K, split)
MIXED = [("q_o_2048_128_1920", 4096, 4096, 4096, (2048, 128, 1920)), ("q_o_3072_896_128", 4096, 4096, 4096, (3072, 896, 128)),
         ("q_o_all_fp4", 4096, 4096, 4096, (4096, 0, 0)), ("down_12288_1024_1024", 4096, 4096, 14336, (12288, 1024, 1024)),
         ("gate_up_0_0_4096", 4096, 14336, 4096, (0, 0, 4096)), ("gate_up_3072_896_128", 4096, 14336, 4096, (3072, 896, 128)),
         # the matching-precision weight mode (bindings.cpp:74-86 -> gemm.cu:26-51 -> w6a6.cu / w8a8.cu): fp6 x fp6 at the fp4 MFMA rate,
         # fp8 x fp8 at the fp8 rate -- the same per-precision roofline formula, weights packed by reorder_quantize_w
         ("q_o_w_0_0_4096", 4096, 4096, 4096, (0, 0, 4096), "w"), ("q_o_w_2048_128_1920", 4096, 4096, 4096, (2048, 128, 1920), "w")]
# the one configuration the reference publishes a number for (mgemm/README.md:34-46, formula mgemm/benchmark/mxf4f6f8_bench.cu:165-167):
# gemm_host_tn, M = 32, N = K = 4096, MXFP6(E3M2) x MXFP4, 0.19270399 ms = 5.5720 TFLOPs on an RTX 5090 -- context, not a same-node comparison
PUBLISHED = {"M": 32, "N": 4096, "K": 4096, "split": (0, 4096, 0), "reference_ms": 0.19270399, "reference_tflops": 5.5720,
             "reference_hardware": "RTX 5090", "source": "mgemm/README.md:34-46"}


def roofline_time_s(m, n, split):
    """t* = sum over segments of 2*M*N*K_seg / peak(segment operand precision): activations fp4 | fp6 | fp8 against fp4
    weights run at the fp4 / fp4 / fp8 MFMA rate."""
    kn, ks, ko = split
    return 2.0 * m * n * (kn + ks) / (PEAK_TFLOPS_FP4 * 1e12) + 2.0 * m * n * ko / (PEAK_TFLOPS_FP8 * 1e12)


def synth_inputs(seed=0, m=M, n=N, k=K):
    """X ~ N(0,1) bf16 with 1 % outlier channels x20; W ~ N(0, 0.02); reorder index = argsort of the
    per-channel mean |x| (reorder_indices.py:64-69).  torch CPU generator, seed fixed."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((m, k), generator=g)
    cols = torch.randperm(k, generator=g)[: k // 100]
    x[:, cols] *= 20.0
    w = torch.randn((n, k), generator=g) * 0.02
    idx = torch.argsort(x.abs().mean(0)).to(torch.int16)
    return x.to(torch.bfloat16), w.to(torch.bfloat16), idx


def cpu_baseline(x, w, idx, budget_s=18.0):
    """the CPU port of QLinearLayer.forward for N = K = 4096 at M in {1, 128, 2048, 4096} -- oracle
    quantize-x (numpy) + dequantise + fp32 torch.matmul on all host cores + bf16 rounding after each segment (the reference's
    rounding order); the weight is packed AND dequantised once outside the timed region (a CPU fake-quant layer would hold
    it that way).  Median of >= 5 repeats after one warm-up, bounded to ~budget_s of CPU time in all."""
    import torch
    from oracle import mx_oracle as o
    cores = torch.get_num_threads()
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    ib = idx.numpy()
    packed = o.qlinear_pack_weight(bits(w), ib, *SPLIT, "w4")
    wdeq = [torch.from_numpy(d.astype(np.float32)) if d is not None else None for d in o.dequant_operand(packed, "w", "w4")]
    fmts = ("fp4", "fp6", "fp8")

    def forward(xb):
        q = o.reorder_quantize(xb, ib, *SPLIT, "x")
        d = torch.zeros((xb.shape[0], N), dtype=torch.float32)
        for i, kseg in enumerate(SPLIT):
            if kseg:
                a = torch.from_numpy(o.dequant_segment(q[i], q[3 + i], xb.shape[0], kseg, fmts[i], np.float32))
                d = (torch.matmul(a, wdeq[i].t()) + d).to(torch.bfloat16).to(torch.float32)   # D = bf16(acc + D), gemm.cu:75-77
        return d

    by_rows, spent = {}, 0.0
    for rows in (1, 128, 2048, 4096):
        xb = bits(x[:rows])
        forward(xb)
        ts = []
        # median of >= 5 repeats after one warm-up; more (up to 9) while the CPU-time budget lasts
        while len(ts) < 5 or (len(ts) < 9 and spent < budget_s * (0.15 if rows < 2048 else 0.6 if rows < 4096 else 1.0)):
            t0 = time.perf_counter()
            forward(xb)
            ts.append(time.perf_counter() - t0)
            spent += ts[-1]
        t = float(np.median(ts))
        by_rows[str(rows)] = {"ms": round(t * 1e3, 3), "repeats": len(ts), "tokens_per_s": round(rows / t, 1),
                              "tflops": round(2.0 * rows * N * K / t / 1e12, 4)}
    return {"value": by_rows["4096"]["tflops"], "unit": "TFLOP/s", "cores": int(cores), "kind": "port",
            "tokens_per_s": by_rows["4096"]["tokens_per_s"], "by_rows": by_rows,
            "sample": f"QLinearLayer.forward port at M in (1, 128, 2048, 4096) token rows, full N=K=4096: oracle quantize-x + "
                      f"dequantise + fp32 torch.matmul + bf16 rounding per segment, median of >= 5 repeats after one warm-up, "
                      f"{spent:.1f} s of CPU time; `value` is the M=4096 figure (the headline GEMM's own shape)"}


def sample_power(out, delay_s=0.6):
    """rocm-smi in a CHILD process (this process never execs), `delay_s` after the call: package power, cap and clocks while
    the caller keeps the GPU busy.  Fills `out` (a dict); silent when rocm-smi is absent."""
    import subprocess
    import threading

    def run():
        try:
            time.sleep(delay_s)
            # rocm-smi is a `#!/usr/bin/env python3` script: under rocprofv3 the child would inherit the profiler's preload, and with a
            # --pmc pass that preload initialises the GPU before the env -> python3 exec, which this pool forbids.  So: the script
            # itself under this interpreter, with the profiler's variables stripped; and no sampling at all under a counter pass.
            env = {k: v for k, v in os.environ.items()
                   if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB") and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER"))}
            if any(k.startswith("ROCPROF") and "PMC" in k for k in os.environ) or os.environ.get("ROCPROF_COUNTERS"):
                out["error"] = "skipped under a rocprofv3 counter pass"
                return
            script = "/opt/rocm/libexec/rocm_smi/rocm_smi.py"
            cmd = [sys.executable, script] if os.path.exists(script) else ["rocm-smi"]
            txt = subprocess.run(cmd + ["--showpower", "--showmaxpower", "--showclocks", "--json"], capture_output=True,
                                 text=True, timeout=20, env=env).stdout
            card = next(iter(json.loads(txt[txt.index("{"):]).values()))
            for k, v in card.items():
                kl = k.lower()
                if "max graphics package power" in kl:
                    out["cap_w"] = float(v)
                elif "package power" in kl:
                    out["package_w"] = float(v)
                elif kl.startswith("sclk clock speed"):
                    out["sclk_mhz"] = float(str(v).strip("()Mhz ").split("Mhz")[0])
        except Exception as e:                      # telemetry only: never fail the benchmark
            out["error"] = str(e)[:80]

    t = threading.Thread(target=run, daemon=True)
    t.start()
    return t


def llama_layer(dev, lib, mixedgemm, x, steps):
    """BASELINE.json metric (ii): qLinear tokens/s on the Llama-3-8B shapes -- the seven linears of ONE decoder layer (q, k, v, o, gate,
    up, down; hidden 4096, kv 1024, intermediate 14336; model/qLlamaLayer.py:265-269,377-387) with the quantizers between them, as this
    library runs them: q / k / v share one `rmsnorm_quantize_x` and are one GEMM over the concatenated weights (FusedQLinear), gate +
    up + silu * up + the quantization for down_proj are one launch (`gate_up_activate`).  Attention, rope and the residual adds are
    outside SURVEY.md section 8 and are NOT in the timed region (o_proj reads a fixed tensor).  Random-init weights, synthetic
    activations; splits (2048,128,1920) for the hidden inputs and (12288,1024,1024) for down_proj."""
    import torch
    H, NKV, I = 4096, 1024, 14336
    in_split, down_split = (2048, 128, 1920), (12288, 1024, 1024)
    g = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda r, c: (torch.randn((r, c), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    idx = torch.argsort(x.float().abs().mean(0)).to(torch.int16)
    pack = lambda w: mixedgemm.reorder_quantize_w4(w, idx, *in_split)
    mm = lambda a, b, **kw: mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], **kw)

    def make_layer(with_cat=False):
        """one decoder layer's packed weights (116 MB): q | k | v, o, gate | up interleaved per 128 features, down"""
        L = {"qkv": pack(torch.cat([rnd(H, H), rnd(NKV, H), rnd(NKV, H)], 0)), "o": pack(rnd(H, H))}
        pg, pu = pack(rnd(I, H)), pack(rnd(I, H))
        L["gu"] = mixedgemm.interleave_gate_up(pg, pu)
        if with_cat:
            L["gu_cat"] = tuple(torch.cat((a, b), 0).contiguous() for a, b in zip(pg, pu))
        del pg, pu
        L["down"] = mixedgemm.downproj_quantize_w4(rnd(H, I), *down_split)
        return L

    layer0 = make_layer(with_cat=True)
    w_qkv, w_o, w_gu, w_gu_cat, w_down = layer0["qkv"], layer0["o"], layer0["gu"], layer0["gu_cat"], layer0["down"]
    normw = torch.ones((H,), dtype=torch.bfloat16, device=dev)

    def prefill(xm, attn):
        qa = mixedgemm.rmsnorm_quantize_x(xm, normw, 1e-5, idx, *in_split)
        mm(qa, w_qkv)
        qo = mixedgemm.reorder_quantize_x(attn, idx, *in_split)
        o = mm(qo, w_o)
        qm = mixedgemm.rmsnorm_quantize_x(o, normw, 1e-5, idx, *in_split)
        qh = mixedgemm.gate_up_activate(qm, w_gu, *down_split)
        return mm(qh, w_down)

    def decode(xm, attn, L=None):
        """the layer as a model runs it at M <= 8 (round 5): input_layernorm + q | k | v in ONE launch (rmsnorm_qlinear_decode), o_proj
        (quantize + GEMM in one launch), post_attention_layernorm + gate | up in ONE launch, down_proj with silu * up + its quantization
        inside: four launches, norms included"""
        L = layer0 if L is None else L
        m = xm.size(0)
        if mixedgemm.rmsnorm_qlinear_decode_supported(m, H + 2 * NKV, *in_split) == 2:
            mixedgemm.rmsnorm_qlinear_decode(xm, normw, 1e-5, idx, *L["qkv"], *in_split)      # norm + quantize + GEMM in one launch
        else:
            mm(mixedgemm.rmsnorm_quantize_x(xm, normw, 1e-5, idx, *in_split), L["qkv"])
        if mixedgemm.qlinear_decode_supported(m, H, *in_split):
            o = mixedgemm.qlinear_decode(attn, idx, *L["o"], *in_split)
        else:
            o = mm(mixedgemm.reorder_quantize_x(attn, idx, *in_split), L["o"])
        if mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, I, *in_split) == 2:
            # round 6: norm + quantize + gate | up GEMM + silu * up + the quantization for down_proj in ONE launch, down_proj a plain GEMM
            return mm(mixedgemm.rmsnorm_gate_up_activate_decode(o, normw, 1e-5, idx, L["gu"], *down_split), L["down"])
        fused_gu = mixedgemm.rmsnorm_qlinear_decode_supported(m, 2 * I, *in_split) == 2
        if fused_gu and mixedgemm.down_activate_decode_supported(m, H, *down_split) == 2:
            gub = mixedgemm.rmsnorm_qlinear_decode(o, normw, 1e-5, idx, *L["gu"], *in_split)   # norm + quantize + gate | up GEMM in one launch ...
            return mixedgemm.down_activate_decode(gub, L["down"], *down_split)                  # ... and down_proj with silu * up + its quantization inside
        qm = mixedgemm.rmsnorm_quantize_x(o, normw, 1e-5, idx, *in_split)
        qh = mixedgemm.gate_up_activate(qm, L["gu"], *down_split)        # M <= 64: GEMM into scratch + the quantizer on it
        return mm(qh, L["down"])

    def launches(m):
        if m > 64:
            return 7
        n = 1 if mixedgemm.rmsnorm_qlinear_decode_supported(m, H + 2 * NKV, *in_split) == 2 else 2
        n += 1 if mixedgemm.qlinear_decode_supported(m, H, *in_split) else 2
        if mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, I, *in_split) == 2:
            return n + 2
        if mixedgemm.rmsnorm_qlinear_decode_supported(m, 2 * I, *in_split) == 2 and mixedgemm.down_activate_decode_supported(m, H, *down_split) == 2:
            return n + 2
        one = "stream" in lib.mm_gate_up_activate_describe(m, I).decode()
        return n + (3 if one else 4)            # rmsnorm_quantize_x, gate | up GEMM (+ activation quantizer: one launch on wide layers at M <= 16), down GEMM

    def measure(fn, xm, attn, reps):
        for _ in range(3):
            fn(xm, attn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            fn(xm, attn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(xm, attn)
        torch.cuda.synchronize()
        t_stream = (time.perf_counter() - t0) / reps
        t_graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn(xm, attn)
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    fn(xm, attn)
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gr.replay()
            torch.cuda.synchronize()
            t_graph = (time.perf_counter() - t0) / reps
            del gr
        except Exception as e:
            print(f"[bench] hipGraph capture of the decoder layer failed ({e})", file=sys.stderr)
        return t_stream, t_graph

    out = {"model": "Llama-3-8B decoder layer, linears + quantizers only (attention / rope / residuals outside the hot path, not timed)",
           "linears": {"q/k/v": [H + 2 * NKV, H], "o": [H, H], "gate/up": [2 * I, H], "down": [H, I]},
           "in_split": list(in_split), "down_split": list(down_split), "by_rows": {}}
    flop_per_row = 2.0 * (H * (H + 2 * NKV) + H * H + 2 * I * H + H * I)
    for m in (4096, 8, 1):
        xm = x[:m].contiguous()
        attn = (x[:m] * 0.5).contiguous()
        fn = prefill if m > 64 else decode
        ts, tg = measure(fn, xm, attn, max(5, steps if m > 64 else 4 * steps))
        ent = {"launches_per_layer": launches(m),
               "us_per_layer_stream": round(ts * 1e6, 1), "us_per_layer_graph": round(tg * 1e6, 1) if tg else None,
               "tokens_per_s_stream": round(m / ts, 1), "tokens_per_s_graph": round(m / tg, 1) if tg else None}
        # `tokens_per_s` by a FIXED rule, not the better of the two: prefill-sized batches (M > 64) are GPU-bound and run as stream
        # launches; decode-sized batches are host-bound from Python and are deployed as one hipGraph of the layer's launches
        use_graph = m <= 64 and tg is not None
        t_rule = tg if use_graph else ts
        ent.update({"tokens_per_s": round(m / t_rule, 1),
                    "tokens_per_s_mode": "one hipGraph of the layer's launches (M <= 64)" if use_graph else "stream launches (M > 64)",
                    "tflops": round(flop_per_row * m / t_rule / 1e12, 2)})
        out["by_rows"][str(m)] = ent
    # ---- decode from HBM: NL layers' weight sets (NL x 116 MB > the 256 MiB Infinity Cache) in rotation inside ONE hipGraph, as consecutive
    # layers of a model present them; against 116 MB per layer at 8 TB/s ----
    NL, ROUNDS = 4, 3
    layers = [layer0] + [make_layer() for _ in range(NL - 1)]
    wbytes = sum(t.numel() for k in ("qkv", "o", "gu", "down") for t in layer0[k])
    out["hbm"] = {}
    out["hbm_note"] = (f"{NL} layers' packed weights ({NL * wbytes / 1e6:.0f} MB) in rotation, {ROUNDS} x {NL} layers per hipGraph replay; "
                       f"hbm_frac = ({wbytes / 1e6:.1f} MB per layer / 8 TB/s) / time per layer")
    for m in (8, 1):
        xm, attn = x[:m].contiguous(), (x[:m] * 0.5).contiguous()
        try:
            for L in layers:
                decode(xm, attn, L)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                decode(xm, attn, layers[0])
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(ROUNDS):
                    for L in layers:
                        decode(xm, attn, L)
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                gr.replay()
            torch.cuda.synchronize()
            t_layer = (time.perf_counter() - t0) / (reps * ROUNDS * NL)
            out["hbm"][str(m)] = {"us_per_layer": round(t_layer * 1e6, 1), "tokens_per_s": round(m / t_layer, 1),
                                  "weight_bytes_per_layer": wbytes, "hbm_frac": round(wbytes / 8e12 / t_layer, 4)}
            del gr
        except Exception as e:
            print(f"[bench] decode-from-HBM measurement failed at M = {m} ({e})", file=sys.stderr)
    del layers
    # the MLP at M = 4096 both ways (events around 10 back-to-back repetitions; quantize_x excluded: it is the same launch in both)
    xm = x[:4096].contiguous()
    qm = mixedgemm.reorder_quantize_x(xm, idx, *in_split)
    gu = torch.empty((4096, 2 * I), dtype=torch.bfloat16, device=dev)

    def three_op():
        mm(qm, w_gu_cat, out=gu)                                          # gate | up as one launch (two separate GEMMs take the same time)
        qh = mixedgemm.activate_quantize_x(ga, gb, *down_split)           # (reads contiguous copies of the two halves)
        return mm(qh, w_down)

    mm(qm, w_gu_cat, out=gu)
    ga, gb = gu[:, :I].contiguous(), gu[:, I:].contiguous()

    def fused():
        return mm(mixedgemm.gate_up_activate(qm, w_gu, *down_split), w_down)

    def ev_time(fn, reps=10):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:        # the decode-sized measurements above let the clocks drop: settle first
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    t3, tf = ev_time(three_op), ev_time(fused)
    out["mlp_M4096"] = {
        "three_op_us": round(t3, 1), "fused_us": round(tf, 1),
        "three_op": "one GEMM over gate | up (N = 28672, bf16 out) -> activate_quantize_x -> down GEMM",
        "fused": "gate_up_activate (gate + up + silu*up + quantize in one launch) -> down GEMM; bit-identical operands for down_proj",
        "kernel": lib.mm_gate_up_activate_describe(4096, I).decode(),
        "hbm_bytes_not_moved": 2 * 2 * 4096 * 2 * I}
    return out


def load_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "gemm_traffic.json")
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


LINE_LIMIT = 6144     # the driver keeps the tail of stdout: the ONE JSON line must fit


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a rendezvous in the environment: start the N ranks as a CHILD process
    (`python -m torch.distributed.run ... bench.py --gpus N ...`), relay rank 0's JSON line and return the child's exit code.
    The parent has not touched the GPU (torch is not even imported) and never execs."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: torch.distributed.run picks its own free rendezvous port (a port found here by bind-then-close could be taken by
    # another job on the box before the child binds it: ADVICE r5)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + list(argv)
    print("[bench] launching the ranks: " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def _num(v, nd=2):
    return None if v is None else round(float(v), nd)


def compact_line(r, details_path=None):
    """The ONE line the driver records: the contract keys as they are, every extra reduced to numbers.  The prose (`note`, `kernel`,
    `timing`, sources) and the full per-case dictionaries go to the details file (--details)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "prewarm_s", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "graph_launch_ms_per_step", "stream_launch_ms_per_step")
    c = {k: r[k] for k in keep if k in r}
    cfg = r.get("config", {})
    c["config"] = {"workload": "BASELINE.json configs[1]: 4096^3 mixed-MX GEMM, (p4,p6,p8)=(0,0,4096), w4 weights",
                   "M": cfg.get("M"), "N": cfg.get("N"), "K": cfg.get("K"), "split": cfg.get("split"), "parallelism": cfg.get("parallelism")}
    if "power" in r:
        pw = r["power"]
        c["power"] = {"package_w": pw.get("package_w"), "cap_w": pw.get("cap_w"), "sclk_mhz": pw.get("sclk_mhz")}
    if "roofline" in r:
        rf = r["roofline"]
        c["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us",
                                                "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch", "frac_of_sustained_mfma_only")}
        c["roofline"]["kernel"] = (rf.get("kernel") or "")[:60]
        if "zero_operands" in rf:
            z = rf["zero_operands"]
            c["roofline"]["zero_operands"] = {"kernel_us": z.get("kernel_us"), "frac_of_peak": z.get("frac_of_peak"),
                                              "package_w": z.get("power", {}).get("package_w"), "sclk_mhz": z.get("power", {}).get("sclk_mhz")}
    if "cpu_baseline" in r:
        cb = r["cpu_baseline"]
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "tokens_per_s": cb.get("tokens_per_s"),
                             "sample": "QLinearLayer.forward CPU port (oracle quantize-x + dequantise + fp32 matmul + bf16 per segment), "
                                       "M=4096 N=K=4096, median of >=5 repeats"}
    if "mixed" in r:       # name: [kernel_us, frac of the per-precision roofline]
        c["mixed"] = {k: [v["kernel_us"], v["frac"]] for k, v in r["mixed"].items()}
        c["mixed_kernels"] = sorted({v["kernel"].split(" x ")[0].replace("mm::", "").replace("::mx_gemm256_kernel", "") for v in r["mixed"].values()})
    if "few_tiles" in r:
        c["few_tiles"] = {k: v["kernel_us"] for k, v in r["few_tiles"].items()}
    if "decode" in r:      # name: [us per launch, TB/s of weight bytes]
        c["decode"] = {k: [v["us_per_launch"], v["weight_stream_TBps"]] for k, v in r["decode"].items()}
    for key in ("small_m", "small_m_hbm"):
        if key in r:       # name: [us per launch, fraction of 8 TB/s on the weight bytes]
            c[key] = {k: [v["us_per_launch"], _num(v["weight_stream_TBps"] / 8.0, 3)] for k, v in r[key].items() if isinstance(v, dict)}
    if "quantizers" in r:  # name: [kernel_us, fraction of 8 TB/s, back-to-back step_us]
        c["quantizers"] = {k: [v["kernel_us"], v["frac_of_8TBps"], v["step_us"]] for k, v in r["quantizers"].items() if isinstance(v, dict)}
    if "qlinear" in r:
        q = r["qlinear"]
        c["qlinear"] = {k: q.get(k) for k in ("tokens_per_s", "forward_us", "forward_us_graph", "quantize_x_kernel_us", "quantize_x_frac_of_8TBps",
                                             "gemm_w_mode_kernel_us")}
    if "llama_layer" in r:
        ll = r["llama_layer"]
        # (since round 5 the decode figures include both RMSNorms: `norms` says so -- not comparable with the round-4 keys of the same name)
        c["llama_layer"] = {m: {"us": e["us_per_layer_graph"] if (int(m) <= 64 and e.get("us_per_layer_graph")) else e["us_per_layer_stream"],
                                "tokens_per_s": e["tokens_per_s"], "launches": e["launches_per_layer"]}
                            for m, e in ll.get("by_rows", {}).items()}
        for m, e in ll.get("hbm", {}).items():          # decode from HBM: weight sets of several layers in rotation
            c["llama_layer"].setdefault(m, {}).update({"hbm_us": e["us_per_layer"], "hbm_frac": e["hbm_frac"]})
        c["llama_layer"]["norms"] = "inside (M <= 8)"
        if "mlp_M4096" in ll:
            c["llama_layer"]["mlp_M4096"] = {"three_op_us": ll["mlp_M4096"]["three_op_us"], "fused_us": ll["mlp_M4096"]["fused_us"]}
    if "published_config" in r:
        c["published_config"] = [r["published_config"]["us_per_launch"], r["published_config"]["tflops"]]
    if "tp" in r:
        t = r["tp"]
        c["tp"] = {"gemm_us": t["gemm_us_max_over_ranks"], "allreduce_us": t["allreduce_us_max_over_ranks"],
                   "allreduce_payload_bytes": t["allreduce_payload_bytes"], "rank0_frac": t["rank0_roofline"].get("frac")}
    if "row_parallel_no_exchange" in r:
        c["row_parallel_no_exchange"] = {k: r["row_parallel_no_exchange"][k] for k in ("value", "unit", "global_rows")}
    if "tp_mlp" in r:
        c["tp_mlp"] = {"mlp_us": r["tp_mlp"]["mlp_us"], "tflops": r["tp_mlp"]["tflops"], "allreduce_payload_bytes": r["tp_mlp"]["allreduce_payload_bytes"]}
    c["legend"] = {"mixed": "[kernel_us, frac of per-precision MFMA roofline]", "decode": "[us, TB/s of weight bytes]",
                   "small_m*": "[us, frac of 8 TB/s on weight bytes]", "quantizers": "[kernel_us, frac of 8 TB/s, step_us]",
                   "published_config": "[us, TFLOP/s]"}
    if details_path:
        c["details"] = details_path
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:          # never lose the contract keys to the driver's tail: drop extras from the back
        for k in ("legend", "published_config", "few_tiles", "small_m", "decode", "qlinear", "mixed_kernels", "small_m_hbm", "quantizers", "llama_layer"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    if len(line) >= LINE_LIMIT:          # still too long (a large `mixed` / `tp` section): only the contract keys, `roofline` and `cpu_baseline` stay
        for k in ("tp_mlp", "row_parallel_no_exchange", "mixed", "tp", "power", "details"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    assert len(line) < LINE_LIMIT, f"the bench line has {len(line)} bytes, the driver's record keeps {LINE_LIMIT}"
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline GEMM only (profiling runs)")
    ap.add_argument("--details", default=os.path.join("gpurun_out", "bench_details.json"),
                    help="file for the full result (every note / kernel name / per-case dictionary); the printed line carries numbers only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from micromix_amd import _lib, mixedgemm
    from micromix_amd import tp as tpmod

    lib = _lib.load()  # fail loudly if the HIP library is missing
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

    def mm(a, b, out):
        return mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5], out=out)

    def settle(fn, seconds=SETTLE_S):
        """untimed: back-to-back launches for at least `seconds` so that the clock the timed region sees is the sustained one.
        With several ranks the steps contain collectives, so every rank must run the SAME number of batches: the decision to
        go on is taken on the maximum of the ranks' elapsed times (one tiny all-reduce per batch)."""
        t0 = time.perf_counter()
        while True:
            for _ in range(100):
                fn()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            if elapsed >= seconds:
                return elapsed

    def kernel_us(fn, steps, stat=np.mean):
        """mean duration of the tiled-GEMM dispatch inside fn(): HIP events attached to the dispatch itself (hipExtLaunchKernel
        start/stop events, mm_diag_set_kernel_events) on the stream the kernel runs on -- they bracket exactly what rocprofv3's
        kernel trace reports.  (Events recorded AROUND the call would include the ~4 us launch gap in every sample.)"""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for e0, e1 in evs:      # torch creates the hipEvent_t on first record
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        for e0, e1 in evs:
            lib.mm_diag_set_kernel_events(e0.cuda_event, e1.cuda_event)
            fn()
        lib.mm_diag_set_kernel_events(None, None)
        torch.cuda.synchronize()
        return float(stat([e0.elapsed_time(e1) for e0, e1 in evs])) * 1e3

    extra = {}
    if world == 1:
        b = mixedgemm.reorder_quantize_w4(w, idx, *SPLIT)
        a = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def step():
            mm(a, b, out)
        parallelism = "single GPU"
    else:
        layer = tpmod.TPShardedLinear(w, idx, *SPLIT, rank=rank, world=world, group=dist.group.WORLD)
        a = layer.quantize_x(x)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def step():
            layer.matmul_allreduce(a, out=out)
        parallelism = f"tp{world}: K-shard (row-parallel) + RCCL all-reduce(bf16 [M,N])"

    for _ in range(args.warmup):
        step()
    barrier()
    # The same K steps are also captured as ONE hipGraph holding K GEMM launches (single GPU) and replayed after the timed
    # region as an auxiliary figure (consecutive launches from a stream leave a ~4 us gap at every kernel boundary, a graph about
    # half of that on most boxes: tools/graph_gap.py).  Capture happens here, outside any timed region; MICROMIX_BENCH_GRAPH=0 skips it.
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
    power = {}
    sampler = sample_power(power) if rank == 0 and world == 1 else None
    prewarm_s = settle(step)
    if sampler is not None:
        sampler.join(timeout=25)
        settle(step, 0.3)           # (rocm-smi may have outlasted the settle phase)
    # timed region: exactly K steps as plain stream launches, nothing else
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    extra_launch = "K stream launches"
    # the same K steps once more as ONE hipGraph replay, timed the same way (auxiliary figure: which of the two modes is
    # faster differs between devices of the pool -- graph 55.6 vs stream 53.9 us on one box, 54 vs 57-60 on another -- so
    # `value` is pinned to the stream launches, the mode roofline.kernel_us is measured in)
    graph_ms = None
    stream_ms = dt * 1e3 / args.steps
    if graph is not None:
        barrier()
        t1 = time.perf_counter()
        graph.replay()
        barrier()
        graph_ms = (time.perf_counter() - t1) * 1e3 / args.steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    flop = 2.0 * M * N * K
    value = flop / (ms_per_step * 1e-3) / 1e12

    result = {
        "metric": "mixed-MX GEMM TFLOPS (Llama-3-8B 4096x4096x4096, all-MXFP8 activations, MXFP4 weights)",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "prewarm_s": round(prewarm_s, 2),
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "mxfp8 x mxfp4 -> f32 acc -> bf16",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: single 4096x4096x4096 mixed-MX GEMM, (p4,p6,p8)=(0,0,4096), "
                               "w4 weights (production QLinearLayer mode)",
                   "M": M, "N": N, "K": K, "split": list(SPLIT), "weight_mode": "w4", "parallelism": parallelism,
                   "launch": extra_launch},
    }
    if graph_ms is not None:
        result["graph_launch_ms_per_step"] = round(graph_ms, 5)
    result["stream_launch_ms_per_step"] = round(stream_ms, 5)
    if power:
        result["power"] = dict(power, sampled="rocm-smi during the settle phase (back-to-back GEMM launches)")

    if rank == 0 and world == 1:
        traffic = load_traffic()
        kus = kernel_us(step, args.steps)
        achieved = flop / (kus * 1e-6) / 1e12
        wmode = 1   # MM_W_FP4
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_FP8, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_TFLOPS_FP8, 4),
            "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
            "traffic_source": (f"profiles/gemm_traffic.json ({traffic.get('collected', 'rocprofv3 --pmc passes')}); "
                               "NOT measured in this run") if traffic else None,
            "kernel": lib.mm_matmul_describe(M, N, *SPLIT, wmode, 0, 0).decode(),
            "kernel_us": round(kus, 2), "kernel_us_source": "HIP events attached to each GEMM dispatch, K stream launches after the settle phase",
            "algorithmic_flop_per_launch": flop,
            "algorithmic_bytes_per_launch": M * K + N * K // 2 + (M + N) * K // 32 + 2 * M * N,
            "note": "peak = dense fp8-operand scaled-MFMA rate; A is fp8 so the fp8 rate applies to the whole launch",
            "pool_spread": "the same binary measures 48.2-53.4 us on this pool's boxes (they hold 1.94-2.11 GHz at the 1400 W cap): "
                           "round-over-round changes of `frac` below ~8 % on this launch are box variance, not code (`zero_operands` is the box-independent figure)",
            # context, not part of the contract: what a loop of nothing but register-operand fp8 x fp4 MFMAs sustains on this chip
            # at its 1400 W package cap (tools/mfma_energy.py, profiles/r02_mfma_shapes.txt) -- the tiled GEMM runs AT that cap
            # (`power`), see DESIGN.md section 4.2
            "sustained_mfma_only_tflops": SUSTAINED_MFMA_FP8_FP4, "frac_of_sustained_mfma_only": round(achieved / SUSTAINED_MFMA_FP8_FP4, 4),
        }
        if not args.no_extras:
            # context, not the headline: the SAME launch on operands whose codes are all zero -- the same instructions and memory
            # traffic, but the multipliers barely switch, so the package stays under its power cap and the clock at its maximum.
            # The gap to `kernel_us` is what the power cap costs on the bench's random data (tools/gemm_data_power.py).
            az = mixedgemm.reorder_quantize_x(torch.zeros_like(x), idx, *SPLIT)
            bz = mixedgemm.reorder_quantize_w4(torch.zeros_like(w), idx, *SPLIT)
            fz = lambda: mm(az, bz, out)
            pz = {}
            thz = sample_power(pz, delay_s=0.4)
            settle(fz, 1.0)
            thz.join(timeout=25)
            kz = kernel_us(fz, args.steps)
            result["roofline"]["zero_operands"] = {
                "kernel_us": round(kz, 2), "tflops": round(flop / (kz * 1e-6) / 1e12, 1), "frac_of_peak": round(flop / (kz * 1e-6) / 1e12 / PEAK_TFLOPS_FP8, 4),
                "power": pz, "note": "same launch, all-zero operand codes: not power-capped (context for `power` / `frac`; never `value`)"}
            del az, bz
            settle(step, 0.5)      # back to the bench data's clock / power state for what follows
    if rank == 0 and world == 1 and not args.no_extras:
        # ---- mixed splits against their per-precision rooflines (t* = sum_seg 2*M*N*K_seg / peak(seg)) ----
        mixed = {}
        for name, mm_, nn_, kk_, split, *wm in MIXED:
            w_match = bool(wm) and wm[0] == "w"
            if (mm_, nn_, kk_) == (M, N, K):
                xs, ws, ids = x, w, idx
            else:
                xc, wc, ic = synth_inputs(1, mm_, nn_, kk_)
                xs, ws, ids = xc.to(dev), wc.to(dev), ic.to(dev)
            am = mixedgemm.reorder_quantize_x(xs, ids, *split)
            bm = (mixedgemm.reorder_quantize_w if w_match else mixedgemm.reorder_quantize_w4)(ws, ids, *split)
            om = out if (mm_, nn_) == (M, N) else torch.empty((mm_, nn_), dtype=torch.bfloat16, device=dev)
            f = lambda: mm(am, bm, om)
            settle(f, 0.4)
            # the weight mode mixedgemm.matmul derives from the tensor shapes (bindings.cpp:74): with KS = KO = 0 the S / O weights
            # compare equal (both empty) -> MM_W_MATCH, which the library runs on the fp4-weight kernels (capi.hip: weights_fp4)
            wmode_call = 0 if (w_match or (split[1] == 0 and split[2] == 0)) else 1
            desc = lib.mm_matmul_describe(mm_, nn_, *split, wmode_call, 0, 0).decode()
            if " + " in desc:       # two launches (tail balancing): events around K back-to-back calls, launch gaps included
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    f()
                e1.record()
                torch.cuda.synchronize()
                us, how = e0.elapsed_time(e1) * 1e3 / args.steps, "events around K back-to-back calls (two launches per call, gaps included)"
            else:
                us, how = kernel_us(f, args.steps), "HIP events attached to the dispatch"
            tstar = roofline_time_s(mm_, nn_, split) * 1e6
            mixed[name] = {"M": mm_, "N": nn_, "K": kk_, "split": list(split), "kernel_us": round(us, 2), "kernel_us_source": how,
                           "tflops": round(2.0 * mm_ * nn_ * kk_ / us / 1e6, 1), "roofline_us": round(tstar, 2),
                           "frac": round(tstar / us, 4), "kernel": desc, "weight_mode": "w" if w_match else "w4"}
            del am, bm
        result["mixed"] = mixed

        # ---- launches with few tiles: the first M rows of the same activations against the q/o weights (64-row tiles) ----
        few, fsplit = {}, (2048, 128, 1920)
        bf = mixedgemm.reorder_quantize_w4(w, idx, *fsplit)
        for m_ in (128, 256, 512):
            af = mixedgemm.reorder_quantize_x(x[:m_].contiguous(), idx, *fsplit)
            of = torch.empty((m_, N), dtype=torch.bfloat16, device=dev)
            f = lambda: mm(af, bf, of)
            settle(f, 0.3)
            us = kernel_us(f, args.steps, np.median)     # 12 us kernels: one late dispatch would move a mean by 10 %
            few[f"q_o_M{m_}"] = {"M": m_, "N": N, "K": K, "split": list(fsplit), "kernel_us": round(us, 2), "kernel_us_stat": "median",
                                 "tflops": round(2.0 * m_ * N * K / us / 1e6, 1),
                                 "kernel": lib.mm_matmul_describe(m_, N, *fsplit, 1, 0, 0).decode()}
        # k/v projection (N = 1024): at M = 128 its 32 tiles take the split-K that reduces inside the launch (mixedgemm.matmul keeps
        # the zeroed ticket workspace); the same shape with split_k=False beside it
        wkv = w[:1024].contiguous()
        bkv = mixedgemm.reorder_quantize_w4(wkv, idx, *fsplit)
        for m_ in (128, 256):
            akv = mixedgemm.reorder_quantize_x(x[:m_].contiguous(), idx, *fsplit)
            okv = torch.empty((m_, 1024), dtype=torch.bfloat16, device=dev)
            ent = {"M": m_, "N": 1024, "K": K, "split": list(fsplit), "kernel_us_stat": "median"}
            for key, kw in (("kernel_us", {}), ("kernel_us_unsplit", {"split_k": False})):
                f = lambda: mixedgemm.matmul(akv[0], bkv[0], akv[1], bkv[1], akv[2], bkv[2], akv[3], bkv[3], akv[4], bkv[4], akv[5], bkv[5], out=okv, **kw)
                settle(f, 0.2)
                ent[key] = round(kernel_us(f, args.steps, np.median), 2)
            need = lib.mm_matmul_workspace_bytes(m_, 1024, *fsplit, 1, 4)
            ent["kernel"] = lib.mm_matmul_describe(m_, 1024, *fsplit, 1, 4, need).decode()
            few[f"k_v_M{m_}"] = ent
        del bkv
        # ---- decode: one token through QLinearLayer.forward as ONE launch (mm_qlinear_decode: quantize + GEMM fused, M <= 8) ----
        dec = {}
        xd = x[:1].contiguous()
        od = torch.empty((1, N), dtype=torch.bfloat16, device=dev)
        stream_ptr = torch.cuda.current_stream().cuda_stream
        wp = [t.data_ptr() if t.numel() else None for t in bf]
        fd = lambda: lib.mm_qlinear_decode(xd.data_ptr(), idx.data_ptr(), *wp, 1, N, *fsplit, 1, 0, None, od.data_ptr(), stream_ptr)
        assert fd() == 0
        settle(fd, 0.2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fd()
        torch.cuda.synchronize()
        t_dec = (time.perf_counter() - t0) / 200
        wbytes = N * K // 2 + N * K // 32
        dec["q_o_M1"] = {"M": 1, "N": N, "K": K, "split": list(fsplit), "us_per_launch": round(t_dec * 1e6, 2),
                         "weight_stream_TBps": round(wbytes / t_dec / 1e12, 3),
                         "note": "back-to-back launches through the C ABI (direct ctypes calls)"}
        # the largest decode GEMMs of a Llama-3-8B layer: gate / up (N = 14336), M = 1, the shape QLinearLayer sends to the fused kernel
        ng = 14336
        wg = (torch.randn((ng, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        bg = mixedgemm.reorder_quantize_w4(wg, idx, *fsplit)
        del wg
        og = torch.empty((1, ng), dtype=torch.bfloat16, device=dev)
        wpg = [t.data_ptr() if t.numel() else None for t in bg]
        fg = lambda: lib.mm_qlinear_decode(xd.data_ptr(), idx.data_ptr(), *wpg, 1, ng, *fsplit, 1, 0, None, og.data_ptr(), stream_ptr)
        assert fg() == 0
        settle(fg, 0.2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fg()
        torch.cuda.synchronize()
        t_g = (time.perf_counter() - t0) / 200
        gbytes = ng * K // 2 + ng * K // 32
        dec["gate_up_M1"] = {"M": 1, "N": ng, "K": K, "split": list(fsplit), "us_per_launch": round(t_g * 1e6, 2),
                             "weight_stream_TBps": round(gbytes / t_g / 1e12, 3),
                             "note": "mm_qlinear_decode: quantize + GEMM in one launch; 448 workgroups of the weight-streaming kernel, each quantizes the row into LDS while its first weight slabs are in flight"}
        del bg
        result["decode"] = dec
        # ---- the largest GEMMs of a decoder layer at M = 16 / 32 / 64 (speculative / batched decode): weight bytes against 8 TB/s ----
        sm = {}
        for name, nn_, kk_, sp in (("gate_up", 14336, K, fsplit), ("gate_up_fused", 28672, K, fsplit), ("down", 4096, 14336, (12288, 1024, 1024))):
            xc, wc, ic = synth_inputs(1, 64, nn_, kk_)
            xs, ws, ids = xc.to(dev), wc.to(dev), ic.to(dev)
            bs_ = mixedgemm.reorder_quantize_w4(ws, ids, *sp)
            del ws
            for m_ in (16, 32, 64):
                as_ = mixedgemm.reorder_quantize_x(xs[:m_].contiguous(), ids, *sp)
                os_ = torch.empty((m_, nn_), dtype=torch.bfloat16, device=dev)
                pa = [t.data_ptr() if t.numel() else None for t in (as_[0], bs_[0], as_[1], bs_[1], as_[2], bs_[2], as_[3], bs_[3], as_[4], bs_[4], as_[5], bs_[5])]
                wsb = lib.mm_matmul_workspace_bytes(m_, nn_, *sp, 1, 4)
                wst = mixedgemm.split_workspace(dev, wsb) if wsb else None
                fs = lambda: lib.mm_matmul_ws(*pa, m_, nn_, *sp, 1, 4 if wst is not None else 0, None, os_.data_ptr(),
                                              wst.data_ptr() if wst is not None else None, wst.numel() if wst is not None else 0, stream_ptr)
                assert fs() == 0
                settle(fs, 0.15)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    fs()
                torch.cuda.synchronize()
                t_s = (time.perf_counter() - t0) / 200
                wb_ = nn_ * kk_ // 2 + nn_ * kk_ // 32
                sm[f"{name}_M{m_}"] = {"M": m_, "N": nn_, "K": kk_, "split": list(sp), "us_per_launch": round(t_s * 1e6, 2),
                                      "weight_stream_TBps": round(wb_ / t_s / 1e12, 3),
                                      "kernel": lib.mm_matmul_describe(m_, nn_, *sp, 1, 4 if wsb else 0, wsb).decode()[:90]}
            del bs_
        sm["note"] = ("back-to-back direct C-ABI launches (mm_matmul_ws with the stream's split-K workspace where the plan wants one); the SAME "
                      "weights every launch, i.e. served by the 256 MiB Infinity Cache when the allocator's placement lets them stay there (the same "
                      "launch measures ~7 us in one process and ~9 us in another) -- `small_m_hbm` rotates through 12 weight sets")
        result["small_m"] = sm
        # ---- the same launches with the weights coming from HBM: 12 different gate_proj-sized weight sets in rotation (12 x 31 MB > the
        #      256 MiB Infinity Cache), as consecutive layers of a model would present them ----
        hb = {}
        nn_, kk_, sp = 14336, K, fsplit
        sets = []
        for r_ in range(12):
            wr = (torch.randn((nn_, kk_), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
            sets.append(mixedgemm.reorder_quantize_w4(wr, idx, *sp))
            del wr
        xh = torch.randn((16, kk_), device=dev, dtype=torch.float32).to(torch.bfloat16)
        oh = torch.empty((16, nn_), dtype=torch.bfloat16, device=dev)
        wb_ = nn_ * kk_ // 2 + nn_ * kk_ // 32
        for m_ in (1, 16):
            ah = mixedgemm.reorder_quantize_x(xh[:m_].contiguous(), idx, *sp)
            calls = []
            for bs_ in sets:
                pa = [t.data_ptr() if t.numel() else None for t in (ah[0], bs_[0], ah[1], bs_[1], ah[2], bs_[2], ah[3], bs_[3], ah[4], bs_[4], ah[5], bs_[5])]
                calls.append(pa)
            def frot(n_=12):
                for pa in calls[:n_]:
                    lib.mm_matmul(*pa, m_, nn_, *sp, 1, 0, None, oh.data_ptr(), stream_ptr)
            settle(frot, 0.15)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                frot()
            torch.cuda.synchronize()
            t_r = (time.perf_counter() - t0) / (20 * 12)
            hb[f"gate_up_M{m_}"] = {"M": m_, "N": nn_, "K": kk_, "split": list(sp), "us_per_launch": round(t_r * 1e6, 2),
                                    "weight_stream_TBps": round(wb_ / t_r / 1e12, 3), "weight_sets": 12,
                                    "kernel": lib.mm_matmul_describe(m_, nn_, *sp, 1, 0, 0).decode()[:90]}
        hb["note"] = "mm_matmul on 12 weight sets in rotation (374 MB of packed weights: every launch streams its 31 MB from HBM); against 8 TB/s"
        result["small_m_hbm"] = hb
        del sets
        del bf
        result["few_tiles"] = few

        # ---- the reference's one published configuration: gemm_host_tn, M = 32, N = K = 4096, MXFP6 x MXFP4 ----
        pm, pn, pk, psplit = PUBLISHED["M"], PUBLISHED["N"], PUBLISHED["K"], PUBLISHED["split"]
        bp = mixedgemm.reorder_quantize_w4(w, idx, *psplit)
        ap = mixedgemm.reorder_quantize_x(x[:pm].contiguous(), idx, *psplit)
        op = torch.empty((pm, pn), dtype=torch.bfloat16, device=dev)
        ptr = lambda t: t.data_ptr() if t.numel() else None
        sptr = torch.cuda.current_stream().cuda_stream
        pargs = [ptr(t) for t in (ap[0], bp[0], ap[1], bp[1], ap[2], bp[2], ap[3], bp[3], ap[4], bp[4], ap[5], bp[5])]
        fp = lambda: lib.mm_matmul(*pargs, pm, pn, *psplit, 1, 0, None, op.data_ptr(), sptr)
        assert fp() == 0
        settle(fp, 0.2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fp()
        e1.record()
        torch.cuda.synchronize()
        t_pub = e0.elapsed_time(e1) * 1e-3 / 200
        result["published_config"] = {
            "M": pm, "N": pn, "K": pk, "split": list(psplit), "formats": "MXFP6(E3M2) x MXFP4, w4 weights",
            "us_per_launch": round(t_pub * 1e6, 2), "tflops": round(2.0 * pm * pn * pk / t_pub / 1e12, 2),
            "kernel": lib.mm_matmul_describe(pm, pn, *psplit, 1, 0, 0).decode() if pm > 64 else "mm_matmul weight-streaming kernel (M <= 64)",
            "timing": "200 back-to-back launches through the C ABI between two events on the launch stream (launch gaps included)",
            "reference": {k: PUBLISHED[k] for k in ("reference_ms", "reference_tflops", "reference_hardware", "source")},
            "note": "context only: the reference's figure is its hand-written CuTe kernel on other hardware; same shape, formats and TFLOPS formula",
        }
        del bp, ap

        # ---- the other quantizers of the path (section 8f): algorithmic bytes against 8 TB/s, timed BOTH ways -- `kernel_us` = events
        # attached to each dispatch (the kernel's own duration, what rocprofv3's kernel trace reports), `step_us` = back-to-back launches
        # between two events (kernel boundaries included: what a caller that queues them pays per launch) ----
        def timed_direct(fn, reps=100):
            assert fn() == 0
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(reps):
                fn()
            a1.record()
            torch.cuda.synchronize()
            return a0.elapsed_time(a1) * 1e-3 / reps

        def both_ways(fn, byts, reps=100):
            step = timed_direct(fn, reps)
            kern = kernel_us(fn, min(reps, 50)) * 1e-6
            return {"kernel_us": round(kern * 1e6, 2), "step_us": round(step * 1e6, 2), "GBps": round(byts / kern / 1e9, 1),
                    "frac_of_8TBps": round(byts / kern / 8e12, 4), "step_frac_of_8TBps": round(byts / step / 8e12, 4)}
        quant = {}
        qsplit = (2048, 128, 1920)
        qo2 = mixedgemm.reorder_quantize_x(x, idx, *qsplit)
        out_b = M * (qsplit[0] // 2 + qsplit[1] * 3 // 4 + qsplit[2]) + M * K // 32
        normw = torch.ones((K,), dtype=torch.bfloat16, device=dev)
        quant["rmsnorm_quantize_x"] = dict({"rows": M, "K": K, "split": list(qsplit)}, **both_ways(
            lambda: lib.mm_rmsnorm_quantize(x.data_ptr(), normw.data_ptr(), 1e-5, M, K, idx.data_ptr(), *qsplit, 0, *[ptr(t) for t in qo2], sptr),
            2 * M * K + out_b + 4 * K))
        quant["reorder_quantize_x"] = dict({"rows": M, "K": K, "split": list(qsplit)}, **both_ways(
            lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *qsplit, 0, *[ptr(t) for t in qo2], sptr), 2 * M * K + out_b + 2 * K))
        inter, asplit = 14336, (12288, 1024, 1024)
        ga = torch.randn((M, inter), device=dev, dtype=torch.float32).to(torch.bfloat16)
        gb = torch.randn((M, inter), device=dev, dtype=torch.float32).to(torch.bfloat16)
        qa = mixedgemm.activate_quantize_x(ga, gb, *asplit)
        act_b = 2 * 2 * M * inter + M * (asplit[0] // 2 + asplit[1] * 3 // 4 + asplit[2]) + M * inter // 32
        quant["activate_quantize_x"] = dict({"rows": M, "K": inter, "split": list(asplit)}, **both_ways(
            lambda: lib.mm_activate_quantize(ga.data_ptr(), gb.data_ptr(), M, *asplit, *[ptr(t) for t in qa], sptr), act_b, 50))
        quant["timing"] = ("kernel_us: HIP events attached to each dispatch (mm_diag_set_kernel_events), mean; step_us: back-to-back direct "
                           "C-ABI launches between two events on the launch stream; frac_of_8TBps is the kernel's, step_frac_of_8TBps the step's")
        result["quantizers"] = quant
        del ga, gb, qa, qo2

        # ---- the full QLinearLayer.forward hot path, and the "w" weight mode ----
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
            mm(q, b, out)
        settle(fwd, 0.4)
        t_fwd = timed(fwd)
        t_fwd_graph = None
        if graph is not None:       # the same K forwards as one hipGraph, reported BESIDE the stream figure (never instead of it)
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
                t_fwd_graph = (time.perf_counter() - t) / args.steps
                del gq
            except Exception as e:
                print(f"[bench] hipGraph capture of the forward failed ({e})", file=sys.stderr)
        # quantizer kernel alone: direct C-ABI calls on preallocated outputs (the op-level call spends ~20 us of host
        # time on six allocations, which would hide the 11 us kernel)
        qo = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
        pp = lambda t: t.data_ptr() if t.numel() else None
        stream = torch.cuda.current_stream().cuda_stream
        # (events around 100 back-to-back launches, as the `quantizers` section: a wall clock around --steps launches charges the final
        # synchronize to them -- with the driver's 20 steps that is ~1.5 us on a 9 us kernel)
        q_bytes = 2 * M * K + M * K + M * K // 32 + 2 * K
        qx_both = both_ways(lambda: lib.mm_reorder_quantize(x.data_ptr(), M, K, idx.data_ptr(), *SPLIT, 0, pp(qo[0]), pp(qo[1]), pp(qo[2]),
                                                            pp(qo[3]), pp(qo[4]), pp(qo[5]), stream), q_bytes)
        bw = mixedgemm.reorder_quantize_w(w, idx, *SPLIT)
        fw = lambda: mm(a, bw, out)
        us_w = kernel_us(fw, args.steps)
        result["qlinear"] = {
            "tokens_per_s": round(M / t_fwd, 1), "forward_us": round(t_fwd * 1e6, 2), "forward_launch": "stream launches (quantize_x + matmul per forward)",
            "forward_us_graph": round(t_fwd_graph * 1e6, 2) if t_fwd_graph else None,
            "tokens_per_s_graph": round(M / t_fwd_graph, 1) if t_fwd_graph else None,
            "quantize_x_kernel_us": qx_both["kernel_us"], "quantize_x_step_us": qx_both["step_us"],
            "quantize_x_timing": "kernel_us: events attached to each dispatch; step_us: 100 back-to-back direct C-ABI launches between two events",
            "quantize_x_GBps": qx_both["GBps"], "quantize_x_frac_of_8TBps": qx_both["frac_of_8TBps"],
            "quantize_x_step_frac_of_8TBps": qx_both["step_frac_of_8TBps"],
            "gemm_w_mode_kernel_us": round(us_w, 2), "gemm_w_mode_tflops": round(flop / us_w / 1e6, 2),
        }
        del bw
        result["llama_layer"] = llama_layer(dev, lib, mixedgemm, x, args.steps)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(x_cpu, w_cpu, idx_cpu)
    if world > 1:
        # ---- GEMM and all-reduce separately (events on the compute stream; the all-reduce is waited for on that stream) ----
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        for e0, e1, e2 in ev:
            e0.record()
            if not layer.empty:
                layer.ops.matmul(a, layer.packed_w, out=out)
            else:
                out.zero_()
            e1.record()
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
            e2.record()
        barrier()
        gemm_us = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e3
        ar_us = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) * 1e3
        widths = layer.shard_widths
        shard_flop = 2.0 * M * N * sum(widths)
        tstar = roofline_time_s(M, N, widths) * 1e6
        stats = torch.tensor([gemm_us, ar_us], dtype=torch.float64, device=dev)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        extra["tp"] = {
            "layout": "K-shard (row-parallel), one all-reduce per linear", "shard_columns_rank0": widths,
            "gemm_us_max_over_ranks": round(float(stats[0]), 2), "allreduce_us_max_over_ranks": round(float(stats[1]), 2),
            "allreduce_payload_bytes": M * N * 2,
            "rank0_roofline": {"bound": "mfma", "kernel_us": round(gemm_us, 2), "roofline_us": round(tstar, 2),
                               "frac": round(tstar / gemm_us, 4) if gemm_us > 0 else None,
                               "achieved": round(shard_flop / gemm_us / 1e6, 1) if gemm_us > 0 else None, "unit": "TFLOP/s"},
            "note": "events on the compute stream: gemm = the shard's fused GEMM, allreduce = RCCL sum of the bf16 [M,N] partials",
        }
        if not args.no_extras:
            # the same GEMM with rows instead of K split across the GPUs (every rank multiplies its own 4096 token rows by the
            # replicated weights, no exchange): what the node delivers when the linear layer is used data-parallel
            b = mixedgemm.reorder_quantize_w4(w, idx, *SPLIT)
            a2 = mixedgemm.reorder_quantize_x(x, idx, *SPLIT)
            for _ in range(args.warmup):
                mm(a2, b, out)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                mm(a2, b, out)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            extra["row_parallel_no_exchange"] = {"value": round(world * flop / (float(t.item()) / args.steps) / 1e12, 2), "unit": "TFLOP/s",
                                                 "scaling": "weak", "global_rows": world * M}
            del b, a2
            # ---- Megatron pairing: Llama-3-8B MLP (hidden 4096, intermediate 14336), ONE all-reduce per MLP ----
            hid, inter = 4096, 14336
            g = torch.Generator(device=dev).manual_seed(7)
            rnd = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.02).to(torch.bfloat16)
            mlp = tpmod.TPMLP(rnd(inter, hid), rnd(inter, hid), rnd(hid, inter), idx, (2048, 128, 1920), (12288, 1024, 1024),
                              rank=rank, world=world, group=dist.group.WORLD)
            for _ in range(max(3, args.warmup // 10)):
                mlp(x)
            barrier()
            n_mlp = max(5, args.steps // 10)
            t0 = time.perf_counter()
            for _ in range(n_mlp):
                mlp(x)
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mlp_s = float(t.item()) / n_mlp
            extra["tp_mlp"] = {
                "layout": "gate/up column-parallel -> activate_quantize_x on the local slice -> down row-parallel -> one all-reduce",
                "tokens": M, "hidden": hid, "intermediate": inter, "mlp_us": round(mlp_s * 1e6, 1),
                "tflops": round(2.0 * M * hid * inter * 3 / mlp_s / 1e12, 1), "allreduce_payload_bytes": M * hid * 2,
                "allreduce_payload_bytes_if_every_linear_were_k_sharded": 2 * M * inter * 2 + M * hid * 2}
    if rank == 0:
        result.update(extra)
        details = None
        try:
            dpath = args.details if os.path.isabs(args.details) else os.path.join(ROOT, args.details)
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            with open(dpath, "w") as f:
                json.dump(result, f, indent=1)
            details = args.details
        except OSError as e:
            print(f"[bench] could not write the details file ({e})", file=sys.stderr)
        print(compact_line(result, details), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
