"""Turns a tools/profile.sh output directory into the files committed under profiles/ (written to gpurun_out/profiles_<tag>/)."""
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", f"profiles_{tag}")   # merged back by gpurun; copy into profiles/ afterwards
os.makedirs(prof, exist_ok=True)
lines = []
CMD = "python3 bench.py --steps 20 --warmup 5"
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    lines.append(f"== rocprofv3 --kernel-trace --stats -- {CMD}  (every kernel of the run; the GEMM kernel symbol also serves the mixed splits) ==")
    for r in rows:
        lines.append(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}%")
# The headline GEMM (4096^3, split (0,0,4096), w4) = the first uninterrupted run of launches of the 256x256-tile kernel: warm-up,
# graph capture warm-up, first graph replay, the settle phase, the timed stream-launch pass (`value`), the graph replay and, last,
# the pass with HIP events attached (`roofline.kernel_us`, `steps` launches).  Anything after the next quantizer launch is another split.
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    run, started = [], False
    for r in rows:
        is_gemm = "g256::mx_gemm256_kernel<true, false>" in r["Kernel_Name"]
        if is_gemm:
            started = True
            run.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif started and "reorder_quantize" in r["Kernel_Name"]:
            break
    if len(run) >= 60:
        steps = 20
        lines.append(f"headline GEMM: {len(run)} consecutive launches in the trace; ALL: avg {sum(run)/len(run):.2f} us min {min(run):.2f} max {max(run):.2f}")
        for name, t in (("event pass (last 20 launches = roofline.kernel_us)", run[-steps:]), ("hipGraph replay (auxiliary figure, 20 launches before those)", run[-2 * steps:-steps]),
                        ("timed region (`value`: K stream launches, 20 launches before those)", run[-3 * steps:-2 * steps]), ("first 26 launches (warm-up, on the clock ramp)", run[:26])):
            lines.append(f"  {name}: kernel-trace avg {sum(t)/len(t):.2f} us min {min(t):.2f} max {max(t):.2f}")
bench_line = [l for l in open(out + "/bench_under_rocprof.log") if l.startswith("{")]
if bench_line:
    lines.append("== bench.py JSON line of the same (profiled) run ==")
    lines.append(bench_line[-1].strip())
plain = [l for l in open(out + "/bench_plain.log") if l.startswith("{")] if os.path.exists(out + "/bench_plain.log") else []
if plain:
    lines.append(f"== bench.py JSON line of an unprofiled run of the same command on the same box ({CMD}) ==")
    lines.append(plain[-1].strip())
traffic = None
SMALL = ("few", "kv", "decode", "stream", "gateup")     # launches that are not ONE round of 256 eight-wave workgroups: ratios only
for name in ("fp8", "fp8w", "fp4", "mixed", "mixed3072", "down") + SMALL:   # (gateup: 1792 workgroups of the fused gate / up kernel)
    d = os.path.join(root, "gpurun_out", f"pmc_{tag}_{name}")
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    if not agg:
        continue
    head = open(os.path.join(d, "summary.txt")).readline().strip() if os.path.exists(os.path.join(d, "summary.txt")) else name
    lines.append(f"== rocprofv3 --pmc passes -- python3 tools/pmc_target.py ({head}; 10 x quantize_x + matmul" +
                 (", 4096^3) ==" if name not in SMALL else "; 10 x quantize_x + mm_gate_up_activate) ==" if name == "gateup" else "; M <= 8: + 10 x the fused decode kernel) =="))
    stat = {}
    for (k, c), v in sorted(agg.items()):
        if "gemm" in k or "reorder" in k or "decode" in k:
            lines.append(f"{k:50s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
            stat[("gemm" if "gemm" in k else "decode" if "decode" in k else "quant", c)] = sum(v) / len(v)
    g = lambda c: stat.get(("gemm", c))
    if name in SMALL:
        for kind in ("gemm", "decode"):
            h = lambda c: stat.get((kind, c))
            if h("SQ_WAVE_CYCLES") and h("SQ_WAIT_ANY") is not None:
                lines.append(f"derived ({kind} kernel): waves waiting (s_waitcnt/barrier) {h('SQ_WAIT_ANY')/h('SQ_WAVE_CYCLES'):.3f} of wave time, issue stalls "
                             f"{h('SQ_WAIT_INST_ANY')/h('SQ_WAVE_CYCLES'):.3f}; matrix pipe busy {h('SQ_VALU_MFMA_BUSY_CYCLES'):.0f} SIMD-cycles of "
                             f"{h('SQ_BUSY_CYCLES'):.0f} SQ-busy cycles")
            if h("SQ_LDS_BANK_CONFLICT") is not None and h("SQ_LDS_IDX_ACTIVE"):
                lines.append(f"derived ({kind} kernel): LDS bank-conflict cycles {h('SQ_LDS_BANK_CONFLICT'):.0f} = {h('SQ_LDS_BANK_CONFLICT')/h('SQ_LDS_IDX_ACTIVE'):.4f} of the LDS-active cycles")
            if h("FETCH_SIZE") is not None and h("WRITE_SIZE") is not None:
                lines.append(f"derived ({kind} kernel): HBM bytes per launch {int(2*h('FETCH_SIZE')*1024 + h('WRITE_SIZE')*1024)} (2 x FETCH_SIZE + WRITE_SIZE)"
                             + (f", L2 hit rate {h('TCC_HIT_sum')/(h('TCC_HIT_sum')+h('TCC_MISS_sum')):.3f}" if h("TCC_HIT_sum") else ""))
        continue
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
        # SQ_WAVE_CYCLES counts quad-cycles summed over the 2048 waves (two per SIMD); MFMA busy counts cycles summed over the 1024 SIMDs
        wave_cycles = g("SQ_WAVE_CYCLES") * 4 / 2048
        lines.append(f"derived: kernel ~{wave_cycles:.0f} shader cycles per wave; matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/1024:.0f} cycles per SIMD "
                     f"= {g('SQ_VALU_MFMA_BUSY_CYCLES')/1024/wave_cycles:.3f} of the kernel; waves waiting (s_waitcnt/barrier) {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.3f}, "
                     f"issue stalls {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f} of wave time")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"derived: LDS bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT'):.0f} = {g('SQ_LDS_BANK_CONFLICT')/g('SQ_LDS_IDX_ACTIVE'):.4f} of the LDS-active cycles")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        fetch_b = 2.0 * g("FETCH_SIZE") * 1024.0     # FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950
        write_b = g("WRITE_SIZE") * 1024.0
        t = {"hbm_bytes_per_launch": int(fetch_b + write_b), "fetch_bytes_corrected": int(fetch_b), "write_bytes": int(write_b),
             "collected": f"profiles/{tag}_summary.txt, split {head}",
             "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md",
             "l2_hit_rate": (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))) if g("TCC_HIT_sum") else None}
        lines.append("derived: " + json.dumps(t))
        if name == "fp8":
            traffic = t
    q = lambda c: stat.get(("quant", c))
    if name == "fp8" and q("FETCH_SIZE") is not None and q("WRITE_SIZE") is not None:
        lines.append(f"derived: quantize_x HBM bytes per launch: {int(2*q('FETCH_SIZE')*1024 + q('WRITE_SIZE')*1024)} (algorithmic 50864128)")
if traffic:
    json.dump(traffic, open(os.path.join(prof, "gemm_traffic.json"), "w"), indent=1)
open(os.path.join(prof, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
