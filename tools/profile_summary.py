"""Turns a tools/profile.sh output directory into the committed files under profiles/."""
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", f"profiles_{tag}")   # merged back by gpurun; copy into profiles/ afterwards
os.makedirs(prof, exist_ok=True)
lines = []
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    lines.append("== rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 50 ==")
    for r in rows:
        lines.append(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}%")
# bench.py's launches of the w4 GEMM kernel inside the same trace (W = 50, K = 200): 50 warm-up launches, 1 launch that warms the
# capture stream, 200 of the untimed first graph replay, then launches 252..451 = the timed region (`value`, one hipGraph) and
# launches 452..651 = the pass with events attached to each dispatch (`roofline.kernel_us`)
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
         if "mx_gemm256_kernel<true, false>" in r["Kernel_Name"]]
    if len(d) >= 651:
        for name, t in (("timed region (GEMM launches 252..451, hipGraph)", d[251:451]), ("event pass (GEMM launches 452..651)", d[451:651])):
            lines.append(f"{name}: kernel-trace avg {sum(t)/len(t):.2f} us min {min(t):.2f} max {max(t):.2f}")
bench_line = [l for l in open(out + "/bench_under_rocprof.log") if l.startswith("{")]
if bench_line:
    lines.append("== bench.py JSON line of the same (profiled) run; its event-based kernel_us reads ~4 us high under the profiler ==")
    lines.append(bench_line[-1].strip())
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
lines.append("== rocprofv3 --pmc passes -- python3 tools/pmc_target.py (10 x quantize_x + matmul, bench shape) ==")
stat = {}
for (k, c), v in sorted(agg.items()):
    if "gemm" in k or "reorder" in k:
        lines.append(f"{k:50s} {c:26s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
        kind = "gemm" if "gemm" in k else ("quant" if "<false" in k else "quant_w4")
        stat[(kind, c)] = sum(v) / len(v)
g = lambda c: stat.get(("gemm", c))
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    fetch_b = 2.0 * g("FETCH_SIZE") * 1024.0     # FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950
    write_b = g("WRITE_SIZE") * 1024.0
    traffic = {"hbm_bytes_per_launch": int(fetch_b + write_b), "fetch_bytes_corrected": int(fetch_b), "write_bytes": int(write_b),
               "source": f"profiles/{tag}_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md",
               "l2_hit_rate": (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))) if g("TCC_HIT_sum") else None}
    json.dump(traffic, open(os.path.join(prof, "gemm_traffic.json"), "w"), indent=1)
    lines.append("== derived ==")
    lines.append(json.dumps(traffic))
    q = lambda c: stat.get(("quant", c))
    if q("FETCH_SIZE") is not None and q("WRITE_SIZE") is not None:
        lines.append(f"quantize_x HBM bytes per launch: {int(2*q('FETCH_SIZE')*1024 + q('WRITE_SIZE')*1024)} (algorithmic 50864128)")
plain = os.path.join(root, "gpurun_out", "bench_plain.log")
if os.path.exists(plain):
    bl = [l for l in open(plain) if l.startswith("{")]
    if bl:
        lines.append("== bench.py JSON line of an unprofiled run on the same box ==")
        lines.append(bl[-1].strip())
open(os.path.join(prof, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
