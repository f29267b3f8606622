"""Summarises rocprofv3 --pmc csv output directories (one per pass) per kernel."""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "mx_gemm" in k or "reorder" in k:
        print(f"{k:62s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g} min={min(v):.5g} max={max(v):.5g}")
