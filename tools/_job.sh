cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_try.json 2> gpurun_out/r03_bench_try.err; tail -3 gpurun_out/r03_bench_try.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_try.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","graph_launch_ms_per_step","stream_launch_ms_per_step","power"):
    print(k, d.get(k))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","kernel_us")})
for k,v in d["mixed"].items(): print(k, v["kernel_us"], v["frac"])
print("few", {k:v["kernel_us"] for k,v in d["few_tiles"].items()})
print("decode", d["decode"])
print("published", {k:d["published_config"][k] for k in ("us_per_launch","tflops")})
print("quantizers", {k:(v["kernel_us"], v["frac_of_8TBps"]) for k,v in d["quantizers"].items() if isinstance(v, dict)})
print("qlinear", d["qlinear"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
