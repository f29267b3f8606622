#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box:  bash tools/profile.sh r02
#   1. the driver's own command (`python3 bench.py --steps 20 --warmup 5`) unprofiled, then under
#      rocprofv3 --kernel-trace --stats: per-kernel averages + the headline GEMM's launches picked out of the trace
#   2. PMC passes (one counter group per pass, no trace domains: MI355X_MICROARCH.md) on 10 x (quantize_x + matmul) for the
#      bench split, the all-fp4 split, the mixed splits (2048,128,1920) and (3072,896,128) and down_proj (12288,1024,1024), K = 14336,
#      and on the launches with few tiles: q/o at M = 128 (64x64 loader/compute tiles), k/v at M = 128 (in-kernel split-K) and
#      M = 1 (first weight-streaming kernel + the fused decode kernel), gate_proj at M = 16 (mx_gemm_stream_kernel); and on the fused gate / up launch (mm_gate_up_activate) at M = 4096
#   3. gemm_traffic.json = HBM bytes per GEMM launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half of the bytes
#      of a wide streaming read), read back by bench.py for roofline.traffic
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
for cfg in "fp8 0,0,4096" "fp8w w:0,0,4096" "fp4 4096,0,0" "mixed 2048,128,1920" "mixed3072 3072,896,128" "down 12288,1024,1024@4096x4096x14336" \
           "few 2048,128,1920@128x4096x4096" "kv 2048,128,1920@128x1024x4096" "decode 2048,128,1920@1x4096x4096" \
           "stream 2048,128,1920@16x14336x4096" "gateup act:2048,128,1920"; do
  set -- $cfg
  bash tools/pmc_gemm.sh ${TAG}_$1 $2 > /dev/null 2>&1
done
python3 tools/profile_summary.py $OUT $TAG
# the summaries are now under gpurun_out/profiles_$TAG/ (merged back by gpurun): copy them into profiles/; delete $OUT and
# gpurun_out/pmc_${TAG}_* on the box first if the raw traces would push gpurun_out/ past its 64 MiB merge limit
