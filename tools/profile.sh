#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box:  bash tools/profile.sh r01
#   1. kernel-trace stats of the SAME command the driver runs (bench.py), summary copied to profiles/
#   2. PMC passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) on the bench shape
#   3. profiles/gemm_traffic.json = HBM bytes per GEMM launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE
#      reports half of the bytes of a wide streaming read), read back by bench.py for roofline.traffic
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 bench.py --steps 200 --warmup 50 > $REPO/gpurun_out/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 50 > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_target.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_target.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 tools/pmc_target.py > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 tools/pmc_target.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/pmc_target.py > $OUT/pmc_sq.log 2>&1
python3 tools/profile_summary.py $OUT $TAG
