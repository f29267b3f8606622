#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats of bench.py, then PMC passes.
# Run on the GPU box:  bash tools/profile.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_target.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_target.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 tools/pmc_target.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_lds -- python3 tools/pmc_target.py > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -50
