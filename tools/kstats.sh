#!/bin/bash
# rocprofv3 kernel-trace stats of tools/pmc_target.py (bench shapes): prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats_${1:-x}
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/pmc_target.py > $OUT/run.log 2>&1
cat $OUT/*/*_kernel_stats.csv | cut -c1-220
