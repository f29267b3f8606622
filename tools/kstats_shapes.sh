#!/bin/bash
# rocprofv3 kernel-trace stats of tools/bench_shapes.py for the given M values: per-kernel average durations
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats_shapes
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_shapes.py "$@" > $OUT/run.log 2>&1
cat $OUT/*/*_kernel_stats.csv | cut -c1-160
