#!/bin/bash
# PMC passes for the GEMM kernel only, one split per run: bash tools/pmc_gemm.sh TAG KN,KS,KO
# (separate --pmc passes per counter group, no trace domains: MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -u
TAG=${1:-x}; SPLIT=${2:-0,0,4096}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/pmc_target.py $SPLIT > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/pmc_target.py $SPLIT > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- python3 tools/pmc_target.py $SPLIT > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F8 --output-format csv -d $OUT/sq -- python3 tools/pmc_target.py $SPLIT > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq2 -- python3 tools/pmc_target.py $SPLIT > $OUT/sq2.log 2>&1
echo "== split $SPLIT" > $OUT/summary.txt
python3 tools/pmc_summary.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
