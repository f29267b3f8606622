#!/bin/bash
# round-3 baseline of the round-2 kernels: boundary cost (reference vs fused rounding) and PMC for the splits furthest from roofline
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
python tools/gemm_clock.py 0,0,4096 4096,0,0 2048,128,1920 3072,896,128
python tools/gemm_clock.py fused 2048,128,1920 3072,896,128
python tools/gemm_clock.py K=14336 12288,1024,1024
python tools/gemm_clock.py K=14336 fused 12288,1024,1024
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_base_clock.txt
bash tools/pmc_gemm.sh r03_3072 3072,896,128 > /dev/null 2>&1
bash tools/pmc_gemm.sh r03_down 12288,1024,1024@4096x4096x14336 > /dev/null 2>&1
cat gpurun_out/r03_base_clock.txt gpurun_out/pmc_r03_3072/summary.txt gpurun_out/pmc_r03_down/summary.txt
