#!/bin/bash
# Round 6, VERDICT r5 item 2: what bounds the L2 -> LDS operand delivery of the 256 x 256 tile's K loop, and do the cheap levers move it?
# Needs the variants of tools/build_gemm_variants.sh (see the call in profiles/r06_delivery.txt).  Run on the GPU box:
#   bash tools/delivery_probes.sh > gpurun_out/r06_delivery_raw.txt 2>&1
cd "$(dirname "$0")/.."
D=$PWD/micromix_amd/lib/dbg
SPLITS="4096,0,0 0,0,4096 3072,896,128"
clock() { MICROMIX_HIP_LIB=$D/lib_$1.so python3 tools/gemm_clock.py $SPLITS "${@:2}" 2>&1 | grep -v "^$"; }
echo "== A. in-kernel loop cycles / clock / phase stamps (tools/gemm_clock.py), M = 4096: 256 workgroups, one per CU"
for v in instr tile00 xcd0 nomfma nomfma_noread nodma ns2 kd128 gh2 gh8 gh16; do echo "-- $v"; clock $v; done
echo "== B. half of the CUs: M = 2048 pinned to 256 x 256 tiles (128 workgroups, 16 per XCD)"
for v in instr nomfma nomfma_noread; do echo "-- $v"; MICROMIX_GEMM_TILE=256 clock $v M=2048; done
echo "== C. a quarter: M = 1024 pinned to 256 x 256 tiles (64 workgroups, 8 per XCD)"
for v in instr nomfma_noread; do echo "-- $v"; MICROMIX_GEMM_TILE=256 clock $v M=1024; done
echo "== D. wall time of the levers (tools/time_cases.py: back-to-back launches, median of 7 x 20), two passes, alternating"
CASES="4096,4096,4096:4096,0,0 4096,4096,4096:0,0,4096 4096,4096,4096:3072,896,128 4096,4096,4096:2048,128,1920 4096,4096,14336:12288,1024,1024 4096,14336,4096:3072,896,128"
for pass in 1 2; do
  for v in instr gh2 gh8 gh16 ns2 kd128; do MICROMIX_HIP_LIB=$D/lib_$v.so python3 tools/time_cases.py $CASES 2>&1 | grep "^{"; done
done
