#!/bin/bash
# runs a tool under every variant library given: tools/run_variants.sh "tools/gemm_clock.py 4096,0,0" default dbg1 dbg2 ...
# ("default" = no MICROMIX_HIP_LIB: the product library; tools/gemm_clock.py then takes lib_instr.so by itself)
cmd=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset MICROMIX_HIP_LIB; else export MICROMIX_HIP_LIB=$PWD/micromix_amd/lib/dbg/lib_$v.so; fi
  echo "=== $v"; timeout 300 python $cmd 2>&1 | grep -v amdgpu.ids
done
