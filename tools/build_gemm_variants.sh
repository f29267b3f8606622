#!/bin/bash
# Instrumented variants of the tiled GEMM (mx_gemm256.hip) for the delivery probes (tools/delivery_probes.py), built in parallel:
#   tools/build_gemm_variants.sh NAME1 "FLAGS1" NAME2 "FLAGS2" ...     -> micromix_amd/lib/dbg/lib_NAME.so
# Every variant gets -DMM_INSTRUMENT (in-kernel clock stamps; mm_diag_set_clock_buffer).  Only mx_gemm256.hip is recompiled per variant;
# capi.hip and reorder_quantize.hip (the instrumented exports) once; the other objects are the default build's (python -m micromix_amd.build).
cd "$(dirname "$0")/.."
D=micromix_amd/lib/dbg; O=micromix_amd/lib/obj; S=micromix_amd/csrc
mkdir -p $D
CC="hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc"
$CC -DMM_INSTRUMENT -c $S/capi.hip -o $D/capi_instr.o &
$CC -DMM_INSTRUMENT -c $S/reorder_quantize.hip -o $D/reorder_quantize_instr.o &
names=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2; names+=($name)
  ( $CC -DMM_INSTRUMENT $flags -c $S/mx_gemm256.hip -o $D/mx_gemm256_$name.o 2>&1 | grep -E "error|spill" ) &
  # at most 7 compilers at once (8 cores, 64 GiB)
  while [ $(jobs -r | wc -l) -ge 7 ]; do sleep 1; done
done
wait
others=$(ls $O/*.o | grep -v "/mx_gemm256.o" | grep -v "/capi.o" | grep -v "/reorder_quantize.o" | grep -v "/diag.o")
for name in "${names[@]}"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $others $D/capi_instr.o $D/reorder_quantize_instr.o $D/mx_gemm256_$name.o -o $D/lib_$name.so && echo "built $name"
done
