#!/bin/bash
# builds a kernel-developer variant of the library: tools/build_variant.sh NAME -DFLAG=1 ... -> micromix_amd/lib/dbg/lib_NAME.so
# (selected at run time with MICROMIX_HIP_LIB=<path>); several can be built in parallel from a shell loop
cd "$(dirname "$0")/.."
mkdir -p micromix_amd/lib/dbg
name=$1; shift
S=micromix_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -fno-gpu-rdc "$@" $S/capi.hip $S/reorder_quantize.hip $S/direct_quantize.hip \
  $S/rmsnorm_quantize.hip $S/mx_gemm.hip $S/mx_gemm256.hip $S/mx_gemm_skinny.hip $S/qlinear_decode.hip \
  -o micromix_amd/lib/dbg/lib_$name.so 2>&1 | grep -E "error|warning: v|spill" ; echo "built $name"
