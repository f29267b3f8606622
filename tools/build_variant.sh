#!/bin/bash
# builds a kernel-developer variant of the library: tools/build_variant.sh NAME -DFLAG=1 ... -> micromix_amd/lib/dbg/lib_NAME.so
# (selected at run time with MICROMIX_HIP_LIB=<path>); several can be built in parallel from a shell loop
cd "$(dirname "$0")/.."
mkdir -p micromix_amd/lib/dbg
name=$1; shift
S=micromix_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -fno-gpu-rdc "$@" $S/capi.hip $S/reorder_quantize.hip $S/direct_quantize.hip \
  $S/rmsnorm_quantize.hip $S/mx_gemm.hip $S/mx_gemm256.hip $S/mx_gemm_tiles_small.hip $S/mx_gemm_skinny.hip $S/mx_gemm_stream.hip $S/qlinear_decode.hip \
  -o micromix_amd/lib/dbg/lib_$name.so 2>&1 | grep -E "error|warning: v|spill"
# the accumulator-register guard on THIS variant's flags (an extra -S pass of mx_gemm256.hip; MM_SKIP_ACC_CHECK=1 skips it)
if [ -z "${MM_SKIP_ACC_CHECK:-}" ]; then python3 tools/check_acc_regs.py "$@" | tail -1; fi
echo "built $name"
