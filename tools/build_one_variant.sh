#!/bin/bash
# A variant of ONE translation unit linked against the default build's other objects (seconds instead of minutes):
#   tools/build_one_variant.sh NAME mx_gemm_stream -DFLAG=1 ...   -> micromix_amd/lib/dbg/lib_NAME.so   (run python -m micromix_amd.build first)
cd "$(dirname "$0")/.."
mkdir -p micromix_amd/lib/dbg
name=$1; unit=$2; shift 2
O=micromix_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc "$@" -c micromix_amd/csrc/$unit.hip -o micromix_amd/lib/dbg/${unit}_$name.o 2>&1 | grep -E "error|spill"
others=$(ls $O/*.o | grep -v "/$unit.o" | grep -v "/diag.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $others micromix_amd/lib/dbg/${unit}_$name.o -o micromix_amd/lib/dbg/lib_$name.so && echo "built $name"
