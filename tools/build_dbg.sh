#!/bin/bash
# builds ablation variants of the library into micromix_amd/lib/dbg/lib_dbg<N>.so (see MM_DBG in mx_gemm256.hip)
cd "$(dirname "$0")/.."
mkdir -p micromix_amd/lib/dbg
for d in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -fno-gpu-rdc -DMM_DBG=$d micromix_amd/csrc/capi.hip micromix_amd/csrc/reorder_quantize.hip micromix_amd/csrc/direct_quantize.hip micromix_amd/csrc/rmsnorm_quantize.hip micromix_amd/csrc/mx_gemm.hip micromix_amd/csrc/mx_gemm256.hip micromix_amd/csrc/mx_gemm_skinny.hip micromix_amd/csrc/qlinear_decode.hip micromix_amd/csrc/diag.hip -o micromix_amd/lib/dbg/lib_dbg$d.so 2>&1 | grep -E " error"
done
