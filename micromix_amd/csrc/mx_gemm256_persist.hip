// mm::g256p: the PERSISTENT kernels of the 256 x 256 tile (round 5; mx_gemm_tile.inc "Persistent launches", mm_tid).
// A namespace -- and a translation unit -- of its own, so that the one-tile kernels of mx_gemm256.hip compile exactly as before.
// Default off (MICROMIX_GEMM_PERSIST=1; profiles/r05_persist_ab.txt).
#include "mx_gemm_prelude.h"

namespace mm {

#define MM_NS g256p
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 4
#define MM_TM 2
#define MM_TN 4
#define MM_PERSIST_NS 1
#define MM_ACC MM_ACC_CLOBBER
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET

hipError_t launch_g256p(bool act, const GemmArgs &a, int grid, hipStream_t stream) {
    static DynamicLdsOnce done[2];
    if (act) return launch_tile(g256p::mx_gemm256_persist_kernel<true>, done[1], g256p::LDS_BUDGET, grid, g256p::NT, a, stream);
    return launch_tile(g256p::mx_gemm256_persist_kernel<false>, done[0], g256p::LDS_BUDGET, grid, g256p::NT, a, stream);
}

}  // namespace mm
