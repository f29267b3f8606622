// mm::g256w: the 256 x 256 tile with ONE wave per SIMD -- 4 waves x 128 x 128, 256 accumulators per wave (round 5).
// Default off (MICROMIX_GEMM_W1=1; profiles/r05_w1_ab.txt).  A translation unit of its own: compiled in parallel with mx_gemm256.hip.
#include "mx_gemm_prelude.h"

namespace mm {

// The same 256 x 256 tile with ONE wave per SIMD (round 5): 4 waves as 2 x 2, 128 x 128 outputs = 4 x 4 MFMA tiles
// per wave, all 256 AGPRs are accumulators, 4 + 4 fragment reads per 16 MFMAs instead of 2 + 4 per 8.
#define MM_NS g256w
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 2
#define MM_TM 4
#define MM_TN 4
#define MM_W1 1
#define MM_ACC MM_ACC_CLOBBER256
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET

hipError_t launch_g256w(const GemmArgs &a, int tiles, hipStream_t stream) {
    static DynamicLdsOnce done;
    return launch_tile(g256w::mx_gemm256_kernel<true, false>, done, g256w::Lds<true>::TOTAL, tiles, g256w::NT, a, stream);
}

}  // namespace mm
