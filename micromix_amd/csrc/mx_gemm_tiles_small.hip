// The tiles for launches that cannot fill the chip with 128-row tiles -- 128 x 128 (8 waves), 64 x 128 / 64 x 64 (4 compute + 4 loader
// waves), 32 x 64 (2 + 2) -- in a translation unit of their own, so that hipcc compiles them beside mx_gemm256.hip (round 6: one file
// took 100 s of a 105 s build).  plan_tiles / plan_small_split in mx_gemm256.hip choose; the launchers below are all that file sees.
#include "mx_gemm_prelude.h"

namespace mm {

#define MM_NS g64
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 4
#define MM_TM 1
#define MM_TN 2
#define MM_ACC MM_ACC_CLOBBER
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET
#define MM_NS g32
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 2
#define MM_TM 1
#define MM_TN 2
#define MM_ACC MM_ACC_CLOBBER32
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET
#define MM_NS g32n
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 2
#define MM_TM 1
#define MM_TN 1
#define MM_ACC MM_ACC_CLOBBER32
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET
#define MM_NS g16
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 1
#define MM_TM 1
#define MM_TN 1
#define MM_ACC MM_ACC_CLOBBER32
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET


static_assert((size_t)g32n::NACC * g32n::NT * 4 == SMALL_PART_BYTES_64x64 && (size_t)g32::NACC * g32::NT * 4 == SMALL_PART_BYTES_64x128,
              "partial-sum slot sizes of the in-kernel split-K (mx_gemm_prelude.h)");

// kind: 64 = 128 x 128 tiles (g64), 32 = 64 x 128 (g32), 33 = 64 x 64 (g32n), 16 = 32 x 64 (g16); splitk: the in-kernel split-K
// variants (split_tile_reduce; 64 x 128 and 64 x 64 tiles only)
hipError_t launch_small_tile(int kind, bool w4, bool splitk, int wgs, const GemmArgs &a, hipStream_t stream) {
    static DynamicLdsOnce done[12];
    if (splitk) {
        if (kind == 33)
            return w4 ? launch_tile(g32n::mx_gemm256_kernel<true, true>, done[8], g32n::Lds<true>::TOTAL, wgs, g32n::NTHREADS, a, stream)
                      : launch_tile(g32n::mx_gemm256_kernel<false, true>, done[9], g32n::Lds<false>::TOTAL, wgs, g32n::NTHREADS, a, stream);
        if (kind == 32)
            return w4 ? launch_tile(g32::mx_gemm256_kernel<true, true>, done[10], g32::Lds<true>::TOTAL, wgs, g32::NTHREADS, a, stream)
                      : launch_tile(g32::mx_gemm256_kernel<false, true>, done[11], g32::Lds<false>::TOTAL, wgs, g32::NTHREADS, a, stream);
        return hipErrorInvalidValue;
    }
    switch (kind) {
        case 64:
            return w4 ? launch_tile(g64::mx_gemm256_kernel<true, false>, done[0], g64::Lds<true>::TOTAL, wgs, g64::NT, a, stream)
                      : launch_tile(g64::mx_gemm256_kernel<false, false>, done[1], g64::Lds<false>::TOTAL, wgs, g64::NT, a, stream);
        case 32:
            return w4 ? launch_tile(g32::mx_gemm256_kernel<true, false>, done[2], g32::Lds<true>::TOTAL, wgs, g32::NTHREADS, a, stream)
                      : launch_tile(g32::mx_gemm256_kernel<false, false>, done[3], g32::Lds<false>::TOTAL, wgs, g32::NTHREADS, a, stream);
        case 33:
            return w4 ? launch_tile(g32n::mx_gemm256_kernel<true, false>, done[4], g32n::Lds<true>::TOTAL, wgs, g32n::NTHREADS, a, stream)
                      : launch_tile(g32n::mx_gemm256_kernel<false, false>, done[5], g32n::Lds<false>::TOTAL, wgs, g32n::NTHREADS, a, stream);
        case 16:
            return w4 ? launch_tile(g16::mx_gemm256_kernel<true, false>, done[6], g16::Lds<true>::TOTAL, wgs, g16::NTHREADS, a, stream)
                      : launch_tile(g16::mx_gemm256_kernel<false, false>, done[7], g16::Lds<false>::TOTAL, wgs, g16::NTHREADS, a, stream);
        default: return hipErrorInvalidValue;
    }
}

// grouped launch (launch_mx_gemm256_grouped has filled ga.first_block[]; total = all groups' tiles)
hipError_t launch_small_tile_grouped(int kind, bool w4, int total, const GroupedTileArgs &ga, hipStream_t stream) {
    static DynamicLdsOnce done[6];
    auto go = [&](auto kern, DynamicLdsOnce &d, int lds, int threads) -> hipError_t {
        if (hipError_t e = d.ensure(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(total), dim3(threads), lds, stream, ga);
        return hipGetLastError();
    };
    switch (kind) {
        case 33: return w4 ? go(g32n::mx_gemm256_grouped_kernel<true>, done[0], g32n::Lds<true>::TOTAL, g32n::NTHREADS)
                           : go(g32n::mx_gemm256_grouped_kernel<false>, done[1], g32n::Lds<false>::TOTAL, g32n::NTHREADS);
        case 32: return w4 ? go(g32::mx_gemm256_grouped_kernel<true>, done[2], g32::Lds<true>::TOTAL, g32::NTHREADS)
                           : go(g32::mx_gemm256_grouped_kernel<false>, done[3], g32::Lds<false>::TOTAL, g32::NTHREADS);
        case 64: return w4 ? go(g64::mx_gemm256_grouped_kernel<true>, done[4], g64::Lds<true>::TOTAL, g64::NT)
                           : go(g64::mx_gemm256_grouped_kernel<false>, done[5], g64::Lds<false>::TOTAL, g64::NT);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mm
