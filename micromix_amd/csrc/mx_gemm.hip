// Fused three-segment mixed-precision block-scaled GEMM for gfx950 (CDNA4): shape dispatch.
//
// Computes what the reference runs as up to three chained CUTLASS SM120 GEMMs
// (mgemm/src/gemm.cu:26-78; per-segment kernels mgemm/src/w4a4.cu, w4a6.cu, w4a8.cu,
// w6a6.cu, w8a8.cu):
//     D = bf16(A_N*B_N^T);  D = bf16(A_S*B_S^T + D);  D = bf16(A_O*B_O^T + D)
// with per-32-element UE8M0 scales on both operands, fp32 accumulation.  Here it is ONE
// launch (two with split-K): the K loop walks the N|S|O segments with the fp32 accumulator
// in registers, switching the MFMA operand formats per segment; in the default
// MM_ROUND_PER_SEGMENT mode the accumulator is rounded through bf16 at each segment
// boundary so that the result follows the reference's rounding order without D ever
// leaving the register file.
//
// Common to every kernel:
//  * v_mfma_scale_f32_32x32x64_f8f6f4: lane l supplies row/col (l & 31); its scale VGPR
//    (byte picked by op_sel) carries the UE8M0 scale of K block (l >> 5).  For fp4/fp6 the
//    lane's registers hold exactly that block; for fp8 they hold two 16-element halves
//    (see load_frag) -- layouts verified on hardware by tests/test_hw_gpu.py.
//  * the ACTIVATION tile is the MFMA row operand (srcA), the WEIGHT tile the column operand.
//  * the reference's SF atom (128 rows x 4 blocks = 512 B, 16 B per (row%32)) is read
//    directly: one dword per lane holds the 4 block scales of its row for a 128-K slab.
//
// Kernels by token count M:
//  * M <= 64 : mx_gemm_skinny.hip -- weight-streaming, 32 features per workgroup, K split over the 8 waves (for 48 < M <= 64 only
//                                    when no round of 32 x 64 tiles fits, for 32 < M <= 48 while N / 32 workgroups fit half a round
//                                    and K is not split, for M <= 32 while they fit three rounds: mx_gemm_small_m_uses_tiles)
//  * M  > 64 : mx_gemm256.hip     -- LDS-DMA pipelined 256x256 tiles when they fill the chip, else 128x256 / 128x128 tiles,
//                                    64x128 / 64x64 / 32x64 tiles with loader and compute waves for the smallest launches, or split-K
//                                    through a caller-provided workspace (plan_tiles / plan_splits, fitted to measurements)
#include <stdlib.h>

#include "mx_common.h"
#include "mx_kernels.h"

namespace mm {

hipError_t launch_mx_gemm(const GemmArgs &a, bool w4, hipStream_t stream) {
    if (a.M == 0 || a.N == 0) return hipSuccess;
    static const int skinny_max = getenv("MICROMIX_SKINNY_MAX_M") ? atoi(getenv("MICROMIX_SKINNY_MAX_M")) : 64;   // kernel-developer override (<= 64)
    if (a.M <= skinny_max && !mx_gemm_small_m_uses_tiles(a.M, a.N, a.K, w4, a.ws ? a.ws_bytes : 0, a.force_split != 0))
        return mx_gemm_stream_supported(a.M, a.N, a.K, w4) ? launch_mx_gemm_stream(a, w4, stream) : launch_mx_gemm_skinny(a, w4, stream);
    return launch_mx_gemm256(a, w4, stream);
}

}  // namespace mm
