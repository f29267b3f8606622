// 256x256-tile fused three-segment MX GEMM for gfx950 -- the large-M path of mm_matmul.
//
// Same arithmetic as mx_gemm.hip (see there for the reference citations: gemm.cu:26-78, w4a4.cu,
// w4a6.cu, w4a8.cu, w6a6.cu, w8a8.cu); different machine mapping, chosen from measurements on MI355X
// (tools/mfma_rate.py, profiles/):
//  * v_mfma_scale_f32_32x32x64_f8f6f4 sustains ~3.9 PF with the fp8 operand as srcA but only ~3.0-3.5 PF
//    with it as srcB, and fp6(A) x fp4(B) beats fp4(A) x fp6(B) the same way.  Activations are the wider
//    (or equal) format in every segment, so the ACTIVATION tile is srcA (MFMA rows = tokens) and the
//    WEIGHT tile srcB (MFMA columns = output features).
//  * a 128x128 tile needs ~47 B/clk/CU from L2 at that MFMA rate (L2 peak ~56): 256x256 halves it.
//  * 8 waves as 4 (tokens) x 2 (features), two per SIMD: each wave owns 64 x 128 outputs = 2 x 4 MFMA tiles and per
//    64-deep K step reads 2 activation + 4 weight fragments for 8 MFMAs.  (A 4-wave / one-per-SIMD variant with 128 x
//    128 per wave was measured: an LDS-DMA instruction costs the issuing wave ~180 cycles, more than the 64-cycle MFMA
//    it is meant to hide behind, so a single wave per SIMD leaves the matrix pipe idle ~45 % of the time.)
//  * the 128 fp32 accumulators of a wave live in a[0:127], driven by inline-asm MFMAs.  They are outside the
//    compiler's register allocation on purpose: with compiler-owned accumulators the three fused segment loops
//    made hipcc copy and spill them (hundreds of scratch dwords per lane, measured on ROCm 7.2).
//  * one 128-deep K slab per pipeline stage, three LDS stages (two with fp8 weights); operands AND the
//    scale-factor atoms arrive by LDS-DMA (buffer_load_dwordx4 ... lds) issued up to three slabs ahead behind
//    counted vmcnt waits, one DMA instruction between consecutive MFMAs so that their issue hides in the MFMA
//    shadow; one workgroup barrier per slab, placed between the two K steps, after the second step's
//    fragments are already in registers.
//  * with tokens on MFMA rows the accumulator has the feature index on the lane, so the epilogue transposes
//    each wave's tile through LDS (free at that point) and stores whole 256-byte rows.
#include <stdlib.h>
#include <type_traits>

#include "mx_acc_regs.h"
#ifndef MM_L2_PREFETCH
#define MM_L2_PREFETCH 0
#endif
#ifndef MM_DBG
#define MM_DBG 0  // kernel-developer ablation switches: 1 = no MFMA, 2 = no DMA (results are garbage)
#endif
#include "mx_common.h"
#include "mx_kernels.h"

namespace mm {

// hipcc parses __device__ bodies in its host pass as well; gfx950 inline asm and target builtins only exist in
// the device pass, so those few bodies are compiled for the device only.
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_DEVICE_ONLY(...)
#endif

#define MM_NS g256
#define MM_TM 2
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_TM
#define MM_NS g128
#define MM_TM 1
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_TM


template <class KernelT>
static hipError_t launch_tile(KernelT kern, bool &attr_done, int lds_bytes, int tiles, int threads, const GemmArgs &a,
                              hipStream_t stream) {
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_mx_gemm256(const GemmArgs &a, bool w4, hipStream_t stream) {
    static bool done[4] = {false, false, false, false};
    const int tn = (a.N + 255) / 256;
    const int tiles256 = ((a.M + 255) / 256) * tn, tiles128 = ((a.M + 127) / 128) * tn;
    // 256-row tiles move the fewest L2->LDS bytes per flop; use them when they (nearly) fill the 256 CUs, otherwise halve
    // the tile height so that twice as many workgroups exist.
    static int force = -1;
    if (force < 0) {
        const char *env = getenv("MICROMIX_GEMM_TILE");   // kernel-developer override: 256 or 128
        force = env ? atoi(env) : 0;
    }
    const bool use128 = force == 128 || (force != 256 && tiles256 < 192);
    if (!use128) {
        if (w4) return launch_tile(g256::mx_gemm256_kernel<true>, done[0], g256::Lds<true>::TOTAL, tiles256, g256::NT, a, stream);
        return launch_tile(g256::mx_gemm256_kernel<false>, done[1], g256::Lds<false>::TOTAL, tiles256, g256::NT, a, stream);
    }
    if (w4) return launch_tile(g128::mx_gemm256_kernel<true>, done[2], g128::Lds<true>::TOTAL, tiles128, g128::NT, a, stream);
    return launch_tile(g128::mx_gemm256_kernel<false>, done[3], g128::Lds<false>::TOTAL, tiles128, g128::NT, a, stream);
}

}  // namespace mm
