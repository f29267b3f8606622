// Device-side helpers shared by the gfx950 kernels: MX element encoders, UE8M0
// block-scale rule and the scale-factor tensor layout.
//
// Semantics restate (not copy) the MicroMix reference:
//   element formats / maxima   mgemm/src/reorder.cu:17-19, include/reorder.cuh:37-41
//   scale rule                 mgemm/src/reorder.cu:175-209   (e = ceil(log2(amax/FMAX)))
//   SF layout                  mgemm/include/sm120_sf_layout.h:170-173
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm {

// Hardware format codes of v_mfma_scale_f32_*_f8f6f4 (cbsz / blgp fields).
enum : int { HW_FP8 = 0, HW_BF8 = 1, HW_FP6 = 2, HW_BF6 = 3, HW_FP4 = 4 };
// Element kinds used by this library (the reference's "FP6" is E3M2 = hardware BF6).
enum : int { EL_FP4 = 0, EL_FP6 = 1, EL_FP8 = 2 };

template <int EL> struct ElemTraits;
template <> struct ElemTraits<EL_FP4> {
    static constexpr int EB = 2, MB = 1, BIAS = 1, MAXCODE = 0x7, HW = HW_FP4;
    static constexpr int GROUP_BYTES = 16;            // 32 elements
    static constexpr uint32_t FMAX_MANT = 0x400000;   // 6  = 1.5  * 2^2
    static constexpr int FMAX_EXP = 2;
};
template <> struct ElemTraits<EL_FP6> {
    static constexpr int EB = 3, MB = 2, BIAS = 3, MAXCODE = 0x1F, HW = HW_BF6;
    static constexpr int GROUP_BYTES = 24;
    static constexpr uint32_t FMAX_MANT = 0x600000;   // 28 = 1.75 * 2^4
    static constexpr int FMAX_EXP = 4;
};
template <> struct ElemTraits<EL_FP8> {
    static constexpr int EB = 4, MB = 3, BIAS = 7, MAXCODE = 0x7E, HW = HW_FP8;
    static constexpr int GROUP_BYTES = 32;
    static constexpr uint32_t FMAX_MANT = 0x600000;   // 448 = 1.75 * 2^8
    static constexpr int FMAX_EXP = 8;
};

// Smallest e with FMAX * 2^e >= amax (amax given as the fp32 bit pattern of a
// non-negative bf16 value); zero block -> -1 (scale 0.5, SF byte 126);
// clamped to [-127, 127] so that e + 127 is a valid UE8M0 byte.
template <int EL>
__device__ __forceinline__ int scale_exponent(uint32_t amax_bits) {
    using T = ElemTraits<EL>;
    const int exp = (int)(amax_bits >> 23);
    const uint32_t mant = amax_bits & 0x7FFFFFu;
    int e = exp - 127 - T::FMAX_EXP + (mant > T::FMAX_MANT ? 1 : 0);
    e = exp == 0 ? -127 : e;
    e = e < -127 ? -127 : (e > 127 ? 127 : e);
    return amax_bits == 0 ? -1 : e;
}

// the same with the format's FMAX = (1 + fmax_mant / 2^23) * 2^fmax_exp as run-time values (lanes of one wave in different segments)
__device__ __forceinline__ int scale_exponent_rt(uint32_t amax_bits, int fmax_exp, uint32_t fmax_mant) {
    const int exp = (int)(amax_bits >> 23);
    const uint32_t mant = amax_bits & 0x7FFFFFu;
    int e = exp - 127 - fmax_exp + (mant > fmax_mant ? 1 : 0);
    e = exp == 0 ? -127 : e;
    e = e < -127 ? -127 : (e > 127 ? 127 : e);
    return amax_bits == 0 ? -1 : e;
}

// fp32 -> element code, round-to-nearest-even, saturating, sign kept on zero.
template <int EL>
__device__ __forceinline__ uint32_t encode(float x) {
    using T = ElemTraits<EL>;
    constexpr int EMIN = 1 - T::BIAS;
    constexpr int SHIFT = 23 - T::MB;
    const uint32_t u = __float_as_uint(x);
    const uint32_t sign = u >> 31;
    const uint32_t a = u & 0x7FFFFFFFu;
    // target-subnormal range: RNE by a magic fp32 add whose ulp is the subnormal quantum
    constexpr uint32_t MAGIC = (uint32_t)(127 + EMIN - T::MB + 23) << 23;
    const uint32_t code_s = __float_as_uint(__uint_as_float(a) + __uint_as_float(MAGIC)) - MAGIC;
    // target-normal range
    const uint32_t r = a + ((1u << (SHIFT - 1)) - 1u) + ((a >> SHIFT) & 1u);
    const uint32_t code_n = (r >> SHIFT) - ((uint32_t)(127 - T::BIAS) << T::MB);
    uint32_t code = a < ((uint32_t)(127 + EMIN) << 23) ? code_s : code_n;
    code = code > (uint32_t)T::MAXCODE ? (uint32_t)T::MAXCODE : code;
    return code | (sign << (T::EB + T::MB));
}

// Byte offset of the scale of (row r, 32-block j) inside one segment of kseg columns.
__host__ __device__ __forceinline__ size_t sf_offset(int r, int j, int kseg) {
    return (size_t)(r >> 7) * (size_t)(kseg >> 7) * 512u + (size_t)(j >> 2) * 512u + (size_t)(r & 31) * 16u +
           (size_t)((r >> 5) & 3) * 4u + (size_t)(j & 3);
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// fp32 -> bf16 bits, RNE.  Finite inputs only on the paths that use it (accumulators); a NaN
// accumulator stays a NaN through the quiet-bit branch.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    const uint32_t r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    const bool nan = (u & 0x7FFFFFFFu) > 0x7F800000u;
    return nan ? ((u >> 16) | 0x40u) : r;
}

// 16-byte store of packed codes.  WT = write-through (sc0 sc1): the quantizer's output is read next by another kernel on any XCD, so
// keeping it dirty in this XCD's L2 only leaves a flush for the end of the kernel.  It pays where a wave instruction writes whole
// lines -- the fp4 segments, 16 contiguous bytes per lane: 4096 x 4096 all-fp4 9.56 -> 8.26 us, (2048,128,1920) 11.2 -> 9.1 us --
// and costs where a lane's 32 bytes go out as two half-covered instructions (fp8: 11.1 -> 11.7 us), so fp8 / fp6 stay plain.
template <bool WT>
__device__ __forceinline__ void store16(uint8_t *out, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WT) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u vv = {a, b, c, d};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out), "v"(vv) : "memory");
        return;
    }
#endif
    *reinterpret_cast<uint4 *>(out) = make_uint4(a, b, c, d);
}

// The four scale bytes of a row and one 128-column slab: one dword, write-through.  A 128-byte line of a scale tensor collects the
// entries of 32 rows, every row from another workgroup, so these are the most scattered stores of the quantizers (16 lines per wave
// instruction, 4 bytes each): 1 % of the bytes but 0.7 us of reorder_quantize's 9.3 at 4096 x 4096 (measured without them).  As
// plain stores their dirty lines wait in the L2s for the end of the kernel; write-through takes 0.35 us off (8.65 -> 8.3 us on
// (2048,128,1920)).  Dealing the rows to workgroups so that all writers of a line share an XCD (one L2 merges them) did not help.
// (reorder_quantize and rmsnorm_quantize; in direct_quantize.hip, whose waves write the scales of 64 consecutive groups, the same
// write-through COSTS 3 us of 41: it keeps plain stores.)
__device__ __forceinline__ void store_scale_dword(uint8_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<uint32_t *>(p) = v;
#endif
}

}  // namespace mm
