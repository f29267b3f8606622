// Shared by the quantizer kernels: one 32-element group (two bf16 per register) -> packed MX codes.
#pragma once
#include "mx_common.h"

namespace mm {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf32 __attribute__((ext_vector_type(32)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));

// two fp32 -> packed bf16 pair {lo, hi}, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
}

// v[i] = {element 2i (low half), element 2i+1 (high half)} as bf16 bits; out = RNE(v / scale), scale = the fp32 bit pattern
// (127 + e) << 23.  The converters read the scale operand as E8M0 -- only its exponent field counts -- so the pattern 0 means
// 2^-127, which no normal fp32 can express: a block whose absmax is below FMAX * 2^-127 (e = -127, scale byte 0) needs no path of
// its own (it used to have one, 2^127 times the value through an integer encoder: 1500 instructions of cold code per format that
// set the kernels' register count).  tests/probe_tiny_scale.py / tests/test_hw_gpu.py: scale patterns 0, 2^-127 and 2^-130 as
// denormals all convert every in-range bf16, denormals included, exactly as the oracle's encoder of 2^127 * x does.
// CDNA4 MX converters (v_cvt_scalef32_pk_fp4_bf16 / _pk_fp8_bf16 / _pk32_bf6_bf16: dst = RNE(src / scale), saturating) --
// tests/test_hw_gpu.py checks them code-for-code against the oracle's encoders for every finite bf16.
// GLOBAL_OUT: `out` is global memory (the stand-alone quantizers), so the fp4 codes may leave by a write-through store; the fused
// decode kernel quantizes into LDS with the same code and leaves it false.
template <int EL, bool GLOBAL_OUT = false>
__device__ __forceinline__ void convert_group(const uint32_t (&v)[16], float scale, uint8_t *__restrict__ out) {
    if constexpr (EL == EL_FP8) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf2 a, b;
            __builtin_memcpy(&a, &v[2 * i], 4);
            __builtin_memcpy(&b, &v[2 * i + 1], 4);
            s2 r = {0, 0};
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, a, scale, false);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, b, scale, true);
            __builtin_memcpy(&w[i], &r, 4);
        }
        store16<false>(out, w[0], w[1], w[2], w[3]);
        store16<false>(out + 16, w[4], w[5], w[6], w[7]);
    } else if constexpr (EL == EL_FP4) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t r = 0;
            bf2 a;
            __builtin_memcpy(&a, &v[4 * i], 4);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(r, a, scale, 0);
            __builtin_memcpy(&a, &v[4 * i + 1], 4);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(r, a, scale, 1);
            __builtin_memcpy(&a, &v[4 * i + 2], 4);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(r, a, scale, 2);
            __builtin_memcpy(&a, &v[4 * i + 3], 4);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(r, a, scale, 3);
            w[i] = r;
        }
        store16<GLOBAL_OUT>(out, w[0], w[1], w[2], w[3]);
    } else {
        bf32 x;
        __builtin_memcpy(&x, v, 64);
        const u6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(x, scale);
        uint2 *o = reinterpret_cast<uint2 *>(out);
        o[0] = make_uint2(r[0], r[1]);
        o[1] = make_uint2(r[2], r[3]);
        o[2] = make_uint2(r[4], r[5]);
    }
}

// (reorder_quantize.hip, rmsnorm_quantize.hip)
// The staged row is swizzled by 16-byte chunks: chunk q lives at q ^ ((q >> 4) & 3), i.e. inside every 1 KB the four 256-byte
// (= 64-bank) rows are rotated against each other.  A random reorder index does not notice; an index that walks the columns in
// order (the identity of mgemm/test.py, or a calibration that leaves whole runs in place) would otherwise send lane g to
// byte 64 g + 2 i: four banks for 32 lanes, an eight-way conflict (21.2 us at 4096 x 4096 instead of 11.2; tools/time_quant.py
// QUANT_IDX=identity); with the swizzle it is two-way.
__device__ __forceinline__ int swizzle_chunk(int q) { return q ^ ((q >> 4) & 3); }
// the same on two packed 16-bit byte offsets: bits 8-9 of each half are XORed into its bits 4-5
__device__ __forceinline__ uint32_t swizzle_offsets(uint32_t two) { return two ^ ((two >> 4) & 0x00300030u); }

// One 32-element group: gather (two bf16 per VGPR), block absmax, UE8M0 scale, convert, pack, store; returns the
// scale byte.  `ix` holds BYTE offsets into the staged row (index << 1), two per register.
// Conversion uses the CDNA4 MX converters (v_cvt_scalef32_pk_fp4_bf16 / _pk_fp8_bf16 / _pk32_bf6_bf16: dst =
// RNE(src / scale), saturating) -- tests/test_hw_gpu.py checks them code-for-code against the oracle's encoders for every
// finite bf16, and tests/test_quantize_gpu.py checks the kernel's bytes.
// gather_group is the same for every element format, so a wave whose lanes sit in different segments (an fp6 segment of four
// groups next to 60 fp8 groups) runs it once; only finish_group -- scale, conversion, store -- diverges.
__device__ __forceinline__ uint32_t gather_group(const uint8_t *__restrict__ row, const uint32_t (&ix)[16], uint32_t (&v)[16]) {
    us2 amax2 = {0, 0};   // v[i] = {element 2i (low half), element 2i+1 (high half)}
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t lo = *reinterpret_cast<const uint16_t *>(row + (ix[i] & 0xFFFFu));
        const uint32_t hi = *reinterpret_cast<const uint16_t *>(row + (ix[i] >> 16));
        v[i] = lo | (hi << 16);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
    }
    return amax2[0] > amax2[1] ? amax2[0] : amax2[1];   // bf16 magnitude bits of the block's absmax
}
template <int EL, bool GLOBAL_OUT = false>
__device__ __forceinline__ uint32_t finish_group(const uint32_t (&v)[16], uint32_t amax, uint8_t *__restrict__ out) {
    const int e = scale_exponent<EL>(amax << 16);   // -127 ... 127
    convert_group<EL, GLOBAL_OUT>(v, __uint_as_float((uint32_t)(127 + e) << 23), out);  // scale 2^e as an E8M0 pattern (see convert_group)
    return (uint32_t)(e + 127);
}
template <int EL, bool GLOBAL_OUT = false>
__device__ __forceinline__ uint32_t quantize_group(const uint8_t *__restrict__ row, const uint32_t (&ix)[16],
                                                   uint8_t *__restrict__ out) {
    uint32_t v[16];
    const uint32_t amax = gather_group(row, ix, v);
    return finish_group<EL, GLOBAL_OUT>(v, amax, out);
}

// The fp6 / fp8 codes of one row leave from an image of the row's [S | O] codes in LDS (written by the lanes that own the groups,
// read here after a barrier): 16 bytes per lane side by side, whole lines, write-through.  Straight from the lanes a store
// instruction covers 16 of every 32 (24) bytes: half lines, 3 us of the 11.3 of reorder_quantize at 4096 x 4096 all-fp8 (measured:
// no stores 8.3 us; the first half only 11.9; the same bytes as fully covered write-through instructions 8.9; fully covered
// but plain 11.6).  At most two chunks per thread: the image has at most K bytes and the workgroup at least K / 32 threads.
// (NCH: chunks per thread; rmsnorm_quantize_kernel<*, 2> covers K bytes with K / 64 threads: four)
template <int NCH = 2>
__device__ __forceinline__ void store_code_image(const uint8_t *image, int bytesS, int bytesO, uint8_t *oS, uint8_t *oO, int r) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int off = (threadIdx.x + i * blockDim.x) * 16;
        if (off < bytesS + bytesO) {
            const uint4 v = *reinterpret_cast<const uint4 *>(image + off);
            uint8_t *dst = off < bytesS ? oS + (size_t)r * bytesS + off : oO + (size_t)r * bytesO + (off - bytesS);
            store16<true>(dst, v.x, v.y, v.z, v.w);
        }
    }
}

}  // namespace mm
