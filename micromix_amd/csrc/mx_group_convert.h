// Shared by the quantizer kernels: one 32-element group (two bf16 per register) -> packed MX codes.
#pragma once
#include "mx_common.h"

namespace mm {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf32 __attribute__((ext_vector_type(32)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));

// e == -127 (every element of the block is below FMAX * 2^-127): 2^127 times the value, integer encoder.
// Inlined on purpose: as a __noinline__ call it cost 40 % of the kernel's time (15.3 vs 10.9 us at 4096 x 4096) although
// it is practically never taken -- the call site pins the caller's registers.
#define MM_TINY_INLINE __forceinline__
// two fp32 -> packed bf16 pair {lo, hi}, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
}

// round to an integer, half away from zero (roundf), clamp to the format's range: the extra step of the reference's rmsnorm
// quantizer (rmsnorm.cu:262-267), before its bf16 rounding.  x + copysign(0.5, x) is exact or, for |x| < 2^-17, rounds to
// +-0.5 and truncates to 0 either way.
template <int EL>
__device__ __forceinline__ float integer_round_clamp(float x) {
    constexpr float FM = EL == EL_FP4 ? 6.0f : (EL == EL_FP6 ? 28.0f : 448.0f);
    const float r = __builtin_truncf(x + __builtin_copysignf(0.5f, x));
    return __builtin_fminf(__builtin_fmaxf(r, -FM), FM);
}
template <int EL>
__device__ __forceinline__ float integer_round_clamp_bf16(float x) {
    return bf16_bits_to_f32(f32_to_bf16_bits(integer_round_clamp<EL>(x)));
}

template <int EL, bool INT_ROUND = false>
__device__ MM_TINY_INLINE void quantize_group_tiny(const uint32_t *__restrict__ v, uint8_t *__restrict__ out) {
    const float rs = __uint_as_float(254u << 23);  // 2^127
    uint32_t c[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float lo = bf16_bits_to_f32(v[i] & 0xFFFFu) * rs, hi = bf16_bits_to_f32(v[i] >> 16) * rs;
        if constexpr (INT_ROUND) {
            lo = integer_round_clamp_bf16<EL>(lo);
            hi = integer_round_clamp_bf16<EL>(hi);
        }
        c[2 * i] = encode<EL>(lo);
        c[2 * i + 1] = encode<EL>(hi);
    }
    if constexpr (EL == EL_FP8) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = c[4 * i] | (c[4 * i + 1] << 8) | (c[4 * i + 2] << 16) | (c[4 * i + 3] << 24);
        uint4 *o = reinterpret_cast<uint4 *>(out);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else if constexpr (EL == EL_FP4) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) x |= (c[8 * i + k] & 0xFu) << (4 * k);  // element 2i in the low nibble
            w[i] = x;
        }
        *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        // dense little-endian 6-bit stream: 32 codes -> 192 bits -> three 64-bit words
        unsigned long long w[3] = {0ull, 0ull, 0ull};
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i, word = bit >> 6, off = bit & 63;
            const unsigned long long code = c[i] & 0x3Fu;
            w[word] |= code << off;
            if (off > 58) w[word + 1] |= code >> (64 - off);
        }
        unsigned long long *o = reinterpret_cast<unsigned long long *>(out);
        o[0] = w[0];
        o[1] = w[1];
        o[2] = w[2];
    }
}

// v[i] = {element 2i (low half), element 2i+1 (high half)} as bf16 bits; out = RNE(v / scale), scale a normal fp32 power of two.
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
template <int EL, bool GLOBAL_OUT = false>
__device__ __forceinline__ uint32_t quantize_group(const uint8_t *__restrict__ row, const uint32_t (&ix)[16],
                                                   uint8_t *__restrict__ out) {
    uint32_t v[16];  // v[i] = {element 2i (low half), element 2i+1 (high half)}
    us2 amax2 = {0, 0};
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
    const uint32_t amax = amax2[0] > amax2[1] ? amax2[0] : amax2[1];
    const int e = scale_exponent<EL>(amax << 16);
    if (e == -127) {
        quantize_group_tiny<EL>(v, out);
        return 0u;
    }
    convert_group<EL, GLOBAL_OUT>(v, __uint_as_float((uint32_t)(127 + e) << 23), out);  // scale 2^e, a normal fp32
    return (uint32_t)(e + 127);
}

}  // namespace mm
