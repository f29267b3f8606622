// The arithmetic of the reorder-free ("direct") MX quantizer, shared by direct_quantize.hip (activate_quantize_x, downproj_quantize_w)
// and by the fused gate/up epilogue of the tiled GEMM (mx_gemm_tile.inc: write_tile_act), so that both produce the same bytes.
// Restates mgemm/src/activate.cu:44-202: v = silu(float(a)) * float(b) in fp32; per 32 values amax = max |v|,
// scale = amax > 1e-6 ? 2^ceil(log2(amax / FMAX)) : 1.0 (byte 127), q = RNE_fmt(v / scale) -- see direct_quantize.hip for the stated
// deviations (exact exponent, hardware exp2 / rcp).
#pragma once
#include "mx_common.h"
// v_cvt_scalef32_2xpk16_bf6_f32 takes two 16-float operands: element order of the packed output, checked on hardware by
// tests/test_hw_gpu.py::test_f32_converters
#ifndef MM_BF6_LO
#define MM_BF6_LO(i) (2 * (i))
#define MM_BF6_HI(i) (2 * (i) + 1)
#endif

namespace mm {

typedef short ds2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned du6 __attribute__((ext_vector_type(6)));

// smallest e with FMAX * 2^e >= amax (amax > 0, any fp32), clamped to [-127, 127]
template <int EL>
__device__ __forceinline__ int scale_exponent_f32(float amax) {
    using T = ElemTraits<EL>;
    const uint32_t a = __float_as_uint(amax);
    const int exp = (int)(a >> 23);
    const uint32_t mant = a & 0x7FFFFFu;
    int e = exp - 127 - T::FMAX_EXP + (mant > T::FMAX_MANT ? 1 : 0);
    e = exp == 0 ? -127 : e;
    return e < -127 ? -127 : (e > 127 ? 127 : e);
}

// (GLOBAL_OUT: the fp4 codes leave through a write-through global store; false for destinations in LDS)
template <int EL, bool GLOBAL_OUT = true>
__device__ __forceinline__ uint32_t quantize32(const float (&v)[32], uint8_t *__restrict__ out) {
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
    int e = 0;                                   // scale 1.0
    if (amax > 1e-6f) e = scale_exponent_f32<EL>(amax);
    const int ec = e < -126 ? -126 : e;          // 2^-127 is not a normal fp32; only reachable for amax < FMAX * 2^-127
    const float scale = __uint_as_float((uint32_t)(127 + ec) << 23);
    if constexpr (EL == EL_FP8) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ds2 r = {0, 0};
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, v[4 * i], v[4 * i + 1], scale, false);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, v[4 * i + 2], v[4 * i + 3], scale, true);
            __builtin_memcpy(&w[i], &r, 4);
        }
        uint4 *o = reinterpret_cast<uint4 *>(out);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else if constexpr (EL == EL_FP4) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t r = 0;
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[8 * i], v[8 * i + 1], scale, 0);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[8 * i + 2], v[8 * i + 3], scale, 1);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[8 * i + 4], v[8 * i + 5], scale, 2);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[8 * i + 6], v[8 * i + 7], scale, 3);
            w[i] = r;
        }
        store16<GLOBAL_OUT>(out, w[0], w[1], w[2], w[3]);      // write-through: see store16 (mx_group_convert.h)
    } else {
        f16v lo, hi;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            lo[i] = v[MM_BF6_LO(i)];
            hi[i] = v[MM_BF6_HI(i)];
        }
        const du6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(lo, hi, scale);
        uint2 *o = reinterpret_cast<uint2 *>(out);
        o[0] = make_uint2(r[0], r[1]);
        o[1] = make_uint2(r[2], r[3]);
        o[2] = make_uint2(r[4], r[5]);
    }
    return (uint32_t)(e + 127);
}

// silu(x) * b = x / (1 + e^-x) * b with the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each; a few fp32 ulps in
// total, like the reference's CUDA expf, whose bits are not reproducible on other hardware either).  The full-precision expf +
// IEEE divide made activate_quantize_x ALU bound at 2.4 TB/s.
__device__ __forceinline__ float silu_mul(float x, float b) {
#ifdef MM_ACT_NOSILU      // ablation (results wrong): what the transcendentals cost in the fused epilogue
    return x * b;
#else
    const float ex = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return (x * __builtin_amdgcn_rcpf(1.0f + ex)) * b;
#endif
}

// eight bf16 (one 16-byte chunk) -> fp32
__device__ __forceinline__ void unpack8(const uint4 t, float *f) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(w[k] << 16);
        f[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
    }
}

}  // namespace mm
