// RMSNorm pieces shared by rmsnorm_quantize.hip (the stand-alone launch) and mx_decode_quant.h (the norm inside the decode launches):
// v = bf16((x * w) * rvar) per 32-column group, the block exponent, the reference's integer rounding (rmsnorm.cu:190-267).
#pragma once
#include "mx_group_convert.h"

namespace mm {

// One group in three steps, the first two the same for every element format, so that a wave whose lanes sit in different
// segments runs them once (only the conversion diverges):
//   rms_gather: v = bf16((x * w) * rvar) for the group's 32 columns, returns the absmax's bf16 magnitude bits.
//     PRODUCTS: `row` holds fp32 products x * w at the byte offsets in `ix` (wg unused); else bf16 x at `ix` and the weights in wg
template <bool PRODUCTS>
__device__ __forceinline__ uint32_t rms_gather(const uint8_t *__restrict__ row, const uint32_t (&ix)[16], const uint32_t (&wg)[16],
                                               float rvar, uint32_t (&v)[16]) {
    typedef float f2 __attribute__((ext_vector_type(2)));   // two-wide fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32)
    us2 amax2 = {0, 0};
    const f2 rvar2 = {rvar, rvar};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f2 xw;
        if constexpr (PRODUCTS) {
            xw = f2{*reinterpret_cast<const float *>(row + (ix[i] & 0xFFFFu)), *reinterpret_cast<const float *>(row + (ix[i] >> 16))};
        } else {
            const f2 x = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + (ix[i] & 0xFFFFu))),
                          bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + (ix[i] >> 16)))};
            const f2 w = {bf16_bits_to_f32(wg[i] & 0xFFFFu), bf16_bits_to_f32(wg[i] >> 16)};
            xw = x * w;
        }
        // (x * w) is exact in fp32 (two 8-bit significands); one rounding in the multiply by rvar, one to bf16
        const f2 r = xw * rvar2;
        v[i] = pack_bf16x2(r[0], r[1]);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
    }
    return amax2[0] > amax2[1] ? amax2[0] : amax2[1];
}
//   rms_scale: the block's exponent e for the lane's format (FMAX given by fexp / fmant); with INT_ROUND v becomes
//     round(v * 2^-e), half away from zero, and the conversion's scale 1; returns the scale pattern for convert_group.
//     (e = -127, a block below FMAX * 2^-127, is no special case: 2^127 * (1 + 2^-10) is a normal fp32, and the converters read the
//     scale pattern 0 as 2^-127 -- see convert_group.)
template <bool INT_ROUND>
__device__ __forceinline__ float rms_scale(uint32_t (&v)[16], uint32_t amax, int fexp, uint32_t fmant, int &e) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    e = scale_exponent_rt(amax << 16, fexp, fmant);
    if constexpr (INT_ROUND) {
        // t = v * 2^-e is exact and has 8 significant bits, so t * (1 + 2^-10) is exact in fp32 too, lies strictly between t and the
        // next point an 8-bit value could occupy, and is never a tie: rounding IT to nearest-even (v_rndne_f32) is rounding t half
        // away from zero (a tie k + 0.5 moves off the tie, away from zero; a non-tie is at least one 8-bit step from the nearest
        // tie, four times the nudge).  |t| >= 128 is an integer already and the nudge stays below 0.5.  The reference's clamp to
        // +-FMAX cannot bite: e is the smallest exponent with FMAX * 2^e >= amax, so |t| <= FMAX, an integer.
        // (Was trunc(t + copysign(0.5, t)): 9 VALU operations per pair, now 6.)
        // (e <= 126 for every finite bf16 absmax: FMAX * 2^126 >= 1.5 * 2^128 is beyond the format; only an inf / NaN block reaches
        // e = 127, whose exponent field would be 0 here -- a denormal multiplier -- so the multiplier's exponent is capped: such a
        // block stays inf / NaN through the conversion either way)
        const int em = e > 126 ? 126 : e;
        const float rs = __uint_as_float(((uint32_t)(127 - em) << 23) | 0x2000u);  // 2^-e * (1 + 2^-10)
        const f2 rs2 = {rs, rs};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f2 t = f2{bf16_bits_to_f32(v[i] & 0xFFFFu), bf16_bits_to_f32(v[i] >> 16)} * rs2;
            v[i] = pack_bf16x2(__builtin_rintf(t[0]), __builtin_rintf(t[1]));
        }
        return 1.0f;
    } else {
        return __uint_as_float((uint32_t)(127 + e) << 23);
    }
}
// the sum of squares of one of the reference's group threads: elements 8 (i T + t) + j, i = 0 .. 3, j = 0 .. 7, one after the other
// (rmsnorm.cu:143-160; `chunk(q)` returns 16-byte chunk q of the row as four dwords)
template <class Chunk>
__device__ __forceinline__ float rms_thread_sum(int t, int T, Chunk chunk) {
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 q = chunk(i * T + t);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = bf16_bits_to_f32(w[k] & 0xFFFFu), b = bf16_bits_to_f32(w[k] >> 16);
            sum = __builtin_fmaf(a, a, sum);  // a*a is exact: the fused and the unfused forms round identically
            sum = __builtin_fmaf(b, b, sum);
        }
    }
    return sum;
}
// the halving tree s[t] += s[t + stride], stride = P/2 ... 1, over P <= 256 partial sums (zero padded), walked by ONE wave: lane l holds
// s[l + 64 j]; then rvar = 1 / sqrt(sum / K + eps) with the correctly rounded divide and square root (see rmsnorm_quantize.hip)
__device__ __forceinline__ float rms_tree_rvar(const float *part, int P, int lane, int K, float eps) {
    float v[4] = {part[lane], 0.0f, 0.0f, 0.0f};
    const int n0 = P >> 6;      // 1, 2 or 4
    if (n0 > 1) v[1] = part[lane + 64];
    if (n0 > 2) { v[2] = part[lane + 128]; v[3] = part[lane + 192]; }
    if (n0 > 2) { v[0] += v[2]; v[1] += v[3]; }
    if (n0 > 1) v[0] += v[1];
    float s = v[0];
#pragma unroll
    for (int st = 32; st >= 1; st >>= 1) s += __shfl_down(s, st, 64);
    s = __shfl(s, 0, 64);
    return __fdiv_rn(1.0f, __builtin_sqrtf(__fdiv_rn(s, (float)K) + eps));
}

}  // namespace mm
