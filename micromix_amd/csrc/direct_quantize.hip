// Reorder-free MX quantizers for gfx950: the fused silu(gate)*up activation quantizer
// that feeds down_proj, and the weight-side quantizers whose column order was already folded into the producing layers.
//
// Restates (does not copy) mgemm/src/activate.cu:
//   activate_quantize_kernel_with_cute_layout        :44-202   v = silu(float(a)) * float(b), silu(x) = x / (1 + expf(-x))
//   downproj_quantize_kernel_with_cute_layout[_w4]   :208-500  v = float(w)
// For every row and 32-wide group of NATURAL column order: amax = max|v| (fp32);
//   scale = amax > 1e-6 ? 2^ceil(log2(amax / FMAX)) : 1.0   (note: 1.0 / byte 127 for an empty block, unlike the reorder
//   kernel's 0.5 / byte 126);  q = RNE_fmt(clamp(v / scale))  -- a single rounding from fp32, no bf16 step.
// Stated deviation: the exponent is the smallest e with FMAX * 2^e >= amax computed exactly; the reference evaluates
// ceilf(log2f(amax / FMAX)) in fp32, which can come out one lower when amax is within an fp32 ulp above FMAX * 2^k.
//
// MI355X mapping: pure streaming (HBM-bound): one thread per group, 64 contiguous input bytes per operand per lane
// (4 KiB contiguous per wave), the CDNA4 fp32 MX converters (v_cvt_scalef32_pk_fp4_f32 / _pk_fp8_f32 / _2xpk16_bf6_f32),
// scale bytes of a lane quad merged into one dword store.
#include "mx_common.h"
// v_cvt_scalef32_2xpk16_bf6_f32 takes two 16-float operands: element order of the packed output, checked on hardware by
// tests/test_hw_gpu.py::test_f32_converters
#ifndef MM_BF6_LO
#define MM_BF6_LO(i) (2 * (i))
#define MM_BF6_HI(i) (2 * (i) + 1)
#endif
#include "mx_kernels.h"

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

template <int EL>
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
        store16<true>(out, w[0], w[1], w[2], w[3]);      // write-through: see store16 (mx_group_convert.h)
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

__device__ __forceinline__ void load32(const uint16_t *__restrict__ p, float (&f)[32]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 t = q[i];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[8 * i + 2 * k] = __uint_as_float(w[k] << 16);
            f[8 * i + 2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
        }
    }
}

// MODE 0: silu(A) * B -> fp4|fp6|fp8;  MODE 1: A -> fp4|fp6|fp8;  MODE 2: A -> fp4|fp4|fp4
template <int MODE>
__global__ void __launch_bounds__(256)
direct_quantize_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B, int rows, int KN, int KS, int KO,
                       uint8_t *__restrict__ oN, uint8_t *__restrict__ oS, uint8_t *__restrict__ oO,
                       uint8_t *__restrict__ sfN, uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    const int K = KN + KS + KO, G = K >> 5, gN = KN >> 5, gS = KS >> 5;
    const long long total = (long long)rows * G;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(id / G), g = (int)(id - (long long)r * G);
        float v[32];
        load32(A + (size_t)r * K + (size_t)g * 32, v);
        if constexpr (MODE == 0) {
            float b[32];
            load32(B + (size_t)r * K + (size_t)g * 32, b);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                // silu(x) * b = x / (1 + e^-x) * b with the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each; a
                // few fp32 ulps in total, like the reference's CUDA expf, whose bits are not reproducible on other hardware
                // either).  The full-precision expf + IEEE divide made this kernel ALU bound at 2.4 TB/s.
                const float e = __builtin_amdgcn_exp2f(v[i] * -1.4426950408889634f);
                v[i] = (v[i] * __builtin_amdgcn_rcpf(1.0f + e)) * b[i];
            }
        }
        uint32_t byte;
        uint8_t *sf;
        int j, kseg;
        if (g < gN) {
            j = g; kseg = KN; sf = sfN;
            byte = quantize32<EL_FP4>(v, oN + (size_t)r * (KN >> 1) + j * 16);
        } else if (g < gN + gS) {
            j = g - gN; kseg = KS; sf = sfS;
            if constexpr (MODE == 2) byte = quantize32<EL_FP4>(v, oS + (size_t)r * (KS >> 1) + j * 16);
            else byte = quantize32<EL_FP6>(v, oS + (size_t)r * (KS / 4 * 3) + j * 24);
        } else {
            j = g - gN - gS; kseg = KO; sf = sfO;
            if constexpr (MODE == 2) byte = quantize32<EL_FP4>(v, oO + (size_t)r * (KO >> 1) + j * 16);
            else byte = quantize32<EL_FP8>(v, oO + (size_t)r * KO + j * 32);
        }
        // G and the segment widths are multiples of 4 groups: a lane quad always holds 4 consecutive blocks of one row
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
        const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
        if ((g & 3) == 0) *reinterpret_cast<uint32_t *>(sf + sf_offset(r, j, kseg)) = byte | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
}

hipError_t launch_direct_quantize(const void *A, const void *B, int rows, int KN, int KS, int KO, int mode, uint8_t *oN,
                                  uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const long long total = (long long)rows * ((KN + KS + KO) / 32);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    auto kern = mode == 0 ? direct_quantize_kernel<0> : (mode == 1 ? direct_quantize_kernel<1> : direct_quantize_kernel<2>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint16_t *)A, (const uint16_t *)B, rows, KN, KS, KO,
                       oN, oS, oO, sfN, sfS, sfO);
    return hipGetLastError();
}

}  // namespace mm
