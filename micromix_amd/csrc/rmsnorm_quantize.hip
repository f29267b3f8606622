// RMSNorm fused with column-reorder + per-32-group MXFP4/MXFP6/MXFP8 quantize for gfx950.
//
// Reference: rmsnorm_bf16_mixed_kernel, mgemm/src/rmsnorm.cu:95-312 (binding bindings.cpp:257-303).  Per row:
//   rvar  = 1 / sqrt(sum(x^2) / K + eps)                     fp32, the reference's summation order (see below)
//   v[i]  = bf16((float(x[idx[i]]) * float(w[idx[i]])) * rvar)                                    rmsnorm.cu:190-195
//   per 32-group of reordered positions: amax, e = ceil(log2(amax / FMAX)) (amax == 0 -> scale 0.5)       :216-245
//   q[i]  = RNE_fmt(bf16(clamp(round(v[i] * 2^-e), +-FMAX)))     -- the reference rounds to an INTEGER first     :262-267
// (MM_RMS_NO_INTEGER_ROUND drops the round(); the result is then reorder_quantize of the normalised row.)
//
// Summation order (bit-exact against oracle/mx_oracle.py: rmsnorm_rvar): group thread t of T = K/32 adds the squares of
// elements i*K/4 + 8t + j, i = 0..3, j = 0..7, one after the other -- these are exactly the four 16-byte chunks the thread
// stages into LDS -- and the T partial sums meet in the halving tree s[t] += s[t + stride] over the next power of two,
// zero padded (the reference hard-codes that tree for T = 128 and is wrong for its other K; see the oracle's note).
// fp32 divide and square root are the correctly rounded ones (the reference's rsqrt() approximation is not reproducible):
// __fdiv_rn and __builtin_sqrtf.  NOT __fsqrt_rn, which on this toolchain is the native v_sqrt_f32 -- one ulp off for 15 % of
// arguments (tools/probe_rvar.hip); rounds 1-2 used it, and about one row in 7000 then had an element one code off against the
// oracle (found by tests/quant_stress.py in round 3; the fixed test cases had happened to miss it).
//
// Layout as reorder_quantize.hip: one workgroup strides over rows, thread t owns reordered group t, its 32 indices stay in
// registers for the whole launch, the row is staged in LDS by coalesced 16-byte loads.
//   K <= 8192 (rmsnorm_quantize_products_kernel): a thread keeps the norm weights of the four chunks it STAGES (natural order:
//   four coalesced 16-byte loads) and stages the fp32 products x * w -- exact, two 8-bit significands -- so the gather reads the
//   product with one ds_read_b32 and there is no gather of the weights at all.  The 32-bit row is laid out in two planes
//   (elements 0-3 of every chunk, then elements 4-7) so that the staging writes stay whole 16-byte lanes side by side.  Every
//   wave sums the partial sums by itself after ONE barrier (same tree, same order): two barriers per row instead of four.
//   K > 8192 (rmsnorm_quantize_kernel): the 16-bit row and 32 gathered norm weights per thread in registers (the 32-bit row
//   of K = 32768 would not fit the LDS).
#include "mx_group_convert.h"
#include "mx_kernels.h"

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
//   then convert_group<EL> into global memory (fp4: 16 bytes per lane, whole lines as they are) or into the LDS image of the row's
//   fp6 / fp8 codes (store_code_image, mx_group_convert.h).
template <bool INT_ROUND, bool PRODUCTS>
__device__ __forceinline__ uint32_t rms_group(const uint8_t *__restrict__ row, const uint32_t (&ix)[16], const uint32_t (&wg)[16],
                                              float rvar, int seg, int j, int r, int KN, uint8_t *oN, uint8_t *image, int bytesS) {
    uint32_t v[16];
    const uint32_t amax = rms_gather<PRODUCTS>(row, ix, wg, rvar, v);
    const int fexp = seg == 0 ? ElemTraits<EL_FP4>::FMAX_EXP : seg == 1 ? ElemTraits<EL_FP6>::FMAX_EXP : ElemTraits<EL_FP8>::FMAX_EXP;
    const uint32_t fmant = seg == 0 ? ElemTraits<EL_FP4>::FMAX_MANT : seg == 1 ? ElemTraits<EL_FP6>::FMAX_MANT : ElemTraits<EL_FP8>::FMAX_MANT;
    int e;
    const float scale = rms_scale<INT_ROUND>(v, amax, fexp, fmant, e);
    if (seg == 0) convert_group<EL_FP4, true>(v, scale, oN + (size_t)r * (KN >> 1) + j * 16);
    else if (seg == 1) convert_group<EL_FP6>(v, scale, image + j * 24);
    else convert_group<EL_FP8>(v, scale, image + bytesS + j * 32);
    return (uint32_t)(e + 127);
}

template <bool INT_ROUND, int MAXT>
__global__ void __launch_bounds__(MAXT)
rmsnorm_quantize_kernel(const uint16_t *__restrict__ src, const uint16_t *__restrict__ weight, float eps, int rows, int K,
                        const int16_t *__restrict__ idx, int KN, int KS, int KO, uint8_t *__restrict__ oN,
                        uint8_t *__restrict__ oS, uint8_t *__restrict__ oO, uint8_t *__restrict__ sfN,
                        uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [K bf16 row][P floats of partial sums]
    const int T = K >> 5;  // group threads = stager threads: thread t stages chunks t, T + t, 2T + t, 3T + t
    const int g = threadIdx.x;
    const bool active = g < T;
    float *part = reinterpret_cast<float *>(smem + (size_t)K * 2);
    int P = 64;
    while (P < T) P <<= 1;
    const int bytesS = KS / 4 * 3;
    uint8_t *image = smem + (size_t)K * 2 + (size_t)(P > (int)blockDim.x ? P : (int)blockDim.x) * 4;   // [row][partial sums][image of the S | O codes]

    // the norm weights of this thread's 32 columns: the weight vector is staged in LDS (coalesced) and gathered from there
    // with the same byte offsets as the row (32 scattered 2-byte global loads per thread cost more than the two rows a
    // workgroup typically processes)
    uint32_t ix[16], wg[16];
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            reinterpret_cast<uint4 *>(smem)[swizzle_chunk(i * T + g)] = reinterpret_cast<const uint4 *>(weight)[i * T + g];
    }
    __syncthreads();
    if (active) {
        const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)g * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = p[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t both = swizzle_offsets((w[k] << 1) & 0xFFFEFFFEu);       // byte offsets into a staged (chunk-swizzled) [K] bf16 vector
                const uint32_t b0 = both & 0xFFFFu, b1 = both >> 16;
                ix[4 * i + k] = b0 | (b1 << 16);
                wg[4 * i + k] = (uint32_t)*reinterpret_cast<const uint16_t *>(smem + b0) |
                                ((uint32_t)*reinterpret_cast<const uint16_t *>(smem + b1) << 16);
            }
        }
    }
    __syncthreads();   // the row staging below reuses the same LDS bytes
    const int gN = KN >> 5, gS = KS >> 5;
    int seg, j, kseg;
    if (g < gN) { seg = 0; j = g; kseg = KN; }
    else if (g < gN + gS) { seg = 1; j = g - gN; kseg = KS; }
    else { seg = 2; j = g - gN - gS; kseg = KO; }

    // the next row's four chunks are loaded into registers before the current row is processed (HBM latency hides under
    // the gather), and stored to LDS -- squares summed on the way -- after the barrier that ends the current row
    uint4 stage[4];
    auto fetch = [&](int r) {
        if (active && r < rows) {
            const uint4 *grow = reinterpret_cast<const uint4 *>(src + (size_t)r * K);
#pragma unroll
            for (int i = 0; i < 4; ++i) stage[i] = grow[i * T + g];
        }
    };
    auto stage_and_sum = [&]() -> float {
        float sum = 0.0f;
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                reinterpret_cast<uint4 *>(smem)[swizzle_chunk(i * T + g)] = stage[i];
                const uint32_t w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf16_bits_to_f32(w[k] & 0xFFFFu), b = bf16_bits_to_f32(w[k] >> 16);
                    sum = __builtin_fmaf(a, a, sum);  // a*a is exact: the fused and the unfused forms round identically
                    sum = __builtin_fmaf(b, b, sum);
                }
            }
        }
        return sum;
    };
    // (the 1024-thread variant, K > 16384, has only 128 registers per lane: it fetches each row when it needs it instead
    // of one row ahead)
    constexpr bool PREFETCH = MAXT <= 512;
    if constexpr (PREFETCH) fetch(blockIdx.x);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        if constexpr (!PREFETCH) fetch(r);
        const float sum = stage_and_sum();
        if constexpr (PREFETCH) fetch(r + gridDim.x);
        part[g] = sum;              // threads T.. contribute the zero padding (P < 2 * blockDim.x)
        if (g + (int)blockDim.x < P) part[g + blockDim.x] = 0.0f;
        __syncthreads();
        for (int stride = P >> 1; stride >= 64; stride >>= 1) {
            if (g < stride) part[g] += part[g + stride];
            __syncthreads();
        }
        if (g < 64) {
            float s = part[g];
#pragma unroll
            for (int stride = 32; stride >= 1; stride >>= 1) s += __shfl_down(s, stride, 64);
            if (g == 0) part[0] = __fdiv_rn(1.0f, __builtin_sqrtf(__fdiv_rn(s, (float)K) + eps));
        }
        __syncthreads();
        const float rvar = part[0];
        if (active) {
            const uint8_t *row = smem;
            uint32_t byte;
            uint8_t *sf;
            byte = rms_group<INT_ROUND, false>(row, ix, wg, rvar, seg, j, r, KN, oN, image, bytesS);
            sf = seg == 0 ? sfN : seg == 1 ? sfS : sfO;
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            if ((g & 3) == 0)
                store_scale_dword(sf + sf_offset(r, j, kseg), byte | (b1 << 8) | (b2 << 16) | (b3 << 24));
        }
        __syncthreads();  // the row and part[] are rewritten by the next iteration
        store_code_image(image, bytesS, KO, oS, oO, r);   // (read here, rewritten only after the next iteration's first barrier)
    }
}

// K <= 8192: the row is staged as fp32 products x * w (see the header).  LDS: [plane A: K floats' first halves][plane B][P partial sums];
// element c = 8q + e lives at byte (e < 4 ? 0 : 2K) + 16 q' + 4(e & 3), q' = swizzle_chunk(q) (mx_group_convert.h), so chunk q's two
// halves are two conflict-free 16-byte writes.
template <bool INT_ROUND>
// (167 VGPRs: six workgroups per CU.  Bounding it to 128 for eight -- __launch_bounds__(256, 4) -- spills 38 registers: 14.4 -> 25.9 us.)
__global__ void __launch_bounds__(256)
rmsnorm_quantize_products_kernel(const uint16_t *__restrict__ src, const uint16_t *__restrict__ weight, float eps, int rows, int K,
                                 const int16_t *__restrict__ idx, int KN, int KS, int KO, uint8_t *__restrict__ oN,
                                 uint8_t *__restrict__ oS, uint8_t *__restrict__ oO, uint8_t *__restrict__ sfN,
                                 uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int T = K >> 5;
    const int g = threadIdx.x;
    const bool active = g < T;
    const uint32_t planeB = (uint32_t)K * 2;
    float *part = reinterpret_cast<float *>(smem + (size_t)K * 4);
    int P = 64;
    while (P < T) P <<= 1;
    const int bytesS = KS / 4 * 3;
    uint8_t *image = smem + (size_t)K * 4 + (size_t)(P > (int)blockDim.x ? P : (int)blockDim.x) * 4;   // [planes][partial sums][image of the S | O codes]

    uint32_t ix[16];
    const uint32_t none[16] = {};
    uint4 wch[4];
    if (active) {
        const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)g * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = p[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // both 16-bit column indices of the register at once (no carries between the halves: every intermediate stays below
                // 2^16): chunk q = c >> 3, swizzled (q ^ ((q >> 4) & 3)), byte = plane(c & 4) + 16 q' + 4 (c & 3)
                uint32_t q2 = (w[k] >> 3) & 0x0FFF0FFFu;                 // K <= 8192: q < 1024
                q2 ^= (q2 >> 4) & 0x00030003u;
                ix[4 * i + k] = (q2 << 4) + ((w[k] & 0x00030003u) << 2) + ((w[k] >> 2) & 0x00010001u) * planeB;   // < 4K <= 32768 each
            }
            wch[i] = reinterpret_cast<const uint4 *>(weight)[i * T + g];
        }
    }
    const int gN = KN >> 5, gS = KS >> 5;
    int seg, j, kseg;
    if (g < gN) { seg = 0; j = g; kseg = KN; }
    else if (g < gN + gS) { seg = 1; j = g - gN; kseg = KS; }
    else { seg = 2; j = g - gN - gS; kseg = KO; }

    uint4 stage[4];
    auto fetch = [&](int r) {
        if (active && r < rows) {
            const uint4 *grow = reinterpret_cast<const uint4 *>(src + (size_t)r * K);
#pragma unroll
            for (int i = 0; i < 4; ++i) stage[i] = grow[i * T + g];
        }
    };
    auto stage_and_sum = [&]() -> float {
        float sum = 0.0f;
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t x[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
                const uint32_t w[4] = {wch[i].x, wch[i].y, wch[i].z, wch[i].w};
                float pr[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf16_bits_to_f32(x[k] & 0xFFFFu), b = bf16_bits_to_f32(x[k] >> 16);
                    sum = __builtin_fmaf(a, a, sum);  // a*a is exact: the fused and the unfused forms round identically
                    sum = __builtin_fmaf(b, b, sum);
                    pr[2 * k] = a * bf16_bits_to_f32(w[k] & 0xFFFFu);
                    pr[2 * k + 1] = b * bf16_bits_to_f32(w[k] >> 16);
                }
                const int q = swizzle_chunk(i * T + g);
                reinterpret_cast<float4 *>(smem)[q] = make_float4(pr[0], pr[1], pr[2], pr[3]);
                reinterpret_cast<float4 *>(smem + planeB)[q] = make_float4(pr[4], pr[5], pr[6], pr[7]);
            }
        }
        return sum;
    };
    fetch(blockIdx.x);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float sum = stage_and_sum();
        fetch(r + gridDim.x);
        part[g] = sum;              // threads T.. contribute the zero padding (P < 2 * blockDim.x)
        if (g + (int)blockDim.x < P) part[g + blockDim.x] = 0.0f;
        __syncthreads();
        // every wave walks the halving tree s[t] += s[t + stride], stride = P/2 ... 1, by itself: lane l holds s[l + 64 j]
        float rvar;
        {
            const int l = g & 63;
            float v[4] = {part[l], 0.0f, 0.0f, 0.0f};
            const int n0 = P >> 6;      // 1, 2 or 4
            if (n0 > 1) v[1] = part[l + 64];
            if (n0 > 2) { v[2] = part[l + 128]; v[3] = part[l + 192]; }
            if (n0 > 2) { v[0] += v[2]; v[1] += v[3]; }
            if (n0 > 1) v[0] += v[1];
            float s = v[0];
#pragma unroll
            for (int stride = 32; stride >= 1; stride >>= 1) s += __shfl_down(s, stride, 64);
            s = __shfl(s, 0, 64);
            rvar = __fdiv_rn(1.0f, __builtin_sqrtf(__fdiv_rn(s, (float)K) + eps));
        }
        if (active) {
            const uint8_t *row = smem;
            uint32_t byte;
            uint8_t *sf;
            byte = rms_group<INT_ROUND, true>(row, ix, none, rvar, seg, j, r, KN, oN, image, bytesS);
            sf = seg == 0 ? sfN : seg == 1 ? sfS : sfO;
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            if ((g & 3) == 0)
                store_scale_dword(sf + sf_offset(r, j, kseg), byte | (b1 << 8) | (b2 << 16) | (b3 << 24));
        }
        __syncthreads();  // the row and part[] are rewritten by the next iteration
        store_code_image(image, bytesS, KO, oS, oO, r);   // (read here, rewritten only after the next iteration's first barrier)
    }
}

hipError_t launch_rmsnorm_quantize(const void *src, const void *weight, float eps, int rows, int K, const int16_t *idx, int KN,
                                   int KS, int KO, bool integer_round, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN,
                                   uint8_t *sfS, uint8_t *sfO, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const int T = K / 32;
    const int threads = (T + 63) / 64 * 64;
    int P = 64;
    while (P < T) P <<= 1;
    const bool products = threads <= 256;   // K <= 8192: the 32-bit product row (see the header)
    const size_t lds = (size_t)K * (products ? 4 : 2) + (size_t)(P > threads ? P : threads) * 4 + (size_t)KS / 4 * 3 + KO;
    // 256 / 512 / 1024 threads: K <= 8192 / 16384 / 32768 (the 1024-thread variant is limited to 128 registers and spills a few)
    auto kern = products ? (integer_round ? rmsnorm_quantize_products_kernel<true> : rmsnorm_quantize_products_kernel<false>)
              : threads <= 512 ? (integer_round ? rmsnorm_quantize_kernel<true, 512> : rmsnorm_quantize_kernel<false, 512>)
                               : (integer_round ? rmsnorm_quantize_kernel<true, 1024> : rmsnorm_quantize_kernel<false, 1024>);
    // K > ~21000: row + partial sums + the image of the fp6 / fp8 codes pass the default 64 KiB limit of dynamic LDS (K = 32768: 100 KiB)
    static DynamicLdsOnce attr[6];
    if (lds > 48 * 1024) {
        const int which = (threads <= 256 ? 0 : threads <= 512 ? 1 : 2) * 2 + (integer_round ? 1 : 0);
        if (hipError_t e = attr[which].ensure(reinterpret_cast<const void *>(kern), 104 * 1024); e != hipSuccess) return e;
    }
    const int per_cu = OccupancyCache::get(integer_round ? 4 : 5, reinterpret_cast<const void *>(kern), threads, lds);
    const int cus = device_cus();
    int blocks = cus * per_cu;
    blocks = rows < blocks ? rows : blocks;
    MM_LAUNCH(kern, dim3(blocks), dim3(threads), lds, stream, (const uint16_t *)src, (const uint16_t *)weight, eps, rows,
                       K, idx, KN, KS, KO, oN, oS, oO, sfN, sfS, sfO);
    return hipGetLastError();
}

}  // namespace mm
