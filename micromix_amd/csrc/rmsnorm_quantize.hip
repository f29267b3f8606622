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
// fp32 divide and square root are the correctly rounded ones (the reference's rsqrt() approximation is not reproducible).
//
// Layout as reorder_quantize.hip: one workgroup strides over rows, thread t owns reordered group t, its 32 indices AND its
// 32 gathered norm weights stay in registers for the whole launch, the row is staged in LDS by coalesced 16-byte loads.
#include "mx_group_convert.h"
#include "mx_kernels.h"

namespace mm {

template <int EL, bool INT_ROUND>
__device__ __forceinline__ uint32_t rms_quantize_group(const uint8_t *__restrict__ row, const uint32_t (&ix)[16],
                                                       const uint32_t (&wg)[16], float rvar, uint8_t *__restrict__ out) {
    typedef float f2 __attribute__((ext_vector_type(2)));   // two-wide fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32)
    uint32_t v[16];
    us2 amax2 = {0, 0};
    const f2 rvar2 = {rvar, rvar};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f2 x = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + (ix[i] & 0xFFFFu))),
                      bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + (ix[i] >> 16)))};
        const f2 w = {bf16_bits_to_f32(wg[i] & 0xFFFFu), bf16_bits_to_f32(wg[i] >> 16)};
        // (x * w) is exact in fp32 (two 8-bit significands); one rounding in the multiply by rvar, one to bf16
        const f2 r = (x * w) * rvar2;
        v[i] = pack_bf16x2(r[0], r[1]);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
    }
    const uint32_t amax = amax2[0] > amax2[1] ? amax2[0] : amax2[1];
    const int e = scale_exponent<EL>(amax << 16);
    if (e == -127) {
        quantize_group_tiny<EL, INT_ROUND>(v, out);
        return 0u;
    }
    if constexpr (INT_ROUND) {
        // round(v * 2^-e) half away from zero = trunc(t + copysign(0.5, t)).  The reference's clamp to +-FMAX cannot bite here:
        // e is the smallest exponent with FMAX * 2^e >= amax, so |t| <= FMAX, an integer.
        const float rs = __uint_as_float((uint32_t)(127 - e) << 23);  // 2^-e: v * 2^-e is exact
        const f2 rs2 = {rs, rs};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f2 t = f2{bf16_bits_to_f32(v[i] & 0xFFFFu), bf16_bits_to_f32(v[i] >> 16)} * rs2;
            t = t + f2{__builtin_copysignf(0.5f, t[0]), __builtin_copysignf(0.5f, t[1])};
            v[i] = pack_bf16x2(__builtin_truncf(t[0]), __builtin_truncf(t[1]));
        }
        convert_group<EL, true>(v, 1.0f, out);
    } else {
        convert_group<EL, true>(v, __uint_as_float((uint32_t)(127 + e) << 23), out);
    }
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

    // the norm weights of this thread's 32 columns: the weight vector is staged in LDS (coalesced) and gathered from there
    // with the same byte offsets as the row (32 scattered 2-byte global loads per thread cost more than the two rows a
    // workgroup typically processes)
    uint32_t ix[16], wg[16];
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            reinterpret_cast<uint4 *>(smem)[i * T + g] = reinterpret_cast<const uint4 *>(weight)[i * T + g];
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
                const uint32_t b0 = (w[k] & 0xFFFFu) << 1, b1 = (w[k] >> 16) << 1;   // byte offsets into a staged [K] bf16 vector
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
                reinterpret_cast<uint4 *>(smem)[i * T + g] = stage[i];
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
            if (g == 0) part[0] = __fdiv_rn(1.0f, __fsqrt_rn(__fdiv_rn(s, (float)K) + eps));
        }
        __syncthreads();
        const float rvar = part[0];
        if (active) {
            const uint8_t *row = smem;
            uint32_t byte;
            uint8_t *sf;
            if (seg == 0) {
                byte = rms_quantize_group<EL_FP4, INT_ROUND>(row, ix, wg, rvar, oN + (size_t)r * (KN >> 1) + j * 16);
                sf = sfN;
            } else if (seg == 1) {
                byte = rms_quantize_group<EL_FP6, INT_ROUND>(row, ix, wg, rvar, oS + (size_t)r * (KS / 4 * 3) + j * 24);
                sf = sfS;
            } else {
                byte = rms_quantize_group<EL_FP8, INT_ROUND>(row, ix, wg, rvar, oO + (size_t)r * KO + j * 32);
                sf = sfO;
            }
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            if ((g & 3) == 0)
                *reinterpret_cast<uint32_t *>(sf + sf_offset(r, j, kseg)) = byte | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        __syncthreads();  // the row and part[] are rewritten by the next iteration
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
    const size_t lds = (size_t)K * 2 + (size_t)(P > threads ? P : threads) * 4;
    // 256 / 512 / 1024 threads: K <= 8192 / 16384 / 32768 (the 1024-thread variant is limited to 128 registers and spills a few)
    auto kern = threads <= 256 ? (integer_round ? rmsnorm_quantize_kernel<true, 256> : rmsnorm_quantize_kernel<false, 256>)
              : threads <= 512 ? (integer_round ? rmsnorm_quantize_kernel<true, 512> : rmsnorm_quantize_kernel<false, 512>)
                               : (integer_round ? rmsnorm_quantize_kernel<true, 1024> : rmsnorm_quantize_kernel<false, 1024>);
    // K = 32768: 64 KiB of row + 4 KiB of partial sums, above the default 64 KiB limit of dynamic LDS
    static DynamicLdsOnce attr[6];
    if (lds > 48 * 1024) {
        const int which = (threads <= 256 ? 0 : threads <= 512 ? 1 : 2) * 2 + (integer_round ? 1 : 0);
        if (hipError_t e = attr[which].ensure(reinterpret_cast<const void *>(kern), 72 * 1024); e != hipSuccess) return e;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), threads, lds) != hipSuccess ||
        per_cu < 1)
        per_cu = 1;
    const int cus = device_cus();
    int blocks = cus * per_cu;
    blocks = rows < blocks ? rows : blocks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, (const uint16_t *)src, (const uint16_t *)weight, eps, rows,
                       K, idx, KN, KS, KO, oN, oS, oO, sfN, sfS, sfO);
    return hipGetLastError();
}

}  // namespace mm
