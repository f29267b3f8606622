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
#include <stdlib.h>

#include "mx_group_convert.h"
#include "mx_rms_convert.h"
#include "mx_kernels.h"

namespace mm {

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

// GPT = groups per thread.  8192 < K <= 16384: one (K / 32 <= 512 threads).  16384 < K <= 32768: TWO -- thread g plays the reference's group
// threads t = g and g + 512 one after the other, with their indices and norm weights in 2 x 32 registers.  (Rounds 1-5 ran that range with
// 1024 threads, i.e. 128 registers per lane, which this kernel does not fit in: it spilled 12-13 registers, reloaded inside the row
// loop.  Round 6: no kernel of the library may use scratch -- 512 threads have 256 registers each.)
template <bool INT_ROUND, int GPT>
__global__ void __launch_bounds__(512)
rmsnorm_quantize_kernel(const uint16_t *__restrict__ src, const uint16_t *__restrict__ weight, float eps, int rows, int K,
                        const int16_t *__restrict__ idx, int KN, int KS, int KO, uint8_t *__restrict__ oN,
                        uint8_t *__restrict__ oS, uint8_t *__restrict__ oO, uint8_t *__restrict__ sfN,
                        uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [K bf16 row][P floats of partial sums]
    const int T = K >> 5;  // the reference's group threads = stager threads: group thread t stages chunks t, T + t, 2T + t, 3T + t
    const int g = threadIdx.x, NTH = (int)blockDim.x;
    // this thread's group threads: t_u = g + u * NTH (GPT = 1: NTH >= T, so t_0 = g alone)
    bool active[GPT];
#pragma unroll
    for (int u = 0; u < GPT; ++u) active[u] = g + u * NTH < T;
    float *part = reinterpret_cast<float *>(smem + (size_t)K * 2);
    int P = 64;
    while (P < T) P <<= 1;
    const int bytesS = KS / 4 * 3;
    const int pslots = P > GPT * NTH ? P : GPT * NTH;
    uint8_t *image = smem + (size_t)K * 2 + (size_t)pslots * 4;   // [row][partial sums][image of the S | O codes]

    // the norm weights of this thread's 32 columns: the weight vector is staged in LDS (coalesced) and gathered from there
    // with the same byte offsets as the row (32 scattered 2-byte global loads per thread cost more than the two rows a
    // workgroup typically processes)
    uint32_t ix[GPT][16], wg[GPT][16];
#pragma unroll
    for (int u = 0; u < GPT; ++u)
        if (active[u]) {
            const int t = g + u * NTH;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                reinterpret_cast<uint4 *>(smem)[swizzle_chunk(i * T + t)] = reinterpret_cast<const uint4 *>(weight)[i * T + t];
        }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GPT; ++u)
        if (active[u]) {
            const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)(g + u * NTH) * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 q = p[i];
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t both = swizzle_offsets((w[k] << 1) & 0xFFFEFFFEu);       // byte offsets into a staged (chunk-swizzled) [K] bf16 vector
                    const uint32_t b0 = both & 0xFFFFu, b1 = both >> 16;
                    ix[u][4 * i + k] = b0 | (b1 << 16);
                    wg[u][4 * i + k] = (uint32_t)*reinterpret_cast<const uint16_t *>(smem + b0) |
                                       ((uint32_t)*reinterpret_cast<const uint16_t *>(smem + b1) << 16);
                }
            }
        }
    __syncthreads();   // the row staging below reuses the same LDS bytes
    const int gN = KN >> 5, gS = KS >> 5;
    int seg[GPT], j[GPT], kseg[GPT];
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        const int t = g + u * NTH;
        if (t < gN) { seg[u] = 0; j[u] = t; kseg[u] = KN; }
        else if (t < gN + gS) { seg[u] = 1; j[u] = t - gN; kseg[u] = KS; }
        else { seg[u] = 2; j[u] = t - gN - gS; kseg[u] = KO; }
    }

    // GPT = 1: the next row's four chunks are loaded into registers before the current row is processed (HBM latency hides under
    // the gather), and stored to LDS -- squares summed on the way -- after the barrier that ends the current row.
    // GPT = 2 fetches each row when it needs it (2 x 4 chunks: the registers go to the second group's indices and weights)
    constexpr bool PREFETCH = GPT == 1;
    uint4 stage[GPT][4];
    auto fetch = [&](int r) {
        if (r < rows) {
            const uint4 *grow = reinterpret_cast<const uint4 *>(src + (size_t)r * K);
#pragma unroll
            for (int u = 0; u < GPT; ++u)
                if (active[u]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) stage[u][i] = grow[i * T + g + u * NTH];
                }
        }
    };
    // (the partial sum of group thread t_u, in the reference's order: its four chunks one after the other)
    auto stage_and_sum = [&](int u) -> float {
        float sum = 0.0f;
        if (active[u]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                reinterpret_cast<uint4 *>(smem)[swizzle_chunk(i * T + g + u * NTH)] = stage[u][i];
                const uint32_t w[4] = {stage[u][i].x, stage[u][i].y, stage[u][i].z, stage[u][i].w};
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
    if constexpr (PREFETCH) fetch(blockIdx.x);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        if constexpr (!PREFETCH) fetch(r);
        float sum[GPT];
#pragma unroll
        for (int u = 0; u < GPT; ++u) sum[u] = stage_and_sum(u);
        if constexpr (PREFETCH) fetch(r + gridDim.x);
        // part[t] for t < T, zeros up to P (P <= 2 * GPT * NTH: group threads past T contribute the zero padding)
#pragma unroll
        for (int u = 0; u < GPT; ++u) part[g + u * NTH] = sum[u];
        if (g + GPT * NTH < P) part[g + GPT * NTH] = 0.0f;
        __syncthreads();
        for (int stride = P >> 1; stride >= 64; stride >>= 1) {
            if (g < stride) part[g] += part[g + stride];          // (stride <= 512 <= NTH whenever GPT = 2; GPT = 1: P <= 2 NTH)
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
#pragma unroll
        for (int u = 0; u < GPT; ++u)
            if (active[u]) {
                const uint8_t *row = smem;
                const uint32_t byte = rms_group<INT_ROUND, false>(row, ix[u], wg[u], rvar, seg[u], j[u], r, KN, oN, image, bytesS);
                uint8_t *sf = seg[u] == 0 ? sfN : seg[u] == 1 ? sfS : sfO;
                const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
                const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
                const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
                if ((g & 3) == 0)
                    store_scale_dword(sf + sf_offset(r, j[u], kseg[u]), byte | (b1 << 8) | (b2 << 16) | (b3 << 24));
            }
        __syncthreads();  // the row and part[] are rewritten by the next iteration
        // (the image is read here and rewritten only after the next iteration's first barrier; at most K bytes = 2 GPT chunks per thread)
        store_code_image<2 * GPT>(image, bytesS, KO, oS, oO, r);
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

// ---------------------------------------------------------------------------------------------------------
// K <= 8192, round 5: the rows arrive by LDS-DMA in a ring of R slots (rmsnorm_quantize_ring_kernel).
// Why: the products kernel above keeps 167 VGPRs (six workgroups of two waves per CU) and one 8 KB row in flight per workgroup:
// ~48 KB of loads in flight per CU, where reorder_quantize_kernel -- the same bytes, 0.72 of 8 TB/s -- has 13-16 workgroups' rows in
// flight.  Here a row costs no registers while it travels: `buffer_load_dwordx4 ... lds` (four 1 KB pieces per wave and row, the
// chunk swizzle of mx_group_convert.h applied on the SOURCE address) straight into slot r % R, R - 1 rows ahead, behind ONE counted
// s_waitcnt vmcnt.  The row stays bf16 (8 KB per slot at K = 4096); the norm weights of the thread's 32 columns are gathered once
// into 16 registers (as in rmsnorm_quantize_kernel); the sum of squares reads the thread's four chunks back from LDS in the
// reference's order; every wave walks the halving tree by itself.  Two barriers per row:
//   B1(r): row r has landed for every wave AND every wave is done with row r-1 (its slot is refilled right behind B1: row r+R-1);
//          the [S | O] code image of row r-1 is complete: it leaves here (store_code_image);
//   B2(r): the T partial sums are in LDS.
// vmcnt counts loads, stores and LDS-DMA together in issue order.  Per iteration a wave issues: [wait] B1, the image stores of row
// r-1 (0-2), the four DMAs of row r+R-1, B2, then the stores of row r: the scale dword (always: lane 0 of every wave owns one) and the
// fp4 codes (when the wave has a lane in the fp4 segment).  So younger than row r's own DMAs are, per ring step, at least 4 DMAs + 1
// store: vmcnt(5 (R - 2)) never lets one of row r's pieces stay in flight, and lets the rows behind it (all of them but at most one
// piece per step, when a wave issues both stores) travel on.  When the next row does not exist nothing was issued for it: vmcnt(0).
// Every wave issues exactly four DMAs per row (rows of K <= 8192 are at most 16 pieces on at most 4 waves); a piece past the row's
// end lies past the descriptor's range, fetches nothing and lands in a dump area of its own.
// ---------------------------------------------------------------------------------------------------------
typedef int rms_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rms_rsrc_t rms_make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    rms_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// one buffer_load_dwordx4 ... lds: 64 lanes x 16 B -> LDS bytes [lds, lds + 1024) in lane order (M0 = wave-uniform base; the compiler
// owns M0, so it is saved and restored; s_nop 4 / 0: SALU -> VMEM and M0 -> LDS-DMA wait states)
__device__ __forceinline__ void rms_dma16(const rms_rsrc_t &rsrc, int voff, int soff, unsigned lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds), "s"(soff)
                 : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void rms_wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

template <bool INT_ROUND, int R>
__global__ void __launch_bounds__(256)
rmsnorm_quantize_ring_kernel(const uint16_t *__restrict__ src, const uint16_t *__restrict__ weight, float eps, int rows, int K,
                             const int16_t *__restrict__ idx, int KN, int KS, int KO, uint8_t *__restrict__ oN,
                             uint8_t *__restrict__ oS, uint8_t *__restrict__ oO, uint8_t *__restrict__ sfN,
                             uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO, int slot_bytes) {
    static_assert(R >= 3 && 5 * (R - 2) < 64, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [R slots][1 KB dump][P floats of partial sums][image of the S | O codes]
    const int T = K >> 5;
    // The counted wait below, vmcnt(5 (R - 2)), relies on EVERY wave issuing at least one store per row (the scale dword of its lane 0),
    // i.e. on lane 0 of every wave owning a group: blockDim.x == roundup64(K / 32), as launch_rmsnorm_quantize launches it.  A launch
    // with surplus waves (DMA pieces but no stores) would let a row be read before it has landed: refuse it loudly.
    if ((int)blockDim.x != ((T + 63) & ~63)) __builtin_trap();
    const int g = threadIdx.x;
    const bool active = g < T;
    const int wave = __builtin_amdgcn_readfirstlane(g >> 6), lane = g & 63, nw = (int)blockDim.x >> 6;
    uint8_t *const dump = smem + (size_t)R * slot_bytes;
    float *part = reinterpret_cast<float *>(dump + 1024);
    int P = 64;
    while (P < T) P <<= 1;
    const int bytesS = KS / 4 * 3;
    uint8_t *image = reinterpret_cast<uint8_t *>(part) + (size_t)(P > (int)blockDim.x ? P : (int)blockDim.x) * 4;
    const unsigned lds0 = (unsigned)(unsigned long long)smem;
    const int pieces = (K + 511) >> 9;                       // 1 KB pieces of a row
    // this lane's source offset inside a piece: LDS chunk c = 64 i + p holds source chunk swizzle_chunk(c) = 64 i + swizzle_chunk(p)
    const int voff = swizzle_chunk(lane) * 16;

    // rows r0, r0 + stride, ...: DMA of the n-th of them into slot n % R
    const int stride = (int)gridDim.x;
    auto issue_row = [&](int r, int slot) {
        const rms_rsrc_t rs = rms_make_rsrc(src + (size_t)r * K, (unsigned)K * 2u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave + i * nw;                    // piece index (wave-uniform); past the row: nothing fetched, lands in the dump
            const bool real = pc < pieces;
            rms_dma16(rs, real ? voff : 0x7FFFFF00, __builtin_amdgcn_readfirstlane(real ? pc * 1024 : 0),
                      __builtin_amdgcn_readfirstlane(real ? lds0 + (unsigned)slot * (unsigned)slot_bytes + (unsigned)pc * 1024u
                                                          : lds0 + (unsigned)R * (unsigned)slot_bytes));
        }
    };

    // ---- prologue: the thread's 32 column offsets and norm weights.  The weight vector is staged in the LAST slot (free until the
    // first refill, behind B1 of the first row) by plain loads; the first R - 1 rows are requested behind those loads ----
    uint32_t ix[16], wg[16];
    uint8_t *const wslot = smem + (size_t)(R - 1) * slot_bytes;
    // The index and weight loads are issued from inline asm (the compiler would wait for them where it first touches them -- before the
    // DMAs are even requested), the first R - 1 rows follow at once, and ONE counted wait certifies the plain loads: they are older
    // than the `dmas` LDS-DMA instructions behind them.
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u wq[4] = {}, iq[4] = {};
    {
        [[maybe_unused]] const uint4 *ip = reinterpret_cast<const uint4 *>(idx + (size_t)(active ? g : 0) * 32), *wp = reinterpret_cast<const uint4 *>(weight);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(iq[i]) : "v"(ip + i) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wq[i]) : "v"(wp + (active ? i * T + g : 0)) : "memory");
#endif
        }
    }
    int r = (int)blockIdx.x;
    int dmas = 0;
#pragma unroll
    for (int n = 0; n < R - 1; ++n)
        if (r + n * stride < rows) { issue_row(r + n * stride, n); dmas += 4; }
    if (dmas >= 12) rms_wait_vmcnt<12>();
    else if (dmas == 8) rms_wait_vmcnt<8>();
    else if (dmas == 4) rms_wait_vmcnt<4>();
    else rms_wait_vmcnt<0>();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(iq[i]), "+v"(wq[i]));     // valid from here on
#endif
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<v4u *>(wslot)[swizzle_chunk(i * T + g)] = wq[i];
    }
    __syncthreads();        // the staged weights are visible
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w[4] = {iq[i][0], iq[i][1], iq[i][2], iq[i][3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t both = swizzle_offsets((w[k] << 1) & 0xFFFEFFFEu);       // byte offsets into a staged (chunk-swizzled) [K] bf16 vector
                const uint32_t b0 = both & 0xFFFFu, b1 = both >> 16;
                ix[4 * i + k] = both;
                wg[4 * i + k] = (uint32_t)*reinterpret_cast<const uint16_t *>(wslot + b0) |
                                ((uint32_t)*reinterpret_cast<const uint16_t *>(wslot + b1) << 16);
            }
        }
    }
    const int gN = KN >> 5, gS = KS >> 5;
    int seg, j, kseg;
    if (g < gN) { seg = 0; j = g; kseg = KN; }
    else if (g < gN + gS) { seg = 1; j = g - gN; kseg = KS; }
    else { seg = 2; j = g - gN - gS; kseg = KO; }

    int slot = 0, prev = -1;
    for (int it = 0; r < rows; r += stride, ++it) {
        // row r has landed (this wave's pieces; the barrier extends it to every wave's)
        // (the first R - 1 iterations: the rows requested by the prologue have no stores between them -- only the DMAs are counted)
        if (r + (R - 2) * stride >= rows) rms_wait_vmcnt<0>();
        else if (it < R - 1) rms_wait_vmcnt<4 * (R - 2)>();
        else rms_wait_vmcnt<5 * (R - 2)>();
        __syncthreads();                                                                  // B1
        if (prev >= 0) store_code_image(image, bytesS, KO, oS, oO, prev);                 // the previous row's fp6 / fp8 codes leave
        {
            const int rn = r + (R - 1) * stride;
            int sn = slot + (R - 1);
            sn = sn >= R ? sn - R : sn;
            if (rn < rows) issue_row(rn, sn);                                             // into the slot the previous row has left
        }
        const uint8_t *row = smem + (size_t)slot * slot_bytes;
        float sum = 0.0f;
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 q = reinterpret_cast<const uint4 *>(row)[swizzle_chunk(i * T + g)];
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf16_bits_to_f32(w[k] & 0xFFFFu), b = bf16_bits_to_f32(w[k] >> 16);
                    sum = __builtin_fmaf(a, a, sum);  // a*a is exact: the fused and the unfused forms round identically
                    sum = __builtin_fmaf(b, b, sum);
                }
            }
        }
        part[g] = sum;              // threads T.. contribute the zero padding (P < 2 * blockDim.x)
        if (g + (int)blockDim.x < P) part[g + blockDim.x] = 0.0f;
        __syncthreads();                                                                  // B2
        const float rvar = rms_tree_rvar(part, P, lane, K, eps);
        if (active) {
            const uint32_t byte = rms_group<INT_ROUND, false>(row, ix, wg, rvar, seg, j, r, KN, oN, image, bytesS);
            uint8_t *sf = seg == 0 ? sfN : seg == 1 ? sfS : sfO;
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            if ((g & 3) == 0)
                store_scale_dword(sf + sf_offset(r, j, kseg), byte | (b1 << 8) | (b2 << 16) | (b3 << 24));
        }
        prev = r;
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    __syncthreads();
    if (prev >= 0) store_code_image(image, bytesS, KO, oS, oO, prev);
}

hipError_t launch_rmsnorm_quantize(const void *src, const void *weight, float eps, int rows, int K, const int16_t *idx, int KN,
                                   int KS, int KO, bool integer_round, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN,
                                   uint8_t *sfS, uint8_t *sfO, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const int T = K / 32;
    const int groups_per_thread = T > 512 ? 2 : 1;     // K > 16384: rmsnorm_quantize_kernel<*, 2>, 512 threads
    // (the ring kernel's counted vmcnt needs exactly roundup64(K / 32) threads -- every wave's lane 0 owns a group; it traps otherwise)
    const int threads = ((T + groups_per_thread - 1) / groups_per_thread + 63) / 64 * 64;
    int P = 64;
    while (P < T) P <<= 1;
    const bool products = threads <= 256;   // K <= 8192: the 32-bit product row (see the header)
    // K <= 8192, round 5: the LDS-DMA row ring (MICROMIX_RMS_RING=0: the products kernel; =3 / 4: ring depth)
    // Measured (round 5, tools/time_rmsnorm.py, K = 4096, us, products / ring of 3): 256 rows 7.5 / 5.8 -- the ring kernel's prologue
    // requests its first rows before anything else -- but 4096 rows 11.6 / 14.7: with several rows per workgroup the launch is bound by
    // the per-row instruction stream (the integer rounding alone is 1.5 us of it), not by bytes in flight, and the ring kernel reads its
    // row back from LDS for the sum of squares.  So: the ring while a workgroup sees at most ~2 rows (rows <= 2 x CUs), the products
    // kernel beyond.  MICROMIX_RMS_RING=0 never, =3 / =4 always with that depth.
    static const int ring_pin = getenv("MICROMIX_RMS_RING") ? atoi(getenv("MICROMIX_RMS_RING")) : -1;
    const int ring = ring_pin >= 0 ? ring_pin : (rows <= 2 * device_cus() ? 3 : 0);
    if (products && ring >= 3) {
        const int slot = ((K * 2 + 1023) / 1024) * 1024;
        const int R = ring >= 4 ? 4 : 3;
        const size_t lds_r = (size_t)R * slot + 1024 + (size_t)(P > threads ? P : threads) * 4 + (size_t)KS / 4 * 3 + KO;
        auto kr = R == 4 ? (integer_round ? rmsnorm_quantize_ring_kernel<true, 4> : rmsnorm_quantize_ring_kernel<false, 4>)
                         : (integer_round ? rmsnorm_quantize_ring_kernel<true, 3> : rmsnorm_quantize_ring_kernel<false, 3>);
        static DynamicLdsOnce rattr[4];
        if (lds_r > 48 * 1024)
            if (hipError_t e = rattr[(R == 4 ? 2 : 0) + (integer_round ? 1 : 0)].ensure(reinterpret_cast<const void *>(kr), 104 * 1024); e != hipSuccess) return e;
        const int per_cu = OccupancyCache::get(integer_round ? 6 : 7, reinterpret_cast<const void *>(kr), threads, lds_r);
        int blocks = device_cus() * per_cu;
        blocks = rows < blocks ? rows : blocks;
        MM_LAUNCH(kr, dim3(blocks), dim3(threads), lds_r, stream, (const uint16_t *)src, (const uint16_t *)weight, eps, rows, K, idx, KN, KS, KO,
                  oN, oS, oO, sfN, sfS, sfO, slot);
        return hipGetLastError();
    }
    const int pslots = P > groups_per_thread * threads ? P : groups_per_thread * threads;
    const size_t lds = (size_t)K * (products ? 4 : 2) + (size_t)pslots * 4 + (size_t)KS / 4 * 3 + KO;
    // 256 / 512 / 512 threads: K <= 8192 / 16384 / 32768 (beyond 16384 a thread plays two of the reference's group threads)
    auto kern = products ? (integer_round ? rmsnorm_quantize_products_kernel<true> : rmsnorm_quantize_products_kernel<false>)
              : groups_per_thread == 1 ? (integer_round ? rmsnorm_quantize_kernel<true, 1> : rmsnorm_quantize_kernel<false, 1>)
                               : (integer_round ? rmsnorm_quantize_kernel<true, 2> : rmsnorm_quantize_kernel<false, 2>);
    // K > ~21000: row + partial sums + the image of the fp6 / fp8 codes pass the default 64 KiB limit of dynamic LDS (K = 32768: 100 KiB)
    static DynamicLdsOnce attr[6];
    if (lds > 48 * 1024) {
        const int which = (products ? 0 : groups_per_thread == 1 ? 1 : 2) * 2 + (integer_round ? 1 : 0);
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
