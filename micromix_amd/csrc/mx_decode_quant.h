// Phase 1 of the decode-sized QLinearLayer.forward kernels (qlinear_decode.hip, and the QUANT mode of mx_gemm_stream.hip): every
// workgroup quantizes the M <= 8 activation rows into LDS by itself -- reorder + per-32 absmax + UE8M0 scale + MXFP4/6/8 codes,
// reorder.cu:94-269 per group, through the shared quantize_group of mx_group_convert.h: the bytes of reorder_quantize_x.
#pragma once
#include "mx_group_convert.h"
#include "mx_direct_convert.h"
#include "mx_rms_convert.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DQ_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define MM_DQ_SCHED_BARRIER() do { } while (0)
#endif

namespace mm {
namespace dq {

struct QuantIn {
    const uint16_t *X;      // [M, K] bf16
    const int16_t *idx;     // [K]
    int K[3];
    int M;
    int stage_rows;         // activation rows staged in LDS at a time (launcher: as many as fit)
    int mode;               // 0: X = bf16 rows, reordered by idx (reorder_quantize_x); 1: X = [M, 2 K] bf16 with 128 gate | 128 up columns
                            // alternating (mm_gate_up_activate's scratch layout), quantized as activate_quantize_x: silu(gate) * up in
                            // natural column order (activate_rows_to_lds)
    int early;              // quantize_rows_early applies: one batch (stage_rows >= M), at most one group per thread, at most EARLY_RL row loads per thread
    // RMSNorm in front of the quantization (mm_rmsnorm_qlinear_decode; mode 0, K <= 8192): v = bf16((x * w) * rvar) with the reference's
    // summation order and integer rounding (rmsnorm.cu:95-312) -- the bytes of mm_rmsnorm_quantize
    const uint16_t *norm_w; // [K] bf16, or null: no norm
    float eps;
    int int_round;          // the reference's round() before the conversion (0: MM_RMS_NO_INTEGER_ROUND)
};
constexpr int EARLY_RL = 4;
constexpr int EARLY_WL = 2;      // with the norm: 16-byte chunks of the weight vector per thread (K / 8 <= NT * EARLY_WL)

// LDS map: [staged bf16 rows | opN | opS | opO | scale bytes]; row r of a segment at op + r * pitch, its scale bytes at
// scales + r * Gt + (first group of the segment)
struct LdsMap {
    uint8_t *opN, *opS, *opO, *scales;
    int pN, pS, pO, Gt, gN, gS;
};
__host__ __device__ inline size_t operand_bytes(int M, const int K[3]) {   // quantized rows + their scale bytes
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    return (size_t)M * (K[0] / 2 + K[1] / 4 * 3 + K[2] + Kt / 32);
}
// LDS behind the (16-byte rounded) operands when the norm runs inside the launch: [K bf16 norm weights | M x P partial sums | M rvar]
__host__ __device__ inline int rms_pow2(int T) { int P = 64; while (P < T) P <<= 1; return P; }
__host__ __device__ inline size_t rms_bytes(int M, const int K[3]) {
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    return Kt * 2 + (size_t)M * rms_pow2((int)(Kt >> 5)) * 4 + 64;
}
constexpr int RMS_MAX_K = 8192;      // the wave-local halving tree covers P <= 256 partial sums

// one group with the norm: gather x and w (same byte offsets into the staged row and the staged weight vector), v = bf16((x w) rvar),
// block exponent, the reference's integer rounding, conversion into LDS; returns the scale byte (rmsnorm_quantize.hip: rms_group)
template <int EL>
__device__ __forceinline__ uint32_t rms_quantize_group(const uint8_t *__restrict__ row, const uint8_t *__restrict__ wrow, const uint32_t (&ix)[16],
                                                       float rvar, bool int_round, uint8_t *__restrict__ out) {
    // (rms_gather of mx_rms_convert.h with the weights read where they are used: 16 registers fewer in a phase that shares its
    // kernel with asm-owned accumulators)
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint32_t v[16];
    us2 amax2 = {0, 0};
    const f2 rvar2 = {rvar, rvar};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t lo = ix[i] & 0xFFFFu, hi = ix[i] >> 16;
        const f2 x = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + lo)), bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + hi))};
        const f2 w = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(wrow + lo)), bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(wrow + hi))};
        const f2 r = (x * w) * rvar2;          // (x * w) is exact in fp32; one rounding in the multiply by rvar, one to bf16
        v[i] = pack_bf16x2(r[0], r[1]);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
        // (the scheduler would otherwise hoist all 64 two-byte LDS reads in front of the arithmetic: 64 live values, and the kernel's
        // register count decides how many workgroups a CU holds)
        if ((i & 1) == 1) { MM_DQ_SCHED_BARRIER(); }
    }
    const uint32_t amax = amax2[0] > amax2[1] ? amax2[0] : amax2[1];
    int e;
    const float scale = int_round ? rms_scale<true>(v, amax, ElemTraits<EL>::FMAX_EXP, ElemTraits<EL>::FMAX_MANT, e)
                                  : rms_scale<false>(v, amax, ElemTraits<EL>::FMAX_EXP, ElemTraits<EL>::FMAX_MANT, e);
    convert_group<EL>(v, scale, out);
    return (uint32_t)(e + 127);
}

// phase 1: quantize the M activation rows into LDS (reorder.cu:94-269 per group, shared quantize_group)
// `staged()` runs once, right after the first batch of rows has been staged (every global load of that batch has landed): a caller
// with loads of its own that the compiler does not track (mx_gemm_stream.hip's DMA ring) issues them there, so that they fly during
// the arithmetic and the compiler's own wait counts never have to cover them.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// RMS: the norm runs in front of the quantization (a.norm_w etc.).  A template parameter, not a run-time test: with both paths in one
// kernel the register allocation is the larger path's, and the streaming kernels' quantizing variants lost a resident workgroup per CU
// to it (87 -> 121 VGPRs; gate/up at M = 1 9.3 -> 13.4 us, round 5).
template <int NT, bool RMS = false, class Hook = NoHook>
__device__ __forceinline__ LdsMap quantize_rows_to_lds(const QuantIn &a, uint8_t *smem, Hook staged = Hook()) {
    const int Kt = a.K[0] + a.K[1] + a.K[2], Gt = Kt >> 5;
    const int gN = a.K[0] >> 5, gS = a.K[1] >> 5;
    const int pN = a.K[0] >> 1, pS = (a.K[1] >> 2) * 3, pO = a.K[2];      // packed bytes per row and segment
    uint8_t *stage = smem;
    uint8_t *opN = stage + (size_t)a.stage_rows * Kt * 2, *opS = opN + a.M * pN, *opO = opS + a.M * pS;
    uint8_t *scales = opO + a.M * pO;
    // with the norm: [norm weights | partial sums | rvar] behind the 16-byte rounded operands (rms_bytes)
    constexpr bool rms = RMS;
    const int P = rms_pow2(Gt);
    uint8_t *wvec = opN + ((operand_bytes(a.M, a.K) + 15) & ~(size_t)15);
    float *part = reinterpret_cast<float *>(wvec + (size_t)Kt * 2), *rvar = part + (size_t)a.M * P;

    // ---- phase 1: quantize the M activation rows into LDS (reorder.cu:94-269 per group, shared quantize_group) ----
    // stage_rows rows are staged at a time and their (row, group) pairs are spread over all 512 threads
    auto load_ix = [&](int g, uint32_t (&ix)[16]) {
        const uint4 *p = reinterpret_cast<const uint4 *>(a.idx + (size_t)g * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = p[i];
            ix[4 * i] = (q.x << 1) & 0xFFFEFFFEu;      // byte offsets into the staged row, two per register
            ix[4 * i + 1] = (q.y << 1) & 0xFFFEFFFEu;
            ix[4 * i + 2] = (q.z << 1) & 0xFFFEFFFEu;
            ix[4 * i + 3] = (q.w << 1) & 0xFFFEFFFEu;
        }
    };
    for (int r0 = 0; r0 < a.M; r0 += a.stage_rows) {
        const int nr = (a.M - r0) < a.stage_rows ? (a.M - r0) : a.stage_rows;
        // the indices of this thread's first (row, group) pair are requested BEFORE the rows are staged, so the two global
        // round trips overlap
        uint32_t ix[16];
        const int t0 = threadIdx.x;
        if (t0 < nr * Gt) load_ix(t0 % Gt, ix);
        const uint4 *grow = reinterpret_cast<const uint4 *>(a.X + (size_t)r0 * Kt);
        for (int c = threadIdx.x; c < nr * (Kt >> 3); c += NT) reinterpret_cast<uint4 *>(stage)[c] = grow[c];
        if constexpr (rms) if (r0 == 0)
            for (int c = threadIdx.x; c < (Kt >> 3); c += NT) reinterpret_cast<uint4 *>(wvec)[c] = reinterpret_cast<const uint4 *>(a.norm_w)[c];
        __syncthreads();
        if (r0 == 0) staged();
        if constexpr (rms) {
            // the reference's sum of squares per row: Gt group threads' partial sums (zero padded to P), then the halving tree, one wave per row
            for (int u = threadIdx.x; u < nr * P; u += NT) {
                const int rr = u / P, t = u - rr * P;
                const uint4 *row4 = reinterpret_cast<const uint4 *>(stage + (size_t)rr * Kt * 2);
                part[u] = t < Gt ? rms_thread_sum(t, Gt, [&](int q) { return row4[q]; }) : 0.0f;
            }
            __syncthreads();
            for (int rr = threadIdx.x >> 6; rr < nr; rr += NT / 64) {
                const float rv = rms_tree_rvar(part + rr * P, P, threadIdx.x & 63, Kt, a.eps);
                if ((threadIdx.x & 63) == 0) rvar[r0 + rr] = rv;
            }
            __syncthreads();
        }
        for (int t = t0; t < nr * Gt; t += NT) {
            const int rr = t / Gt, g = t - rr * Gt, r = r0 + rr;
            const uint8_t *row = stage + (size_t)rr * Kt * 2;
            if (t != t0) load_ix(g, ix);
            uint32_t byte;
            if constexpr (rms) {
                const float rv = rvar[r];
                if (g < gN) byte = rms_quantize_group<EL_FP4>(row, wvec, ix, rv, a.int_round != 0, opN + r * pN + g * 16);
                else if (g < gN + gS) byte = rms_quantize_group<EL_FP6>(row, wvec, ix, rv, a.int_round != 0, opS + r * pS + (g - gN) * 24);
                else byte = rms_quantize_group<EL_FP8>(row, wvec, ix, rv, a.int_round != 0, opO + r * pO + (g - gN - gS) * 32);
            } else {
                if (g < gN) byte = quantize_group<EL_FP4>(row, ix, opN + r * pN + g * 16);
                else if (g < gN + gS) byte = quantize_group<EL_FP6>(row, ix, opS + r * pS + (g - gN) * 24);
                else byte = quantize_group<EL_FP8>(row, ix, opO + r * pO + (g - gN - gS) * 32);
            }
            scales[r * Gt + g] = (uint8_t)byte;
        }
        __syncthreads();
    }

    LdsMap m;
    m.opN = opN; m.opS = opS; m.opO = opO; m.scales = scales;
    m.pN = pN; m.pS = pS; m.pO = pO; m.Gt = Gt; m.gN = gN; m.gS = gS;
    return m;
}

// The same phase with ALL of its global loads issued through inline asm before the caller's own untracked loads (`request()`: the
// first slabs of mx_gemm_stream.hip's DMA ring, AFTER_LOADS vector-memory instructions when it returns true), and ONE counted wait:
// the rows and indices are older than the ring, so vmcnt(AFTER_LOADS) certifies them while the weights stay in flight -- they are
// requested ~1 us earlier than from the hook of quantize_rows_to_lds, which has to wait for the staged rows first.
// Preconditions (QuantIn::early, set by the launcher): stage_rows >= M, M * K / 32 <= NT, M * K / 8 <= NT * EARLY_RL.
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DQ_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_DQ_DEVICE_ONLY(...)
#endif
typedef unsigned dq_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dq_v4u gload16(const void *p) {
    dq_v4u d = {0u, 0u, 0u, 0u};
    MM_DQ_DEVICE_ONLY(asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p) : "memory");)
    return d;
}
template <int NT, int AFTER_LOADS, bool RMS, class Request>
__device__ __forceinline__ LdsMap quantize_rows_early(const QuantIn &a, uint8_t *smem, Request request) {
    const int Kt = a.K[0] + a.K[1] + a.K[2], Gt = Kt >> 5;
    const int gN = a.K[0] >> 5, gS = a.K[1] >> 5;
    const int pN = a.K[0] >> 1, pS = (a.K[1] >> 2) * 3, pO = a.K[2];
    uint8_t *stage = smem;
    uint8_t *opN = stage + (size_t)a.stage_rows * Kt * 2, *opS = opN + a.M * pN, *opO = opS + a.M * pS;
    uint8_t *scales = opO + a.M * pO;
    const int t = threadIdx.x, groups = a.M * Gt, chunks = a.M * (Kt >> 3);
    const int rr = t < groups ? t / Gt : 0, g = t < groups ? t - rr * Gt : 0;
    const uint4 *ip = reinterpret_cast<const uint4 *>(a.idx + (size_t)g * 32);
    const uint4 *grow = reinterpret_cast<const uint4 *>(a.X);
    // with the norm: [norm weights | partial sums | rvar] behind the 16-byte rounded operands (rms_bytes); one more load per thread
    // (the launcher admits the early path with the norm only when the weight vector is at most EARLY_WL chunks per thread)
    constexpr bool rms = RMS;
    const int P = rms_pow2(Gt);
    uint8_t *wvec = opN + ((operand_bytes(a.M, a.K) + 15) & ~(size_t)15);
    float *part = reinterpret_cast<float *>(wvec + (size_t)Kt * 2), *rvar = part + (size_t)a.M * P;
    dq_v4u iq[4], rq[EARLY_RL], wq[EARLY_WL] = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) iq[i] = gload16(ip + i);
    // Which chunks of the rows a thread loads.  Without the norm: chunk t + k NT (side by side).  With it (round 6): group thread (rr, g)
    // loads ITS OWN four chunks of the reference's summation order, i Gt + g of row rr (rmsnorm.cu:143-160) -- every chunk exactly once,
    // M Gt <= NT group threads x 4 -- so that its partial sum of squares comes straight from these registers: no second pass over the
    // staged row, one barrier fewer (tools/stream_clock.py: the phase with the norm was 5.8 us of a 17 us launch).
    static_assert(!RMS || EARLY_RL == 4, "a group thread's four chunks");
    auto row_chunk = [&](int k) { return rms ? (t < groups ? rr * (Kt >> 3) + k * Gt + g : -1) : (t + k * NT < chunks ? t + k * NT : -1); };
#pragma unroll
    for (int k = 0; k < EARLY_RL; ++k) {
        const int c = row_chunk(k);
        rq[k] = gload16(grow + (c >= 0 ? c : chunks - 1));       // (nothing to load: the last chunk again, not stored)
    }
    if constexpr (rms) {                                                                                   // (see the wait below)
#pragma unroll
        for (int k = 0; k < EARLY_WL; ++k) {
            const int c = t + k * NT;
            wq[k] = gload16(reinterpret_cast<const uint4 *>(a.norm_w) + (c < (Kt >> 3) ? c : 0));
        }
    }
    const bool requested = request();
    if (requested) { MM_DQ_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER_LOADS) : "memory");) }
    else { MM_DQ_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");) }
#pragma unroll
    for (int i = 0; i < 4; ++i) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(iq[i]));) }
#pragma unroll
    for (int k = 0; k < EARLY_RL; ++k) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(rq[k]));) }
    if constexpr (rms) {
#pragma unroll
        for (int k = 0; k < EARLY_WL; ++k) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(wq[k]));) }
    }
#pragma unroll
    for (int k = 0; k < EARLY_RL; ++k) {
        const int c = row_chunk(k);
        if (c >= 0) reinterpret_cast<dq_v4u *>(stage)[c] = rq[k];
    }
    if constexpr (rms) {
#pragma unroll
        for (int k = 0; k < EARLY_WL; ++k) {
            const int c = t + k * NT;
            if (c < (Kt >> 3)) reinterpret_cast<dq_v4u *>(wvec)[c] = wq[k];
        }
        // the reference's partial sums (rms_thread_sum's arithmetic on the group thread's own chunks, i = 0 .. 3, eight elements each, one
        // after the other) and the zero padding up to P
        if (t < groups) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x0 = bf16_bits_to_f32(rq[i][k] & 0xFFFFu), x1 = bf16_bits_to_f32(rq[i][k] >> 16);
                    sum = __builtin_fmaf(x0, x0, sum);
                    sum = __builtin_fmaf(x1, x1, sum);
                }
            part[rr * P + g] = sum;
        }
        for (int u = t; u < a.M * (P - Gt); u += NT) {
            const int r2 = u / (P - Gt);
            part[r2 * P + Gt + (u - r2 * (P - Gt))] = 0.0f;
        }
    }
    __syncthreads();
    if constexpr (rms) {      // the halving tree by one wave per row (as quantize_rows_to_lds)
        for (int r2 = t >> 6; r2 < a.M; r2 += NT / 64) {
            const float rv = rms_tree_rvar(part + r2 * P, P, t & 63, Kt, a.eps);
            if ((t & 63) == 0) rvar[r2] = rv;
        }
        __syncthreads();
    }
    if (t < groups) {
        uint32_t ix[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ix[4 * i] = (iq[i][0] << 1) & 0xFFFEFFFEu;      // byte offsets into the staged row, two per register
            ix[4 * i + 1] = (iq[i][1] << 1) & 0xFFFEFFFEu;
            ix[4 * i + 2] = (iq[i][2] << 1) & 0xFFFEFFFEu;
            ix[4 * i + 3] = (iq[i][3] << 1) & 0xFFFEFFFEu;
        }
        const uint8_t *row = stage + (size_t)rr * Kt * 2;
        uint32_t byte;
        if constexpr (rms) {
            const float rv = rvar[rr];
            if (g < gN) byte = rms_quantize_group<EL_FP4>(row, wvec, ix, rv, a.int_round != 0, opN + rr * pN + g * 16);
            else if (g < gN + gS) byte = rms_quantize_group<EL_FP6>(row, wvec, ix, rv, a.int_round != 0, opS + rr * pS + (g - gN) * 24);
            else byte = rms_quantize_group<EL_FP8>(row, wvec, ix, rv, a.int_round != 0, opO + rr * pO + (g - gN - gS) * 32);
        } else {
            if (g < gN) byte = quantize_group<EL_FP4>(row, ix, opN + rr * pN + g * 16);
            else if (g < gN + gS) byte = quantize_group<EL_FP6>(row, ix, opS + rr * pS + (g - gN) * 24);
            else byte = quantize_group<EL_FP8>(row, ix, opO + rr * pO + (g - gN - gS) * 32);
        }
        scales[rr * Gt + g] = (uint8_t)byte;
    }
    __syncthreads();
    LdsMap m;
    m.opN = opN; m.opS = opS; m.opO = opO; m.scales = scales;
    m.pN = pN; m.pS = pS; m.pO = pO; m.Gt = Gt; m.gN = gN; m.gS = gS;
    return m;
}

// The activation quantizer as phase 1 (mm_down_activate_decode): h = silu(gate) * up per 32 consecutive intermediate features, the
// arithmetic of direct_quantize.hip (activate.cu:44-202; shared quantize32 / silu_mul: the same bytes), codes and scale bytes into the LDS
// map above with K = the intermediate size in natural order.  `request()` as in quantize_rows_to_lds: called once, after the first
// batch of loads has landed.
template <int NT, class Hook>
__device__ __forceinline__ LdsMap activate_rows_to_lds(const QuantIn &a, uint8_t *smem, Hook request) {
    const int Kt = a.K[0] + a.K[1] + a.K[2], Gt = Kt >> 5;
    const int gN = a.K[0] >> 5, gS = a.K[1] >> 5;
    const int pN = a.K[0] >> 1, pS = (a.K[1] >> 2) * 3, pO = a.K[2];
    uint8_t *opN = smem, *opS = opN + a.M * pN, *opO = opS + a.M * pS;
    uint8_t *scales = opO + a.M * pO;
    const int groups = a.M * Gt;
    for (int t = threadIdx.x; t - (int)threadIdx.x < groups; t += NT) {      // (every thread walks every pass: request() must run everywhere)
        const bool live = t < groups;
        const int r = live ? t / Gt : 0, g = live ? t - r * Gt : 0;
        // group g of row r: 32 gate values at element r * 2 K + (g / 4) * 256 + (g % 4) * 32, the 32 up values 128 elements further
        const uint4 *pa = reinterpret_cast<const uint4 *>(a.X + (size_t)r * (size_t)(2 * Kt) + (size_t)(g >> 2) * 256u + (size_t)(g & 3) * 32u);
        uint4 xa[4], xb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xa[i] = pa[i];
            xb[i] = pa[16 + i];
        }
        if (t == (int)threadIdx.x) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(xa[i].x), "+v"(xa[i].y), "+v"(xb[i].x), "+v"(xb[i].y));) }   // the loads have landed
            request();
        }
        if (live) {
            float v[32];
#pragma unroll
            for (int i = 0; i < 4; ++i) unpack8(xa[i], v + 8 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float b[8];
                unpack8(xb[i], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * i + e] = silu_mul(v[8 * i + e], b[e]);
            }
            uint32_t byte;
            if (g < gN) byte = quantize32<EL_FP4, false>(v, opN + r * pN + g * 16);
            else if (g < gN + gS) byte = quantize32<EL_FP6, false>(v, opS + r * pS + (g - gN) * 24);
            else byte = quantize32<EL_FP8, false>(v, opO + r * pO + (g - gN - gS) * 32);
            scales[r * Gt + g] = (uint8_t)byte;
        }
    }
    __syncthreads();
    LdsMap m;
    m.opN = opN; m.opS = opS; m.opO = opO; m.scales = scales;
    m.pN = pN; m.pS = pS; m.pO = pO; m.Gt = Gt; m.gN = gN; m.gS = gS;
    return m;
}

// The same with the gate | up values loaded through inline asm IN FRONT of the caller's own untracked loads (`request()`: the first slabs
// of the weight ring, AFTER_LOADS vector-memory instructions when it returns true) and ONE counted wait -- as quantize_rows_early: the
// weights are requested ~1 us earlier than from the hook above, which has to wait for the values first (round 6).
// Precondition (QuantIn::early with mode 1, set by the launcher): ONE pass, M * K / 32 <= NT.
template <int NT, int AFTER_LOADS, class Request>
__device__ __forceinline__ LdsMap activate_rows_early(const QuantIn &a, uint8_t *smem, Request request) {
    const int Kt = a.K[0] + a.K[1] + a.K[2], Gt = Kt >> 5;
    const int gN = a.K[0] >> 5, gS = a.K[1] >> 5;
    const int pN = a.K[0] >> 1, pS = (a.K[1] >> 2) * 3, pO = a.K[2];
    uint8_t *opN = smem, *opS = opN + a.M * pN, *opO = opS + a.M * pS;
    uint8_t *scales = opO + a.M * pO;
    const int groups = a.M * Gt, t = threadIdx.x;
    const bool live = t < groups;
    const int r = live ? t / Gt : 0, g = live ? t - r * Gt : 0;
    const uint4 *pa = reinterpret_cast<const uint4 *>(a.X + (size_t)r * (size_t)(2 * Kt) + (size_t)(g >> 2) * 256u + (size_t)(g & 3) * 32u);
    dq_v4u qa[4], qb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa[i] = gload16(pa + i);
        qb[i] = gload16(pa + 16 + i);
    }
    const bool requested = request();
    if (requested) { MM_DQ_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER_LOADS) : "memory");) }
    else { MM_DQ_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");) }
#pragma unroll
    for (int i = 0; i < 4; ++i) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(qa[i]), "+v"(qb[i]));) }
    if (live) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 4; ++i) unpack8(make_uint4(qa[i][0], qa[i][1], qa[i][2], qa[i][3]), v + 8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float b[8];
            unpack8(make_uint4(qb[i][0], qb[i][1], qb[i][2], qb[i][3]), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 * i + e] = silu_mul(v[8 * i + e], b[e]);
        }
        uint32_t byte;
        if (g < gN) byte = quantize32<EL_FP4, false>(v, opN + r * pN + g * 16);
        else if (g < gN + gS) byte = quantize32<EL_FP6, false>(v, opS + r * pS + (g - gN) * 24);
        else byte = quantize32<EL_FP8, false>(v, opO + r * pO + (g - gN - gS) * 32);
        scales[r * Gt + g] = (uint8_t)byte;
    }
    __syncthreads();
    LdsMap m;
    m.opN = opN; m.opS = opS; m.opO = opO; m.scales = scales;
    m.pN = pN; m.pS = pS; m.pO = pO; m.Gt = Gt; m.gN = gN; m.gS = gS;
    return m;
}

}  // namespace dq
}  // namespace mm
