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
    int early;              // quantize_rows_early applies: one batch (stage_rows >= M), at most one half group per thread (dq::early_fits)
    // RMSNorm in front of the quantization (mm_rmsnorm_qlinear_decode; mode 0, K <= 8192): v = bf16((x * w) * rvar) with the reference's
    // summation order and integer rounding (rmsnorm.cu:95-312) -- the bytes of mm_rmsnorm_quantize
    const uint16_t *norm_w; // [K] bf16, or null: no norm
    float eps;
    int int_round;          // the reference's round() before the conversion (0: MM_RMS_NO_INTEGER_ROUND)
};
constexpr int EARLY_RL = 2;      // 16-byte chunks of the rows per thread (M K / 8 <= NT * EARLY_RL follows from early_fits)
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

// ---------------------------------------------------------------------------------------------------------
// Two lanes per group (round 6).  Lanes (2p, 2p + 1) of a wave share reordered group p: lane half h = lane & 1 owns pairs 8h .. 8h + 7
// of the group's sixteen (elements 16h .. 16h + 15), eight index and eight value registers instead of sixteen each.
// Why: with one lane per group the phase is a serial walk of 32 (with the norm: 64) two-byte LDS gathers and their arithmetic on K / 32
// of the workgroup's threads -- 5.9 us of a 17 us launch (tools/stream_clock.py) -- and its sixteen + sixteen registers put the norm's
// 32-feature streaming kernel ONE register over the two-workgroups-per-CU budget (gate/up N = 14336 at M = 1: 15.7 us; a bounding
// experiment with half the gather per thread: 9.7).  The absmax of the halves meets by one DPP exchange; fp8 / fp4 codes leave from both
// lanes (16 / 8 bytes each); the fp6 converter wants all 32 values in one lane: lane 0 fetches its partner's eight registers by DPP.
// Same values, same roundings, same bytes (tests/test_decode_gpu.py, test_rmsnorm_decode_gpu.py: bit-identity with the stand-alone
// quantizers and the oracle).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_partner(uint32_t x) {      // the other lane of the pair (quad_perm [1, 0, 3, 2])
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
}
// this lane's half of the group's index registers: byte offsets into the staged row, two per register
__device__ __forceinline__ void load_ix_half(const int16_t *idx, int g, int h, uint32_t (&ix)[8]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)g * 32 + 16 * h);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 q = p[i];
        ix[4 * i] = (q.x << 1) & 0xFFFEFFFEu;
        ix[4 * i + 1] = (q.y << 1) & 0xFFFEFFFEu;
        ix[4 * i + 2] = (q.z << 1) & 0xFFFEFFFEu;
        ix[4 * i + 3] = (q.w << 1) & 0xFFFEFFFEu;
    }
}
// (one lane per group: all sixteen)
__device__ __forceinline__ void load_ix(const int16_t *idx, int g, uint32_t (&ix)[16]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)g * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 q = p[i];
        ix[4 * i] = (q.x << 1) & 0xFFFEFFFEu;
        ix[4 * i + 1] = (q.y << 1) & 0xFFFEFFFEu;
        ix[4 * i + 2] = (q.z << 1) & 0xFFFEFFFEu;
        ix[4 * i + 3] = (q.w << 1) & 0xFFFEFFFEu;
    }
}
// gather_group / rms_quantize_group's gather on a half: v = the lane's 16 values (two bf16 per register), returns the half's absmax bits
__device__ __forceinline__ uint32_t gather_half(const uint8_t *__restrict__ row, const uint32_t (&ix)[8], uint32_t (&v)[8]) {
    us2 amax2 = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = *reinterpret_cast<const uint16_t *>(row + (ix[i] & 0xFFFFu));
        const uint32_t hi = *reinterpret_cast<const uint16_t *>(row + (ix[i] >> 16));
        v[i] = lo | (hi << 16);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
    }
    return amax2[0] > amax2[1] ? amax2[0] : amax2[1];
}
__device__ __forceinline__ uint32_t rms_gather_half(const uint8_t *__restrict__ row, const uint8_t *__restrict__ wrow, const uint32_t (&ix)[8],
                                                    float rvar, uint32_t (&v)[8]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    us2 amax2 = {0, 0};
    const f2 rvar2 = {rvar, rvar};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = ix[i] & 0xFFFFu, hi = ix[i] >> 16;
        const f2 x = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + lo)), bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(row + hi))};
        const f2 w = {bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(wrow + lo)), bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(wrow + hi))};
        const f2 r = (x * w) * rvar2;          // (x * w) is exact in fp32; one rounding in the multiply by rvar, one to bf16
        v[i] = pack_bf16x2(r[0], r[1]);
        const uint32_t mag = v[i] & 0x7FFF7FFFu;
        us2 m;
        __builtin_memcpy(&m, &mag, 4);
        amax2 = __builtin_elementwise_max(amax2, m);
        if ((i & 1) == 1) { MM_DQ_SCHED_BARRIER(); }     // (as rms_quantize_group: keep the gather's live values few)
    }
    return amax2[0] > amax2[1] ? amax2[0] : amax2[1];
}
// the conversion of a half (convert_group's instructions on eight registers); `out` = the GROUP's first byte
template <int EL>
__device__ __forceinline__ void convert_half(const uint32_t (&v)[8], float scale, int h, uint8_t *__restrict__ out) {
    if constexpr (EL == EL_FP8) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf2 a, b;
            __builtin_memcpy(&a, &v[2 * i], 4);
            __builtin_memcpy(&b, &v[2 * i + 1], 4);
            s2 r = {0, 0};
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, a, scale, false);
            r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, b, scale, true);
            __builtin_memcpy(&w[i], &r, 4);
        }
        *reinterpret_cast<uint4 *>(out + 16 * h) = make_uint4(w[0], w[1], w[2], w[3]);
    } else if constexpr (EL == EL_FP4) {
        uint32_t w[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
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
        *reinterpret_cast<uint2 *>(out + 8 * h) = make_uint2(w[0], w[1]);
    } else {
        // all 32 values in one lane: {own half, partner's half} in lane 0 (lane 1 assembles them the other way round and stores nothing)
        uint32_t full[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t other = lane_partner(v[i]);
            full[i] = h ? other : v[i];
            full[8 + i] = h ? v[i] : other;
        }
        bf32 x;
        __builtin_memcpy(&x, full, 64);
        const u6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(x, scale);
        if (h == 0) {
            uint2 *o = reinterpret_cast<uint2 *>(out);
            o[0] = make_uint2(r[0], r[1]);
            o[1] = make_uint2(r[2], r[3]);
            o[2] = make_uint2(r[4], r[5]);
        }
    }
}
// rms_scale<true>'s integer rounding on a half (same arithmetic per pair)
__device__ __forceinline__ void int_round_half(uint32_t (&v)[8], int e) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int em = e > 126 ? 126 : e;
    const float rs = __uint_as_float(((uint32_t)(127 - em) << 23) | 0x2000u);  // 2^-e * (1 + 2^-10), see rms_scale
    const f2 rs2 = {rs, rs};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f2 t = f2{bf16_bits_to_f32(v[i] & 0xFFFFu), bf16_bits_to_f32(v[i] >> 16)} * rs2;
        v[i] = pack_bf16x2(__builtin_rintf(t[0]), __builtin_rintf(t[1]));
    }
}
// quantize_group / rms_quantize_group for the lane pair: returns the group's scale byte (in both lanes)
template <int EL>
__device__ __forceinline__ uint32_t quantize_half(const uint8_t *__restrict__ row, const uint32_t (&ix)[8], int h, uint8_t *__restrict__ out) {
    uint32_t v[8];
    uint32_t amax = gather_half(row, ix, v);
    const uint32_t other = lane_partner(amax);
    amax = amax > other ? amax : other;
    const int e = scale_exponent<EL>(amax << 16);
    convert_half<EL>(v, __uint_as_float((uint32_t)(127 + e) << 23), h, out);
    return (uint32_t)(e + 127);
}
template <int EL>
__device__ __forceinline__ uint32_t rms_quantize_half(const uint8_t *__restrict__ row, const uint8_t *__restrict__ wrow, const uint32_t (&ix)[8], float rvar,
                                                      bool int_round, int h, uint8_t *__restrict__ out) {
    uint32_t v[8];
    uint32_t amax = rms_gather_half(row, wrow, ix, rvar, v);
    const uint32_t other = lane_partner(amax);
    amax = amax > other ? amax : other;
    const int e = scale_exponent_rt(amax << 16, ElemTraits<EL>::FMAX_EXP, ElemTraits<EL>::FMAX_MANT);
    float scale = __uint_as_float((uint32_t)(127 + e) << 23);
    if (int_round) {
        int_round_half(v, e);
        scale = 1.0f;
    }
    convert_half<EL>(v, scale, h, out);
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
// LPG: lanes per group.  2: lane pairs (above).  1: one lane per group.  0: chosen per launch -- pairs when every slot of a batch of staged
// rows fits ONE pass of the workgroup, else one lane per group: a pass is a chain of LDS round trips (gather, convert, store) that half the
// work per lane does not halve, so pairs lose wherever they double the passes (q/o at M = 4: 6.4 -> 7.3 us, M = 8: 8.5 -> 9.9) and win
// where they do not (M = 1, 2).  Both paths in one kernel cost the larger one's registers: kernels at one workgroup per CU anyway take 0,
// the norm's 32-feature streaming kernel 2 (its register count decides between one and two workgroups per CU).
template <int NT, bool RMS = false, int LPG = 0, class Hook = NoHook>
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

    // ---- phase 1: quantize the M activation rows into LDS (reorder.cu:94-269 per group) ----
    // stage_rows rows are staged at a time and their slots are spread over all NT threads.  Pairs: slot u = 2 (rr Gt + g) + h -- NT is
    // even, so a thread keeps its half h and lanes (2p, 2p + 1) always sit in the same group; else slot u = rr Gt + g
    static_assert(NT % 64 == 0, "lane pairs");
    static_assert(LPG >= 0 && LPG <= 2, "lanes per group");
    const int h = threadIdx.x & 1;
    const int batch = a.M < a.stage_rows ? a.M : a.stage_rows;
    const bool pairs = LPG == 2 || (LPG == 0 && 2 * batch * Gt <= NT);
    constexpr int NIX = LPG == 2 ? 8 : 16;
    uint32_t ix[NIX];
    uint32_t (&ixh)[8] = reinterpret_cast<uint32_t (&)[8]>(ix);
    for (int r0 = 0; r0 < a.M; r0 += a.stage_rows) {
        const int nr = (a.M - r0) < a.stage_rows ? (a.M - r0) : a.stage_rows;
        const int slots = pairs ? nr * Gt * 2 : nr * Gt;
        // the indices of this thread's first slot are requested BEFORE the rows are staged, so the two global round trips overlap
        const int t0 = threadIdx.x;
        if (t0 < slots) {
            if (pairs) load_ix_half(a.idx, (t0 >> 1) % Gt, h, ixh);
            else if constexpr (LPG != 2) load_ix(a.idx, t0 % Gt, ix);
        }
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
        if (pairs) {
            for (int t = t0; t < slots; t += NT) {
                const int rr = (t >> 1) / Gt, g = (t >> 1) - rr * Gt, r = r0 + rr;
                const uint8_t *row = stage + (size_t)rr * Kt * 2;
                if (t != t0) load_ix_half(a.idx, g, h, ixh);      // (LPG 2 only: LPG 0 takes pairs when there is one pass)
                uint32_t byte;
                if constexpr (rms) {
                    const float rv = rvar[r];
                    if (g < gN) byte = rms_quantize_half<EL_FP4>(row, wvec, ixh, rv, a.int_round != 0, h, opN + r * pN + g * 16);
                    else if (g < gN + gS) byte = rms_quantize_half<EL_FP6>(row, wvec, ixh, rv, a.int_round != 0, h, opS + r * pS + (g - gN) * 24);
                    else byte = rms_quantize_half<EL_FP8>(row, wvec, ixh, rv, a.int_round != 0, h, opO + r * pO + (g - gN - gS) * 32);
                } else {
                    if (g < gN) byte = quantize_half<EL_FP4>(row, ixh, h, opN + r * pN + g * 16);
                    else if (g < gN + gS) byte = quantize_half<EL_FP6>(row, ixh, h, opS + r * pS + (g - gN) * 24);
                    else byte = quantize_half<EL_FP8>(row, ixh, h, opO + r * pO + (g - gN - gS) * 32);
                }
                if (h == 0) scales[r * Gt + g] = (uint8_t)byte;
            }
        } else if constexpr (LPG != 2) {
            for (int t = t0; t < slots; t += NT) {
                const int rr = t / Gt, g = t - rr * Gt, r = r0 + rr;
                const uint8_t *row = stage + (size_t)rr * Kt * 2;
                if (t != t0) load_ix(a.idx, g, ix);
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
        }
        __syncthreads();
    }

    LdsMap m;
    m.opN = opN; m.opS = opS; m.opO = opO; m.scales = scales;
    m.pN = pN; m.pS = pS; m.pO = pO; m.Gt = Gt; m.gN = gN; m.gS = gS;
    return m;
}

// The same phase with ALL of its global loads issued through inline asm before the caller's own untracked loads (`request()`: the
// first slabs of mx_gemm_stream.hip's DMA ring and its scale image; it returns the number of vector-memory instructions it issued -- at
// most 63, wave-uniform, 0: none), and ONE counted wait: the rows and indices are older than those, so vmcnt(that number) certifies them while the weights stay in flight -- they are
// requested ~1 us earlier than from the hook of quantize_rows_to_lds, which has to wait for the staged rows first.
// Preconditions (QuantIn::early, set by the launcher: early_fits): stage_rows >= M and at most NPASS (row, group, half) slots per thread,
// 2 M K / 32 <= NPASS NT -- which bounds the rows at EARLY_RL and the norm's weight vector at EARLY_WL 16-byte chunks per thread and pass.
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
// NPASS: slots per thread.  1 on the eight-wave kernels, whose register count decides how many workgroups a CU holds; 2 on the four-wave
// 64-feature kernels (two workgroups per CU either way), where it keeps M = 2 at K = 4096 on this path (fused gate + up 13.8 us; through the
// staged path 16.3).
// s_waitcnt vmcnt(n) for a wave-uniform n known at run time only (0 .. 63: the immediate is picked by six scalar compares)
template <int LO, int HI>
__device__ __forceinline__ void wait_vmcnt_range(int n) {
    if constexpr (LO == HI) {
        MM_DQ_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO) : "memory");)
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (n <= MID) wait_vmcnt_range<LO, MID>(n);
        else wait_vmcnt_range<MID + 1, HI>(n);
    }
}
__device__ __forceinline__ void wait_vmcnt(int n) { wait_vmcnt_range<0, 63>(n < 0 ? 0 : (n > 63 ? 63 : n)); }
__host__ __device__ inline bool early_fits(int M, int Kt, int NT, int npass) { return 2 * (size_t)M * (size_t)(Kt >> 5) <= (size_t)NT * npass; }
template <int NT, bool RMS, int NPASS, class Request, class Landed>
__device__ __forceinline__ LdsMap quantize_rows_early(const QuantIn &a, uint8_t *smem, Request request, Landed landed) {
    const int Kt = a.K[0] + a.K[1] + a.K[2], Gt = Kt >> 5;
    const int gN = a.K[0] >> 5, gS = a.K[1] >> 5;
    const int pN = a.K[0] >> 1, pS = (a.K[1] >> 2) * 3, pO = a.K[2];
    uint8_t *stage = smem;
    uint8_t *opN = stage + (size_t)a.stage_rows * Kt * 2, *opS = opN + a.M * pN, *opO = opS + a.M * pS;
    uint8_t *scales = opO + a.M * pO;
    // slot u = 2 (rr Gt + g) + h: half h of group g of row rr (two lanes per group, see above); this thread's slots: t + p NT
    const int t = threadIdx.x, slots = a.M * Gt * 2, chunks = a.M * (Kt >> 3), h = t & 1;
    int rr[NPASS], g[NPASS];
    bool live[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int u = t + p * NT;
        live[p] = u < slots;
        rr[p] = live[p] ? (u >> 1) / Gt : 0;
        g[p] = live[p] ? (u >> 1) - rr[p] * Gt : 0;
    }
    const uint4 *grow = reinterpret_cast<const uint4 *>(a.X);
    // with the norm: [norm weights | partial sums | rvar] behind the 16-byte rounded operands (rms_bytes); EARLY_WL NPASS more loads per thread
    constexpr bool rms = RMS;
    constexpr int RL = EARLY_RL * NPASS, WL = EARLY_WL * NPASS;
    const int P = rms_pow2(Gt);
    uint8_t *wvec = opN + ((operand_bytes(a.M, a.K) + 15) & ~(size_t)15);
    float *part = reinterpret_cast<float *>(wvec + (size_t)Kt * 2), *rvar = part + (size_t)a.M * P;
    dq_v4u iq[NPASS][2], rq[RL], wq[WL] = {};
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) iq[p][i] = gload16(reinterpret_cast<const uint4 *>(a.idx + (size_t)g[p] * 32 + 16 * h) + i);
    // Which chunks of the rows a thread loads.  Without the norm: chunk t + k NT (side by side; M K / 8 = 2 slots <= RL NT).  With it: the
    // lane pair of group thread (rr, g) loads ITS four chunks of the reference's summation order, i Gt + g of row rr (rmsnorm.cu:143-160)
    // -- lane h the chunks i = 2h, 2h + 1; every chunk exactly once -- so that the partial sum of squares comes straight from these
    // registers: no second pass over the staged row, one barrier fewer.
    static_assert(EARLY_RL == 2, "a lane's two chunks of its group thread's four");
    auto row_chunk = [&](int k) {
        const int p = k >> 1;
        return rms ? (live[p] ? rr[p] * (Kt >> 3) + (2 * h + (k & 1)) * Gt + g[p] : -1) : (t + k * NT < chunks ? t + k * NT : -1);
    };
#pragma unroll
    for (int k = 0; k < RL; ++k) {
        const int c = row_chunk(k);
        rq[k] = gload16(grow + (c >= 0 ? c : chunks - 1));       // (nothing to load: the last chunk again, not stored)
    }
    if constexpr (rms) {                                                                                   // (see the wait below)
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int c = t + k * NT;
            wq[k] = gload16(reinterpret_cast<const uint4 *>(a.norm_w) + (c < (Kt >> 3) ? c : 0));
        }
    }
    wait_vmcnt(request());      // (everything older than the caller's requests: this phase's loads)
    // `landed()`: requests of the caller's that may NOT travel in front of any wave's rows -- a CU's memory pipe returns its waves' loads in
    // issue order, so a slow request (scale bytes from HBM) issued before a later wave's rows holds those up, and the phase's barrier waits
    // for every wave's.  Here this wave's loads are home and every other wave's have long been issued.
    landed();
#pragma unroll
    for (int p = 0; p < NPASS; ++p) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(iq[p][0]), "+v"(iq[p][1]));) }
#pragma unroll
    for (int k = 0; k < RL; ++k) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(rq[k]));) }
    if constexpr (rms) {
#pragma unroll
        for (int k = 0; k < WL; ++k) { MM_DQ_DEVICE_ONLY(asm volatile("" : "+v"(wq[k]));) }
    }
#pragma unroll
    for (int k = 0; k < RL; ++k) {
        const int c = row_chunk(k);
        if (c >= 0) reinterpret_cast<dq_v4u *>(stage)[c] = rq[k];
    }
    if constexpr (rms) {
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int c = t + k * NT;
            if (c < (Kt >> 3)) reinterpret_cast<dq_v4u *>(wvec)[c] = wq[k];
        }
        // the reference's partial sums: rms_thread_sum's chain over the group thread's chunks i = 0 .. 3, eight elements each, one after
        // the other.  Lane 0 of the pair holds chunks 0, 1, lane 1 chunks 2, 3: the chain runs over the lane's own sixteen elements from
        // zero (right in lane 0), crosses to the partner, and runs over them again from there (right in lane 1, which stores it)
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
            if (live[p]) {
                auto chain = [&](float sum) {
#pragma unroll
                    for (int i = 0; i < EARLY_RL; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t two = rq[EARLY_RL * p + i][k];
                            const float x0 = bf16_bits_to_f32(two & 0xFFFFu), x1 = bf16_bits_to_f32(two >> 16);
                            sum = __builtin_fmaf(x0, x0, sum);
                            sum = __builtin_fmaf(x1, x1, sum);
                        }
                    return sum;
                };
                const float first = chain(0.0f);
                const float sum = chain(__uint_as_float(lane_partner(__float_as_uint(first))));
                if (h == 1) part[rr[p] * P + g[p]] = sum;
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
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
        if (live[p]) {
            uint32_t ix[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ix[4 * i] = (iq[p][i][0] << 1) & 0xFFFEFFFEu;      // byte offsets into the staged row, two per register
                ix[4 * i + 1] = (iq[p][i][1] << 1) & 0xFFFEFFFEu;
                ix[4 * i + 2] = (iq[p][i][2] << 1) & 0xFFFEFFFEu;
                ix[4 * i + 3] = (iq[p][i][3] << 1) & 0xFFFEFFFEu;
            }
            const int r = rr[p], gg = g[p];
            const uint8_t *row = stage + (size_t)r * Kt * 2;
            uint32_t byte;
            if constexpr (rms) {
                const float rv = rvar[r];
                if (gg < gN) byte = rms_quantize_half<EL_FP4>(row, wvec, ix, rv, a.int_round != 0, h, opN + r * pN + gg * 16);
                else if (gg < gN + gS) byte = rms_quantize_half<EL_FP6>(row, wvec, ix, rv, a.int_round != 0, h, opS + r * pS + (gg - gN) * 24);
                else byte = rms_quantize_half<EL_FP8>(row, wvec, ix, rv, a.int_round != 0, h, opO + r * pO + (gg - gN - gS) * 32);
            } else {
                if (gg < gN) byte = quantize_half<EL_FP4>(row, ix, h, opN + r * pN + gg * 16);
                else if (gg < gN + gS) byte = quantize_half<EL_FP6>(row, ix, h, opS + r * pS + (gg - gN) * 24);
                else byte = quantize_half<EL_FP8>(row, ix, h, opO + r * pO + (gg - gN - gS) * 32);
            }
            if (h == 0) scales[r * Gt + gg] = (uint8_t)byte;
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
// of the weight ring; it returns the number of vector-memory instructions it issued) and ONE counted wait -- as quantize_rows_early: the
// weights are requested ~1 us earlier than from the hook above, which has to wait for the values first (round 6).
// Precondition (QuantIn::early with mode 1, set by the launcher): ONE pass, M * K / 32 <= NT.
// (Round 6, measured and dropped: FOUR lanes per group -- a lane owns one 16-byte chunk of gate and of up, the absmax meets by DPP, every
// load instruction covers contiguous runs of 256 bytes.  down_proj at M = 1 the same 9.4 us, M = 2 11.0 -> 12.4, M = 4 13.7 -> 17.7: with
// K = 14336 one lane per group already keeps 448 of 512 threads busy, so a thread's chain of silu does not get shorter, there are only
// more passes; lane-contiguous loads alone were worth 0.35 us.)
template <int NT, class Request, class Landed>
__device__ __forceinline__ LdsMap activate_rows_early(const QuantIn &a, uint8_t *smem, Request request, Landed landed) {
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
    wait_vmcnt(request());      // (everything older than the caller's requests: this phase's loads)
    // `landed()`: requests of the caller's that may NOT travel in front of any wave's rows -- a CU's memory pipe returns its waves' loads in
    // issue order, so a slow request (scale bytes from HBM) issued before a later wave's rows holds those up, and the phase's barrier waits
    // for every wave's.  Here this wave's loads are home and every other wave's have long been issued.
    landed();
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
