// Decode-sized QLinearLayer.forward in ONE launch (M <= 8 token rows): reorder + quantize of the activations fused into the
// weight-streaming GEMM of mx_gemm_skinny.hip.
//
// Reference semantics: reorder_quantize_x (mgemm/src/reorder.cu:94-269) followed by matmul (gemm.cu:26-78) (+ bias,
// qLinearLayer.py:58-74) -- the same arithmetic as the two separate kernels, bit for bit (the group quantizer is the shared
// quantize_group of mx_group_convert.h, the GEMM part follows mx_gemm_skinny.hip).
//
// Why: at M = 1 the quantize kernel (4.9 us) and the GEMM (6 us) are both at their launch + one-memory-round-trip floor, so
// the pair costs two floors.  Here every workgroup (32 output features, 8 waves) first quantizes the M activation rows into
// LDS by itself -- the rows are a few KB, re-reading them from L2 in every workgroup is free compared with a second launch --
// and then runs the skinny GEMM with the activation fragments and scales coming from LDS and only the weights from HBM.
#ifndef MM_DECODE_DEPTH
#define MM_DECODE_DEPTH 4   // weight slabs a wave requests at once (1 = one memory round trip per slab)
#endif
#ifndef MM_DECODE_PREFETCH
#define MM_DECODE_PREFETCH 2   // A/B builds: 2 = first slabs requested from the quantization phase's hook, 1 = the same code path but requested behind the phase, 0 = the round-5 path
#endif
#include <type_traits>
#include <utility>

#include "mx_decode_quant.h"
#include "mx_kernels.h"

namespace mm {
namespace decode {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int NT = 512, NW = 8, BN = 32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const uint8_t *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)nb, 0x00020000);
}

template <int EL> struct G { static constexpr int BYTES = EL == EL_FP8 ? 128 : (EL == EL_FP6 ? 96 : 64); };

// weight fragment from global memory (as mx_gemm_skinny.hip)
template <int EL>
__device__ __forceinline__ v8i load_wfrag(__amdgpu_buffer_rsrc_t rsrc, int rowoff, int slab, int h, int kb) {
    const int so = slab * G<EL>::BYTES;
    v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EL == EL_FP8) {
        const v4i lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + (4 * h + kb) * 16, so, 0);
        const v4i hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + (4 * h + 2 + kb) * 16, so, 0);
        r = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else if constexpr (EL == EL_FP4) {
        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + (2 * h + kb) * 16, so, 0);
        r = v8i{v[0], v[1], v[2], v[3], 0, 0, 0, 0};
    } else {
        const int o = rowoff + (2 * h + kb) * 24;
        const v2i a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, so, 0);
        const v2i b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 8, so, 0);
        const v2i c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 16, so, 0);
        r = v8i{a[0], a[1], b[0], b[1], c[0], c[1], 0, 0};
    }
    return r;
}

// activation fragment from the LDS copy of the quantized row (`p` = row base + slab * BYTES); same register layouts
template <int EL>
__device__ __forceinline__ v8i lds_xfrag(const uint8_t *p, int h, int kb) {
    v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EL == EL_FP8) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(p + (4 * h + kb) * 16);
        const uint4 hi = *reinterpret_cast<const uint4 *>(p + (4 * h + 2 + kb) * 16);
        r = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    } else if constexpr (EL == EL_FP4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + (2 * h + kb) * 16);
        r = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    } else {
        const uint2 *q = reinterpret_cast<const uint2 *>(p + (2 * h + kb) * 24);
        const uint2 a = q[0], b = q[1], c = q[2];
        r = v8i{(int)a.x, (int)a.y, (int)b.x, (int)b.y, (int)c.x, (int)c.y, 0, 0};
    }
    return r;
}

struct Args {
    const uint16_t *X;      // [M, K] bf16
    const int16_t *idx;     // [K]
    const uint8_t *W[3];    // packed weight segments
    const uint8_t *SFW[3];
    int K[3];
    int M, N;
    int sfw_row_tiles;
    int round_per_segment;
    const uint16_t *bias;
    uint16_t *D;
    int stage_rows;         // activation rows staged in LDS at a time (launcher: as many as fit)
    const uint16_t *norm_w; // RMSNorm in front of the quantization (null: none), see dq::QuantIn
    float eps;
    int int_round;
};

// this wave's slabs of one segment; xl = LDS base of the segment's quantized rows (pitch xp bytes), sl = LDS base of the
// segment's scale bytes (pitch sp bytes per row, 4 consecutive bytes per slab)
// r0: the wave's slabs of the first r0 rounds (w, w + 8, ...) have been taken care of (Prefetch)
template <int XEL, int WEL>
__device__ __forceinline__ void run_segment(v16f &acc, const uint8_t *xl, int xp, const uint8_t *sl, int sp, const uint8_t *W,
                                            const uint8_t *SFW, int nslab, int M, int N, int n0, int sfw_tiles, int r0 = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kb = lane >> 5;
    const int wrb = nslab * G<WEL>::BYTES;
    int wrows = N - n0;
    wrows = wrows > BN ? BN : wrows;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W + (size_t)n0 * wrb, (unsigned)wrows * (unsigned)wrb);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(SFW, (unsigned)sfw_tiles * (unsigned)nslab * 512u);
    const int n = n0 + li;
    const int sfw_off = (n >> 7) * nslab * 512 + (n & 31) * 16 + ((n >> 5) & 3) * 4;
    const bool valid = li < M;   // MFMA rows = tokens; rows past M are zero
    const uint8_t *xrow = xl + (valid ? li : 0) * xp;
    const uint8_t *srow = sl + (valid ? li : 0) * sp;
    const int sh = 8 * kb;
    // The weights are the only global loads of this loop and depend on nothing: a wave requests DEPTH of its slabs at once
    // (wave-uniform tests for the tail) and then walks them, instead of paying one memory round trip per slab.
    constexpr int DEPTH = MM_DECODE_DEPTH;
    for (int s0 = wave + r0 * NW; s0 < nslab; s0 += DEPTH * NW) {
        v8i wf[DEPTH][2];
        int swr[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int s = s0 + j * NW;
            if (s < nslab) {
                swr[j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, sfw_off, s * 512, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h) wf[j][h] = load_wfrag<WEL>(rw, li * wrb, s, h, kb);
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int s = s0 + j * NW;
            if (s < nslab) {
                const int sw = swr[j] >> sh;
                v8i xf[2];
                int sx = 0;
                const v8i zero = {0, 0, 0, 0, 0, 0, 0, 0};
                xf[0] = xf[1] = zero;
                if (valid) {
                    sx = (int)(*reinterpret_cast<const uint32_t *>(srow + 4 * s) >> sh);
#pragma unroll
                    for (int h = 0; h < 2; ++h) xf[h] = lds_xfrag<XEL>(xrow + s * G<XEL>::BYTES, h, kb);
                }
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[0], wf[j][0], acc, ElemTraits<XEL>::HW, ElemTraits<WEL>::HW, 0, sx, 0, sw);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[1], wf[j][1], acc, ElemTraits<XEL>::HW, ElemTraits<WEL>::HW, 2, sx, 2, sw);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: the first weight slabs are requested BEFORE the activation rows are quantized (fp4 weights, the production mode).
// Until then a workgroup first quantized its rows (rows and indices from L2: ~1 us of latency, ~1 us of arithmetic) and only then asked
// for its weights (another round trip: ~1 us from the Infinity Cache, ~2 us from HBM) -- and q/k/v/o at K = 4096 are ONE round of
// requests per wave.  Now every wave requests its first slabs (weights + the rows' scale dwords, straight into registers, through inline
// asm: hipcc's own wait counts never have to cover them) from the quantization phase's hook, i.e. as soon as the rows are staged, and
// they travel under the phase's arithmetic.
// NOT earlier: requested in front of the rows (dq::quantize_rows_early, one counted wait) the launch got 1.5-2.4 us SLOWER (q/o at M = 1
// 5.6 -> 7.1 us, q | k | v with the norm 9.1 -> 11.5): a CU's memory pipe takes the waves' vector-memory instructions in issue order at
// 30-45 cycles each, so the rows of waves 1-7 queued behind the weight requests of the waves in front of them, and the phase's first
// barrier waits for every wave's rows (measured, round 6).
// The association of the sums is unchanged (slab s of a segment belongs to wave s % 8, a wave adds its slabs in order, the eight partial
// sums meet in wave order): bit-identical to mx_gemm_skinny.hip as before.
//   Per segment the first PRE_R rounds (slabs w and w + 8 of wave w): at K = 4096 that is every slab of every split; a slab that does
//   not exist for this wave requests nothing.
// ---------------------------------------------------------------------------------------------------------
typedef int rsrc4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc4_t make_rsrc4(const uint8_t *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    rsrc4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_QD_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_QD_DEVICE_ONLY(...)
#endif
// (s_nop 4: SALU / readfirstlane results may not be read by a VMEM instruction for five wait states)
// `d` is read-write ("+v"): the destination IS the register that holds the caller's zero when the request is skipped (a wave-uniform
// branch around the statement), so no copy of a pending register is ever needed where the two paths meet
__device__ __forceinline__ void asm_load16(v4i &d, const rsrc4_t &rs, int voff, int soff) {
    MM_QD_DEVICE_ONLY(asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");)
}
__device__ __forceinline__ void asm_load4(int &d, const rsrc4_t &rs, int voff, int soff) {
    MM_QD_DEVICE_ONLY(asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "+v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");)
}
constexpr int PRE_R = 2;                // rounds of a segment (slabs w, w + 8 of wave w) that travel under the quantization phase
template <int FEAT> struct PreSlab;     // the registers one request fills
template <> struct PreSlab<16> { v4i w[1]; int sw; };     // lane (row l & 15, K block h = l >> 4): 16 bytes of the row + its scale dword
template <> struct PreSlab<32> { v4i w[2]; int sw; };     // lane (row l & 31, kb = l >> 5): K blocks kb and 2 + kb

// One segment's first PRE_R rounds.  fp4 weights: the weight rows of every segment are 64 bytes per 128-deep slab.  The segment is a
// compile-time property of the OBJECT (one per segment in the kernels), so that requests and MFMAs are straight-line code per
// segment -- a first version with run-time (slot -> segment) dispatch was ~1100 instructions of selects and register copies per wave
// and cost the launch 1.3 us (tools/time_decode_ab.py, round 6).
template <int FEAT>
struct SegPrefetch {
    PreSlab<FEAT> q[PRE_R];
    __device__ __forceinline__ void request(const uint8_t *W, const uint8_t *SFW, int nslab, int N, int n0, int sfw_tiles) {
        constexpr int LI = FEAT - 1, SHIFT = FEAT == 16 ? 4 : 5;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int li = lane & LI, hb = lane >> SHIFT;        // K block of the lane inside a 64-deep (FEAT 32) / 128-deep (FEAT 16) step
#pragma unroll
        for (int j = 0; j < PRE_R; ++j) {
            q[j].sw = 0;
            q[j].w[0] = v4i{0, 0, 0, 0};
            if constexpr (FEAT == 32) q[j].w[1] = v4i{0, 0, 0, 0};
        }
        if (wave >= nslab) return;                            // (wave-uniform: also the segments that do not exist)
        int wrows = N - n0;
        wrows = wrows > FEAT ? FEAT : wrows;
        const int n = n0 + li, wrb = nslab * 64;
        const rsrc4_t rw = make_rsrc4(W + (size_t)n0 * wrb, (unsigned)wrows * (unsigned)wrb);
        const rsrc4_t rs = make_rsrc4(SFW, (unsigned)sfw_tiles * (unsigned)nslab * 512u);
        const int sfw_off = (n >> 7) * nslab * 512 + (n & 31) * 16 + ((n >> 5) & 3) * 4;
        const int woff = li * wrb + hb * 16;
#pragma unroll
        for (int j = 0; j < PRE_R; ++j) {
            const int sl = wave + NW * j;
            if (sl < nslab) {
                asm_load4(q[j].sw, rs, sfw_off, sl * 512);
                asm_load16(q[j].w[0], rw, woff, sl * 64);
                if constexpr (FEAT == 32) asm_load16(q[j].w[1], rw, woff + 32, sl * 64);
            }
        }
    }
    // (after the wave's s_waitcnt vmcnt(0): the registers are the asm statement's own from here on)
    __device__ __forceinline__ void landed() {
#pragma unroll
        for (int j = 0; j < PRE_R; ++j) {
            MM_QD_DEVICE_ONLY(asm volatile("" : "+v"(q[j].sw), "+v"(q[j].w[0]));)
            if constexpr (FEAT == 32) { MM_QD_DEVICE_ONLY(asm volatile("" : "+v"(q[j].w[1]));) }
        }
    }
};
__device__ __forceinline__ void wait_all_loads() { MM_QD_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");) }

using dq::LdsMap;
template <bool RMS, int LPG>
__device__ __forceinline__ LdsMap quantize_rows_to_lds(const Args &a, uint8_t *smem) {
    dq::QuantIn q;
    q.X = a.X; q.idx = a.idx; q.M = a.M; q.stage_rows = a.stage_rows;
    q.K[0] = a.K[0]; q.K[1] = a.K[1]; q.K[2] = a.K[2];
    q.mode = 0; q.early = 0;
    q.norm_w = a.norm_w; q.eps = a.eps; q.int_round = a.int_round;
    return dq::quantize_rows_to_lds<NT, RMS, LPG>(q, smem);
}
// ... with `request()` (vector-memory instructions the compiler does not track) called from the phase's hook, once the first batch of
// rows is staged
template <bool RMS, int LPG, class Request>
__device__ __forceinline__ LdsMap quantize_rows_to_lds(const Args &a, uint8_t *smem, Request request) {
    dq::QuantIn q;
    q.X = a.X; q.idx = a.idx; q.M = a.M; q.stage_rows = a.stage_rows;
    q.K[0] = a.K[0]; q.K[1] = a.K[1]; q.K[2] = a.K[2];
    q.mode = 0; q.early = 0;
    q.norm_w = a.norm_w; q.eps = a.eps; q.int_round = a.int_round;
    return dq::quantize_rows_to_lds<NT, RMS, LPG>(q, smem, [&]() { request(); });
}

// LPG: lanes per reordered group in the quantization phase (dq::quantize_rows_to_lds): 2 when the launch's (row, group, half) slots fit
// one pass of the workgroup (M <= 2 at K = 4096), else 1 -- kernels of their own: with both paths in one kernel the one-lane path ran
// 0.3 - 2 us slower than alone (q | k | v with the norm at M = 8: 16.1 -> 18.1 us)
template <bool W4, bool RMS = false, int LPG = 1>
__global__ void __launch_bounds__(NT) qlinear_decode_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [row stage | opN | opS | opO | scales]
    __shared__ float red[3][NW][4][64];      // per segment and wave: accumulator registers 0 .. 3 (token rows 0 .. 7)
    // fp4 weights: the first slabs of the workgroup's first feature block are requested in front of the quantization (Prefetch)
    [[maybe_unused]] SegPrefetch<32> pfN, pfS, pfO;
    LdsMap L;
    constexpr bool PF = W4 && MM_DECODE_PREFETCH != 0;
    auto request = [&]() {
        const int n0 = (int)blockIdx.x * BN;
        pfN.request(a.W[0], a.SFW[0], a.K[0] >> 7, a.N, n0, a.sfw_row_tiles);
        pfS.request(a.W[1], a.SFW[1], a.K[1] >> 7, a.N, n0, a.sfw_row_tiles);
        pfO.request(a.W[2], a.SFW[2], a.K[2] >> 7, a.N, n0, a.sfw_row_tiles);
    };
    if constexpr (PF && MM_DECODE_PREFETCH == 2) L = quantize_rows_to_lds<RMS, LPG>(a, smem, request);
    else L = quantize_rows_to_lds<RMS, LPG>(a, smem);
    if constexpr (PF && MM_DECODE_PREFETCH == 1) request();
    const uint8_t *opN = L.opN, *opS = L.opS, *opO = L.opO, *scales = L.scales;
    const int pN = L.pN, pS = L.pS, pO = L.pO, Gt = L.Gt, gN = L.gN, gS = L.gS;

    // ---- phase 2: weight-streaming GEMM, 32 features per workgroup, K split over the 8 waves (mx_gemm_skinny.hip) ----
    // A workgroup walks the feature blocks blockIdx.x, blockIdx.x + gridDim.x, ...: with more blocks than resident workgroups
    // (fused gate + up: N = 28672 -> 896 blocks) the quantization of phase 1 -- as long as the streaming of one block's 64 KB of
    // weights -- is paid once per RESIDENT workgroup instead of once per block.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg[3] = {a.K[0] >> 7, a.K[1] >> 7, a.K[2] >> 7};
    for (int n0 = blockIdx.x * BN; n0 < a.N; n0 += gridDim.x * BN) {
        v16f accN, accS, accO;
#pragma unroll
        for (int i = 0; i < 16; ++i) accN[i] = accS[i] = accO[i] = 0.0f;
        int r0 = 0;
        if constexpr (PF) {
            if (n0 == (int)blockIdx.x * BN) {        // the block whose first slabs were requested under the quantization
                wait_all_loads();
                pfN.landed(); pfS.landed(); pfO.landed();
                const int wv = __builtin_amdgcn_readfirstlane(wave), li = lane & 31, kb = lane >> 5, sh = 8 * kb;
                const bool valid = li < a.M;
                const int rr = valid ? li : 0;
                // (the arithmetic of run_segment, slab by slab in the same order)
                auto steps = [&](auto EL_, v16f &acc, const SegPrefetch<32> &pf, const uint8_t *xl, int xp, const uint8_t *sl, int nslab) {
                    constexpr int XEL = decltype(EL_)::value;
#pragma unroll
                    for (int j = 0; j < PRE_R; ++j) {
                        const int s = wv + NW * j;
                        if (s < nslab) {
                            v8i xf[2];
                            const v8i zero = {0, 0, 0, 0, 0, 0, 0, 0};
                            xf[0] = xf[1] = zero;
                            int sx = 0;
                            if (valid) {
                                sx = (int)(*reinterpret_cast<const uint32_t *>(sl + rr * Gt + 4 * s) >> sh);
#pragma unroll
                                for (int h = 0; h < 2; ++h) xf[h] = lds_xfrag<XEL>(xl + rr * xp + s * G<XEL>::BYTES, h, kb);
                            }
                            const int sw = pf.q[j].sw >> sh;
                            const v4i a0 = pf.q[j].w[0], a1 = pf.q[j].w[1];
                            const v8i w0 = {a0[0], a0[1], a0[2], a0[3], 0, 0, 0, 0}, w1 = {a1[0], a1[1], a1[2], a1[3], 0, 0, 0, 0};
                            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[0], w0, acc, ElemTraits<XEL>::HW, ElemTraits<EL_FP4>::HW, 0, sx, 0, sw);
                            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[1], w1, acc, ElemTraits<XEL>::HW, ElemTraits<EL_FP4>::HW, 2, sx, 2, sw);
                        }
                    }
                };
                steps(std::integral_constant<int, EL_FP4>{}, accN, pfN, opN, pN, scales, nseg[0]);
                steps(std::integral_constant<int, EL_FP6>{}, accS, pfS, opS, pS, scales + gN, nseg[1]);
                steps(std::integral_constant<int, EL_FP8>{}, accO, pfO, opO, pO, scales + gN + gS, nseg[2]);
                r0 = PRE_R;
            }
        }
        if (nseg[0]) run_segment<EL_FP4, EL_FP4>(accN, opN, pN, scales, Gt, a.W[0], a.SFW[0], nseg[0], a.M, a.N, n0, a.sfw_row_tiles, r0);
        if (nseg[1]) run_segment<EL_FP6, (W4 ? EL_FP4 : EL_FP6)>(accS, opS, pS, scales + gN, Gt, a.W[1], a.SFW[1], nseg[1], a.M, a.N, n0, a.sfw_row_tiles, r0);
        if (nseg[2]) run_segment<EL_FP8, (W4 ? EL_FP4 : EL_FP8)>(accO, opO, pO, scales + gN + gS, Gt, a.W[2], a.SFW[2], nseg[2], a.M, a.N, n0, a.sfw_row_tiles, r0);

        // cross-wave reduction with the reference's rounding chain (as mx_gemm_skinny.hip: per segment the eight waves' sums in wave order, then
        // D = bf16(segment + D)).  M <= 8: token row m = (i & 3) + 8 (i >> 2) + 4 (l >> 5) of accumulator register i is a row of the launch for
        // i < 4 only, so a wave leaves those four registers of every present segment side by side and ONE barrier pair serves the three segments
        // (it was sixteen registers and a barrier pair per segment: six barriers and 48 LDS writes per lane behind every feature block).
        __syncthreads();     // (the previous block's sums have been read)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (nseg[0]) red[0][wave][i][lane] = accN[i];
            if (nseg[1]) red[1][wave][i][lane] = accS[i];
            if (nseg[2]) red[2][wave][i][lane] = accO[i];
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            const int e = threadIdx.x, l = e & 63, i = e >> 6;
            float run = 0.0f;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (nseg[g]) {
                    float s = 0.0f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += (&red[g][w][0][0])[e];
                    s += run;
                    run = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
                }
            }
            const int m = i + 4 * (l >> 5), n = n0 + (l & 31);
            if (m < a.M && n < a.N) {
                uint32_t b = f32_to_bf16_bits(run);
                if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
                a.D[(size_t)m * a.N + n] = (uint16_t)b;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 16-feature variant (v_mfma_scale_f32_16x16x128_f8f6f4: one MFMA per 128-deep slab, tokens on 16 rows): twice the workgroups
// for the same N.  Used while N/32 workgroups would leave half of the CUs idle (N <= 4096 on 256 CUs), where a workgroup's
// weight stream is latency bound (~25 GB/s per CU): down_proj at M = 1 14.8 -> ~10 us.
// Register layouts (tests/test_hw_gpu.py): lane l = (row/col l & 15, K block h = l >> 4); fp4/fp6 lanes hold the 32 elements of
// block h, fp8 lanes hold K = 16h + [0,16) and 64 + 16h + [0,16); the scale byte of a lane belongs to block h.
// ---------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BN16 = 16;

template <int EL>
__device__ __forceinline__ v8i load_wfrag16(__amdgpu_buffer_rsrc_t rsrc, int rowoff, int slab, int h) {
    const int so = slab * G<EL>::BYTES;
    v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EL == EL_FP8) {
        const v4i lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + h * 16, so, 0);
        const v4i hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + 64 + h * 16, so, 0);
        r = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else if constexpr (EL == EL_FP4) {
        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowoff + h * 16, so, 0);
        r = v8i{v[0], v[1], v[2], v[3], 0, 0, 0, 0};
    } else {
        const int o = rowoff + h * 24;
        const v2i a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, so, 0);
        const v2i b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 8, so, 0);
        const v2i c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 16, so, 0);
        r = v8i{a[0], a[1], b[0], b[1], c[0], c[1], 0, 0};
    }
    return r;
}

template <int EL>
__device__ __forceinline__ v8i lds_xfrag16(const uint8_t *p, int h) {
    v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (EL == EL_FP8) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(p + h * 16);
        const uint4 hi = *reinterpret_cast<const uint4 *>(p + 64 + h * 16);
        r = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    } else if constexpr (EL == EL_FP4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + h * 16);
        r = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    } else {
        const uint2 *q = reinterpret_cast<const uint2 *>(p + h * 24);
        const uint2 a = q[0], b = q[1], c = q[2];
        r = v8i{(int)a.x, (int)a.y, (int)b.x, (int)b.y, (int)c.x, (int)c.y, 0, 0};
    }
    return r;
}

template <int XEL, int WEL>
__device__ __forceinline__ void run_segment16(v4f &acc, const uint8_t *xl, int xp, const uint8_t *sl, int sp, const uint8_t *W,
                                              const uint8_t *SFW, int nslab, int M, int N, int n0, int sfw_tiles, int r0 = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, h = lane >> 4;
    const int wrb = nslab * G<WEL>::BYTES;
    int wrows = N - n0;
    wrows = wrows > BN16 ? BN16 : wrows;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W + (size_t)n0 * wrb, (unsigned)wrows * (unsigned)wrb);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(SFW, (unsigned)sfw_tiles * (unsigned)nslab * 512u);
    const int n = n0 + li;
    const int sfw_off = (n >> 7) * nslab * 512 + (n & 31) * 16 + ((n >> 5) & 3) * 4;
    const bool valid = li < M;
    const uint8_t *xrow = xl + (valid ? li : 0) * xp;
    const uint8_t *srow = sl + (valid ? li : 0) * sp;
    const int sh = 8 * h;
    constexpr int DEPTH = MM_DECODE_DEPTH;   // as run_segment
    for (int s0 = wave + r0 * NW; s0 < nslab; s0 += DEPTH * NW) {
        v8i wf[DEPTH];
        int swr[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int s = s0 + j * NW;
            if (s < nslab) {
                swr[j] = __builtin_amdgcn_raw_buffer_load_b32(rsw, sfw_off, s * 512, 0);
                wf[j] = load_wfrag16<WEL>(rw, li * wrb, s, h);
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int s = s0 + j * NW;
            if (s < nslab) {
                int sx = 0;
                v8i xf = {0, 0, 0, 0, 0, 0, 0, 0};
                if (valid) {
                    sx = (int)(*reinterpret_cast<const uint32_t *>(srow + 4 * s) >> sh);
                    xf = lds_xfrag16<XEL>(xrow + s * G<XEL>::BYTES, h);
                }
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xf, wf[j], acc, ElemTraits<XEL>::HW, ElemTraits<WEL>::HW, 0, sx, 0,
                                                                       swr[j] >> sh);
            }
        }
    }
}

template <bool W4, bool RMS = false, int LPG = 1>
__global__ void __launch_bounds__(NT) qlinear_decode16_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ float red[3][NW][4][64];        // the waves' partial sums of N, S, O side by side: ONE barrier (round 6; was two per segment)
    const int n0 = blockIdx.x * BN16;
    // fp4 weights: the first slabs are requested in front of the quantization (Prefetch)
    [[maybe_unused]] SegPrefetch<16> pfN, pfS, pfO;
    LdsMap L;
    constexpr bool PF = W4 && MM_DECODE_PREFETCH != 0;
    auto request = [&]() {
        pfN.request(a.W[0], a.SFW[0], a.K[0] >> 7, a.N, n0, a.sfw_row_tiles);
        pfS.request(a.W[1], a.SFW[1], a.K[1] >> 7, a.N, n0, a.sfw_row_tiles);
        pfO.request(a.W[2], a.SFW[2], a.K[2] >> 7, a.N, n0, a.sfw_row_tiles);
    };
    if constexpr (PF && MM_DECODE_PREFETCH == 2) L = quantize_rows_to_lds<RMS, LPG>(a, smem, request);
    else L = quantize_rows_to_lds<RMS, LPG>(a, smem);
    if constexpr (PF && MM_DECODE_PREFETCH == 1) request();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg[3] = {a.K[0] >> 7, a.K[1] >> 7, a.K[2] >> 7};
    v4f accN = {0, 0, 0, 0}, accS = {0, 0, 0, 0}, accO = {0, 0, 0, 0};
    int r0 = 0;
    if constexpr (PF) {
        wait_all_loads();
        pfN.landed(); pfS.landed(); pfO.landed();
        const int wv = __builtin_amdgcn_readfirstlane(wave), li = lane & 15, h = lane >> 4, sh = 8 * h;
        const bool valid = li < a.M;
        const int rr = valid ? li : 0;
        // (the arithmetic of run_segment16, slab by slab in the same order)
        auto steps = [&](auto EL_, v4f &acc, const SegPrefetch<16> &pf, const uint8_t *xl, int xp, const uint8_t *sl, int nslab) {
            constexpr int XEL = decltype(EL_)::value;
#pragma unroll
            for (int j = 0; j < PRE_R; ++j) {
                const int s = wv + NW * j;
                if (s < nslab) {
                    int sx = 0;
                    v8i xf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (valid) {
                        sx = (int)(*reinterpret_cast<const uint32_t *>(sl + rr * L.Gt + 4 * s) >> sh);
                        xf = lds_xfrag16<XEL>(xl + rr * xp + s * G<XEL>::BYTES, h);
                    }
                    const v4i a0 = pf.q[j].w[0];
                    const v8i w0 = {a0[0], a0[1], a0[2], a0[3], 0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xf, w0, acc, ElemTraits<XEL>::HW, ElemTraits<EL_FP4>::HW, 0, sx, 0, pf.q[j].sw >> sh);
                }
            }
        };
        steps(std::integral_constant<int, EL_FP4>{}, accN, pfN, L.opN, L.pN, L.scales, nseg[0]);
        steps(std::integral_constant<int, EL_FP6>{}, accS, pfS, L.opS, L.pS, L.scales + L.gN, nseg[1]);
        steps(std::integral_constant<int, EL_FP8>{}, accO, pfO, L.opO, L.pO, L.scales + L.gN + L.gS, nseg[2]);
        r0 = PRE_R;
    }
    if (nseg[0]) run_segment16<EL_FP4, EL_FP4>(accN, L.opN, L.pN, L.scales, L.Gt, a.W[0], a.SFW[0], nseg[0], a.M, a.N, n0, a.sfw_row_tiles, r0);
    if (nseg[1]) run_segment16<EL_FP6, (W4 ? EL_FP4 : EL_FP6)>(accS, L.opS, L.pS, L.scales + L.gN, L.Gt, a.W[1], a.SFW[1], nseg[1], a.M, a.N, n0, a.sfw_row_tiles, r0);
    if (nseg[2]) run_segment16<EL_FP8, (W4 ? EL_FP4 : EL_FP8)>(accO, L.opO, L.pO, L.scales + L.gN + L.gS, L.Gt, a.W[2], a.SFW[2], nseg[2], a.M, a.N, n0, a.sfw_row_tiles, r0);

    // cross-wave reduction with the reference's rounding chain, segment by segment in the order N, S, O (the same sums in the same order as
    // mx_gemm_skinny.hip); threads 0..255 own one output element each.  Every wave leaves the partial sums of all present segments first.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (nseg[0]) red[0][wave][i][lane] = accN[i];
        if (nseg[1]) red[1][wave][i][lane] = accS[i];
        if (nseg[2]) red[2][wave][i][lane] = accO[i];
    }
    __syncthreads();
    float run = 0.0f;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (nseg[g]) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += (&red[g][w][0][0])[threadIdx.x];
                s += run;
                run = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
            }
        }
    }
    if (threadIdx.x < 256) {
        const int l = threadIdx.x & 63, i = threadIdx.x >> 6;
        const int m = 4 * (l >> 4) + i;            // D[4 * (lane >> 4) + register][lane & 15]
        const int n = n0 + (l & 15);
        if (m < a.M && n < a.N) {
            uint32_t b = f32_to_bf16_bits(run);
            if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
            a.D[(size_t)m * a.N + n] = (uint16_t)b;
        }
    }
}

}  // namespace decode

// dynamic LDS: the quantized rows and scales of all M rows + as many staged bf16 rows as fit next to the 32 KB reduction buffer
constexpr size_t DECODE_LDS_MAX = 126 * 1024;
static size_t decode_operand_bytes(int M, const int K[3], bool rms = false) {
    return rms ? ((dq::operand_bytes(M, K) + 15) & ~(size_t)15) + dq::rms_bytes(M, K) : dq::operand_bytes(M, K);
}

// features per workgroup: 16 while 32 would leave half of the CUs without a workgroup
static int decode_features(int N) { return 2 * ((N + 31) / 32) <= device_cus() ? 16 : 32; }

// 0: cannot run; 1: can run; 2: can run and is expected to beat quantize + GEMM.  Every workgroup repeats the quantization, in
// ceil(M * K/32 / 512) passes of ~1.4 us, and the workgroups take ceil(N/features / CUs) rounds; measured on MI355X the fused
// kernel wins while rounds * passes <= 2 (q/o up to M = 8, gate/up and down up to M = 2-4) and loses beyond.
int qlinear_decode_supported(int M, int N, const int K[3], bool rms, bool w4) {
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    if (rms && Kt > (size_t)dq::RMS_MAX_K) return 0;
    if (M < 1 || M > 8 || Kt * 2 + decode_operand_bytes(M, K, rms) > DECODE_LDS_MAX) return 0;   // at least one staged row must fit
    // layers wide enough for the streaming kernel's workgroup-local quantization (mx_gemm_stream.hip): it beats quantize + GEMM at
    // M = 1 and 2 (gate/up 10.6-12.7 -> 8.5-9.3 us, fused gate + up 18.3-19.0 -> 16.2-17.7) and ties or loses from M = 4 on
    // (every workgroup repeats the quantization: beyond ~4 rounds of workgroups -- not measured, N > 32768 -- one separate quantize launch is cheaper)
    // With the norm inside (round 5, tools/time_decode.py, K = 4096, us, rmsnorm_quantize + GEMM / one launch): every workgroup repeats
    // the sum of squares too.  Streaming kernel: gate | up N = 28672 (F = 4 on 4 waves) M = 1 18.3 / 17.4 (16.0 once the early-request phase took the norm), M = 4 18.3 / 28.2; N = 14336
    // (F = 2 on 8 waves: 105 + 24 registers, ONE workgroup per CU) M = 1 11.4 / 16.0.  First fused kernel: q | k | v M = 1 11.5 / 9.1,
    // M = 4 11.6 / 11.8; q/o M = 1 9.3 / 7.9, M = 4 9.7 / 10.5.
    if (qlinear_stream_supported(M, N, K, rms, w4)) {
        // (round 6, lane pairs in the quantization phase: the 32-feature norm kernel holds two workgroups per CU -- N = 14336 M = 1 13.3 / 9.5,
        // M = 2 13.6 / 11.8; gate | up M = 1 17.8 / 15.0, M = 2 17.8 / 16.7.  Only while the slots fit the early-request phase, 2 M K / 32 <= 512:
        // at K = 8192 M = 2 goes through the staged path, 13.0 -> 13.9)
        // Beyond K = 4096 the staged row, the operands and the norm's weight vector leave room for ONE workgroup per CU: one round of them
        // only (Llama-3-70B shapes, K = 8192, M = 1: q/o N = 8192 12.8 / 11.1; q | k | v N = 10240 16.7 / 18.6)
        if (rms) return (M <= 2 && N <= 32768 && 2 * (size_t)M * (Kt / 32) <= 512 && (Kt <= 4096 || (N + 31) / 32 <= device_cus())) ? 2 : 1;
        return (M <= 2 && N <= 32768) ? 2 : 1;
    }
    const int feat = decode_features(N), cus = device_cus();
    const int rounds = ((N + feat - 1) / feat + cus - 1) / cus;
    const int passes = (int)((M * (Kt / 32) + decode::NT - 1) / decode::NT);
    if (rms) return (rounds == 1 && M <= 2) ? 2 : 1;
    return rounds * passes <= 2 ? 2 : 1;
}

hipError_t launch_qlinear_decode(const void *X, const int16_t *idx, const uint8_t *const W[3], const uint8_t *const SFW[3],
                                 int M, int N, const int K[3], bool w4, int round_per_segment, const void *bias, void *D,
                                 hipStream_t stream, const NormArgs &norm) {
    using namespace decode;
    const bool rms = norm.weight != nullptr;
    if (qlinear_stream_supported(M, N, K, rms, w4)) return launch_qlinear_stream(X, idx, W, SFW, M, N, K, w4, round_per_segment, bias, D, stream, norm);
    Args a;
    a.norm_w = (const uint16_t *)norm.weight;
    a.eps = norm.eps;
    a.int_round = norm.int_round;
    a.X = (const uint16_t *)X;
    a.idx = idx;
    for (int i = 0; i < 3; ++i) {
        a.W[i] = W[i];
        a.SFW[i] = SFW[i];
        a.K[i] = K[i];
    }
    a.M = M;
    a.N = N;
    a.sfw_row_tiles = (N + 127) / 128;
    a.round_per_segment = round_per_segment;
    a.bias = (const uint16_t *)bias;
    a.D = (uint16_t *)D;
    const size_t Kt = (size_t)K[0] + K[1] + K[2], ops = decode_operand_bytes(M, K, rms);
    int stage_rows = (int)((DECODE_LDS_MAX - ops) / (Kt * 2));
    stage_rows = stage_rows > M ? M : stage_rows;
    a.stage_rows = stage_rows;
    const size_t lds = (size_t)stage_rows * Kt * 2 + ops;
    const bool f16 = decode_features(N) == 16;
    // lane pairs in the quantization phase when one pass of the workgroup covers every (row, group, half) slot (see the kernels)
    const bool pairs = 2 * (size_t)stage_rows * (Kt / 32) <= (size_t)NT;
    auto pick = [&](auto lpg_) {
        constexpr int L = decltype(lpg_)::value;
        return rms ? (f16 ? (w4 ? qlinear_decode16_kernel<true, true, L> : qlinear_decode16_kernel<false, true, L>)
                          : (w4 ? qlinear_decode_kernel<true, true, L> : qlinear_decode_kernel<false, true, L>))
                   : (f16 ? (w4 ? qlinear_decode16_kernel<true, false, L> : qlinear_decode16_kernel<false, false, L>)
                          : (w4 ? qlinear_decode_kernel<true, false, L> : qlinear_decode_kernel<false, false, L>));
    };
    auto kern = pairs ? pick(std::integral_constant<int, 2>{}) : pick(std::integral_constant<int, 1>{});
    static DynamicLdsOnce done[16];
    if (hipError_t e = done[(pairs ? 8 : 0) + (rms ? 4 : 0) + (f16 ? 2 : 0) + (w4 ? 0 : 1)].ensure(reinterpret_cast<const void *>(kern), (int)DECODE_LDS_MAX); e != hipSuccess)
        return e;
    const int feat = f16 ? BN16 : BN;
    int blocks = (N + feat - 1) / feat;
    if (!f16) {
        // The 32-feature kernel walks its feature blocks (see its phase 2): one workgroup per CU.  Measured (tools/time_decode.py,
        // K = 4096, (2048,128,1920), one box, us at M = 1 / 4 / 8): gate/up N = 14336 one workgroup per block 13.0 / 16.0 / 20.2,
        // one per CU 11.4 / 13.4 / 15.8, two per CU 13.1 / 16.1 / 20.2; fused gate + up N = 28672: 26.1 / 32.3 / 40.2 against
        // 20.9 / 23.1 / 25.4 (two per CU 22.7 / 26.6 / 31.3).  MICROMIX_DECODE_PERSIST=0 restores one workgroup per block (A/B runs).
        static const bool persist = [] { const char *e = getenv("MICROMIX_DECODE_PERSIST"); return !(e && e[0] == '0'); }();
        const int resident = device_cus();
        if (persist && blocks > resident) blocks = resident;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, stream, a);
    return hipGetLastError();
}

}  // namespace mm
