// Skinny (M <= 64) fused three-segment MX GEMM for gfx950: the decode / small-batch path of mm_matmul.
//
// Same arithmetic and reference citations as mx_gemm.hip (gemm.cu:26-78; per-segment bf16 rounding order of the chained
// reference kernels).  At these token counts the product is bound by streaming the packed weights once from HBM
// (N*K/2 bytes in the production MXFP4 weight mode), not by the MFMA, so the mapping is the opposite of the large-M kernel:
//  * one workgroup = 32 output features x all tokens; its 8 waves split K between them (slab s goes to wave s % 8) and
//    every wave streams its weight rows straight into registers -- no LDS staging, the operand is read exactly once
//    (cdna_hip_programming.md: "GEMV / M <= 16 decode weights: load straight to VGPRs");
//  * activations (M x K, a few hundred KB) and the scale-factor atoms are read by every wave from L2;
//  * MFMA 32x32x64 with the tokens on the rows (padded to 32 or 64, rows past M read as zero through the buffer
//    descriptor's range check) and the 32 features on the columns;
//  * each wave keeps ONE accumulator set PER SEGMENT (16 registers per 32 tokens), the eight partial sums of a segment
//    meet in LDS, and the chain D = bf16(N); D = bf16(S + D); D = bf16(O + D) is applied on the reduced values, so the
//    reference's rounding order is reproduced exactly although K is split across waves.
#include <stdlib.h>

#include "mx_common.h"
#include "mx_kernels.h"

namespace mm {
namespace skinny {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
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

// One lane's fragment for MFMA step h of slab `slab`: row byte offset `rowoff` inside the descriptor, kb = lane >> 5.
// Register layouts as in mx_gemm256.hip (fp8: two 16-element halves; fp4/fp6: block 2h + kb).
template <int EL>
__device__ __forceinline__ v8i load_frag(__amdgpu_buffer_rsrc_t rsrc, int rowoff, int slab, int h, int kb) {
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
        typedef int v2i __attribute__((ext_vector_type(2)));
        const int o = rowoff + (2 * h + kb) * 24;
        const v2i a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, so, 0);
        const v2i b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 8, so, 0);
        const v2i c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 16, so, 0);
        r = v8i{a[0], a[1], b[0], b[1], c[0], c[1], 0, 0};
    }
    return r;
}

// One segment: this wave's share of the slabs (slab = wave, wave + 8, ...), accumulated into acc[TM].
template <int XEL, int WEL, int TM>
__device__ __forceinline__ void run_segment(v16f (&acc)[TM], const uint8_t *X, const uint8_t *W, const uint8_t *SFX,
                                            const uint8_t *SFW, int nslab, int M, int N, int n0, int sfx_tiles, int sfw_tiles) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kb = lane >> 5;
    const int xrb = nslab * G<XEL>::BYTES, wrb = nslab * G<WEL>::BYTES;
    int wrows = N - n0;
    wrows = wrows > BN ? BN : wrows;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, (unsigned)M * (unsigned)xrb);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W + (size_t)n0 * wrb, (unsigned)wrows * (unsigned)wrb);
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(SFX, (unsigned)sfx_tiles * (unsigned)nslab * 512u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(SFW, (unsigned)sfw_tiles * (unsigned)nslab * 512u);
    // scale dword of (row, slab): atom (row >> 7, slab) + (row & 31) * 16 + ((row >> 5) & 3) * 4
    const int n = n0 + li;
    const int sfw_off = (n >> 7) * nslab * 512 + (n & 31) * 16 + ((n >> 5) & 3) * 4;
    int sfx_off[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) sfx_off[t] = li * 16 + t * 4;  // rows t*32 + li < 128: atom row-tile 0
    const int sh = 8 * kb;

    // Two slabs in flight per wave: the loads of slab s + 8 are issued before the MFMAs of slab s wait for theirs (one slab at a
    // time, a wave paid a full memory round trip per slab: four in a row at K = 4096).
    struct Ops {
        v8i xf[TM][2], wf[2];
        int sx[TM], sw;
    };
    auto load = [&](Ops &o, int s) {
        o.sw = __builtin_amdgcn_raw_buffer_load_b32(rsw, sfw_off, s * 512, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) o.wf[h] = load_frag<WEL>(rw, li * wrb, s, h, kb);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            o.sx[t] = __builtin_amdgcn_raw_buffer_load_b32(rsx, sfx_off[t], s * 512, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) o.xf[t][h] = load_frag<XEL>(rx, (t * 32 + li) * xrb, s, h, kb);
        }
    };
    auto mfma = [&](const Ops &o) {
        const int sw = o.sw >> sh;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int sx = o.sx[t] >> sh;
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(o.xf[t][0], o.wf[0], acc[t], ElemTraits<XEL>::HW,
                                                                    ElemTraits<WEL>::HW, 0, sx, 0, sw);
            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(o.xf[t][1], o.wf[1], acc[t], ElemTraits<XEL>::HW,
                                                                    ElemTraits<WEL>::HW, 2, sx, 2, sw);
        }
    };
    Ops o0, o1;
    int s = wave;
    if (s < nslab) load(o0, s);
    while (s < nslab) {
        const int s1 = s + NW, s2 = s + 2 * NW;
        if (s1 < nslab) load(o1, s1);
        mfma(o0);
        if (s1 >= nslab) break;
        if (s2 < nslab) load(o0, s2);
        mfma(o1);
        s = s2;
    }
}

template <bool W4, int TM>
__device__ __forceinline__ void skinny_body(const GemmArgs &a) {
    // TM = 1 (two workgroups per CU at 128 registers): a finished segment's partial sums wait in LDS, not in registers -- with one
    // accumulator set per segment alive to the end (48 registers) beside two slabs of operands the kernels spilled 2 (fp4 weights) and
    // 17 (fp8 weights) registers to scratch.  PARK = LDS images that can hold a segment while a later one runs: two for
    // TM = 1 (2 x 32 KB; the third segment goes from the registers into image 0 once N has been summed), none beyond the one
    // reduction image for TM = 2 (64 KB each, one workgroup per CU, 256 registers: nothing spills there).
    constexpr int PARK = TM == 1 ? 2 : 1;
    __shared__ float red[PARK][NW][TM][16][64];
    const int n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg[3] = {a.K[0] >> 7, a.K[1] >> 7, a.K[2] >> 7};

    // cross-wave reduction, one segment at a time; thread `threadIdx.x` owns elements e = threadIdx.x + 512 * j of the
    // TM x 16 x 64 accumulator image and carries the running D through the reference's rounding chain
    constexpr int PER = TM * 16 * 64 / NT;  // 2 * TM
    float run[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) run[j] = 0.0f;
    auto park = [&](const v16f (&acc)[TM], int img) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[img][wave][t][i][lane] = acc[t][i];
    };
    auto sum_image = [&](int img) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = threadIdx.x + NT * j;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += (&red[img][w][0][0][0])[e];
            s += run[j];
            run[j] = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
        }
    };
    auto zero = [&](v16f (&acc)[TM]) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    };
    if constexpr (TM == 1) {
        v16f acc[TM];
        // N -> image 0, S -> image 1 (each wave writes its own slots: no barrier), O stays in the registers until N has been summed
        if (nseg[0]) { zero(acc); run_segment<EL_FP4, EL_FP4, TM>(acc, a.X[0], a.W[0], a.SFX[0], a.SFW[0], nseg[0], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles); park(acc, 0); }
        if (nseg[1]) { zero(acc); run_segment<EL_FP6, (W4 ? EL_FP4 : EL_FP6), TM>(acc, a.X[1], a.W[1], a.SFX[1], a.SFW[1], nseg[1], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles); park(acc, 1); }
        if (nseg[2]) { zero(acc); run_segment<EL_FP8, (W4 ? EL_FP4 : EL_FP8), TM>(acc, a.X[2], a.W[2], a.SFX[2], a.SFW[2], nseg[2], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles); }
        __syncthreads();
        if (nseg[0]) sum_image(0);
        if (nseg[1]) sum_image(1);
        if (nseg[2]) {
            __syncthreads();          // every thread is past its reads of image 0
            park(acc, 0);
            __syncthreads();
            sum_image(0);
        }
    } else {
        v16f accN[TM], accS[TM], accO[TM];
        zero(accN);
        zero(accS);
        zero(accO);
        if (nseg[0]) run_segment<EL_FP4, EL_FP4, TM>(accN, a.X[0], a.W[0], a.SFX[0], a.SFW[0], nseg[0], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);
        if (nseg[1]) run_segment<EL_FP6, (W4 ? EL_FP4 : EL_FP6), TM>(accS, a.X[1], a.W[1], a.SFX[1], a.SFW[1], nseg[1], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);
        if (nseg[2]) run_segment<EL_FP8, (W4 ? EL_FP4 : EL_FP8), TM>(accO, a.X[2], a.W[2], a.SFX[2], a.SFW[2], nseg[2], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);
        auto reduce = [&](const v16f (&acc)[TM]) {
            __syncthreads();
            park(acc, 0);
            __syncthreads();
            sum_image(0);
        };
        if (nseg[0]) reduce(accN);
        if (nseg[1]) reduce(accS);
        if (nseg[2]) reduce(accO);
    }

    // element e = (t, i, l): token row t*32 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), feature n0 + (l & 31)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = threadIdx.x + NT * j;
        const int l = e & 63, i = (e >> 6) & 15, t = e >> 10;
        const int m = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        const int n = n0 + (l & 31);
        if (m < a.M && n < a.N) {
            if (a.out_f32) {     // MM_OUT_F32: the fp32 sum itself
                reinterpret_cast<float *>(a.D)[(size_t)m * a.N + n] = run[j];
                continue;
            }
            uint32_t b = f32_to_bf16_bits(run[j]);
            if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
            a.D[(size_t)m * a.N + n] = (uint16_t)b;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 16-feature variant (v_mfma_scale_f32_16x16x128_f8f6f4, one MFMA per slab and 16-token tile): twice the workgroups for the same N.
// Used while N/32 workgroups would leave half of the CUs idle: a workgroup's weight stream is latency bound (~25 GB/s per CU),
// so for N <= 4096 the extra workgroups nearly halve the time of the long-K layers (down_proj).
// Register layouts (tests/test_hw_gpu.py): lane l = (row/col l & 15, K block h = l >> 4); fp4/fp6 lanes hold the 32 elements of
// block h, fp8 lanes hold K = 16h + [0,16) and 64 + 16h + [0,16); the scale byte of a lane belongs to block h.
// ---------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BN16 = 16;

template <int EL>
__device__ __forceinline__ v8i load_frag16(__amdgpu_buffer_rsrc_t rsrc, int rowoff, int slab, int h) {
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
        typedef int v2i __attribute__((ext_vector_type(2)));
        const int o = rowoff + h * 24;
        const v2i a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, so, 0);
        const v2i b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 8, so, 0);
        const v2i c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o + 16, so, 0);
        r = v8i{a[0], a[1], b[0], b[1], c[0], c[1], 0, 0};
    }
    return r;
}

// T16 token tiles of 16 rows each (M <= 16 * T16)
template <int XEL, int WEL, int T16>
__device__ __forceinline__ void run_segment16(v4f (&acc)[T16], const uint8_t *X, const uint8_t *W, const uint8_t *SFX,
                                              const uint8_t *SFW, int nslab, int M, int N, int n0, int sfx_tiles, int sfw_tiles) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, h = lane >> 4;
    const int xrb = nslab * G<XEL>::BYTES, wrb = nslab * G<WEL>::BYTES;
    int wrows = N - n0;
    wrows = wrows > BN16 ? BN16 : wrows;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, (unsigned)M * (unsigned)xrb);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W + (size_t)n0 * wrb, (unsigned)wrows * (unsigned)wrb);
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(SFX, (unsigned)sfx_tiles * (unsigned)nslab * 512u);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(SFW, (unsigned)sfw_tiles * (unsigned)nslab * 512u);
    const int n = n0 + li;
    const int sfw_off = (n >> 7) * nslab * 512 + (n & 31) * 16 + ((n >> 5) & 3) * 4;
    int sfx_off[T16];
#pragma unroll
    for (int t = 0; t < T16; ++t) sfx_off[t] = ((t & 1) * 16 + li) * 16 + (t >> 1) * 4;   // token row 16t + li < 64: atom row-tile 0
    const int sh = 8 * h;
    for (int s = wave; s < nslab; s += NW) {
        const int sw = __builtin_amdgcn_raw_buffer_load_b32(rsw, sfw_off, s * 512, 0) >> sh;
        const v8i wf = load_frag16<WEL>(rw, li * wrb, s, h);
        int sx[T16];
        v8i xf[T16];
#pragma unroll
        for (int t = 0; t < T16; ++t) {
            sx[t] = __builtin_amdgcn_raw_buffer_load_b32(rsx, sfx_off[t], s * 512, 0) >> sh;
            xf[t] = load_frag16<XEL>(rx, (t * 16 + li) * xrb, s, h);
        }
#pragma unroll
        for (int t = 0; t < T16; ++t)
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xf[t], wf, acc[t], ElemTraits<XEL>::HW, ElemTraits<WEL>::HW, 0,
                                                                     sx[t], 0, sw);
    }
}

template <bool W4, int T16>
__device__ __forceinline__ void skinny16_body(const GemmArgs &a) {
    __shared__ float red[NW][T16 * 4][64];
    const int n0 = blockIdx.x * BN16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg[3] = {a.K[0] >> 7, a.K[1] >> 7, a.K[2] >> 7};
    v4f accN[T16], accS[T16], accO[T16];
#pragma unroll
    for (int t = 0; t < T16; ++t) accN[t] = accS[t] = accO[t] = v4f{0, 0, 0, 0};
    if (nseg[0]) run_segment16<EL_FP4, EL_FP4, T16>(accN, a.X[0], a.W[0], a.SFX[0], a.SFW[0], nseg[0], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);
    if (nseg[1]) run_segment16<EL_FP6, (W4 ? EL_FP4 : EL_FP6), T16>(accS, a.X[1], a.W[1], a.SFX[1], a.SFW[1], nseg[1], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);
    if (nseg[2]) run_segment16<EL_FP8, (W4 ? EL_FP4 : EL_FP8), T16>(accO, a.X[2], a.W[2], a.SFX[2], a.SFW[2], nseg[2], a.M, a.N, n0, a.sfx_row_tiles, a.sfw_row_tiles);

    // cross-wave reduction per segment with the reference's rounding chain; element e = (t, i, l) of the T16 x 4 x 64
    // accumulator image: token 16t + 4 * (l >> 4) + i, feature n0 + (l & 15); threads 0..255 own elements e = tid + 256 j
    float run[T16];
#pragma unroll
    for (int j = 0; j < T16; ++j) run[j] = 0.0f;
    auto reduce = [&](const v4f (&acc)[T16]) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][t * 4 + i][lane] = acc[t][i];
        __syncthreads();
        if (threadIdx.x < 256) {
#pragma unroll
            for (int j = 0; j < T16; ++j) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += (&red[w][0][0])[threadIdx.x + 256 * j];
                s += run[j];
                run[j] = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
            }
        }
    };
    if (nseg[0]) reduce(accN);
    if (nseg[1]) reduce(accS);
    if (nseg[2]) reduce(accO);
    if (threadIdx.x < 256) {
        const int l = threadIdx.x & 63, i = threadIdx.x >> 6;
        const int n = n0 + (l & 15);
#pragma unroll
        for (int j = 0; j < T16; ++j) {
            const int m = 16 * j + 4 * (l >> 4) + i;            // D[4 * (lane >> 4) + register][lane & 15] per token tile
            if (m < a.M && n < a.N) {
                if (a.out_f32) {     // MM_OUT_F32: the fp32 sum itself
                    reinterpret_cast<float *>(a.D)[(size_t)m * a.N + n] = run[j];
                    continue;
                }
                uint32_t b = f32_to_bf16_bits(run[j]);
                if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
                a.D[(size_t)m * a.N + n] = (uint16_t)b;
            }
        }
    }
}

template <bool W4, int TM>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(TM == 1 ? 4 : 2, TM == 1 ? 4 : 2))) mx_gemm_skinny_kernel(GemmArgs a) { skinny_body<W4, TM>(a); }
template <bool W4, int T16>
__global__ void __launch_bounds__(NT) mx_gemm_skinny16_kernel(GemmArgs a) { skinny16_body<W4, T16>(a); }

// Grouped launch (MoE experts): blockIdx.y picks one of up to MM_MAX_GROUPS problems that share
// N, the K split and the weight mode but have their own operands, outputs and token counts; the argument blocks travel in the
// kernel arguments, so no device scratch is needed.  The token-tile count is the maximum over the groups (rows past a group's
// M read as zeros and are not stored); a group with M = 0 returns at once.
template <bool W4, int TM>
__global__ void __launch_bounds__(NT) mx_gemm_skinny_grouped_kernel(GroupedGemmArgs ga) {
    const GemmArgs &a = ga.g[blockIdx.y];
    if (a.M > 0) skinny_body<W4, TM>(a);
}
template <bool W4, int T16>
__global__ void __launch_bounds__(NT) mx_gemm_skinny16_grouped_kernel(GroupedGemmArgs ga) {
    const GemmArgs &a = ga.g[blockIdx.y];
    if (a.M > 0) skinny16_body<W4, T16>(a);
}

}  // namespace skinny

hipError_t launch_mx_gemm_skinny(const GemmArgs &a, bool w4, hipStream_t stream) {
    using namespace skinny;
    const int cus = device_cus();
    const int blocks = (a.N + BN - 1) / BN;
    static const int force16 = getenv("MICROMIX_SKINNY16") ? atoi(getenv("MICROMIX_SKINNY16")) : 0;   // kernel-developer override
    if (2 * blocks <= cus || (force16 && a.M <= 16)) {   // 16 features per workgroup while 32 would leave half of the CUs idle
        const int b16 = (a.N + BN16 - 1) / BN16;
#define MM_L16(T_)                                                                                              \
    do {                                                                                                        \
        if (w4) hipLaunchKernelGGL((mx_gemm_skinny16_kernel<true, T_>), dim3(b16), dim3(NT), 0, stream, a);     \
        else hipLaunchKernelGGL((mx_gemm_skinny16_kernel<false, T_>), dim3(b16), dim3(NT), 0, stream, a);       \
    } while (0)
        if (a.M <= 16) MM_L16(1);
        else if (a.M <= 32) MM_L16(2);
        else if (a.M <= 48) MM_L16(3);
        else MM_L16(4);
#undef MM_L16
    } else if (a.M <= 32) {
        if (w4) hipLaunchKernelGGL((mx_gemm_skinny_kernel<true, 1>), dim3(blocks), dim3(NT), 0, stream, a);
        else hipLaunchKernelGGL((mx_gemm_skinny_kernel<false, 1>), dim3(blocks), dim3(NT), 0, stream, a);
    } else {
        if (w4) hipLaunchKernelGGL((mx_gemm_skinny_kernel<true, 2>), dim3(blocks), dim3(NT), 0, stream, a);
        else hipLaunchKernelGGL((mx_gemm_skinny_kernel<false, 2>), dim3(blocks), dim3(NT), 0, stream, a);
    }
    return hipGetLastError();
}

// all groups must have M <= 64 and share N, K[], round_per_segment; returns hipErrorInvalidValue otherwise
hipError_t launch_mx_gemm_skinny_grouped(const GroupedGemmArgs &ga, int max_m, bool w4, hipStream_t stream) {
    using namespace skinny;
    if (ga.ngroups < 1 || ga.ngroups > MM_MAX_GROUPS || max_m < 1 || max_m > 64) return hipErrorInvalidValue;
    const int cus = device_cus();
    const int N = ga.g[0].N, blocks = (N + BN - 1) / BN;
    // with G groups in one launch there are G x blocks workgroups: 16 features per workgroup while that still leaves CUs idle
    if (2 * blocks * ga.ngroups <= cus) {
        const dim3 grid((N + BN16 - 1) / BN16, ga.ngroups);
#define MM_G16(T_)                                                                                                     \
    do {                                                                                                               \
        if (w4) hipLaunchKernelGGL((mx_gemm_skinny16_grouped_kernel<true, T_>), grid, dim3(NT), 0, stream, ga);       \
        else hipLaunchKernelGGL((mx_gemm_skinny16_grouped_kernel<false, T_>), grid, dim3(NT), 0, stream, ga);         \
    } while (0)
        if (max_m <= 16) MM_G16(1);
        else if (max_m <= 32) MM_G16(2);
        else if (max_m <= 48) MM_G16(3);
        else MM_G16(4);
#undef MM_G16
    } else {
        const dim3 grid(blocks, ga.ngroups);
        if (max_m <= 32) {
            if (w4) hipLaunchKernelGGL((mx_gemm_skinny_grouped_kernel<true, 1>), grid, dim3(NT), 0, stream, ga);
            else hipLaunchKernelGGL((mx_gemm_skinny_grouped_kernel<false, 1>), grid, dim3(NT), 0, stream, ga);
        } else {
            if (w4) hipLaunchKernelGGL((mx_gemm_skinny_grouped_kernel<true, 2>), grid, dim3(NT), 0, stream, ga);
            else hipLaunchKernelGGL((mx_gemm_skinny_grouped_kernel<false, 2>), grid, dim3(NT), 0, stream, ga);
        }
    }
    return hipGetLastError();
}

}  // namespace mm
