// Fused three-segment mixed-precision block-scaled GEMM for gfx950 (CDNA4).
//
// Computes what the reference runs as up to three chained CUTLASS SM120 GEMMs
// (mgemm/src/gemm.cu:26-78; per-segment kernels mgemm/src/w4a4.cu, w4a6.cu, w4a8.cu,
// w6a6.cu, w8a8.cu):
//     D = bf16(A_N*B_N^T);  D = bf16(A_S*B_S^T + D);  D = bf16(A_O*B_O^T + D)
// with per-32-element UE8M0 scales on both operands, fp32 accumulation.  Here it is ONE
// launch: the K loop walks the N|S|O segments with a single fp32 accumulator in
// registers, switching the MFMA operand formats per segment; in the default
// MM_ROUND_PER_SEGMENT mode the accumulator is rounded through bf16 at each segment
// boundary so that the result follows the reference's rounding order without D ever
// leaving the register file.
//
// MI355X mapping:
//  * v_mfma_scale_f32_32x32x64_f8f6f4: lane l supplies row/col (l & 31); its scale VGPR
//    (byte picked by op_sel) carries the UE8M0 scale of K block (l >> 5).  For fp4/fp6 the
//    lane's registers hold exactly that block; for fp8 they hold two 16-element halves
//    (see load_frag) -- layouts verified on hardware by tests/test_hw_gpu.py.
//  * the MFMA "A" (row) operand is the WEIGHT tile and the "B" (column) operand the
//    ACTIVATION tile: the accumulator then has the token index on the lane and 4
//    consecutive output features in consecutive registers -> packed 8-byte bf16 stores.
//  * the packed operands are copied global->LDS with 16-byte LDS-DMA
//    (global_load_lds_dwordx4), rows kept dense (128 B fp8 / 96->128 B fp6 / 64 B fp4
//    per 128-K slab) and XOR-swizzled on the SOURCE address so that the ds_read_b128
//    fragment reads are bank-conflict free.
//  * the reference's SF atom (128 rows x 4 blocks = 512 B, 16 B per (row%32)) is read
//    directly: one dword per lane holds the 4 block scales of its row for a 128-K slab.
#include "mx_common.h"
#include "mx_kernels.h"

namespace mm {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------
// per-format geometry of one 128-K slab of one operand row
// ---------------------------------------------------------------------------------
template <int EL> struct Slab;
template <> struct Slab<EL_FP8> { static constexpr int GBYTES = 128, PITCH = 128, VALID_CHUNKS = 8; };
template <> struct Slab<EL_FP6> { static constexpr int GBYTES = 96, PITCH = 128, VALID_CHUNKS = 6; };
template <> struct Slab<EL_FP4> { static constexpr int GBYTES = 64, PITCH = 64, VALID_CHUNKS = 4; };

// XOR applied to the 16-byte chunk index of a row (same involution on the DMA source
// address and on the fragment read address).
template <int PITCH> __device__ __forceinline__ int swz(int row) {
    if constexpr (PITCH == 128) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Stage ROWS rows of one operand's 128-K slab into LDS (lane-linear image, 16 B per lane).
//   g_base   : first byte of the operand segment
//   row0     : first row of the tile; rows clamped to [0, nrows-1]
//   rowbytes : bytes per operand row in global memory
//   koff     : byte offset of the slab inside a row
template <int EL, int ROWS, int NT>
__device__ __forceinline__ void stage_tile(const uint8_t *__restrict__ g_base, int row0, int nrows, size_t rowbytes,
                                           size_t koff, uint8_t *lds_tile) {
    using S = Slab<EL>;
    constexpr int CPR = S::PITCH / 16;             // chunks per LDS row
    constexpr int ITERS = ROWS * CPR / NT;
    static_assert(ROWS * CPR % NT == 0, "tile must be a whole number of wave instructions");
    const int tid = threadIdx.x;
    const int wave_first = tid & ~63;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int slot = it * NT + tid;
        const int row = slot / CPR;
        int c = (slot % CPR) ^ swz<S::PITCH>(row);
        if constexpr (S::VALID_CHUNKS < CPR) c = c < S::VALID_CHUNKS ? c : c - S::VALID_CHUNKS;  // fp6 pad: any valid bytes
        int grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;
        const uint8_t *src = g_base + (size_t)grow * rowbytes + koff + (size_t)c * 16;
        uint8_t *dst = lds_tile + (size_t)(it * NT + wave_first) * 16;  // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
}

// One lane's 32-element fragment for MFMA step h (K = 64h .. 64h+63 of the slab) of LDS row `row`;
// kb = lane >> 5.  fp4 / fp6: the lane holds the 32 consecutive elements of K block 2h + kb.
// fp8 (measured on gfx950, tools/fp8_probe.py): registers 0-3 hold K = 64h + 16kb + [0,16) and
// registers 4-7 hold K = 64h + 32 + 16kb + [0,16) -- the scale byte still belongs to block 2h + kb.
template <int EL>
__device__ __forceinline__ v8i load_frag(const uint8_t *lds_tile, int row, int h, int kb) {
    using S = Slab<EL>;
    const int sw = swz<S::PITCH>(row);
    const uint8_t *rp = lds_tile + row * S::PITCH;
    v8i r;
    if constexpr (EL == EL_FP8) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(rp + (((4 * h + kb) ^ sw) << 4));
        const uint4 hi = *reinterpret_cast<const uint4 *>(rp + (((4 * h + 2 + kb) ^ sw) << 4));
        r = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    } else if constexpr (EL == EL_FP4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(rp + (((2 * h + kb) ^ sw) << 4));
        r = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    } else {
        uint2 t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = 3 * (2 * h + kb) + i;  // 8-byte unit inside the 96-byte row
            t[i] = *reinterpret_cast<const uint2 *>(rp + ((((u >> 1) ^ sw)) << 4) + ((u & 1) << 3));
        }
        r = v8i{(int)t[0].x, (int)t[0].y, (int)t[1].x, (int)t[1].y, (int)t[2].x, (int)t[2].y, 0, 0};
    }
    return r;
}

// ---------------------------------------------------------------------------------
// kernel configuration
// ---------------------------------------------------------------------------------
template <int WAVES_N_, int WAVES_M_, int TN_, int TM_> struct Cfg {
    static constexpr int WAVES_N = WAVES_N_, WAVES_M = WAVES_M_, TN = TN_, TM = TM_;
    static constexpr int BN = WAVES_N * TN * 32;  // weight rows (output features) per workgroup
    static constexpr int BM = WAVES_M * TM * 32;  // activation rows (tokens) per workgroup
    static constexpr int NT = 64 * WAVES_N * WAVES_M;
    static constexpr int W_TILE_BYTES = BN * 128;  // sized for the widest format
    static constexpr int X_TILE_BYTES = BM * 128;
    static constexpr int STAGE_BYTES = W_TILE_BYTES + X_TILE_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
};


// One segment: acc += W_seg * X_seg^T over Kseg, XEL/WEL element kinds.
template <class C, int XEL, int WEL>
__device__ __forceinline__ void run_segment(v16f (&acc)[C::TN][C::TM], const uint8_t *__restrict__ X,
                                            const uint8_t *__restrict__ W, const uint8_t *__restrict__ SFX,
                                            const uint8_t *__restrict__ SFW, int Kseg, int M, int N, int m0, int n0,
                                            int sfx_tiles, int sfw_tiles, uint8_t *smem) {
    using SX = Slab<XEL>;
    using SW = Slab<WEL>;
    const int nslab = Kseg >> 7;
    const size_t x_rowbytes = (size_t)(Kseg >> 7) * SX::GBYTES;
    const size_t w_rowbytes = (size_t)(Kseg >> 7) * SW::GBYTES;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wn = wave / C::WAVES_M, wm = wave % C::WAVES_M;
    const int li = lane & 31, kb = lane >> 5;

    // scale addressing: dword (row & 31) * 16 + ((row >> 5) & 3) * 4 of atom (row >> 7, slab)
    size_t sfw_off[C::TN], sfx_off[C::TM];
#pragma unroll
    for (int t = 0; t < C::TN; ++t) {
        const int r = n0 + (wn * C::TN + t) * 32;  // first row of this 32-row MFMA tile
        int rt = r >> 7;
        rt = rt < sfw_tiles ? rt : sfw_tiles - 1;
        sfw_off[t] = (size_t)rt * (size_t)nslab * 512u + (size_t)li * 16u + (size_t)((r >> 5) & 3) * 4u;
    }
#pragma unroll
    for (int t = 0; t < C::TM; ++t) {
        const int r = m0 + (wm * C::TM + t) * 32;
        int rt = r >> 7;
        rt = rt < sfx_tiles ? rt : sfx_tiles - 1;
        sfx_off[t] = (size_t)rt * (size_t)nslab * 512u + (size_t)li * 16u + (size_t)((r >> 5) & 3) * 4u;
    }

    uint8_t *stage0 = smem, *stage1 = smem + C::STAGE_BYTES;

    // prologue: slab 0
    stage_tile<WEL, C::BN, C::NT>(W, n0, N, w_rowbytes, 0, stage0);
    stage_tile<XEL, C::BM, C::NT>(X, m0, M, x_rowbytes, 0, stage0 + C::W_TILE_BYTES);
    uint32_t sw_cur[C::TN], sx_cur[C::TM];
#pragma unroll
    for (int t = 0; t < C::TN; ++t) sw_cur[t] = *reinterpret_cast<const uint32_t *>(SFW + sfw_off[t]);
#pragma unroll
    for (int t = 0; t < C::TM; ++t) sx_cur[t] = *reinterpret_cast<const uint32_t *>(SFX + sfx_off[t]);
    __syncthreads();

    for (int s = 0; s < nslab; ++s) {
        uint8_t *cur = (s & 1) ? stage1 : stage0;
        uint8_t *nxt = (s & 1) ? stage0 : stage1;
        uint32_t sw_nxt[C::TN], sx_nxt[C::TM];
        if (s + 1 < nslab) {
            stage_tile<WEL, C::BN, C::NT>(W, n0, N, w_rowbytes, (size_t)(s + 1) * SW::GBYTES, nxt);
            stage_tile<XEL, C::BM, C::NT>(X, m0, M, x_rowbytes, (size_t)(s + 1) * SX::GBYTES, nxt + C::W_TILE_BYTES);
#pragma unroll
            for (int t = 0; t < C::TN; ++t)
                sw_nxt[t] = *reinterpret_cast<const uint32_t *>(SFW + sfw_off[t] + (size_t)(s + 1) * 512u);
#pragma unroll
            for (int t = 0; t < C::TM; ++t)
                sx_nxt[t] = *reinterpret_cast<const uint32_t *>(SFX + sfx_off[t] + (size_t)(s + 1) * 512u);
        }
        const uint8_t *wt = cur, *xt = cur + C::W_TILE_BYTES;
        // lane's scale byte for K block (2*h + kb): shift once by 8*kb, then op_sel 0 / 2 picks h.
        int swv[C::TN], sxv[C::TM];
#pragma unroll
        for (int t = 0; t < C::TN; ++t) swv[t] = (int)(sw_cur[t] >> (8 * kb));
#pragma unroll
        for (int t = 0; t < C::TM; ++t) sxv[t] = (int)(sx_cur[t] >> (8 * kb));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v8i wf[C::TN], xf[C::TM];
#pragma unroll
            for (int t = 0; t < C::TN; ++t) wf[t] = load_frag<WEL>(wt, (wn * C::TN + t) * 32 + li, h, kb);
#pragma unroll
            for (int t = 0; t < C::TM; ++t) xf[t] = load_frag<XEL>(xt, (wm * C::TM + t) * 32 + li, h, kb);
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm) {
                    if (h == 0)
                        acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            wf[tn], xf[tm], acc[tn][tm], ElemTraits<WEL>::HW, ElemTraits<XEL>::HW, 0, swv[tn], 0, sxv[tm]);
                    else
                        acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            wf[tn], xf[tm], acc[tn][tm], ElemTraits<WEL>::HW, ElemTraits<XEL>::HW, 2, swv[tn], 2, sxv[tm]);
                }
        }
        if (s + 1 < nslab) {
#pragma unroll
            for (int t = 0; t < C::TN; ++t) sw_cur[t] = sw_nxt[t];
#pragma unroll
            for (int t = 0; t < C::TM; ++t) sx_cur[t] = sx_nxt[t];
        }
        __syncthreads();  // drains the LDS-DMA of slab s+1 (vmcnt(0)) and fences reads of slab s
    }
}

template <class C>
__device__ __forceinline__ void round_acc_bf16(v16f (&acc)[C::TN][C::TM]) {
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tn][tm][i] = bf16_bits_to_f32(f32_to_bf16_bits(acc[tn][tm][i]));
}

template <class C, bool W4>
__global__ void __launch_bounds__(C::NT) mx_gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tiles_n = (a.N + C::BN - 1) / C::BN;
    const int tile = blockIdx.x;
    const int n0 = (tile % tiles_n) * C::BN;
    const int m0 = (tile / tiles_n) * C::BM;

    v16f acc[C::TN][C::TM];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tn][tm][i] = 0.0f;

    bool any = false;
    if (a.K[0]) {
        run_segment<C, EL_FP4, EL_FP4>(acc, a.X[0], a.W[0], a.SFX[0], a.SFW[0], a.K[0], a.M, a.N, m0, n0,
                                       a.sfx_row_tiles, a.sfw_row_tiles, smem);
        any = true;
    }
    if (a.K[1]) {
        if (any && a.round_per_segment) round_acc_bf16<C>(acc);
        run_segment<C, EL_FP6, (W4 ? EL_FP4 : EL_FP6)>(acc, a.X[1], a.W[1], a.SFX[1], a.SFW[1], a.K[1], a.M, a.N, m0, n0,
                                                       a.sfx_row_tiles, a.sfw_row_tiles, smem);
        any = true;
    }
    if (a.K[2]) {
        if (any && a.round_per_segment) round_acc_bf16<C>(acc);
        run_segment<C, EL_FP8, (W4 ? EL_FP4 : EL_FP8)>(acc, a.X[2], a.W[2], a.SFX[2], a.SFW[2], a.K[2], a.M, a.N, m0, n0,
                                                       a.sfx_row_tiles, a.sfw_row_tiles, smem);
    }

    // epilogue.  32x32 accumulator: lane -> column (token) l & 31; register i -> row (feature)
    // (i & 3) + 8 * (i >> 2) + 4 * (l >> 5).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave / C::WAVES_M, wm = wave % C::WAVES_M;
    const int li = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int m = m0 + (wm * C::TM + tm) * 32 + li;
        if (m >= a.M) continue;
        uint16_t *drow = a.D + (size_t)m * a.N;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int nb = n0 + (wn * C::TN + tn) * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g;
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t b = f32_to_bf16_bits(acc[tn][tm][4 * g + i]);
                    if (a.bias != nullptr && n + i < a.N)
                        b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n + i]));
                    o[i] = b;
                }
                if (n + 3 < a.N) {
                    *reinterpret_cast<uint2 *>(drow + n) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (n + i < a.N) drow[n + i] = (uint16_t)o[i];
                }
            }
        }
    }
}

template <class C>
static hipError_t launch_cfg(const GemmArgs &a, bool w4, hipStream_t stream) {
    const int tiles_n = (a.N + C::BN - 1) / C::BN, tiles_m = (a.M + C::BM - 1) / C::BM;
    auto kern = w4 ? mx_gemm_kernel<C, true> : mx_gemm_kernel<C, false>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[w4 ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done[w4 ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_n * tiles_m), dim3(C::NT), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_mx_gemm(const GemmArgs &a, bool w4, hipStream_t stream) {
    if (a.M == 0 || a.N == 0) return hipSuccess;
    if (a.M > 128) return launch_mx_gemm256(a, w4, stream);     // large-M path (mx_gemm256.hip)
    if (a.M <= 64) return launch_mx_gemm_skinny(a, w4, stream);  // decode / small batch (mx_gemm_skinny.hip)
    return launch_cfg<Cfg<2, 2, 2, 2>>(a, w4, stream);
}

}  // namespace mm
