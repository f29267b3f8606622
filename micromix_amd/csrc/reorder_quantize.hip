// Fused column-reorder + per-32-group absmax + UE8M0 scale + MXFP4/MXFP6/MXFP8
// quantize + pack for gfx950.
//
// What it computes is the reference's reorder_quantize_mixed_kernel
// (mgemm/src/reorder.cu:94-269) and reorder_quantize_mxfp4_kernel (:271-432):
// for every row and every 32-wide group g of *reordered* columns,
//   v[i] = row[idx[32g+i]];  amax = max|v|;  e = ceil(log2(amax/FMAX)) (amax==0 -> -1);
//   q[i] = RNE_fmt(v[i] * 2^-e);  SF byte = e + 127 at sf_offset(row, g - seg_start/32).
//
// How it is laid out for MI355X (HBM-bound: 2*K bytes in, ~K/2..K bytes out per row):
//  * one workgroup owns a strided set of rows; thread t owns group t for every one
//    of those rows, so its 32 reorder indices live in 16 VGPRs for the whole launch
//    (the reference re-reads each index from global memory per element);
//  * the row is staged in LDS with 16-byte coalesced loads; the next row's loads are
//    issued before the current row's gather so HBM latency hides under the LDS work;
//  * all scale arithmetic is integer/exponent arithmetic (no log2/ceil/ldexp/divide); the element conversion is
//    one hardware MX-converter instruction per 2 (fp4/fp8) or 32 (fp6) elements.
#include "mx_common.h"
#include "mx_group_convert.h"
#include "mx_instrument.h"
#include "mx_kernels.h"

namespace mm {

#if MM_CLOCKS
// instrumented variant only: per workgroup {start, row staged, group quantized and stored, end} in 100 MHz ticks (tools/quant_clock.py)
__device__ unsigned long long *g_quant_clock = nullptr;
#endif

// rows first_row, first_row + row_stride, ... of one [rows, K] matrix (one workgroup's share)
template <bool W4>
__device__ __forceinline__ void reorder_quantize_body(const uint16_t *__restrict__ src, int rows, int K, const int16_t *__restrict__ idx,
                                                      int KN, int KS, int KO, uint8_t *__restrict__ oN, uint8_t *__restrict__ oS,
                                                      uint8_t *__restrict__ oO, uint8_t *__restrict__ sfN, uint8_t *__restrict__ sfS,
                                                      uint8_t *__restrict__ sfO, int first_row, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int G = (KN + KS + KO) >> 5;  // groups produced; K is the input row length (>= 32 * G)
    const int g = threadIdx.x;
    const bool active = g < G;
    const int nchunk = K >> 3;  // 16-byte chunks per row
    // fp6 / fp8 codes go through an image of the row's [S | O] codes behind the staged row (store_code_image, mx_group_convert.h): 11.3 -> 9.3 us
    const int bytesS = KS / 4 * 3, bytesO = KO;
    uint8_t *image = smem + (size_t)K * 2;

    uint32_t ix[16];
    if (active) {
        const uint4 *p = reinterpret_cast<const uint4 *>(idx + (size_t)g * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = p[i];
            // keep BYTE offsets (index << 1; indices are < 32768 so each half stays within 16 bits)
            ix[4 * i] = swizzle_offsets((q.x << 1) & 0xFFFEFFFEu);
            ix[4 * i + 1] = swizzle_offsets((q.y << 1) & 0xFFFEFFFEu);
            ix[4 * i + 2] = swizzle_offsets((q.z << 1) & 0xFFFEFFFEu);
            ix[4 * i + 3] = swizzle_offsets((q.w << 1) & 0xFFFEFFFEu);
        }
    }
    // Which segment this thread's group falls in (positions in reordered order).
    const int gN = KN >> 5, gS = KS >> 5;
    int seg, j, kseg;
    if (g < gN) { seg = 0; j = g; kseg = KN; }
    else if (g < gN + gS) { seg = 1; j = g - gN; kseg = KS; }
    else { seg = 2; j = g - gN - gS; kseg = KO; }

#if MM_CLOCKS
    unsigned long long *ck = g_quant_clock != nullptr ? g_quant_clock + 4 * (size_t)blockIdx.x : nullptr;
    if (ck != nullptr && threadIdx.x == 0) ck[0] = __builtin_amdgcn_s_memrealtime();
#endif
    uint4 stage[4];
    int r = first_row;
    if (r < rows) {
        const uint4 *grow = reinterpret_cast<const uint4 *>(src + (size_t)r * K);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
            if (c < nchunk) t = grow[c];
            stage[i] = t;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) reinterpret_cast<uint4 *>(smem)[swizzle_chunk(c)] = stage[i];
        }
    }
    __syncthreads();
#if MM_CLOCKS
    if (ck != nullptr && threadIdx.x == 0) ck[1] = __builtin_amdgcn_s_memrealtime();
#endif
    for (; r < rows; r += row_stride) {
        const int rn = r + row_stride;
        if (rn < rows) {
            const uint4 *grow = reinterpret_cast<const uint4 *>(src + (size_t)rn * K);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = threadIdx.x + i * blockDim.x;
                uint4 t = make_uint4(0u, 0u, 0u, 0u);
                if (c < nchunk) t = grow[c];
                stage[i] = t;
            }
        }
        if (active) {
            const uint8_t *row = smem;
            uint32_t byte;
            uint8_t *sf;
            uint32_t v[16];
            const uint32_t amax = gather_group(row, ix, v);   // once per wave, whatever segments its lanes are in
            if (seg == 0) {
                byte = finish_group<EL_FP4, true>(v, amax, oN + (size_t)r * (KN >> 1) + j * 16);
                sf = sfN;
            } else if (seg == 1) {
                if constexpr (W4) byte = finish_group<EL_FP4, true>(v, amax, oS + (size_t)r * (KS >> 1) + j * 16);
                else byte = finish_group<EL_FP6>(v, amax, image + j * 24);                 // into the LDS image (see above)
                sf = sfS;
            } else {
                if constexpr (W4) byte = finish_group<EL_FP4, true>(v, amax, oO + (size_t)r * (KO >> 1) + j * 16);
                else byte = finish_group<EL_FP8>(v, amax, image + bytesS + j * 32);
                sf = sfO;
            }
            // the 4 block scales of one row and one 128-column slab are 4 consecutive bytes of the SF layout: gather them
            // from the lane quad (segment widths are multiples of 128, so a quad never straddles segments) and store a dword
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            if ((g & 3) == 0)
                store_scale_dword(sf + sf_offset(r, j, kseg), byte | (b1 << 8) | (b2 << 16) | (b3 << 24));
        }
#if MM_CLOCKS
        if (ck != nullptr && threadIdx.x == 0 && r == first_row) ck[2] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();
        if constexpr (!W4) store_code_image(image, bytesS, bytesO, oS, oO, r);
        if (rn < rows) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = threadIdx.x + i * blockDim.x;
                if (c < nchunk) reinterpret_cast<uint4 *>(smem)[swizzle_chunk(c)] = stage[i];
            }
        }
        __syncthreads();
    }
#if MM_CLOCKS
    if (ck != nullptr && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ck[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <bool W4, int MAXT>
__global__ void __launch_bounds__(MAXT)
reorder_quantize_kernel(const uint16_t *__restrict__ src, int rows, int K, const int16_t *__restrict__ idx, int KN,
                        int KS, int KO, uint8_t *__restrict__ oN, uint8_t *__restrict__ oS, uint8_t *__restrict__ oO,
                        uint8_t *__restrict__ sfN, uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    reorder_quantize_body<W4>(src, rows, K, idx, KN, KS, KO, oN, oS, oO, sfN, sfS, sfO, blockIdx.x, gridDim.x);
}

// Grouped launch (MoE experts: every expert quantizes its own token rows with its own reorder index; reference: the
// per-expert reorder_quantize_x calls of qMixtralLayer.py:507-519): blockIdx.y picks one of up to MM_MAX_GROUPS argument blocks.
template <bool W4, int MAXT>
__global__ void __launch_bounds__(MAXT) reorder_quantize_grouped_kernel(GroupedQuantArgs ga) {
    const QuantArgs &q = ga.g[blockIdx.y];
    if ((int)blockIdx.x < q.rows)
        reorder_quantize_body<W4>(q.src, q.rows, ga.K, q.idx, ga.KN, ga.KS, ga.KO, q.o[0], q.o[1], q.o[2], q.sf[0], q.sf[1], q.sf[2],
                                  blockIdx.x, gridDim.x);
}

#if MM_CLOCKS
hipError_t set_quant_clock_buffer(unsigned long long *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_quant_clock), &buf, sizeof(buf)); }
#endif

hipError_t launch_reorder_quantize(const void *src, int rows, int K, const int16_t *idx, int KN, int KS, int KO, bool w4,
                                   uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                                   hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const int G = (KN + KS + KO) / 32;
    const int stagers = K / 32;  // a row of K bf16 is staged as K/8 16-byte chunks, at most 4 per thread
    const int threads = ((G > stagers ? G : stagers) + 63) / 64 * 64;
    const size_t lds = (size_t)K * 2 + (w4 ? 0 : (size_t)KS / 4 * 3 + KO);   // the staged row + the image of its fp6 / fp8 codes
    auto kern = threads <= 256 ? (w4 ? reorder_quantize_kernel<true, 256> : reorder_quantize_kernel<false, 256>)
                               : (w4 ? reorder_quantize_kernel<true, 1024> : reorder_quantize_kernel<false, 1024>);
    // K > 21845: the row and the image together pass the default 64 KiB limit of dynamic LDS (at most 96 KiB, K = 32768)
    static DynamicLdsOnce big_lds;
    if (lds > 48 * 1024)
        if (hipError_t e = big_lds.ensure(reinterpret_cast<const void *>(reorder_quantize_kernel<false, 1024>), 96 * 1024); e != hipSuccess) return e;
    // one resident wave of workgroups (no tail), each striding over the rows
    const int per_cu = OccupancyCache::get(w4 ? 0 : 1, reinterpret_cast<const void *>(kern), threads, lds);
    const int cus = device_cus();
    int blocks = cus * per_cu;
    // (fewer, fatter workgroups that keep their reorder indices for 2 / 4 / 8 rows were measured: 11.1 -> 11.7 / 15.6 / 23.9 us;
    // one row per workgroup with the occupancy limited to 12 / 8 / 4 workgroups per CU, so that rounds of workgroups overlap
    // their load / gather / store phases: 11.3 / 11.4 / 13.7 us)
    blocks = rows < blocks ? rows : blocks;
    MM_LAUNCH(kern, dim3(blocks), dim3(threads), lds, stream, (const uint16_t *)src, rows, K, idx, KN, KS, KO, oN,
                       oS, oO, sfN, sfS, sfO);
    return hipGetLastError();
}

hipError_t launch_reorder_quantize_grouped(const GroupedQuantArgs &ga, int max_rows, bool w4, hipStream_t stream) {
    if (ga.ngroups < 1 || ga.ngroups > MM_MAX_GROUPS || max_rows < 1) return hipErrorInvalidValue;
    const int G = (ga.KN + ga.KS + ga.KO) / 32, stagers = ga.K / 32;
    const int threads = ((G > stagers ? G : stagers) + 63) / 64 * 64;
    const size_t lds = (size_t)ga.K * 2 + (w4 ? 0 : (size_t)ga.KS / 4 * 3 + ga.KO);
    auto kern = threads <= 256 ? (w4 ? reorder_quantize_grouped_kernel<true, 256> : reorder_quantize_grouped_kernel<false, 256>)
                               : (w4 ? reorder_quantize_grouped_kernel<true, 1024> : reorder_quantize_grouped_kernel<false, 1024>);
    static DynamicLdsOnce big_lds;
    if (lds > 48 * 1024)
        if (hipError_t e = big_lds.ensure(reinterpret_cast<const void *>(reorder_quantize_grouped_kernel<false, 1024>), 96 * 1024); e != hipSuccess) return e;
    const int per_cu = OccupancyCache::get(w4 ? 2 : 3, reinterpret_cast<const void *>(kern), threads, lds);
    const int cus = device_cus();
    int bx = cus * per_cu / ga.ngroups;   // one resident wave of workgroups shared by the groups
    bx = bx < 1 ? 1 : bx;
    bx = max_rows < bx ? max_rows : bx;
    hipLaunchKernelGGL(kern, dim3(bx, ga.ngroups), dim3(threads), lds, stream, ga);
    return hipGetLastError();
}

}  // namespace mm
