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
// MI355X mapping: pure streaming (HBM-bound).  A wave takes 64 consecutive groups (4 KiB per operand) per step:
//   1. it reads them as four instructions of 16 bytes per lane SIDE BY SIDE (1 KiB contiguous each) and turns them over in a
//      wave-private 4 KiB of LDS (16-byte chunks, swizzled so that neither side conflicts), after which every lane holds one whole
//      group -- 32 elements -- as the MX converters want them (v_cvt_scalef32_pk_fp4_f32 / _pk_fp8_f32 / _2xpk16_bf6_f32);
//   2. fp4 codes (16 bytes per lane, side by side) leave straight from the lanes, write-through;
//   3. fp6 / fp8 codes (24 / 32 bytes per lane) go back into the same LDS (32-byte slot per lane) and leave as 8 / 16 bytes per lane
//      side by side, write-through, each piece finding its own row and segment offset;
//   4. the scale bytes of a lane quad are merged into one dword store.
// Rounds 1-2 loaded 64 contiguous bytes per lane, so each load instruction covered 16 of every 64 bytes -- a quarter of each
// line -- and each fp8 store instruction half of each line.  Measured at M = 4096, K = 14336 (tools/time_activate.py,
// profiles/notes_r03.md section 17), (12288,1024,1024) / all-fp8 / all-fp4: 51.4 / 66.2 / 51.8 us; loads side by side (results
// wrong) 41.9 / 46.6 / 41.1; a first rewrite with a lane quad per group (eight elements per lane, 4- and 8-byte stores side by
// side, fp6 through the old path) 45.4 / 52.8 / 46.8, of which the small stores were 4 ... 12 us.
#include "mx_common.h"
#include "mx_direct_convert.h"
#include "mx_kernels.h"

namespace mm {

// write-through store of 8 bytes (whole lines per wave instruction when the lanes' pieces lie side by side)
__device__ __forceinline__ void store8_wt(uint8_t *out, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u vv = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out), "v"(vv) : "memory");
#endif
}

// 16-byte chunk i of group m (of the wave's 64) lives in slot 4 m + (i ^ ((m >> 2) & 3)): the writers (lane 4 q + c writes chunk c of
// group 16 k + q) fill whole 64-byte runs, the readers (lane m reads chunk i of group m, a 64-byte stride) spread over all banks
__device__ __forceinline__ int turn_slot(int m, int i) { return 4 * m + (i ^ ((m >> 2) & 3)); }

// row / group of the wave's group number m (row r0, group g0 = the wave's first)
__device__ __forceinline__ void locate(int r0, int g0, int m, int G, int &r, int &g) {
    g = g0 + m;
    r = r0;
    if (g >= G) {
        const unsigned q = (unsigned)g / (unsigned)G;
        r += (int)q;
        g -= (int)(q * (unsigned)G);
    }
}

// MODE 0: silu(A) * B -> fp4|fp6|fp8;  MODE 1: A -> fp4|fp6|fp8;  MODE 2: A -> fp4|fp4|fp4
// INTER (MODE 0 only): A and B are the two halves of ONE row-major [rows, 2 K] matrix whose columns alternate between 128 columns of
// A and 128 columns of B -- the output of a GEMM over gate / up weight rows interleaved per 128 features (mm_gate_up_activate's
// small-M path): group g of row r lies at element r * 2 K + (g / 4) * 256 + (g % 4) * 32 of A, and 128 elements further in B.
template <int MODE, bool INTER = false>
__global__ void __launch_bounds__(256)
direct_quantize_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B, int rows, int KN, int KS, int KO,
                       uint8_t *__restrict__ oN, uint8_t *__restrict__ oS, uint8_t *__restrict__ oO,
                       uint8_t *__restrict__ sfN, uint8_t *__restrict__ sfS, uint8_t *__restrict__ sfO) {
    __shared__ uint4 turn[4][MODE == 0 ? 2 : 1][256];   // per wave: 4 KiB per operand
    const int K = KN + KS + KO, G = K >> 5, gN = KN >> 5, gS = KS >> 5;
    const long long total = (long long)rows * G;      // groups, 64 bytes of input each, consecutive in memory
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 *ta = turn[wave][0], *tb = turn[wave][MODE == 0 ? 1 : 0];
    uint8_t *image = reinterpret_cast<uint8_t *>(ta);   // the fp6 / fp8 codes reuse operand A's 4 KiB: 32 bytes per lane
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long base = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)); base < total; base += step) {
        // 1. four instructions of 16 bytes per lane side by side; instruction k brings chunk (lane & 3) of group 16 k + (lane >> 2)
        {
            const uint4 *pa = reinterpret_cast<const uint4 *>(A) + base * 4 + lane;
            const uint4 *pb = reinterpret_cast<const uint4 *>(B) + base * 4 + lane;
            uint4 xa[4], xb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long gi = base + 16 * k + (lane >> 2);
                const bool valid = gi < total;
                if constexpr (INTER) {
                    const long long r = gi / G;
                    const int g = (int)(gi - r * G);
                    const size_t at = ((size_t)r * (size_t)(2 * K) + (size_t)(g >> 2) * 256u + (size_t)(g & 3) * 32u) / 8u + (lane & 3);
                    xa[k] = valid ? reinterpret_cast<const uint4 *>(A)[at] : make_uint4(0u, 0u, 0u, 0u);
                    xb[k] = valid ? reinterpret_cast<const uint4 *>(B)[at] : make_uint4(0u, 0u, 0u, 0u);
                } else {
                    xa[k] = valid ? pa[64 * k] : make_uint4(0u, 0u, 0u, 0u);
                    if constexpr (MODE == 0) xb[k] = valid ? pb[64 * k] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int slot = turn_slot(16 * k + (lane >> 2), lane & 3);
                ta[slot] = xa[k];
                if constexpr (MODE == 0) tb[slot] = xb[k];
            }
        }
        __builtin_amdgcn_wave_barrier();   // (one wave: its LDS instructions complete in order; this only stops the compiler)
        // row and group of the wave's first group (wave-uniform; 32-bit arithmetic whenever the group count allows it)
        int r0, g0;
        if (total < 0x7FFFFFFFll) { r0 = (int)((unsigned)base / (unsigned)G); g0 = (int)((unsigned)base - (unsigned)r0 * (unsigned)G); }
        else { r0 = (int)(base / G); g0 = (int)(base - (long long)r0 * G); }
        // 2. one whole group per lane
        if (base + lane < total) {
            float v[32];
#pragma unroll
            for (int i = 0; i < 4; ++i) unpack8(ta[turn_slot(lane, i)], v + 8 * i);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float b[8];
                    unpack8(tb[turn_slot(lane, i)], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[8 * i + e] = silu_mul(v[8 * i + e], b[e]);     // (mx_direct_convert.h)
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // every lane has read its operands before a lane's codes overwrite operand A's slots
            int r, g;
            locate(r0, g0, lane, G, r, g);
            uint32_t byte;
            uint8_t *sf;
            int j, kseg;
            if (g < gN) {
                j = g; kseg = KN; sf = sfN;
                byte = quantize32<EL_FP4>(v, oN + (size_t)r * (KN >> 1) + j * 16);
            } else if (g < gN + gS) {
                j = g - gN; kseg = KS; sf = sfS;
                if constexpr (MODE == 2) byte = quantize32<EL_FP4>(v, oS + (size_t)r * (KS >> 1) + j * 16);
                else byte = quantize32<EL_FP6>(v, image + lane * 32);
            } else {
                j = g - gN - gS; kseg = KO; sf = sfO;
                if constexpr (MODE == 2) byte = quantize32<EL_FP4>(v, oO + (size_t)r * (KO >> 1) + j * 16);
                else byte = quantize32<EL_FP8>(v, image + lane * 32);
            }
            // G and the segment widths are multiples of 4 groups: a lane quad always holds 4 consecutive blocks of one row
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0x55, 0xF, 0xF, false);
            const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xAA, 0xF, 0xF, false);
            const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)byte, 0xFF, 0xF, 0xF, false);
            // (a plain store: write-through, which pays in the row-per-workgroup quantizers -- store_scale_dword -- costs 3 us of 41 here)
            if ((g & 3) == 0) *reinterpret_cast<uint32_t *>(sf + sf_offset(r, j, kseg)) = byte | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        if constexpr (MODE != 2) {
            __builtin_amdgcn_wave_barrier();
            // 3. the codes in the image leave side by side: fp8 as 16-byte pieces (piece t = half t & 1 of group t >> 1), fp6 as 8-byte
            //    pieces (piece t = third t % 3 of group t / 3); a piece whose group has another format (or lies past the end) is skipped
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = lane + 64 * i, m = t >> 1;
                int r, g;
                locate(r0, g0, m, G, r, g);
                if (base + m < total && g >= gN + gS) {
                    const uint4 c = *reinterpret_cast<const uint4 *>(image + m * 32 + (t & 1) * 16);
                    store16<true>(oO + (size_t)r * KO + (size_t)(g - gN - gS) * 32 + (t & 1) * 16, c.x, c.y, c.z, c.w);
                }
            }
            if (gS > 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int t = lane + 64 * i, m = t / 3, p = t - 3 * m;
                    int r, g;
                    locate(r0, g0, m, G, r, g);
                    if (base + m < total && g >= gN && g < gN + gS) {
                        const uint2 c = *reinterpret_cast<const uint2 *>(image + m * 32 + p * 8);
                        store8_wt(oS + (size_t)r * (KS / 4 * 3) + (size_t)(g - gN) * 24 + p * 8, c.x, c.y);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the image is read before the next step's operands overwrite it
        }
    }
}

hipError_t launch_direct_quantize(const void *A, const void *B, int rows, int KN, int KS, int KO, int mode, uint8_t *oN,
                                  uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    const long long total = (long long)rows * ((KN + KS + KO) / 32);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // mode 3 = mode 0 on the interleaved [rows, 2 K] layout (A = the matrix, B = A + 128 elements; the caller passes both)
    auto kern = mode == 0 ? direct_quantize_kernel<0> : mode == 1 ? direct_quantize_kernel<1> : mode == 2 ? direct_quantize_kernel<2>
                                                                                                           : direct_quantize_kernel<0, true>;
    MM_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint16_t *)A, (const uint16_t *)B, rows, KN, KS, KO,
                       oN, oS, oO, sfN, sfS, sfO);
    return hipGetLastError();
}

}  // namespace mm
