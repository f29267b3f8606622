// libmicromix_diag.so (include/micromix_diag.h) -- NOT linked into the product library.
// Hardware diagnostics exported through a C ABI (mm_diag_*): single scaled-MFMA issue and the
// CDNA4 MX converter instructions, so that the tests can pin (a) the operand/scale/accumulator
// register layouts the GEMM kernel relies on and (b) the oracle's element encoders against
// AMD's own hardware implementation of the OCP MX formats.  Not on the product path.
#include <hip/hip_runtime.h>
#include "../../include/micromix_diag.h"
#include "../../include/micromix_hip.h"   // status codes only
#include "mx_common.h"

namespace mm {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FA, int FB, int OPSEL>
__global__ void diag_mfma32(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *out) {
    const int l = threadIdx.x;
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, OPSEL, sa[l], OPSEL, sb[l]);
    out[l] = acc;
}

template <int FA, int FB, int OPSEL>
__global__ void diag_mfma16(const v8i *a, const v8i *b, const int *sa, const int *sb, v4f *out) {
    const int l = threadIdx.x;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, FA, FB, OPSEL, sa[l], OPSEL, sb[l]);
    out[l] = acc;
}

template <int FA, int FB, int OPSEL>
static void launch_mfma(int shape, const void *a, const void *b, const void *sa, const void *sb, void *out, hipStream_t s) {
    if (shape == 32)
        hipLaunchKernelGGL((diag_mfma32<FA, FB, OPSEL>), dim3(1), dim3(64), 0, s, (const v8i *)a, (const v8i *)b,
                           (const int *)sa, (const int *)sb, (v16f *)out);
    else
        hipLaunchKernelGGL((diag_mfma16<FA, FB, OPSEL>), dim3(1), dim3(64), 0, s, (const v8i *)a, (const v8i *)b,
                           (const int *)sa, (const int *)sb, (v4f *)out);
}

template <int FA, int FB>
static void dispatch_opsel(int shape, int opsel, const void *a, const void *b, const void *sa, const void *sb, void *out,
                           hipStream_t s) {
    switch (opsel) {
        case 0: launch_mfma<FA, FB, 0>(shape, a, b, sa, sb, out, s); break;
        case 1: launch_mfma<FA, FB, 1>(shape, a, b, sa, sb, out, s); break;
        case 2: launch_mfma<FA, FB, 2>(shape, a, b, sa, sb, out, s); break;
        default: launch_mfma<FA, FB, 3>(shape, a, b, sa, sb, out, s); break;
    }
}

template <int FA>
static void dispatch_fb(int fb, int shape, int opsel, const void *a, const void *b, const void *sa, const void *sb,
                        void *out, hipStream_t s) {
    switch (fb) {
        case EL_FP4: dispatch_opsel<FA, HW_FP4>(shape, opsel, a, b, sa, sb, out, s); break;
        case EL_FP6: dispatch_opsel<FA, HW_BF6>(shape, opsel, a, b, sa, sb, out, s); break;
        default: dispatch_opsel<FA, HW_FP8>(shape, opsel, a, b, sa, sb, out, s); break;
    }
}

typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef __bf16 v32bf __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef short v2s __attribute__((ext_vector_type(2)));

// one thread converts one 32-element group with the hardware MX converters; writes one code per byte
__global__ void diag_hw_convert(const uint16_t *src, int ngroups, float scale, int el, uint8_t *out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint16_t *p = src + (size_t)g * 32;
    uint8_t *o = out + (size_t)g * 32;
    if (el == EL_FP4) {
        for (int i = 0; i < 32; i += 8) {
            unsigned w = 0;
            v2bf x0, x1, x2, x3;
            __builtin_memcpy(&x0, p + i, 4);
            __builtin_memcpy(&x1, p + i + 2, 4);
            __builtin_memcpy(&x2, p + i + 4, 4);
            __builtin_memcpy(&x3, p + i + 6, 4);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(w, x0, scale, 0);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(w, x1, scale, 1);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(w, x2, scale, 2);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(w, x3, scale, 3);
            for (int k = 0; k < 8; ++k) o[i + k] = (w >> (4 * k)) & 0xF;
        }
    } else if (el == EL_FP8) {
        for (int i = 0; i < 32; i += 4) {
            v2s w = {0, 0};
            v2bf x0, x1;
            __builtin_memcpy(&x0, p + i, 4);
            __builtin_memcpy(&x1, p + i + 2, 4);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, x0, scale, false);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, x1, scale, true);
            unsigned u;
            __builtin_memcpy(&u, &w, 4);
            for (int k = 0; k < 4; ++k) o[i + k] = (u >> (8 * k)) & 0xFF;
        }
    } else {
        v32bf x;
        __builtin_memcpy(&x, p, 64);
        v6u w = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(x, scale);
        unsigned long long lo = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
        unsigned long long mi = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
        unsigned long long hi = (unsigned long long)w[4] | ((unsigned long long)w[5] << 32);
        unsigned long long ws[3] = {lo, mi, hi};
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i, word = bit >> 6, off = bit & 63;
            unsigned long long v = ws[word] >> off;
            if (off > 58) v |= ws[word + 1] << (64 - off);
            o[i] = (uint8_t)(v & 0x3F);
        }
    }
}

// Register-only issue-rate loop: every wave issues `iters` x 8 independent scaled MFMAs.
template <int FA, int FB, int SHAPE>
__global__ void __launch_bounds__(256) diag_mfma_rate(const v8i *seed, int iters, float *sink) {
    const int l = threadIdx.x & 63;
    v8i a = seed[l], b = seed[64 + l];
    const int sa = 127, sb = 127;
    if constexpr (SHAPE == 32) {
        v16f acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], FA, FB, 0, sa, 0, sb);
        }
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += acc[j][0] + acc[j][15];
        if (t == 12345.678f) sink[0] = t;
    } else {
        v4f acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[j], FA, FB, 0, sa, 0, sb);
        }
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += acc[j][0] + acc[j][3];
        if (t == 12345.678f) sink[0] = t;
    }
}

template <int FA, int FB>
static void launch_rate(int shape, int blocks, const void *seed, int iters, float *sink, hipStream_t s) {
    if (shape == 32)
        hipLaunchKernelGGL((diag_mfma_rate<FA, FB, 32>), dim3(blocks), dim3(256), 0, s, (const v8i *)seed, iters, sink);
    else
        hipLaunchKernelGGL((diag_mfma_rate<FA, FB, 16>), dim3(blocks), dim3(256), 0, s, (const v8i *)seed, iters, sink);
}

template <int FA>
static void rate_fb(int fb, int shape, int blocks, const void *seed, int iters, float *sink, hipStream_t s) {
    switch (fb) {
        case EL_FP4: launch_rate<FA, HW_FP4>(shape, blocks, seed, iters, sink, s); break;
        case EL_FP6: launch_rate<FA, HW_BF6>(shape, blocks, seed, iters, sink, s); break;
        default: launch_rate<FA, HW_FP8>(shape, blocks, seed, iters, sink, s); break;
    }
}

}  // namespace mm

extern "C" {

int mm_diag_mfma_rate(int shape, int el_a, int el_b, int blocks, int iters, const void *seed_regs, void *sink,
                      mm_stream_t stream) {
    if ((shape != 32 && shape != 16) || el_a < 0 || el_a > 2 || el_b < 0 || el_b > 2 || blocks <= 0 || iters <= 0)
        return MM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (el_a) {
        case mm::EL_FP4: mm::rate_fb<mm::HW_FP4>(el_b, shape, blocks, seed_regs, iters, (float *)sink, s); break;
        case mm::EL_FP6: mm::rate_fb<mm::HW_BF6>(el_b, shape, blocks, seed_regs, iters, (float *)sink, s); break;
        default: mm::rate_fb<mm::HW_FP8>(el_b, shape, blocks, seed_regs, iters, (float *)sink, s); break;
    }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

int mm_diag_mfma(int shape, int el_a, int el_b, int opsel, const void *a_regs, const void *b_regs, const void *scale_a,
                 const void *scale_b, void *out, mm_stream_t stream) {
    if ((shape != 32 && shape != 16) || el_a < 0 || el_a > 2 || el_b < 0 || el_b > 2 || opsel < 0 || opsel > 3)
        return MM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (el_a) {
        case mm::EL_FP4: mm::dispatch_fb<mm::HW_FP4>(el_b, shape, opsel, a_regs, b_regs, scale_a, scale_b, out, s); break;
        case mm::EL_FP6: mm::dispatch_fb<mm::HW_BF6>(el_b, shape, opsel, a_regs, b_regs, scale_a, scale_b, out, s); break;
        default: mm::dispatch_fb<mm::HW_FP8>(el_b, shape, opsel, a_regs, b_regs, scale_a, scale_b, out, s); break;
    }
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

int mm_diag_hw_convert(const void *src_bf16, int n, float scale, int el, uint8_t *out_codes, mm_stream_t stream) {
    if (n < 0 || (n % 32) || el < 0 || el > 2) return MM_ERR_BAD_ARG;
    if (n == 0) return MM_OK;
    const int groups = n / 32;
    hipLaunchKernelGGL(mm::diag_hw_convert, dim3((groups + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t *)src_bf16, groups, scale, el, out_codes);
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}
}

// ---------------------------------------------------------------------------------------------------------
// L2 -> CU read-bandwidth microbenchmark (kernel-developer tool, not on the product path).
//   mode 0: global_load_dwordx4 to registers, 1 KiB contiguous per wave instruction
//   mode 1: LDS-DMA, 1 KiB contiguous per wave instruction
//   mode 2: LDS-DMA, 8 rows x 128 B per wave instruction, rows `stride` bytes apart (the GEMM's operand pattern)
// Every workgroup (256 threads) moves `kb_per_iter` KiB per iteration from a hot region of `region` bytes.
// ---------------------------------------------------------------------------------------------------------
namespace mm {
typedef int rsrc4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) diag_l2_bw(const uint8_t *buf, unsigned region, int stride, int kb_per_iter, int iters,
                                                  int mode, float *sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long v = (unsigned long long)buf;
    rsrc4_t rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xFFFFu));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)region);
    rsrc[3] = 0x00020000;
    const unsigned mask = region - 1;  // region is a power of two
    const unsigned base = (blockIdx.x * 65536u) & mask;
    float acc = 0.0f;
    if (mode == 4 || mode == 5) {
        // register path with a deep queue: every wave keeps 12 x 16 B per lane in flight (mode 5 also stores them to LDS)
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (int)region, 0x00020000);
        for (int it = 0; it < iters; ++it) {
            for (int p0 = 0; p0 < kb_per_iter / 4; p0 += 12) {
                v4i_t q[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const unsigned off = (base + (unsigned)(it & 31) * 49152u + (unsigned)((p0 + j) * 4 + wave) * 1024u + lane * 16u) & mask;
                    q[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    if (mode == 5) *reinterpret_cast<v4i_t *>(smem + ((j * 4 + wave) * 64 + lane) * 16) = q[j];
                    else acc += __int_as_float(q[j][0] ^ q[j][1] ^ q[j][2] ^ q[j][3]);
                }
            }
        }
        if (acc == 12345.678f) sink[0] = acc + smem[tid];
        return;
    }
    for (int it = 0; it < iters; ++it) {
        for (int p = 0; p < kb_per_iter / 4; ++p) {  // each wave moves 1 KiB per instruction, 4 waves
            unsigned off;
            if (mode == 2) {
                const int row = (p * 32 + wave * 8 + (lane >> 3));
                off = (base + (unsigned)row * (unsigned)stride + (unsigned)(it & 31) * 128u + (lane & 7) * 16u) & mask;
            } else {
                off = (base + (unsigned)(it & 31) * 49152u + (unsigned)(p * 4 + wave) * 1024u + lane * 16u) & mask;
            }
            if (mode == 0) {
                const uint4 q = *reinterpret_cast<const uint4 *>(buf + off);
                acc += __uint_as_float(q.x ^ q.y ^ q.z ^ q.w);
            } else {
                const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(((p & 31) * 4 + wave) * 1024));
                unsigned keep;
                asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"((int)off), "s"(rsrc), "s"(lds)
                             : "memory");
            }
        }
        if (mode != 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc + smem[tid];
}

// One pass over `rows` weight rows of `pitch` bytes, 32 rows per 512-thread workgroup, every load of a wave issued before the
// first use (tools/stream_floor.py: what a launch that only streams the weights of a skinny GEMM costs, by access pattern).
//   pattern 0: coalesced -- the workgroup's 32 * pitch bytes as one range, a wave-instruction = 1 KiB contiguous
//   pattern 1: the weight-streaming kernel's -- lane (li, kb) reads 16 B of row li; 64-byte slab s of a row goes to wave s % 8
//   pattern 2: as 1, but a wave takes both 64-byte halves of a 128-byte line (slabs 2w, 2w + 1, 2w + 16, 2w + 17, ...)
template <int NL>
__global__ void __launch_bounds__(512) diag_stream_once(const uint8_t *buf, int pitch, int pattern, float *sink) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kb = lane >> 5;
    const uint8_t *base = buf + (size_t)blockIdx.x * 32 * pitch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 32 * pitch, 0x00020000);
    v4i_t q[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        int off;
        if (pattern == 0) off = (j * 8 + wave) * 1024 + lane * 16;
        else {
            const int pair = j >> 1, h = j & 1;                       // two loads per 64-byte slab
            const int slab = pattern == 1 ? wave + 8 * pair : 2 * wave + (pair & 1) + 16 * (pair >> 1);
            off = li * pitch + slab * 64 + (2 * h + kb) * 16;
        }
        q[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
    int acc = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) acc ^= q[j][0] ^ q[j][1] ^ q[j][2] ^ q[j][3];
    if (acc == 0x12345678) sink[0] = 1.0f;
}
}  // namespace mm

extern "C" int mm_diag_l2_bw(const void *buf, unsigned region, int stride, int kb_per_iter, int iters, int mode, int blocks,
                             void *sink, mm_stream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mm::diag_l2_bw), hipFuncAttributeMaxDynamicSharedMemorySize,
                                131072) != hipSuccess)
            return MM_ERR_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(mm::diag_l2_bw, dim3(blocks), dim3(256), mode >= 4 ? 49152 : 131072, (hipStream_t)stream, (const uint8_t *)buf, region, stride,
                       kb_per_iter, iters, mode, (float *)sink);
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

// rows % 32 == 0; pitch = bytes per row (1024, 2048 or 4096: every wave issues pitch / 256 loads of 16 B per lane, all before the first use)
extern "C" int mm_diag_stream_once(const void *buf, int rows, int pitch, int pattern, void *sink, mm_stream_t stream) {
    if (rows % 32 || (pitch != 1024 && pitch != 2048 && pitch != 4096)) return MM_ERR_BAD_ARG;
    const dim3 grid(rows / 32), block(512);
    if (pitch == 1024) hipLaunchKernelGGL(mm::diag_stream_once<4>, grid, block, 0, (hipStream_t)stream, (const uint8_t *)buf, pitch, pattern, (float *)sink);
    else if (pitch == 2048) hipLaunchKernelGGL(mm::diag_stream_once<8>, grid, block, 0, (hipStream_t)stream, (const uint8_t *)buf, pitch, pattern, (float *)sink);
    else hipLaunchKernelGGL(mm::diag_stream_once<16>, grid, block, 0, (hipStream_t)stream, (const uint8_t *)buf, pitch, pattern, (float *)sink);
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}
