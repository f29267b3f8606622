/*
 * micromix_diag.h -- C ABI of libmicromix_diag.so: hardware probes and microbenchmarks for gfx950.
 *
 * NOT part of the product library (libmicromix_hip.so, include/micromix_hip.h).  The GPU tests use the probes to pin the
 * register layouts the GEMM kernels rely on and the oracle's element encoders against the CDNA4 hardware
 * (tests/test_hw_gpu.py); tools/ uses the microbenchmarks (tools/mfma_rate.py, mfma_energy.py, l2_bw.py).
 */
#ifndef MICROMIX_DIAG_H
#define MICROMIX_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *mm_stream_t; /* hipStream_t */

/*
 *   mm_diag_mfma: one wave issues one v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4.
 *     shape 32|16; el_a/el_b 0=fp4 1=fp6(E3M2) 2=fp8(E4M3); opsel 0..3 applied to both scales;
 *     a_regs/b_regs [64 lanes][8] int32; scale_a/scale_b [64] int32; out [64][16|4] float.
 *   mm_diag_hw_convert: v_cvt_scalef32_pk_{fp4,fp8}_bf16 / pk32_bf6_bf16 on n (multiple of 32)
 *     bf16 values with one scale; out_codes gets one element code per byte.
 */
int mm_diag_mfma(int shape, int el_a, int el_b, int opsel, const void *a_regs, const void *b_regs, const void *scale_a,
                 const void *scale_b, void *out, mm_stream_t stream);
int mm_diag_hw_convert(const void *src_bf16, int n, float scale, int el, uint8_t *out_codes, mm_stream_t stream);
/* Issue-rate microbenchmark: `blocks` workgroups of 4 waves, each wave issues iters*8 independent scaled
 * MFMAs on register operands taken from seed_regs ([128][8] int32).  flops = blocks*4*iters*8*2*M*N*K. */
int mm_diag_mfma_rate(int shape, int el_a, int el_b, int blocks, int iters, const void *seed_regs, void *sink,
                      mm_stream_t stream);
/* L2 -> CU read-bandwidth microbenchmark: `blocks` workgroups each move kb_per_iter KiB per iteration from a hot region;
 * mode 0 = register loads, 1 = contiguous LDS-DMA, 2 = LDS-DMA of 8 x 128-byte rows `stride` bytes apart. */
int mm_diag_l2_bw(const void *buf, unsigned region, int stride, int kb_per_iter, int iters, int mode, int blocks, void *sink,
                  mm_stream_t stream);
/* One pass over `rows` rows of `pitch` bytes (1024, 2048 or 4096; rows % 32 == 0), 32 rows per 512-thread workgroup, every load issued
 * before the first use, nothing computed or written: what a launch that only streams a small-M GEMM's weights costs, by access pattern
 * (0 lane-contiguous, 1 lane = row with 64-byte slab s on wave s % 8, 2 the same with whole 128-byte lines per wave). tools/stream_floor.py */
int mm_diag_stream_once(const void *buf, int rows, int pitch, int pattern, void *sink, mm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICROMIX_DIAG_H */
