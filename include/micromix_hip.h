/*
 * micromix_hip.h -- C ABI of libmicromix_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the MicroMix `mixedgemm` extension's MX hot path.  Every
 * entry point takes plain device pointers, sizes and a HIP stream; no torch
 * types.  The caller (the Python `mixedgemm` module, or a C++/pybind binding a
 * reference maintainer writes) owns allocation, shape
 * derivation and stream/device selection.
 *
 * Reference interfaces replaced (paths relative to the MicroMix tree):
 *   mm_reorder_quantize  <- run_reorder_quantize_x   mgemm/src/reorder.cu:434-469
 *                           run_reorder_quantize_w   mgemm/src/reorder.cu:471-506
 *                           run_reorder_quantize_w4  mgemm/src/reorder.cu:508-543
 *                           (bindings: mgemm/src/bindings.cpp:104-151,155-202,206-253)
 *   mm_matmul            <- matmul_host / matmul_w4_host   mgemm/src/gemm.cu:26-78
 *                           (binding: mgemm/src/bindings.cpp:50-102)
 *   mm_rmsnorm_quantize  <- run_rmsnorm_bf16_mixed   mgemm/src/rmsnorm.cu:314-352 (binding bindings.cpp:257-303)
 *   mm_activate_quantize / mm_downproj_quantize <- mgemm/src/activate.cu:510-551 (bindings.cpp:307-387)
 *   mm_sf_bytes_x / _w   <- SF allocation sizes      mgemm/src/bindings.cpp:120-123,170-172
 *   mm_sf_offset         <- SF layout atom            mgemm/include/sm120_sf_layout.h:170-173
 *
 * All functions are asynchronous with respect to the host (kernels are queued
 * on `stream`) and return an mm_status code; they never exit() or abort().
 */
#ifndef MICROMIX_HIP_H
#define MICROMIX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mm_stream_t; /* hipStream_t */

enum mm_status {
    MM_OK = 0,
    MM_ERR_BAD_SPLIT = 1,   /* KN+KS+KO != K, or not multiples of 128 (reference: "Value error in run_reorder_quantize_*") */
    MM_ERR_BAD_ARG = 2,     /* null pointer / negative size / K too large for int16 indices */
    MM_ERR_LAUNCH = 3,      /* HIP launch error; text via mm_last_error() */
    MM_ERR_UNSUPPORTED = 4, /* configuration not built */
    MM_ERR_NO_DEVICE = 5    /* no gfx950 device */
};

/* mode argument of mm_reorder_quantize */
enum mm_quant_mode {
    MM_QUANT_MIXED = 0, /* segments -> MXFP4 | MXFP6(E3M2) | MXFP8(E4M3)   (reorder_quantize_x / _w)  */
    MM_QUANT_W4 = 1     /* segments -> MXFP4 | MXFP4 | MXFP4                (reorder_quantize_w4)       */
};

/* wmode argument of mm_matmul: format of the B (weight) segments */
enum mm_weight_mode {
    MM_W_MATCH = 0, /* B = fp4 | fp6 | fp8  (matmul_host,    gemm.cu:26-51) */
    MM_W_FP4 = 1    /* B = fp4 | fp4 | fp4  (matmul_w4_host, gemm.cu:53-78) */
};

/* flags argument of mm_matmul */
enum mm_matmul_flags {
    MM_ROUND_PER_SEGMENT = 0, /* default: accumulator rounded through bf16 after each segment, as the
                                 reference's three chained kernels do (gemm.cu:75-77) */
    MM_ROUND_ONCE = 1,        /* single fp32 accumulator across segments, one bf16 rounding */
    MM_SPLIT_K_ALWAYS = 2,    /* mm_matmul_ws: split K whenever the shape allows it, not only where the cost model expects a
                                 gain (tests and tuning) */
    MM_OUT_F32 = 8,           /* mm_matmul / mm_matmul_ws: D is [M,N] FP32 and receives the fp32 accumulator itself, unrounded.  For the
                                 partial products of a K-sharded (row-parallel) tensor-parallel layer: the ranks' partials are summed in
                                 fp32 and rounded to bf16 once, so the result does not degrade with the number of ranks.  Needs
                                 MM_ROUND_ONCE and bias_bf16 == NULL (the bias is added after the reduction); mm_matmul_grouped and mm_qlinear_decode
                                 return MM_ERR_UNSUPPORTED for it. */
    MM_WS_TICKETS_ZEROED = 4  /* mm_matmul_ws / mm_matmul_workspace_bytes / mm_matmul_describe: the first MM_WS_TICKET_BYTES of the
                                 workspace are ZERO (the caller cleared them once, when it created the workspace; every launch
                                 leaves them zero again).  Enables the split-K whose reduction runs inside the GEMM launch (64-row
                                 tiles, launches with few tiles): its workgroups count their arrivals per tile there.  Without the
                                 flag the workspace may hold anything, and that path is not taken. */
};
#define MM_WS_TICKET_BYTES 4096

int mm_version(void); /* major * 10000 + minor * 100 + patch */
const char *mm_strerror(int status);
/* Text of the last HIP error seen by this thread ("" if none). */
const char *mm_last_error(void);

/* Scale-factor tensor geometry (one tensor per segment). */
size_t mm_sf_bytes_x(int M, int Kseg); /* (M/128+1)*128 * Kseg/32, bindings.cpp:120-123 */
size_t mm_sf_bytes_w(int N, int Kseg); /* ceil(N/128)*128 * Kseg/32, bindings.cpp:170-172 */
size_t mm_sf_offset(int row, int block, int Kseg);

/*
 * Fused column-reorder + per-32-group absmax + E8M0 scale + MXFP4/6/8 quantize + pack.
 *   src_bf16       [rows, K] bf16, row-major, contiguous
 *   reorder_index  [K] int16: output column j takes input column reorder_index[j]
 *   KN,KS,KO       widths of the three reordered segments; multiples of 128; KN+KS+KO == K
 *   oN [rows,KN/2]; oS [rows,3*KS/4] (mixed) or [rows,KS/2] (w4); oO [rows,KO] (mixed) or [rows,KO/2] (w4)
 *   sfN,sfS,sfO    UE8M0 bytes in the layout of mm_sf_offset; buffers of at least
 *                  mm_sf_bytes_x(rows,Kseg) (activations) / mm_sf_bytes_w(rows,Kseg) (weights) bytes.
 *                  Only the bytes of real rows are written (the reference leaves padding uninitialised).
 * Pointers of zero-width segments may be NULL.
 */
int mm_reorder_quantize(const void *src_bf16, int rows, int K, const int16_t *reorder_index, int KN, int KS, int KO,
                        int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                        mm_stream_t stream);

/*
 * Same kernel, gathering only a SUBSET of the input columns: `index` has KN+KS+KO (<= K_in) entries, each
 * < K_in, and the rows of src are K_in wide.  This is what a K-sharded (row-parallel) tensor-parallel rank
 * runs: it quantizes just its 128-aligned slices of the reordered segments (micromix_amd/tp.py).  No
 * counterpart in the reference, which has no tensor parallelism.
 */
int mm_reorder_quantize_gather(const void *src_bf16, int rows, int K_in, const int16_t *index, int KN, int KS, int KO,
                               int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                               mm_stream_t stream);

/*
 * Reorder-free quantizers (reference: mgemm/src/activate.cu:44-202, 208-500; bindings.cpp:307-387).  Natural column order,
 * scale = amax > 1e-6 ? 2^ceil(log2(amax/FMAX)) : 1.0, a single RNE rounding from fp32.  Every SF buffer holds
 * mm_sf_bytes_x(rows, Kseg) bytes (the reference sizes the weight variants that way too, bindings.cpp:348-350).
 *   mm_activate_quantize : v = silu(A) * B, A and B [rows, KN+KS+KO] bf16 -> fp4 | fp6 | fp8   (activate_quantize_x); silu in fp32
 *                          with the hardware exp2 / reciprocal (a few fp32 ulps, as the reference's CUDA expf)
 *   mm_downproj_quantize : v = W;  mode MM_QUANT_MIXED -> fp4 | fp6 | fp8 (downproj_quantize_w),
 *                                  mode MM_QUANT_W4    -> fp4 | fp4 | fp4 (downproj_quantize_w4)
 */
int mm_activate_quantize(const void *A_bf16, const void *B_bf16, int rows, int KN, int KS, int KO, uint8_t *oN, uint8_t *oS,
                         uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, mm_stream_t stream);
int mm_downproj_quantize(const void *W_bf16, int rows, int KN, int KS, int KO, int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO,
                         uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, mm_stream_t stream);

/*
 * RMSNorm fused with reorder + quantize (reference: rmsnorm_bf16_mixed_kernel, mgemm/src/rmsnorm.cu:95-312; binding
 * rmsnorm_quantize_x, bindings.cpp:257-303).  v = bf16(x[idx] * w[idx] * rsqrt(mean(x^2) + eps)), then the mixed quantizer of
 * mm_reorder_quantize on v (zero block -> byte 126) -- with the reference's extra step (rmsnorm.cu:262-267): the scaled value
 * is rounded to an integer (half away from zero), clamped and rounded through bf16 before the element conversion.
 *   flags  MM_RMS_REFERENCE (0): as the reference;  MM_RMS_NO_INTEGER_ROUND: without that step
 *   X_bf16 [rows, K], W_bf16 [K] norm weight, reorder_index [K] int16; outputs as mm_reorder_quantize(MM_QUANT_MIXED),
 *   SF buffers mm_sf_bytes_x(rows, Kseg).  Any K % 128 == 0 up to 32768 (the reference compiles 3072, 3584, 4096, 5120 and its
 *   block reduction is only right for 4096).
 */
enum mm_rmsnorm_flags { MM_RMS_REFERENCE = 0, MM_RMS_NO_INTEGER_ROUND = 1 };
int mm_rmsnorm_quantize(const void *X_bf16, const void *W_bf16, float eps, int rows, int K, const int16_t *reorder_index, int KN,
                        int KS, int KO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS,
                        uint8_t *sfO, mm_stream_t stream);

/*
 * Three-segment mixed-precision block-scaled GEMM:
 *   D[m,n] = bf16( sum over segments, blocks b:  2^(sfa[m,b]-127) * 2^(sfb[n,b]-127) * sum_{k in b} a[m,k]*b[n,k] ) (+ bias[n])
 *   A segments: AN [M,KN/2] fp4, AS [M,3KS/4] fp6(E3M2), AO [M,KO] fp8(E4M3)
 *   B segments: wmode MM_W_MATCH: same formats as A;  MM_W_FP4: all fp4 ([N,Kseg/2])
 *   bias_bf16   optional [N] bf16 (NULL for none), added after the final rounding and rounded
 *               again, i.e. exactly `y = matmul(...); y = y + bias` (qLinearLayer.py:68-71)
 *   D_bf16      [M,N] bf16 row-major; fully overwritten (no pre-zeroing needed)
 */
int mm_matmul(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
              const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
              const uint8_t *SFAO, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO, int wmode, int flags,
              const void *bias_bf16, void *D_bf16, mm_stream_t stream);

/*
 * mm_matmul with a caller-owned scratch buffer.  For shapes with few output tiles (medium M or small N; M > 32) the GEMM
 * splits K across workgroups, which needs room for fp32 partial sums; the library never allocates, so the caller passes
 *   workspace        device buffer of at least mm_matmul_workspace_bytes(...) bytes, 16-byte aligned, not shared with a
 *                    call that may run concurrently on another stream (NULL or too small: same results, without the split)
 * mm_matmul_workspace_bytes returns 0 when the shape would not be split.  Results are deterministic either way; they differ
 * from the unsplit kernel only in fp32 summation order (the bf16 rounding chain of MM_ROUND_PER_SEGMENT is kept).
 */
size_t mm_matmul_workspace_bytes(int M, int N, int KN, int KS, int KO, int wmode, int flags);
int mm_matmul_ws(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
                 const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
                 const uint8_t *SFAO, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO, int wmode, int flags,
                 const void *bias_bf16, void *D_bf16, void *workspace, size_t workspace_bytes, mm_stream_t stream);
/*
 * gate_proj + up_proj + silu(gate) * up + MX quantization for down_proj as ONE launch (M > 64).  Replaces the reference's
 *     gate = gate_proj(x); up = up_proj(x); h = act_fn(gate) * up; down_proj quantizes h        (model/qLlamaLayer.py:377-387)
 * and this library's own three-op form  mm_matmul (twice) -> mm_activate_quantize (mgemm/src/activate.cu:44-202,
 * bindings.cpp:307-334) with the bytes of the latter: gate and up are rounded to bf16 as mm_matmul rounds them, silu * up and the
 * quantization are those of mm_activate_quantize, so the outputs are bit-identical to that pair -- without the [M, 2 I] bf16
 * round trip through HBM.
 *   A*, SFA*         the quantized activations x, as for mm_matmul ([M, K], split KN | KS | KO)
 *   B*, SFB*         ONE packed fp4 weight of 2 I rows (MM_W_FP4 layout: [2I, KN/2], [2I, KS/2], [2I, KO/2] + scale tensors): rows
 *                    [256 j, 256 j + 128) are gate_proj's rows [128 j, 128 j + 128), rows [256 j + 128, 256 j + 256) up_proj's rows of
 *                    the same indices (both packed with x's reorder index; scale tensors interleave the same way, one 128-row tile =
 *                    (Kseg/128) * 512 bytes).  I must be a multiple of 128.
 *   DN, DS, DO       the consumer's (down_proj's) split of its K = I input features, natural column order as mm_activate_quantize
 *   o*, sf*          the consumer's activation operands: [M, DN/2], [M, 3 DS/4], [M, DO] and scale tensors of mm_sf_bytes_x(M, D*) bytes
 *   flags            MM_ROUND_PER_SEGMENT (default) or MM_ROUND_ONCE: the rounding of gate / up, as mm_matmul
 *   workspace        only for M <= 64 (mm_gate_up_activate_workspace_bytes(M, I) > 0): M * 2 I bf16 values of scratch, 16-byte aligned;
 *                    those sizes run the weight-streaming GEMM into it and the activation quantizer on it (same bytes, two launches)
 */
size_t mm_gate_up_activate_workspace_bytes(int M, int I);
int mm_gate_up_activate(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
                        const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
                        const uint8_t *SFAO, const uint8_t *SFBO, int M, int I, int KN, int KS, int KO, int DN, int DS, int DO,
                        int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, void *workspace,
                        size_t workspace_bytes, mm_stream_t stream);
/* The same for decode-sized batches, from the bf16 activations: reorder + quantize + gate | up GEMM in one launch (mm_qlinear_decode on the
 * interleaved weight, M <= 8, mm_qlinear_decode_supported(M, 2 I, ...) != 0, else MM_ERR_UNSUPPORTED) into `workspace` (M * 2 I bf16
 * values, 16-byte aligned), then silu(gate) * up + the MX quantization for down_proj: two launches instead of the three of
 * mm_reorder_quantize -> mm_gate_up_activate, the same bytes.  flags as mm_gate_up_activate. */
int mm_gate_up_activate_decode(const void *X_bf16, const int16_t *reorder_index, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO,
                               const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int I, int KN, int KS, int KO, int DN,
                               int DS, int DO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                               void *workspace, size_t workspace_bytes, mm_stream_t stream);
/* ... with the RMSNorm in front (post_attention_layernorm -> gate / up -> act_fn -> the quantization for down_proj; round 6, version >= 510):
 * mm_rmsnorm_quantize -> mm_gate_up_activate, the same bytes.  On wide layers (2 I / 64 >= the CUs, K <= 8192) and M <= 4 it is ONE
 * weight-streaming launch with the norm, the quantization of x, the GEMM, silu(gate) * up and the consumer's quantization inside
 * (`workspace` unused); otherwise two launches (`workspace` as mm_gate_up_activate_decode).  Since round 6 mm_gate_up_activate (M <= 32)
 * and mm_gate_up_activate_decode (M <= 4) run as one such launch too on layers that wide; down_proj is then a plain mm_matmul on o* / sf*.
 * The _supported queries (also for mm_gate_up_activate_decode): 0 cannot run; 1 runs; 2 runs as ONE launch and is expected to be the
 * fastest form of the MLP's first half (M <= 2; beyond that mm_rmsnorm_quantize / mm_reorder_quantize -> mm_gate_up_activate wins).
 * flags: MM_ROUND_* | MM_NORM_NO_INTEGER_ROUND. */
int mm_gate_up_activate_decode_supported(int M, int I, int KN, int KS, int KO);
int mm_rmsnorm_gate_up_activate_decode_supported(int M, int I, int KN, int KS, int KO);
int mm_rmsnorm_gate_up_activate_decode(const void *X_bf16, const void *norm_weight_bf16, float eps, const int16_t *reorder_index, const uint8_t *BN,
                                       const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M,
                                       int I, int KN, int KS, int KO, int DN, int DS, int DO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO,
                                       uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, void *workspace, size_t workspace_bytes, mm_stream_t stream);
/* The other half of a decode-sized MLP: down_proj straight from the bf16 gate | up matrix GU [M, 2 I] (128 gate columns alternating with
 * the 128 up columns of the same indices: the layout mm_gate_up_activate's scratch and mm_qlinear_decode on an interleaved weight produce;
 * 16-byte aligned).  Every workgroup of the weight-streaming GEMM computes silu(gate) * up and quantizes it for its own use -- the
 * bytes of mm_activate_quantize (activate.cu:44-202) -- so the result equals mm_matmul on mm_activate_quantize's output, in ONE launch
 * instead of two.  (DN, DS, DO) = down_proj's split of the I intermediate features in natural column order; B / SFB = its packed weights
 * (mm_downproj_quantize); wmode / flags / bias as mm_matmul.  M <= 4: mm_down_activate_decode_supported() returns 0 if the shape cannot
 * run, 1 if it can, 2 if it is expected to beat the two-launch form. */
int mm_down_activate_decode_supported(int M, int N, int DN, int DS, int DO);
/* ... for the weight mode the launch will run in (MM_W_FP4 / MM_W_MATCH).  The five-argument form answers for matching-precision weights,
 * whose ring / reduction tail in LDS is the larger one (64 KB against 48 KB): what it accepts launches in either mode, but it turns away
 * long-K shapes that fit with fp4 weights (round 6; version >= 500) */
int mm_down_activate_decode_supported_w(int M, int N, int DN, int DS, int DO, int wmode);
int mm_down_activate_decode(const void *GU_bf16, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS,
                            const uint8_t *SFBO, int M, int N, int DN, int DS, int DO, int wmode, int flags, const void *bias_bf16, void *D_bf16,
                            mm_stream_t stream);
/* which kernels mm_gate_up_activate launches for (M, I) (thread-local buffer, as mm_matmul_describe) */
const char *mm_gate_up_activate_describe(int M, int I);

/*
 * Re-arms a workspace for MM_WS_TICKETS_ZEROED: queues a one-workgroup kernel on `stream` that clears its first MM_WS_TICKET_BYTES
 * (a kernel node when the stream is being captured into a hipGraph; `workspace` must be 16-byte aligned).  Call it when the workspace is created, inside every graph capture that uses a
 * workspace of its own, and after a launch that did not complete (a fault or a reset mid-launch can leave a ticket counter of the
 * in-kernel split-K non-zero, and a later launch on that workspace would then never see its last arrival and never write that tile).
 */
int mm_matmul_ws_reset(void *workspace, size_t workspace_bytes, mm_stream_t stream);

/*
 * Grouped GEMM (MoE experts; reference caller: the per-expert loop of model/qMixtralLayer.py:507-519, one matmul per expert and
 * linear): `ngroups` independent products D_g = matmul(A_g, B_g) that share N, the (KN, KS, KO) split, the weight mode and the
 * flags but have their own operands, token counts and outputs.  Groups of at most 64 token rows share launches of the
 * weight-streaming kernels, larger groups launches of the tiled kernels, 8 groups per launch.  Results are bit-identical to
 * ngroups calls of mm_matmul (which never splits K).  `groups` is a HOST array (copied into the kernel arguments).
 */
typedef struct mm_quant_group {
    const void *src_bf16;           /* [rows, K] bf16: the token rows routed to this expert */
    const int16_t *reorder_index;   /* [K]: this expert's own index */
    uint8_t *oN, *oS, *oO, *sfN, *sfS, *sfO; /* outputs as mm_reorder_quantize */
    int rows;                        /* 0 = skip */
} mm_quant_group;
/* mm_reorder_quantize for `ngroups` independent row sets that share K and the split (the per-expert reorder_quantize_x calls of
 * qMixtralLayer.py:507-519), 8 groups per launch; `groups` is a HOST array.  Bit-identical to the separate calls. */
int mm_reorder_quantize_grouped(const mm_quant_group *groups, int ngroups, int K, int KN, int KS, int KO, int mode, mm_stream_t stream);

typedef struct mm_group {
    const uint8_t *AN, *AS, *AO, *SFAN, *SFAS, *SFAO; /* activations of this group: [M, KN/2], [M, 3KS/4], [M, KO] + scales */
    const uint8_t *BN, *BS, *BO, *SFBN, *SFBS, *SFBO; /* its packed weights */
    const void *bias_bf16;                             /* optional [N] */
    void *D;                                           /* [M, N] bf16 */
    int M;                                             /* token rows of this group (0 = skip) */
} mm_group;
int mm_matmul_grouped(const mm_group *groups, int ngroups, int N, int KN, int KS, int KO, int wmode, int flags, mm_stream_t stream);

/*
 * QLinearLayer.forward for decode-sized inputs in ONE launch (reference: qLinearLayer.py:58-74 = reorder_quantize_x + matmul
 * (+ bias)): every workgroup quantizes the M activation rows into LDS itself and then streams its weight rows.  Bit-identical to
 * mm_reorder_quantize(MM_QUANT_MIXED) followed by mm_matmul.  mm_qlinear_decode_supported() returns 0 when the shape cannot run
 * (needs 1 <= M <= 8 and the quantized rows in LDS; mm_qlinear_decode then returns MM_ERR_UNSUPPORTED), 1 when it can, 2 when it
 * can and is expected to be faster than the two calls (every workgroup repeats the quantization, so many rows x many
 * workgroup rounds lose).
 *   X_bf16 [M, KN+KS+KO] bf16, reorder_index [K] int16, B / SFB as mm_matmul, flags MM_ROUND_*, bias optional, D [M, N] bf16
 */
int mm_qlinear_decode_supported(int M, int N, int KN, int KS, int KO);
int mm_qlinear_decode_supported_w(int M, int N, int KN, int KS, int KO, int wmode);      /* as mm_down_activate_decode_supported_w */
int mm_qlinear_decode(const void *X_bf16, const int16_t *reorder_index, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO,
                      const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO,
                      int wmode, int flags, const void *bias_bf16, void *D_bf16, mm_stream_t stream);

/*
 * The same with the RMSNorm that precedes q/k/v and gate/up in the reference's decoder layers inside the launch (reference:
 * rmsnorm_quantize_x, mgemm/src/rmsnorm.cu:95-352 / bindings.cpp:257-303, followed by matmul; model/qLlamaLayer.py input_layernorm ->
 * q/k/v, post_attention_layernorm -> gate/up): every workgroup computes the row's sum of squares in the reference's summation order,
 * v = bf16((x * w) * rvar), the reference's integer rounding, and quantizes.  Bit-identical to mm_rmsnorm_quantize followed by
 * mm_matmul.  Needs K <= 8192 besides the conditions of mm_qlinear_decode; X and norm_weight 16-byte aligned.
 *   flags: MM_ROUND_* | MM_NORM_NO_INTEGER_ROUND (= mm_rmsnorm_quantize's MM_RMS_NO_INTEGER_ROUND)
 */
#define MM_NORM_NO_INTEGER_ROUND 0x100
int mm_rmsnorm_qlinear_decode_supported(int M, int N, int KN, int KS, int KO);
int mm_rmsnorm_qlinear_decode_supported_w(int M, int N, int KN, int KS, int KO, int wmode);
int mm_rmsnorm_qlinear_decode(const void *X_bf16, const void *norm_weight_bf16, float eps, const int16_t *reorder_index, const uint8_t *BN,
                              const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int N,
                              int KN, int KS, int KO, int wmode, int flags, const void *bias_bf16, void *D_bf16, mm_stream_t stream);

/* Which kernel(s) and how many workgroups mm_matmul / mm_matmul_ws launch for this problem on the CURRENT device (the same
 * decision code as the launcher; workspace_bytes = 0 means "no workspace", i.e. never split-K).  Returns a string in a
 * thread-local buffer, valid until the calling thread's next call.  Used by bench.py to name the kernel it timed. */
const char *mm_matmul_describe(int M, int N, int KN, int KS, int KO, int wmode, int flags, size_t workspace_bytes);

/* bindings.cpp:700 `m.def("test_function", ...)`: the reference module's liveness probe; returns the same constant string. */
const char *mm_test_function(void);

/*
 * Measurement hook -- UNSTABLE, not part of the drop-in interface (no reference counterpart; bench.py and tools/ use it; it changes
 * no result).  THREAD-LOCAL: it affects only launches made by the calling thread.
 *
 * mm_diag_set_kernel_events: while a pair of hipEvent_t is registered, every tiled-GEMM launch (M > 64, and the M > 32 shapes that
 *   run on tiles; mm_gate_up_activate's fused launch) and every quantizer launch (mm_reorder_quantize, mm_rmsnorm_quantize,
 *   mm_activate_quantize, mm_downproj_quantize) of this thread attaches them to its own dispatch (hipExtLaunchKernel start/stop events), so hipEventElapsedTime
 *   gives the kernel's duration as rocprofv3 reports it, without the launch gap that events recorded around the call include.
 *   NULL, NULL disables.  Host-side only: the kernels are the same with and without it.
 *
 * The in-kernel clock stamps (mm_diag_set_clock_buffer) and the MM_DBG ablation switches exist only in the instrumented
 * developer variant of the library (-DMM_INSTRUMENT, csrc/mx_instrument.h, tools/build_variant.sh); the default library neither
 * exports that symbol nor contains the stores.  The hardware microbenchmarks and probes (mm_diag_mfma, mm_diag_hw_convert,
 * mm_diag_mfma_rate, mm_diag_l2_bw, mm_diag_stream_once) live in libmicromix_diag.so, declared in include/micromix_diag.h.
 */
int mm_diag_set_kernel_events(void *start_event, void *stop_event);
#ifdef MM_INSTRUMENT
/* device buffer of 4 x 8 bytes per workgroup: {main-loop s_memtime delta, main-loop s_memrealtime delta, start tick,
 * end-of-epilogue tick delta} of every workgroup of the large-M GEMM (in-kernel clock = ratio x 100 MHz); NULL disables. */
int mm_diag_set_clock_buffer(void *buf);
/* device buffer of 4 x 8 bytes per workgroup of reorder_quantize_kernel: {start, row staged, first group stored, end} ticks */
int mm_diag_set_quant_clock_buffer(void *buf);
#endif

#ifdef __cplusplus
}
#endif
#endif /* MICROMIX_HIP_H */
