// extern "C" surface of libmicromix_hip.so -- see include/micromix_hip.h.
#include "../../include/micromix_hip.h"
#include "mx_common.h"
#include "mx_kernels.h"

#include <stdio.h>
#include <string.h>


static thread_local char g_last_error[256] = "";
// measurement hooks, per calling thread (see mm_diag_set_kernel_events / mm_diag_set_clock_buffer in the header)
static thread_local mm::DiagEvents g_ev = {nullptr, nullptr};
namespace mm { DiagEvents &diag_events() { return g_ev; } }
static thread_local unsigned long long *g_clock_buf = nullptr;

static int fail_hip(hipError_t e, const char *where) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", where, hipGetErrorString(e));
    return MM_ERR_LAUNCH;
}

// The weight mode a launch really runs in: MM_W_FP4 as given; MM_W_MATCH with KS = KO = 0 has only the N segment, whose weights are
// fp4 in both modes, so it takes the fp4-weight kernels too (the matching-precision 256 x 256 kernel sits at its register limit).
static bool weights_fp4(int wmode, int KS, int KO) { return wmode == MM_W_FP4 || (KS == 0 && KO == 0); }

static bool split_ok(int K, int KN, int KS, int KO) {
    return KN >= 0 && KS >= 0 && KO >= 0 && (KN % 128) == 0 && (KS % 128) == 0 && (KO % 128) == 0 && KN + KS + KO == K &&
           K > 0;
}

extern "C" {

int mm_version(void) { return 510; /* 0.5.1: + mm_rmsnorm_gate_up_activate_decode(_supported), mm_gate_up_activate_decode_supported; mm_gate_up_activate(_decode) one launch at decode sizes; 0.5.0: + the *_supported_w queries (weight mode); 0.4.0: + mm_rmsnorm_qlinear_decode(_supported) (0.3.0: + mm_gate_up_activate(_decode), mm_down_activate_decode, mm_matmul_ws_reset; 0.2.0: diagnostics moved to libmicromix_diag.so, + mm_test_function) */ }

const char *mm_test_function(void) { return "Hello from test_function!"; /* bindings.cpp:700 */ }

const char *mm_strerror(int status) {
    switch (status) {
        case MM_OK: return "ok";
        case MM_ERR_BAD_SPLIT: return "KN, KS, KO must be non-negative multiples of 128 that sum to K";
        case MM_ERR_BAD_ARG: return "bad argument";
        case MM_ERR_LAUNCH: return "HIP error";
        case MM_ERR_UNSUPPORTED: return "unsupported configuration";
        case MM_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

const char *mm_last_error(void) { return g_last_error; }

size_t mm_sf_bytes_x(int M, int Kseg) { return (size_t)(M / 128 + 1) * 128u * (size_t)(Kseg / 32); }
size_t mm_sf_bytes_w(int N, int Kseg) { return (size_t)((N + 127) / 128) * 128u * (size_t)(Kseg / 32); }
size_t mm_sf_offset(int row, int block, int Kseg) { return mm::sf_offset(row, block, Kseg); }

int mm_reorder_quantize(const void *src_bf16, int rows, int K, const int16_t *reorder_index, int KN, int KS, int KO,
                        int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                        mm_stream_t stream) {
    if (!split_ok(K, KN, KS, KO)) return MM_ERR_BAD_SPLIT;
    if (rows < 0 || K > 32768 || (mode != MM_QUANT_MIXED && mode != MM_QUANT_W4)) return MM_ERR_BAD_ARG;
    if (rows == 0) return MM_OK;
    if (!src_bf16 || !reorder_index) return MM_ERR_BAD_ARG;
    if ((KN && (!oN || !sfN)) || (KS && (!oS || !sfS)) || (KO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    hipError_t e = mm::launch_reorder_quantize(src_bf16, rows, K, reorder_index, KN, KS, KO, mode == MM_QUANT_W4, oN, oS,
                                               oO, sfN, sfS, sfO, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_reorder_quantize");
}

int mm_reorder_quantize_gather(const void *src_bf16, int rows, int K_in, const int16_t *index, int KN, int KS, int KO,
                               int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                               mm_stream_t stream) {
    if (!split_ok(KN + KS + KO, KN, KS, KO) || K_in <= 0 || (K_in % 128) || KN + KS + KO > K_in) return MM_ERR_BAD_SPLIT;
    if (rows < 0 || K_in > 32768 || (mode != MM_QUANT_MIXED && mode != MM_QUANT_W4)) return MM_ERR_BAD_ARG;
    if (rows == 0) return MM_OK;
    if (!src_bf16 || !index) return MM_ERR_BAD_ARG;
    if ((KN && (!oN || !sfN)) || (KS && (!oS || !sfS)) || (KO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    hipError_t e = mm::launch_reorder_quantize(src_bf16, rows, K_in, index, KN, KS, KO, mode == MM_QUANT_W4, oN, oS, oO,
                                               sfN, sfS, sfO, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_reorder_quantize_gather");
}

int mm_activate_quantize(const void *A_bf16, const void *B_bf16, int rows, int KN, int KS, int KO, uint8_t *oN, uint8_t *oS,
                         uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, mm_stream_t stream) {
    if (!split_ok(KN + KS + KO, KN, KS, KO)) return MM_ERR_BAD_SPLIT;
    if (rows < 0) return MM_ERR_BAD_ARG;
    if (rows == 0) return MM_OK;
    if (!A_bf16 || !B_bf16) return MM_ERR_BAD_ARG;
    if ((KN && (!oN || !sfN)) || (KS && (!oS || !sfS)) || (KO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    hipError_t e = mm::launch_direct_quantize(A_bf16, B_bf16, rows, KN, KS, KO, 0, oN, oS, oO, sfN, sfS, sfO, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_activate_quantize");
}

int mm_downproj_quantize(const void *W_bf16, int rows, int KN, int KS, int KO, int mode, uint8_t *oN, uint8_t *oS, uint8_t *oO,
                         uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, mm_stream_t stream) {
    if (!split_ok(KN + KS + KO, KN, KS, KO)) return MM_ERR_BAD_SPLIT;
    if (rows < 0 || (mode != MM_QUANT_MIXED && mode != MM_QUANT_W4)) return MM_ERR_BAD_ARG;
    if (rows == 0) return MM_OK;
    if (!W_bf16) return MM_ERR_BAD_ARG;
    if ((KN && (!oN || !sfN)) || (KS && (!oS || !sfS)) || (KO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    hipError_t e = mm::launch_direct_quantize(W_bf16, nullptr, rows, KN, KS, KO, mode == MM_QUANT_W4 ? 2 : 1, oN, oS, oO, sfN, sfS,
                                              sfO, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_downproj_quantize");
}

int mm_rmsnorm_quantize(const void *X_bf16, const void *W_bf16, float eps, int rows, int K, const int16_t *reorder_index, int KN,
                        int KS, int KO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS,
                        uint8_t *sfO, mm_stream_t stream) {
    if (!split_ok(K, KN, KS, KO)) return MM_ERR_BAD_SPLIT;
    if (rows < 0 || K > 32768) return MM_ERR_BAD_ARG;
    if (rows == 0) return MM_OK;
    if (!X_bf16 || !W_bf16 || !reorder_index) return MM_ERR_BAD_ARG;
    if ((KN && (!oN || !sfN)) || (KS && (!oS || !sfS)) || (KO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    hipError_t e = mm::launch_rmsnorm_quantize(X_bf16, W_bf16, eps, rows, K, reorder_index, KN, KS, KO,
                                               !(flags & MM_RMS_NO_INTEGER_ROUND), oN, oS, oO, sfN, sfS, sfO, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_rmsnorm_quantize");
}

// (the five-argument queries answer for the matching-precision weight mode, whose ring / reduction tail is the larger one: what they
// accept launches in either mode; the _w forms take the weight mode and accept the long-K fp4 shapes the 48 KB tail leaves room for)
int mm_qlinear_decode_supported_w(int M, int N, int KN, int KS, int KO, int wmode) {
    if (N < 0 || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0 || KN + KS + KO > 32768) return 0;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return 0;
    const int K[3] = {KN, KS, KO};
    return mm::qlinear_decode_supported(M, N, K, false, weights_fp4(wmode, KS, KO));
}
int mm_qlinear_decode_supported(int M, int N, int KN, int KS, int KO) { return mm_qlinear_decode_supported_w(M, N, KN, KS, KO, (KS | KO) ? MM_W_MATCH : MM_W_FP4); }

int mm_qlinear_decode(const void *X_bf16, const int16_t *reorder_index, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO,
                      const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO,
                      int wmode, int flags, const void *bias_bf16, void *D_bf16, mm_stream_t stream) {
    if (M < 0 || N < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return MM_ERR_BAD_SPLIT;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return MM_ERR_BAD_ARG;
    if (flags & MM_OUT_F32) return MM_ERR_UNSUPPORTED;    // fp32 partial sums come from mm_matmul only
    if (M == 0 || N == 0) return MM_OK;
    if (!mm_qlinear_decode_supported_w(M, N, KN, KS, KO, wmode)) return MM_ERR_UNSUPPORTED;
    if (!X_bf16 || !reorder_index || !D_bf16) return MM_ERR_BAD_ARG;
    if ((KN && (!BN || !SFBN)) || (KS && (!BS || !SFBS)) || (KO && (!BO || !SFBO))) return MM_ERR_BAD_ARG;
    const uint8_t *W[3] = {BN, BS, BO}, *SFW[3] = {SFBN, SFBS, SFBO};
    const int K[3] = {KN, KS, KO};
    hipError_t e = mm::launch_qlinear_decode(X_bf16, reorder_index, W, SFW, M, N, K, weights_fp4(wmode, KS, KO),
                                             (flags & MM_ROUND_ONCE) ? 0 : 1, bias_bf16, D_bf16, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_qlinear_decode");
}

int mm_rmsnorm_qlinear_decode_supported_w(int M, int N, int KN, int KS, int KO, int wmode) {
    if (N < 0 || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0 || KN + KS + KO > 32768) return 0;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return 0;
    const int K[3] = {KN, KS, KO};
    return mm::qlinear_decode_supported(M, N, K, true, weights_fp4(wmode, KS, KO));
}
int mm_rmsnorm_qlinear_decode_supported(int M, int N, int KN, int KS, int KO) { return mm_rmsnorm_qlinear_decode_supported_w(M, N, KN, KS, KO, (KS | KO) ? MM_W_MATCH : MM_W_FP4); }

int mm_rmsnorm_qlinear_decode(const void *X_bf16, const void *norm_weight_bf16, float eps, const int16_t *reorder_index, const uint8_t *BN,
                              const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int N,
                              int KN, int KS, int KO, int wmode, int flags, const void *bias_bf16, void *D_bf16, mm_stream_t stream) {
    if (M < 0 || N < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return MM_ERR_BAD_SPLIT;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return MM_ERR_BAD_ARG;
    if (flags & ~(MM_ROUND_ONCE | MM_NORM_NO_INTEGER_ROUND)) return MM_ERR_BAD_ARG;
    if (M == 0 || N == 0) return MM_OK;
    if (!mm_rmsnorm_qlinear_decode_supported_w(M, N, KN, KS, KO, wmode)) return MM_ERR_UNSUPPORTED;
    if (!X_bf16 || !norm_weight_bf16 || !reorder_index || !D_bf16) return MM_ERR_BAD_ARG;
    if (((uintptr_t)X_bf16 & 15) || ((uintptr_t)norm_weight_bf16 & 15)) return MM_ERR_BAD_ARG;      // rows and weights are staged in 16-byte pieces
    if ((KN && (!BN || !SFBN)) || (KS && (!BS || !SFBS)) || (KO && (!BO || !SFBO))) return MM_ERR_BAD_ARG;
    const uint8_t *W[3] = {BN, BS, BO}, *SFW[3] = {SFBN, SFBS, SFBO};
    const int K[3] = {KN, KS, KO};
    const mm::NormArgs norm = {norm_weight_bf16, eps, (flags & MM_NORM_NO_INTEGER_ROUND) ? 0 : 1};
    hipError_t e = mm::launch_qlinear_decode(X_bf16, reorder_index, W, SFW, M, N, K, weights_fp4(wmode, KS, KO),
                                             (flags & MM_ROUND_ONCE) ? 0 : 1, bias_bf16, D_bf16, (hipStream_t)stream, norm);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_rmsnorm_qlinear_decode");
}

size_t mm_matmul_workspace_bytes(int M, int N, int KN, int KS, int KO, int wmode, int flags) {
    if (M <= 0 || N <= 0 || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128)) return 0;
    const int K[3] = {KN, KS, KO};
    return mm::mx_gemm_workspace_bytes(M, N, K, weights_fp4(wmode, KS, KO), (flags & MM_SPLIT_K_ALWAYS) != 0, (flags & MM_WS_TICKETS_ZEROED) != 0);
}

const char *mm_matmul_describe(int M, int N, int KN, int KS, int KO, int wmode, int flags, size_t workspace_bytes) {
    if (M <= 0 || N <= 0 || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return "none";
    const int K[3] = {KN, KS, KO};
    if (M <= 64 && !mm::mx_gemm_small_m_uses_tiles(M, N, K, weights_fp4(wmode, KS, KO), workspace_bytes, (flags & MM_SPLIT_K_ALWAYS) != 0))
        return mm::mx_gemm_stream_supported(M, N, K, weights_fp4(wmode, KS, KO)) ? "mm::stream::mx_gemm_stream_kernel (weight streaming, M <= 64)"
                                                                         : "mm::skinny::mx_gemm_skinny*_kernel (weight streaming, M <= 64)";
    return mm::describe_mx_gemm256(M, N, K, weights_fp4(wmode, KS, KO), workspace_bytes, (flags & MM_SPLIT_K_ALWAYS) != 0, (flags & MM_WS_TICKETS_ZEROED) != 0);
}

int mm_matmul(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
              const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
              const uint8_t *SFAO, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO, int wmode, int flags,
              const void *bias_bf16, void *D_bf16, mm_stream_t stream) {
    return mm_matmul_ws(AN, BN, AS, BS, AO, BO, SFAN, SFBN, SFAS, SFBS, SFAO, SFBO, M, N, KN, KS, KO, wmode, flags, bias_bf16,
                        D_bf16, nullptr, 0, stream);
}

int mm_matmul_ws(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
                 const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
                 const uint8_t *SFAO, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO, int wmode, int flags,
                 const void *bias_bf16, void *D_bf16, void *workspace, size_t workspace_bytes, mm_stream_t stream) {
    if (M < 0 || N < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128)) return MM_ERR_BAD_SPLIT;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return MM_ERR_BAD_ARG;
    if (M == 0 || N == 0) return MM_OK;
    if (!D_bf16) return MM_ERR_BAD_ARG;
    const bool out_f32 = (flags & MM_OUT_F32) != 0;
    if (out_f32 && (!(flags & MM_ROUND_ONCE) || bias_bf16)) return MM_ERR_BAD_ARG;   // fp32 partial sums: no chain rounding, no bias
    if ((KN && (!AN || !BN || !SFAN || !SFBN)) || (KS && (!AS || !BS || !SFAS || !SFBS)) ||
        (KO && (!AO || !BO || !SFAO || !SFBO)))
        return MM_ERR_BAD_ARG;
    if (KN + KS + KO == 0) {  // reference: C = zeros, no segment runs (gemm.cu:48-50)
        hipError_t e = hipMemsetAsync(D_bf16, 0, (size_t)M * N * (out_f32 ? 4 : 2), (hipStream_t)stream);
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_matmul(memset)");
    }
    mm::GemmArgs a;
    a.X[0] = AN; a.X[1] = AS; a.X[2] = AO;
    a.W[0] = BN; a.W[1] = BS; a.W[2] = BO;
    a.SFX[0] = SFAN; a.SFX[1] = SFAS; a.SFX[2] = SFAO;
    a.SFW[0] = SFBN; a.SFW[1] = SFBS; a.SFW[2] = SFBO;
    a.K[0] = KN; a.K[1] = KS; a.K[2] = KO;
    a.M = M; a.N = N;
    // the 128-row tiles that hold scales of real rows.  The reference allocates M/128 + 1 tiles (bindings.cpp:120) and never
    // writes the last one when M % 128 == 0; counting only the written tiles keeps every kernel's scale reads inside ANY tensor
    // that holds the scales of its M rows, whoever allocated it.
    a.sfx_row_tiles = (M + 127) / 128;
    a.sfw_row_tiles = (N + 127) / 128;
    a.round_per_segment = (flags & MM_ROUND_ONCE) ? 0 : 1;
    a.bias = (const uint16_t *)bias_bf16;
    a.D = (uint16_t *)D_bf16;
    a.out_f32 = out_f32 ? 1 : 0;
    a.act = 0;
    a.clock_out = g_clock_buf;
    a.ev_start = g_ev.start;
    a.ev_stop = g_ev.stop;
    a.ws = (float *)workspace;
    a.ws_bytes = workspace ? workspace_bytes : 0;
    a.splits = 0;
    a.tickets = nullptr;
    a.tickets_zeroed = (workspace && (flags & MM_WS_TICKETS_ZEROED)) ? 1 : 0;
    a.n_tile0 = a.n_tiles = 0;
    a.force_split = (flags & MM_SPLIT_K_ALWAYS) ? 1 : 0;
    a.split_first[0] = a.split_first[1] = a.split_first[2] = a.split_first[3] = 0;
    hipError_t e = mm::launch_mx_gemm(a, weights_fp4(wmode, KS, KO), (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_matmul");
}

// A kernel, not hipMemsetAsync: captured into a hipGraph (ROCm 7.2, MI355X) a memset node in front of the GEMM gave the right result
// on the first replay only (later replays were wrong as if the clearing were no longer ordered before the kernel); a kernel node
// replays in order (tools/probe_capture.py, profiles/notes_r04.md).
static __global__ void ws_reset_kernel(uint4 *p) { p[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u); }

size_t mm_gate_up_activate_workspace_bytes(int M, int I) {
    if (M <= 0 || I <= 0 || (I % 128)) return 0;
    return mm::mx_gemm_act_supported(M, 2 * I) ? 0 : (size_t)M * (size_t)(2 * I) * sizeof(uint16_t);
}

const char *mm_gate_up_activate_describe(int M, int I) {
    if (M <= 0 || I <= 0 || (I % 128)) return "none";
    const int Kany[3] = {0, 0, 4096};      // (the answer depends on K only for K in the tens of thousands: LDS of the scale images)
    if (!mm::mx_gemm_act_supported(M, 2 * I) && mm::gate_up_act_stream_supported(M, 2 * I, Kany, false, false))
        return "mm::stream::mx_gemm_stream_act_kernel (weight streaming with silu(gate) * up and the consumer's quantization inside, M <= 32)";
    return mm::describe_mx_gemm_act(M, 2 * I);
}

// the fused gate | up weight with the activation inside the weight-streaming launch (mx_gemm_stream.hip, ACT)
static mm::GemmArgs act_stream_args(const uint8_t *BN, const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO,
                                    int M, int I, int KN, int KS, int KO, int DN, int DS, int DO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO,
                                    uint8_t *sfN, uint8_t *sfS, uint8_t *sfO) {
    mm::GemmArgs a = {};
    a.W[0] = BN; a.W[1] = BS; a.W[2] = BO;
    a.SFW[0] = SFBN; a.SFW[1] = SFBS; a.SFW[2] = SFBO;
    a.K[0] = KN; a.K[1] = KS; a.K[2] = KO;
    a.M = M; a.N = 2 * I;
    a.sfx_row_tiles = 1;
    a.sfw_row_tiles = (2 * I + 127) / 128;
    a.round_per_segment = (flags & MM_ROUND_ONCE) ? 0 : 1;
    a.act = 1;
    a.act_K[0] = DN; a.act_K[1] = DS; a.act_K[2] = DO;
    a.act_o[0] = oN; a.act_o[1] = oS; a.act_o[2] = oO;
    a.act_sf[0] = sfN; a.act_sf[1] = sfS; a.act_sf[2] = sfO;
    return a;
}

int mm_gate_up_activate(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
                        const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
                        const uint8_t *SFAO, const uint8_t *SFBO, int M, int I, int KN, int KS, int KO, int DN, int DS, int DO,
                        int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, void *workspace,
                        size_t workspace_bytes, mm_stream_t stream) {
    if (M < 0 || I < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return MM_ERR_BAD_SPLIT;
    if (!split_ok(DN + DS + DO, DN, DS, DO) || DN + DS + DO != I) return MM_ERR_BAD_SPLIT;
    if (flags & ~MM_ROUND_ONCE) return MM_ERR_BAD_ARG;
    if (M == 0) return MM_OK;
    if ((KN && (!AN || !BN || !SFAN || !SFBN)) || (KS && (!AS || !BS || !SFAS || !SFBS)) || (KO && (!AO || !BO || !SFAO || !SFBO)))
        return MM_ERR_BAD_ARG;
    if ((DN && (!oN || !sfN)) || (DS && (!oS || !sfS)) || (DO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    const int N = 2 * I;
    const int Kin[3] = {KN, KS, KO};
    if (!mm::mx_gemm_act_supported(M, N) && mm::gate_up_act_stream_supported(M, N, Kin, false, false)) {
        // M <= 16 on a wide layer (round 6): ONE weight-streaming launch with the activation inside -- no scratch, the same bytes
        mm::GemmArgs a = act_stream_args(BN, BS, BO, SFBN, SFBS, SFBO, M, I, KN, KS, KO, DN, DS, DO, flags, oN, oS, oO, sfN, sfS, sfO);
        a.X[0] = AN; a.X[1] = AS; a.X[2] = AO;
        a.SFX[0] = SFAN; a.SFX[1] = SFAS; a.SFX[2] = SFAO;
        hipError_t e = mm::launch_gate_up_act_stream(a, (hipStream_t)stream);
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_gate_up_activate");
    }
    if (!mm::mx_gemm_act_supported(M, N)) {
        // M <= 64: the weight-streaming GEMM into the caller's scratch (columns alternate 128 gate | 128 up), then the activation
        // quantizer on that layout: the same bytes as the fused epilogue (tests/test_gate_up_gpu.py)
        if (!workspace || workspace_bytes < (size_t)M * N * sizeof(uint16_t) || ((uintptr_t)workspace & 15)) return MM_ERR_BAD_ARG;
        const int st = mm_matmul(AN, BN, AS, BS, AO, BO, SFAN, SFBN, SFAS, SFBS, SFAO, SFBO, M, N, KN, KS, KO, MM_W_FP4, flags, nullptr,
                                 workspace, stream);
        if (st != MM_OK) return st;
        hipError_t e = mm::launch_direct_quantize(workspace, (const uint16_t *)workspace + 128, M, DN, DS, DO, 3, oN, oS, oO, sfN, sfS, sfO,
                                                  (hipStream_t)stream);
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_gate_up_activate");
    }
    mm::GemmArgs a;
    a.X[0] = AN; a.X[1] = AS; a.X[2] = AO;
    a.W[0] = BN; a.W[1] = BS; a.W[2] = BO;
    a.SFX[0] = SFAN; a.SFX[1] = SFAS; a.SFX[2] = SFAO;
    a.SFW[0] = SFBN; a.SFW[1] = SFBS; a.SFW[2] = SFBO;
    a.K[0] = KN; a.K[1] = KS; a.K[2] = KO;
    a.M = M; a.N = N;
    a.sfx_row_tiles = (M + 127) / 128;
    a.sfw_row_tiles = (N + 127) / 128;
    a.round_per_segment = (flags & MM_ROUND_ONCE) ? 0 : 1;
    a.bias = nullptr;
    a.D = nullptr;
    a.out_f32 = 0;
    a.act = 1;
    a.act_K[0] = DN; a.act_K[1] = DS; a.act_K[2] = DO;
    a.act_o[0] = oN; a.act_o[1] = oS; a.act_o[2] = oO;
    a.act_sf[0] = sfN; a.act_sf[1] = sfS; a.act_sf[2] = sfO;
    a.clock_out = g_clock_buf;
    a.ev_start = g_ev.start;
    a.ev_stop = g_ev.stop;
    a.ws = nullptr; a.ws_bytes = 0; a.splits = 0; a.tickets = nullptr; a.tickets_zeroed = 0;
    a.n_tile0 = a.n_tiles = 0;
    a.force_split = 0;
    a.split_first[0] = a.split_first[1] = a.split_first[2] = a.split_first[3] = 0;
    hipError_t e = mm::launch_mx_gemm_act(a, (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_gate_up_activate");
}

int mm_gate_up_activate_decode(const void *X_bf16, const int16_t *reorder_index, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO,
                               const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M, int I, int KN, int KS, int KO, int DN,
                               int DS, int DO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                               void *workspace, size_t workspace_bytes, mm_stream_t stream) {
    if (M < 0 || I < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return MM_ERR_BAD_SPLIT;
    if (!split_ok(DN + DS + DO, DN, DS, DO) || DN + DS + DO != I) return MM_ERR_BAD_SPLIT;
    if (flags & ~MM_ROUND_ONCE) return MM_ERR_BAD_ARG;
    if (M == 0) return MM_OK;
    const int N = 2 * I;
    if (!mm_qlinear_decode_supported_w(M, N, KN, KS, KO, MM_W_FP4)) return MM_ERR_UNSUPPORTED;
    if ((DN && (!oN || !sfN)) || (DS && (!oS || !sfS)) || (DO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    const int Kin[3] = {KN, KS, KO};
    if (mm::gate_up_act_stream_supported(M, N, Kin, true, false)) {
        // M <= 4 on a wide layer (round 6): quantization, GEMM, silu(gate) * up and the consumer's quantization in ONE launch
        if (!X_bf16 || !reorder_index || ((uintptr_t)X_bf16 & 15)) return MM_ERR_BAD_ARG;
        if ((KN && (!BN || !SFBN)) || (KS && (!BS || !SFBS)) || (KO && (!BO || !SFBO))) return MM_ERR_BAD_ARG;
        const mm::GemmArgs a = act_stream_args(BN, BS, BO, SFBN, SFBS, SFBO, M, I, KN, KS, KO, DN, DS, DO, flags, oN, oS, oO, sfN, sfS, sfO);
        hipError_t e = mm::launch_gate_up_act_stream_decode(X_bf16, reorder_index, a, (hipStream_t)stream);
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_gate_up_activate_decode");
    }
    if (!workspace || workspace_bytes < (size_t)M * N * sizeof(uint16_t) || ((uintptr_t)workspace & 15)) return MM_ERR_BAD_ARG;
    // quantize + gate | up GEMM in one launch into the scratch (columns alternate 128 gate | 128 up), then the activation quantizer on
    // that layout: the bytes of mm_reorder_quantize -> mm_gate_up_activate (tests/test_gate_up_gpu.py)
    const int st = mm_qlinear_decode(X_bf16, reorder_index, BN, BS, BO, SFBN, SFBS, SFBO, M, N, KN, KS, KO, MM_W_FP4, flags, nullptr, workspace, stream);
    if (st != MM_OK) return st;
    hipError_t e = mm::launch_direct_quantize(workspace, (const uint16_t *)workspace + 128, M, DN, DS, DO, 3, oN, oS, oO, sfN, sfS, sfO,
                                              (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_gate_up_activate_decode");
}

// 2: ONE launch and expected to be the fastest way through the MLP's first half (M <= 2: at M = 3, 4 every workgroup repeating the
// quantization loses to mm_rmsnorm_quantize / mm_reorder_quantize -> mm_gate_up_activate, itself one launch at M <= 16 --
// tools/time_mlp_decode.py: Llama-3-8B MLP at M = 1 / 2 / 4 23.1 / 25.3 / 36.9 us against 26.3 / 26.7 / 27.2); 1: runs; 0: cannot
int mm_rmsnorm_gate_up_activate_decode_supported(int M, int I, int KN, int KS, int KO) {
    if (M < 1 || I < 128 || (I % 128) || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return 0;
    const int Kin[3] = {KN, KS, KO};
    if (mm::gate_up_act_stream_supported(M, 2 * I, Kin, true, true)) return M <= 2 ? 2 : 1;
    return mm_rmsnorm_qlinear_decode_supported_w(M, 2 * I, KN, KS, KO, MM_W_FP4) ? 1 : 0;
}
int mm_gate_up_activate_decode_supported(int M, int I, int KN, int KS, int KO) {
    if (M < 1 || I < 128 || (I % 128) || KN < 0 || KS < 0 || KO < 0 || (KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return 0;
    const int Kin[3] = {KN, KS, KO};
    if (mm::gate_up_act_stream_supported(M, 2 * I, Kin, true, false)) return M <= 2 ? 2 : 1;
    return mm_qlinear_decode_supported_w(M, 2 * I, KN, KS, KO, MM_W_FP4) ? 1 : 0;
}

int mm_rmsnorm_gate_up_activate_decode(const void *X_bf16, const void *norm_weight_bf16, float eps, const int16_t *reorder_index, const uint8_t *BN,
                                       const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS, const uint8_t *SFBO, int M,
                                       int I, int KN, int KS, int KO, int DN, int DS, int DO, int flags, uint8_t *oN, uint8_t *oS, uint8_t *oO,
                                       uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, void *workspace, size_t workspace_bytes, mm_stream_t stream) {
    if (M < 0 || I < 0 || KN < 0 || KS < 0 || KO < 0) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128) || KN + KS + KO == 0) return MM_ERR_BAD_SPLIT;
    if (!split_ok(DN + DS + DO, DN, DS, DO) || DN + DS + DO != I) return MM_ERR_BAD_SPLIT;
    if (flags & ~(MM_ROUND_ONCE | MM_NORM_NO_INTEGER_ROUND)) return MM_ERR_BAD_ARG;
    if (M == 0) return MM_OK;
    const int Kin[3] = {KN, KS, KO};
    if (!mm_rmsnorm_gate_up_activate_decode_supported(M, I, KN, KS, KO)) return MM_ERR_UNSUPPORTED;
    if (!X_bf16 || !norm_weight_bf16 || !reorder_index || ((uintptr_t)X_bf16 & 15) || ((uintptr_t)norm_weight_bf16 & 15)) return MM_ERR_BAD_ARG;
    if ((KN && (!BN || !SFBN)) || (KS && (!BS || !SFBS)) || (KO && (!BO || !SFBO))) return MM_ERR_BAD_ARG;
    if ((DN && (!oN || !sfN)) || (DS && (!oS || !sfS)) || (DO && (!oO || !sfO))) return MM_ERR_BAD_ARG;
    if (mm::gate_up_act_stream_supported(M, 2 * I, Kin, true, true)) {
        const mm::GemmArgs a = act_stream_args(BN, BS, BO, SFBN, SFBS, SFBO, M, I, KN, KS, KO, DN, DS, DO, flags, oN, oS, oO, sfN, sfS, sfO);
        const mm::NormArgs norm = {norm_weight_bf16, eps, (flags & MM_NORM_NO_INTEGER_ROUND) ? 0 : 1};
        hipError_t e = mm::launch_gate_up_act_stream_decode(X_bf16, reorder_index, a, (hipStream_t)stream, norm);
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_rmsnorm_gate_up_activate_decode");
    }
    // two launches: norm + quantize + gate | up GEMM into the scratch, then the activation quantizer on it (the same bytes)
    const int N = 2 * I;
    if (!workspace || workspace_bytes < (size_t)M * N * sizeof(uint16_t) || ((uintptr_t)workspace & 15)) return MM_ERR_BAD_ARG;
    const int st = mm_rmsnorm_qlinear_decode(X_bf16, norm_weight_bf16, eps, reorder_index, BN, BS, BO, SFBN, SFBS, SFBO, M, N, KN, KS, KO, MM_W_FP4,
                                             flags, nullptr, workspace, stream);
    if (st != MM_OK) return st;
    hipError_t e = mm::launch_direct_quantize(workspace, (const uint16_t *)workspace + 128, M, DN, DS, DO, 3, oN, oS, oO, sfN, sfS, sfO,
                                              (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_rmsnorm_gate_up_activate_decode");
}

int mm_down_activate_decode_supported_w(int M, int N, int DN, int DS, int DO, int wmode) {
    if (M < 1 || N < 1 || DN < 0 || DS < 0 || DO < 0 || (DN % 128) || (DS % 128) || (DO % 128) || DN + DS + DO == 0) return 0;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return 0;
    const int K[3] = {DN, DS, DO};
    if (!mm::down_activate_stream_supported(M, N, K, weights_fp4(wmode, DS, DO))) return 0;
    return M <= 2 ? 2 : 1;      // every workgroup repeats silu * up + the quantization: one pass of its threads up to M = 2 at I = 14336
}
int mm_down_activate_decode_supported(int M, int N, int DN, int DS, int DO) { return mm_down_activate_decode_supported_w(M, N, DN, DS, DO, (DS | DO) ? MM_W_MATCH : MM_W_FP4); }

int mm_down_activate_decode(const void *GU_bf16, const uint8_t *BN, const uint8_t *BS, const uint8_t *BO, const uint8_t *SFBN, const uint8_t *SFBS,
                            const uint8_t *SFBO, int M, int N, int DN, int DS, int DO, int wmode, int flags, const void *bias_bf16, void *D_bf16,
                            mm_stream_t stream) {
    if (M < 0 || N < 0 || DN < 0 || DS < 0 || DO < 0) return MM_ERR_BAD_ARG;
    if ((DN % 128) || (DS % 128) || (DO % 128) || DN + DS + DO == 0) return MM_ERR_BAD_SPLIT;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return MM_ERR_BAD_ARG;
    if (flags & ~MM_ROUND_ONCE) return MM_ERR_BAD_ARG;
    if (M == 0 || N == 0) return MM_OK;
    if (!mm_down_activate_decode_supported_w(M, N, DN, DS, DO, wmode)) return MM_ERR_UNSUPPORTED;
    if (!GU_bf16 || !D_bf16 || ((uintptr_t)GU_bf16 & 15)) return MM_ERR_BAD_ARG;
    if ((DN && (!BN || !SFBN)) || (DS && (!BS || !SFBS)) || (DO && (!BO || !SFBO))) return MM_ERR_BAD_ARG;
    const uint8_t *W[3] = {BN, BS, BO}, *SFW[3] = {SFBN, SFBS, SFBO};
    const int K[3] = {DN, DS, DO};
    hipError_t e = mm::launch_down_activate_stream(GU_bf16, W, SFW, M, N, K, weights_fp4(wmode, DS, DO), (flags & MM_ROUND_ONCE) ? 0 : 1, bias_bf16, D_bf16,
                                                   (hipStream_t)stream);
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_down_activate_decode");
}

int mm_matmul_ws_reset(void *workspace, size_t workspace_bytes, mm_stream_t stream) {
    if (!workspace || workspace_bytes < MM_WS_TICKET_BYTES || ((uintptr_t)workspace & 15)) return MM_ERR_BAD_ARG;
    static_assert(MM_WS_TICKET_BYTES % 16 == 0 && MM_WS_TICKET_BYTES / 16 <= 1024, "one workgroup clears the ticket words");
    hipLaunchKernelGGL(ws_reset_kernel, dim3(1), dim3(MM_WS_TICKET_BYTES / 16), 0, (hipStream_t)stream, (uint4 *)workspace);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MM_OK : fail_hip(e, "mm_matmul_ws_reset");
}

int mm_reorder_quantize_grouped(const mm_quant_group *groups, int ngroups, int K, int KN, int KS, int KO, int mode, mm_stream_t stream) {
    if (ngroups < 0 || (ngroups > 0 && !groups)) return MM_ERR_BAD_ARG;
    if (!split_ok(K, KN, KS, KO)) return MM_ERR_BAD_SPLIT;
    if (mode != MM_QUANT_MIXED && mode != MM_QUANT_W4) return MM_ERR_BAD_ARG;
    if (K > 32768) return MM_ERR_BAD_ARG;
    for (int i = 0; i < ngroups; ++i) {
        const mm_quant_group &g = groups[i];
        if (g.rows < 0) return MM_ERR_BAD_ARG;
        if (g.rows == 0) continue;
        if (!g.src_bf16 || !g.reorder_index || (KN && (!g.oN || !g.sfN)) || (KS && (!g.oS || !g.sfS)) || (KO && (!g.oO || !g.sfO)))
            return MM_ERR_BAD_ARG;
    }
    mm::GroupedQuantArgs ga;
    ga.K = K; ga.KN = KN; ga.KS = KS; ga.KO = KO;
    int count = 0, max_rows = 0;
    auto flush = [&]() -> int {
        if (count == 0) return MM_OK;
        ga.ngroups = count;
        hipError_t e = mm::launch_reorder_quantize_grouped(ga, max_rows, mode == MM_QUANT_W4, (hipStream_t)stream);
        count = 0;
        max_rows = 0;
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_reorder_quantize_grouped");
    };
    for (int i = 0; i < ngroups; ++i) {
        const mm_quant_group &g = groups[i];
        if (g.rows == 0) continue;
        mm::QuantArgs &q = ga.g[count];
        q.src = (const uint16_t *)g.src_bf16;
        q.idx = g.reorder_index;
        q.o[0] = g.oN; q.o[1] = g.oS; q.o[2] = g.oO;
        q.sf[0] = g.sfN; q.sf[1] = g.sfS; q.sf[2] = g.sfO;
        q.rows = g.rows;
        max_rows = g.rows > max_rows ? g.rows : max_rows;
        if (++count == mm::MM_MAX_GROUPS) {
            const int st = flush();
            if (st != MM_OK) return st;
        }
    }
    return flush();
}

int mm_matmul_grouped(const mm_group *groups, int ngroups, int N, int KN, int KS, int KO, int wmode, int flags, mm_stream_t stream) {
    if (ngroups < 0 || N < 0 || KN < 0 || KS < 0 || KO < 0 || (ngroups > 0 && !groups)) return MM_ERR_BAD_ARG;
    if ((KN % 128) || (KS % 128) || (KO % 128)) return MM_ERR_BAD_SPLIT;
    if (wmode != MM_W_MATCH && wmode != MM_W_FP4) return MM_ERR_BAD_ARG;
    if (flags & MM_OUT_F32) return MM_ERR_UNSUPPORTED;    // fp32 partial sums come from mm_matmul only
    if (ngroups == 0 || N == 0) return MM_OK;
    for (int i = 0; i < ngroups; ++i) {
        const mm_group &g = groups[i];
        if (g.M < 0) return MM_ERR_BAD_ARG;
        if (g.M == 0) continue;
        if (!g.D || (KN && (!g.AN || !g.BN || !g.SFAN || !g.SFBN)) || (KS && (!g.AS || !g.BS || !g.SFAS || !g.SFBS)) ||
            (KO && (!g.AO || !g.BO || !g.SFAO || !g.SFBO)))
            return MM_ERR_BAD_ARG;
    }
    // groups of at most 64 token rows share launches of the weight-streaming kernels, larger groups launches of the tiled
    // kernels (up to MM_MAX_GROUPS argument blocks per launch, carried in the kernel arguments)
    auto fill = [&](mm::GemmArgs &a, const mm_group &g) {
        a.X[0] = g.AN; a.X[1] = g.AS; a.X[2] = g.AO;
        a.W[0] = g.BN; a.W[1] = g.BS; a.W[2] = g.BO;
        a.SFX[0] = g.SFAN; a.SFX[1] = g.SFAS; a.SFX[2] = g.SFAO;
        a.SFW[0] = g.SFBN; a.SFW[1] = g.SFBS; a.SFW[2] = g.SFBO;
        a.K[0] = KN; a.K[1] = KS; a.K[2] = KO;
        a.M = g.M; a.N = N;
        a.sfx_row_tiles = (g.M + 127) / 128;
        a.sfw_row_tiles = (N + 127) / 128;
        a.round_per_segment = (flags & MM_ROUND_ONCE) ? 0 : 1;
        a.bias = (const uint16_t *)g.bias_bf16;
        a.D = (uint16_t *)g.D;
        a.out_f32 = 0;
        a.act = 0;
        a.clock_out = nullptr;
        a.ev_start = a.ev_stop = nullptr;
        a.ws = nullptr; a.ws_bytes = 0; a.splits = 0; a.force_split = 0; a.n_tile0 = a.n_tiles = 0;
        a.tickets = nullptr; a.tickets_zeroed = 0;
        a.split_first[0] = a.split_first[1] = a.split_first[2] = a.split_first[3] = 0;
    };
    if (KN + KS + KO == 0) {   // no segment: every output is zero (gemm.cu:48-50)
        for (int i = 0; i < ngroups; ++i)
            if (groups[i].M) {
                const int st = mm_matmul(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         nullptr, groups[i].M, N, 0, 0, 0, wmode, flags, nullptr, groups[i].D, stream);
                if (st != MM_OK) return st;
            }
        return MM_OK;
    }
    mm::GroupedGemmArgs small;
    mm::GroupedTileArgs big;
    int nsmall = 0, nbig = 0, max_m = 0;
    auto flush_small = [&]() -> int {
        if (nsmall == 0) return MM_OK;
        small.ngroups = nsmall;
        const int Ks[3] = {KN, KS, KO};
        hipError_t e = mm::mx_gemm_stream_grouped_supported(max_m, nsmall, N, Ks)
                           ? mm::launch_mx_gemm_stream_grouped(small, max_m, weights_fp4(wmode, KS, KO), (hipStream_t)stream)
                           : mm::launch_mx_gemm_skinny_grouped(small, max_m, weights_fp4(wmode, KS, KO), (hipStream_t)stream);
        nsmall = 0;
        max_m = 0;
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_matmul_grouped");
    };
    auto flush_big = [&]() -> int {
        if (nbig == 0) return MM_OK;
        big.ngroups = nbig;
        hipError_t e = mm::launch_mx_gemm256_grouped(big, weights_fp4(wmode, KS, KO), (hipStream_t)stream);
        nbig = 0;
        return e == hipSuccess ? MM_OK : fail_hip(e, "mm_matmul_grouped");
    };
    for (int i = 0; i < ngroups; ++i) {
        const mm_group &g = groups[i];
        if (g.M == 0) continue;
        if (g.M > 64) {
            fill(big.g[nbig], g);
            if (++nbig == mm::MM_MAX_GROUPS) {
                const int st = flush_big();
                if (st != MM_OK) return st;
            }
        } else {
            fill(small.g[nsmall], g);
            max_m = g.M > max_m ? g.M : max_m;
            if (++nsmall == mm::MM_MAX_GROUPS) {
                const int st = flush_small();
                if (st != MM_OK) return st;
            }
        }
    }
    const int st = flush_small();
    return st != MM_OK ? st : flush_big();
}

int mm_diag_set_kernel_events(void *start_event, void *stop_event) {
    g_ev.start = (hipEvent_t)start_event;
    g_ev.stop = (hipEvent_t)stop_event;
    return MM_OK;
}

#ifdef MM_INSTRUMENT
int mm_diag_set_quant_clock_buffer(void *buf) { return mm::set_quant_clock_buffer((unsigned long long *)buf) == hipSuccess ? MM_OK : MM_ERR_LAUNCH; }
// only in the instrumented variant (csrc/mx_instrument.h): the default library has neither this symbol nor the in-kernel stores
int mm_diag_set_clock_buffer(void *buf) {
    g_clock_buf = (unsigned long long *)buf;
    return MM_OK;
}
#endif

}  // extern "C"
