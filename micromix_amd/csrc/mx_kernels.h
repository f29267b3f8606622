// Host-side launch interface between capi.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

namespace mm {

// Per-device launcher state.  One process may drive several GPUs (the reference's --multi_gpu mode places the layers of ONE
// process on several devices, model/parallel_utils.py:135-156): a kernel's dynamic-LDS limit and the CU count belong to the
// device that is current at the launch, so both are cached per device id, never per process.
constexpr int MM_MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < MM_MAX_DEVICES) ? d : 0;
}
inline int device_cus() {
    static std::atomic<int> cus[MM_MAX_DEVICES];
    const int dev = current_device();
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}
// raises a kernel's dynamic shared memory limit once per device (calling it twice is harmless, so no lock)
struct DynamicLdsOnce {
    std::atomic<bool> done[MM_MAX_DEVICES];
    hipError_t ensure(const void *kernel, int bytes) {
        const int dev = current_device();
        if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done[dev].store(true, std::memory_order_release);
        return e;
    }
};

// resident workgroups per CU of `kernel` at (threads, dynamic LDS): the occupancy query costs ~1 us of host time per launch, so the
// last answer is kept per calling thread and kernel slot (a layer stack alternates between a handful of shapes)
struct OccupancyCache {
    struct Entry { const void *kernel; int dev, threads; size_t lds; int per_cu; };
    static int get(int slot, const void *kernel, int threads, size_t lds) {
        static thread_local Entry last[8] = {};
        Entry &e = last[slot & 7];
        const int dev = current_device();
        if (e.kernel == kernel && e.dev == dev && e.threads == threads && e.lds == lds && e.per_cu > 0) return e.per_cu;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        e = Entry{kernel, dev, threads, lds, per_cu};
        return per_cu;
    }
};

// Measurement hook (mm_diag_set_kernel_events): the calling thread's pair of events, both null unless registered.  The quantizer
// launchers (MM_LAUNCH) and the tiled GEMM (GemmArgs::ev_start / ev_stop) attach them to their dispatch, so that their elapsed time is
// the kernel's own duration -- what rocprofv3's kernel trace reports -- without the launch gap that events around a call include.
struct DiagEvents { hipEvent_t start, stop; };
DiagEvents &diag_events();     // thread-local, defined in capi.hip
#define MM_LAUNCH(kern, grid, block, lds, stream, ...)                                                                      \
    do {                                                                                                                   \
        const mm::DiagEvents &ev_ = mm::diag_events();                                                                     \
        if (ev_.start != nullptr && ev_.stop != nullptr)                                                                   \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ev_.start, ev_.stop, 0, __VA_ARGS__);                    \
        else                                                                                                               \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                               \
    } while (0)

constexpr int MM_MAX_SPLITS = 16;   // in-kernel split-K: splits per tile

struct GemmArgs {
    const uint8_t *X[3];    // activation segments  (AN, AS, AO)
    const uint8_t *W[3];    // weight segments      (BN, BS, BO)
    const uint8_t *SFX[3];  // activation scales
    const uint8_t *SFW[3];  // weight scales
    int K[3];
    int M, N;
    int sfx_row_tiles;      // allocated 128-row tiles in the activation SF tensors
    int sfw_row_tiles;
    int round_per_segment;
    const uint16_t *bias;   // optional [N] bf16
    uint16_t *D;            // [M, N] bf16 (out_f32 = 0)
    int out_f32;            // MM_OUT_F32: D is [M, N] fp32 instead -- the accumulator as it is, no rounding (tensor-parallel partial sums)
    float *ws;              // split-K workspace (NULL: never split), see mm_matmul_ws
    size_t ws_bytes;
    int force_split;        // MM_SPLIT_K_ALWAYS
    int n_tile0, n_tiles;   // launcher: this launch covers 256-feature tile columns [n_tile0, n_tile0 + n_tiles) (0, 0 = all)
    int splits;             // filled in by the launcher when it splits K
    int split_first[4];     // splits [split_first[i], split_first[i+1]) work on segment i
    unsigned short split_cut[MM_MAX_SPLITS + 1];   // in-kernel split-K: split q walks 128-deep K units [split_cut[q], split_cut[q+1]) of N | S | O
    unsigned short split_slot[MM_MAX_SPLITS + 1];  // ... and leaves its partial sums in slots split_slot[q] ... of its tile; [splits] = slots per tile
    unsigned long long slot_seg;    // ... two bits per slot: the segment (0 N, 1 S, 2 O) whose partial sum the slot holds
    unsigned *tickets;              // in-kernel split-K (4-wave tiles): one counter per tile at the head of the workspace, zero between launches
    int tickets_zeroed;             // MM_WS_TICKETS_ZEROED: the caller vouches for that
    // Fused gate / up epilogue (mm_gate_up_activate; launch_mx_gemm_act): W holds 2 * I rows, 128 gate features alternating with
    // the 128 up features of the same index, and a 256-feature tile quantizes silu(gate) * up of its 128 intermediate features as
    // activate_quantize_x does -- into act_o / act_sf, the operand tensors of the consumer GEMM whose K split is act_K -- instead of
    // writing D.
    int act;
    int act_K[3];
    uint8_t *act_o[3];
    uint8_t *act_sf[3];
    hipEvent_t ev_start, ev_stop;   // diagnostics only (mm_diag_set_kernel_events): recorded at the GEMM dispatch itself
    unsigned long long *clock_out;  // diagnostics only (mm_diag_set_clock_buffer): per workgroup {shader cycles, 100 MHz ticks}
};

constexpr int MM_MAX_GROUPS = 8;   // argument blocks of one grouped launch travel in the kernel arguments (8 x ~230 B)
struct GroupedGemmArgs {
    GemmArgs g[MM_MAX_GROUPS];
    int ngroups;
};

struct GroupedTileArgs {
    GemmArgs g[MM_MAX_GROUPS];
    int first_block[MM_MAX_GROUPS + 1];   // prefix sum of the groups' tile counts
    int ngroups;
};
hipError_t launch_mx_gemm256_grouped(GroupedTileArgs &ga, bool w4, hipStream_t stream);   // fills first_block[]

struct QuantArgs {
    const uint16_t *src;   // [rows, K] bf16
    const int16_t *idx;    // [KN + KS + KO]
    uint8_t *o[3];         // packed segments
    uint8_t *sf[3];        // scale tensors
    int rows;
};
struct GroupedQuantArgs {
    QuantArgs g[MM_MAX_GROUPS];
    int ngroups, K, KN, KS, KO;
};
hipError_t launch_reorder_quantize_grouped(const GroupedQuantArgs &ga, int max_rows, bool w4, hipStream_t stream);

hipError_t set_quant_clock_buffer(unsigned long long *buf);   // -DMM_INSTRUMENT only
hipError_t launch_reorder_quantize(const void *src, int rows, int K, const int16_t *idx, int KN, int KS, int KO, bool w4,
                                   uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO,
                                   hipStream_t stream);
// mode: 0 silu(A) * B, 1 A (mixed), 2 A (all fp4), 3 = 0 with A | B interleaved per 128 columns in one [rows, 2 K] matrix
hipError_t launch_direct_quantize(const void *A, const void *B, int rows, int KN, int KS, int KO, int mode, uint8_t *oN,
                                  uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO, hipStream_t stream);
hipError_t launch_rmsnorm_quantize(const void *src, const void *weight, float eps, int rows, int K, const int16_t *idx, int KN,
                                   int KS, int KO, bool integer_round, uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN,
                                   uint8_t *sfS, uint8_t *sfO, hipStream_t stream);
// RMSNorm in front of the quantization of the decode launches (mm_rmsnorm_qlinear_decode): weight == nullptr means no norm
struct NormArgs { const void *weight; float eps; int int_round; };
constexpr NormArgs NO_NORM = {nullptr, 0.0f, 1};
int qlinear_decode_supported(int M, int N, const int K[3], bool rms = false, bool w4 = false);
hipError_t launch_qlinear_decode(const void *X, const int16_t *idx, const uint8_t *const W[3], const uint8_t *const SFW[3],
                                 int M, int N, const int K[3], bool w4, int round_per_segment, const void *bias, void *D,
                                 hipStream_t stream, const NormArgs &norm = NO_NORM);
hipError_t launch_mx_gemm(const GemmArgs &a, bool w4, hipStream_t stream);
hipError_t launch_mx_gemm256(const GemmArgs &a, bool w4, hipStream_t stream);
bool mx_gemm_act_supported(int M, int N);                                 // the tiled kernels with the fused gate / up epilogue take this shape
hipError_t launch_mx_gemm_act(const GemmArgs &a, hipStream_t stream);    // fp4 weights only
const char *describe_mx_gemm_act(int M, int N);
hipError_t launch_mx_gemm_skinny(const GemmArgs &a, bool w4, hipStream_t stream);
// second-generation weight-streaming kernel (mx_gemm_stream.hip): M <= 64 (one to four 16-token tiles)
bool mx_gemm_stream_supported(int M, int N, const int K[3], bool w4);
hipError_t launch_mx_gemm_stream(const GemmArgs &a, bool w4, hipStream_t stream);
bool mx_gemm_stream_grouped_supported(int max_m, int ngroups, int N, const int K[3]);
hipError_t launch_mx_gemm_stream_grouped(const GroupedGemmArgs &ga, int max_m, bool w4, hipStream_t stream);
// ... with silu(gate) * up + its quantization inside every workgroup (mm_down_activate_decode)
bool down_activate_stream_supported(int M, int N, const int K[3], bool w4 = false);   // w4: the ring / reduction tail of fp4 weights is 48 KB, not 64
hipError_t launch_down_activate_stream(const void *GU, const uint8_t *const W[3], const uint8_t *const SFW[3], int M, int N, const int K[3],
                                       bool w4, int round_per_segment, const void *bias, void *D, hipStream_t stream);
// ... with the quantization of the M <= 8 activation rows inside every workgroup (mm_qlinear_decode)
bool qlinear_stream_supported(int M, int N, const int K[3], bool rms = false, bool w4 = false);
hipError_t launch_qlinear_stream(const void *X, const int16_t *idx, const uint8_t *const W[3], const uint8_t *const SFW[3], int M, int N,
                                 const int K[3], bool w4, int round_per_segment, const void *bias, void *D, hipStream_t stream,
                                 const NormArgs &norm = NO_NORM);
// ... the fused gate | up weight (fp4) with silu(gate) * up and the quantization for the consumer inside the launch (a.act_K / act_o / act_sf;
// round 6): from the quantized activations (M <= 16) or, `X` / `idx` / `norm`, from the bf16 rows (M <= 4)
bool gate_up_act_stream_supported(int M, int N, const int K[3], bool from_bf16, bool rms);
hipError_t launch_gate_up_act_stream(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gate_up_act_stream_decode(const void *X, const int16_t *idx, const GemmArgs &a, hipStream_t stream, const NormArgs &norm = NO_NORM);
hipError_t launch_mx_gemm_skinny_grouped(const GroupedGemmArgs &ga, int max_m, bool w4, hipStream_t stream);
size_t mx_gemm_workspace_bytes(int M, int N, const int K[3], bool w4, bool force, bool tickets_zeroed);
bool mx_gemm_small_m_uses_tiles(int M, int N, const int K[3], bool w4, size_t ws_bytes, bool force_split);
const char *describe_mx_gemm256(int M, int N, const int K[3], bool w4, size_t ws_bytes, bool force_split, bool tickets_zeroed);   // thread-local buffer  // 0 when mm_matmul would not split K for this shape

}  // namespace mm
