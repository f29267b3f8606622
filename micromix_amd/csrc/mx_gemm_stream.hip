// Weight-streaming three-segment MX GEMM for M <= 64 on gfx950, second generation (round 4): the decode / small-batch path of
// mm_matmul, and -- with the quantization of the activation rows inside every workgroup -- of mm_qlinear_decode.  Same arithmetic and
// reference citations as mx_gemm_skinny.hip (gemm.cu:26-78: D = bf16(N); D = bf16(S + D); D = bf16(O + D), fp32 accumulation inside
// a segment).
//
// Why a second kernel.  A launch that ONLY streams the 29 MB of gate_proj's packed weights takes 3.7 us with lane-contiguous loads
// and 4.9 us with the first kernel's pattern (lane = weight row: 64 separate 16-byte requests per instruction; tools/stream_floor.py;
// the weights of one layer stay in the Infinity Cache between back-to-back launches).  The first weight-streaming kernel needed
// 11.5 us at M = 16.  Ablations on a register-ring version of it (profiles/r04_stream_ablation.txt) showed that the time was the sum
// of its vector-memory instructions, at 30-45 cycles each whatever they fetched -- weights 2.8 us, activations 3.3 us, the three scale
// dwords per slab 2.7 us -- and of its instruction stream, not a chain of round trips.  So here EVERY load is lane-contiguous and the
// MFMA's (lane = row) operand layout is made in LDS:
//  * one workgroup = 16 F output features x all tokens (T16 tiles of 16), its NW waves split K: the 128-deep slabs of N | S | O are
//    numbered through (j = 0 .. T-1) and slab j goes to wave j % NW, whatever segment it belongs to;
//  * a slab's operand tile (16 rows x C 16-byte chunks per row; C = 4 / 6 / 8 for fp4 / fp6 / fp8) goes global -> LDS directly
//    (buffer_load_dwordx4 ... lds, inline asm: no staging registers, no ds_write) into the wave's PRIVATE ring of D slots.  A DMA
//    instruction writes its 64 x 16 bytes to LDS in lane order, so WHICH chunk a lane fetches decides the LDS image: lane p = C q + c
//    of a piece fetches chunk c ^ s(row) of row q (s = row >> 2 for 4 chunks per row, (row >> 1) & 7 for 8; fp6 tiles stay row-major):
//    four / eight consecutive lanes cover one row's contiguous bytes, and the 16 lanes of a ds_read_b128 phase (16 rows, same chunk)
//    hit 16 different bank groups.  The scale bytes of the slab's rows are one 8-byte load per lane over each 512-byte atom, handed to
//    the lanes that need them by ds_bpermute (as one dword per lane at a stride of 8 bytes the same 512 bytes cost the launch 2 us),
//    issued before the tiles (they come from L2 and would otherwise queue behind the weights);
//  * every wave keeps D slabs in flight, issued and consumed in slab order in whole rounds of D steps with the same number L of
//    vector-memory instructions per step, so a counted s_waitcnt vmcnt((D - 1) L) before a step leaves exactly the D - 1 younger
//    slabs outstanding (a wave with cnt slabs starts with (D - cnt % D) % D phantom steps that re-request its first slab and skip the
//    arithmetic).  No barrier in the loop: LDS-DMA completion is what vmcnt counts.  More waves with a shallow ring (8 x 2) beat
//    fewer with a deep one; the same tiles through a register ring + ds_write measured the same;
//  * instruction count is the currency: a wave64 VALU instruction holds its SIMD for 4 cycles, a 5 us launch leaves a wave a few
//    hundred of them.  Everything that depends only on (segment, lane) -- descriptors, byte offsets, LDS addresses -- is computed
//    once; a step selects by segment with three typed code paths; the accumulators live in asm-owned AGPRs (through the builtin hipcc
//    copied the three accumulator sets at the joins of the segment branch);
//  * consuming a slab = reading the MFMA fragments (lane (row l & 15, K block l >> 4)) and v_mfma_scale_f32_16x16x128_f8f6f4 with the
//    tokens on the rows; each segment has its own accumulators (a 16 x 16 fp32 tile is 4 registers);
//  * ONE barrier (eight tiles per wave: one per segment): every wave leaves its partial sums in LDS, then thread o sums the partials of
//    output o segment by segment and applies the reference's rounding chain on the reduced values.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "mx_common.h"
#include "mx_decode_quant.h"
#include "mx_kernels.h"

namespace mm {
namespace stream {

// hipcc parses __device__ bodies in its host pass as well; gfx950 inline asm only exists in the device pass
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_DEVICE_ONLY(...)
#endif

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int rsrc_t __attribute__((ext_vector_type(4)));

// kernel-developer ablations (tools/build_one_variant.sh): 1 no activation loads, 2 no scale loads, 4 no MFMAs, 8 no reduction / store,
// 16 no weight loads
#ifndef MM_STREAM_DBG
#define MM_STREAM_DBG 0
#endif
#ifndef MM_STREAM_ACCZERO_EARLY
#define MM_STREAM_ACCZERO_EARLY 0
#endif


// launches without the quantization that take their scales from scale images (see stream_body): one token tile, and two with fp4 weights
__host__ __device__ constexpr bool scale_images_for(int T16, bool W4) { return T16 == 1 || (T16 == 2 && W4); }

// 128-bit raw buffer descriptor {base_lo, base_hi(16 bits) | stride 0, num_records (bytes), flags}, every word provably
// wave-uniform so that it can be bound to an "s" operand
__device__ __forceinline__ rsrc_t make_rsrc(const uint8_t *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// LDS byte address of a pointer into the workgroup's LDS (the low half of the flat address; see mx_gemm_tile.inc)
__device__ __forceinline__ unsigned lds_address(const uint8_t *p) { return (unsigned)(unsigned long long)p; }

// one buffer_load_dwordx4 ... lds: 64 lanes x 16 B -> LDS bytes [lds, lds + 1024) in lane order; per-lane source = base + voff + soff
// (s_nop 4: SALU results may not be read by a VMEM instruction for 5 states; s_nop 0: one state between the M0 write and the DMA).
// (The same tiles through registers -- buffer_load_dwordx4 into a register ring, ds_write_b128 in lane order when a slab is consumed --
// measured the same times at 118 instead of 53 VGPRs: profiles/r04_stream_ablation.txt, section 6.)
__device__ __forceinline__ void dma16(const rsrc_t &rsrc, int voff, int soff, unsigned lds) {
    MM_DEVICE_ONLY(unsigned keep;
                   asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                                : "=&s"(keep)
                                : "v"(voff), "s"(rsrc), "s"(lds), "s"(soff)
                                : "memory");)
}
// 8 bytes per lane into registers, NOT tracked by the compiler (the counted waits below order it).  Lane-contiguous on purpose:
// the same 512 bytes as one dword per lane at a stride of 8 bytes cost the launch 1.5-2.3 us (profiles/r04_stream_ablation.txt).
__device__ __forceinline__ v2i load_atom(const rsrc_t &rsrc, int voff, int soff) {
    v2i d = {0, 0};
    MM_DEVICE_ONLY(asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(d) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");)
    return d;
}

// a slab's two 512-byte scale atoms, 8 bytes per lane: lane l holds, of atom row l >> 1, the row groups 2 (l & 1) and 2 (l & 1) + 1
// (four block scales each)
struct Slot { v2i sw, sx; };

template <int N>
__device__ __forceinline__ void wait_slot(Slot &q) {     // at most N vector-memory instructions outstanding; q's registers are valid after it
    MM_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q.sw), "+v"(q.sx) : "n"(N) : "memory");)
}

typedef int v6i __attribute__((ext_vector_type(6)));
template <int EL> struct Frag;                                    // the registers a lane holds of a 128-deep operand row
template <> struct Frag<0> { typedef v4i type; static constexpr int HW = HW_FP4; };
template <> struct Frag<1> { typedef v6i type; static constexpr int HW = HW_BF6; };
template <> struct Frag<2> { typedef v8i type; static constexpr int HW = HW_FP8; };

// The accumulators live in a[0 : 12 F T16 - 1], outside the compiler's register allocation (as in mx_gemm256.hip): tile i of segment
// g is a[4 (g F T16 + i) .. + 3].  Through the builtin -- or through asm on compiler-owned registers -- hipcc copied the three
// accumulator sets at the joins of the segment branch (8-16 moves per step).  s_nop 1: two wait states between a just-written
// source VGPR (the scale shift) and the MFMA -- hipcc pads nothing in front of an asm statement.  Scale bytes: byte 0 of sx / sw.
#define MM_STREAM_ACC12 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11"
#define MM_STREAM_ACC24 MM_STREAM_ACC12,"a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23"
#define MM_STREAM_ACC48 MM_STREAM_ACC24,"a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"
#define MM_STREAM_ACC96 MM_STREAM_ACC48,"a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71", \
                        "a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95"
#define MM_STREAM_ACC192 MM_STREAM_ACC96,"a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119", \
                         "a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143", \
                         "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167", \
                         "a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"
// (NACC = 12 F T16 accumulator registers: the clobber list names exactly those, the rest of the register file stays the compiler's)
#define MM_STREAM_ASM_ACC(NACC, ...)                                                  \
    do {                                                                              \
        if constexpr ((NACC) <= 12) { MM_DEVICE_ONLY(asm volatile(__VA_ARGS__ : MM_STREAM_ACC12);) }       \
        else if constexpr ((NACC) <= 24) { MM_DEVICE_ONLY(asm volatile(__VA_ARGS__ : MM_STREAM_ACC24);) }  \
        else if constexpr ((NACC) <= 48) { MM_DEVICE_ONLY(asm volatile(__VA_ARGS__ : MM_STREAM_ACC48);) }  \
        else if constexpr ((NACC) <= 96) { MM_DEVICE_ONLY(asm volatile(__VA_ARGS__ : MM_STREAM_ACC96);) }  \
        else { MM_DEVICE_ONLY(asm volatile(__VA_ARGS__ : MM_STREAM_ACC192);) }                             \
    } while (0)
template <int XEL, int WEL, int TILE, int NACC>
__device__ __forceinline__ void mfma16(const typename Frag<XEL>::type &x, const typename Frag<WEL>::type &w, int sx, int sw) {
    static_assert(4 * TILE + 3 < NACC && NACC <= 192, "a[0:191]");
    MM_STREAM_ASM_ACC(NACC, "s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 a[%c6:%c7], %0, %1, a[%c6:%c7], %2, %3 op_sel_hi:[0,0,0] cbsz:%c4 blgp:%c5"
                      :
                      : "v"(x), "v"(w), "v"(sx), "v"(sw), "i"(Frag<XEL>::HW), "i"(Frag<WEL>::HW), "i"(4 * TILE), "i"(4 * TILE + 3));
}
template <int REG, int NACC>
__device__ __forceinline__ void acc_zero() { MM_STREAM_ASM_ACC(NACC, "v_accvgpr_write_b32 a[%c0], 0" : : "i"(REG)); }
template <int REG>
__device__ __forceinline__ float acc_read() {
    float r = 0.0f;
    MM_DEVICE_ONLY(asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "i"(REG));)
    return r;
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&fn) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (fn(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// one slot of a wave's LDS ring: F weight tiles (one 1 KB piece each with fp4 weights, two otherwise), T16 activation tiles of two pieces
// (QUANT: the workgroup quantizes the activation rows itself, mx_decode_quant.h -- no activation tiles, no activation scale atom.  QUANT or
// one token tile: the scales come from the wave's scale images, see scale_image_bytes: no scale atoms per slab)
template <int F, int T16, bool W4, bool QUANT = false>
struct Ring {
    static constexpr int WP = W4 ? 1 : 2;
    static constexpr int W_BYTES = F * WP * 1024, X_BYTES = QUANT ? 0 : T16 * 2048, SLOT = W_BYTES + X_BYTES;
    // vector-memory instructions of a slab, at least: the second piece of an activation tile is requested only where it holds rows the
    // launch has (fp4 tiles are one piece; M <= 8: rows 0 .. 7 sit in the first piece of every format).  The counted waits use this
    // minimum -- with longer slabs behind it a wait lets at most one instruction fewer stay in flight, never one too many.
    static constexpr int LOADS = ((MM_STREAM_DBG & 16) ? 0 : F * WP) + ((MM_STREAM_DBG & 1) || QUANT ? 0 : T16) + ((MM_STREAM_DBG & 2) || QUANT || scale_images_for(T16, W4) ? 0 : 2);
};

// chunks per row of a segment's 128-deep slab: fp4 4, fp6 6, fp8 8 (x 16 bytes)
__device__ __forceinline__ constexpr int chunks_of(int g) { return g == 0 ? 4 : (g == 1 ? 6 : 8); }

#ifndef MM_STREAM_CLOCK     // kernel-developer build: wave 0 of every workgroup leaves 100 MHz timestamps (start, ring primed, loop done, barrier passed, end)
#define MM_STREAM_CLOCK 0
#endif
#if MM_STREAM_CLOCK
__device__ unsigned long long *g_stream_clock;
#define MM_STAMP(i) do { if (threadIdx.x == 0 && g_stream_clock) g_stream_clock[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MM_STAMP(i) do { } while (0)
#endif

// QUANT (mm_qlinear_decode, M <= 8): X / SFX of `a` are unused; the workgroup quantizes the bf16 rows of `qi` into LDS itself, after
// its first batch of rows is staged it requests its first D slabs of weights, and the activation fragments and scales of a slab come
// from that LDS copy (rows in the reference's packed layout, as in qlinear_decode.hip).  `qbytes` = the quantization's LDS range in
// front of the rings.
// QUANT kernels: the scale image.  A slab's 512-byte scale atom carries the scales of 128 weight rows; a workgroup wants 16 F of them, and
// a launch of these kernels is the sum of its vector-memory instructions (30-45 cycles each on a CU, whatever they fetch): with one atom
// load per slab the scales cost as much as the fp4 weights themselves (down_proj at M = 1, tools/stream_clock.py: loop 5.2 us, without the
// scale loads 2.8; the whole launch 11.0 -> 8.6).  So a wave gathers the scale dwords of ALL of its slabs in front of everything else --
// one dword per lane = (slab, weight row), 64 / (16 F) slabs per instruction, global -> LDS directly -- into a private image
// [instruction k][lane] of dwords, and a step reads its scale bytes from there.
__host__ __device__ constexpr int scale_image_wave_bytes(int F, int NW, int T) {      // T = 128-deep slabs of the launch
    const int spi = 4 / F, per_wave = (T + NW - 1) / NW;
    return (per_wave + spi - 1) / spi * 256;
}
__host__ __device__ constexpr int scale_image_bytes(int F, int NW, int T) { return NW * scale_image_wave_bytes(F, NW, T); }
// both images of a launch without the quantization (weights' rows 16 F, activations' 16 T16): one token tile only, see stream_body
__host__ __device__ constexpr int scale_images_bytes(int F, int T16, int NW, int T, bool W4) {
    return scale_images_for(T16, W4) ? scale_image_bytes(F, NW, T) + scale_image_bytes(T16, NW, T) : 0;
}
constexpr int STREAM_LDS_MAX = 160 * 1024;      // (the 64-token configuration on 16 features uses all of a CU's LDS)
// one global_load_lds_dword: lane l's dword at `p` -> LDS byte lds + 4 l (M0 = lds; counted by vmcnt like the ring's DMA)
// (M0 is saved and restored by the caller, once around its loop: m0_save / m0_restore)
__device__ __forceinline__ void dma4(const uint8_t *p, unsigned lds) {
    MM_DEVICE_ONLY(asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(p), "s"(lds) : "memory");)
}
__device__ __forceinline__ unsigned m0_save() {
    unsigned keep = 0;
    MM_DEVICE_ONLY(asm volatile("s_mov_b32 %0, m0" : "=s"(keep) : : "memory");)
    return keep;
}
__device__ __forceinline__ void m0_restore(unsigned keep) { MM_DEVICE_ONLY(asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");) }

// slots per thread of dq::quantize_rows_early (see there): 2 on the four-wave kernels, whose register count costs no resident workgroup
template <int NW> constexpr int EARLY_NPASS = NW <= 4 ? 2 : 1;
// ACT (round 6; F = 4, one token tile, fp4 weights): the fused gate | up weight of mm_gate_up_activate -- 128 gate rows alternating with the
// 128 up rows of the same indices -- with the activation INSIDE the launch.  Workgroup b takes the 32 gate rows 256 (b / 4) + 32 (b % 4) ..
// and the 32 up rows 128 further on (its four 16-row tiles: gate, gate, up, up), i.e. both halves of ONE 32-column group of the
// intermediate activation: after the reduction it computes h = silu(gate) * up on the bf16-rounded outputs (activate.cu:44-202, the
// arithmetic of direct_quantize.hip), quantizes the group as mm_activate_quantize does and writes the consumer's (down_proj's) operand
// bytes and scale byte -- a.act_o / a.act_sf / a.act_K -- instead of D.  The same bytes as GEMM -> mm_activate_quantize, one launch less,
// and down_proj becomes a plain mm_matmul (M = 1: 7.3 us against 9.2 for mm_down_activate_decode).
template <int F, int T16, int D, int NW, bool W4, bool QUANT = false, bool RMS = false, bool ACT = false>
__device__ __forceinline__ void stream_body(const GemmArgs &a, const dq::QuantIn &qi = dq::QuantIn(), int qbytes = 0) {
    static_assert(T16 <= 4, "64 token rows: row groups 0 and 1 of the activation scale atoms, both in the 8 bytes a lane loads");
    static_assert(!QUANT || T16 == 1, "M <= 8");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];  // [QUANT: staged rows | quantized rows | scale bytes] the waves' rings [NW][D][Ring::SLOT]; then the reduction image over the rings
    uint8_t *const smem = smem_all + qbytes;      // (QUANT: the scale image sits behind the rings, under the part of the reduction image they leave free)
    MM_STAMP(0);
    using RG = Ring<F, T16, W4, QUANT>;
    static_assert((D - 1) * RG::LOADS < 64, "vmcnt is a 6-bit counter");
    constexpr int BN = 16 * F, ACC = F * T16, NT = 64 * NW;
    static_assert(!ACT || (F == 4 && T16 <= 2 && W4 && (T16 == 1 || (NW == 4 && !QUANT))), "gate, gate, up, up");
    // first weight row of the workgroup; tile f starts trow(f) rows further on
    const int n0 = ACT ? 256 * ((int)blockIdx.x >> 2) + 32 * ((int)blockIdx.x & 3) : (int)blockIdx.x * BN;
    auto trow = [](int f) { return ACT ? (f & 1) * 16 + (f >> 1) * 128 : 16 * f; };
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, h = lane >> 4, sh = 8 * h;
    const int ns[3] = {a.K[0] >> 7, a.K[1] >> 7, a.K[2] >> 7};
    const int c1 = ns[0], c2 = ns[0] + ns[1], T = c2 + ns[2];
    const int wrows = ACT ? 160 : (a.N - n0 > BN ? BN : a.N - n0);       // (ACT: N is a multiple of 256: rows n0 .. n0 + 159 exist)
    const bool half_tile = T16 == 1 && a.M <= 8;        // rows 0 .. 7 of a tile are in its first piece: the second is never requested (rows
                                                        // 8 .. 15 of the LDS image then hold stale bytes; their outputs are never stored)
    const unsigned ring = __builtin_amdgcn_readfirstlane(lds_address(smem) + wave * D * RG::SLOT);   // LDS byte address of slot 0
    const uint8_t *const ringp = smem + wave * D * RG::SLOT;

    // this wave's slabs: j = wave + NW * i, i = 0 .. cnt-1
    const int cnt = wave < T ? (T - wave + NW - 1) / NW : 0;
    // The wave's scale images (see scale_image_bytes): QUANT kernels and, without the quantization, launches of one token tile (M <= 16:
    // there the two atoms were two of a slab's five or six vector-memory instructions; with more token tiles the images would cost the
    // second workgroup its place in the CU's LDS).  The weights' image [k][lane = (slab 4 / F k + k', weight row)], then -- not QUANT -- the
    // activations' [k][lane = (slab 4 k + k', token row)], behind the rings.
    // QUANT: WHEN the image is requested matters: a CU's memory pipe returns its waves' loads in issue order and these come from HBM / the
    // Infinity Cache.  In front of the activation rows (all waves at once, at the start) the quantization phase grew by 0.5 us; behind a
    // wave's rows but possibly in front of a later wave's by 0.8-1.3 (tools/stream_clock.py).  So: once the wave's own rows have landed
    // (the early reorder phase's `landed` hook; fused gate + up at M = 1 12.8 -> 12.0 us against in front), in front for the other phases
    // (below); the loop starts with one wait for everything requested under the phase.  Not QUANT: in front of the ring's first slabs.
    constexpr bool SIMG = (QUANT || scale_images_for(T16, W4)) && !(MM_STREAM_DBG & 2);
    [[maybe_unused]] const uint8_t *simg = nullptr, *simgx = nullptr;       // + 64 F i + 64 f (+ 64 i): the scale byte of (slab i, this lane's row and K block) -- see consume_g
    auto request_scales = [&]() {
      if constexpr (SIMG) {
        const unsigned keep_m0 = m0_save();
        // one image: `rows16` 16-row tiles per slab, image row r = row row_of(r) of the tensor (a segment's atoms: [row >> 7][slab]); segment bases b0 .. b2
        auto gather = [&](auto rows16_, const uint8_t *img, auto row_of, const uint8_t *b0, const uint8_t *b1, const uint8_t *b2) {
            constexpr int R = 16 * decltype(rows16_)::value, SPI = 64 / R;
            const unsigned img_lds = __builtin_amdgcn_readfirstlane(lds_address(img));
            const int qd = lane / R, n = row_of(lane % R), tile = n >> 7;
            const int aoff = (n & 31) * 16 + ((n >> 5) & 3) * 4;      // (atom row n & 31, row group (n >> 5) & 3): the dword of row n
            // (the three pointers as opaque scalars: hipcc otherwise turns the per-lane select between them into a per-lane LOAD from the
            // kernel-argument segment, with a wait for every outstanding load behind it, in every trip of the loop; the slab counts
            // likewise: the select between ns[0 .. 2] became an index into a copy of the array in scratch)
            int ns0 = ns[0], ns1 = ns[1], ns2 = ns[2];
            MM_DEVICE_ONLY(asm volatile("" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(ns0), "+s"(ns1), "+s"(ns2));)
            const uint8_t *const safe = ns0 ? b0 : (ns1 ? b1 : b2);
#pragma unroll 1
            for (int k = 0; k * SPI < cnt; ++k) {
                const int i = k * SPI + qd, j = wave + NW * i;
                // (selects, not arrays indexed by the lane's segment: those would live in scratch)
                const bool g0 = j < c1, g1 = j < c2;
                const int sl = j - (g0 ? 0 : (g1 ? c1 : c2)), nsg = g0 ? ns0 : (g1 ? ns1 : ns2);
                const uint8_t *base = g0 ? b0 : (g1 ? b1 : b2);
                // (slabs past the wave's last: any valid address; their dwords are never read)
                dma4(i < cnt ? base + ((size_t)tile * nsg + (size_t)sl) * 512 + aoff : safe, img_lds + k * 256);
            }
        };
        const int wimg = scale_image_wave_bytes(F, NW, T);
        const uint8_t *img = smem + NW * D * RG::SLOT;
        gather(std::integral_constant<int, F>{}, img + wave * wimg, [&](int r) { return n0 + trow(r >> 4) + (r & 15); }, a.SFW[0], a.SFW[1], a.SFW[2]);
        if constexpr (!QUANT) {
            const int ximg = scale_image_wave_bytes(T16, NW, T);
            gather(std::integral_constant<int, T16>{}, img + NW * wimg + wave * ximg, [](int r) { return r; }, a.SFX[0], a.SFX[1], a.SFX[2]);
        }
        m0_restore(keep_m0);
      }
    };
    if constexpr (SIMG) {
        const int wimg = scale_image_wave_bytes(F, NW, T);
        simg = smem + NW * D * RG::SLOT + wave * wimg + 4 * li + h;
        if constexpr (!QUANT) simgx = smem + NW * D * RG::SLOT + NW * wimg + wave * scale_image_wave_bytes(T16, NW, T) + 4 * li + h;
    }

    // ---- everything that depends on (segment, lane) only ----
    // piece k of a 16-row tile with C chunks per row: lane p fetches (row, chunk); its 16 bytes land at LDS byte 1024 k + 16 p
    //   C = 4 (one piece):   row p >> 2,       chunk (p & 3) ^ (row >> 2)
    //   C = 8 (two pieces):  row 8 k + (p >> 3), chunk (p & 7) ^ ((row >> 1) & 7)
    //   C = 6 (1.5 pieces):  chunk e = 64 k + p of the tile in row-major order (rows of 96 bytes stay contiguous in LDS), nothing past e = 95
    constexpr int OOB = 0x7FFFFF00;     // a byte offset past every descriptor's range: the lane fetches nothing (the DMA writes zeros)
    const int e1 = 64 + lane;
    const int r4 = lane >> 2, r8 = lane >> 3;
    const int ra[3] = {r4, lane / 6, r8}, ca[3] = {(lane & 3) ^ (r4 >> 2), lane % 6, (lane & 7) ^ ((r8 >> 1) & 7)};
    const int rb[3] = {0, e1 / 6, 8 + r8}, cb[3] = {0, e1 % 6, (lane & 7) ^ (((8 + r8) >> 1) & 7)};
    const bool vb[3] = {false, lane < 32, true};
    rsrc_t rw[3], rx[3], rsw[3], rsx[3];
    int wva[3], wvb[3], xva[3], xvb[3], wpitch[3], xpitch[3];               // byte offsets of this lane's chunks in tile 0, slab 0
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int xc = chunks_of(g), wc = W4 ? 4 : xc, gw = W4 ? 0 : g;
        xpitch[g] = ns[g] * xc * 16;
        wpitch[g] = ns[g] * wc * 16;
        // (the workgroup's own rows as the descriptor's range: rows past N read as zeros, offsets stay small)
        rw[g] = make_rsrc(a.W[g] + (size_t)n0 * (size_t)wpitch[g], (unsigned)wrows * (unsigned)wpitch[g]);
        rsw[g] = make_rsrc(a.SFW[g], (unsigned)a.sfw_row_tiles * (unsigned)ns[g] * 512u);
        if constexpr (!QUANT) {
            rx[g] = make_rsrc(a.X[g], (unsigned)a.M * (unsigned)xpitch[g]);
            rsx[g] = make_rsrc(a.SFX[g], (unsigned)a.sfx_row_tiles * (unsigned)ns[g] * 512u);
        }
        wva[g] = ra[gw] * wpitch[g] + ca[gw] * 16;
        wvb[g] = vb[gw] ? rb[gw] * wpitch[g] + cb[gw] * 16 : OOB;
        xva[g] = ra[g] * xpitch[g] + ca[g] * 16;
        xvb[g] = vb[g] ? rb[g] * xpitch[g] + cb[g] * 16 : OOB;
    }
    // scale atoms: the dword at byte 8 l + 4 p of a slab's 512-byte atom is (atom row l >> 1, row group 2 (l & 1) + p), its four bytes
    // the scales of the slab's four 32-blocks.  The workgroup's features share one scale tile (n0 >> 7).
    const int sf_lane = lane * 8;
    bool sfw_hi[F];                                                       // wave-uniform: the feature tile's row group is odd
    int sfw_src[F], sfx_src[T16];                                         // ds_bpermute byte index of the lane that loaded (row, row group)
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int rg = ((n0 + 16 * f) >> 5) & 3;
        sfw_hi[f] = (rg & 1) != 0;
        sfw_src[f] = 4 * (2 * ((n0 + 16 * f + li) & 31) + (rg >> 1));
    }
#pragma unroll
    for (int t = 0; t < T16; ++t) sfx_src[t] = 4 * (2 * ((16 * t + li) & 31));   // token row 16 t + li: atom row (.. & 31), row group t >> 1
    // fragment reads: lane (row li, K block h) holds 16 B at chunk h (fp4), 24 B at byte 24 h (fp6), chunks h and 4 + h (fp8) of its row
    const int s8 = (li >> 1) & 7;
    const int rd4 = (4 * li + (h ^ (li >> 2))) * 16, rd6 = li * 96 + 24 * h;
    const int rd8a = (li >> 3) * 1024 + (8 * (li & 7) + (h ^ s8)) * 16, rd8b = (li >> 3) * 1024 + (8 * (li & 7) + ((4 + h) ^ s8)) * 16;

    static_assert(12 * ACC <= 192, "a[0:191]");
#if MM_STREAM_ACCZERO_EARLY       // A/B builds only (round 4's placement: in front of the quantization phase -- not safe with the norm inside, see below)
    static_for<12 * ACC>([&](auto r_) { acc_zero<decltype(r_)::value, 12 * ACC>(); });
#endif

    // slab s of segment G into slot d: RG::LOADS vector-memory instructions, whatever the segment
    auto issue_g = [&](Slot &q, int d, auto G_, int s) {
        constexpr int G = decltype(G_)::value, XC = chunks_of(G), WC = W4 ? 4 : XC;
        const unsigned base = ring + d * RG::SLOT;
        // the scale atoms first: they come from L2 and would otherwise queue behind the slab's weight tiles
        if constexpr (!(MM_STREAM_DBG & 2)) {
            if constexpr (!SIMG) {
                q.sw = load_atom(rsw[G], sf_lane, (s + (n0 >> 7) * ns[G]) * 512);
                q.sx = load_atom(rsx[G], sf_lane, s * 512);
            }
        }
        if constexpr (!(MM_STREAM_DBG & 16)) {
#pragma unroll
            // The row-tile term goes into the per-lane offset, NOT into soffset: a raw buffer's range check covers inst_offset + voffset
            // only, so with it in soffset a lane whose row lies inside the descriptor's range would fetch rows past N (the
            // last workgroup when N is not a multiple of 16 F).  Unsigned: OOB + the term stays past every range and below 2^32.
            for (int f = 0; f < F; ++f) {
                const int ft = trow(f) * wpitch[G];
                dma16(rw[G], (int)((unsigned)wva[G] + (unsigned)ft), s * WC * 16, base + f * RG::WP * 1024);
                if constexpr (!W4) dma16(rw[G], (int)((unsigned)wvb[G] + (unsigned)ft), s * WC * 16, base + f * RG::WP * 1024 + 1024);
            }
        }
        if constexpr (!(MM_STREAM_DBG & 1) && !QUANT) {
#pragma unroll
            for (int t = 0; t < T16; ++t) {      // (as above: token rows past M -- M = 17 .. 31 with two tiles -- must fail the range check)
                const int tt = 16 * t * xpitch[G];
                dma16(rx[G], (int)((unsigned)xva[G] + (unsigned)tt), s * XC * 16, base + RG::W_BYTES + t * 2048);
                if constexpr (G > 0) {       // (fp4: 16 rows x 4 chunks are one piece)
                    if (!half_tile) dma16(rx[G], (int)((unsigned)xvb[G] + (unsigned)tt), s * XC * 16, base + RG::W_BYTES + t * 2048 + 1024);
                }
            }
        }
    };
    auto frag = [&](const uint8_t *tile, auto EL_) {
        constexpr int EL = decltype(EL_)::value;
        if constexpr (EL == 0) {
            return *reinterpret_cast<const v4i *>(tile + rd4);
        } else if constexpr (EL == 1) {
            const v2i p0 = *reinterpret_cast<const v2i *>(tile + rd6), p1 = *reinterpret_cast<const v2i *>(tile + rd6 + 8),
                      p2 = *reinterpret_cast<const v2i *>(tile + rd6 + 16);
            return v6i{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]};
        } else {
            const v4i p = *reinterpret_cast<const v4i *>(tile + rd8a), p2 = *reinterpret_cast<const v4i *>(tile + rd8b);
            return v8i{p[0], p[1], p[2], p[3], p2[0], p2[1], p2[2], p2[3]};
        }
    };
    // QUANT: this lane's row of the workgroup's own quantized activations (rows past M: row 0 again -- their outputs are never stored)
    dq::LdsMap L = {};
    const uint8_t *qx[3] = {nullptr, nullptr, nullptr}, *qs[3] = {nullptr, nullptr, nullptr};
    auto consume_g = [&](const Slot &q, int d, auto G_, int s, [[maybe_unused]] int i) {
        constexpr int G = decltype(G_)::value, GW = W4 ? 0 : G;
        const uint8_t *base = ringp + d * RG::SLOT;
        // scales: the dword of (row, row group) from the lane that loaded it, shifted to this lane's K block
        int sx[T16], sw[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if constexpr (SIMG) sw[f] = (int)simg[i * (64 * F) + 64 * f];      // (dword 16 F i + 16 f + li of the image, byte h)
            else if constexpr (QUANT || scale_images_for(T16, W4)) sw[f] = 0;   // (MM_STREAM_DBG & 2)
            else sw[f] = __builtin_amdgcn_ds_bpermute(sfw_src[f], sfw_hi[f] ? q.sw[1] : q.sw[0]) >> sh;
        }
        typename Frag<G>::type xv[T16];
        typename Frag<GW>::type wv[F];
        if constexpr (QUANT) {
            sx[0] = (int)(*reinterpret_cast<const uint32_t *>(qs[G] + 4 * s) >> sh);
            const uint8_t *r = qx[G] + s * (chunks_of(G) * 16);          // (qx: row base + this lane's K block: 16 h, 24 h, 16 h)
            if constexpr (G == 0) {
                xv[0] = *reinterpret_cast<const v4i *>(r);
            } else if constexpr (G == 1) {
                const v2i p0 = *reinterpret_cast<const v2i *>(r), p1 = *reinterpret_cast<const v2i *>(r + 8), p2 = *reinterpret_cast<const v2i *>(r + 16);
                xv[0] = v6i{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]};
            } else {
                const v4i p = *reinterpret_cast<const v4i *>(r), p2 = *reinterpret_cast<const v4i *>(r + 64);
                xv[0] = v8i{p[0], p[1], p[2], p[3], p2[0], p2[1], p2[2], p2[3]};
            }
        } else {
#pragma unroll
            for (int t = 0; t < T16; ++t) {
                if constexpr (SIMG) sx[t] = (int)simgx[i * (64 * T16) + 64 * t];
                else if constexpr (scale_images_for(T16, W4)) sx[t] = 0;        // (MM_STREAM_DBG & 2)
                else sx[t] = __builtin_amdgcn_ds_bpermute(sfx_src[t], t >= 2 ? q.sx[1] : q.sx[0]) >> sh;
            }
#pragma unroll
            for (int t = 0; t < T16; ++t) xv[t] = frag(base + RG::W_BYTES + t * 2048, std::integral_constant<int, G>{});
        }
#pragma unroll
        for (int f = 0; f < F; ++f) wv[f] = frag(base + f * RG::WP * 1024, std::integral_constant<int, GW>{});
        static_for<ACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value, f = i / T16, t = i % T16;
            if constexpr (!(MM_STREAM_DBG & 4)) mfma16<G, GW, G * ACC + i, 12 * ACC>(xv[t], wv[f], sx[t], sw[f]);
            else { MM_DEVICE_ONLY(asm volatile("" ::"v"(xv[t]), "v"(wv[f]), "v"(sx[t]), "v"(sw[f]));) }
        });
    };
    // slab j of the launch (wave-uniform): segment g, slab j - (first slab of g)
    auto issue = [&](Slot &q, int d, int j) {
        if (j < c1) issue_g(q, d, std::integral_constant<int, 0>{}, j);
        else if (j < c2) issue_g(q, d, std::integral_constant<int, 1>{}, j - c1);
        else issue_g(q, d, std::integral_constant<int, 2>{}, j - c2);
    };
    auto consume = [&](const Slot &q, int d, int j, int i) {      // i = (j - wave) / NW
        if (j < c1) consume_g(q, d, std::integral_constant<int, 0>{}, j, i);
        else if (j < c2) consume_g(q, d, std::integral_constant<int, 1>{}, j - c1, i);
        else consume_g(q, d, std::integral_constant<int, 2>{}, j - c2, i);
    };

    // `shift` phantom steps in front make the step count a multiple of D
    const int rounds = (cnt + D - 1) / D, shift = rounds * D - cnt;
    auto slab_of = [&](int step) { const int i = step - shift; return wave + NW * (i > 0 ? i : 0); };
    Slot q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d].sw = q[d].sx = v2i{0, 0};
    auto prime = [&]() {      // returns the number of vector-memory instructions it issued (the early quantization phases count them)
        if (cnt > 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) issue(q[d], d, slab_of(d));
        }
        return cnt > 0 ? D * RG::LOADS : 0;
    };
    auto scales = [&]() { if (cnt > 0) request_scales(); };
    if constexpr (QUANT) {
        // (the activation mode has no norm: its 32 + 32 registers stay out of the norm kernels' allocation)
        // (the staged phases and the activation mode -- whose own loads are many and whose arithmetic is long -- take the image in front:
        // down_proj at M = 1 9.2 against 9.3 us, M = 4 13.7 against 14.8)
        if (!qi.early || qi.mode == 1) scales();
        if (!RMS && qi.mode == 1 && qi.early) L = dq::activate_rows_early<NT>(qi, smem_all, prime, [] {});
        else if (!RMS && qi.mode == 1) L = dq::activate_rows_to_lds<NT>(qi, smem_all, [&]() { prime(); });
        // (the ring's first slabs stay in front of the wait: from the landed hook as well, fused gate + up at M = 1 11.9 -> 12.4 us)
        else if (qi.early) L = dq::quantize_rows_early<NT, RMS, EARLY_NPASS<NW>>(qi, smem_all, prime, scales);
        // (whatever fits one pass of lane pairs went the early way: the staged path runs one lane per group -- except in the norm's
        // eight-wave 32-feature kernel, whose register count decides between one and two workgroups per CU)
        else L = dq::quantize_rows_to_lds<NT, RMS, (RMS && NW == 8 && F == 2) ? 2 : 1>(qi, smem_all, [&]() { prime(); });
        const int rr = li < a.M ? li : 0;
        qx[0] = L.opN + rr * L.pN + 16 * h;
        qx[1] = L.opS + rr * L.pS + 24 * h;
        qx[2] = L.opO + rr * L.pO + 16 * h;
        qs[0] = L.scales + rr * L.Gt;
        qs[1] = qs[0] + L.gN;
        qs[2] = qs[1] + L.gS;
    } else {
        scales();      // (older than the ring's first slabs: the first step's counted wait covers the images)
        prime();
    }
    // the accumulators are cleared HERE, behind the quantization phase: nothing asm-owned is live while the compiler allocates that
    // phase's registers (with the norm inside it once parked values in a[2:3]; the build guard caught it)
#if !MM_STREAM_ACCZERO_EARLY
    static_for<12 * ACC>([&](auto r_) { acc_zero<decltype(r_)::value, 12 * ACC>(); });
#endif
    if (cnt > 0) {
        MM_STAMP(1);
        // (QUANT: the scale image is younger than the ring's first slabs, the counted waits below would not cover it; all of it was
        // requested a quantization phase ago)
        if constexpr (QUANT) { MM_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");) }
        for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int step = r * D + d;
                wait_slot<(D - 1) * RG::LOADS>(q[d]);
                if (step >= shift) consume(q[d], d, slab_of(step), step - shift);
                MM_DEVICE_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");)   // the slot's fragments are in registers before it is refilled
                issue(q[d], d, slab_of(step + D));
            }
        }
        auto last = [&](auto d_) {      // the last round: nothing is requested any more, the younger slabs are the later slots only
            constexpr int d = decltype(d_)::value;
            const int step = (rounds - 1) * D + d;
            wait_slot<(D - 1 - d) * RG::LOADS>(q[d]);
            if (step >= shift) consume(q[d], d, slab_of(step), step - shift);
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (last(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, D>{});
    }

    MM_STAMP(2);
    MM_DEVICE_ONLY(asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");)    // the last MFMA's result may be read (18 wait states)
    if constexpr (MM_STREAM_DBG & 8) {
        float t = 0.0f;
        static_for<12 * ACC>([&](auto r_) { t += acc_read<decltype(r_)::value>(); });
        if (t == 12345.678f) a.D[0] = 1;
        return;
    }
    // ---- cross-wave reduction: the partial sums of the present segments side by side and one barrier; with eight 16 x 16 tiles per
    //      wave (M > 32 on 32 features) one segment at a time through one image (the side-by-side image would not leave room for a
    //      second workgroup on the CU) ----
    const int p0 = ns[0] ? 1 : 0, p1 = ns[1] ? 1 : 0, p2 = ns[2] ? 1 : 0, P = p0 + p1 + p2;
    constexpr int IMG = ACC * 4 * 64;                   // floats per wave and segment
    constexpr bool SEQ = ACC >= 8;
    constexpr int OUTS = ACC * 256, PER = (OUTS + NT - 1) / NT;
    float *const red = reinterpret_cast<float *>(smem);
    float run[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) run[k] = 0.0f;
    auto chain = [&](int k, float s) {      // D = bf16(segment sum + D): the reference's rounding between its chained GEMMs
        s += run[k];
        run[k] = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
    };
    if constexpr (SEQ) {
        static_for<3>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            const int present = g == 0 ? p0 : (g == 1 ? p1 : p2);
            if (present) {
                __syncthreads();       // every wave is done with its ring / the previous segment's image has been read
                static_for<4 * ACC>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;      // register r & 3 of tile r >> 2
                    red[(size_t)wave * IMG + r * 64 + lane] = acc_read<4 * g * ACC + r>();
                });
                __syncthreads();
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int o = threadIdx.x + k * NT;
                    float s = 0.0f;
                    if (o < OUTS) {
#pragma unroll
                        for (int w = 0; w < NW; ++w) s += red[(size_t)w * IMG + o];
                    }
                    chain(k, s);
                }
            }
        });
    } else {
        __syncthreads();       // every wave is done with its ring: the reduction image reuses it
        float *mine = red + (size_t)wave * P * IMG;
        int slot = 0;
        static_for<3>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            const int present = g == 0 ? p0 : (g == 1 ? p1 : p2);
            if (present) {
                // register q of a tile holds token rows 4 (l >> 4) + q: with M <= 3 tokens the registers q >= M are rows past M, whose
                // outputs are neither summed nor stored -- M = 1 leaves a quarter of the image's writes (the reduction was 1.6 us of the
                // fused gate + up launch's 13 at M = 1)
                static_for<4>([&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    if (q < a.M) {
                        static_for<ACC>([&](auto i_) {
                            constexpr int r = 4 * decltype(i_)::value + q;
                            mine[slot * IMG + r * 64 + lane] = acc_read<4 * g * ACC + r>();
                        });
                    }
                });
                ++slot;
            }
        });
        __syncthreads();
        MM_STAMP(3);
        if constexpr (T16 == 1) {
            // One token tile: only M of a tile's 16 token rows exist.  The live outputs are numbered through -- (tile f, token m, feature
            // column) = 16 M per tile -- and spread over the workgroup's threads: at M = 1 on 64 features that is one output for each lane of
            // wave 0 instead of four passes over a 1024-entry image of which 64 are live (`reduce+store` 1.7 us of the fused gate + up launch,
            // tools/stream_clock.py).  The same sums in the same order: per segment the waves' partial sums in wave order, then the chain.
            const int rows = a.M < 16 ? a.M : 16, nlive = ACC * 16 * rows;
            if constexpr (ACT) {
                // thread (token m, column c of the workgroup's 32-column group): gate = tile c >> 4, up = tile 2 + (c >> 4), same lane and
                // register; h = silu(bf16(gate)) * bf16(up) in fp32 (silu_mul, mx_direct_convert.h), quantized as direct_quantize.hip does
                const int gi = blockIdx.x;          // the workgroup's group of the intermediate activation (wave-uniform segment)
                const int gN = a.act_K[0] >> 5, gS = a.act_K[1] >> 5;
                const int seg = gi < gN ? 0 : (gi < gN + gS ? 1 : 2);
                // (QUANT: the group buffer in the staged rows' range, dead since the quantization phase -- 2 KB behind the reduction image
                // would cost M = 2 its second workgroup per CU)
                float *hbuf = QUANT ? reinterpret_cast<float *>(smem_all) : red + (size_t)NW * P * IMG;
                for (int t = threadIdx.x; t < 32 * rows; t += NT) {
                    const int m = t >> 5, c = t & 31, col = c & 15;
                    const int og = (c >> 4) * 256 + (m & 3) * 64 + 16 * (m >> 2) + col, ou = og + 512;
                    float rg = 0.0f, ru = 0.0f;
                    for (int sl = 0; sl < P; ++sl) {
                        float sg = 0.0f, su = 0.0f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            sg += red[((size_t)w * P + sl) * IMG + og];
                            su += red[((size_t)w * P + sl) * IMG + ou];
                        }
                        sg += rg;
                        su += ru;
                        rg = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(sg)) : sg;
                        ru = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(su)) : su;
                    }
                    const float hv = silu_mul(bf16_bits_to_f32(f32_to_bf16_bits(rg)), bf16_bits_to_f32(f32_to_bf16_bits(ru)));
                    if (seg == 1) {
                        hbuf[t] = hv;      // fp6: the converter wants the 32 values in one lane (below)
                    } else {
                        // fp4 / fp8: the 32 lanes of the token quantize their group together -- quantize32's arithmetic (mx_direct_convert.h):
                        // the absmax through four DPP steps inside the rows of 16 lanes and one swap across them, the scale in every lane,
                        // the even lane of a pair converts both values and stores their byte (fp4) / two bytes (fp8)
                        float am = fabsf(hv);
                        auto dpp_max = [&](auto ctrl_) {
                            constexpr int ctrl = decltype(ctrl_)::value;
                            am = fmaxf(am, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(am), ctrl, 0xF, 0xF, false)));
                        };
                        dpp_max(std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
                        dpp_max(std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
                        dpp_max(std::integral_constant<int, 0x141>{});      // row_half_mirror
                        dpp_max(std::integral_constant<int, 0x140>{});      // row_mirror
                        am = fmaxf(am, __uint_as_float((uint32_t)__builtin_amdgcn_ds_swizzle((int)__float_as_uint(am), 0x401F)));      // lane ^ 16
                        const float other = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(hv), 0xB1, 0xF, 0xF, false));
                        int e = 0;
                        if (seg == 0) {
                            if (am > 1e-6f) e = scale_exponent_f32<EL_FP4>(am);
                            const float scale = __uint_as_float((uint32_t)(127 + (e < -126 ? -126 : e)) << 23);
                            if ((c & 1) == 0) {
                                const uint32_t r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, hv, other, scale, 0);
                                a.act_o[0][(size_t)m * (a.act_K[0] >> 1) + gi * 16 + (c >> 1)] = (uint8_t)r;
                            }
                            if (c == 0) a.act_sf[0][sf_offset(m, gi, a.act_K[0])] = (uint8_t)(e + 127);
                        } else {
                            if (am > 1e-6f) e = scale_exponent_f32<EL_FP8>(am);
                            const float scale = __uint_as_float((uint32_t)(127 + (e < -126 ? -126 : e)) << 23);
                            if ((c & 1) == 0) {
                                ds2 r = {0, 0};
                                r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, hv, other, scale, false);
                                *reinterpret_cast<uint16_t *>(a.act_o[2] + (size_t)m * a.act_K[2] + (gi - gN - gS) * 32 + c) = (uint16_t)r[0];
                            }
                            if (c == 0) a.act_sf[2][sf_offset(m, gi - gN - gS, a.act_K[2])] = (uint8_t)(e + 127);
                        }
                    }
                }
                if (seg == 1) {      // (wave-uniform: every thread takes the barrier or none)
                    __syncthreads();
                    if ((int)threadIdx.x < rows) {
                        const int m = threadIdx.x;
                        float v[32];
#pragma unroll
                        for (int i = 0; i < 32; ++i) v[i] = hbuf[m * 32 + i];
                        const uint32_t byte = quantize32<EL_FP6, true>(v, a.act_o[1] + (size_t)m * ((a.act_K[1] >> 2) * 3) + (gi - gN) * 24);
                        a.act_sf[1][sf_offset(m, gi - gN, a.act_K[1])] = (uint8_t)byte;
                    }
                }
                MM_STAMP(4);
                return;
            }
            for (int t = threadIdx.x; t < nlive; t += NT) {
                const int f = t / (16 * rows), rem = t - f * 16 * rows, m = rem >> 4, col = rem & 15;
                const int o = f * 256 + (m & 3) * 64 + 16 * (m >> 2) + col;      // register m & 3 of lane 16 (m >> 2) + col of tile f
                float run1 = 0.0f;
                for (int sl = 0; sl < P; ++sl) {
                    float s = 0.0f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += red[((size_t)w * P + sl) * IMG + o];
                    s += run1;
                    run1 = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(s)) : s;
                }
                const int n = n0 + 16 * f + col;
                if (n < a.N) {
                    if (a.out_f32) {
                        reinterpret_cast<float *>(a.D)[(size_t)m * a.N + n] = run1;
                    } else {
                        uint32_t b = f32_to_bf16_bits(run1);
                        if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
                        a.D[(size_t)m * a.N + n] = (uint16_t)b;
                    }
                }
            }
            MM_STAMP(4);
            return;
        }
        // Segment by segment, the thread's PER outputs side by side: the PER * NW reads of a segment are independent and travel together,
        // then the chain.  (Round 6: with the outputs outside and the segments inside, every segment of every output was its own LDS
        // round trip -- 12 in a row for 64 features x 3 segments: `reduce+store` 1.9 us of a 14 us launch, tools/stream_clock.py.)
        // Outputs of token rows past M are neither summed nor stored.
        bool live[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int o = threadIdx.x + k * NT;
            const int l = o & 63, r = (o >> 6) & 3, t = (o >> 8) % T16;
            live[k] = o < OUTS && 16 * t + 4 * (l >> 4) + r < a.M;
        }
        for (int sl = 0; sl < P; ++sl) {
            float s[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int o = threadIdx.x + k * NT;
                s[k] = 0.0f;
                if (live[k]) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) s[k] += red[((size_t)w * P + sl) * IMG + o];
                }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) chain(k, s[k]);
        }
    }
    if constexpr (ACT && T16 > 1) {
        // ACT with two token tiles (17 .. 32 tokens; the single-tile launches left through the live-output reduction above).  256 threads,
        // 8 outputs each: o = thread + 256 k is (tile k = f T16 + t, register r = wave, lane l) -- so run[f T16 + t] (f = 0, 1: gate) and
        // run[(f + 2) T16 + t] (up) of one thread belong to the same token 16 t + 4 (l >> 4) + r and to columns (l & 15) + 16 f of the
        // workgroup's group, whose other columns sit in the other lanes of the DPP row: the group's absmax is four DPP steps, no LDS.
        static_assert(NT == 256 && PER == 4 * T16, "tile = k");
        const int l = threadIdx.x & 63, r = threadIdx.x >> 6, col = l & 15;
        const int gi = blockIdx.x, gN = a.act_K[0] >> 5, gS = a.act_K[1] >> 5;
        const int seg = gi < gN ? 0 : (gi < gN + gS ? 1 : 2);
        float *hbuf = red;                      // (fp6 groups only, behind a barrier: the reduction image is done with)
        if (seg == 1) __syncthreads();
        static_for<T16>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            const int m = 16 * t + 4 * (l >> 4) + r;
            const bool live = m < a.M;
            auto bf = [](float x) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); };
            const float h0 = silu_mul(bf(run[t]), bf(run[2 * T16 + t])), h1 = silu_mul(bf(run[T16 + t]), bf(run[3 * T16 + t]));
            if (seg == 1) {
                if (live) { hbuf[m * 32 + col] = h0; hbuf[m * 32 + 16 + col] = h1; }
            } else {
                float am = fmaxf(fabsf(h0), fabsf(h1));
                auto dpp_max = [&](auto ctrl_) {
                    constexpr int ctrl = decltype(ctrl_)::value;
                    am = fmaxf(am, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(am), ctrl, 0xF, 0xF, false)));
                };
                dpp_max(std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
                dpp_max(std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
                dpp_max(std::integral_constant<int, 0x141>{});      // row_half_mirror
                dpp_max(std::integral_constant<int, 0x140>{});      // row_mirror
                const float o0 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(h0), 0xB1, 0xF, 0xF, false));
                const float o1 = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(h1), 0xB1, 0xF, 0xF, false));
                int e = 0;
                if (seg == 0) {
                    if (am > 1e-6f) e = scale_exponent_f32<EL_FP4>(am);
                    const float scale = __uint_as_float((uint32_t)(127 + (e < -126 ? -126 : e)) << 23);
                    if (live && (col & 1) == 0) {
                        uint8_t *out = a.act_o[0] + (size_t)m * (a.act_K[0] >> 1) + gi * 16 + (col >> 1);
                        out[0] = (uint8_t)__builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, h0, o0, scale, 0);
                        out[8] = (uint8_t)__builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, h1, o1, scale, 0);
                    }
                    if (live && col == 0) a.act_sf[0][sf_offset(m, gi, a.act_K[0])] = (uint8_t)(e + 127);
                } else {
                    if (am > 1e-6f) e = scale_exponent_f32<EL_FP8>(am);
                    const float scale = __uint_as_float((uint32_t)(127 + (e < -126 ? -126 : e)) << 23);
                    if (live && (col & 1) == 0) {
                        uint8_t *out = a.act_o[2] + (size_t)m * a.act_K[2] + (gi - gN - gS) * 32 + col;
                        ds2 q0 = {0, 0}, q1 = {0, 0};
                        q0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q0, h0, o0, scale, false);
                        q1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q1, h1, o1, scale, false);
                        *reinterpret_cast<uint16_t *>(out) = (uint16_t)q0[0];
                        *reinterpret_cast<uint16_t *>(out + 16) = (uint16_t)q1[0];
                    }
                    if (live && col == 0) a.act_sf[2][sf_offset(m, gi - gN - gS, a.act_K[2])] = (uint8_t)(e + 127);
                }
            }
        });
        if (seg == 1) {
            __syncthreads();
            if ((int)threadIdx.x < a.M) {
                const int m = threadIdx.x;
                float v[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = hbuf[m * 32 + i];
                const uint32_t byte = quantize32<EL_FP6, true>(v, a.act_o[1] + (size_t)m * ((a.act_K[1] >> 2) * 3) + (gi - gN) * 24);
                a.act_sf[1][sf_offset(m, gi - gN, a.act_K[1])] = (uint8_t)byte;
            }
        }
        MM_STAMP(4);
        return;
    }
    // output o = (i = f * T16 + t, r, l): token 16 t + 4 (l >> 4) + r, feature n0 + 16 f + (l & 15)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int o = threadIdx.x + k * NT;
        const int l = o & 63, r = (o >> 6) & 3, i = o >> 8;
        const int f = i / T16, t = i % T16;
        const int m = 16 * t + 4 * (l >> 4) + r, n = n0 + 16 * f + (l & 15);
        if (o < OUTS && m < a.M && n < a.N) {
            if (a.out_f32) {
                reinterpret_cast<float *>(a.D)[(size_t)m * a.N + n] = run[k];
            } else {
                uint32_t b = f32_to_bf16_bits(run[k]);
                if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bf16_bits_to_f32(a.bias[n]));
                a.D[(size_t)m * a.N + n] = (uint16_t)b;
            }
        }
    }
    MM_STAMP(4);
}

template <int F, int T16, int D, int NW, bool W4>
__global__ void __launch_bounds__(64 * NW) mx_gemm_stream_kernel(GemmArgs a) { stream_body<F, T16, D, NW, W4>(a); }

template <int F, int T16, int D, int NW, bool W4>
static hipError_t launch_one(const GemmArgs &a, hipStream_t stream) {
    const int present = (a.K[0] ? 1 : 0) + (a.K[1] ? 1 : 0) + (a.K[2] ? 1 : 0);
    // [rings | scale images] while the slabs stream, then the reduction image over both
    const int red_bytes = NW * (F * T16 >= 8 ? 1 : present) * F * T16 * 4 * 64 * (int)sizeof(float);
    const int stage_bytes = NW * D * Ring<F, T16, W4>::SLOT + scale_images_bytes(F, T16, NW, (a.K[0] + a.K[1] + a.K[2]) >> 7, W4);
    const int lds = red_bytes > stage_bytes ? red_bytes : stage_bytes;
    if (lds > STREAM_LDS_MAX) return hipErrorInvalidValue;      // (mx_gemm_stream_supported keeps callers away from this)
    static DynamicLdsOnce once;
    if (lds > 65536) {
        hipError_t e = once.ensure(reinterpret_cast<const void *>(mx_gemm_stream_kernel<F, T16, D, NW, W4>), STREAM_LDS_MAX);
        if (e != hipSuccess) return e;
    }
    const int blocks = (a.N + 16 * F - 1) / (16 * F);
    MM_LAUNCH((mx_gemm_stream_kernel<F, T16, D, NW, W4>), dim3(blocks), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}

// Grouped launch (MoE experts): blockIdx.y picks one of up to MM_MAX_GROUPS problems that share N, the K split and the weight mode
// (as mx_gemm_skinny_grouped_kernel); T16 covers the largest group, a group with M = 0 returns at once.
template <int F, int T16, int D, int NW, bool W4>
__global__ void __launch_bounds__(64 * NW) mx_gemm_stream_grouped_kernel(GroupedGemmArgs ga) {
    const GemmArgs &a = ga.g[blockIdx.y];
    if (a.M > 0) stream_body<F, T16, D, NW, W4>(a);
}

template <int F, int T16, int D, int NW, bool W4>
static hipError_t launch_grouped_one(const GroupedGemmArgs &ga, hipStream_t stream) {
    const GemmArgs &a = ga.g[0];
    const int present = (a.K[0] ? 1 : 0) + (a.K[1] ? 1 : 0) + (a.K[2] ? 1 : 0);
    const int red_bytes = NW * (F * T16 >= 8 ? 1 : present) * F * T16 * 4 * 64 * (int)sizeof(float);
    const int stage_bytes = NW * D * Ring<F, T16, W4>::SLOT + scale_images_bytes(F, T16, NW, (a.K[0] + a.K[1] + a.K[2]) >> 7, W4);
    const int lds = red_bytes > stage_bytes ? red_bytes : stage_bytes;
    if (lds > STREAM_LDS_MAX) return hipErrorInvalidValue;      // (mx_gemm_stream_grouped_supported keeps callers away from this)
    static DynamicLdsOnce once;
    if (lds > 65536) {
        hipError_t e = once.ensure(reinterpret_cast<const void *>(mx_gemm_stream_grouped_kernel<F, T16, D, NW, W4>), STREAM_LDS_MAX);
        if (e != hipSuccess) return e;
    }
    const int blocks = (a.N + 16 * F - 1) / (16 * F);
    hipLaunchKernelGGL((mx_gemm_stream_grouped_kernel<F, T16, D, NW, W4>), dim3(blocks, ga.ngroups), dim3(64 * NW), lds, stream, ga);
    return hipGetLastError();
}

// mm_qlinear_decode on the streaming kernel: the workgroup quantizes the M <= 8 rows itself (QUANT)
template <int F, int D, int NW, bool W4>
__global__ void __launch_bounds__(64 * NW) mx_qlinear_stream_kernel(GemmArgs a, dq::QuantIn qi, int qbytes) {
    stream_body<F, 1, D, NW, W4, true>(a, qi, qbytes);
}
// ... with the RMSNorm in front of the quantization (mm_rmsnorm_qlinear_decode): kernels of their own, see dq::quantize_rows_to_lds
// (105 VGPRs + 24 accumulators at F = 2: ONE 8-wave workgroup per CU where the kernel without the norm holds two.  Bounding it with
// __launch_bounds__(512, 4) made hipcc split the file 64 + 64, park values in a[0:23] and copy pending load destinations before their
// wait -- the build guard refused all of it.  The Llama layer's wide launch, gate | up at N = 28672, runs F = 4 on 4 waves: three
// workgroups per CU with and without the norm.)
template <int F, int D, int NW, bool W4>
__global__ void __launch_bounds__(64 * NW) mx_qlinear_stream_rms_kernel(GemmArgs a, dq::QuantIn qi, int qbytes) {
    stream_body<F, 1, D, NW, W4, true, true>(a, qi, qbytes);
}

// ... and with the activation inside (ACT, see stream_body): the fused gate | up weight, 64 rows (32 gate + 32 up) x 4 waves
template <bool RMS>
__global__ void __launch_bounds__(256) mx_qlinear_stream_act_kernel(GemmArgs a, dq::QuantIn qi, int qbytes) {
    stream_body<4, 1, 2, 4, true, true, RMS, true>(a, qi, qbytes);
}
template <int T16>
__global__ void __launch_bounds__(256) mx_gemm_stream_act_kernel(GemmArgs a) { stream_body<4, T16, 2, 4, true, false, false, true>(a); }
constexpr int ACT_GROUP_BYTES = 16 * 32 * 4;       // the 32 values of a group for up to 16 tokens, behind the reduction image

template <int F, int D, int NW, bool W4, bool RMS = false, bool ACT = false>
static hipError_t launch_quant(const GemmArgs &a, dq::QuantIn qi, hipStream_t stream) {
    static_assert(!ACT || (F == 4 && D == 2 && NW == 4 && W4), "mx_qlinear_stream_act_kernel");
    if constexpr (ACT && !RMS) {
        if (qi.norm_w != nullptr) return launch_quant<F, D, NW, W4, true, true>(a, qi, stream);
    }
    if constexpr (!ACT && !RMS && ((F == 4 && D == 2 && NW == 4) || (F == 2 && D == 2 && NW == 8) || (F == 1 && NW == 8 && (D == 3 || D == 4)))) {
        if (qi.norm_w != nullptr) return launch_quant<F, D, NW, W4, true>(a, qi, stream);      // (the configurations the default dispatch uses)
    }
    if (!RMS && qi.norm_w != nullptr) return hipErrorInvalidValue;       // (a kernel-developer override picked a configuration without a norm variant)
    const int present = (a.K[0] ? 1 : 0) + (a.K[1] ? 1 : 0) + (a.K[2] ? 1 : 0);
    const size_t Kt = (size_t)a.K[0] + a.K[1] + a.K[2];
    // [rings | scale image] while the slabs stream, then the reduction image over both
    const size_t red_bytes = (size_t)NW * present * F * 4 * 64 * sizeof(float);      // (ACT: the group buffer lies in the staged rows' range)
    const size_t ring_bytes = (size_t)NW * D * Ring<F, 1, W4, true>::SLOT + scale_image_bytes(F, NW, (int)(Kt >> 7));
    const size_t tail = red_bytes > ring_bytes ? red_bytes : ring_bytes;
    const size_t ops = ((dq::operand_bytes(a.M, a.K) + 15) & ~(size_t)15) + (qi.norm_w != nullptr ? ((dq::rms_bytes(a.M, a.K) + 15) & ~(size_t)15) : 0);
    // staged bf16 rows: all M if two workgroups still fit a CU's 160 KB, else as many as one workgroup can hold (at least one)
    constexpr size_t LDS_CU = 160 * 1024, LDS_WG = 156 * 1024;
    size_t rows = qi.mode == 1 ? 0 : a.M;       // (mode 1 quantizes straight from global memory: nothing is staged)
    if (qi.mode != 1 && 2 * (ops + tail + rows * Kt * 2) > LDS_CU) {
        const size_t two = LDS_CU / 2 > ops + tail ? (LDS_CU / 2 - ops - tail) / (Kt * 2) : 0;
        const size_t one = LDS_WG > ops + tail ? (LDS_WG - ops - tail) / (Kt * 2) : 0;
        rows = two >= 1 ? two : one;
        if (rows > (size_t)a.M) rows = a.M;
    }
    if (rows < 1 && qi.mode != 1) return hipErrorInvalidValue;
    qi.stage_rows = (int)rows;
    qi.early = (qi.mode != 1 && rows >= (size_t)a.M && dq::early_fits(a.M, (int)Kt, 64 * NW, EARLY_NPASS<NW>)) ? 1 : 0;
    if (qi.mode == 1) qi.early = (size_t)a.M * (Kt / 32) <= 64u * NW ? 1 : 0;      // activate_rows_early: one pass of the workgroup's threads
    static const int early_on = getenv("MICROMIX_DECODE_EARLY") ? atoi(getenv("MICROMIX_DECODE_EARLY")) : 1;   // kernel-developer override
    if (!early_on) qi.early = 0;
    const size_t qbytes = rows * Kt * 2 + ops, lds = qbytes + tail;
    if (lds > LDS_WG) return hipErrorInvalidValue;      // (every mode: the supported() predicates keep callers away from this)
    static DynamicLdsOnce once;
    auto kern = [] {
        if constexpr (ACT) return mx_qlinear_stream_act_kernel<RMS>;
        else if constexpr (RMS) return mx_qlinear_stream_rms_kernel<F, D, NW, W4>;
        else return mx_qlinear_stream_kernel<F, D, NW, W4>;
    }();
    if (lds > 65536) {
        hipError_t e = once.ensure(reinterpret_cast<const void *>(kern), (int)LDS_WG);
        if (e != hipSuccess) return e;
    }
    const int blocks = (a.N + 16 * F - 1) / (16 * F);
    MM_LAUNCH(kern, dim3(blocks), dim3(64 * NW), lds, stream, a, qi, (int)qbytes);
    return hipGetLastError();
}

}  // namespace stream

#if MM_STREAM_CLOCK
extern "C" int mm_diag_set_stream_clock(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(stream::g_stream_clock), &buf, sizeof(buf)) == hipSuccess ? 0 : 3;
}
#endif

// the grouped launch on the streaming kernel: every group M <= 64 (one to four token tiles); the workgroups of all groups count towards
// filling the CUs
// one token tile: the rings of the (F, D, NW) the dispatch may pick (at most 8 waves x 4 slots x (1 + 2) KB with fp4 weights, (2 + 2) KB
// without) and both scale images (at most 8 waves, 16 rows per slab: the larger of the two geometries) must fit a workgroup's LDS
// 16 features x two token tiles x 8 waves with fp4 weights: `depth` slots of 1 + 4 KB and both images
static bool two_tile_fits(int depth, int T) {
    return 8 * depth * 5 * 1024 + stream::scale_image_bytes(1, 8, T) + stream::scale_image_bytes(2, 8, T) <= stream::STREAM_LDS_MAX;
}
static bool stream_images_fit(int M, const int K[3], bool w4) {
    if (M > 32 || (M > 16 && !w4)) return true;
    const int T = (K[0] + K[1] + K[2]) >> 7;
    if (M > 16) return two_tile_fits(2, T);      // (the 32-feature configuration: 4 waves x 2 slots x 6 KB + smaller images)
    const int rings = 8 * (T <= 32 ? 4 : 3) * (w4 ? 3 : 4) * 1024;
    return rings + 2 * stream::scale_image_bytes(1, 8, T) <= stream::STREAM_LDS_MAX;
}
bool mx_gemm_stream_grouped_supported(int max_m, int ngroups, int N, const int K[3]) {
    static const int on = getenv("MICROMIX_STREAM_GROUPED") ? atoi(getenv("MICROMIX_STREAM_GROUPED")) : 1;   // kernel-developer override
    (void)N;
    static const int max_tokens = getenv("MICROMIX_STREAM_GROUPED_MAX_M") ? atoi(getenv("MICROMIX_STREAM_GROUPED_MAX_M")) : 64;
    return on && max_m >= 1 && max_m <= max_tokens && max_m <= 64 && ngroups >= 1 && ngroups <= MM_MAX_GROUPS && stream_images_fit(max_m, K, false) && stream_images_fit(max_m, K, true);      // (either weight mode)
}
hipError_t launch_mx_gemm_stream_grouped(const GroupedGemmArgs &ga, int max_m, bool w4, hipStream_t stream) {
    using namespace stream;
    const bool wide = (ga.g[0].N + 31) / 32 * ga.ngroups >= device_cus();
#define MM_STREAM_G(F_, T_, D_, NW_)                                                 \
    (w4 ? launch_grouped_one<F_, T_, D_, NW_, true>(ga, stream) : launch_grouped_one<F_, T_, D_, NW_, false>(ga, stream))
    if (max_m <= 16) return wide ? MM_STREAM_G(2, 1, 2, 8) : MM_STREAM_G(1, 1, 3, 8);
    if (max_m <= 32 && w4) {      // (as launch_mx_gemm_stream: scale images with two token tiles)
        const int slabs = (ga.g[0].K[0] + ga.g[0].K[1] + ga.g[0].K[2]) >> 7;
        if (wide) return launch_grouped_one<2, 2, 2, 4, true>(ga, stream);
        return two_tile_fits(3, slabs) ? launch_grouped_one<1, 2, 3, 8, true>(ga, stream) : launch_grouped_one<1, 2, 2, 8, true>(ga, stream);
    }
    if (max_m <= 32) return wide ? MM_STREAM_G(2, 2, 3, 4) : MM_STREAM_G(1, 2, 3, 8);
    if (max_m <= 48) return wide ? MM_STREAM_G(2, 3, 2, 4) : MM_STREAM_G(1, 3, 2, 8);
    return wide ? MM_STREAM_G(2, 4, 2, 4) : MM_STREAM_G(1, 4, 2, 8);
#undef MM_STREAM_G
}

bool mx_gemm_stream_supported(int M, int N, const int K[3], bool w4) {
    static const int on = getenv("MICROMIX_STREAM") ? atoi(getenv("MICROMIX_STREAM")) : 1;   // kernel-developer override
    if (!on || M > 64) return false;
    // few features, short K, a handful of tokens (q/k/v/o at M <= 8): the launch is all start-up, and the first kernel's is shorter
    // (q/o at M = 1: 4.65 against 5.0-5.5 us; from M = 16 on the two meet)
    if (M <= 8 && (N + 15) / 16 <= device_cus() && K[0] + K[1] + K[2] <= 8192) return false;
    return stream_images_fit(M, K, w4);
}

// LDS that the ring / reduction tail of the quantizing launches may need beside the quantization's own range: the largest of the
// instantiations launch_quant is called with -- matching-precision weights, NW * D * SLOT = 8 * 2 * 4096 = 4 * 2 * 8192 = 8 * 4 * 2048
// = 64 KB (fp4 weights: 48 KB).  The supported() predicates budget this worst case so that a shape they accept always launches
// (they budgeted 48 KB whatever the weight mode).
constexpr size_t stream_tail_budget(bool w4) { return (w4 ? 48 : 64) * 1024; }
// ... with the scale image (stream::scale_image_bytes) of the (F, NW) the dispatch below picks for N output features behind the rings
// (NW * D * SLOT <= 32 KB with fp4 weights, 64 KB without): under the reduction image where that is the larger one
static size_t quant_tail_budget(int N, size_t Kt, bool w4) {
    const int cus = device_cus(), T = (int)(Kt >> 7);
    const size_t simg = (N + 31) / 32 > 2 * cus ? stream::scale_image_bytes(4, 4, T)
                                                : ((N + 31) / 32 >= cus ? stream::scale_image_bytes(2, 8, T) : stream::scale_image_bytes(1, 8, T));
    const size_t rings = (w4 ? 32 : 64) * 1024 + simg;
    return rings > stream_tail_budget(w4) ? rings : stream_tail_budget(w4);
}

// 1 if mm_qlinear_decode can run on the streaming kernel (the quantized rows, one staged row and the rings fit a workgroup's LDS)
bool qlinear_stream_supported(int M, int N, const int K[3], bool rms, bool w4) {
    static const int on = getenv("MICROMIX_DECODE_STREAM") ? atoi(getenv("MICROMIX_DECODE_STREAM")) : 1;   // kernel-developer override
    // Measured against the first fused kernel (qlinear_decode.hip; tools/time_decode.py, profiles/r04_stream_ablation.txt section 8):
    // it wins where N / 32 fills the CUs and one pass quantizes the rows (gate/up at M = 1 / 2 / 4: 12.5 / 11.8 / 14.3 -> 10.2 / 9.2 /
    // 12.3 us) and loses on q/k/v/o (few workgroups: all start-up) and at M = 8 (two passes per workgroup: quantize + GEMM wins there).
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    const size_t norm = rms ? dq::rms_bytes(M, K) : 0;
    if (rms && Kt > (size_t)dq::RMS_MAX_K) return false;
    const size_t need = dq::operand_bytes(M, K) + norm + Kt * 2 + quant_tail_budget(N, Kt, w4) + 64;
    if (on == 2 && M >= 1 && M <= 8) return need <= 156 * 1024;     // (A/B runs: every shape that fits)
    if (!on || M < 1 || M > 4 || (N + 31) / 32 < device_cus()) return false;
    return need <= 156 * 1024;
}

hipError_t launch_qlinear_stream(const void *X, const int16_t *idx, const uint8_t *const W[3], const uint8_t *const SFW[3], int M, int N,
                                 const int K[3], bool w4, int round_per_segment, const void *bias, void *D, hipStream_t stream,
                                 const NormArgs &norm) {
    using namespace stream;
    GemmArgs a = {};
    dq::QuantIn qi = {};
    qi.X = (const uint16_t *)X;
    qi.idx = idx;
    qi.M = M;
    qi.norm_w = (const uint16_t *)norm.weight;
    qi.eps = norm.eps;
    qi.int_round = norm.int_round;
    for (int g = 0; g < 3; ++g) {
        a.W[g] = W[g];
        a.SFW[g] = SFW[g];
        a.K[g] = qi.K[g] = K[g];
    }
    a.M = M;
    a.N = N;
    a.sfx_row_tiles = 1;
    a.sfw_row_tiles = (N + 127) / 128;
    a.round_per_segment = round_per_segment;
    a.bias = (const uint16_t *)bias;
    a.D = (uint16_t *)D;
    const bool wide = (N + 31) / 32 >= device_cus();
    // kernel-developer overrides: 64 features x 4 waves with this many slots (-1: never) / ring slots (2, 3, 4).  Ignored with the norm inside
    // the launch: only the default (F, D, NW) configurations have norm variants (an override made a supported shape fail at launch)
    static const int qf4_env = getenv("MICROMIX_DECODE_F4") ? atoi(getenv("MICROMIX_DECODE_F4")) : 0;
    static const int qd_env = getenv("MICROMIX_DECODE_DEPTH") ? atoi(getenv("MICROMIX_DECODE_DEPTH")) : 2;
    const int qf4 = norm.weight != nullptr ? 0 : qf4_env, qd = norm.weight != nullptr ? 2 : qd_env;
    if (wide && qf4 == 2) return w4 ? launch_quant<4, 2, 4, true>(a, qi, stream) : launch_quant<4, 2, 4, false>(a, qi, stream);
    if (wide && qf4 == 3) return w4 ? launch_quant<4, 3, 4, true>(a, qi, stream) : launch_quant<4, 3, 4, false>(a, qi, stream);
    if (wide && qf4 == 4) return w4 ? launch_quant<4, 4, 4, true>(a, qi, stream) : launch_quant<4, 4, 4, false>(a, qi, stream);
    // more than one round of 32-feature workgroups (two per CU): 64 features x 4 waves halve the workgroups that repeat the quantization and
    // run in one round (fused gate + up, N = 28672, M = 1: 17.3 -> 14.2 us, from HBM 18.7 -> 16.1; at N = 14336 it loses 1 us from HBM)
    if (wide && qf4 == 0 && (N + 31) / 32 > 2 * device_cus()) return w4 ? launch_quant<4, 2, 4, true>(a, qi, stream) : launch_quant<4, 2, 4, false>(a, qi, stream);
    if (wide && qd == 3) return w4 ? launch_quant<2, 3, 8, true>(a, qi, stream) : launch_quant<2, 3, 8, false>(a, qi, stream);
    if (wide && qd == 4) return w4 ? launch_quant<2, 4, 8, true>(a, qi, stream) : launch_quant<2, 4, 8, false>(a, qi, stream);
    if (wide) return w4 ? launch_quant<2, 2, 8, true>(a, qi, stream) : launch_quant<2, 2, 8, false>(a, qi, stream);
    if (((K[0] + K[1] + K[2]) >> 7) <= 32) return w4 ? launch_quant<1, 4, 8, true>(a, qi, stream) : launch_quant<1, 4, 8, false>(a, qi, stream);
    return w4 ? launch_quant<1, 3, 8, true>(a, qi, stream) : launch_quant<1, 3, 8, false>(a, qi, stream);
}

// The fused gate | up launch with the activation inside (stream_body ACT): N = 2 I rows of weights, 128 gate | 128 up alternating
bool gate_up_act_stream_supported(int M, int N, const int K[3], bool from_bf16, bool rms) {
    static const int on = getenv("MICROMIX_GATE_UP_ACT_STREAM") ? atoi(getenv("MICROMIX_GATE_UP_ACT_STREAM")) : 1;   // kernel-developer override
    if (!on || M < 1 || N < 256 || (N % 256) != 0) return false;
    if ((N + 63) / 64 < device_cus()) return false;          // 64-row workgroups: at least one per CU, else the two-launch form
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    if (!from_bf16) {      // one or two token tiles: 4 waves x 2 slots x (4 + 2 T16) KB and both scale images
        const int T16 = M <= 16 ? 1 : 2;
        return M <= 32 && 4 * 2 * (4 + 2 * T16) * 1024 + stream::scale_images_bytes(4, T16, 4, (int)(Kt >> 7), true) <= stream::STREAM_LDS_MAX;
    }
    if (M > 4 || (rms && Kt > (size_t)dq::RMS_MAX_K)) return false;
    const size_t norm = rms ? ((dq::rms_bytes(M, K) + 15) & ~(size_t)15) : 0;
    const size_t tail = 48 * 1024;      // (three segments' reduction image; the rings + scale image stay below; the group buffer lies in the staged row's range)
    return Kt * 2 >= (size_t)stream::ACT_GROUP_BYTES && dq::operand_bytes(M, K) + 16 + norm + Kt * 2 + tail + 64 <= 156 * 1024 &&
           stream::scale_image_bytes(4, 4, (int)(Kt >> 7)) <= 16 * 1024;
}
template <int T16>
static hipError_t launch_gate_up_act_tiles(const GemmArgs &a, hipStream_t stream) {
    using namespace stream;
    const int present = (a.K[0] ? 1 : 0) + (a.K[1] ? 1 : 0) + (a.K[2] ? 1 : 0);
    // (two token tiles: eight tiles per wave reduce one segment at a time through one image; the fp6 group buffer lies inside it)
    const int red_bytes = 4 * (4 * T16 >= 8 ? 1 : present) * 4 * T16 * 4 * 64 * (int)sizeof(float) + (T16 == 1 ? ACT_GROUP_BYTES : 0);
    const int stage_bytes = 4 * 2 * Ring<4, T16, true>::SLOT + scale_images_bytes(4, T16, 4, (a.K[0] + a.K[1] + a.K[2]) >> 7, true);
    const int lds = red_bytes > stage_bytes ? red_bytes : stage_bytes;
    if (lds > STREAM_LDS_MAX) return hipErrorInvalidValue;
    static DynamicLdsOnce once;
    if (lds > 65536) {
        hipError_t e = once.ensure(reinterpret_cast<const void *>(mx_gemm_stream_act_kernel<T16>), STREAM_LDS_MAX);
        if (e != hipSuccess) return e;
    }
    MM_LAUNCH(mx_gemm_stream_act_kernel<T16>, dim3(a.N / 64), dim3(256), lds, stream, a);
    return hipGetLastError();
}
hipError_t launch_gate_up_act_stream(const GemmArgs &a, hipStream_t stream) {
    return a.M <= 16 ? launch_gate_up_act_tiles<1>(a, stream) : launch_gate_up_act_tiles<2>(a, stream);
}
hipError_t launch_gate_up_act_stream_decode(const void *X, const int16_t *idx, const GemmArgs &a, hipStream_t stream, const NormArgs &norm) {
    using namespace stream;
    dq::QuantIn qi = {};
    qi.X = (const uint16_t *)X;
    qi.idx = idx;
    qi.M = a.M;
    qi.norm_w = (const uint16_t *)norm.weight;
    qi.eps = norm.eps;
    qi.int_round = norm.int_round;
    for (int g = 0; g < 3; ++g) qi.K[g] = a.K[g];
    return launch_quant<4, 2, 4, true, false, true>(a, qi, stream);
}

// mm_down_activate_decode: down_proj at M <= 4 straight from the bf16 gate | up matrix -- every workgroup computes silu(gate) * up and
// quantizes it for its own use (the bytes of mm_activate_quantize), K = the intermediate size in natural order
bool down_activate_stream_supported(int M, int N, const int K[3], bool w4) {
    const size_t Kt = (size_t)K[0] + K[1] + K[2];
    return M >= 1 && M <= 4 && dq::operand_bytes(M, K) + quant_tail_budget(N, Kt, w4) + 64 <= 156 * 1024;
}
hipError_t launch_down_activate_stream(const void *GU, const uint8_t *const W[3], const uint8_t *const SFW[3], int M, int N, const int K[3],
                                       bool w4, int round_per_segment, const void *bias, void *D, hipStream_t stream) {
    using namespace stream;
    GemmArgs a = {};
    dq::QuantIn qi = {};
    qi.X = (const uint16_t *)GU;
    qi.M = M;
    qi.mode = 1;
    for (int g = 0; g < 3; ++g) {
        a.W[g] = W[g];
        a.SFW[g] = SFW[g];
        a.K[g] = qi.K[g] = K[g];
    }
    a.M = M;
    a.N = N;
    a.sfx_row_tiles = 1;
    a.sfw_row_tiles = (N + 127) / 128;
    a.round_per_segment = round_per_segment;
    a.bias = (const uint16_t *)bias;
    a.D = (uint16_t *)D;
    const bool wide = (N + 31) / 32 >= device_cus();
    if (wide) return w4 ? launch_quant<2, 2, 8, true>(a, qi, stream) : launch_quant<2, 2, 8, false>(a, qi, stream);
    // (A deeper ring does nothing here -- seven or fourteen slots, a wave's whole share of K = 14336 in flight at once, measured the same
    // 10.6 us as three, round 6, and with the weights coming from HBM 11.6 against 11.5: the launch is the sum of its vector-memory
    // instructions, not a chain of round trips.  The scale image is what took 1.6 us off it.)
    return w4 ? launch_quant<1, 3, 8, true>(a, qi, stream) : launch_quant<1, 3, 8, false>(a, qi, stream);
}

#ifndef MM_STREAM_SWEEP      // kernel-developer build: every (F, D, NW) combination, picked by MICROMIX_STREAM_CFG="F,D,NW"
#define MM_STREAM_SWEEP 0
#endif

hipError_t launch_mx_gemm_stream(const GemmArgs &a, bool w4, hipStream_t stream) {
    using namespace stream;
    const int cus = device_cus();
    // 32 features per workgroup once those fill the CUs, 16 below (q/o at N = 4096: 256 workgroups instead of 128)
    const bool wide = (a.N + 31) / 32 >= cus;
#define MM_STREAM(F_, T_, D_, NW_)                                                   \
    (w4 ? launch_one<F_, T_, D_, NW_, true>(a, stream) : launch_one<F_, T_, D_, NW_, false>(a, stream))
#if MM_STREAM_SWEEP
    static int cf = 0, cd = 0, cn = 0;
    static const bool have = getenv("MICROMIX_STREAM_CFG") && sscanf(getenv("MICROMIX_STREAM_CFG"), "%d,%d,%d", &cf, &cd, &cn) == 3;
    if (have) {
#define MM_TRY(F_, D_, NW_)                                                                              \
    if (cf == F_ && cd == D_ && cn == NW_) return a.M <= 16 ? MM_STREAM(F_, 1, D_, NW_) : (a.M <= 32 ? MM_STREAM(F_, 2, D_, NW_) : MM_STREAM(F_, 4, D_, NW_));
        if (cf == 4 && cd == 2 && cn == 4) return a.M <= 16 ? MM_STREAM(4, 1, 2, 4) : (a.M <= 32 ? MM_STREAM(4, 2, 2, 4) : MM_STREAM(4, 4, 2, 4));
        if (cf == 4 && cd == 2 && cn == 8) return a.M <= 16 ? MM_STREAM(4, 1, 2, 8) : (a.M <= 32 ? MM_STREAM(4, 2, 2, 8) : hipErrorInvalidValue);
        if (cf == 4 && cd == 3 && cn == 4) return a.M <= 16 ? MM_STREAM(4, 1, 3, 4) : (a.M <= 32 ? MM_STREAM(4, 2, 3, 4) : hipErrorInvalidValue);
        MM_TRY(1, 4, 4) MM_TRY(1, 6, 4) MM_TRY(1, 4, 8) MM_TRY(1, 2, 8) MM_TRY(1, 3, 8) MM_TRY(1, 2, 16)
        MM_TRY(2, 3, 4) MM_TRY(2, 4, 4) MM_TRY(2, 4, 8) MM_TRY(2, 3, 8) MM_TRY(2, 2, 8) MM_TRY(2, 2, 16) MM_TRY(2, 1, 16)
#undef MM_TRY
        return hipErrorInvalidValue;
    }
#endif
    // (slots, waves) from sweeps on three boxes (profiles/r04_stream_ablation.txt, section 7): more waves with a shallow ring beat fewer
    // with a deep one; 32 tokens x 32 features: four waves, so that two workgroups fit a CU's LDS
    // more than one round of 32-feature workgroups (fused gate + up, N = 28672): 64 features x 4 waves run in one round -- the same
    // time with resident weights, 1 us less at M = 16 when they come from HBM (18.2 -> 17.1; section 19 of the record)
    if (a.M <= 16 && (a.N + 31) / 32 > 2 * cus) return MM_STREAM(4, 1, 2, 4);
    const int slabs = (a.K[0] + a.K[1] + a.K[2]) >> 7;
    // (few features: four slots when a wave's slabs are at most four -- K <= 4096: everything is requested at once, no phantom steps)
    if (a.M <= 16) return wide ? MM_STREAM(2, 1, 2, 8) : (slabs <= 32 ? MM_STREAM(1, 1, 4, 8) : MM_STREAM(1, 1, 3, 8));
    // 17 .. 32 tokens.  fp4 weights take their scales from images (round 6): on 32 features two slots instead of three keep the second
    // workgroup's place in the LDS (gate/up 10.1-10.8 -> 9.4-9.8 us); on 16 features (one workgroup per CU) three slots while rings and
    // images fit, else two (down_proj 12.0 -> 11.1)
    if (a.M <= 32 && w4) {
        if (wide) return launch_one<2, 2, 2, 4, true>(a, stream);
        return two_tile_fits(3, slabs) ? launch_one<1, 2, 3, 8, true>(a, stream) : launch_one<1, 2, 2, 8, true>(a, stream);
    }
    if (a.M <= 32) return wide ? MM_STREAM(2, 2, 3, 4) : MM_STREAM(1, 2, 3, 8);
    // 33 .. 64 tokens: three / four token tiles; two slots of 8 / 10 KB and four waves, so that two workgroups fit a CU's LDS
    if (a.M <= 48) return wide ? MM_STREAM(2, 3, 2, 4) : MM_STREAM(1, 3, 2, 8);
    return wide ? MM_STREAM(2, 4, 2, 4) : MM_STREAM(1, 4, 2, 8);
#undef MM_STREAM
}

}  // namespace mm
