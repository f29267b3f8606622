// 256x256-tile fused three-segment MX GEMM for gfx950 -- the large-M path of mm_matmul.
//
// Same arithmetic as mx_gemm.hip (see there for the reference citations: gemm.cu:26-78, w4a4.cu,
// w4a6.cu, w4a8.cu, w6a6.cu, w8a8.cu); different machine mapping, chosen from measurements on MI355X
// (tools/mfma_rate.py, profiles/):
//  * v_mfma_scale_f32_32x32x64_f8f6f4 sustains ~3.9 PF with the fp8 operand as srcA but only ~3.0-3.5 PF
//    with it as srcB, and fp6(A) x fp4(B) beats fp4(A) x fp6(B) the same way.  Activations are the wider
//    (or equal) format in every segment, so the ACTIVATION tile is srcA (MFMA rows = tokens) and the
//    WEIGHT tile srcB (MFMA columns = output features).
//  * a 128x128 tile needs ~47 B/clk/CU from L2 at that MFMA rate (L2 peak ~56): 256x256 halves it.
//  * 8 waves as 4 (tokens) x 2 (features), two per SIMD: each wave owns 64 x 128 outputs = 2 x 4 MFMA tiles and per
//    64-deep K step reads 2 activation + 4 weight fragments for 8 MFMAs.  (A 4-wave / one-per-SIMD variant with 128 x
//    128 per wave was measured: an LDS-DMA instruction costs the issuing wave ~180 cycles, more than the 64-cycle MFMA
//    it is meant to hide behind, so a single wave per SIMD leaves the matrix pipe idle ~45 % of the time.)
//  * the 128 fp32 accumulators of a wave live in a[0:127], driven by inline-asm MFMAs.  They are outside the
//    compiler's register allocation on purpose: with compiler-owned accumulators the three fused segment loops
//    made hipcc copy and spill them (hundreds of scratch dwords per lane, measured on ROCm 7.2).
//  * one 128-deep K slab per pipeline stage, three LDS stages (two with fp8 weights); operands AND the
//    scale-factor atoms arrive by LDS-DMA (buffer_load_dwordx4 ... lds) issued up to three slabs ahead behind
//    counted vmcnt waits, one DMA instruction between consecutive MFMAs so that their issue hides in the MFMA
//    shadow; one workgroup barrier per slab, placed between the two K steps, after the second step's
//    fragments are already in registers.
//  * with tokens on MFMA rows the accumulator has the feature index on the lane, so the epilogue transposes
//    each wave's tile through LDS (free at that point) and stores whole 256-byte rows.
#include <type_traits>

#include "mx_acc_regs.h"
#ifndef MM_L2_PREFETCH
#define MM_L2_PREFETCH 0
#endif
#ifndef MM_DBG
#define MM_DBG 0  // kernel-developer ablation switches: 1 = no MFMA, 2 = no DMA (results are garbage)
#endif
#include "mx_common.h"
#include "mx_kernels.h"

namespace mm {

// hipcc parses __device__ bodies in its host pass as well; gfx950 inline asm and target builtins only exist in
// the device pass, so those few bodies are compiled for the device only.
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_DEVICE_ONLY(...)
#endif

namespace g256 {

constexpr int BM = 256, BN = 256, NT = 512;
constexpr int TM = 2, TN = 4;      // MFMA tiles per wave: 64 tokens x 128 features
constexpr int NACC = 16 * TM * TN; // accumulator registers per lane: a[0 : NACC-1]
constexpr int X_TILE = BM * 128;   // bytes, sized for fp8 / padded fp6
constexpr int SF_BYTES = 1024;     // two 512-byte SF atoms (256 rows x 4 blocks)

template <bool W4> struct Lds {
    static constexpr int W_TILE = W4 ? BN * 64 : BN * 128;
    static constexpr int OFF_W = X_TILE;
    static constexpr int OFF_SFX = X_TILE + W_TILE;
    static constexpr int OFF_SFW = OFF_SFX + SF_BYTES;
    static constexpr int STAGE = OFF_SFW + SF_BYTES;
    // A slab is consumed in ~1.1 us but its DMA takes ~3 us to land when ANY of its lines misses L2 (measured:
    // DMA-only loop = 1.55 us per slab with two slabs in flight, independent of the bytes moved).  So the DMA runs
    // NSTAGE slabs ahead: three 50 KB stages fit with fp4 weights, two with fp8 weights.  (An optional cooperative L2
    // prefetch further ahead exists, see SlabDma::prefetch.)
    static constexpr int NSTAGE = W4 ? 3 : 2;
    static constexpr int OFF_DUMP = NSTAGE * STAGE;   // 8 KiB that the L2-prefetch DMAs write into (never read)
    static constexpr int TOTAL = OFF_DUMP + 8192;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    MM_DEVICE_ONLY(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");)
}
// workgroup barrier that does NOT drain the DMA queue (a __syncthreads() would add vmcnt(0))
__device__ __forceinline__ void barrier_lds_only() {
    MM_DEVICE_ONLY(asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");)
}

// ---------------------------------------------------------------------------------------------------------
// fragment types: the register tuple handed to the MFMA has exactly the width the format needs
// ---------------------------------------------------------------------------------------------------------
template <int EL> struct Frag;
template <> struct Frag<EL_FP8> { typedef int type __attribute__((ext_vector_type(8))); static constexpr int PITCH = 128, GBYTES = 128; };
template <> struct Frag<EL_FP6> { typedef int type __attribute__((ext_vector_type(6))); static constexpr int PITCH = 128, GBYTES = 96; };
template <> struct Frag<EL_FP4> { typedef int type __attribute__((ext_vector_type(4))); static constexpr int PITCH = 64, GBYTES = 64; };

// Per-lane byte offsets (inside an operand tile) of the pieces of a fragment, for MFMA tile 0 of the wave and K
// step h.  Tile t adds the compile-time constant t * 32 * PITCH (the swizzle term only depends on row & 31).
// Layouts (measured, tests/test_hw_gpu.py): fp4/fp6 lanes hold the 32 consecutive elements of K block 2h + kb;
// fp8 lanes hold K = 64h + 16kb + [0,16) in registers 0-3 and K = 64h + 32 + 16kb + [0,16) in registers 4-7.
template <int EL> struct FragOfs;
template <> struct FragOfs<EL_FP8> {
    int o[2][2];
    __device__ __forceinline__ void init(int row, int kb) {
        const int sw = (row >> 1) & 7, rb = row * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            o[h][0] = rb + (((4 * h + kb) ^ sw) << 4);
            o[h][1] = rb + (((4 * h + 2 + kb) ^ sw) << 4);
        }
    }
};
template <> struct FragOfs<EL_FP4> {
    int o[2][1];
    __device__ __forceinline__ void init(int row, int kb) {
        const int sw = (row >> 2) & 3, rb = row * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) o[h][0] = rb + (((2 * h + kb) ^ sw) << 4);
    }
};
template <> struct FragOfs<EL_FP6> {
    int o[2][3];
    __device__ __forceinline__ void init(int row, int kb) {
        const int sw = (row >> 1) & 7, rb = row * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int u = 3 * (2 * h + kb) + i;  // 8-byte unit of the 96-byte row
                o[h][i] = rb + (((u >> 1) ^ sw) << 4) + ((u & 1) << 3);
            }
    }
};

template <int EL, int H>
__device__ __forceinline__ typename Frag<EL>::type load_frag(const uint8_t *tile, const FragOfs<EL> &fo, int t) {
    const uint8_t *p = tile + t * 32 * Frag<EL>::PITCH;
    typename Frag<EL>::type r;
    if constexpr (EL == EL_FP8) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(p + fo.o[H][0]);
        const uint4 hi = *reinterpret_cast<const uint4 *>(p + fo.o[H][1]);
        r = typename Frag<EL>::type{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    } else if constexpr (EL == EL_FP4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + fo.o[H][0]);
        r = typename Frag<EL>::type{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    } else {
        const uint2 t0 = *reinterpret_cast<const uint2 *>(p + fo.o[H][0]);
        const uint2 t1 = *reinterpret_cast<const uint2 *>(p + fo.o[H][1]);
        const uint2 t2 = *reinterpret_cast<const uint2 *>(p + fo.o[H][2]);
        r = typename Frag<EL>::type{(int)t0.x, (int)t0.y, (int)t1.x, (int)t1.y, (int)t2.x, (int)t2.y};
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// accumulators in a[0:255]: tile (tm, tn) = a[16 * (tn * TM + tm) .. + 15]
// ---------------------------------------------------------------------------------------------------------
// FENCE: the statement also carries a "memory" clobber, so hipcc cannot hoist the LDS reads that follow it in the source
// above it (used on the first MFMAs after a barrier: the matrix pipe gets work before the wave spends ~100 cycles
// issuing the next fragments' ds_reads).
template <int T, int XEL, int WEL, int H, bool FENCE = false>
__device__ __forceinline__ void mfma_tile(const typename Frag<XEL>::type &x, const typename Frag<WEL>::type &w, int sx,
                                          int sw) {
    // s_nop 1: two wait states between a just-written source VGPR (a compiler v_mov assembling the tuple, the
    // scale shift) and the MFMA that reads it -- hipcc pads nothing inside or in front of an asm statement.
#if !(MM_DBG & 1)
    if constexpr (FENCE) {
        MM_DEVICE_ONLY(asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 a[%c7:%c8], %0, %1, a[%c7:%c8], %2, %3 "
                                    "op_sel_hi:[%c4,%c4,0] cbsz:%c5 blgp:%c6"
                                    :
                                    : "v"(x), "v"(w), "v"(sx), "v"(sw), "i"(H), "i"(ElemTraits<XEL>::HW),
                                      "i"(ElemTraits<WEL>::HW), "i"(16 * T), "i"(16 * T + 15)
                                    : MM_ACC_CLOBBER, "memory");)
    } else {
        MM_DEVICE_ONLY(asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 a[%c7:%c8], %0, %1, a[%c7:%c8], %2, %3 "
                                    "op_sel_hi:[%c4,%c4,0] cbsz:%c5 blgp:%c6"
                                    :
                                    : "v"(x), "v"(w), "v"(sx), "v"(sw), "i"(H), "i"(ElemTraits<XEL>::HW),
                                      "i"(ElemTraits<WEL>::HW), "i"(16 * T), "i"(16 * T + 15)
                                    : MM_ACC_CLOBBER);)
    }
#else
    MM_DEVICE_ONLY(asm volatile("" ::"v"(x), "v"(w), "v"(sx), "v"(sw));)
#endif
}

template <int I>
__device__ __forceinline__ float acc_read() {
    float v = 0.0f;
    MM_DEVICE_ONLY(asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(I) : MM_ACC_CLOBBER);)
    return v;
}
template <int I>
__device__ __forceinline__ void acc_write(float v) {
    MM_DEVICE_ONLY(asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(v), "i"(I) : MM_ACC_CLOBBER);)
}
// an MFMA result may be read by a VALU instruction only 18+ wait states after the MFMA issued (16-pass XDL op)
__device__ __forceinline__ void acc_settle() { MM_DEVICE_ONLY(asm volatile("s_nop 15\n\ts_nop 7" ::: MM_ACC_CLOBBER);) }

template <int I, int N>
struct AccLoop {
    static __device__ __forceinline__ void zero() {
        acc_write<I>(0.0f);
        AccLoop<I + 1, N>::zero();
    }
    static __device__ __forceinline__ void round_bf16() {
        acc_write<I>(bf16_bits_to_f32(f32_to_bf16_bits(acc_read<I>())));
        AccLoop<I + 1, N>::round_bf16();
    }
};
template <int N>
struct AccLoop<N, N> {
    static __device__ __forceinline__ void zero() {}
    static __device__ __forceinline__ void round_bf16() {}
};

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA of one slab, cut into pieces so that one piece can be issued between two MFMAs
// ---------------------------------------------------------------------------------------------------------
// The DMA instructions are issued from inline asm, NOT through the raw_buffer_load_lds builtin: hipcc (ROCm 7.2)
// orders every later ds_read behind a pending builtin LDS-DMA with an s_waitcnt vmcnt(0), which drains the whole
// prefetch queue once per K slab (measured: the loop then runs at DMA latency, ~47 % MFMA utilisation).  With asm
// the compiler does not see the LDS writes; ordering is ours: counted s_waitcnt vmcnt(N), then the workgroup
// barrier, then the reads (all asm statements carry a "memory" clobber).
typedef int rsrc_t __attribute__((ext_vector_type(4)));

// 128-bit raw buffer descriptor {base_lo, base_hi(16 bits) | stride 0, num_records (bytes), flags}, every word made
// provably wave-uniform so that it can be bound to an "s" operand.
__device__ __forceinline__ rsrc_t make_rsrc(const uint8_t *base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(v >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// one buffer_load_dwordx4 ... lds: 64 lanes x 16 B -> LDS bytes [lds_addr, lds_addr + 1024) (M0 = wave-uniform base,
// the hardware adds lane * 16); per-lane source = descriptor base + voff + soff.
//   s_nop 4 : SALU/readfirstlane results (descriptor, soffset) may not be read by a VMEM instruction for 5 states
//   s_nop 0 : one state between the M0 write and the LDS-DMA that reads it
// M0 belongs to the compiler, so it is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const rsrc_t &rsrc, int voff, int soff, unsigned lds_addr) {
#if (MM_DBG & 2)
    MM_DEVICE_ONLY(asm volatile("" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff));)
    return;
#endif
    MM_DEVICE_ONLY(unsigned keep;
                   asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                                : "=&s"(keep)
                                : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
                                : "memory");)
}

// 4-byte variant, used only to pull a cache line into L2 ahead of time (the LDS bytes are never read)
__device__ __forceinline__ void dma4(const rsrc_t &rsrc, int voff, int soff, unsigned lds_addr) {
#if (MM_DBG & 2)
    MM_DEVICE_ONLY(asm volatile("" ::"v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff));)
    return;
#endif
    MM_DEVICE_ONLY(unsigned keep;
                   asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                "buffer_load_dword %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                                : "=&s"(keep)
                                : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
                                : "memory");)
}

__device__ __forceinline__ unsigned lds_address(const uint8_t *p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const uint8_t *)p;
}

// One operand of one segment: descriptor over the tile's rows (rows past the matrix edge read as zero and their
// outputs are masked) and the per-lane offset of DMA piece 0; piece `it` adds it * ROWS_PER_IT * rowbytes.
template <int EL>
struct OperandDma {
    static constexpr int CPR = Frag<EL>::PITCH / 16;       // 16-byte chunks per LDS row
    static constexpr int PIECES = 256 * CPR / NT;          // DMA instructions per thread per slab
    static constexpr int ROWS_PER_IT = NT / CPR;
    rsrc_t rsrc;
    int voff0;
    int rowbytes;
    __device__ __forceinline__ void init(const uint8_t *base, int row0, int nrows, int nslab) {
        rowbytes = __builtin_amdgcn_readfirstlane(nslab * Frag<EL>::GBYTES);
        int rows = nrows - row0;
        rows = rows > 256 ? 256 : rows;
        rsrc = make_rsrc(base + (size_t)row0 * (unsigned)rowbytes, (unsigned)rows * (unsigned)rowbytes);
        const int tid = threadIdx.x;
        const int row = tid / CPR;
        int c;
        if constexpr (CPR == 4) c = (tid & 3) ^ ((row >> 2) & 3);
        else c = (tid & 7) ^ ((row >> 1) & 7);
#if (MM_DBG & 4)
        c = tid & (CPR - 1);  // ablation: linear source order
#endif
        // ROWS_PER_IT is a multiple of 32 for both pitches, so the swizzle term is the same for every piece
        if constexpr (EL == EL_FP6) c = c >= 6 ? c - 6 : c;  // fp6 rows are 96 B: the pad chunks re-read valid bytes
        voff0 = row * rowbytes + c * 16;
    }
    // L2 prefetch of a SLICE of the slab: one wave instruction = 1 KiB = NT/CPR/.. rows starting at tile row `row0`
    // (same per-lane pattern as a DMA piece; the bytes land in a dump area nobody reads).
    __device__ __forceinline__ void prefetch(int slab, int row0, const uint8_t *lds_dump, int wave) const {
        const int lane = threadIdx.x & 63;
        const int row = row0 + lane / CPR;
        dma16(rsrc, row * rowbytes + (lane % CPR) * 16, __builtin_amdgcn_readfirstlane(slab * Frag<EL>::GBYTES),
              __builtin_amdgcn_readfirstlane(lds_address(lds_dump) + wave * 1024));
    }
    template <int IT>
    __device__ __forceinline__ void piece(int slab, const uint8_t *lds_tile, int wave) const {
        dma16(rsrc, voff0 + IT * ROWS_PER_IT * rowbytes, __builtin_amdgcn_readfirstlane(slab * Frag<EL>::GBYTES),
              __builtin_amdgcn_readfirstlane(lds_address(lds_tile) + (IT * NT + wave * 64) * 16));
    }
};

// the two SF atoms (256 rows x 4 blocks = 1 KiB) of one slab: ONE DMA instruction of one wave
struct SfDma {
    rsrc_t rsrc;
    int voff;
    __device__ __forceinline__ void init(const uint8_t *sf, int row0, int row_tiles, int nslab) {
        rsrc = make_rsrc(sf, (unsigned)row_tiles * (unsigned)nslab * 512u);
        const int lane = threadIdx.x & 63;
        int rt = (row0 >> 7) + (lane >> 5);
        rt = rt < row_tiles ? rt : row_tiles - 1;
        voff = rt * nslab * 512 + (lane & 31) * 16;
    }
    __device__ __forceinline__ void issue(int slab, const uint8_t *lds_sf) const {
        dma16(rsrc, voff, __builtin_amdgcn_readfirstlane(slab * 512), __builtin_amdgcn_readfirstlane(lds_address(lds_sf)));
    }
};

template <bool W4, int XEL, int WEL>
struct SlabDma {
    using L = Lds<W4>;
    OperandDma<XEL> x;
    OperandDma<WEL> w;
    SfDma sfx, sfw;
    int wave;
    int rot, nslab;  // the K loop of this tile starts at slab `rot` and wraps around (see mx_gemm256_kernel)
    // logical slab -> slab of the operand; logical indices past the end are clamped to the last slab (the extra DMAs
    // and prefetches of the last iterations are issued anyway so that every iteration queues the same number of
    // memory operations and ONE vmcnt immediate is right; they land in stages nobody reads any more)
    __device__ __forceinline__ int phys(int s) const {
        s = s < nslab ? s : nslab - 1;
        int p = s + rot;
        p = p >= nslab ? p - nslab : p;
        return __builtin_amdgcn_readfirstlane(p);
    }
    // Cooperative L2 prefetch.  The 8 tiles of an XCD chunk that share an activation panel each pull 1/8 of its rows
    // (32 rows = 4 instructions, waves 0-3), the 4 tiles that share a weight panel each pull 1/4 of it (64 rows = 4
    // instructions at 64 B/row, waves 4-7): ONE extra 1-KiB DMA per wave and slab, PF_AHEAD slabs ahead, so that the
    // compulsory L2 misses are taken long before the operand DMA needs the lines.
    // Off by default: measured 3-4 % SLOWER on the production fp8 x fp4 shape (the kernel is power-limited there and the
    // extra 1/12 of L2 traffic costs clock), although it halves the DMA-only loop time.  -DMM_L2_PREFETCH=1 enables it.
    static constexpr int NPF = MM_L2_PREFETCH;
    int pf_xrow, pf_wrow;  // first tile row of this workgroup's slice
    __device__ __forceinline__ void prefetch(int slab_logical, uint8_t *smem) const {
        if constexpr (NPF == 0) return;
        const int slab = phys(slab_logical);
        constexpr int XR = 64 / OperandDma<XEL>::CPR, WR = 64 / OperandDma<WEL>::CPR;  // rows per instruction
        if (wave < 4) x.prefetch(slab, pf_xrow + wave * XR, smem + L::OFF_DUMP, wave);
        else w.prefetch(slab, pf_wrow + (wave - 4) * WR, smem + L::OFF_DUMP, wave);
    }
    static constexpr int NPIECES = OperandDma<XEL>::PIECES + OperandDma<WEL>::PIECES + 1;
    static_assert(NPIECES <= TM * TN + 1, "one DMA piece per MFMA slot");
    // piece P of slab `slab` into `stage`; pieces beyond NPIECES are no-ops
    template <int P>
    __device__ __forceinline__ void piece(int slab_logical, uint8_t *stage) const {
        const int slab = phys(slab_logical);
        if constexpr (P < OperandDma<XEL>::PIECES) {
            x.template piece<P>(slab, stage, wave);
        } else if constexpr (P < OperandDma<XEL>::PIECES + OperandDma<WEL>::PIECES) {
#if (MM_DBG & 8)
            x.template piece<0>(slab, stage + L::OFF_W, wave);  // ablation: weight DMA re-reads activation bytes (same count)
#elif (MM_DBG & 16)
            sfx.issue(slab, stage + L::OFF_SFX);              // ablation: weight DMA replaced by a tiny repeated read
#else
            w.template piece<P - OperandDma<XEL>::PIECES>(slab, stage + L::OFF_W, wave);
#endif
        } else if constexpr (P == NPIECES - 1) {
            // every wave issues exactly NPIECES DMA instructions per slab (so that one vmcnt immediate is right for
            // all of them): even waves bring the activation scales, odd waves the weight scales (the duplicate copy
            // writes the same bytes).
            if (wave & 1) sfw.issue(slab, stage + L::OFF_SFW);
            else sfx.issue(slab, stage + L::OFF_SFX);
        }
    }
    template <int P = 0>
    __device__ __forceinline__ void all(int slab, uint8_t *stage) const {
        if constexpr (P < NPIECES) {
            piece<P>(slab, stage);
            all<P + 1>(slab, stage);
        }
    }
};

struct Scales {
    int x[TM];
    int w[TN];
};

template <bool W4>
__device__ __forceinline__ void load_scales(Scales &sc, const uint8_t *stage, int wm, int wn, int li, int kb) {
    using L = Lds<W4>;
    // SF atom = 128 rows: 16 bytes per (row & 31) = 4 row groups x 4 K blocks.  A wave's 64 activation rows are half an
    // atom (row groups 2*(wm&1) + {0,1} of atom wm>>1); its 128 weight rows are atom wn.
    const uint2 sx = *reinterpret_cast<const uint2 *>(stage + L::OFF_SFX + (wm >> 1) * 512 + li * 16 + (wm & 1) * 8);
    const uint4 sw = *reinterpret_cast<const uint4 *>(stage + L::OFF_SFW + wn * 512 + li * 16);
    const int sh = 8 * kb;  // the lane's K block inside the 64-deep step; op_sel_hi then picks the step
    sc.x[0] = (int)(sx.x >> sh);
    sc.x[1] = (int)(sx.y >> sh);
    sc.w[0] = (int)(sw.x >> sh);
    sc.w[1] = (int)(sw.y >> sh);
    sc.w[2] = (int)(sw.z >> sh);
    sc.w[3] = (int)(sw.w >> sh);
}

// TM*TN MFMAs of K step H on fragment set (xc, wc).  Interleaved with them, in this issue order:
//   after MFMA 0 .. TM-1        : the next step's activation fragment t is read (load_x(t))
//   after MFMA TM .. TM+TN-1    : the next step's weight fragment t is read      (load_w(t))
//   after MFMA p >= DMA_FIRST   : DMA piece p - DMA_FIRST of slab dma_slab (DMA = true only), the rest after the last MFMA
// The first TM+TN MFMAs are memory fences for the compiler, so the reads stay behind them.
template <bool W4, int XEL, int WEL, int H, bool DMA, int P = 0, class LX, class LW>
__device__ __forceinline__ void mfma_step(const typename Frag<XEL>::type (&xc)[TM], const typename Frag<WEL>::type (&wc)[TN],
                                          const Scales &sc, const SlabDma<W4, XEL, WEL> &dma, int dma_slab,
                                          uint8_t *dma_stage, int pf_slab, uint8_t *smem, LX load_x, LW load_w) {
    constexpr int NPIECES = SlabDma<W4, XEL, WEL>::NPIECES;
    constexpr int DMA_FIRST = 1;
    if constexpr (P < TM * TN) {
        constexpr int tn = P / TM, tm = P % TM;
        mfma_tile<tn * TM + tm, XEL, WEL, H, (P < TM + TN)>(xc[tm], wc[tn], sc.x[tm], sc.w[tn]);
        if constexpr (P < TM) load_x(std::integral_constant<int, P>{});
        else if constexpr (P < TM + TN) load_w(std::integral_constant<int, P - TM>{});
        if constexpr (DMA) {
            if constexpr (P >= DMA_FIRST) dma.template piece<P - DMA_FIRST>(dma_slab, dma_stage);
            if constexpr (P == TM * TN - 1) {
                if constexpr (NPIECES > TM * TN - DMA_FIRST) dma.template piece<TM * TN - DMA_FIRST>(dma_slab, dma_stage);
                if constexpr (NPIECES > TM * TN - DMA_FIRST + 1) dma.template piece<TM * TN - DMA_FIRST + 1>(dma_slab, dma_stage);
                dma.prefetch(pf_slab, smem);  // last in the queue order of the iteration
            }
        }
        mfma_step<W4, XEL, WEL, H, DMA, P + 1>(xc, wc, sc, dma, dma_slab, dma_stage, pf_slab, smem, load_x, load_w);
    }
}

// One segment (Q = 0: N, 1: S, 2: O) with its own two-stage DMA pipeline.  On entry no DMA is in flight and every
// wave is past its last LDS read.
template <bool W4, int Q, int XEL, int WEL>
__device__ __forceinline__ void run_segment(const GemmArgs &a, int nslab, int m0, int n0, int rot_num, int rot_den,
                                            int role_x, int role_w, uint8_t *smem) {
    using L = Lds<W4>;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kb = lane >> 5;

    SlabDma<W4, XEL, WEL> dma;
    dma.x.init(a.X[Q], m0, a.M, nslab);
    dma.w.init(a.W[Q], n0, a.N, nslab);
    dma.sfx.init(a.SFX[Q], m0, a.sfx_row_tiles, nslab);
    dma.sfw.init(a.SFW[Q], n0, a.sfw_row_tiles, nslab);
    dma.wave = wave;
    dma.pf_xrow = role_x * 32;
    dma.pf_wrow = role_w * 64;
    dma.nslab = nslab;
    dma.rot = __builtin_amdgcn_readfirstlane((rot_num * nslab) / rot_den);

    constexpr int NS = L::NSTAGE;
    constexpr int NP = SlabDma<W4, XEL, WEL>::NPIECES;
    constexpr int NPF = SlabDma<W4, XEL, WEL>::NPF;
    constexpr int PER_ITER = NP + NPF;   // memory operations every loop iteration queues, in the order DMA.., prefetch..
    constexpr int PF_AHEAD = 2 * NS;     // the L2 prefetch runs this many slabs ahead of the MFMAs
    // prologue: request the first NS slabs (indices past the end are clamped duplicates), wait for slab 0 only
    dma.all(0, smem);
    dma.all(1, smem + L::STAGE);
    if constexpr (NS == 3) dma.all(2, smem + 2 * L::STAGE);
    wait_vmcnt<(NS - 1) * NP>();
    barrier_lds_only();

    FragOfs<XEL> fx;
    FragOfs<WEL> fw;
    fx.init(wm * (TM * 32) + li, kb);
    fw.init(wn * (TN * 32) + li, kb);
    typename Frag<XEL>::type x0[TM], x1[TM];
    typename Frag<WEL>::type w0[TN], w1[TN];
    Scales sc;
#pragma unroll
    for (int t = 0; t < TM; ++t) x0[t] = load_frag<XEL, 0>(smem, fx, t);
#pragma unroll
    for (int t = 0; t < TN; ++t) w0[t] = load_frag<WEL, 0>(smem + L::OFF_W, fw, t);
    load_scales<W4>(sc, smem, wm, wn, li, kb);

    int cur = 0;
    for (int s = 0; s < nslab; ++s) {
        const int nxt = (cur + 1 == NS) ? 0 : cur + 1;
        uint8_t *st_cur = smem + cur * L::STAGE;
        uint8_t *st_nxt = smem + nxt * L::STAGE;
        // K step 0 on (x0, w0); the step-1 fragments are read from the same stage in the shadow of its first MFMAs
        mfma_step<W4, XEL, WEL, 0, false>(
            x0, w0, sc, dma, 0, nullptr, 0, nullptr,
            [&](auto t) { x1[decltype(t)::value] = load_frag<XEL, 1>(st_cur, fx, decltype(t)::value); },
            [&](auto t) { w1[decltype(t)::value] = load_frag<WEL, 1>(st_cur + L::OFF_W, fw, decltype(t)::value); });
        // Wait for this thread's share of slab s+1 and nothing younger.  Slab s+1 was requested in iteration s+1-NS
        // (or in the prologue); everything queued after it may stay in flight:
        //   s >= NS-1 : that iteration's prefetches + (NS-2) whole iterations
        //   s <  NS-1 : the prologue's later slabs + s whole iterations
        if (s >= NS - 1) wait_vmcnt<NPF + (NS - 2) * PER_ITER>();
        else if (NS == 3 && s == 1) wait_vmcnt<PER_ITER>();
        else wait_vmcnt<(NS - 2) * NP>();
        // then the barrier: every wave has its step-1 fragments in registers, so stage[cur] is free, and slab s+1 is
        // complete in LDS
        barrier_lds_only();
        // K step 1 on (x1, w1); meanwhile step 0 of the next slab is read from the next stage, the DMA of slab s+NS
        // refills stage[cur] (one DMA instruction per MFMA) and a slice of slab s+2*NS is pulled into L2.
        Scales scn;
        load_scales<W4>(scn, st_nxt, wm, wn, li, kb);
        mfma_step<W4, XEL, WEL, 1, true>(
            x1, w1, sc, dma, s + NS, st_cur, s + PF_AHEAD, smem,
            [&](auto t) { x0[decltype(t)::value] = load_frag<XEL, 0>(st_nxt, fx, decltype(t)::value); },
            [&](auto t) { w0[decltype(t)::value] = load_frag<WEL, 0>(st_nxt + L::OFF_W, fw, decltype(t)::value); });
        sc = scn;
        cur = nxt;
    }
    wait_vmcnt<0>();
}

template <bool W4>
__global__ void __launch_bounds__(NT) mx_gemm256_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // ---- tile mapping: blocks that share an XCD (blockIdx % 8) get a contiguous chunk of a grouped order ----
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int id = blockIdx.x;
    // K-loop rotation: the operand rows are a power-of-two number of bytes apart (K = 4096 -> 4 KiB / 2 KiB), so
    // tiles that walk K in lockstep ask the L2 for the SAME few channels at the same time (measured: 5.8 TB/s of
    // L2->LDS traffic at 84 % hit rate, far below the L2's rate).  Tile j of an XCD therefore starts its K loop
    // j/32 of the way in and wraps around; sums are order-independent up to fp32 rounding.
    int rot_num, rot_den;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = id & 7, j = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        rot_den = q + 1;
        rot_num = 0;  // K-loop rotation off: measured neutral to slightly negative (it destroys the temporal sharing of
                      // panels in L2); kept as a knob
        (void)j;
    }
    constexpr int GH = 4;  // tiles of a group share their weight panels; 4 x 8 tiles per XCD chunk at 16 x 16
    const int per_group = GH * tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * GH;
    const int gh = (tiles_m - first_m) < GH ? (tiles_m - first_m) : GH;
    const int in = id - grp * per_group;
    const int m0 = __builtin_amdgcn_readfirstlane((first_m + in % gh) * BM);
    const int n0 = __builtin_amdgcn_readfirstlane((in / gh) * BN);
    // cooperative-prefetch roles inside the XCD chunk (speed only): tiles with the same m share the activation panel (8
    // consecutive n per chunk at GH = 4), tiles with the same n share the weight panel (gh of them)
    const int role_x = __builtin_amdgcn_readfirstlane((in / gh) & 7);
    const int role_w = __builtin_amdgcn_readfirstlane((in % gh) & 3);

    unsigned long long t0 = 0, r0 = 0;
    if (a.clock_out != nullptr) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int n0s = a.K[0] >> 7, n1s = a.K[1] >> 7, n2s = a.K[2] >> 7;
#if !(MM_DBG & 128)
    AccLoop<0, NACC>::zero();
#endif

    if (n0s) run_segment<W4, 0, EL_FP4, EL_FP4>(a, n0s, m0, n0, rot_num, rot_den, role_x, role_w, smem);
    if (n1s) {
        if (n0s) {
            if (a.round_per_segment) {
                acc_settle();
                AccLoop<0, NACC>::round_bf16();
            }
            __syncthreads();  // the previous segment's last LDS reads are done before the stages are refilled
        }
        run_segment<W4, 1, EL_FP6, (W4 ? EL_FP4 : EL_FP6)>(a, n1s, m0, n0, rot_num, rot_den, role_x, role_w, smem);
    }
    if (n2s) {
        if (n0s | n1s) {
            if (a.round_per_segment) {
                acc_settle();
                AccLoop<0, NACC>::round_bf16();
            }
            __syncthreads();
        }
        run_segment<W4, 2, EL_FP8, (W4 ? EL_FP4 : EL_FP8)>(a, n2s, m0, n0, rot_num, rot_den, role_x, role_w, smem);
    }
    __syncthreads();  // every wave is done with the operand stages: LDS is reused for the transpose
    acc_settle();
    if (a.clock_out != nullptr && threadIdx.x == 0) {
        a.clock_out[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        a.clock_out[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
        a.clock_out[4 * blockIdx.x + 2] = r0;  // absolute start (100 MHz ticks)
    }

    // ---- epilogue: bf16 (+bias) -> LDS [32 rows][128 cols] per wave -> 16-byte row-contiguous stores ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    uint8_t *reg = smem + wave * 8192;
    float bias[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * 128 + tn * 32 + li;
        bias[tn] = (a.bias != nullptr && n < a.N) ? bf16_bits_to_f32(a.bias[n]) : 0.0f;
    }
    const bool vec_ok = (a.N & 7) == 0;
    auto store_rows = [&](int tm) {
        // wave-local hand-over: the same wave reads back what it wrote (LDS is in order per wave, no barrier needed)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rowl = j * 4 + (lane >> 4), ch = lane & 15;
            const uint4 v = *reinterpret_cast<const uint4 *>(reg + rowl * 256 + ch * 16);
            const int m = m0 + wm * (TM * 32) + tm * 32 + rowl;
            const int n = n0 + wn * 128 + ch * 8;
            if (m < a.M && !(MM_DBG & 32)) {
                uint16_t *dst = a.D + (size_t)m * a.N + n;
                if (vec_ok && n + 7 < a.N) {
                    *reinterpret_cast<uint4 *>(dst) = v;
                } else {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < a.N) dst[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    };
#define MM_EPI_TILE(TM_, TN_)                                                                                      \
    {                                                                                                              \
        float v[16];                                                                                               \
        v[0] = acc_read<16 * (TN_ * TM + TM_) + 0>();   v[1] = acc_read<16 * (TN_ * TM + TM_) + 1>();              \
        v[2] = acc_read<16 * (TN_ * TM + TM_) + 2>();   v[3] = acc_read<16 * (TN_ * TM + TM_) + 3>();              \
        v[4] = acc_read<16 * (TN_ * TM + TM_) + 4>();   v[5] = acc_read<16 * (TN_ * TM + TM_) + 5>();              \
        v[6] = acc_read<16 * (TN_ * TM + TM_) + 6>();   v[7] = acc_read<16 * (TN_ * TM + TM_) + 7>();              \
        v[8] = acc_read<16 * (TN_ * TM + TM_) + 8>();   v[9] = acc_read<16 * (TN_ * TM + TM_) + 9>();              \
        v[10] = acc_read<16 * (TN_ * TM + TM_) + 10>(); v[11] = acc_read<16 * (TN_ * TM + TM_) + 11>();            \
        v[12] = acc_read<16 * (TN_ * TM + TM_) + 12>(); v[13] = acc_read<16 * (TN_ * TM + TM_) + 13>();            \
        v[14] = acc_read<16 * (TN_ * TM + TM_) + 14>(); v[15] = acc_read<16 * (TN_ * TM + TM_) + 15>();            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
            uint32_t b = f32_to_bf16_bits(v[r]);                                                                   \
            if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bias[TN_]);                          \
            const int rowl = (r & 3) + 8 * (r >> 2) + 4 * hi;                                                      \
            *reinterpret_cast<uint16_t *>(reg + rowl * 256 + (TN_ * 32 + li) * 2) = (uint16_t)b;                   \
        }                                                                                                          \
    }
#define MM_EPI_ROW(TM_)                                                                                            \
    MM_EPI_TILE(TM_, 0) MM_EPI_TILE(TM_, 1) MM_EPI_TILE(TM_, 2) MM_EPI_TILE(TM_, 3) store_rows(TM_);
#if !(MM_DBG & 64)
    MM_EPI_ROW(0)
    MM_EPI_ROW(1)
#endif
#undef MM_EPI_ROW
#undef MM_EPI_TILE
    if (a.clock_out != nullptr && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left
        a.clock_out[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

}  // namespace g256

hipError_t launch_mx_gemm256(const GemmArgs &a, bool w4, hipStream_t stream) {
    using namespace g256;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static bool attr_done[2] = {false, false};
    if (w4) {
        if (!attr_done[1]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mx_gemm256_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, Lds<true>::TOTAL);
            if (e != hipSuccess) return e;
            attr_done[1] = true;
        }
        hipLaunchKernelGGL(mx_gemm256_kernel<true>, dim3(tiles), dim3(NT), Lds<true>::TOTAL, stream, a);
    } else {
        if (!attr_done[0]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mx_gemm256_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, Lds<false>::TOTAL);
            if (e != hipSuccess) return e;
            attr_done[0] = true;
        }
        hipLaunchKernelGGL(mx_gemm256_kernel<false>, dim3(tiles), dim3(NT), Lds<false>::TOTAL, stream, a);
    }
    return hipGetLastError();
}

}  // namespace mm
