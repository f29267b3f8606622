// Tiled fused three-segment MX GEMM for gfx950 -- the M > 64 path of mm_matmul: 256x256 tiles when they fill the chip, 128x256 /
// 128x128 tiles (same 8-wave body) or 64x128 / 64x64 tiles (4 compute + 4 loader waves) for launches with fewer tiles, split-K
// through a caller-provided workspace for the fewest; plan_tiles() chooses, mm_matmul_describe() reports the choice.
// The notes below are about the 256x256 tile; mx_gemm_tile.inc documents the body and the 64-row pipeline.
//
// Arithmetic and reference citations: see mx_gemm.hip (gemm.cu:26-78, w4a4.cu, w4a6.cu, w4a8.cu, w6a6.cu,
// w8a8.cu).  Machine mapping, chosen from measurements on MI355X
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
#include "mx_gemm_prelude.h"

namespace mm {

#define MM_NS g256
#ifdef MM_G256_MAX_STAGES          // developer probe (tools/delivery_probes.sh): the 256-row tile's 128-deep segments on two stages
#define MM_MAX_STAGES MM_G256_MAX_STAGES
#else
#define MM_MAX_STAGES 3
#endif
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 4
#define MM_TM 2
#define MM_TN 4
#define MM_ACC MM_ACC_CLOBBER
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET
#define MM_NS g128
#define MM_MAX_STAGES 3
#define MM_LDS_BUDGET (160 * 1024)
#define MM_WM 4
#define MM_TM 1
#define MM_TN 4
#define MM_ACC MM_ACC_CLOBBER
#include "mx_gemm_tile.inc"
#undef MM_NS
#undef MM_WM
#undef MM_TM
#undef MM_TN
#undef MM_ACC
#undef MM_MAX_STAGES
#undef MM_LDS_BUDGET

// ---------------------------------------------------------------------------------------------------------
// split-K for shapes with few tiles (medium M, or small N): 128 x 256 tiles x `splits` workgroups, each on a slab range
// of one segment, raw fp32 partial sums in the caller's workspace, then a reduction kernel that adds them in split
// order (deterministic) and applies the reference's rounding chain D = bf16(N); D = bf16(S + D); D = bf16(O + D).
// ---------------------------------------------------------------------------------------------------------
constexpr int SPLIT_WG_FLOATS = g128::NACC * g128::NT;  // 32768 floats = 128 KiB of partial sums per workgroup

__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs a, int tiles_n, int total) {
    // one thread = four consecutive accumulator registers of one lane of the GEMM workgroup (16 bytes of every split's partial sums,
    // the threads' pieces side by side): four consecutive tokens of one feature
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= total) return;
    // e = (tile, register quad r4, thread tid of the GEMM workgroup, 4 registers): the layout AccLoop::store writes
    const int tid = (e >> 2) & (g128::NT - 1), r4 = (e >> 11) & (g128::NACC / 4 - 1), tile = e >> 15;
    const float *p = a.ws + (size_t)tile * a.splits * SPLIT_WG_FLOATS + (e & (SPLIT_WG_FLOATS - 1));
    float run[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int seg = 0; seg < 3; ++seg) {
        if (a.split_first[seg + 1] == a.split_first[seg]) continue;
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = a.split_first[seg]; q < a.split_first[seg + 1]; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)q * SPLIT_WG_FLOATS);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = s[i] + run[i];
            run[i] = a.round_per_segment ? bf16_bits_to_f32(f32_to_bf16_bits(t)) : t;
        }
    }
    // accumulator layout of the tile kernel: wave = 2 * wm + wn owns tokens wm*32.., features wn*128..; register r of
    // MFMA tile tn = r >> 4: token (r & 3) + 8 * ((r >> 2) & 3) + 4 * (lane >> 5), feature tn*32 + (lane & 31); r = 4 * r4 + i
    const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
    const int m = (tile / tiles_n) * g128::BM + wm * 32 + 8 * (r4 & 3) + 4 * (lane >> 5);      // + i
    const int n = (tile % tiles_n) * g128::BN + wn * 128 + (r4 >> 2) * 32 + (lane & 31);
    if (n >= a.N) return;
    if (a.out_f32) {     // MM_OUT_F32: the fp32 sums themselves
        float *d32 = reinterpret_cast<float *>(a.D) + (size_t)m * a.N + n;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (m + i < a.M) d32[(size_t)i * a.N] = run[i];
        return;
    }
    const float bias = a.bias != nullptr ? bf16_bits_to_f32(a.bias[n]) : 0.0f;
    uint16_t *dst = a.D + (size_t)m * a.N + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t b = f32_to_bf16_bits(run[i]);
        if (a.bias != nullptr) b = f32_to_bf16_bits(bf16_bits_to_f32(b) + bias);
        if (m + i < a.M) dst[(size_t)i * a.N] = (uint16_t)b;     // a wave's 32 lanes of one token: 64 bytes side by side
    }
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Number of K splits for (M, N, K) and the segment each split works on; 0 = do not split.
// Cost model fitted to MI355X measurements (tools/bench_shapes.py): a workgroup of a launch with few tiles walks one
// 128-deep slab in ~0.7 us (DMA-latency bound), and every split adds tiles * 128 KiB of fp32 partial sums that are
// written and read back at ~4 TB/s (0.064 us per tile and split), plus ~5 us for the second launch:
//     t(S) = 0.7 * slabs / S + 0.064 * tiles * S (+ 5)   ->   S* = 3.3 * sqrt(slabs / tiles)
// Split only when that beats the unsplit launch (the tile plan_tiles would pick at these tile counts, see below) by a margin.
// `force` (MM_SPLIT_K_ALWAYS, tests and tuning) skips the model.  MICROMIX_SPLITK=0 disables splitting, =S pins the split count.
static int plan_splits(int M, int N, const int K[3], bool w4, bool force, int first[4]) {
    static const int pinned = env_int("MICROMIX_SPLITK", -1);
    if (pinned == 0 || M <= 32) return 0;
    const int tiles = ((M + g128::BM - 1) / g128::BM) * ((N + g128::BN - 1) / g128::BN);
    int n[3], total = 0, nonempty = 0;
    for (int i = 0; i < 3; ++i) {
        n[i] = K[i] >> 7;
        total += n[i];
        nonempty += n[i] > 0;
    }
    int cap = device_cus() / tiles; // one wave of workgroups on the CUs
    cap = cap > 16 ? 16 : cap;
    cap = cap > total / 2 ? total / 2 : cap;  // at least two slabs per split on average
    if (cap < 2 || cap < nonempty) return 0;
    int S;
    if (pinned > 0 || force) {
        S = pinned > 0 ? pinned : cap;
    } else {
        S = (int)(3.3f * sqrtf((float)total / (float)tiles) + 0.5f);
        S = S < nonempty ? nonempty : S;
        S = S > cap ? cap : S;
        // unsplit: 128 x 128 tiles walk a slab in ~0.5 us; the 4-wave tiles (when plan_tiles would pick them: one round of
        // workgroups) in ~0.32 us (64 x 64) / ~0.36 us (64 x 128).  Margins fitted to tools/mid_m_sweep.py (down_proj, K = 14336:
        // split at M <= 384, unsplit 64 x 128 tiles at M = 512; k/v and q/o at K = 4096: never split).
        const int t32n = ((M + 63) / 64) * ((N + 63) / 64), t32 = ((M + 63) / 64) * ((N + 127) / 128);
        const int t64 = ((M + 127) / 128) * ((N + 127) / 128);
        const int cus = device_cus();
        const bool g32n_fits = t32n <= cus, g32_fits = !g32n_fits && t32 <= cus && 2 * t64 <= cus;   // as plan_tiles
        const float unsplit = (g32n_fits ? 0.32f : g32_fits ? 0.36f : 0.5f) * total, margin = g32_fits ? 0.95f : 0.85f;
        const float split = 0.7f * total / S + 0.064f * tiles * S + 5.0f;
        if (S < 2 || split > margin * unsplit) return 0;
    }
    S = S > cap ? cap : S;
    if (S < 2 || S < nonempty) return 0;
    // segments get splits in proportion to their slab counts, at least one each and never more than their slabs
    int parts[3], used = 0;
    for (int i = 0; i < 3; ++i) {
        parts[i] = n[i] ? (S * n[i]) / total : 0;
        if (n[i] && parts[i] == 0) parts[i] = 1;
        used += parts[i];
    }
    while (used != S) {
        int best = -1;
        for (int i = 0; i < 3; ++i) {
            if (!n[i]) continue;
            if (used < S) {  // give a split to the segment with the longest slab run per split
                if (parts[i] < n[i] && (best < 0 || n[i] * parts[best] > n[best] * parts[i])) best = i;
            } else {         // take one from the segment with the shortest
                if (parts[i] > 1 && (best < 0 || n[i] * parts[best] < n[best] * parts[i])) best = i;
            }
        }
        if (best < 0) break;
        parts[best] += used < S ? 1 : -1;
        used += used < S ? 1 : -1;
    }
    first[0] = 0;
    for (int i = 0; i < 3; ++i) first[i + 1] = first[i] + parts[i];
    return first[3];
}

// ---------------------------------------------------------------------------------------------------------
// In-kernel split-K of the 4-wave tiles (split_tile_reduce in mx_gemm_tile.inc): a launch whose 64 x 64 tiles occupy at most half
// of the CUs cuts K into `splits` equal slab ranges, one workgroup each, and the last workgroup of a tile to finish
// reduces.  Workspace: [tickets: MM_TICKET_BYTES, one 32-bit counter per tile, zero between launches][tiles x (<= splits + 2)
// partial-sum slots].
// What it is worth, measured on MI355X.  Phases of ONE cold launch (tools/split_clock.py, in-kernel clock stamps): a workgroup needs
// ~2.5-3.5 us from its start to its first finished slab and ~0.3 us per 128-deep slab after that; the split adds ~0.6 us (partial
// sums acknowledged, ticket) and one round trip of ~2 us for the reducing workgroup's reads (the partial sums are written through to
// memory: the eight XCDs' L2s are not coherent for plain accesses).  Kernel durations in a queue of back-to-back launches
// (tools/split_small_sweep.py: dispatch events, settled; K = 4096, (2048,128,1920)), unsplit / 2 / 4 / 6 splits:
//   32 tiles (k/v, N = 1024, M = 128)   11.3 / 10.7 /  9.5 /  8.9 us        64 tiles (N = 2048, M = 128)  12.7 / 11.2 / 11.5 / 15.2
//   48 tiles (N = 1024, M = 192)        11.4 / 10.9 / 10.0 / 12.6           128 tiles (q/o, M = 128)      12.9 / 12.0 / 19.8 / 19.9
//   64 tiles (N = 1024, M = 256)        11.2 / 11.2 / 10.8 / 13.9           256 tiles (q/o, M = 256)      12.8 / 22.7
// More than one round of workgroups (tiles x splits > CUs) always loses.  Two splits at 64 ... 128 tiles are a wash: an A/B on a second
// box (alternating processes, profiles/notes_r03.md section 14) had the unsplit launch at 11.4 / 11.4 / 12.7 us and two splits at
// 12.0 / 12.3 / 12.0 for q/o at M = 128.  Hence the rule below: six splits up to 32 tiles, four up to 48, none above, at least four
// slabs per split.  MICROMIX_SPLIT_SMALL=0 disables it, =<tile>:<S> (tile 33 = 64 x 64,
// 32 = 64 x 128; the latter never won) pins a plan (tools, tests); MM_SPLIT_K_ALWAYS relaxes the rule to "any split that fits one
// round of workgroups".
// ---------------------------------------------------------------------------------------------------------
struct SmallSplit {
    int kind;     // 0 = none, 33 = 64 x 64 tiles, 32 = 64 x 128 tiles
    int splits;
    int tiles;
};
constexpr size_t small_part_bytes(int kind) { return kind == 33 ? SMALL_PART_BYTES_64x64 : SMALL_PART_BYTES_64x128; }   // (mx_gemm_prelude.h; asserted in mx_gemm_tiles_small.hip)
constexpr size_t MM_TICKET_BYTES = 4096;   // = mm_matmul_ticket_bytes(): room for 1024 tiles (a launch of this kind has at most CUs / 2)
static size_t small_ticket_bytes(int) { return MM_TICKET_BYTES; }
// a split leaves one partial sum per segment it touches: at most splits + 2 slots per tile
static size_t small_split_bytes(const SmallSplit &p) { return p.kind ? small_ticket_bytes(p.tiles) + (size_t)p.tiles * (p.splits + 2) * small_part_bytes(p.kind) : 0; }

static SmallSplit plan_small_split(int M, int N, const int K[3], bool force) {
    SmallSplit none{0, 0, 0};
    static const char *pin = getenv("MICROMIX_SPLIT_SMALL");
    if (M <= 64 || (pin && pin[0] == '0' && pin[1] == 0)) return none;   // (M <= 64: the weight-streaming kernels, mx_gemm.hip)
    const int cus = device_cus();
    const int total = (K[0] >> 7) + (K[1] >> 7) + (K[2] >> 7);
    const int t32n = ((M + 63) / 64) * ((N + 63) / 64), t32 = ((M + 63) / 64) * ((N + 127) / 128);
    if (pin && pin[0]) {
        int kind = 0, S = 0;
        if (sscanf(pin, "%d:%d", &kind, &S) == 2 && (kind == 33 || kind == 32) && S >= 2 && S <= MM_MAX_SPLITS && S <= total)
            return SmallSplit{kind, S, kind == 33 ? t32n : t32};
        return none;
    }
    if (force) {    // tests: the most splits that fit one round of workgroups, 64 x 64 tiles while those fit, else 64 x 128
        for (int kind : {33, 32}) {
            const int tiles = kind == 33 ? t32n : t32;
            int S = cus / (tiles > 0 ? tiles : 1);
            S = S > MM_MAX_SPLITS ? MM_MAX_SPLITS : S;
            S = S > total / 2 ? total / 2 : S;
            if (S >= 2) return SmallSplit{kind, S, tiles};
        }
        return none;
    }
    // measured rule (see above): six splits up to 32 tiles, four up to 48, none above (two splits at 64 ... 128 tiles measured equal
    // to the unsplit launch within the box-to-box spread); never more than one round of workgroups, at least four slabs per split
    // (longer K -- down_proj, 112 slabs -- stays with the two-launch split-K of the 128-row tiles, which cuts it into up to 16)
    if (total > 64) return none;
    int S = t32n <= 32 ? 6 : t32n <= 48 ? 4 : 0;
    S = S > total / 4 ? total / 4 : S;
    S = (S == 5 || S == 3) ? S - 1 : S;
    return S >= 2 && t32n * S <= cus ? SmallSplit{33, S, t32n} : none;
}

size_t mx_gemm_workspace_bytes(int M, int N, const int K[3], bool w4, bool force, bool tickets_zeroed) {
    const SmallSplit sp = tickets_zeroed ? plan_small_split(M, N, K, force) : SmallSplit{0, 0, 0};
    if (sp.kind) return small_split_bytes(sp);
    int first[4];
    const int S = plan_splits(M, N, K, w4, force, first);
    if (S == 0) return 0;
    const size_t tiles = (size_t)((M + g128::BM - 1) / g128::BM) * ((N + g128::BN - 1) / g128::BN);
    // with MM_WS_TICKETS_ZEROED the head of the workspace belongs to the ticket counters: the partial sums start behind it
    return (tickets_zeroed ? MM_TICKET_BYTES : 0) + tiles * S * SPLIT_WG_FLOATS * sizeof(float);
}

// Which kernel(s) a problem runs on: decided once here, used by the launcher and by mm_matmul_describe.
enum TileKind { TK_SPLITK, TK_SMALL_SPLIT, TK_G64, TK_G256_TAIL, TK_G256, TK_G128, TK_G32, TK_G32N, TK_G16 };
struct TilePlan {
    TileKind kind;
    int tn, tiles256, tiles128, tiles64, tiles32, tiles32n, tiles16, tm256, tm128, tail_cols;
    int splits, split_first[4];
    SmallSplit small;
};

static TilePlan plan_tiles(int M, int N, const int K[3], bool w4, bool have_ws, size_t ws_bytes, bool force_split, bool tickets_zeroed) {
    TilePlan p{};
    p.tn = (N + 255) / 256;
    p.tm256 = (M + 255) / 256;
    p.tm128 = (M + 127) / 128;
    p.tiles256 = p.tm256 * p.tn;
    p.tiles128 = p.tm128 * p.tn;
    p.tiles64 = p.tm128 * ((N + 127) / 128);
    p.tiles32 = ((M + 63) / 64) * ((N + 127) / 128);
    p.tiles32n = ((M + 63) / 64) * ((N + 63) / 64);
    p.tiles16 = ((M + 31) / 32) * ((N + 63) / 64);
    if (have_ws) {
        p.small = tickets_zeroed ? plan_small_split(M, N, K, force_split) : SmallSplit{0, 0, 0};
        if (p.small.kind && p.small.tiles * 4 <= (int)MM_TICKET_BYTES && small_split_bytes(p.small) <= ws_bytes) {
            p.kind = TK_SMALL_SPLIT;
            p.splits = p.small.splits;
            return p;
        }
        p.small = SmallSplit{0, 0, 0};
        p.splits = plan_splits(M, N, K, w4, force_split, p.split_first);
        if (p.splits && (tickets_zeroed ? MM_TICKET_BYTES : 0) + (size_t)p.tiles128 * p.splits * SPLIT_WG_FLOATS * sizeof(float) <= ws_bytes) {
            p.kind = TK_SPLITK;
            return p;
        }
        p.splits = 0;
    }
    static const int force = env_int("MICROMIX_GEMM_TILE", 0);   // kernel-developer override: 256, 128 or 64
    static const int tail_split = env_int("MICROMIX_GEMM_TAIL", 1);
    const int cus = device_cus();
    if (force == 32 || force == 33 || force == 16) {
        p.kind = force == 32 ? TK_G32 : force == 33 ? TK_G32N : TK_G16;
        return p;
    }
    // 64 x 64 tiles on at most half of the CUs (q/o at M <= 128: 128 tiles): 32 x 64 tiles (two compute + two loader waves) double the
    // workgroups; every workgroup then walks its K with half the LDS traffic per slab (round 4, tools/time_cases.py)
    static const int use16 = env_int("MICROMIX_GEMM_TILE16", 1);
    if (force == 0 && use16 && p.tiles16 <= 2 * cus && (2 * p.tiles32n <= cus || M <= 32)) {   // (M <= 32: one row of tiles, see mx_gemm_small_m_uses_tiles)
        p.kind = TK_G16;
        return p;
    }
    // Fewer than a CU's worth of 128-row tiles: the 4-wave tiles (64 rows, loader / compute waves, see mx_gemm_tile.inc) fill more
    // CUs.  64 x 64 while those fit one round of workgroups (M <= 256 at N = 4096: 11-12 us against 17 us at K = 4096), 64 x 128
    // while THEY fit one round and the 128 x 128 tiles would leave half of the CUs idle (M = 512 at N = 4096: 14.7 against 17.7).
    if (force == 0 && p.tiles32n <= cus) {
        p.kind = TK_G32N;
        return p;
    }
    if (force == 0 && p.tiles32 <= cus && 2 * p.tiles64 <= cus) {
        p.kind = TK_G32;
        return p;
    }
    // One workgroup fits per CU, so a launch runs in rounds of `cus` tiles.  256-row tiles move the fewest L2->LDS bytes per
    // flop; a round of 128-row tiles takes ~0.62 of a round of 256-row tiles (measured).  So 128-row tiles pay exactly when
    // they still fit in ONE round (tiles128 <= cus, i.e. at most half of the CUs would get a 256-row tile): M=2048, N=4096
    // 46.5 -> 33 us; with 160 256-row tiles (320 128-row tiles = two rounds) the 256-row tiles win, 62 vs 76 us.
    const bool use128 = force == 128 || (force != 256 && force != 64 && p.tiles128 <= cus);
    // ... and when even the 128-row tiles would occupy at most half of the CUs, 128 x 128 tiles double the workgroups once
    // more (1.5x the L2->LDS bytes per flop, which does not matter while half of the chip idles): M = 1024, N = 4096.
    if (force == 64 || (force == 0 && 2 * p.tiles128 <= cus && p.tiles64 <= cus)) {
        p.kind = TK_G64;
        return p;
    }
    // Tail balancing: tiles256 = q * CUs + R runs q + 1 rounds and the last one leaves CUs idle.
    // When R <= CUs / 2 and R is a whole number of tile columns, those columns are run as 128-row tiles instead (2R
    // workgroups of half the work: the last round takes half the time).  gate/up at M = 4096: 896 tiles = 3.5 rounds.
    const int rem = p.tiles256 % cus;
    if (!use128 && tail_split && force == 0 && p.tiles256 > cus && rem > 0 && 2 * rem <= cus && rem % p.tm256 == 0) {
        p.kind = TK_G256_TAIL;
        p.tail_cols = rem / p.tm256;
        return p;
    }
    p.kind = use128 ? TK_G128 : TK_G256;
    return p;
}

// 32 < M <= 64: the weight-streaming kernel (two token tiles per workgroup, N / 32 workgroups) against the tiles.  Measured
// (tools/mid_m_sweep.py 40 48 56 64): the skinny kernel costs ~10-13 us per ROUND of its workgroups at K = 4096, the 64-row tiles
// 11-14 us flat, so the tiles win once N / 32 exceeds the CUs (gate/up, N = 14336: 22-25 -> 14 us); and a long K that the model
// would split (down_proj, K = 14336: 25-30 us skinny) goes to the split-K tiles when the caller brought a workspace.
bool mx_gemm_small_m_uses_tiles(int M, int N, const int K[3], bool w4, size_t ws_bytes, bool force_split) {
    // 16 < M <= 32: only with more than three rounds of 32-feature workgroups (fused gate + up, N = 28672: the 64-row tiles take
    // 19.2-20.0 us at M = 17 / 24 / 32, the weight-streaming kernel 20.7 / 22.1 / 23.9; at N <= 24576 the latter wins or ties:
    // tools/time_skinny_big_n.py, profiles/notes_r03.md section 24)
    // (round 4: with the 32 x 64 tiles also M <= 16 -- N = 28672, us at M = 1 / 8 / 16 / 24 / 32: weight-streaming or 64 x 64 tiles
    // 17.4 / 18.6 / 20.3 / 18.6 / 18.6, 32 x 64 tiles 16.7 / 16.8 / 16.7 / 16.9 / 16.9; at N <= 14336 the weight-streaming kernel wins by 1-4 us)
    // (round 4, later: the second weight-streaming kernel, mx_gemm_stream.hip, takes 14.3-14.6 us at N = 28672 and M <= 16, 20.2 at
    // M = 32 where the 32 x 64 tiles take 16.8; MICROMIX_SMALL_M_TILES=1 restores the rule above for A/B runs)
    static const int tiles_le16 = env_int("MICROMIX_SMALL_M_TILES", 0);
    if (M <= 16 && !tiles_le16) return false;
    if (M <= 32) return (N + 31) / 32 > 3 * device_cus();
    if (M <= 32 || M > 64) return M > 64;
    static const int mid_stream = env_int("MICROMIX_MID_M_STREAM", 0);      // kernel-developer override: 32 < M <= 64 always on the weight-streaming kernel
    if (mid_stream) return false;
    if ((N + 31) / 32 > device_cus()) return true;
    // 32 < M <= 64 on one round of 32 x 64 tiles (round 4; profiles/r04_small_tiles.txt, (2048,128,1920), us): a workgroup of those
    // tiles walks K = 4096 in ~9.8 us however many of them there are.  The second weight-streaming kernel (mx_gemm_stream.hip, three /
    // four token tiles) takes 6.7 / 6.8 / 7.7 at M = 33 / 48 / 64 for N = 4096 and 14.0 / 14.7 / 17.3 on down_proj's K = 14336 (the
    // two-launch split-K: 19.5), but 14.1-16.3 against the tiles' 13.5-14.0 at N = 14336: so the tiles from N / 32 > CUs / 2 on, the
    // streaming kernel below, and the split-K plan only when the caller forces it.
    {
        static const int small16 = env_int("MICROMIX_SMALL_M_TILE16", 1);
        const int t16 = ((M + 31) / 32) * ((N + 63) / 64);
        if (small16 && t16 <= device_cus() && 2 * ((N + 31) / 32) > device_cus()) return true;
    }
    if (ws_bytes == 0 || !force_split) return false;
    const size_t need = mx_gemm_workspace_bytes(M, N, K, w4, force_split, false);   // (the in-kernel split starts above M = 64)
    return need > 0 && need <= ws_bytes;
}

const char *describe_mx_gemm256(int M, int N, const int K[3], bool w4, size_t ws_bytes, bool force_split, bool tickets_zeroed) {
    static thread_local char buf[192];
    const TilePlan p = plan_tiles(M, N, K, w4, ws_bytes > 0, ws_bytes, force_split, tickets_zeroed);
    const char *w = w4 ? "true" : "false";
    switch (p.kind) {
        case TK_SPLITK: snprintf(buf, sizeof(buf), "mm::g128::mx_gemm256_kernel<%s,true> x %d workgroups (128x256 tiles, split-K %d) + mm::splitk_reduce_kernel", w, p.tiles128 * p.splits, p.splits); break;
        case TK_SMALL_SPLIT: snprintf(buf, sizeof(buf), "mm::%s::mx_gemm256_kernel<%s,true> x %d workgroups (%s tiles, in-kernel split-K %d)", p.small.kind == 33 ? "g32n" : "g32", w, p.small.tiles * p.small.splits, p.small.kind == 33 ? "64x64" : "64x128", p.small.splits); break;
        case TK_G64: snprintf(buf, sizeof(buf), "mm::g64::mx_gemm256_kernel<%s,false> x %d workgroups (128x128 tiles)", w, p.tiles64); break;
        case TK_G32: snprintf(buf, sizeof(buf), "mm::g32::mx_gemm256_kernel<%s,false> x %d workgroups (64x128 tiles)", w, p.tiles32); break;
        case TK_G32N: snprintf(buf, sizeof(buf), "mm::g32n::mx_gemm256_kernel<%s,false> x %d workgroups (64x64 tiles)", w, p.tiles32n); break;
        case TK_G16: snprintf(buf, sizeof(buf), "mm::g16::mx_gemm256_kernel<%s,false> x %d workgroups (32x64 tiles)", w, p.tiles16); break;
        case TK_G256_TAIL:
            snprintf(buf, sizeof(buf), "mm::g256::mx_gemm256_kernel<%s,false> x %d workgroups (256x256 tiles) + mm::g128::mx_gemm256_kernel<%s,false> x %d (last %d tile columns as 128x256 tiles)", w, p.tm256 * (p.tn - p.tail_cols), w, p.tm128 * p.tail_cols, p.tail_cols);
            break;
        case TK_G256:
            snprintf(buf, sizeof(buf), "mm::g256::mx_gemm256_kernel<%s,false> x %d workgroups (256x256 tiles)", w, p.tiles256);
            break;
        default: snprintf(buf, sizeof(buf), "mm::g128::mx_gemm256_kernel<%s,false> x %d workgroups (128x256 tiles)", w, p.tiles128); break;
    }
    return buf;
}

hipError_t launch_mx_gemm256(const GemmArgs &a, bool w4, hipStream_t stream) {
    static DynamicLdsOnce done[12];
    const TilePlan p = plan_tiles(a.M, a.N, a.K, w4, a.ws != nullptr, a.ws_bytes, a.force_split != 0, a.tickets_zeroed != 0);
    switch (p.kind) {
        case TK_SPLITK: {
            GemmArgs b = a;
            b.splits = p.splits;
            if (a.tickets_zeroed) b.ws = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(a.ws) + MM_TICKET_BYTES);   // the ticket counters stay zero
            for (int i = 0; i < 4; ++i) b.split_first[i] = p.split_first[i];
            hipError_t e = w4 ? launch_tile(g128::mx_gemm256_kernel<true, true>, done[4], g128::Lds<true>::TOTAL, p.tiles128 * p.splits, g128::NT, b, stream)
                              : launch_tile(g128::mx_gemm256_kernel<false, true>, done[5], g128::Lds<false>::TOTAL, p.tiles128 * p.splits, g128::NT, b, stream);
            if (e != hipSuccess) return e;
            const int total = p.tiles128 * SPLIT_WG_FLOATS;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, stream, b, p.tn, total);
            return hipGetLastError();
        }
        case TK_SMALL_SPLIT: {
            GemmArgs b = a;
            b.splits = p.small.splits;
            b.tickets = reinterpret_cast<unsigned *>(a.ws);
            b.ws = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(a.ws) + small_ticket_bytes(p.small.tiles));
            {
                // equal slab ranges; a cut inside the fp4 segment moves down to an even unit (256-deep slabs stay whole lines)
                // when that leaves the range before it non-empty
                const int n0s = a.K[0] >> 7, total = n0s + (a.K[1] >> 7) + (a.K[2] >> 7), S = p.small.splits;
                for (int q = 0; q <= S; ++q) {
                    int c = (int)((long long)total * q / S);
                    if (c < n0s && (c & 1) && q > 0 && q < S && c - 1 > b.split_cut[q - 1]) c -= 1;
                    b.split_cut[q] = (unsigned short)c;
                }
                const int off[4] = {0, n0s, n0s + (a.K[1] >> 7), total};
                int slot = 0;
                b.slot_seg = 0;
                for (int q = 0; q < S; ++q) {
                    b.split_slot[q] = (unsigned short)slot;
                    for (int sg = 0; sg < 3; ++sg) {
                        const int lo = b.split_cut[q] > off[sg] ? b.split_cut[q] : off[sg], hi = b.split_cut[q + 1] < off[sg + 1] ? b.split_cut[q + 1] : off[sg + 1];
                        if (hi > lo) b.slot_seg |= (unsigned long long)sg << (2 * slot++);
                    }
                }
                b.split_slot[S] = (unsigned short)slot;
            }
            return launch_small_tile(p.small.kind, w4, true, p.small.tiles * p.small.splits, b, stream);
        }
        case TK_G64: return launch_small_tile(64, w4, false, p.tiles64, a, stream);
        case TK_G32: return launch_small_tile(32, w4, false, p.tiles32, a, stream);
        case TK_G32N: return launch_small_tile(33, w4, false, p.tiles32n, a, stream);
        case TK_G16: return launch_small_tile(16, w4, false, p.tiles16, a, stream);
        case TK_G256_TAIL: {
            const int c = p.tail_cols;
            GemmArgs lo = a, hi = a;
            lo.n_tile0 = 0;
            lo.n_tiles = p.tn - c;
            hi.n_tile0 = p.tn - c;
            hi.n_tiles = c;
            hipError_t e = w4 ? launch_tile(g256::mx_gemm256_kernel<true, false>, done[0], g256::Lds<true>::TOTAL, p.tm256 * (p.tn - c), g256::NT, lo, stream)
                           : launch_tile(g256::mx_gemm256_kernel<false, false>, done[1], g256::Lds<false>::TOTAL, p.tm256 * (p.tn - c), g256::NT, lo, stream);
            if (e != hipSuccess) return e;
            return w4 ? launch_tile(g128::mx_gemm256_kernel<true, false>, done[2], g128::Lds<true>::TOTAL, p.tm128 * c, g128::NT, hi, stream)
                      : launch_tile(g128::mx_gemm256_kernel<false, false>, done[3], g128::Lds<false>::TOTAL, p.tm128 * c, g128::NT, hi, stream);
        }
        case TK_G256:
            if (w4) return launch_tile(g256::mx_gemm256_kernel<true, false>, done[0], g256::Lds<true>::TOTAL, p.tiles256, g256::NT, a, stream);
            return launch_tile(g256::mx_gemm256_kernel<false, false>, done[1], g256::Lds<false>::TOTAL, p.tiles256, g256::NT, a, stream);
        default:
            if (w4) return launch_tile(g128::mx_gemm256_kernel<true, false>, done[2], g128::Lds<true>::TOTAL, p.tiles128, g128::NT, a, stream);
            return launch_tile(g128::mx_gemm256_kernel<false, false>, done[3], g128::Lds<false>::TOTAL, p.tiles128, g128::NT, a, stream);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tiled GEMM with the fused gate / up epilogue (write_tile_act in mx_gemm_tile.inc; mm_gate_up_activate): 256-feature tiles only
// (one tile = 128 gate + the 128 up features of the same indices), i.e. the 256 x 256 and 128 x 256 kernels; no split-K.  M <= 64
// stays with the weight-streaming kernels + the stand-alone activation quantizer (capi.hip).
// ---------------------------------------------------------------------------------------------------------
bool mx_gemm_act_supported(int M, int N) { return M > 64 && N > 0 && (N % 256) == 0; }

struct ActPlan { bool use128; int tm256, tm128, tn, tail_cols; };
static ActPlan plan_act(int M, int N) {
    ActPlan p{};
    p.tn = N / 256;
    p.tm256 = (M + 255) / 256;
    p.tm128 = (M + 127) / 128;
    const int cus = device_cus();
    p.use128 = p.tm128 * p.tn <= cus;          // as plan_tiles: 128-row tiles while they fit one round of workgroups
    const int tiles256 = p.tm256 * p.tn, rem = tiles256 % cus;
    // tail balancing as plan_tiles: the last, less-than-half round of 256-row tiles runs as 128-row tiles
    if (!p.use128 && tiles256 > cus && rem > 0 && 2 * rem <= cus && rem % p.tm256 == 0) p.tail_cols = rem / p.tm256;
    return p;
}

const char *describe_mx_gemm_act(int M, int N) {
    static thread_local char buf[192];
    if (!mx_gemm_act_supported(M, N)) return "weight-streaming GEMM + mm::direct_quantize_kernel<0,true> (M <= 64)";
    const ActPlan p = plan_act(M, N);
    if (p.use128) snprintf(buf, sizeof(buf), "mm::g128::mx_gemm256_act_kernel x %d workgroups (128x256 tiles, fused silu*up + quantize)", p.tm128 * p.tn);
    else if (p.tail_cols) snprintf(buf, sizeof(buf), "mm::g256::mx_gemm256_act_kernel x %d workgroups (256x256 tiles) + mm::g128::mx_gemm256_act_kernel x %d", p.tm256 * (p.tn - p.tail_cols),
                                   p.tm128 * p.tail_cols);
    else snprintf(buf, sizeof(buf), "mm::g256::mx_gemm256_act_kernel x %d workgroups (256x256 tiles, fused silu*up + quantize)", p.tm256 * p.tn);
    return buf;
}

hipError_t launch_mx_gemm_act(const GemmArgs &a, hipStream_t stream) {
    static DynamicLdsOnce done[2];
    if (!mx_gemm_act_supported(a.M, a.N)) return hipErrorInvalidValue;
    const ActPlan p = plan_act(a.M, a.N);
    if (p.use128) return launch_tile(g128::mx_gemm256_act_kernel, done[1], g128::Lds<true>::TOTAL, p.tm128 * p.tn, g128::NT, a, stream);
    if (p.tail_cols) {
        GemmArgs lo = a, hi = a;
        lo.n_tile0 = 0;
        lo.n_tiles = p.tn - p.tail_cols;
        hi.n_tile0 = p.tn - p.tail_cols;
        hi.n_tiles = p.tail_cols;
        hipError_t e = launch_tile(g256::mx_gemm256_act_kernel, done[0], g256::Lds<true>::TOTAL, p.tm256 * lo.n_tiles, g256::NT, lo, stream);
        if (e != hipSuccess) return e;
        return launch_tile(g128::mx_gemm256_act_kernel, done[1], g128::Lds<true>::TOTAL, p.tm128 * hi.n_tiles, g128::NT, hi, stream);
    }
    return launch_tile(g256::mx_gemm256_act_kernel, done[0], g256::Lds<true>::TOTAL, p.tm256 * p.tn, g256::NT, a, stream);
}

// grouped launch of the tiled kernels: the tile size is chosen for the sum of the groups' tiles with the same round rule as a
// single problem (no split-K, no tail balancing)
hipError_t launch_mx_gemm256_grouped(GroupedTileArgs &ga, bool w4, hipStream_t stream) {
    if (ga.ngroups < 1 || ga.ngroups > MM_MAX_GROUPS) return hipErrorInvalidValue;
    const int cus = device_cus();
    const int N = ga.g[0].N;
    auto tiles = [&](int bm, int bn) {
        int t = 0;
        for (int i = 0; i < ga.ngroups; ++i) t += ((ga.g[i].M + bm - 1) / bm) * ((N + bn - 1) / bn);
        return t;
    };
    const int t128 = tiles(128, 256), t64 = tiles(128, 128), t32 = tiles(64, 128), t32n = tiles(64, 64);
    int bm, bn;
    if (t32n <= cus) { bm = 64; bn = 64; }                         // the 4-wave tiles, as for a single problem (plan_tiles)
    else if (t32 <= cus && 2 * t64 <= cus) { bm = 64; bn = 128; }
    else if (2 * t128 <= cus && t64 <= cus) { bm = 128; bn = 128; }
    else if (t128 <= cus) { bm = 128; bn = 256; }
    else { bm = 256; bn = 256; }
    ga.first_block[0] = 0;
    for (int i = 0; i < ga.ngroups; ++i)
        ga.first_block[i + 1] = ga.first_block[i] + ((ga.g[i].M + bm - 1) / bm) * ((N + bn - 1) / bn);
    for (int i = ga.ngroups + 1; i <= MM_MAX_GROUPS; ++i) ga.first_block[i] = ga.first_block[ga.ngroups];
    const int total = ga.first_block[ga.ngroups];
    static DynamicLdsOnce done[10];
    auto go = [&](auto kern, DynamicLdsOnce &d, int lds, int threads) -> hipError_t {
        if (hipError_t e = d.ensure(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(total), dim3(threads), lds, stream, ga);
        return hipGetLastError();
    };
    if (bm == 64) return launch_small_tile_grouped(bn == 64 ? 33 : 32, w4, total, ga, stream);
    if (bm == 256) return w4 ? go(g256::mx_gemm256_grouped_kernel<true>, done[0], g256::Lds<true>::TOTAL, g256::NT)
                             : go(g256::mx_gemm256_grouped_kernel<false>, done[1], g256::Lds<false>::TOTAL, g256::NT);
    if (bn == 256) return w4 ? go(g128::mx_gemm256_grouped_kernel<true>, done[2], g128::Lds<true>::TOTAL, g128::NT)
                             : go(g128::mx_gemm256_grouped_kernel<false>, done[3], g128::Lds<false>::TOTAL, g128::NT);
    return launch_small_tile_grouped(64, w4, total, ga, stream);
}

}  // namespace mm
