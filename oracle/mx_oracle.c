/*
 * CPU oracle (C restatement) for the MicroMix mgemm hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Second, independent restatement of the reference algorithm.  Where
 * oracle/mx_oracle.py uses integer-exact forms, this file follows the
 * reference's float formulas literally (log2f/ceilf/ldexpf, bf16 rounding of the
 * scale, reciprocal multiply, clamp, bf16 rounding, RNE conversion), so that the
 * two can be checked against each other.  Nothing under micromix_amd/ links or
 * loads this file.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and cannot
 * be compiled here (needs nvcc + CUTLASS; /root/reference/cutlass is empty).
 *
 * Reference lines followed (relative to /root/reference):
 *   mgemm/src/reorder.cu:17-19      FP4_MAX 6, FP6_MAX 28, FP8_MAX 448
 *   mgemm/src/reorder.cu:30-33      PackFp4 {low, high}
 *   mgemm/src/reorder.cu:54-63      pack_4_fp6_to_3_bytes
 *   mgemm/src/reorder.cu:94-269     reorder_quantize_mixed_kernel
 *   mgemm/src/reorder.cu:271-432    reorder_quantize_mxfp4_kernel
 *   mgemm/include/sm120_sf_layout.h:170-173   scale-factor atom / tiling
 *   mgemm/src/gemm.cu:26-78         three-segment dispatcher (beta = 0,1,1)
 *   mgemm/src/w4a4.cu:27,176        fp32 accumulator, alpha=1 beta=0
 *   mgemm/src/w4a6.cu:178           alpha=1 beta=1
 *   mgemm/src/activate.cu:29,101    silu(x) = x / (1 + expf(-x)); v = silu(float(a)) * float(b)
 *   mgemm/src/activate.cu:104-160   per 32 values: amax, scale = amax > 1e-6 ? 2^ceil(log2(amax/FMAX)) : 1, q = RNE(clamp(v/scale))
 *   mgemm/src/activate.cu:208-500   the same on float(w) for the down_proj weights (mixed and all-fp4)
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { FMT_FP4 = 0, FMT_FP6 = 1, FMT_FP8 = 2 };

static const float FMAXV[3] = {6.0f, 28.0f, 448.0f};
static const int EBITS[3] = {2, 3, 4};
static const int MBITS[3] = {1, 2, 3};
static const int BIAS[3] = {1, 3, 7};
static const int MAXCODE[3] = {0x7, 0x1F, 0x7E};

static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

/* value of a magnitude code, by the format definition */
static float decode_mag(int code, int fmt) {
    int mb = MBITS[fmt], bias = BIAS[fmt];
    int e = code >> mb, m = code & ((1 << mb) - 1);
    if (e == 0) return ldexpf((float)m, 1 - bias - mb);
    return ldexpf(1.0f + (float)m / (float)(1 << mb), e - bias);
}

float mxo_decode(int code, int fmt) {
    int sb = EBITS[fmt] + MBITS[fmt];
    float v = decode_mag(code & ((1 << sb) - 1), fmt);
    return (code >> sb) & 1 ? -v : v;
}

/* Definitional encoder: nearest representable magnitude, ties to the even code,
 * saturating at the largest finite value, sign bit kept (also on zero). */
int mxo_encode_search(float x, int fmt) {
    uint32_t u;
    memcpy(&u, &x, 4);
    int sign = (int)(u >> 31);
    float a = fabsf(x);
    int best = 0;
    double bestd = INFINITY;
    for (int c = 0; c <= MAXCODE[fmt]; ++c) {
        double d = fabs((double)decode_mag(c, fmt) - (double)a);
        if (d < bestd || (d == bestd && (c & 1) == 0)) {
            bestd = d;
            best = c;
        }
    }
    if (!(a <= FMAXV[fmt])) best = MAXCODE[fmt]; /* satfinite, incl. inf/nan */
    return best | (sign << (EBITS[fmt] + MBITS[fmt]));
}

/* Bit-twiddling encoder (used on the bulk path; tests check it == search). */
int mxo_encode_fast(float x, int fmt) {
    int mb = MBITS[fmt], bias = BIAS[fmt];
    uint32_t u;
    memcpy(&u, &x, 4);
    int sign = (int)(u >> 31);
    uint32_t a = u & 0x7FFFFFFFu;
    int emin = 1 - bias;
    int64_t code;
    if (a < ((uint32_t)(127 + emin) << 23)) {
        float af;
        memcpy(&af, &a, 4);
        code = (int64_t)nearbyint((double)af * ldexp(1.0, mb - emin)); /* RNE */
    } else {
        int shift = 23 - mb;
        uint64_t r = (uint64_t)a + ((1u << (shift - 1)) - 1) + ((a >> shift) & 1u);
        code = (int64_t)(r >> shift) - ((int64_t)(127 - bias) << mb);
    }
    if (code > MAXCODE[fmt]) code = MAXCODE[fmt];
    return (int)code | (sign << (EBITS[fmt] + MBITS[fmt]));
}

/* reorder.cu:179-180 literally: scale exponent from the block absmax.
 * Returns e with scale = 2^e.  The clamp to [-127,127] is ours (the reference's
 * bf16 scale underflows to 0 below 2^-133 and its behaviour is then undefined). */
int mxo_scale_exponent_literal(float maxv, int fmt) {
    if (maxv == 0.0f) return -1; /* scale = 0.5 */
    float ratio = maxv / FMAXV[fmt];
    int e = (int)ceilf(log2f(ratio));
    if (e < -127) e = -127;
    if (e > 127) e = 127;
    return e;
}

static inline int64_t sf_offset(int64_t r, int64_t j, int64_t kseg) {
    return (r / 128) * (kseg / 128) * 512 + (j / 4) * 512 + (r % 32) * 16 + ((r / 32) % 4) * 4 + (j % 4);
}

int64_t mxo_sf_offset(int64_t r, int64_t j, int64_t kseg) { return sf_offset(r, j, kseg); }

/* One row, one 32-group: gather, absmax, scale, quantize, pack (reorder.cu:153-268). */
static void quantize_group(const uint16_t *row, const int16_t *idx32, int fmt, uint8_t *out, uint8_t *sfbyte) {
    float v[32];
    float maxv = 0.0f;
    for (int i = 0; i < 32; ++i) {
        v[i] = bf16_to_f32(row[(uint16_t)idx32[i]]);
        float a = fabsf(v[i]);
        maxv = a > maxv ? a : maxv;
    }
    int e = mxo_scale_exponent_literal(maxv, fmt);
    /* scale = converterScale(ldexpf(1, e)) -- a power of two, exact in bf16 for e >= -126 */
    double r_scale = ldexp(1.0, -e); /* 1.0 / scale, exact */
    *sfbyte = (uint8_t)(e + 127);
    uint8_t codes[32];
    float lo = -FMAXV[fmt], hi = FMAXV[fmt];
    for (int i = 0; i < 32; ++i) {
        float s = (float)((double)v[i] * r_scale);  /* exact power-of-two scaling */
        s = s < lo ? lo : (s > hi ? hi : s);        /* clamp (never bites) */
        s = bf16_to_f32(f32_to_bf16(s));            /* converterScale -> bf16 */
        codes[i] = (uint8_t)mxo_encode_fast(s, fmt);
    }
    if (fmt == FMT_FP8) {
        memcpy(out, codes, 32);
    } else if (fmt == FMT_FP6) {
        for (int i = 0; i < 32; i += 4) {
            uint8_t a = codes[i] & 0x3F, b = codes[i + 1] & 0x3F, c = codes[i + 2] & 0x3F, d = codes[i + 3] & 0x3F;
            uint8_t *o = out + (i / 4) * 3;
            o[0] = (uint8_t)(a | ((b & 0x03) << 6));
            o[1] = (uint8_t)((b >> 2) | ((c & 0x0F) << 4));
            o[2] = (uint8_t)((c >> 4) | (d << 2));
        }
    } else {
        for (int i = 0; i < 32; i += 2) out[i / 2] = (uint8_t)((codes[i] & 0xF) | ((codes[i + 1] & 0xF) << 4));
    }
}

static int group_bytes(int fmt) { return fmt == FMT_FP8 ? 32 : (fmt == FMT_FP6 ? 24 : 16); }

/* mode 0: mixed (fp4|fp6|fp8) ; mode 1: all segments fp4 (weights, "w4").
 * SF buffers must be pre-sized by the caller; only valid bytes are written. */
int mxo_reorder_quantize(const uint16_t *x, int rows, int K, const int16_t *idx, int KN, int KS, int KO, int mode,
                         uint8_t *oN, uint8_t *oS, uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO) {
    if (KN < 0 || KS < 0 || KO < 0 || KN + KS + KO != K || (KN % 128) || (KS % 128) || (KO % 128)) return 1;
    int fN = FMT_FP4, fS = mode ? FMT_FP4 : FMT_FP6, fO = mode ? FMT_FP4 : FMT_FP8;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const uint16_t *row = x + (int64_t)r * K;
        for (int g = 0; g < K / 32; ++g) {
            int fmt, j, kseg;
            uint8_t *out, *sf;
            if (g < KN / 32) {
                fmt = fN; j = g; kseg = KN;
                out = oN + (int64_t)r * (KN / 32) * group_bytes(fN);
                sf = sfN;
            } else if (g < (KN + KS) / 32) {
                fmt = fS; j = g - KN / 32; kseg = KS;
                out = oS + (int64_t)r * (KS / 32) * group_bytes(fS);
                sf = sfS;
            } else {
                fmt = fO; j = g - (KN + KS) / 32; kseg = KO;
                out = oO + (int64_t)r * (KO / 32) * group_bytes(fO);
                sf = sfO;
            }
            quantize_group(row, idx + g * 32, fmt, out + (int64_t)j * group_bytes(fmt), &sf[sf_offset(r, j, kseg)]);
        }
    }
    return 0;
}

/* dequantise one segment to fp32 [rows, kseg] (scale applied) */
static void dequant(const uint8_t *p, const uint8_t *sf, int rows, int kseg, int fmt, float *out) {
    int gb = group_bytes(fmt);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int j = 0; j < kseg / 32; ++j) {
            const uint8_t *g = p + ((int64_t)r * (kseg / 32) + j) * gb;
            float s = ldexpf(1.0f, (int)sf[sf_offset(r, j, kseg)] - 127);
            float *o = out + (int64_t)r * kseg + j * 32;
            for (int i = 0; i < 32; ++i) {
                int code;
                if (fmt == FMT_FP8) code = g[i];
                else if (fmt == FMT_FP4) code = (g[i / 2] >> ((i & 1) * 4)) & 0xF;
                else {
                    int bit = i * 6, byte = bit >> 3, sh = bit & 7;
                    int w = g[byte] | (byte + 1 < 24 ? (g[byte + 1] << 8) : 0);
                    code = (w >> sh) & 0x3F;
                }
                o[i] = mxo_decode(code, fmt) * s;
            }
        }
    }
}

/* D(bf16) = bf16(acc + beta * D) for one segment; fp32 accumulate */
static void seg_gemm(const float *a, const float *b, int M, int N, int K, uint16_t *D, int beta) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float *ar = a + (int64_t)m * K;
        for (int n = 0; n < N; ++n) {
            const float *br = b + (int64_t)n * K;
            float acc = 0.0f;
            for (int k0 = 0; k0 < K; k0 += 32) {
                float part = 0.0f;
                for (int k = k0; k < k0 + 32; ++k) part += ar[k] * br[k];
                acc += part;
            }
            float c = beta ? bf16_to_f32(D[(int64_t)m * N + n]) : 0.0f;
            D[(int64_t)m * N + n] = f32_to_bf16(acc + c);
        }
    }
}

/* wmode 0: B formats fp4|fp6|fp8 ("w"), 1: all fp4 ("w4").  gemm.cu:26-78. */
int mxo_matmul(const uint8_t *AN, const uint8_t *BN, const uint8_t *AS, const uint8_t *BS, const uint8_t *AO,
               const uint8_t *BO, const uint8_t *SFAN, const uint8_t *SFBN, const uint8_t *SFAS, const uint8_t *SFBS,
               const uint8_t *SFAO, const uint8_t *SFBO, int M, int N, int KN, int KS, int KO, int wmode,
               uint16_t *D) {
    const uint8_t *A[3] = {AN, AS, AO}, *B[3] = {BN, BS, BO}, *SFA[3] = {SFAN, SFAS, SFAO}, *SFB[3] = {SFBN, SFBS, SFBO};
    int Ks[3] = {KN, KS, KO};
    int fa[3] = {FMT_FP4, FMT_FP6, FMT_FP8};
    int fb[3] = {FMT_FP4, wmode ? FMT_FP4 : FMT_FP6, wmode ? FMT_FP4 : FMT_FP8};
    memset(D, 0, (size_t)M * N * 2); /* C = torch::zeros (bindings.cpp:72) */
    int beta = 0;                    /* the reference's first launched segment reads the zeroed C */
    for (int s = 0; s < 3; ++s) {
        if (Ks[s] == 0) continue;
        float *a = (float *)malloc((size_t)M * Ks[s] * 4), *b = (float *)malloc((size_t)N * Ks[s] * 4);
        if (!a || !b) { free(a); free(b); return 2; }
        dequant(A[s], SFA[s], M, Ks[s], fa[s], a);
        dequant(B[s], SFB[s], N, Ks[s], fb[s], b);
        seg_gemm(a, b, M, N, Ks[s], D, beta);
        beta = 1;
        free(a);
        free(b);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * Reorder-free ("direct") quantizers, activate.cu -- literal float formulas (expf, ceilf(log2f(.))), natural column order.
 * mode 0: v = silu(a) * b -> fp4 | fp6 | fp8;  mode 1: v = float(a) -> fp4 | fp6 | fp8;  mode 2: v = float(a) -> fp4 | fp4 | fp4.
 * Unlike the reorder kernels there is NO bf16 rounding of the scaled value (one rounding from fp32), and an empty block
 * (amax <= 1e-6) takes scale 1.0 (byte 127).
 * ------------------------------------------------------------------------------------------------------------ */
static void pack_codes(const uint8_t *codes, int fmt, uint8_t *out) {
    if (fmt == FMT_FP8) {
        memcpy(out, codes, 32);
    } else if (fmt == FMT_FP6) {
        for (int i = 0; i < 32; i += 4) {
            uint8_t a = codes[i] & 0x3F, b = codes[i + 1] & 0x3F, c = codes[i + 2] & 0x3F, d = codes[i + 3] & 0x3F;
            uint8_t *o = out + (i / 4) * 3;
            o[0] = (uint8_t)(a | ((b & 0x03) << 6));
            o[1] = (uint8_t)((b >> 2) | ((c & 0x0F) << 4));
            o[2] = (uint8_t)((c >> 4) | (d << 2));
        }
    } else {
        for (int i = 0; i < 32; i += 2) out[i / 2] = (uint8_t)((codes[i] & 0xF) | ((codes[i + 1] & 0xF) << 4));
    }
}

int mxo_direct_quantize(const uint16_t *a, const uint16_t *b, int rows, int KN, int KS, int KO, int mode, uint8_t *oN, uint8_t *oS,
                        uint8_t *oO, uint8_t *sfN, uint8_t *sfS, uint8_t *sfO) {
    const int K = KN + KS + KO;
    if (KN < 0 || KS < 0 || KO < 0 || K <= 0 || (KN % 128) || (KS % 128) || (KO % 128) || mode < 0 || mode > 2) return 1;
    const int fN = FMT_FP4, fS = mode == 2 ? FMT_FP4 : FMT_FP6, fO = mode == 2 ? FMT_FP4 : FMT_FP8;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        for (int g = 0; g < K / 32; ++g) {
            int fmt, j, kseg;
            uint8_t *out, *sf;
            if (g < KN / 32) { fmt = fN; j = g; kseg = KN; out = oN + (int64_t)r * (KN / 32) * group_bytes(fN); sf = sfN; }
            else if (g < (KN + KS) / 32) { fmt = fS; j = g - KN / 32; kseg = KS; out = oS + (int64_t)r * (KS / 32) * group_bytes(fS); sf = sfS; }
            else { fmt = fO; j = g - (KN + KS) / 32; kseg = KO; out = oO + (int64_t)r * (KO / 32) * group_bytes(fO); sf = sfO; }
            float v[32], amax = 0.0f;
            for (int i = 0; i < 32; ++i) {
                const float x = bf16_to_f32(a[(int64_t)r * K + g * 32 + i]);
                v[i] = mode == 0 ? (x / (1.0f + expf(-x))) * bf16_to_f32(b[(int64_t)r * K + g * 32 + i]) : x;
                const float m = fabsf(v[i]);
                amax = m > amax ? m : amax;
            }
            int e = 0;                                           /* scale 1.0 */
            if (amax > 1e-6f) {
                e = (int)ceilf(log2f(amax / FMAXV[fmt]));
                if (e < -127) e = -127;
                if (e > 127) e = 127;
            }
            const int ec = e < -126 ? -126 : e;                  /* the divisor is a normal fp32 */
            uint8_t codes[32];
            for (int i = 0; i < 32; ++i) {
                float q = (float)((double)v[i] * ldexp(1.0, -ec));
                q = q < -FMAXV[fmt] ? -FMAXV[fmt] : (q > FMAXV[fmt] ? FMAXV[fmt] : q);
                codes[i] = (uint8_t)mxo_encode_fast(q, fmt);
            }
            pack_codes(codes, fmt, out + (int64_t)j * group_bytes(fmt));
            sf[sf_offset(r, j, kseg)] = (uint8_t)(e + 127);
        }
    }
    return 0;
}
