/*
 * venc_device.h -- per-lane building blocks of the HIP macroblock path.
 *
 * Everything here is pure per-lane integer arithmetic (no cross-lane traffic,
 * no memory beyond the lane's own bit sink), so the same source is compiled
 *   - by hipcc for gfx950 into the kernels of venc_kernels.hip, and
 *   - by g++ into the host-check library used by the "not gpu" tests
 *     (tests/test_device_functions_cpu.py), which is how the per-lane logic is
 *     verified in a container that has no GPU.  The host-check build is a test
 *     artefact; the product never runs it (the product path fails loudly when
 *     the HIP library is missing).
 *
 * Stages restated (reference: everything behind x264_encoder_encode(),
 * av_encode.c:970; no reference source exists):
 *   a4.2 forward 4x4 integer core transform ("DCT")
 *   a4.3 scalar quantisation at fixed QP (+ Hadamard DC paths)
 *   a4.4 zig-zag + run/level = CAVLC residual block
 *   a4.5 dequantisation + inverse transform (reconstruction)
 */
#ifndef VENC_DEVICE_H
#define VENC_DEVICE_H
#include <stdint.h>

#include "venc_tables.h"

#if defined(__HIPCC__)
#define VENC_DEV __device__ __forceinline__
/* 24-bit multiplies (v_mul_u32_u24 / v_mul_i32_i24, v_mad_*: full rate; v_mul_lo_u32 issues at a quarter of it).  Every
 * use below has both factors inside 24 bits and the product inside 32, see the range note at each call. */
#define VENC_UMUL24(a, b) __umul24((unsigned)(a), (unsigned)(b))
#define VENC_MUL24(a, b) __mul24((int)(a), (int)(b))
#else
#define VENC_DEV static inline
#define VENC_UMUL24(a, b) ((unsigned)(a) * (unsigned)(b))
#define VENC_MUL24(a, b) ((int)(a) * (int)(b))
#endif

/* quantiser constants of one QP, computed on the host once per encoder */
typedef struct {
    int per;   /* qp / 6 */
    int qbits; /* 15 + qp/6 */
    int mf[3]; /* MF[qp%6][class] */
    int v[3];  /* V[qp%6][class] */
    int f_intra, f_inter;
    int zt_inter[3]; /* per class: the largest |w| that quantises to level 0 with the inter offset: (|w| mf + f_inter) >> qbits == 0 */
} venc_qparams;

/* zig-zag scan index -> raster index as a foldable expression (keeps unrolled
 * loops free of table loads so coefficient arrays stay in registers) */
VENC_DEV int venc_zigzag_ct(int i) { return (int)((0xFEB7ADC963258410ull >> (4 * i)) & 15ull); }

VENC_DEV int venc_pos_class(int r) {
    int row = r >> 2, col = r & 3;
    return (!(row & 1) && !(col & 1)) ? 0 : ((row & 1) && (col & 1)) ? 1 : 2;
}

/* ---------------------------------------------------------------- transforms */
VENC_DEV void venc_fdct4(const int x[16], int w[16]) {
    int t[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int s0 = x[4 * i] + x[4 * i + 3], s1 = x[4 * i + 1] + x[4 * i + 2];
        int d0 = x[4 * i] - x[4 * i + 3], d1 = x[4 * i + 1] - x[4 * i + 2];
        t[4 * i] = s0 + s1;
        t[4 * i + 1] = 2 * d0 + d1;
        t[4 * i + 2] = s0 - s1;
        t[4 * i + 3] = d0 - 2 * d1;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int s0 = t[j] + t[12 + j], s1 = t[4 + j] + t[8 + j];
        int d0 = t[j] - t[12 + j], d1 = t[4 + j] - t[8 + j];
        w[j] = s0 + s1;
        w[4 + j] = 2 * d0 + d1;
        w[8 + j] = s0 - s1;
        w[12 + j] = d0 - 2 * d1;
    }
}
/* inverse core transform incl. the final (x + 32) >> 6; rows first (8.5.12.2) */
VENC_DEV void venc_idct4(const int d[16], int r[16]) {
    int f[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[4 * i] + d[4 * i + 2], e1 = d[4 * i] - d[4 * i + 2];
        int e2 = (d[4 * i + 1] >> 1) - d[4 * i + 3], e3 = d[4 * i + 1] + (d[4 * i + 3] >> 1);
        f[4 * i] = e0 + e3;
        f[4 * i + 1] = e1 + e2;
        f[4 * i + 2] = e1 - e2;
        f[4 * i + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = f[j] + f[8 + j], g1 = f[j] - f[8 + j];
        int g2 = (f[4 + j] >> 1) - f[12 + j], g3 = f[4 + j] + (f[12 + j] >> 1);
        r[j] = (g0 + g3 + 32) >> 6;
        r[4 + j] = (g1 + g2 + 32) >> 6;
        r[8 + j] = (g1 - g2 + 32) >> 6;
        r[12 + j] = (g0 - g3 + 32) >> 6;
    }
}

/* ---------------------------------------------------------------- quantiser */
VENC_DEV int venc_quant_mag(int w, int mf, int f, int shift) {
    unsigned a = (unsigned)(w < 0 ? -w : w);
    /* |w| <= 65280 (Hadamard of 16 luma DCs of 4080), mf <= 13107: product < 2^30 */
    int lv = (int)((VENC_UMUL24(a, mf) + (unsigned)f) >> shift);
    return w < 0 ? -lv : lv;
}
/* AC / normal coefficient at raster position r */
VENC_DEV int venc_quant(int w, const venc_qparams *q, int r, int intra) {
    return venc_quant_mag(w, q->mf[venc_pos_class(r)], intra ? q->f_intra : q->f_inter, q->qbits);
}
VENC_DEV int venc_dequant(int lv, const venc_qparams *q, int r) { return VENC_MUL24(lv, q->v[venc_pos_class(r)]) << q->per; } /* |lv| < 2^16, v <= 29 */
/* Intra16x16 luma DC on the un-halved 4x4 Hadamard output */
VENC_DEV int venc_quant_luma_dc(int y, const venc_qparams *q) {
    return venc_quant_mag(y, q->mf[0], 4 * q->f_intra, q->qbits + 2);
}
VENC_DEV int venc_dequant_luma_dc(int f, const venc_qparams *q) {
    int ls = 16 * q->v[0];
    /* |f| < 2^20 (Hadamard of 16 levels), ls <= 464 */
    if (q->per >= 6) return VENC_MUL24(f, ls) << (q->per - 6);
    return (VENC_MUL24(f, ls) + (1 << (5 - q->per))) >> (6 - q->per);
}
VENC_DEV int venc_quant_chroma_dc(int y, const venc_qparams *q, int intra) {
    return venc_quant_mag(y, q->mf[0], 2 * (intra ? q->f_intra : q->f_inter), q->qbits + 1);
}
VENC_DEV int venc_dequant_chroma_dc(int f, const venc_qparams *q) { return (VENC_MUL24(f, 16 * q->v[0]) << q->per) >> 5; } /* |f| < 2^18 */

VENC_DEV int venc_ilog2(uint32_t v);
/* ---------------------------------------------------------------- rate control (spec/CODEC.md section 1a)
 * Picture QP under X264_RC_CRF: IDR base - 3; first P picture of a GOP base; other P pictures base + delta from the mean
 * best SAD per macroblock of the previous picture, 2.4 QP per octave around a mean SAD of 512.  Integer, identical on
 * every lane / block that evaluates it. */
VENC_DEV int venc_rc_ilog2q(uint32_t c) { /* quarter octaves: 4 floor(log2 c) + the two bits below the leading one */
    if (c < 1u) c = 1u;
    const int k = venc_ilog2(c);
    const int frac = k >= 2 ? (int)((c >> (k - 2)) & 3u) : (int)((c << (2 - k)) & 3u);
    return 4 * k + frac;
}
VENC_DEV int venc_rc_qp(int base_qp, int is_idr, int have_history, unsigned long long prev_sad_sum, int nmb) {
    int qp = base_qp;
    if (is_idr)
        qp = base_qp - 3;
    else if (have_history) {
        const uint32_t c = (uint32_t)(prev_sad_sum / (unsigned long long)(nmb > 0 ? nmb : 1));
        const int num = 3 * (venc_rc_ilog2q(c) - 36);
        int delta = num >= 0 ? num / 5 : -((-num + 4) / 5); /* floor division */
        delta = delta < -6 ? -6 : delta > 6 ? 6 : delta;
        qp = base_qp + delta;
    }
    return qp < 12 ? 12 : qp > 51 ? 51 : qp;
}
/* quantiser constants of a QP from the tables (what the host computes once per encoder at fixed QP) */
VENC_DEV void venc_make_qparams(int qp, venc_qparams *q) {
    const int per = qp / 6, rem = qp - 6 * per;
    q->per = per;
    q->qbits = 15 + per;
    for (int k = 0; k < 3; k++) {
        q->mf[k] = venc_mf[rem * 3 + k];
        q->v[k] = venc_v[rem * 3 + k];
    }
    q->f_intra = (1 << q->qbits) / 3;
    q->f_inter = (1 << q->qbits) / 6;
    for (int k = 0; k < 3; k++) q->zt_inter[k] = ((1 << q->qbits) - q->f_inter - 1) / q->mf[k];
}
VENC_DEV int venc_chroma_qp(int qp) { return qp < 30 ? qp : venc_qpc_hi[qp - 30]; }

/* ---------------------------------------------------------------- bit sink */
/* MSB-first 32-bit words into lane-private memory */
typedef struct {
    uint32_t *w;
    uint64_t acc;
    int nacc;   /* valid low bits in acc, < 32 between calls */
    int nwords; /* complete words stored */
} venc_bitsink;

VENC_DEV void venc_bs_init(venc_bitsink *b, uint32_t *w) {
    b->w = w;
    b->acc = 0;
    b->nacc = 0;
    b->nwords = 0;
}
VENC_DEV void venc_bs_put(venc_bitsink *b, uint32_t code, int len) { /* 0 <= len <= 32, code < 2^len */
    b->acc = (b->acc << len) | (uint64_t)code;
    b->nacc += len;
    if (b->nacc >= 32) {
        b->w[b->nwords++] = (uint32_t)(b->acc >> (b->nacc - 32));
        b->nacc -= 32;
    }
}
VENC_DEV int venc_bs_bits(const venc_bitsink *b) { return 32 * b->nwords + b->nacc; }
/* store the partial last word left-aligned; returns total bits */
VENC_DEV int venc_bs_flush(venc_bitsink *b) {
    if (b->nacc > 0) b->w[b->nwords] = (uint32_t)(b->acc << (32 - b->nacc));
    return venc_bs_bits(b);
}
/* The same sink with its first four words in REGISTERS (round 4: the syntax entries of P pictures, venc_cavlc_p.hip).  An entry
 * -- a macroblock header or one 4x4 block -- is 10-30 bits on ordinary content, so its whole string leaves the lane as ONE
 * 16-byte store into a dense array (coalesced across the wave) instead of a word here and there in a 2 KB region per macroblock
 * (what made k_pack read 65x the bytes it wrote).  Words 4 .. 15 (an entry beyond 128 bits: a dense block at a fine quantiser) go
 * to spill[word index].  h0 .. h3 are only ever addressed by constants (selects), so they stay in registers. */
typedef struct {
    uint32_t h0, h1, h2, h3;
    uint32_t *spill;
    uint64_t acc;
    int nacc, nwords;
} venc_bitsink4;
VENC_DEV void venc_bs4_init(venc_bitsink4 *b, uint32_t *spill) {
    b->h0 = b->h1 = b->h2 = b->h3 = 0u;
    b->spill = spill;
    b->acc = 0;
    b->nacc = 0;
    b->nwords = 0;
}
VENC_DEV void venc_bs4_word(venc_bitsink4 *b, uint32_t w) {
    const int n = b->nwords;
    if (n < 4) {
        b->h0 = n == 0 ? w : b->h0;
        b->h1 = n == 1 ? w : b->h1;
        b->h2 = n == 2 ? w : b->h2;
        b->h3 = n == 3 ? w : b->h3;
    } else
        b->spill[n] = w;
    b->nwords = n + 1;
}
VENC_DEV void venc_bs_put(venc_bitsink4 *b, uint32_t code, int len) { /* as above */
    b->acc = (b->acc << len) | (uint64_t)code;
    b->nacc += len;
    if (b->nacc >= 32) {
        venc_bs4_word(b, (uint32_t)(b->acc >> (b->nacc - 32)));
        b->nacc -= 32;
    }
}
VENC_DEV int venc_bs_flush(venc_bitsink4 *b) { /* the partial last word left-aligned (nwords does not count it); returns total bits */
    const int bits = 32 * b->nwords + b->nacc;
    if (b->nacc > 0) {
        venc_bs4_word(b, (uint32_t)(b->acc << (32 - b->nacc)));
        b->nwords--;
    }
    return bits;
}
VENC_DEV int venc_ilog2(uint32_t v) { /* floor(log2(v)), v > 0 */
#if defined(__HIPCC__)
    return 31 - __clz((int)v);
#else
    return 31 - __builtin_clz(v);
#endif
}
VENC_DEV int venc_popc(uint32_t v) {
#if defined(__HIPCC__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}
template <class BS>
VENC_DEV void venc_bs_ue(BS *b, uint32_t v) {
    int k = venc_ilog2(v + 1);
    venc_bs_put(b, v + 1, 2 * k + 1); /* k zeros followed by the (k+1)-bit value */
}
template <class BS>
VENC_DEV void venc_bs_se(BS *b, int v) { venc_bs_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }

/* ---------------------------------------------------------------- CAVLC block
 * c[]: the block's levels in scan order; the block occupies scan positions
 * first .. first+max_coef-1 of c[] (first = 1, max_coef = 15 for AC-only blocks).
 * nC < 0 selects the chroma-DC tables.  All loops over c[] are fully unrolled so
 * that c[] stays in registers.  Returns TotalCoeff. */
VENC_DEV int venc_cavlc_block(venc_bitsink *b, const int c[16], int first, int max_coef, int nC) {
    uint32_t nz = 0, ones = 0, neg = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int k = i - first;
        if (k >= 0 && k < max_coef) {
            int v = c[i];
            nz |= (uint32_t)(v != 0) << k;
            ones |= (uint32_t)(v == 1 || v == -1) << k;
            neg |= (uint32_t)(v < 0) << k;
        }
    }
    int total = venc_popc(nz);
    /* trailing ones: up to three +-1 at the high-frequency end */
    int t1 = 0;
    uint32_t m = nz;
    for (int k = 0; k < 3 && m; k++) {
        int i = venc_ilog2(m);
        if (!((ones >> i) & 1)) break;
        t1++;
        m ^= 1u << i;
    }
    uint32_t tok;
    if (nC < 0)
        tok = venc_vlc_coeff_token_cdc[4 * total + t1];
    else
        tok = venc_vlc_coeff_token[(nC < 2 ? 0 : nC < 4 ? 68 : nC < 8 ? 136 : 204) + 4 * total + t1];
    venc_bs_put(b, tok & 0xff, (int)(tok >> 8));
    if (total == 0) return 0;
    {
        uint32_t mm = nz;
        for (int k = 0; k < t1; k++) {
            int i = venc_ilog2(mm);
            venc_bs_put(b, (neg >> i) & 1, 1);
            mm ^= 1u << i;
        }
    }
    /* remaining levels, high frequency first (m has the trailing ones removed) */
    int sl = (total > 10 && t1 < 3) ? 1 : 0;
    int first_level = 1;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        int k = i - first;
        if (k >= 0 && k < max_coef && ((m >> k) & 1)) {
            int lv = c[i];
            int alv = lv < 0 ? -lv : lv;
            int code = 2 * alv - 2 + (lv < 0);
            if (first_level && t1 < 3) code -= 2;
            first_level = 0;
            if (sl == 0) {
                if (code < 14)
                    venc_bs_put(b, 1, code + 1);
                else if (code < 30)
                    venc_bs_put(b, 16u | (uint32_t)(code - 14), 19);
                else
                    venc_bs_put(b, 4096u | (uint32_t)(code - 30), 28);
            } else {
                if (code < (15 << sl))
                    venc_bs_put(b, (1u << sl) | (uint32_t)(code & ((1 << sl) - 1)), (code >> sl) + 1 + sl);
                else
                    venc_bs_put(b, 4096u | (uint32_t)(code - (15 << sl)), 28);
            }
            if (sl == 0) sl = 1;
            if (alv > (3 << (sl - 1)) && sl < 6) sl++;
        }
    }
    int last = venc_ilog2(nz);
    int total_zeros = last + 1 - total;
    if (total < max_coef) {
        uint32_t tz = (max_coef == 4) ? venc_vlc_total_zeros_cdc[4 * (total - 1) + total_zeros]
                                      : venc_vlc_total_zeros[16 * (total - 1) + total_zeros];
        venc_bs_put(b, tz & 0xff, (int)(tz >> 8));
    }
    /* run_before for every coefficient but the lowest, while zeros remain */
    int zeros_left = total_zeros;
    uint32_t rest = nz;
    while (zeros_left > 0) {
        int i = venc_ilog2(rest);
        rest ^= 1u << i;
        if (!rest) break;
        int run = i - venc_ilog2(rest) - 1;
        uint32_t rb = venc_vlc_run_before[16 * ((zeros_left > 7 ? 7 : zeros_left) - 1) + run];
        venc_bs_put(b, rb & 0xff, (int)(rb >> 8));
        zeros_left -= run;
    }
    return total;
}

/* ---------------------------------------------------------------- CAVLC block, data-driven
 * Same syntax as venc_cavlc_block, but the levels sit in (lane-private) memory and every loop
 * walks the set bits of the non-zero mask instead of all 16 positions: the work is proportional
 * to the number of coefficients (1-4 in a typical P block).  tab = the five VLC tables back to
 * back (see VENC_TAB_* offsets), staged in LDS by the kernel.
 *   c   : 16 levels, scan order; the block occupies c[first .. first+max_coef-1]
 *   nz  : bit k set iff c[first+k] != 0 */
#define VENC_TAB_TOKEN 0
#define VENC_TAB_TOKEN_CDC 272
#define VENC_TAB_TZ 292
#define VENC_TAB_TZ_CDC 532
#define VENC_TAB_RUN 544
#define VENC_TAB_SIZE 656

VENC_DEV uint32_t venc_nzmask_mem(const int16_t *c, int first, int max_coef) {
    uint32_t nz = 0;
    for (int k = 0; k < max_coef; k++) nz |= (uint32_t)(c[first + k] != 0) << k;
    return nz;
}

template <class BS>
VENC_DEV int venc_cavlc_block_mem(BS *b, const int16_t *c, uint32_t nz, int first, int max_coef, int nC,
                                  const uint16_t *tab) {
    const int total = venc_popc(nz);
    int t1 = 0;
    uint32_t m = nz, signs = 0;
    while (t1 < 3 && m) {
        const int i = venc_ilog2(m);
        const int v = c[first + i];
        if (v != 1 && v != -1) break;
        signs = (signs << 1) | (uint32_t)(v < 0);
        t1++;
        m ^= 1u << i;
    }
    const uint32_t tok = nC < 0 ? tab[VENC_TAB_TOKEN_CDC + 4 * total + t1]
                                : tab[VENC_TAB_TOKEN + (nC < 2 ? 0 : nC < 4 ? 68 : nC < 8 ? 136 : 204) + 4 * total + t1];
    venc_bs_put(b, tok & 0xff, (int)(tok >> 8));
    if (total == 0) return 0;
    venc_bs_put(b, signs, t1);
    int sl = (total > 10 && t1 < 3) ? 1 : 0;
    int first_level = 1;
    while (m) {
        const int i = venc_ilog2(m);
        m ^= 1u << i;
        const int lv = c[first + i];
        const int alv = lv < 0 ? -lv : lv;
        int code = 2 * alv - 2 + (lv < 0);
        if (first_level && t1 < 3) code -= 2;
        first_level = 0;
        if (sl == 0) {
            if (code < 14)
                venc_bs_put(b, 1, code + 1);
            else if (code < 30)
                venc_bs_put(b, 16u | (uint32_t)(code - 14), 19);
            else
                venc_bs_put(b, 4096u | (uint32_t)(code - 30), 28);
        } else {
            if (code < (15 << sl))
                venc_bs_put(b, (1u << sl) | (uint32_t)(code & ((1 << sl) - 1)), (code >> sl) + 1 + sl);
            else
                venc_bs_put(b, 4096u | (uint32_t)(code - (15 << sl)), 28);
        }
        if (sl == 0) sl = 1;
        if (alv > (3 << (sl - 1)) && sl < 6) sl++;
    }
    const int total_zeros = venc_ilog2(nz) + 1 - total;
    if (total < max_coef) {
        const uint32_t tz = (max_coef == 4) ? tab[VENC_TAB_TZ_CDC + 4 * (total - 1) + total_zeros]
                                            : tab[VENC_TAB_TZ + 16 * (total - 1) + total_zeros];
        venc_bs_put(b, tz & 0xff, (int)(tz >> 8));
    }
    int zeros_left = total_zeros;
    uint32_t rest = nz;
    while (zeros_left > 0) {
        const int i = venc_ilog2(rest);
        rest ^= 1u << i;
        if (!rest) break;
        const int run = i - venc_ilog2(rest) - 1;
        const uint32_t rb = tab[VENC_TAB_RUN + 16 * ((zeros_left > 7 ? 7 : zeros_left) - 1) + run];
        venc_bs_put(b, rb & 0xff, (int)(rb >> 8));
        zeros_left -= run;
    }
    return total;
}

/* nC from the neighbouring blocks' TotalCoeff (9.2.1) */
VENC_DEV int venc_nc(int availA, int nA, int availB, int nB) {
    if (availA && availB) return (nA + nB + 1) >> 1;
    return availA ? nA : availB ? nB : 0;
}
VENC_DEV int venc_median3(int a, int b, int c) {
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return c < mn ? mn : c > mx ? mx : c;
}
VENC_DEV int venc_clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

#endif
