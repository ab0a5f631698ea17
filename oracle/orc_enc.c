/*
 * orc_enc.c -- CPU oracle encoder (TEST INFRASTRUCTURE ONLY; see orc.h header).
 *
 * PARITY UNPINNED: plain-C restatement of the H.264 Constrained-Baseline subset
 * frozen in spec/CODEC.md.  It stands in for what sits behind the reference's
 * single call x264_encoder_encode() (av_encode.c:970 / :1078) -- the reference
 * has no source for this arithmetic (it is inside un-vendored libx264,
 * Makefile:8), so every stage below cites the spec section it restates and the
 * reference call site whose contract it serves.
 *
 * Stages (north_star order): ME block search -> forward "DCT" (4x4 integer core
 * transform) -> quantisation -> zig-zag / run-level (CAVLC) -> reconstruction.
 */
#include "orc.h"
#include "orc_interp.h"
#include "orc_intra4.h"
#include "orc_deblock.h"
#include "orc_tables.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ bit writer */
typedef struct {
    uint8_t *buf;
    size_t cap, pos;
    uint64_t acc;
    int nacc;
    int overflow;
    uint64_t total_bits;
} bw_t;

static void bw_init(bw_t *b, uint8_t *buf, size_t cap) {
    b->buf = buf;
    b->cap = cap;
    b->pos = 0;
    b->acc = 0;
    b->nacc = 0;
    b->overflow = 0;
    b->total_bits = 0;
}
static void bw_put(bw_t *b, uint32_t val, int n) {
    if (n <= 0) return;
    if (n < 32) val &= (1u << n) - 1u;
    b->acc = (b->acc << n) | val;
    b->nacc += n;
    b->total_bits += (uint64_t)n;
    while (b->nacc >= 8) {
        if (b->pos < b->cap)
            b->buf[b->pos++] = (uint8_t)(b->acc >> (b->nacc - 8));
        else
            b->overflow = 1;
        b->nacc -= 8;
    }
}
static void bw_ue(bw_t *b, uint32_t v) {
    uint32_t code = v + 1;
    int len = 0;
    for (uint32_t t = code; t; t >>= 1) len++;
    bw_put(b, 0, len - 1);
    bw_put(b, code, len);
}
static void bw_se(bw_t *b, int v) { bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(bw_t *b) {
    bw_put(b, 1, 1);
    if (b->nacc) bw_put(b, 0, 8 - b->nacc);
}

/* ------------------------------------------------------------------ transforms */
/* forward 4x4 core transform W = Cf X Cf^T (encoder side of 8.5.12) */
void orc_fdct4x4(const int16_t x[16], int16_t w[16]) {
    int t[16];
    for (int i = 0; i < 4; i++) {
        int a = x[4 * i], b = x[4 * i + 1], c = x[4 * i + 2], d = x[4 * i + 3];
        int s0 = a + d, s1 = b + c, d0 = a - d, d1 = b - c;
        t[4 * i + 0] = s0 + s1;
        t[4 * i + 1] = 2 * d0 + d1;
        t[4 * i + 2] = s0 - s1;
        t[4 * i + 3] = d0 - 2 * d1;
    }
    for (int j = 0; j < 4; j++) {
        int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
        int s0 = a + d, s1 = b + c, d0 = a - d, d1 = b - c;
        w[j] = (int16_t)(s0 + s1);
        w[4 + j] = (int16_t)(2 * d0 + d1);
        w[8 + j] = (int16_t)(s0 - s1);
        w[12 + j] = (int16_t)(d0 - 2 * d1);
    }
}
/* inverse 4x4 transform, 8.5.12.2: rows first, then columns, then (x+32)>>6 */
void orc_idct4x4(const int32_t d[16], int16_t r[16]) {
    int f[16];
    for (int i = 0; i < 4; i++) {
        int d0 = d[4 * i], d1 = d[4 * i + 1], d2 = d[4 * i + 2], d3 = d[4 * i + 3];
        int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
        f[4 * i + 0] = e0 + e3;
        f[4 * i + 1] = e1 + e2;
        f[4 * i + 2] = e1 - e2;
        f[4 * i + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        int f0 = f[j], f1 = f[4 + j], f2 = f[8 + j], f3 = f[12 + j];
        int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        r[j] = (int16_t)((g0 + g3 + 32) >> 6);
        r[4 + j] = (int16_t)((g1 + g2 + 32) >> 6);
        r[8 + j] = (int16_t)((g1 - g2 + 32) >> 6);
        r[12 + j] = (int16_t)((g0 - g3 + 32) >> 6);
    }
}
/* 4x4 Hadamard Y = A X A, A symmetric (8.5.10) */
static void hadamard4x4(const int x[16], int y[16]) {
    int t[16];
    for (int i = 0; i < 4; i++) {
        int a = x[4 * i], b = x[4 * i + 1], c = x[4 * i + 2], d = x[4 * i + 3];
        t[4 * i + 0] = a + b + c + d;
        t[4 * i + 1] = a + b - c - d;
        t[4 * i + 2] = a - b - c + d;
        t[4 * i + 3] = a - b + c - d;
    }
    for (int j = 0; j < 4; j++) {
        int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
        y[j] = a + b + c + d;
        y[4 + j] = a + b - c - d;
        y[8 + j] = a - b - c + d;
        y[12 + j] = a - b + c - d;
    }
}
/* 2x2 transform of chroma DC (8.5.11.1), raster [a b; c d] */
static void hadamard2x2(const int x[4], int y[4]) {
    y[0] = x[0] + x[1] + x[2] + x[3];
    y[1] = x[0] - x[1] + x[2] - x[3];
    y[2] = x[0] + x[1] - x[2] - x[3];
    y[3] = x[0] - x[1] - x[2] + x[3];
}

/* ------------------------------------------------------------------ quantiser */
static inline int quant_f(int qbits, int intra) { return intra ? (1 << qbits) / 3 : (1 << qbits) / 6; }

int orc_quant(int coef, int qp, int raster_pos, int intra) {
    int qbits = 15 + qp / 6;
    uint32_t mf = orc_quant_mf[qp % 6][orc_pos_class(raster_pos)];
    uint32_t a = (uint32_t)(coef < 0 ? -coef : coef);
    int lv = (int)((a * mf + (uint32_t)quant_f(qbits, intra)) >> qbits);
    return coef < 0 ? -lv : lv;
}
int orc_dequant(int level, int qp, int raster_pos) {
    return (level * orc_dequant_v[qp % 6][orc_pos_class(raster_pos)]) << (qp / 6);
}
/* Intra16x16 luma DC: level = (|Y|*MF0 + 4f) >> (qbits+2) on the un-halved Hadamard output */
static int quant_luma_dc(int y, int qp) {
    int qbits = 15 + qp / 6;
    uint32_t a = (uint32_t)(y < 0 ? -y : y);
    int lv = (int)((a * (uint32_t)orc_quant_mf[qp % 6][0] + 4u * (uint32_t)quant_f(qbits, 1)) >> (qbits + 2));
    return y < 0 ? -lv : lv;
}
static int dequant_luma_dc(int f, int qp) { /* 8.5.10 */
    int ls = 16 * orc_dequant_v[qp % 6][0];
    if (qp >= 36) return (f * ls) << (qp / 6 - 6);
    return (f * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
}
static int quant_chroma_dc(int y, int qpc, int intra) {
    int qbits = 15 + qpc / 6;
    uint32_t a = (uint32_t)(y < 0 ? -y : y);
    int lv = (int)((a * (uint32_t)orc_quant_mf[qpc % 6][0] + 2u * (uint32_t)quant_f(qbits, intra)) >> (qbits + 1));
    return y < 0 ? -lv : lv;
}
static int dequant_chroma_dc(int f, int qpc) { /* 8.5.11.2 */
    int ls = 16 * orc_dequant_v[qpc % 6][0];
    return ((f * ls) << (qpc / 6)) >> 5;
}

/* ------------------------------------------------------------------ ME rank */
/* spec/CODEC.md section 1a */
static int rc_ilog2q(uint32_t c) {
    if (c < 1) c = 1;
    int k = 31 - __builtin_clz(c);
    int frac = k >= 2 ? (int)((c >> (k - 2)) & 3u) : (int)((c << (2 - k)) & 3u);
    return 4 * k + frac;
}
int orc_rc_qp(int base_qp, int is_idr, int have_history, uint64_t prev_sad_sum, int nmb) {
    int qp = base_qp;
    if (is_idr)
        qp = base_qp - 3;
    else if (have_history) {
        uint32_t c = (uint32_t)(prev_sad_sum / (uint64_t)(nmb > 0 ? nmb : 1));
        int num = 3 * (rc_ilog2q(c) - 36);
        int delta = num >= 0 ? num / 5 : -((-num + 4) / 5); /* floor division */
        if (delta < -6) delta = -6;
        if (delta > 6) delta = 6;
        qp = base_qp + delta;
    }
    return qp < 12 ? 12 : qp > 51 ? 51 : qp;
}

int orc_me_rank(int dx, int dy) {
    /* position of (dx,dy) in the list of all 33x33 displacements sorted by
     * (|dx|+|dy|, dy, dx) ascending: deterministic tie-break preferring short vectors */
    int l1 = abs(dx) + abs(dy), rank = 0;
    for (int y = -ORC_ME_RANGE; y <= ORC_ME_RANGE; y++)
        for (int x = -ORC_ME_RANGE; x <= ORC_ME_RANGE; x++) {
            int l = abs(x) + abs(y);
            if (l < l1 || (l == l1 && (y < dy || (y == dy && x < dx)))) rank++;
        }
    return rank;
}

/* ------------------------------------------------------------------ CAVLC */
static void cavlc_block(bw_t *b, const int16_t *coef, int max_coef, int nC) {
    /* 7.3.5.3.2 residual_block_cavlc / 9.2 */
    int level[16], run[16];
    int total = 0, last = -1;
    for (int i = 0; i < max_coef; i++)
        if (coef[i]) last = i;
    /* levels listed from the highest-frequency non-zero coefficient downwards */
    int zeros = 0, total_zeros = 0;
    for (int i = last; i >= 0; i--) {
        if (coef[i]) {
            if (total > 0) run[total - 1] = zeros;
            level[total++] = coef[i];
            zeros = 0;
        } else {
            zeros++;
        }
    }
    if (total > 0) {
        run[total - 1] = zeros;
        total_zeros = last + 1 - total;
    }
    int t1 = 0;
    while (t1 < total && t1 < 3 && (level[t1] == 1 || level[t1] == -1)) t1++;

    if (nC < 0) {
        bw_put(b, orc_chroma_dc_coeff_token_bits[4 * total + t1], orc_chroma_dc_coeff_token_len[4 * total + t1]);
    } else {
        int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        bw_put(b, orc_coeff_token_bits[tab][4 * total + t1], orc_coeff_token_len[tab][4 * total + t1]);
    }
    if (total == 0) return;
    for (int i = 0; i < t1; i++) bw_put(b, level[i] < 0, 1);

    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = t1; i < total; i++) {
        int lv = level[i];
        int alv = lv < 0 ? -lv : lv;
        int code = 2 * alv - 2 + (lv < 0);
        if (i == t1 && t1 < 3) code -= 2;
        if (suffix_len == 0) {
            if (code < 14) {
                bw_put(b, 1, code + 1);
            } else if (code < 30) {
                bw_put(b, 1, 15);
                bw_put(b, (uint32_t)(code - 14), 4);
            } else {
                bw_put(b, 1, 16);
                bw_put(b, (uint32_t)(code - 30), 12);
            }
        } else {
            if (code < (15 << suffix_len)) {
                bw_put(b, 1, (code >> suffix_len) + 1);
                bw_put(b, (uint32_t)(code & ((1 << suffix_len) - 1)), suffix_len);
            } else {
                bw_put(b, 1, 16);
                bw_put(b, (uint32_t)(code - (15 << suffix_len)), 12);
            }
        }
        if (suffix_len == 0) suffix_len = 1;
        if (alv > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    if (total < max_coef) {
        if (max_coef == 4)
            bw_put(b, orc_chroma_dc_total_zeros_bits[total - 1][total_zeros],
                   orc_chroma_dc_total_zeros_len[total - 1][total_zeros]);
        else
            bw_put(b, orc_total_zeros_bits[total - 1][total_zeros], orc_total_zeros_len[total - 1][total_zeros]);
    }
    int zeros_left = total_zeros;
    for (int i = 0; i < total - 1 && zeros_left > 0; i++) {
        int zl = zeros_left > 7 ? 7 : zeros_left;
        bw_put(b, orc_run_bits[zl - 1][run[i]], orc_run_len[zl - 1][run[i]]);
        zeros_left -= run[i];
    }
}

int orc_cavlc_block_bits(const int16_t *coef, int max_coef, int nC, uint8_t *out, size_t cap) {
    bw_t b;
    bw_init(&b, out, cap);
    cavlc_block(&b, coef, max_coef, nC);
    int bits = (int)b.total_bits;
    if (b.nacc) bw_put(&b, 0, 8 - b.nacc);
    return b.overflow ? -1 : bits;
}

size_t orc_escape(const uint8_t *rbsp, size_t n, uint8_t *out) {
    size_t o = 0;
    int z = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = rbsp[i];
        if (z == 2 && c <= 3) {
            out[o++] = 3;
            z = 0;
        }
        out[o++] = c;
        z = (c == 0) ? z + 1 : 0;
    }
    return o;
}

/* ------------------------------------------------------------------ encoder */
#define LPAD 32 /* luma border of the padded reference copy */
#define CPAD 16

struct orc_enc {
    orc_params p;
    int mbw, mbh, cw, ch;
    uint8_t *cur[3], *rec[3];
    uint8_t *rec_pre[3]; /* deblock: the reconstruction before the in-loop filter (introspection) */
    uint8_t *refp[3]; /* padded reference planes (edge-replicated, 8.4.2.2.1 clipping) */
    int refp_stride[3];
    long frame_idx;
    int frame_num, idr_count;
    int *slice_rows;
    int8_t *mv;  /* [mb][2] integer-sample vector of the full search */
    int8_t *mvq; /* [mb][2] the coded vector in quarter-sample units: 4 * mv, or refined (spec/CODEC.md section 2a) */
    uint16_t *sad;
    int16_t *levels;
    uint32_t *mb_bits;
    uint8_t *nnz_l, *nnz_cb, *nnz_cr;
    uint8_t *mb_cbp, *mb_i16flags, *mb_i16mode;
    uint8_t *mb_i4;      /* [mb] the macroblock is coded Intra4x4 (I_NxN) */
    int8_t *mb_i4mode;   /* [mb][16] Intra4x4PredMode by luma4x4BlkIdx (-1 everywhere in macroblocks that are not Intra4x4) */
    uint8_t cbp_to_code_intra[48];
    uint16_t rank[33 * 33];
    uint8_t cbp_to_code[48];
    uint8_t **slice_buf;
    size_t slice_cap;
    size_t *slice_len;
    uint8_t *esc_buf;
    int band_s0, band_s1; /* slices coded by this instance */
    int qp_cur;           /* QP of the picture being coded (p.qp at fixed QP; spec/CODEC.md section 1a under rate control) */
    uint64_t prev_sad_sum; /* sum of the best SADs of the previous picture's macroblocks */
    int prev_was_p;
};

static int pick_level(int mbs, int fps_num, int fps_den) {
    static const struct {
        int idc, max_fs;
        long max_mbps;
    } lv[] = {{10, 99, 1485},     {11, 396, 3000},     {12, 396, 6000},     {13, 396, 11880},
              {20, 396, 11880},   {21, 792, 19800},    {22, 1620, 20250},   {30, 1620, 40500},
              {31, 3600, 108000}, {32, 5120, 216000},  {40, 8192, 245760},  {42, 8704, 522240},
              {50, 22080, 589824}, {51, 36864, 983040}, {52, 36864, 2073600}};
    long mbps = 0;
    if (fps_num > 0 && fps_den > 0) mbps = ((long)mbs * fps_num + fps_den - 1) / fps_den;
    for (unsigned i = 0; i < sizeof(lv) / sizeof(lv[0]); i++)
        if (mbs <= lv[i].max_fs && mbps <= lv[i].max_mbps) return lv[i].idc;
    return 52;
}

orc_enc *orc_enc_open(const orc_params *p) {
    if (!p || p->width <= 0 || p->height <= 0 || (p->width & 1) || (p->height & 1)) return NULL;
    if (p->qp < 12 || p->qp > 51) return NULL;
    orc_enc *e = (orc_enc *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->p = *p;
    if (e->p.keyint <= 0) e->p.keyint = 1;
    if (e->p.threads <= 0) e->p.threads = 1;
    e->idr_count = e->p.idr_base;
    e->mbw = (p->width + 15) / 16;
    e->mbh = (p->height + 15) / 16;
    if (e->p.n_slices < 1) e->p.n_slices = 1;
    if (e->p.n_slices > e->mbh) e->p.n_slices = e->mbh;
    e->cw = e->mbw * 16;
    e->ch = e->mbh * 16;
    int nmb = e->mbw * e->mbh;
    for (int k = 0; k < 3; k++) {
        int w = k ? e->cw / 2 : e->cw, h = k ? e->ch / 2 : e->ch;
        int pad = k ? CPAD : LPAD;
        e->cur[k] = (uint8_t *)calloc((size_t)w * h, 1);
        e->rec[k] = (uint8_t *)calloc((size_t)w * h, 1);
        e->refp_stride[k] = w + 2 * pad;
        e->refp[k] = (uint8_t *)calloc((size_t)(w + 2 * pad) * (h + 2 * pad), 1);
    }
    e->slice_rows = (int *)calloc((size_t)e->p.n_slices + 1, sizeof(int));
    {
        int base = e->mbh / e->p.n_slices, rem = e->mbh % e->p.n_slices, r = 0;
        for (int s = 0; s < e->p.n_slices; s++) {
            e->slice_rows[s] = r;
            r += base + (s < rem ? 1 : 0);
        }
        e->slice_rows[e->p.n_slices] = r;
    }
    e->mv = (int8_t *)calloc((size_t)nmb * 2, 1);
    e->mvq = (int8_t *)calloc((size_t)nmb * 2, 1);
    e->sad = (uint16_t *)calloc((size_t)nmb, 2);
    e->levels = (int16_t *)calloc((size_t)nmb * ORC_MB_COEFS, 2);
    e->mb_bits = (uint32_t *)calloc((size_t)nmb, 4);
    e->nnz_l = (uint8_t *)calloc((size_t)nmb * 16, 1);
    e->nnz_cb = (uint8_t *)calloc((size_t)nmb * 4, 1);
    e->nnz_cr = (uint8_t *)calloc((size_t)nmb * 4, 1);
    e->mb_cbp = (uint8_t *)calloc((size_t)nmb, 1);
    e->mb_i16flags = (uint8_t *)calloc((size_t)nmb, 1);
    e->mb_i16mode = (uint8_t *)calloc((size_t)nmb, 1);
    e->mb_i4 = (uint8_t *)calloc((size_t)nmb, 1);
    e->mb_i4mode = (int8_t *)malloc((size_t)nmb * 16);
    memset(e->mb_i4mode, -1, (size_t)nmb * 16);
    for (int dy = -16; dy <= 16; dy++)
        for (int dx = -16; dx <= 16; dx++) e->rank[(dy + 16) * 33 + dx + 16] = (uint16_t)orc_me_rank(dx, dy);
    for (int i = 0; i < 48; i++) e->cbp_to_code[orc_golomb_to_inter_cbp[i]] = (uint8_t)i;
    for (int i = 0; i < 48; i++) e->cbp_to_code_intra[orc_golomb_to_intra_cbp[i]] = (uint8_t)i;
    e->band_s0 = 0;
    e->band_s1 = e->p.n_slices;
    e->slice_buf = (uint8_t **)calloc((size_t)e->p.n_slices, sizeof(uint8_t *));
    e->slice_len = (size_t *)calloc((size_t)e->p.n_slices, sizeof(size_t));
    {
        int maxrows = 0;
        for (int s = 0; s < e->p.n_slices; s++) {
            int r = e->slice_rows[s + 1] - e->slice_rows[s];
            if (r > maxrows) maxrows = r;
        }
        e->slice_cap = (size_t)maxrows * e->mbw * 1600 + 64;
        for (int s = 0; s < e->p.n_slices; s++) e->slice_buf[s] = (uint8_t *)malloc(e->slice_cap);
        e->esc_buf = (uint8_t *)malloc(e->slice_cap * 3 / 2 + 16);
    }
    return e;
}

void orc_enc_close(orc_enc *e) {
    if (!e) return;
    for (int k = 0; k < 3; k++) {
        free(e->cur[k]);
        free(e->rec[k]);
        free(e->refp[k]);
        free(e->rec_pre[k]);
    }
    for (int s = 0; s < e->p.n_slices; s++) free(e->slice_buf[s]);
    free(e->slice_buf);
    free(e->slice_len);
    free(e->esc_buf);
    free(e->slice_rows);
    free(e->mv);
    free(e->mvq);
    free(e->sad);
    free(e->levels);
    free(e->mb_bits);
    free(e->nnz_l);
    free(e->nnz_cb);
    free(e->nnz_cr);
    free(e->mb_cbp);
    free(e->mb_i16flags);
    free(e->mb_i16mode);
    free(e->mb_i4);
    free(e->mb_i4mode);
    free(e);
}

int orc_enc_picture_qp(const orc_enc *e) { return e->qp_cur; }
int orc_enc_mb_width(const orc_enc *e) { return e->mbw; }
int orc_enc_mb_height(const orc_enc *e) { return e->mbh; }
const int8_t *orc_enc_mv(const orc_enc *e) { return e->mv; }
const int8_t *orc_enc_mvq(const orc_enc *e) { return e->mvq; }
const uint16_t *orc_enc_sad(const orc_enc *e) { return e->sad; }
const int16_t *orc_enc_levels(const orc_enc *e) { return e->levels; }
const uint32_t *orc_enc_mb_bits(const orc_enc *e) { return e->mb_bits; }
const int *orc_enc_slice_rows(const orc_enc *e) { return e->slice_rows; }
const uint8_t *orc_enc_recon_unfiltered(const orc_enc *e, int plane) { return e->rec_pre[plane] ? e->rec_pre[plane] : e->rec[plane]; }
const uint8_t *orc_enc_recon(const orc_enc *e, int plane, int *stride) {
    if (stride) *stride = plane ? e->cw / 2 : e->cw;
    return e->rec[plane];
}

/* copy the display-size input into the coded-size frame, replicating the right
 * and bottom edges (the reference hands over w x h planes, av_encode.c:545-547) */
static void ingest_plane(uint8_t *dst, int cw, int ch, const uint8_t *src, int stride, int w, int h) {
    for (int y = 0; y < ch; y++) {
        const uint8_t *s = src + (size_t)(y < h ? y : h - 1) * stride;
        uint8_t *d = dst + (size_t)y * cw;
        memcpy(d, s, (size_t)w);
        for (int x = w; x < cw; x++) d[x] = s[w - 1];
    }
}
/* build the edge-replicated padded reference from the reconstructed frame */
static void pad_plane(uint8_t *dst, int dstride, int pad, const uint8_t *src, int w, int h) {
    for (int y = -pad; y < h + pad; y++) {
        int sy = y < 0 ? 0 : y >= h ? h - 1 : y;
        const uint8_t *s = src + (size_t)sy * w;
        uint8_t *d = dst + (size_t)(y + pad) * dstride;
        memset(d, s[0], (size_t)pad);
        memcpy(d + pad, s, (size_t)w);
        memset(d + pad + w, s[w - 1], (size_t)pad);
    }
}

/* ---- motion estimation: exhaustive +-16 integer-pel, 16x16 luma SAD (a4.1) ---- */
static void me_mb(orc_enc *e, int mx, int my) {
    const uint8_t *cur = e->cur[0] + (size_t)(my * 16) * e->cw + mx * 16;
    int rs = e->refp_stride[0];
    const uint8_t *ref0 = e->refp[0] + (size_t)(my * 16 + LPAD) * rs + mx * 16 + LPAD;
    uint32_t best = 0xffffffffu;
    int bdx = 0, bdy = 0;
    for (int dy = -16; dy <= 16; dy++)
        for (int dx = -16; dx <= 16; dx++) {
            const uint8_t *r = ref0 + (ptrdiff_t)dy * rs + dx;
            uint32_t sad = 0;
            for (int y = 0; y < 16; y++) {
                const uint8_t *c = cur + (size_t)y * e->cw;
                const uint8_t *q = r + (size_t)y * rs;
                for (int x = 0; x < 16; x++) sad += (uint32_t)abs((int)c[x] - (int)q[x]);
            }
            uint32_t key = (sad << 11) | e->rank[(dy + 16) * 33 + dx + 16];
            if (key < best) {
                best = key;
                bdx = dx;
                bdy = dy;
            }
        }
    int mb = my * e->mbw + mx;
    e->mv[2 * mb] = (int8_t)bdx;
    e->mv[2 * mb + 1] = (int8_t)bdy;
    e->mvq[2 * mb] = (int8_t)(4 * bdx);
    e->mvq[2 * mb + 1] = (int8_t)(4 * bdy);
    e->sad[mb] = (uint16_t)(best >> 11);
}

/* ---- sub-sample refinement around the full search's vector (spec/CODEC.md section 2a): a half-sample stage (the vector
 * and its eight neighbours at +-2 quarter units), then with subpel = 2 a quarter-sample stage around that stage's winner
 * (+-1).  A stage's winner is the minimum of (SAD << 4) | index, index 0 = the stage's centre, 1..8 = the neighbours in
 * raster order; predictions by 8.4.2.2.1 (orc_interp.h).  Behind x264_encoder_encode (av_encode.c:970) this is what the
 * reference's presets (av_encode.c:365) ask libx264 for as subpixel refinement. ---- */
static void subpel_mb(orc_enc *e, int mx, int my) {
    const int mb = my * e->mbw + mx;
    const uint8_t *cur = e->cur[0] + (size_t)(my * 16) * e->cw + mx * 16;
    const orc_plane ref = {e->refp[0] + (size_t)LPAD * e->refp_stride[0] + LPAD, e->refp_stride[0], e->cw, e->ch};
    int cx = e->mvq[2 * mb], cy = e->mvq[2 * mb + 1];
    uint32_t csad = e->sad[mb];
    for (int stage = 0; stage < e->p.subpel && stage < 2; stage++) {
        const int step = stage == 0 ? 2 : 1;
        uint32_t best = csad << 4; /* index 0: the centre, whose SAD is known */
        int bx = cx, by = cy, idx = 0;
        for (int ey = -1; ey <= 1; ey++)
            for (int ex = -1; ex <= 1; ex++) {
                if (!ex && !ey) continue;
                idx++;
                const int qx = cx + step * ex, qy = cy + step * ey;
                uint32_t sad = 0;
                for (int y = 0; y < 16; y++)
                    for (int x = 0; x < 16; x++)
                        sad += (uint32_t)abs((int)cur[(size_t)y * e->cw + x] - orc_luma_qpel(&ref, 4 * (mx * 16 + x) + qx, 4 * (my * 16 + y) + qy));
                const uint32_t key = (sad << 4) | (uint32_t)idx;
                if (key < best) best = key, bx = qx, by = qy;
            }
        cx = bx, cy = by, csad = best >> 4;
    }
    e->mvq[2 * mb] = (int8_t)cx;
    e->mvq[2 * mb + 1] = (int8_t)cy;
    e->sad[mb] = (uint16_t)csad;
}

/* stand-alone interpolation (unit tests): bw x bh luma prediction block at (xq, yq) quarter-sample units of a w x h plane */
void orc_luma_interp(const uint8_t *ref, int stride, int w, int h, int xq, int yq, int bw, int bh, uint8_t *dst) {
    const orc_plane r = {ref, stride, w, h};
    orc_luma_pred_block(&r, xq, yq, bw, bh, dst, bw);
}

static inline uint8_t clip_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* ---- chroma of one macroblock: 4 blocks/plane + 2x2 DC (a4.2, a4.3, a4.5) ---- */
static void code_chroma(orc_enc *e, int mx, int my, int plane, const uint8_t pred[64], int intra, int16_t *lv_dc,
                        int16_t *lv_ac /* [4][16] */) {
    int cs = e->cw / 2;
    int qpc = orc_chroma_qp[e->qp_cur];
    const uint8_t *cur = e->cur[plane] + (size_t)(my * 8) * cs + mx * 8;
    uint8_t *rec = e->rec[plane] + (size_t)(my * 8) * cs + mx * 8;
    int dc[4], ydc[4];
    for (int k = 0; k < 4; k++) {
        int bx = (k & 1) * 4, by = (k >> 1) * 4;
        int16_t res[16], w[16];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                res[4 * y + x] = (int16_t)((int)cur[(size_t)(by + y) * cs + bx + x] - (int)pred[(by + y) * 8 + bx + x]);
        orc_fdct4x4(res, w);
        dc[k] = w[0];
        lv_ac[16 * k] = 0;
        for (int i = 1; i < 16; i++) lv_ac[16 * k + i] = (int16_t)orc_quant(w[orc_zigzag4x4[i]], qpc, orc_zigzag4x4[i], intra);
    }
    hadamard2x2(dc, ydc);
    int c[4], f[4];
    for (int k = 0; k < 4; k++) {
        lv_dc[k] = (int16_t)quant_chroma_dc(ydc[k], qpc, intra);
        c[k] = lv_dc[k];
    }
    for (int k = 4; k < 16; k++) lv_dc[k] = 0;
    hadamard2x2(c, f);
    for (int k = 0; k < 4; k++) {
        int bx = (k & 1) * 4, by = (k >> 1) * 4;
        int32_t d[16];
        int16_t r[16];
        memset(d, 0, sizeof(d));
        d[0] = dequant_chroma_dc(f[k], qpc);
        for (int i = 1; i < 16; i++) d[orc_zigzag4x4[i]] = orc_dequant(lv_ac[16 * k + i], qpc, orc_zigzag4x4[i]);
        orc_idct4x4(d, r);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                rec[(size_t)(by + y) * cs + bx + x] = clip_u8((int)pred[(by + y) * 8 + bx + x] + r[4 * y + x]);
    }
}

static int count_nz(const int16_t *c, int from, int to) {
    int n = 0;
    for (int i = from; i < to; i++) n += c[i] != 0;
    return n;
}

/* derive cbp / nnz bookkeeping for chroma from the level record */
static int chroma_cbp_and_nnz(orc_enc *e, int mb, const int16_t *lv) {
    int any_dc = count_nz(lv + 17 * 16, 0, 4) + count_nz(lv + 18 * 16, 0, 4);
    int any_ac = 0;
    for (int k = 0; k < 4; k++) {
        int ncb = count_nz(lv + (19 + k) * 16, 1, 16), ncr = count_nz(lv + (23 + k) * 16, 1, 16);
        e->nnz_cb[mb * 4 + k] = (uint8_t)ncb;
        e->nnz_cr[mb * 4 + k] = (uint8_t)ncr;
        any_ac += ncb + ncr;
    }
    int cbpc = any_ac ? 2 : any_dc ? 1 : 0;
    if (cbpc < 2)
        for (int k = 0; k < 4; k++) e->nnz_cb[mb * 4 + k] = e->nnz_cr[mb * 4 + k] = 0;
    return cbpc;
}

/* ---- Intra16x16 (DC prediction) macroblock: transform, quant, recon ---- */
static void code_intra_chroma_and_flags(orc_enc *e, int mx, int my, int avail_l, int avail_t, int is_i4);

/* test hooks: one Intra4x4 predictor from explicit neighbours, and the availability of a block's upper-right samples */
int orc_intra4_predict(const uint8_t top[8], const uint8_t left[4], int corner, int have_top, int have_left, int have_corner, int mode,
                       uint8_t out[16]) {
    orc_i4_nb nb;
    nb.top = have_top, nb.left = have_left, nb.corner = have_corner, nb.topright = have_top;
    for (int i = 0; i < 8; i++) nb.t[i] = top[i];
    for (int i = 0; i < 4; i++) nb.l[i] = left[i];
    nb.c = corner;
    if (mode < 0 || mode > 8 || !orc_i4_mode_ok(&nb, mode)) return -1;
    orc_i4_predict(&nb, mode, out);
    return 0;
}
int orc_intra4_topright_available(int bx, int by, int aT, int aTR) {
    uint8_t px[64 * 64];
    orc_i4_nb nb;
    memset(px, 1, sizeof(px));
    orc_i4_neighbours(px + 20 * 64 + 20, 64, bx, by, 1, aT, aT, aTR, &nb);
    return nb.topright;
}

/* Intra4x4 luma of one macroblock (8.3.1, spec/CODEC.md section 3): blocks in luma4x4BlkIdx order, each predicted from the
 * reconstruction so far by the usable mode with the smallest SAD against the source (the smaller mode number on ties),
 * then transformed, quantised (all 16 coefficients, intra rounding) and reconstructed.  Returns 1 and leaves levels,
 * counts, modes, the luma half of mb_cbp and the reconstruction in place iff the sum of the blocks' SADs is strictly
 * below sad16; else returns 0 (levels and reconstruction are then overwritten by the Intra16x16 path). */
static int code_luma_intra4x4(orc_enc *e, int mx, int my, int slice_first_row, long sad16) {
    const int mb = my * e->mbw + mx, qp = e->qp_cur;
    const int aL = mx > 0, aT = my > slice_first_row, aTL = aL && aT, aTR = aT && mx + 1 < e->mbw;
    uint8_t *rec = e->rec[0] + (size_t)(my * 16) * e->cw + mx * 16;
    const uint8_t *cur = e->cur[0] + (size_t)(my * 16) * e->cw + mx * 16;
    int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    int8_t modes[16];
    uint8_t nnz[16];
    long total = 0;
    for (int blk = 0; blk < 16; blk++) {
        const int bx = orc_blk_x[blk], by = orc_blk_y[blk];
        orc_i4_nb nb;
        orc_i4_neighbours(rec, e->cw, bx, by, aL, aT, aTL, aTR, &nb);
        uint8_t pred[16], bestp[16];
        long bs = -1;
        int bm = 2;
        for (int m = 0; m < 9; m++) {
            if (!orc_i4_mode_ok(&nb, m)) continue;
            orc_i4_predict(&nb, m, pred);
            long sd = 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) sd += abs((int)cur[(size_t)(4 * by + y) * e->cw + 4 * bx + x] - (int)pred[4 * y + x]);
            if (bs < 0 || sd < bs) bs = sd, bm = m, memcpy(bestp, pred, 16);
        }
        total += bs;
        modes[blk] = (int8_t)bm;
        int16_t res[16], w[16], r[16];
        int32_t d[16];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) res[4 * y + x] = (int16_t)((int)cur[(size_t)(4 * by + y) * e->cw + 4 * bx + x] - (int)bestp[4 * y + x]);
        orc_fdct4x4(res, w);
        for (int i = 0; i < 16; i++) {
            lv[blk * 16 + i] = (int16_t)orc_quant(w[orc_zigzag4x4[i]], qp, orc_zigzag4x4[i], 1);
            d[orc_zigzag4x4[i]] = orc_dequant(lv[blk * 16 + i], qp, orc_zigzag4x4[i]);
        }
        orc_idct4x4(d, r);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) rec[(size_t)(4 * by + y) * e->cw + 4 * bx + x] = clip_u8((int)bestp[4 * y + x] + r[4 * y + x]);
        nnz[blk] = (uint8_t)count_nz(lv + blk * 16, 0, 16);
    }
    if (!(total < sad16)) return 0;
    int cbpl = 0;
    for (int blk = 0; blk < 16; blk++) {
        e->nnz_l[mb * 16 + 4 * orc_blk_y[blk] + orc_blk_x[blk]] = nnz[blk];
        if (nnz[blk]) cbpl |= 1 << (blk >> 2);
    }
    memset(lv + 16 * 16, 0, 16 * sizeof(int16_t)); /* no Intra16x16 DC block */
    memcpy(e->mb_i4mode + (size_t)mb * 16, modes, 16);
    e->mb_i4[mb] = 1;
    e->mb_cbp[mb] = (uint8_t)cbpl; /* (the chroma half follows) */
    return 1;
}

static void code_mb_intra(orc_enc *e, int mx, int my, int slice_first_row) {
    int mb = my * e->mbw + mx;
    int qp = e->qp_cur;
    int avail_l = mx > 0, avail_t = my > slice_first_row;
    int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    memset(lv, 0, ORC_MB_COEFS * sizeof(int16_t));
    e->mv[2 * mb] = e->mv[2 * mb + 1] = 0;
    e->mvq[2 * mb] = e->mvq[2 * mb + 1] = 0;
    e->sad[mb] = 0;

    /* luma DC prediction (8.3.3.3) */
    uint8_t *rec = e->rec[0] + (size_t)(my * 16) * e->cw + mx * 16;
    const uint8_t *cur = e->cur[0] + (size_t)(my * 16) * e->cw + mx * 16;
    int sum = 0, dcp;
    if (avail_t)
        for (int x = 0; x < 16; x++) sum += rec[-(ptrdiff_t)e->cw + x];
    if (avail_l)
        for (int y = 0; y < 16; y++) sum += rec[(size_t)y * e->cw - 1];
    if (avail_l && avail_t)
        dcp = (sum + 16) >> 5;
    else if (avail_l || avail_t)
        dcp = (sum + 8) >> 4;
    else
        dcp = 128;
    /* the 16x16 predictor: DC, or (intra_modes) whichever of vertical (0), horizontal (1), DC (2) has the smallest SAD
     * against the source, the smaller mode number on ties; vertical needs the macroblock above, horizontal the one to the
     * left, inside the slice (8.3.3) */
    uint8_t pred16[256];
    int mode = 2;
    long best16 = 0; /* SAD of the chosen 16x16 predictor */
    memset(pred16, dcp, 256);
    if (e->p.intra_modes) {
        long sad[3] = {-1, -1, 0};
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) sad[2] += abs((int)cur[(size_t)y * e->cw + x] - dcp);
        if (avail_t) {
            sad[0] = 0;
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) sad[0] += abs((int)cur[(size_t)y * e->cw + x] - (int)rec[-(ptrdiff_t)e->cw + x]);
        }
        if (avail_l) {
            sad[1] = 0;
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) sad[1] += abs((int)cur[(size_t)y * e->cw + x] - (int)rec[(size_t)y * e->cw - 1]);
        }
        long best = sad[2];
        for (int m = 1; m >= 0; m--)
            if (sad[m] >= 0 && sad[m] <= best) best = sad[m], mode = m;
        /* intra_modes = 2: plane prediction (8.3.3.4) as a fourth candidate when the left, upper and upper-left macroblocks are
         * in the slice; chosen only when strictly better (ties go to the smaller mode number) */
        uint8_t plane16[256];
        if (e->p.intra_modes >= 2 && avail_l && avail_t) {
            const uint8_t *top = rec - (ptrdiff_t)e->cw; /* top[-1] is the corner sample */
            int H = 0, V = 0;
            for (int k = 0; k < 8; k++) {
                H += (k + 1) * ((int)top[8 + k] - (int)top[6 - k]);
                V += (k + 1) * ((int)rec[(size_t)(8 + k) * e->cw - 1] - (k == 7 ? (int)top[-1] : (int)rec[(size_t)(6 - k) * e->cw - 1]));
            }
            const int a = 16 * ((int)rec[(size_t)15 * e->cw - 1] + (int)top[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
            long s3 = 0;
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) {
                    plane16[16 * y + x] = clip_u8((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
                    s3 += abs((int)cur[(size_t)y * e->cw + x] - (int)plane16[16 * y + x]);
                }
            if (s3 < best) best = s3, mode = 3;
        }
        best16 = best;
        if (mode == 3)
            memcpy(pred16, plane16, 256);
        else if (mode == 0)
            for (int y = 0; y < 16; y++) memcpy(pred16 + 16 * y, rec - (ptrdiff_t)e->cw, 16);
        else if (mode == 1)
            for (int y = 0; y < 16; y++) memset(pred16 + 16 * y, rec[(size_t)y * e->cw - 1], 16);
    }
    e->mb_i16mode[mb] = (uint8_t)mode;
    e->mb_i4[mb] = 0;
    memset(e->mb_i4mode + (size_t)mb * 16, -1, 16);
    /* intra_modes = 3: Intra4x4 (8.3.1) tried against the best Intra16x16 predictor; taken iff the sum of its blocks' SADs is
     * strictly smaller than the 16x16 predictor's (spec/CODEC.md section 3) */
    if (e->p.intra_modes >= 3 && code_luma_intra4x4(e, mx, my, slice_first_row, best16)) {
        code_intra_chroma_and_flags(e, mx, my, avail_l, avail_t, 1);
        return;
    }

    int dc[16], ydc[16];
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            int16_t res[16], w[16];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    res[4 * y + x] = (int16_t)((int)cur[(size_t)(4 * by + y) * e->cw + 4 * bx + x] - (int)pred16[16 * (4 * by + y) + 4 * bx + x]);
            orc_fdct4x4(res, w);
            dc[4 * by + bx] = w[0];
            int blk = -1;
            for (int k = 0; k < 16; k++)
                if (orc_blk_x[k] == bx && orc_blk_y[k] == by) blk = k;
            for (int i = 1; i < 16; i++) lv[blk * 16 + i] = (int16_t)orc_quant(w[orc_zigzag4x4[i]], qp, orc_zigzag4x4[i], 1);
        }
    hadamard4x4(dc, ydc);
    int c[16], f[16];
    for (int i = 0; i < 16; i++) {
        lv[16 * 16 + i] = (int16_t)quant_luma_dc(ydc[orc_zigzag4x4[i]], qp);
        c[orc_zigzag4x4[i]] = lv[16 * 16 + i];
    }
    hadamard4x4(c, f);
    int ac_flag = 0;
    for (int blk = 0; blk < 16; blk++) {
        int bx = orc_blk_x[blk], by = orc_blk_y[blk];
        int32_t d[16];
        int16_t r[16];
        d[0] = dequant_luma_dc(f[4 * by + bx], qp);
        for (int i = 1; i < 16; i++) d[orc_zigzag4x4[i]] = orc_dequant(lv[blk * 16 + i], qp, orc_zigzag4x4[i]);
        orc_idct4x4(d, r);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) rec[(size_t)(4 * by + y) * e->cw + 4 * bx + x] = clip_u8((int)pred16[16 * (4 * by + y) + 4 * bx + x] + r[4 * y + x]);
        int n = count_nz(lv + blk * 16, 1, 16);
        e->nnz_l[mb * 16 + 4 * by + bx] = (uint8_t)n;
        ac_flag |= n != 0;
    }
    if (!ac_flag) memset(e->nnz_l + mb * 16, 0, 16);

    code_intra_chroma_and_flags(e, mx, my, avail_l, avail_t, 0);
    e->mb_cbp[mb] = (uint8_t)((e->mb_cbp[mb] & 0x30) | (ac_flag ? 15 : 0));
}

/* chroma of an intra macroblock (DC prediction, 8.3.4.1-8.3.4.3, per 4x4 chroma block) and the macroblock's flags; the
 * luma half of mb_cbp is left to the caller (Intra16x16: all or nothing; Intra4x4: set by code_luma_intra4x4) */
static void code_intra_chroma_and_flags(orc_enc *e, int mx, int my, int avail_l, int avail_t, int is_i4) {
    int mb = my * e->mbw + mx;
    int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    int cs = e->cw / 2;
    for (int plane = 1; plane <= 2; plane++) {
        uint8_t pred[64];
        const uint8_t *rc = e->rec[plane] + (size_t)(my * 8) * cs + mx * 8;
        for (int k = 0; k < 4; k++) {
            int bx = (k & 1) * 4, by = (k >> 1) * 4;
            int st = 0, sl = 0, v;
            if (avail_t)
                for (int x = 0; x < 4; x++) st += rc[-(ptrdiff_t)cs + bx + x];
            if (avail_l)
                for (int y = 0; y < 4; y++) sl += rc[(size_t)(by + y) * cs - 1];
            if (k == 0 || k == 3) {
                if (avail_t && avail_l)
                    v = (st + sl + 4) >> 3;
                else if (avail_t)
                    v = (st + 2) >> 2;
                else if (avail_l)
                    v = (sl + 2) >> 2;
                else
                    v = 128;
            } else if (k == 1) { /* top-right block: top preferred */
                if (avail_t)
                    v = (st + 2) >> 2;
                else if (avail_l)
                    v = (sl + 2) >> 2;
                else
                    v = 128;
            } else { /* bottom-left block: left preferred */
                if (avail_l)
                    v = (sl + 2) >> 2;
                else if (avail_t)
                    v = (st + 2) >> 2;
                else
                    v = 128;
            }
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) pred[(by + y) * 8 + bx + x] = (uint8_t)v;
        }
        code_chroma(e, mx, my, plane, pred, 1, lv + (16 + plane) * 16, lv + (plane == 1 ? 19 : 23) * 16);
    }
    int cbpc = chroma_cbp_and_nnz(e, mb, lv);
    e->mb_cbp[mb] = (uint8_t)((cbpc << 4) | (is_i4 ? (e->mb_cbp[mb] & 15) : 0));
    e->mb_i16flags[mb] = 1; /* (an intra macroblock: what the in-loop filter's strengths ask) */
}

/* ---- P_L0_16x16 macroblock: MC prediction, transform, quant, recon ---- */
static void code_mb_inter(orc_enc *e, int mx, int my) {
    int mb = my * e->mbw + mx;
    int qp = e->qp_cur;
    const int qx = e->mvq[2 * mb], qy = e->mvq[2 * mb + 1]; /* quarter-sample units */
    int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    memset(lv, 0, ORC_MB_COEFS * sizeof(int16_t));

    /* luma prediction: the reference block at the vector, interpolated (8.4.2.2.1) when the vector is fractional */
    uint8_t predl[256];
    int rs = 16;
    const uint8_t *ref = predl;
    if ((qx | qy) & 3) {
        const orc_plane rp = {e->refp[0] + (size_t)LPAD * e->refp_stride[0] + LPAD, e->refp_stride[0], e->cw, e->ch};
        orc_luma_pred_block(&rp, 4 * mx * 16 + qx, 4 * my * 16 + qy, 16, 16, predl, 16);
    } else {
        rs = e->refp_stride[0];
        ref = e->refp[0] + (size_t)(my * 16 + (qy >> 2) + LPAD) * rs + mx * 16 + (qx >> 2) + LPAD;
    }
    const uint8_t *cur = e->cur[0] + (size_t)(my * 16) * e->cw + mx * 16;
    uint8_t *rec = e->rec[0] + (size_t)(my * 16) * e->cw + mx * 16;
    int cbpl = 0;
    for (int blk = 0; blk < 16; blk++) {
        int bx = orc_blk_x[blk], by = orc_blk_y[blk];
        int16_t res[16], w[16], r[16];
        int32_t d[16];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                res[4 * y + x] = (int16_t)((int)cur[(size_t)(4 * by + y) * e->cw + 4 * bx + x] -
                                           (int)ref[(size_t)(4 * by + y) * rs + 4 * bx + x]);
        orc_fdct4x4(res, w);
        int n = 0;
        for (int i = 0; i < 16; i++) {
            int l = orc_quant(w[orc_zigzag4x4[i]], qp, orc_zigzag4x4[i], 0);
            lv[blk * 16 + i] = (int16_t)l;
            d[orc_zigzag4x4[i]] = orc_dequant(l, qp, orc_zigzag4x4[i]);
            n += l != 0;
        }
        orc_idct4x4(d, r);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                rec[(size_t)(4 * by + y) * e->cw + 4 * bx + x] = clip_u8((int)ref[(size_t)(4 * by + y) * rs + 4 * bx + x] + r[4 * y + x]);
        e->nnz_l[mb * 16 + 4 * by + bx] = (uint8_t)n;
        if (n) cbpl |= 1 << (blk >> 2);
    }
    /* chroma MC (8.4.2.2.2): mvC = luma mv in 1/8 chroma sample units */
    int mvx = qx, mvy = qy;
    int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
    for (int plane = 1; plane <= 2; plane++) {
        int cst = e->refp_stride[plane];
        const uint8_t *rc = e->refp[plane] + (size_t)(my * 8 + yi + CPAD) * cst + mx * 8 + xi + CPAD;
        uint8_t pred[64];
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                int A = rc[(size_t)y * cst + x], B = rc[(size_t)y * cst + x + 1];
                int C = rc[(size_t)(y + 1) * cst + x], D = rc[(size_t)(y + 1) * cst + x + 1];
                pred[y * 8 + x] = (uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
            }
        code_chroma(e, mx, my, plane, pred, 0, lv + (16 + plane) * 16, lv + (plane == 1 ? 19 : 23) * 16);
    }
    int cbpc = chroma_cbp_and_nnz(e, mb, lv);
    e->mb_cbp[mb] = (uint8_t)((cbpc << 4) | cbpl);
    e->mb_i16flags[mb] = 0;
}

/* ---- neighbour-derived syntax: nC (9.2.1) and mvp (8.4.1.3) ---- */
static int nc_from(int availA, int nA, int availB, int nB) {
    if (availA && availB) return (nA + nB + 1) >> 1;
    if (availA) return nA;
    if (availB) return nB;
    return 0;
}
static int luma_nc(const orc_enc *e, int mx, int my, int bx, int by, int srow) {
    int mb = my * e->mbw + mx;
    int availA, availB, nA = 0, nB = 0;
    if (bx > 0) {
        availA = 1;
        nA = e->nnz_l[mb * 16 + 4 * by + bx - 1];
    } else {
        availA = mx > 0;
        if (availA) nA = e->nnz_l[(mb - 1) * 16 + 4 * by + 3];
    }
    if (by > 0) {
        availB = 1;
        nB = e->nnz_l[mb * 16 + 4 * (by - 1) + bx];
    } else {
        availB = my > srow;
        if (availB) nB = e->nnz_l[(mb - e->mbw) * 16 + 12 + bx];
    }
    return nc_from(availA, nA, availB, nB);
}
static int chroma_nc(const orc_enc *e, const uint8_t *nnz, int mx, int my, int bx, int by, int srow) {
    int mb = my * e->mbw + mx;
    int availA, availB, nA = 0, nB = 0;
    if (bx > 0) {
        availA = 1;
        nA = nnz[mb * 4 + 2 * by];
    } else {
        availA = mx > 0;
        if (availA) nA = nnz[(mb - 1) * 4 + 2 * by + 1];
    }
    if (by > 0) {
        availB = 1;
        nB = nnz[mb * 4 + bx];
    } else {
        availB = my > srow;
        if (availB) nB = nnz[(mb - e->mbw) * 4 + 2 + bx];
    }
    return nc_from(availA, nA, availB, nB);
}
static int median3(int a, int b, int c) {
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return c < mn ? mn : c > mx ? mx : c;
}
static void predict_mv(const orc_enc *e, int mx, int my, int srow, int *px, int *py) {
    int mb = my * e->mbw + mx;
    int aA = mx > 0, aB = my > srow, aC = my > srow && mx + 1 < e->mbw, aD = my > srow && mx > 0;
    int ax = 0, ay = 0, bx = 0, by = 0, cx = 0, cy = 0;
    if (aA) ax = e->mvq[2 * (mb - 1)], ay = e->mvq[2 * (mb - 1) + 1];
    if (aB) bx = e->mvq[2 * (mb - e->mbw)], by = e->mvq[2 * (mb - e->mbw) + 1];
    if (aC)
        cx = e->mvq[2 * (mb - e->mbw + 1)], cy = e->mvq[2 * (mb - e->mbw + 1) + 1];
    else if (aD) {
        aC = 1;
        cx = e->mvq[2 * (mb - e->mbw - 1)], cy = e->mvq[2 * (mb - e->mbw - 1) + 1];
    }
    if (!aB && !aC && aA) {
        *px = ax;
        *py = ay;
        return;
    }
    int n = aA + aB + aC;
    if (n == 1) {
        if (aA) *px = ax, *py = ay;
        else if (aB) *px = bx, *py = by;
        else *px = cx, *py = cy;
        return;
    }
    *px = median3(ax, bx, cx);
    *py = median3(ay, by, cy);
}

/* ---- macroblock_layer syntax (7.3.5) ---- */
static void write_chroma_residual(orc_enc *e, bw_t *b, int mx, int my, int srow, const int16_t *lv, int cbpc) {
    if (cbpc == 0) return;
    cavlc_block(b, lv + 17 * 16, 4, -1);
    cavlc_block(b, lv + 18 * 16, 4, -1);
    if (cbpc < 2) return;
    for (int plane = 0; plane < 2; plane++) {
        const uint8_t *nnz = plane ? e->nnz_cr : e->nnz_cb;
        for (int k = 0; k < 4; k++)
            cavlc_block(b, lv + ((plane ? 23 : 19) + k) * 16 + 1, 15, chroma_nc(e, nnz, mx, my, k & 1, k >> 1, srow));
    }
}
static void write_mb_intra(orc_enc *e, bw_t *b, int mx, int my, int srow) {
    int mb = my * e->mbw + mx;
    const int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    int cbp = e->mb_cbp[mb], cbpc = cbp >> 4, acf = (cbp & 15) != 0;
    if (e->mb_i4[mb]) {
        /* I_NxN (7.3.5, 7.3.5.1): mb_type 0; per block prev_intra4x4_pred_mode_flag [+ rem_intra4x4_pred_mode] against
         * predIntra4x4PredMode = min(mode of the block to the left, mode of the block above), DC when either macroblock is
         * missing, a macroblock that is not Intra4x4 counting as DC (8.3.1.1); intra_chroma_pred_mode; coded_block_pattern
         * me(v), Intra column; mb_qp_delta iff cbp != 0; the luma blocks of the 8x8s whose bit is set, 16 coefficients each */
        bw_ue(b, 0);
        const int8_t *own = e->mb_i4mode + (size_t)mb * 16;
        for (int blk = 0; blk < 16; blk++) {
            const int bx = orc_blk_x[blk], by = orc_blk_y[blk];
            int mA, mB;
            if (bx > 0) mA = own[orc_i4_idx(bx - 1, by)];
            else if (mx > 0) mA = e->mb_i4[mb - 1] ? e->mb_i4mode[(size_t)(mb - 1) * 16 + orc_i4_idx(3, by)] : 2;
            else mA = -1;
            if (by > 0) mB = own[orc_i4_idx(bx, by - 1)];
            else if (my > srow) mB = e->mb_i4[mb - e->mbw] ? e->mb_i4mode[(size_t)(mb - e->mbw) * 16 + orc_i4_idx(bx, 3)] : 2;
            else mB = -1;
            const int pm = orc_i4_pred_mode(mA, mB), m = own[blk];
            if (m == pm)
                bw_put(b, 1, 1);
            else {
                bw_put(b, 0, 1);
                bw_put(b, (uint32_t)(m < pm ? m : m - 1), 3);
            }
        }
        bw_ue(b, 0); /* intra_chroma_pred_mode = DC */
        bw_ue(b, e->cbp_to_code_intra[cbp]);
        if (cbp) bw_se(b, 0); /* mb_qp_delta */
        for (int q = 0; q < 4; q++)
            if (cbp & (1 << q))
                for (int k = 0; k < 4; k++) {
                    int blk = 4 * q + k;
                    cavlc_block(b, lv + blk * 16, 16, luma_nc(e, mx, my, orc_blk_x[blk], orc_blk_y[blk], srow));
                }
        write_chroma_residual(e, b, mx, my, srow, lv, cbpc);
        return;
    }
    bw_ue(b, (uint32_t)(1 + e->mb_i16mode[my * e->mbw + mx] + 4 * cbpc + 12 * acf)); /* mb_type, table 7-11: Intra16x16PredMode 0 V, 1 H, 2 DC, 3 plane */
    bw_ue(b, 0);                                                   /* intra_chroma_pred_mode = DC */
    bw_se(b, 0);                                                   /* mb_qp_delta */
    cavlc_block(b, lv + 16 * 16, 16, luma_nc(e, mx, my, 0, 0, srow));
    if (acf)
        for (int blk = 0; blk < 16; blk++)
            cavlc_block(b, lv + blk * 16 + 1, 15, luma_nc(e, mx, my, orc_blk_x[blk], orc_blk_y[blk], srow));
    write_chroma_residual(e, b, mx, my, srow, lv, cbpc);
}
/* P_Skip (7.3.4, 8.4.1.1): a macroblock is skipped when it has no residual (cbp = 0) and its vector equals the vector a
 * decoder infers for a skipped macroblock: zero if A or B is missing or has a zero vector, else the 16x16 median
 * prediction.  (Every neighbour in a P slice of this encoder is a P macroblock with refIdx 0.) */
static int mb_is_skip(const orc_enc *e, int mx, int my, int srow) {
    int mb = my * e->mbw + mx;
    if (e->mb_cbp[mb] != 0) return 0;
    int sx = 0, sy = 0;
    if (mx > 0 && my > srow) {
        const int8_t *a = e->mvq + 2 * (mb - 1), *bb = e->mvq + 2 * (mb - e->mbw);
        if (!((a[0] == 0 && a[1] == 0) || (bb[0] == 0 && bb[1] == 0))) predict_mv(e, mx, my, srow, &sx, &sy);
    }
    return e->mvq[2 * mb] == sx && e->mvq[2 * mb + 1] == sy;
}
static void write_mb_inter(orc_enc *e, bw_t *b, int mx, int my, int srow, int skip_run) {
    int mb = my * e->mbw + mx;
    const int16_t *lv = e->levels + (size_t)mb * ORC_MB_COEFS;
    int cbp = e->mb_cbp[mb], cbpc = cbp >> 4, cbpl = cbp & 15;
    int px, py;
    predict_mv(e, mx, my, srow, &px, &py);
    bw_ue(b, (uint32_t)skip_run); /* mb_skip_run: P_Skip macroblocks since the previous coded one */
    bw_ue(b, 0); /* mb_type P_L0_16x16 */
    bw_se(b, e->mvq[2 * mb] - px); /* mvd_l0 in quarter-sample units */
    bw_se(b, e->mvq[2 * mb + 1] - py);
    bw_ue(b, e->cbp_to_code[cbp]);
    if (cbp) bw_se(b, 0); /* mb_qp_delta */
    for (int q = 0; q < 4; q++)
        if (cbpl & (1 << q))
            for (int k = 0; k < 4; k++) {
                int blk = 4 * q + k;
                cavlc_block(b, lv + blk * 16, 16, luma_nc(e, mx, my, orc_blk_x[blk], orc_blk_y[blk], srow));
            }
    write_chroma_residual(e, b, mx, my, srow, lv, cbpc);
}

/* ---- parameter sets (7.3.2.1, 7.3.2.2) ---- */
static size_t write_nal(uint8_t *out, int ref_idc, int type, const uint8_t *rbsp, size_t n) {
    out[4] = (uint8_t)((ref_idc << 5) | type);
    size_t sz = 1 + orc_escape(rbsp, n, out + 5);
    out[0] = (uint8_t)(sz >> 24);
    out[1] = (uint8_t)(sz >> 16);
    out[2] = (uint8_t)(sz >> 8);
    out[3] = (uint8_t)sz;
    return sz + 4;
}
static size_t write_sps(const orc_enc *e, uint8_t *rbsp, size_t cap) {
    bw_t b;
    bw_init(&b, rbsp, cap);
    bw_put(&b, 66, 8);   /* profile_idc: Baseline */
    bw_put(&b, 0xC0, 8); /* constraint_set0+1: Constrained Baseline */
    bw_put(&b, (uint32_t)pick_level(e->mbw * e->mbh, e->p.fps_num, e->p.fps_den), 8);
    bw_ue(&b, 0); /* sps id */
    bw_ue(&b, 0); /* log2_max_frame_num_minus4 */
    bw_ue(&b, 2); /* pic_order_cnt_type */
    bw_ue(&b, 1); /* max_num_ref_frames */
    bw_put(&b, 0, 1);
    bw_ue(&b, (uint32_t)(e->mbw - 1));
    bw_ue(&b, (uint32_t)(e->mbh - 1));
    bw_put(&b, 1, 1); /* frame_mbs_only */
    bw_put(&b, 1, 1); /* direct_8x8_inference */
    int cr = (e->cw - e->p.width) / 2, cb = (e->ch - e->p.height) / 2;
    if (cr || cb) {
        bw_put(&b, 1, 1);
        bw_ue(&b, 0);
        bw_ue(&b, (uint32_t)cr);
        bw_ue(&b, 0);
        bw_ue(&b, (uint32_t)cb);
    } else
        bw_put(&b, 0, 1);
    bw_put(&b, 1, 1); /* vui_parameters_present */
    if (e->p.sar_w > 0 && e->p.sar_h > 0) {
        bw_put(&b, 1, 1);
        bw_put(&b, 255, 8);
        bw_put(&b, (uint32_t)e->p.sar_w, 16);
        bw_put(&b, (uint32_t)e->p.sar_h, 16);
    } else
        bw_put(&b, 0, 1);
    bw_put(&b, 0, 1); /* overscan */
    bw_put(&b, 0, 1); /* video_signal_type */
    bw_put(&b, 0, 1); /* chroma_loc */
    if (e->p.fps_num > 0 && e->p.fps_den > 0) {
        bw_put(&b, 1, 1);
        bw_put(&b, (uint32_t)e->p.fps_den, 32);
        bw_put(&b, (uint32_t)e->p.fps_num * 2u, 32);
        bw_put(&b, 1, 1);
    } else
        bw_put(&b, 0, 1);
    bw_put(&b, 0, 1); /* nal_hrd */
    bw_put(&b, 0, 1); /* vcl_hrd */
    bw_put(&b, 0, 1); /* pic_struct */
    bw_put(&b, 1, 1); /* bitstream_restriction */
    bw_put(&b, 1, 1); /* mv over pic boundaries */
    bw_ue(&b, 0);
    bw_ue(&b, 0);
    bw_ue(&b, 7);
    bw_ue(&b, 7);
    bw_ue(&b, 0); /* max_num_reorder_frames */
    bw_ue(&b, 1); /* max_dec_frame_buffering */
    bw_trailing(&b);
    return b.pos;
}
static size_t write_pps(const orc_enc *e, uint8_t *rbsp, size_t cap) {
    bw_t b;
    bw_init(&b, rbsp, cap);
    bw_ue(&b, 0);
    bw_ue(&b, 0);
    bw_put(&b, 0, 1); /* CAVLC */
    bw_put(&b, 0, 1);
    bw_ue(&b, 0); /* slice groups */
    bw_ue(&b, 0);
    bw_ue(&b, 0);
    bw_put(&b, 0, 1);
    bw_put(&b, 0, 2);
    bw_se(&b, e->p.qp - 26);
    bw_se(&b, 0);
    bw_se(&b, 0);     /* chroma_qp_index_offset */
    bw_put(&b, 1, 1); /* deblocking_filter_control_present */
    bw_put(&b, 0, 1); /* constrained_intra_pred */
    bw_put(&b, 0, 1); /* redundant_pic_cnt_present */
    bw_trailing(&b);
    return b.pos;
}
static void write_slice_header(const orc_enc *e, bw_t *b, int first_mb, int idr) {
    bw_ue(b, (uint32_t)first_mb);
    bw_ue(b, idr ? 7 : 5); /* slice_type: all slices of the picture I / P */
    bw_ue(b, 0);
    bw_put(b, (uint32_t)e->frame_num, 4);
    if (idr) bw_ue(b, (uint32_t)(e->idr_count & 1));
    if (!idr) {
        bw_put(b, 0, 1); /* num_ref_idx_active_override */
        bw_put(b, 0, 1); /* ref_pic_list_modification_flag_l0 */
    }
    if (idr) {
        bw_put(b, 0, 1);
        bw_put(b, 0, 1);
    } else
        bw_put(b, 0, 1); /* adaptive_ref_pic_marking_mode */
    bw_se(b, e->qp_cur - e->p.qp); /* slice_qp_delta: 0 at fixed QP */
    if (e->p.deblock) {
        bw_ue(b, 2); /* disable_deblocking_filter_idc: filter, but not across slice boundaries (spec/CODEC.md section 5a) */
        bw_se(b, 0); /* slice_alpha_c0_offset_div2 */
        bw_se(b, 0); /* slice_beta_offset_div2 */
    } else
        bw_ue(b, 1); /* disable_deblocking_filter_idc */
}

int orc_enc_set_band(orc_enc *e, int slice_first, int slice_count) {
    if (e->p.rc_mode && slice_count != e->p.n_slices) return -1; /* the rate control's SAD sum spans the picture */
    if (slice_first < 0 || slice_count < 1 || slice_first + slice_count > e->p.n_slices) return -1;
    e->band_s0 = slice_first;
    e->band_s1 = slice_first + slice_count;
    return 0;
}

static void repad(orc_enc *e) {
    pad_plane(e->refp[0], e->refp_stride[0], LPAD, e->rec[0], e->cw, e->ch);
    pad_plane(e->refp[1], e->refp_stride[1], CPAD, e->rec[1], e->cw / 2, e->ch / 2);
    pad_plane(e->refp[2], e->refp_stride[2], CPAD, e->rec[2], e->cw / 2, e->ch / 2);
}

int orc_enc_import_rows(orc_enc *e, int mb_row0, int mb_row1, const uint8_t *y, const uint8_t *u, const uint8_t *v) {
    if (mb_row0 < 0 || mb_row1 > e->mbh || mb_row0 >= mb_row1 || !y || !u || !v) return -1;
    const int n = mb_row1 - mb_row0;
    memcpy(e->rec[0] + (size_t)mb_row0 * 16 * e->cw, y, (size_t)n * 16 * e->cw);
    memcpy(e->rec[1] + (size_t)mb_row0 * 8 * (e->cw / 2), u, (size_t)n * 8 * (e->cw / 2));
    memcpy(e->rec[2] + (size_t)mb_row0 * 8 * (e->cw / 2), v, (size_t)n * 8 * (e->cw / 2));
    repad(e);
    return 0;
}

long orc_enc_encode(orc_enc *e, const uint8_t *y, int sy, const uint8_t *u, int su, const uint8_t *v, int sv,
                    uint8_t *out, size_t out_cap, orc_nal *nals, int *nal_count, int *is_idr) {
    int idr = (e->frame_idx % e->p.keyint) == 0;
    if (idr) e->frame_num = 0;
    ingest_plane(e->cur[0], e->cw, e->ch, y, sy, e->p.width, e->p.height);
    ingest_plane(e->cur[1], e->cw / 2, e->ch / 2, u, su, e->p.width / 2, e->p.height / 2);
    ingest_plane(e->cur[2], e->cw / 2, e->ch / 2, v, sv, e->p.width / 2, e->p.height / 2);

#ifdef _OPENMP
    omp_set_num_threads(e->p.threads);
#endif
    /* picture QP: fixed, or chosen from the previous picture's search results (spec/CODEC.md section 1a) */
    e->qp_cur = e->p.rc_mode ? orc_rc_qp(e->p.qp, idr, e->prev_was_p, e->prev_sad_sum, e->mbw * e->mbh) : e->p.qp;
    if (!idr) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int my = e->slice_rows[e->band_s0]; my < e->slice_rows[e->band_s1]; my++)
            for (int mx = 0; mx < e->mbw; mx++) me_mb(e, mx, my);
        uint64_t sum = 0; /* (the rate control reads the full search's SADs, before any refinement) */
        for (int i = 0; i < e->mbw * e->mbh; i++) sum += e->sad[i];
        e->prev_sad_sum = sum;
        if (e->p.subpel) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int my = e->slice_rows[e->band_s0]; my < e->slice_rows[e->band_s1]; my++)
                for (int mx = 0; mx < e->mbw; mx++) subpel_mb(e, mx, my);
        }
    }
    e->prev_was_p = !idr;
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = e->band_s0; s < e->band_s1; s++) {
        int r0 = e->slice_rows[s], r1 = e->slice_rows[s + 1];
        for (int my = r0; my < r1; my++)
            for (int mx = 0; mx < e->mbw; mx++) {
                if (idr)
                    code_mb_intra(e, mx, my, r0);
                else
                    code_mb_inter(e, mx, my);
            }
        bw_t b;
        bw_init(&b, e->slice_buf[s], e->slice_cap);
        write_slice_header(e, &b, r0 * e->mbw, idr);
        int skip_run = 0;
        for (int my = r0; my < r1; my++)
            for (int mx = 0; mx < e->mbw; mx++) {
                uint64_t before = b.total_bits;
                if (idr)
                    write_mb_intra(e, &b, mx, my, r0);
                else if (mb_is_skip(e, mx, my, r0)) {
                    skip_run++;
                    /* a slice that ends in skipped macroblocks carries their count as a last mb_skip_run; the bits
                     * are booked on the last macroblock of the slice */
                    if (my == r1 - 1 && mx == e->mbw - 1) bw_ue(&b, (uint32_t)skip_run);
                } else {
                    write_mb_inter(e, &b, mx, my, r0, skip_run);
                    skip_run = 0;
                }
                e->mb_bits[my * e->mbw + mx] = (uint32_t)(b.total_bits - before);
            }
        bw_trailing(&b);
        e->slice_len[s] = b.overflow ? (size_t)-1 : b.pos;
    }
    for (int s = e->band_s0; s < e->band_s1; s++)
        if (e->slice_len[s] == (size_t)-1) return -2;

    /* NAL assembly: contiguous, 4-byte length prefixes, parameter sets first
     * (consumer contract av_encode.c:670-751, :802-811) */
    size_t o = 0;
    int n = 0;
    if (idr && e->band_s0 == 0) {
        uint8_t rbsp[128];
        size_t k = write_sps(e, rbsp, sizeof(rbsp));
        if (o + k * 2 + 8 > out_cap) return -1;
        size_t sz = write_nal(out + o, 3, 7, rbsp, k);
        nals[n++] = (orc_nal){7, 3, o, sz};
        o += sz;
        k = write_pps(e, rbsp, sizeof(rbsp));
        if (o + k * 2 + 8 > out_cap) return -1;
        sz = write_nal(out + o, 3, 8, rbsp, k);
        nals[n++] = (orc_nal){8, 3, o, sz};
        o += sz;
    }
    for (int s = e->band_s0; s < e->band_s1; s++) {
        if (o + e->slice_len[s] * 3 / 2 + 8 > out_cap || n >= ORC_MAX_NALS) return -1;
        size_t sz = write_nal(out + o, idr ? 3 : 2, idr ? 5 : 1, e->slice_buf[s], e->slice_len[s]);
        nals[n++] = (orc_nal){idr ? 5 : 1, idr ? 3 : 2, o, sz};
        o += sz;
    }
    *nal_count = n;
    *is_idr = idr;

    if (e->p.deblock)
        for (int k = 0; k < 3; k++) {
            size_t n = (size_t)e->cw * e->ch / (k ? 4 : 1);
            if (!e->rec_pre[k]) e->rec_pre[k] = (uint8_t *)malloc(n);
            memcpy(e->rec_pre[k], e->rec[k], n);
        }
    if (e->p.deblock) { /* in-loop filter of the coded slices (8.7, orc_deblock.h): the picture a decoder outputs and predicts from */
        int nmb = e->mbw * e->mbh;
        uint8_t *intra = (uint8_t *)malloc((size_t)nmb * 3);
        int32_t *slice = (int32_t *)malloc((size_t)nmb * sizeof(int32_t));
        uint8_t *qp = intra + nmb, *en = intra + 2 * nmb;
        memset(intra, idr ? 1 : 0, (size_t)nmb);
        memset(qp, e->qp_cur, (size_t)nmb);
        memset(en, 0, (size_t)nmb);
        for (int s = 0; s < e->p.n_slices; s++)
            for (int mb = e->slice_rows[s] * e->mbw; mb < e->slice_rows[s + 1] * e->mbw; mb++) slice[mb] = s, en[mb] = (uint8_t)(s >= e->band_s0 && s < e->band_s1);
        const orc_dbk k = {{e->rec[0], e->rec[1], e->rec[2]}, {e->cw, e->cw / 2, e->cw / 2}, e->mbw, e->mbh, intra, e->nnz_l, e->mvq, slice, qp, en, orc_chroma_qp};
        orc_deblock_picture(&k);
        free(intra);
        free(slice);
    }
    /* reconstructed frame becomes the (edge-replicated) reference of the next */
    pad_plane(e->refp[0], e->refp_stride[0], LPAD, e->rec[0], e->cw, e->ch);
    pad_plane(e->refp[1], e->refp_stride[1], CPAD, e->rec[1], e->cw / 2, e->ch / 2);
    pad_plane(e->refp[2], e->refp_stride[2], CPAD, e->rec[2], e->cw / 2, e->ch / 2);
    e->frame_idx++;
    e->frame_num = (e->frame_num + 1) & 15;
    if (idr) e->idr_count++;
    return (long)o;
}
