/*
 * orc_dec.c -- minimal H.264 decoder for the subset the oracle encoder emits
 * (TEST INFRASTRUCTURE ONLY; see orc.h).
 *
 * Purpose: self-consistency.  decode(bitstream) must reproduce the encoder's
 * reconstructed frames bit for bit, which catches encoder/recon drift and
 * syntax-order mistakes.  It shares this repo's reading of ITU-T H.264, so it
 * does NOT establish conformance with third-party decoders ("conformance
 * verified by own decoder only").  It consumes exactly
 * what the reference's muxer consumes: 4-byte length-prefixed NAL units
 * (av_encode.c:635-640, :722-727).
 *
 * Written independently of orc_enc.c where practical: reference samples are
 * fetched with coordinate clamping (8.4.2.2.1) instead of a padded copy, VLCs
 * are decoded by table search, transforms follow the spec equations literally.
 */
#include "orc.h"
#include "orc_interp.h"
#include "orc_intra4.h"
#include "orc_deblock.h"
#include "orc_tables.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *buf;
    size_t nbits, pos;
    int err;
} br_t;

static uint32_t br_peek(br_t *b, int n) { /* n <= 24, zero padded past the end */
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        size_t p = b->pos + (size_t)i;
        int bit = p < b->nbits ? (b->buf[p >> 3] >> (7 - (p & 7))) & 1 : 0;
        v = (v << 1) | (uint32_t)bit;
    }
    return v;
}
static uint32_t br_get(br_t *b, int n) {
    uint32_t v = 0;
    while (n > 0) {
        int k = n > 16 ? 16 : n;
        v = (v << k) | br_peek(b, k);
        b->pos += (size_t)k;
        n -= k;
    }
    if (b->pos > b->nbits) b->err = 1;
    return v;
}
static uint32_t br_ue(br_t *b) {
    int z = 0;
    while (z < 32 && br_get(b, 1) == 0 && !b->err) z++;
    if (z >= 32) {
        b->err = 1;
        return 0;
    }
    return (1u << z) - 1u + (z ? br_get(b, z) : 0);
}
static int br_se(br_t *b) {
    uint32_t k = br_ue(b);
    return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}

struct orc_dec {
    int have_sps, have_pps;
    int mbw, mbh, cw, ch, crop_w, crop_h;
    int pic_init_qp, deblock_ctrl;
    int log2_max_frame_num, poc_type;
    uint8_t *cur[3], *ref[3];
    int8_t *mv;
    uint8_t *nnz_l, *nnz_cb, *nnz_cr;
    int *mb_slice; /* slice id of every MB of the current picture, -1 = not decoded */
    uint8_t *mb_i4;  /* [mb] Intra4x4 */
    int8_t *i4mode;  /* [mb][16] Intra4x4PredMode by luma4x4BlkIdx */
    uint8_t *mb_intra, *mb_qp, *mb_dbk; /* per MB: intra, QP_Y, in-loop filter on (disable_deblocking_filter_idc == 2) */
    int n_slices_seen;
    uint8_t *rbsp;
    size_t rbsp_cap;
    char err[256];
};

orc_dec *orc_dec_open(void) { return (orc_dec *)calloc(1, sizeof(orc_dec)); }
void orc_dec_close(orc_dec *d) {
    if (!d) return;
    for (int k = 0; k < 3; k++) {
        free(d->cur[k]);
        free(d->ref[k]);
    }
    free(d->mv);
    free(d->nnz_l);
    free(d->nnz_cb);
    free(d->nnz_cr);
    free(d->mb_slice);
    free(d->mb_intra);
    free(d->mb_i4);
    free(d->i4mode);
    free(d->mb_qp);
    free(d->mb_dbk);
    free(d->rbsp);
    free(d);
}
int orc_dec_width(const orc_dec *d) { return d->cw; }
int orc_dec_height(const orc_dec *d) { return d->ch; }
int orc_dec_crop_width(const orc_dec *d) { return d->crop_w; }
int orc_dec_crop_height(const orc_dec *d) { return d->crop_h; }
const uint8_t *orc_dec_plane(const orc_dec *d, int plane, int *stride) {
    if (stride) *stride = plane ? d->cw / 2 : d->cw;
    return d->cur[plane];
}
const char *orc_dec_error(const orc_dec *d) { return d->err; }

#define FAIL(d, ...)                                     \
    do {                                                 \
        snprintf((d)->err, sizeof((d)->err), __VA_ARGS__); \
        return -1;                                       \
    } while (0)

static int parse_sps(orc_dec *d, br_t *b) {
    int profile = (int)br_get(b, 8);
    br_get(b, 8);
    br_get(b, 8);
    br_ue(b);
    if (profile != 66) FAIL(d, "unsupported profile %d", profile);
    d->log2_max_frame_num = (int)br_ue(b) + 4;
    d->poc_type = (int)br_ue(b);
    if (d->poc_type != 2) FAIL(d, "unsupported poc type %d", d->poc_type);
    br_ue(b);     /* max_num_ref_frames */
    br_get(b, 1); /* gaps */
    int mbw = (int)br_ue(b) + 1, mbh = (int)br_ue(b) + 1;
    if (!br_get(b, 1)) FAIL(d, "interlace unsupported");
    br_get(b, 1);
    int cl = 0, cr = 0, ct = 0, cb = 0;
    if (br_get(b, 1)) {
        cl = (int)br_ue(b);
        cr = (int)br_ue(b);
        ct = (int)br_ue(b);
        cb = (int)br_ue(b);
    }
    if (b->err) FAIL(d, "sps truncated");
    if (!d->have_sps || mbw != d->mbw || mbh != d->mbh) {
        for (int k = 0; k < 3; k++) {
            free(d->cur[k]);
            free(d->ref[k]);
            size_t n = (size_t)(mbw * 16) * (mbh * 16) / (k ? 4 : 1);
            d->cur[k] = (uint8_t *)calloc(n, 1);
            d->ref[k] = (uint8_t *)calloc(n, 1);
        }
        free(d->mv);
        free(d->nnz_l);
        free(d->nnz_cb);
        free(d->nnz_cr);
        free(d->mb_slice);
        free(d->mb_intra);
    free(d->mb_i4);
    free(d->i4mode);
        free(d->mb_qp);
        free(d->mb_dbk);
        size_t nmb = (size_t)mbw * mbh;
        d->mv = (int8_t *)calloc(nmb * 2, 1);
        d->nnz_l = (uint8_t *)calloc(nmb * 16, 1);
        d->nnz_cb = (uint8_t *)calloc(nmb * 4, 1);
        d->nnz_cr = (uint8_t *)calloc(nmb * 4, 1);
        d->mb_slice = (int *)calloc(nmb, sizeof(int));
        d->mb_intra = (uint8_t *)calloc(nmb, 1);
        d->mb_i4 = (uint8_t *)calloc(nmb, 1);
        d->i4mode = (int8_t *)calloc(nmb * 16, 1);
        d->mb_qp = (uint8_t *)calloc(nmb, 1);
        d->mb_dbk = (uint8_t *)calloc(nmb, 1);
    }
    d->mbw = mbw;
    d->mbh = mbh;
    d->cw = mbw * 16;
    d->ch = mbh * 16;
    d->crop_w = d->cw - 2 * (cl + cr);
    d->crop_h = d->ch - 2 * (ct + cb);
    d->have_sps = 1;
    return 0;
}
static int parse_pps(orc_dec *d, br_t *b) {
    br_ue(b);
    br_ue(b);
    if (br_get(b, 1)) FAIL(d, "CABAC unsupported");
    br_get(b, 1);
    if (br_ue(b)) FAIL(d, "slice groups unsupported");
    if (br_ue(b)) FAIL(d, "num_ref_idx_l0_default != 1 unsupported");
    br_ue(b);
    if (br_get(b, 1)) FAIL(d, "weighted pred unsupported");
    br_get(b, 2);
    d->pic_init_qp = 26 + br_se(b);
    br_se(b);
    if (br_se(b)) FAIL(d, "chroma_qp_index_offset unsupported");
    d->deblock_ctrl = (int)br_get(b, 1);
    if (br_get(b, 1)) FAIL(d, "constrained_intra_pred unsupported");
    if (br_get(b, 1)) FAIL(d, "redundant_pic_cnt unsupported");
    if (b->err) FAIL(d, "pps truncated");
    d->have_pps = 1;
    return 0;
}

/* ---- CAVLC parsing by exhaustive table search ---- */
static int vlc_search(br_t *b, const uint8_t *len, const uint8_t *bits, int n) {
    uint32_t pk = br_peek(b, 16);
    for (int i = 0; i < n; i++)
        if (len[i] && (pk >> (16 - len[i])) == bits[i]) {
            b->pos += len[i];
            return i;
        }
    b->err = 1;
    return -1;
}
/* returns TotalCoeff; coef[] (scan order, max_coef entries) filled */
static int parse_block(br_t *b, int16_t *coef, int max_coef, int nC) {
    memset(coef, 0, sizeof(int16_t) * (size_t)max_coef);
    int idx;
    if (nC < 0)
        idx = vlc_search(b, orc_chroma_dc_coeff_token_len, orc_chroma_dc_coeff_token_bits, 20);
    else {
        int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
        idx = vlc_search(b, orc_coeff_token_len[tab], orc_coeff_token_bits[tab], 68);
    }
    if (idx < 0) return -1;
    int total = idx >> 2, t1 = idx & 3;
    if (total == 0) return 0;
    if (total > max_coef) {
        b->err = 1;
        return -1;
    }
    int level[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1) {
            level[i] = br_get(b, 1) ? -1 : 1;
            continue;
        }
        int prefix = 0;
        while (br_get(b, 1) == 0 && !b->err && prefix < 32) prefix++;
        if (prefix > 15) {
            b->err = 1;
            return -1;
        }
        int code = (prefix < 15 ? prefix : 15) << suffix_len; /* 9.2.2.1 */
        int ssize = (prefix == 14 && suffix_len == 0) ? 4 : prefix >= 15 ? prefix - 3 : suffix_len;
        if (ssize) code += (int)br_get(b, ssize);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros_left = 0;
    if (total < max_coef) {
        int tz;
        if (max_coef == 4)
            tz = vlc_search(b, orc_chroma_dc_total_zeros_len[total - 1], orc_chroma_dc_total_zeros_bits[total - 1], 4);
        else
            tz = vlc_search(b, orc_total_zeros_len[total - 1], orc_total_zeros_bits[total - 1], 16);
        if (tz < 0) return -1;
        zeros_left = tz;
    }
    int pos = total + zeros_left - 1; /* scan index of the highest-frequency coefficient */
    if (pos >= max_coef) {
        b->err = 1;
        return -1;
    }
    for (int i = 0; i < total; i++) {
        coef[pos] = (int16_t)level[i];
        int run = 0;
        if (i < total - 1 && zeros_left > 0) {
            int zl = zeros_left > 7 ? 7 : zeros_left;
            run = vlc_search(b, orc_run_len[zl - 1], orc_run_bits[zl - 1], 16);
            if (run < 0) return -1;
            zeros_left -= run;
            if (zeros_left < 0) {
                b->err = 1;
                return -1;
            }
        }
        pos -= 1 + run;
    }
    return total;
}

/* ---- reconstruction helpers following 8.5 literally ---- */
static int cls(int i, int j) { return (!(i & 1) && !(j & 1)) ? 0 : ((i & 1) && (j & 1)) ? 1 : 2; }
static int level_scale(int qp, int i, int j) { return 16 * orc_dequant_v[qp % 6][cls(i, j)]; }
static void inv_transform_add(uint8_t *dst, int stride, const uint8_t *pred, int pstride, int d[4][4]) {
    int f[4][4], h[4][4];
    for (int i = 0; i < 4; i++) {
        int e0 = d[i][0] + d[i][2], e1 = d[i][0] - d[i][2];
        int e2 = (d[i][1] >> 1) - d[i][3], e3 = d[i][1] + (d[i][3] >> 1);
        f[i][0] = e0 + e3;
        f[i][1] = e1 + e2;
        f[i][2] = e1 - e2;
        f[i][3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        int g0 = f[0][j] + f[2][j], g1 = f[0][j] - f[2][j];
        int g2 = (f[1][j] >> 1) - f[3][j], g3 = f[1][j] + (f[3][j] >> 1);
        h[0][j] = g0 + g3;
        h[1][j] = g1 + g2;
        h[2][j] = g1 - g2;
        h[3][j] = g0 - g3;
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int v = pred[i * pstride + j] + ((h[i][j] + 32) >> 6);
            dst[i * stride + j] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}
static void scale_block(const int16_t *coef /*scan order, 16*/, int qp, int skip_dc, int d[4][4]) {
    for (int k = skip_dc; k < 16; k++) {
        int r = orc_zigzag4x4[k], i = r >> 2, j = r & 3;
        int c = coef[k];
        if (qp >= 24)
            d[i][j] = (c * level_scale(qp, i, j)) << (qp / 6 - 4);
        else
            d[i][j] = (c * level_scale(qp, i, j) + (1 << (3 - qp / 6))) >> (4 - qp / 6);
    }
}

static int nc_of(int aA, int nA, int aB, int nB) { return aA && aB ? (nA + nB + 1) >> 1 : aA ? nA : aB ? nB : 0; }
static int mb_avail(const orc_dec *d, int mx, int my, int sid) {
    if (mx < 0 || my < 0 || mx >= d->mbw || my >= d->mbh) return 0;
    return d->mb_slice[my * d->mbw + mx] == sid;
}
static int luma_nc_d(const orc_dec *d, int mx, int my, int bx, int by, int sid) {
    int mb = my * d->mbw + mx, aA, aB, nA = 0, nB = 0;
    if (bx > 0)
        aA = 1, nA = d->nnz_l[mb * 16 + 4 * by + bx - 1];
    else if ((aA = mb_avail(d, mx - 1, my, sid)))
        nA = d->nnz_l[(mb - 1) * 16 + 4 * by + 3];
    if (by > 0)
        aB = 1, nB = d->nnz_l[mb * 16 + 4 * (by - 1) + bx];
    else if ((aB = mb_avail(d, mx, my - 1, sid)))
        nB = d->nnz_l[(mb - d->mbw) * 16 + 12 + bx];
    return nc_of(aA, nA, aB, nB);
}
static int chroma_nc_d(const orc_dec *d, const uint8_t *nnz, int mx, int my, int bx, int by, int sid) {
    int mb = my * d->mbw + mx, aA, aB, nA = 0, nB = 0;
    if (bx > 0)
        aA = 1, nA = nnz[mb * 4 + 2 * by];
    else if ((aA = mb_avail(d, mx - 1, my, sid)))
        nA = nnz[(mb - 1) * 4 + 2 * by + 1];
    if (by > 0)
        aB = 1, nB = nnz[mb * 4 + bx];
    else if ((aB = mb_avail(d, mx, my - 1, sid)))
        nB = nnz[(mb - d->mbw) * 4 + 2 + bx];
    return nc_of(aA, nA, aB, nB);
}

/* chroma residual parse + reconstruction; pred is an 8x8 array per plane */
static int decode_chroma(orc_dec *d, br_t *b, int mx, int my, int sid, int cbpc, int qp, uint8_t pred[2][64]) {
    int mb = my * d->mbw + mx, qpc = orc_chroma_qp[qp], cs = d->cw / 2;
    int16_t dc[2][4], ac[2][4][16];
    memset(dc, 0, sizeof(dc));
    memset(ac, 0, sizeof(ac));
    memset(d->nnz_cb + mb * 4, 0, 4);
    memset(d->nnz_cr + mb * 4, 0, 4);
    if (cbpc) {
        for (int p = 0; p < 2; p++)
            if (parse_block(b, dc[p], 4, -1) < 0) FAIL(d, "chroma dc parse, mb %d", mb);
    }
    if (cbpc == 2)
        for (int p = 0; p < 2; p++) {
            uint8_t *nnz = p ? d->nnz_cr : d->nnz_cb;
            for (int k = 0; k < 4; k++) {
                int n = parse_block(b, ac[p][k] + 1, 15, chroma_nc_d(d, nnz, mx, my, k & 1, k >> 1, sid));
                if (n < 0) FAIL(d, "chroma ac parse, mb %d", mb);
                nnz[mb * 4 + k] = (uint8_t)n;
            }
        }
    for (int p = 0; p < 2; p++) {
        int c0 = dc[p][0], c1 = dc[p][1], c2 = dc[p][2], c3 = dc[p][3];
        int f[4] = {c0 + c1 + c2 + c3, c0 - c1 + c2 - c3, c0 + c1 - c2 - c3, c0 - c1 - c2 + c3};
        uint8_t *dst = d->cur[p + 1] + (size_t)(my * 8) * cs + mx * 8;
        for (int k = 0; k < 4; k++) {
            int dd[4][4];
            memset(dd, 0, sizeof(dd));
            scale_block(ac[p][k], qpc, 1, dd);
            dd[0][0] = ((f[k] * level_scale(qpc, 0, 0)) << (qpc / 6)) >> 5;
            int bx = (k & 1) * 4, by = (k >> 1) * 4;
            inv_transform_add(dst + (size_t)by * cs + bx, cs, pred[p] + by * 8 + bx, 8, dd);
        }
    }
    return 0;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static int med3(int a, int b, int c) { return a > b ? (b > c ? b : a > c ? c : a) : (a > c ? a : b > c ? c : b); }

static int decode_slice(orc_dec *d, br_t *b, int nal_type, int ref_idc) {
    if (!d->have_sps || !d->have_pps) FAIL(d, "slice before parameter sets");
    int first_mb = (int)br_ue(b);
    int st = (int)br_ue(b) % 5;
    br_ue(b);
    br_get(b, d->log2_max_frame_num);
    int idr = nal_type == 5;
    if (idr) br_ue(b);
    if (st != 0 && st != 2) FAIL(d, "slice type %d unsupported", st);
    if (st == 0) {
        if (br_get(b, 1)) FAIL(d, "num_ref_idx override unsupported");
        if (br_get(b, 1)) FAIL(d, "ref list modification unsupported");
    }
    if (ref_idc) {
        if (idr) {
            br_get(b, 1);
            br_get(b, 1);
        } else if (br_get(b, 1))
            FAIL(d, "MMCO unsupported");
    }
    int qp = d->pic_init_qp + br_se(b);
    int dbk = 0; /* in-loop filter of this slice's macroblocks (8.7, orc_deblock.h): off (idc 1) or within the slice (idc 2) */
    if (d->deblock_ctrl) {
        int idc = (int)br_ue(b);
        if (idc != 1 && idc != 2) FAIL(d, "disable_deblocking_filter_idc %d unsupported (filtering across slice boundaries)", idc);
        if (idc == 2) {
            if (br_se(b) != 0 || br_se(b) != 0) FAIL(d, "deblocking filter offsets unsupported");
            dbk = 1;
        }
    } else
        FAIL(d, "deblocking without filter control: unsupported (would filter across slice boundaries)");
    if (b->err) FAIL(d, "slice header truncated");

    /* end of slice data: position of the rbsp_stop_one_bit */
    size_t end = b->nbits;
    while (end > 0 && !((b->buf[(end - 1) >> 3] >> (7 - ((end - 1) & 7))) & 1)) end--;
    if (end == 0) FAIL(d, "no stop bit");
    end--; /* index of the stop bit */

    if (first_mb == 0) { /* new picture */
        for (int k = 0; k < 3; k++) {
            uint8_t *t = d->ref[k];
            d->ref[k] = d->cur[k];
            d->cur[k] = t;
        }
        for (int i = 0; i < d->mbw * d->mbh; i++) d->mb_slice[i] = -1;
        d->n_slices_seen = 0;
    }
    int sid = d->n_slices_seen++;
    int mb = first_mb, nmb = d->mbw * d->mbh, cs = d->cw / 2;
    int pending_skips = 0, need_run = 1; /* slice_data() of 7.3.4, CAVLC: mb_skip_run, that many P_Skip macroblocks,
                                            then -- if more data follows -- one macroblock_layer() */
    for (;;) {
        int skip = 0;
        if (st == 0) {
            if (pending_skips > 0) {
                skip = 1;
                pending_skips--;
            } else if (need_run) {
                if (b->pos >= end) break;
                pending_skips = (int)br_ue(b);
                need_run = 0;
                if (pending_skips > 0) {
                    skip = 1;
                    pending_skips--;
                }
            }
            if (!skip) {
                if (b->pos >= end) break; /* the slice ended with a run of skipped macroblocks */
                need_run = 1;
            }
        } else if (b->pos >= end)
            break;
        if (mb >= nmb) FAIL(d, "too many macroblocks in slice");
        int mx = mb % d->mbw, my = mb / d->mbw;
        int t = skip ? 0 : (int)br_ue(b);
        d->mb_slice[mb] = sid;
        d->mb_dbk[mb] = (uint8_t)dbk;
        int aL = mb_avail(d, mx - 1, my, sid), aT = mb_avail(d, mx, my - 1, sid);
        if (st == 2 || t >= 5) {
            if (st == 0) FAIL(d, "intra MB in P slice unsupported");
            int cbpc;
            uint8_t *dst = d->cur[0] + (size_t)(my * 16) * d->cw + mx * 16;
            d->mb_i4[mb] = (uint8_t)(t == 0);
            memset(d->i4mode + (size_t)mb * 16, -1, 16);
            if (t == 0) {
                /* I_NxN (7.3.5, 7.3.5.1; Baseline: no transform_size_8x8_flag): Intra4x4 */
                int8_t *own = d->i4mode + (size_t)mb * 16;
                for (int k = 0; k < 16; k++) {
                    const int bx = orc_blk_x[k], by = orc_blk_y[k];
                    int mA, mB;
                    if (bx > 0) mA = own[orc_i4_idx(bx - 1, by)];
                    else if (aL) mA = d->mb_i4[mb - 1] ? d->i4mode[(size_t)(mb - 1) * 16 + orc_i4_idx(3, by)] : 2;
                    else mA = -1;
                    if (by > 0) mB = own[orc_i4_idx(bx, by - 1)];
                    else if (aT) mB = d->mb_i4[mb - d->mbw] ? d->i4mode[(size_t)(mb - d->mbw) * 16 + orc_i4_idx(bx, 3)] : 2;
                    else mB = -1;
                    const int pm4 = orc_i4_pred_mode(mA, mB);
                    if (br_get(b, 1))
                        own[k] = (int8_t)pm4;
                    else {
                        const int rem = (int)br_get(b, 3);
                        own[k] = (int8_t)(rem < pm4 ? rem : rem + 1);
                    }
                }
                if (br_ue(b) != 0) FAIL(d, "intra chroma pred mode unsupported");
                const int code = (int)br_ue(b);
                if (code > 47) FAIL(d, "bad coded_block_pattern code %d", code);
                const int cbp = orc_golomb_to_intra_cbp[code];
                cbpc = cbp >> 4;
                if (cbp) qp += br_se(b);
                int16_t cf[16][16];
                memset(cf, 0, sizeof(cf));
                memset(d->nnz_l + mb * 16, 0, 16);
                for (int q = 0; q < 4; q++)
                    if (cbp & (1 << q))
                        for (int k = 4 * q; k < 4 * q + 4; k++) {
                            const int bx = orc_blk_x[k], by = orc_blk_y[k];
                            const int n = parse_block(b, cf[k], 16, luma_nc_d(d, mx, my, bx, by, sid));
                            if (n < 0) FAIL(d, "luma parse mb %d blk %d", mb, k);
                            d->nnz_l[mb * 16 + 4 * by + bx] = (uint8_t)n;
                        }
                const int aTL = mb_avail(d, mx - 1, my - 1, sid), aTR = mb_avail(d, mx + 1, my - 1, sid);
                for (int k = 0; k < 16; k++) {
                    const int bx = orc_blk_x[k], by = orc_blk_y[k];
                    orc_i4_nb nb;
                    orc_i4_neighbours(dst, d->cw, bx, by, aL, aT, aTL, aTR, &nb);
                    if (own[k] < 0 || own[k] > 8 || !orc_i4_mode_ok(&nb, own[k])) FAIL(d, "Intra4x4PredMode %d without its neighbours (mb %d blk %d)", own[k], mb, k);
                    uint8_t pr[16];
                    orc_i4_predict(&nb, own[k], pr);
                    int dd[4][4];
                    scale_block(cf[k], qp, 0, dd);
                    inv_transform_add(dst + (size_t)(4 * by) * d->cw + 4 * bx, d->cw, pr, 4, dd);
                }
            } else {
                if (t < 1 || t > 24) FAIL(d, "mb_type %d unsupported in I slice", t);
                int pm = (t - 1) & 3, acf = (t - 1) >= 12;
                cbpc = ((t - 1) >> 2) % 3;
                if (br_ue(b) != 0) FAIL(d, "intra chroma pred mode unsupported");
                qp += br_se(b);
                int16_t dcl[16], ac[16][16];
                memset(ac, 0, sizeof(ac));
                if (parse_block(b, dcl, 16, luma_nc_d(d, mx, my, 0, 0, sid)) < 0) FAIL(d, "luma dc parse mb %d", mb);
                memset(d->nnz_l + mb * 16, 0, 16);
                if (acf)
                    for (int k = 0; k < 16; k++) {
                        int bx = orc_blk_x[k], by = orc_blk_y[k];
                        int n = parse_block(b, ac[k] + 1, 15, luma_nc_d(d, mx, my, bx, by, sid));
                        if (n < 0) FAIL(d, "luma ac parse mb %d blk %d", mb, k);
                        d->nnz_l[mb * 16 + 4 * by + bx] = (uint8_t)n;
                    }
                /* Intra16x16 DC prediction */
                int sum = 0, dcp;
                if (aT)
                    for (int x = 0; x < 16; x++) sum += dst[-(ptrdiff_t)d->cw + x];
                if (aL)
                    for (int y = 0; y < 16; y++) sum += dst[(size_t)y * d->cw - 1];
                dcp = aT && aL ? (sum + 16) >> 5 : (aT || aL) ? (sum + 8) >> 4 : 128;
                /* the 16x16 predictor (8.3.3.1-8.3.3.3): vertical = the row above, horizontal = the column to the left, DC */
                uint8_t pred16[256];
                if (pm == 0) {
                    if (!aT) FAIL(d, "Intra16x16 vertical prediction without a macroblock above (mb %d)", mb);
                    for (int y = 0; y < 16; y++) memcpy(pred16 + 16 * y, dst - (ptrdiff_t)d->cw, 16);
                } else if (pm == 1) {
                    if (!aL) FAIL(d, "Intra16x16 horizontal prediction without a macroblock to the left (mb %d)", mb);
                    for (int y = 0; y < 16; y++) memset(pred16 + 16 * y, dst[(size_t)y * d->cw - 1], 16);
                } else if (pm == 3) {
                    /* plane (8.3.3.4): p[x, -1], p[-1, y] and the corner p[-1, -1] as arrays indexed from -1 */
                    if (!aT || !aL || !mb_avail(d, mx - 1, my - 1, sid)) FAIL(d, "Intra16x16 plane prediction without its neighbours (mb %d)", mb);
                    int tp[17], lf[17];
                    for (int i = -1; i < 16; i++) {
                        tp[i + 1] = dst[-(ptrdiff_t)d->cw + i];
                        lf[i + 1] = i < 0 ? dst[-(ptrdiff_t)d->cw - 1] : dst[(size_t)i * d->cw - 1];
                    }
                    int hh = 0, vv = 0;
                    for (int k = 0; k <= 7; k++) {
                        hh += (k + 1) * (tp[1 + 8 + k] - tp[1 + 6 - k]);
                        vv += (k + 1) * (lf[1 + 8 + k] - lf[1 + 6 - k]);
                    }
                    const int pa = 16 * (lf[16] + tp[16]), pb = (5 * hh + 32) >> 6, pc = (5 * vv + 32) >> 6;
                    for (int y = 0; y < 16; y++)
                        for (int x = 0; x < 16; x++) {
                            const int v = (pa + pb * (x - 7) + pc * (y - 7) + 16) >> 5;
                            pred16[16 * y + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                        }
                } else
                    memset(pred16, dcp, 256);
                /* luma DC: inverse zig-zag, 4x4 Hadamard, scaling (8.5.10) */
                int c[4][4], f[4][4], tt[4][4];
                for (int k = 0; k < 16; k++) c[orc_zigzag4x4[k] >> 2][orc_zigzag4x4[k] & 3] = dcl[k];
                static const int A[4][4] = {{1, 1, 1, 1}, {1, 1, -1, -1}, {1, -1, -1, 1}, {1, -1, 1, -1}};
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++) {
                        tt[i][j] = 0;
                        for (int k = 0; k < 4; k++) tt[i][j] += A[i][k] * c[k][j];
                    }
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++) {
                        f[i][j] = 0;
                        for (int k = 0; k < 4; k++) f[i][j] += tt[i][k] * A[k][j];
                    }
                for (int k = 0; k < 16; k++) {
                    int bx = orc_blk_x[k], by = orc_blk_y[k];
                    int dd[4][4];
                    memset(dd, 0, sizeof(dd));
                    scale_block(ac[k], qp, 1, dd);
                    int ls = level_scale(qp, 0, 0);
                    dd[0][0] = qp >= 36 ? (f[by][bx] * ls) << (qp / 6 - 6) : (f[by][bx] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
                    uint8_t pr[16];
                    for (int y = 0; y < 4; y++) memcpy(pr + 4 * y, pred16 + 16 * (4 * by + y) + 4 * bx, 4);
                    inv_transform_add(dst + (size_t)(4 * by) * d->cw + 4 * bx, d->cw, pr, 4, dd);
                }
            }
            /* chroma DC prediction */
            uint8_t pred[2][64];
            for (int p = 0; p < 2; p++) {
                const uint8_t *rc = d->cur[p + 1] + (size_t)(my * 8) * cs + mx * 8;
                for (int k = 0; k < 4; k++) {
                    int bx = (k & 1) * 4, by = (k >> 1) * 4, s0 = 0, s1 = 0, v;
                    if (aT)
                        for (int x = 0; x < 4; x++) s0 += rc[-(ptrdiff_t)cs + bx + x];
                    if (aL)
                        for (int y = 0; y < 4; y++) s1 += rc[(size_t)(by + y) * cs - 1];
                    int useT = aT, useL = aL;
                    if (k == 1 && aT) useL = 0;
                    if (k == 2 && aL) useT = 0;
                    if (useT && useL)
                        v = (s0 + s1 + 4) >> 3;
                    else if (useT)
                        v = (s0 + 2) >> 2;
                    else if (useL)
                        v = (s1 + 2) >> 2;
                    else
                        v = 128;
                    for (int y = 0; y < 4; y++) memset(pred[p] + (by + y) * 8 + bx, v, 4);
                }
            }
            d->mv[2 * mb] = d->mv[2 * mb + 1] = 0;
            if (decode_chroma(d, b, mx, my, sid, cbpc, qp, pred) < 0) return -1;
        } else {
            if (t != 0) FAIL(d, "P mb_type %d unsupported", t);
            /* mvp (8.4.1.3.1) */
            int aA = aL, aB = aT, aC = mb_avail(d, mx + 1, my - 1, sid), aD = mb_avail(d, mx - 1, my - 1, sid);
            int ax = 0, ay = 0, bx_ = 0, by_ = 0, cx = 0, cy = 0;
            if (aA) ax = d->mv[2 * (mb - 1)], ay = d->mv[2 * (mb - 1) + 1];
            if (aB) bx_ = d->mv[2 * (mb - d->mbw)], by_ = d->mv[2 * (mb - d->mbw) + 1];
            if (aC)
                cx = d->mv[2 * (mb - d->mbw + 1)], cy = d->mv[2 * (mb - d->mbw + 1) + 1];
            else if (aD)
                aC = 1, cx = d->mv[2 * (mb - d->mbw - 1)], cy = d->mv[2 * (mb - d->mbw - 1) + 1];
            int px, py;
            if (!aB && !aC && aA)
                px = ax, py = ay;
            else if (aA + aB + aC == 1)
                px = aA ? ax : aB ? bx_ : cx, py = aA ? ay : aB ? by_ : cy;
            else
                px = med3(ax, bx_, cx), py = med3(ay, by_, cy);
            int mvx, mvy, cbp = 0;
            if (skip) {
                /* P_Skip (8.4.1.1): the vector is zero when A or B is missing or has a zero vector (every coded
                 * neighbour here is a P macroblock with refIdx 0), else the 16x16 median prediction; no residual */
                if (!aA || !aB || (ax == 0 && ay == 0) || (bx_ == 0 && by_ == 0)) px = py = 0;
                mvx = px, mvy = py;
            } else {
                mvx = br_se(b) + px, mvy = br_se(b) + py; /* quarter-sample units throughout */
                if (mvx < -127 || mvx > 127 || mvy < -127 || mvy > 127) FAIL(d, "vector out of this decoder's range");
            }
            d->mv[2 * mb] = (int8_t)mvx;
            d->mv[2 * mb + 1] = (int8_t)mvy;
            if (!skip) {
                int code = (int)br_ue(b);
                if (code > 47) FAIL(d, "bad cbp code");
                cbp = orc_golomb_to_inter_cbp[code];
                if (cbp) qp += br_se(b);
            }
            /* prediction with coordinate clamping */
            uint8_t predl[256], pred[2][64];
            {   /* luma: 8.4.2.2.1, integer, half- and quarter-sample positions (orc_interp.h) */
                const orc_plane rp = {d->ref[0], d->cw, d->cw, d->ch};
                orc_luma_pred_block(&rp, 4 * mx * 16 + mvx, 4 * my * 16 + mvy, 16, 16, predl, 16);
            }
            int xf = mvx & 7, yf = mvy & 7;
            for (int p = 0; p < 2; p++)
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) {
                        int x0 = clampi(mx * 8 + x + (mvx >> 3), 0, cs - 1), x1 = clampi(mx * 8 + x + (mvx >> 3) + 1, 0, cs - 1);
                        int y0 = clampi(my * 8 + y + (mvy >> 3), 0, d->ch / 2 - 1), y1 = clampi(my * 8 + y + (mvy >> 3) + 1, 0, d->ch / 2 - 1);
                        const uint8_t *r = d->ref[p + 1];
                        int A_ = r[(size_t)y0 * cs + x0], B_ = r[(size_t)y0 * cs + x1], C_ = r[(size_t)y1 * cs + x0], D_ = r[(size_t)y1 * cs + x1];
                        pred[p][8 * y + x] = (uint8_t)(((8 - xf) * (8 - yf) * A_ + xf * (8 - yf) * B_ + (8 - xf) * yf * C_ + xf * yf * D_ + 32) >> 6);
                    }
            memset(d->nnz_l + mb * 16, 0, 16);
            uint8_t *dst = d->cur[0] + (size_t)(my * 16) * d->cw + mx * 16;
            for (int k = 0; k < 16; k++) {
                int bx = orc_blk_x[k], by = orc_blk_y[k];
                int16_t cf[16];
                memset(cf, 0, sizeof(cf));
                if (cbp & (1 << (k >> 2))) {
                    int n = parse_block(b, cf, 16, luma_nc_d(d, mx, my, bx, by, sid));
                    if (n < 0) FAIL(d, "luma parse mb %d blk %d", mb, k);
                    d->nnz_l[mb * 16 + 4 * by + bx] = (uint8_t)n;
                }
                int dd[4][4];
                memset(dd, 0, sizeof(dd));
                scale_block(cf, qp, 0, dd);
                inv_transform_add(dst + (size_t)(4 * by) * d->cw + 4 * bx, d->cw, predl + 16 * 4 * by + 4 * bx, 16, dd);
            }
            if (decode_chroma(d, b, mx, my, sid, cbp >> 4, qp, pred) < 0) return -1;
        }
        if (b->err) FAIL(d, "bitstream error in mb %d", mb);
        d->mb_intra[mb] = (uint8_t)(st == 2 || t >= 5);
        d->mb_qp[mb] = (uint8_t)qp;
        mb++;
    }
    if (b->pos != end) FAIL(d, "slice data overran the stop bit (pos %zu end %zu)", b->pos, end);
    return 0;
}

int orc_dec_decode(orc_dec *d, const uint8_t *data, size_t size) {
    size_t o = 0;
    int got_pic = 0;
    d->err[0] = 0;
    while (o + 4 <= size) {
        size_t n = ((size_t)data[o] << 24) | ((size_t)data[o + 1] << 16) | ((size_t)data[o + 2] << 8) | data[o + 3];
        o += 4;
        if (n < 1 || o + n > size) FAIL(d, "bad NAL length %zu", n);
        int hdr = data[o], type = hdr & 31, ref_idc = (hdr >> 5) & 3;
        if (hdr & 0x80) FAIL(d, "forbidden_zero_bit set");
        if (d->rbsp_cap < n) {
            free(d->rbsp);
            d->rbsp = (uint8_t *)malloc(n);
            d->rbsp_cap = n;
        }
        size_t r = 0;
        int z = 0;
        for (size_t i = 1; i < n; i++) { /* strip emulation prevention */
            uint8_t c = data[o + i];
            if (z >= 2 && c == 3) {
                z = 0;
                continue;
            }
            if (z >= 2 && c < 3) FAIL(d, "start code emulation inside NAL");
            d->rbsp[r++] = c;
            z = c ? 0 : z + 1;
        }
        br_t b = {d->rbsp, r * 8, 0, 0};
        int rc = 0;
        if (type == 7)
            rc = parse_sps(d, &b);
        else if (type == 8)
            rc = parse_pps(d, &b);
        else if (type == 1 || type == 5) {
            rc = decode_slice(d, &b, type, ref_idc);
            got_pic = 1;
        }
        if (rc < 0) return -1;
        o += n;
    }
    if (o != size) FAIL(d, "trailing bytes");
    if (got_pic) {
        for (int i = 0; i < d->mbw * d->mbh; i++)
            if (d->mb_slice[i] < 0) FAIL(d, "macroblock %d not covered by any slice", i);
        /* the picture is complete: in-loop filter (8.7) of the slices that ask for it; the filtered picture is what is
         * output and what the next picture predicts from */
        const orc_dbk k = {{d->cur[0], d->cur[1], d->cur[2]}, {d->cw, d->cw / 2, d->cw / 2}, d->mbw, d->mbh, d->mb_intra, d->nnz_l, d->mv,
                           d->mb_slice, d->mb_qp, d->mb_dbk, orc_chroma_qp};
        orc_deblock_picture(&k);
    }
    return got_pic;
}
