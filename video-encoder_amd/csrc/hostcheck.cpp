/*
 * hostcheck.cpp -- g++ build of the per-lane device functions (venc_device.h)
 * and the synthetic source for CPU-only tests.  TEST ARTEFACT: nothing in the
 * product loads this library; it exists so the lane-level logic that runs on
 * the GPU can be checked against the oracle in a container without a GPU.
 */
#include <string.h>

#include "venc_device.h"
#include "venc_lines.h"
#include "venc_synth.h"

extern "C" {

void hc_fill_qparams(venc_qparams *q, int qp) {
    q->per = qp / 6;
    q->qbits = 15 + qp / 6;
    for (int k = 0; k < 3; k++) {
        q->mf[k] = venc_mf[(qp % 6) * 3 + k];
        q->v[k] = venc_v[(qp % 6) * 3 + k];
    }
    q->f_intra = (1 << q->qbits) / 3;
    q->f_inter = (1 << q->qbits) / 6;
    for (int k = 0; k < 3; k++) q->zt_inter[k] = ((1 << q->qbits) - q->f_inter - 1) / q->mf[k];
}
/* the zero-block test of the transform kernels (venc_inter2.hip blk_all_zero) restated for the host: 1 iff every
 * coefficient from index `first` on quantises to level 0 */
int hc_block_all_zero(const int *w, int qp, int first) {
    venc_qparams q;
    hc_fill_qparams(&q, qp);
    unsigned m[3] = {0, 0, 0};
    for (int r = first; r < 16; r++) {
        const int cls = venc_pos_class(r);
        const unsigned u = (unsigned)(w[r] + q.zt_inter[cls]);
        if (u > m[cls]) m[cls] = u;
    }
    return m[0] <= 2u * (unsigned)q.zt_inter[0] && m[1] <= 2u * (unsigned)q.zt_inter[1] && m[2] <= 2u * (unsigned)q.zt_inter[2];
}
int hc_chroma_qp(int qp) { return qp < 30 ? qp : venc_qpc_hi[qp - 30]; }
int hc_rc_qp(int base, int is_idr, int have_history, unsigned long long sum, int nmb) { return venc_rc_qp(base, is_idr, have_history, sum, nmb); }
/* 1 iff the device-side construction of a QP's quantiser constants equals the host's */
int hc_make_qparams_matches(int qp) {
    venc_qparams a, b;
    memset(&a, 0, sizeof(a));
    memset(&b, 0, sizeof(b));
    hc_fill_qparams(&a, qp);
    venc_make_qparams(qp, &b);
    return memcmp(&a, &b, sizeof(a)) == 0 && venc_chroma_qp(qp) == hc_chroma_qp(qp);
}

/* round 3: the in-loop filter's line functions and the follow-up's entry lookup (venc_lines.h) */
void hc_dbk_line(int *s, int bs, int alpha, int beta, int tc0, int chroma) { venc_dbk_line(s, bs, alpha, beta, tc0, chroma != 0); }
void hc_dbk_line_normal(int *s, int bs, int alpha, int beta, int tc0, int chroma) { /* s = p3 .. q3 as venc_dbk_line takes them */
    venc_dbk_line_normal(chroma != 0, s[1], &s[2], &s[3], &s[4], &s[5], s[6], bs, alpha, beta, tc0);
}
int hc_nz_locate(const uint32_t *part, const uint32_t *cnt32, int rpt, uint32_t k, uint32_t *index) { return venc_nz_locate(part, cnt32, rpt, k, index); }

void hc_fdct4(const int *x, int *w) { venc_fdct4(x, w); }
void hc_idct4(const int *d, int *r) { venc_idct4(d, r); }
int hc_quant(int w, int qp, int r, int intra) {
    venc_qparams q;
    hc_fill_qparams(&q, qp);
    return venc_quant(w, &q, r, intra);
}
int hc_dequant(int lv, int qp, int r) {
    venc_qparams q;
    hc_fill_qparams(&q, qp);
    return venc_dequant(lv, &q, r);
}
int hc_zigzag(int i) { return venc_zigzag_ct(i); }

/* CAVLC block into a byte buffer (MSB first); returns bit count */
int hc_cavlc_block(const int *c, int first, int max_coef, int nC, unsigned char *out, int cap_bytes) {
    uint32_t words[32];
    venc_bitsink b;
    venc_bs_init(&b, words);
    venc_cavlc_block(&b, c, first, max_coef, nC);
    int bits = venc_bs_flush(&b);
    int nb = (bits + 7) / 8;
    if (nb > cap_bytes) return -1;
    for (int i = 0; i < nb; i++) out[i] = (unsigned char)(words[i >> 2] >> (24 - 8 * (i & 3)));
    return bits;
}
/* data-driven variant used by k_cavlc: levels in memory, tables concatenated as the kernel stages them */
int hc_cavlc_block_mem(const short *c16, int first, int max_coef, int nC, unsigned char *out, int cap_bytes) {
    static uint16_t tab[VENC_TAB_SIZE];
    static int init = 0;
    if (!init) {
        memcpy(tab + VENC_TAB_TOKEN, venc_vlc_coeff_token, 272 * 2);
        memcpy(tab + VENC_TAB_TOKEN_CDC, venc_vlc_coeff_token_cdc, 20 * 2);
        memcpy(tab + VENC_TAB_TZ, venc_vlc_total_zeros, 240 * 2);
        memcpy(tab + VENC_TAB_TZ_CDC, venc_vlc_total_zeros_cdc, 12 * 2);
        memcpy(tab + VENC_TAB_RUN, venc_vlc_run_before, 112 * 2);
        init = 1;
    }
    uint32_t words[32];
    venc_bitsink b;
    venc_bs_init(&b, words);
    venc_cavlc_block_mem(&b, c16, venc_nzmask_mem(c16, first, max_coef), first, max_coef, nC, tab);
    int bits = venc_bs_flush(&b);
    int nb = (bits + 7) / 8;
    if (nb > cap_bytes) return -1;
    for (int i = 0; i < nb; i++) out[i] = (unsigned char)(words[i >> 2] >> (24 - 8 * (i & 3)));
    return bits;
}
/* round 4: the sink with four words in registers (venc_bitsink4, the P pictures' entries) must produce the bits of the plain sink:
 * words[0..3] = its head registers, words[4..] = what it spilled; returns the bit count, or -1 - plain count on a mismatch of counts */
int hc_cavlc_block_mem4(const short *c16, int first, int max_coef, int nC, unsigned prefix_code, int prefix_len, unsigned *words) {
    static uint16_t tab[VENC_TAB_SIZE];
    memcpy(tab + VENC_TAB_TOKEN, venc_vlc_coeff_token, 272 * 2);
    memcpy(tab + VENC_TAB_TOKEN_CDC, venc_vlc_coeff_token_cdc, 20 * 2);
    memcpy(tab + VENC_TAB_TZ, venc_vlc_total_zeros, 240 * 2);
    memcpy(tab + VENC_TAB_TZ_CDC, venc_vlc_total_zeros_cdc, 12 * 2);
    memcpy(tab + VENC_TAB_RUN, venc_vlc_run_before, 112 * 2);
    uint32_t spill[32];
    memset(spill, 0, sizeof(spill));
    venc_bitsink4 b;
    venc_bs4_init(&b, spill);
    if (prefix_len) venc_bs_put(&b, prefix_code, prefix_len);
    venc_bs_ue(&b, 5u);
    venc_bs_se(&b, -3);
    venc_cavlc_block_mem(&b, c16, venc_nzmask_mem(c16, first, max_coef), first, max_coef, nC, tab);
    const int bits = venc_bs_flush(&b);
    uint32_t ref[32];
    memset(ref, 0, sizeof(ref));
    venc_bitsink r;
    venc_bs_init(&r, ref);
    if (prefix_len) venc_bs_put(&r, prefix_code, prefix_len);
    venc_bs_ue(&r, 5u);
    venc_bs_se(&r, -3);
    venc_cavlc_block_mem(&r, c16, venc_nzmask_mem(c16, first, max_coef), first, max_coef, nC, tab);
    const int rbits = venc_bs_flush(&r);
    if (bits != rbits) return -1 - rbits;
    words[0] = b.h0, words[1] = b.h1, words[2] = b.h2, words[3] = b.h3;
    for (int i = 4; i < 32; i++) words[i] = spill[i];
    for (int i = 0; i < 32; i++) words[32 + i] = ref[i];
    return bits;
}
int hc_ue_bits(unsigned v, unsigned char *out) {
    uint32_t words[4];
    venc_bitsink b;
    venc_bs_init(&b, words);
    venc_bs_ue(&b, v);
    int bits = venc_bs_flush(&b);
    for (int i = 0; i < (bits + 7) / 8; i++) out[i] = (unsigned char)(words[i >> 2] >> (24 - 8 * (i & 3)));
    return bits;
}
int hc_se_bits(int v, unsigned char *out) {
    uint32_t words[4];
    venc_bitsink b;
    venc_bs_init(&b, words);
    venc_bs_se(&b, v);
    int bits = venc_bs_flush(&b);
    for (int i = 0; i < (bits + 7) / 8; i++) out[i] = (unsigned char)(words[i >> 2] >> (24 - 8 * (i & 3)));
    return bits;
}
/* raw packed VLC tables for the structural test: which 0 coeff_token(272), 1 cdc(20),
 * 2 total_zeros(240), 3 tz cdc(12), 4 run_before(112) */
int hc_table(int which, unsigned short *out) {
    const uint16_t *t;
    int n;
    switch (which) {
        case 0: t = venc_vlc_coeff_token, n = 272; break;
        case 1: t = venc_vlc_coeff_token_cdc, n = 20; break;
        case 2: t = venc_vlc_total_zeros, n = 240; break;
        case 3: t = venc_vlc_total_zeros_cdc, n = 12; break;
        case 4: t = venc_vlc_run_before, n = 112; break;
        default: return -1;
    }
    memcpy(out, t, n * sizeof(uint16_t));
    return n;
}
int hc_cbp_code(int cbp) { return venc_cbp_inter_code[cbp]; }
unsigned char hc_synth_pixel(int plane, int x, int y, int t, int w, int h, unsigned seed) {
    return venc_synth_pixel(plane, x, y, t, w, h, seed);
}
}
