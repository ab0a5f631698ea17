/*
 * h264_syntax.c -- see h264_syntax.h.  Plain C99, no dependencies.
 */
#include "h264_syntax.h"

#include <string.h>

/* ---- MSB-first bit packer over a byte buffer ---- */
typedef struct {
    uint8_t *p;
    size_t cap, n;
    uint32_t hold;
    int held;
    int bad;
} bitpack;

static void bp_bits(bitpack *b, uint32_t v, int n) {
    for (int i = n - 1; i >= 0; i--) {
        b->hold = (b->hold << 1) | ((v >> i) & 1u);
        if (++b->held == 8) {
            if (b->n < b->cap)
                b->p[b->n++] = (uint8_t)b->hold;
            else
                b->bad = 1;
            b->hold = 0;
            b->held = 0;
        }
    }
}
static void bp_ue(bitpack *b, uint32_t v) {
    uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    bp_bits(b, 0, len);
    bp_bits(b, x, len + 1);
}
static void bp_se(bitpack *b, int v) { bp_ue(b, v > 0 ? 2u * (uint32_t)v - 1u : 2u * (uint32_t)(-v)); }
static size_t bp_finish(bitpack *b) { /* rbsp_trailing_bits */
    bp_bits(b, 1, 1);
    while (b->held) bp_bits(b, 0, 1);
    return b->bad ? 0 : b->n;
}

int h264_level_idc(int mbs, int fps_num, int fps_den) {
    /* Table A-1: level, MaxFS (macroblocks), MaxMBPS */
    static const int lim[][3] = {{10, 99, 1485},      {11, 396, 3000},     {12, 396, 6000},     {13, 396, 11880},
                                 {20, 396, 11880},    {21, 792, 19800},    {22, 1620, 20250},   {30, 1620, 40500},
                                 {31, 3600, 108000},  {32, 5120, 216000},  {40, 8192, 245760},  {42, 8704, 522240},
                                 {50, 22080, 589824}, {51, 36864, 983040}, {52, 36864, 2073600}};
    long rate = 0;
    if (fps_num > 0 && fps_den > 0) rate = ((long)mbs * fps_num + fps_den - 1) / fps_den;
    for (size_t i = 0; i < sizeof(lim) / sizeof(lim[0]); i++)
        if (mbs <= lim[i][1] && rate <= lim[i][2]) return lim[i][0];
    return 52;
}

size_t h264_write_sps(const h264_seq *s, uint8_t *rbsp, size_t cap) {
    const int mbw = (s->width + 15) / 16, mbh = (s->height + 15) / 16;
    bitpack b = {rbsp, cap, 0, 0, 0, 0};
    bp_bits(&b, 66, 8);   /* profile_idc = Baseline; byte 5 of the NAL as read at av_encode.c:703 */
    bp_bits(&b, 0xC0, 8); /* constraint_set0_flag | constraint_set1_flag */
    bp_bits(&b, (uint32_t)h264_level_idc(mbw * mbh, s->fps_num, s->fps_den), 8);
    bp_ue(&b, 0);         /* seq_parameter_set_id */
    bp_ue(&b, 0);         /* log2_max_frame_num_minus4 */
    bp_ue(&b, 2);         /* pic_order_cnt_type: output order = decode order */
    bp_ue(&b, 1);         /* max_num_ref_frames */
    bp_bits(&b, 0, 1);    /* gaps_in_frame_num_value_allowed_flag */
    bp_ue(&b, (uint32_t)(mbw - 1));
    bp_ue(&b, (uint32_t)(mbh - 1));
    bp_bits(&b, 1, 1);    /* frame_mbs_only_flag */
    bp_bits(&b, 1, 1);    /* direct_8x8_inference_flag */
    const int crop_r = (mbw * 16 - s->width) / 2, crop_b = (mbh * 16 - s->height) / 2;
    if (crop_r || crop_b) {
        bp_bits(&b, 1, 1);
        bp_ue(&b, 0);
        bp_ue(&b, (uint32_t)crop_r);
        bp_ue(&b, 0);
        bp_ue(&b, (uint32_t)crop_b);
    } else {
        bp_bits(&b, 0, 1);
    }
    bp_bits(&b, 1, 1); /* vui_parameters_present_flag */
    if (s->sar_w > 0 && s->sar_h > 0) {
        bp_bits(&b, 1, 1);
        bp_bits(&b, 255, 8); /* Extended_SAR */
        bp_bits(&b, (uint32_t)s->sar_w, 16);
        bp_bits(&b, (uint32_t)s->sar_h, 16);
    } else {
        bp_bits(&b, 0, 1);
    }
    bp_bits(&b, 0, 3); /* overscan_info, video_signal_type, chroma_loc_info: absent */
    if (s->fps_num > 0 && s->fps_den > 0) {
        bp_bits(&b, 1, 1);
        bp_bits(&b, (uint32_t)s->fps_den, 32);      /* num_units_in_tick */
        bp_bits(&b, 2u * (uint32_t)s->fps_num, 32); /* time_scale */
        bp_bits(&b, 1, 1);                          /* fixed_frame_rate_flag */
    } else {
        bp_bits(&b, 0, 1);
    }
    bp_bits(&b, 0, 3); /* nal_hrd, vcl_hrd, pic_struct: absent */
    bp_bits(&b, 1, 1); /* bitstream_restriction_flag */
    bp_bits(&b, 1, 1); /* motion_vectors_over_pic_boundaries_flag */
    bp_ue(&b, 0);      /* max_bytes_per_pic_denom */
    bp_ue(&b, 0);      /* max_bits_per_mb_denom */
    bp_ue(&b, 7);      /* log2_max_mv_length_horizontal: |mv| <= 64 quarter samples */
    bp_ue(&b, 7);      /* log2_max_mv_length_vertical */
    bp_ue(&b, 0);      /* max_num_reorder_frames */
    bp_ue(&b, 1);      /* max_dec_frame_buffering */
    return bp_finish(&b);
}

size_t h264_write_pps(const h264_seq *s, uint8_t *rbsp, size_t cap) {
    bitpack b = {rbsp, cap, 0, 0, 0, 0};
    bp_ue(&b, 0);      /* pic_parameter_set_id */
    bp_ue(&b, 0);      /* seq_parameter_set_id */
    bp_bits(&b, 0, 1); /* entropy_coding_mode_flag: CAVLC */
    bp_bits(&b, 0, 1); /* bottom_field_pic_order_in_frame_present_flag */
    bp_ue(&b, 0);      /* num_slice_groups_minus1 */
    bp_ue(&b, 0);      /* num_ref_idx_l0_default_active_minus1 */
    bp_ue(&b, 0);      /* num_ref_idx_l1_default_active_minus1 */
    bp_bits(&b, 0, 1); /* weighted_pred_flag */
    bp_bits(&b, 0, 2); /* weighted_bipred_idc */
    bp_se(&b, s->qp - 26);
    bp_se(&b, 0);      /* pic_init_qs_minus26 */
    bp_se(&b, 0);      /* chroma_qp_index_offset */
    bp_bits(&b, 1, 1); /* deblocking_filter_control_present_flag */
    bp_bits(&b, 0, 1); /* constrained_intra_pred_flag */
    bp_bits(&b, 0, 1); /* redundant_pic_cnt_present_flag */
    return bp_finish(&b);
}

/* Emulation prevention (7.4.1): insert 0x03 after two zero bytes when the next byte is <= 3.
 * Nothing can need escaping before the first pair of zero bytes, and in entropy-coded data such a pair turns up
 * about once in 64 KB: the payload is copied in a few large memcpy calls between pairs, found by memchr over the
 * zero bytes (one every ~256 bytes).  The first version copied the stretches between single zero bytes: 2.4 GB/s
 * on the 565 KB I pictures of the 1080p clip, which made host packaging the bottleneck once consecutive
 * batches overlapped. */
static size_t find_zero_pair(const uint8_t *s, size_t from, size_t n) { /* first p >= from with s[p] == s[p+1] == 0, else n */
    while (from + 1 < n) {
        const uint8_t *z = (const uint8_t *)memchr(s + from, 0, n - 1 - from);
        if (!z) return n;
        const size_t p = (size_t)(z - s);
        if (s[p + 1] == 0) return p;
        from = p + 2;
    }
    return n;
}

static size_t escape_into(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t o = 0, i = 0; /* i: next source byte to copy */
    size_t scan = 0;     /* the search for the next pair starts here (a zero that was already copied may open it) */
    while (i < n) {
        const size_t p = find_zero_pair(src, scan, n);
        if (p >= n) {
            memcpy(dst + o, src + i, n - i);
            o += n - i;
            break;
        }
        memcpy(dst + o, src + i, p + 2 - i); /* up to and including the pair */
        o += p + 2 - i;
        i = p + 2;
        if (i >= n) break;
        const uint8_t c = src[i];
        if (c <= 3) {
            dst[o++] = 3;
            dst[o++] = c;
            i++;
            scan = c == 0 ? i - 1 : i;
        } else
            scan = i;
    }
    return o;
}

size_t h264_wrap_nal(uint8_t *dst, int annexb, int ref_idc, int type, const uint8_t *rbsp, size_t n) {
    dst[4] = (uint8_t)(((ref_idc & 3) << 5) | (type & 31));
    const size_t body = 1 + escape_into(dst + 5, rbsp, n);
    if (annexb) {
        dst[0] = 0, dst[1] = 0, dst[2] = 0, dst[3] = 1;
    } else {
        dst[0] = (uint8_t)(body >> 24), dst[1] = (uint8_t)(body >> 16), dst[2] = (uint8_t)(body >> 8), dst[3] = (uint8_t)body;
    }
    return 4 + body;
}
