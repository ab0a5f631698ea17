/*
 * orc.h -- CPU oracle for the per-macroblock encode loop (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED.  The reference's only video-coding line is the library call
 * x264_encoder_encode() at av_encode.c:970 (flush twin av_encode.c:1078); the
 * arithmetic behind it lives in libx264, which is linked un-vendored and
 * un-pinned (Makefile:8) and is absent from /root/reference.  The reference has
 * no tests or golden vectors.  This oracle is therefore a plain-C restatement
 * of the published H.264 Constrained-Baseline algorithm as frozen in
 * spec/CODEC.md, NOT of libx264; "bit-exact" in this repo always means
 * "bit-exact against this oracle".  H.264 conformance is verified by this
 * repo's own decoder (orc_dec.c) only.
 *
 * Interface shape follows the reference's call sites: one encoder object per
 * stream (av_encode.c:369-376), frame in -> length-prefixed NAL array out
 * (av_encode.c:970, consumer contract av_encode.c:670-834).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, link or execute anything under oracle/.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NALS 512
#define ORC_ME_RANGE 16
#define ORC_ME_CANDS 1089 /* 33 x 33 */
#define ORC_MB_COEFS (27 * 16)

/* per-macroblock coefficient record: 27 blocks x 16 int16 levels in scan order.
 *  0..15  luma 4x4 blocks, luma4x4BlkIdx order (Intra16x16: index 0 unused = 0)
 *  16     Intra16x16 luma DC (16 levels, zig-zag order); zero for inter MBs
 *  17,18  Cb, Cr chroma DC (4 levels each, raster 2x2)
 *  19..22 Cb AC blocks 0..3 (levels at 1..15, index 0 = 0)
 *  23..26 Cr AC blocks 0..3
 */

typedef struct orc_enc orc_enc;

typedef struct {
    int width, height;   /* display size (need not be a multiple of 16) */
    int qp;              /* fixed QP, 12..51 */
    int n_slices;        /* MB-row bands per frame, 1..mb_height */
    int keyint;          /* closed GOP length: frame %% keyint == 0 -> IDR */
    int fps_num, fps_den;
    int sar_w, sar_h;    /* 0 -> no aspect_ratio_info */
    int threads;         /* OpenMP threads (slice / MB-row parallel); <=0 -> 1 */
    int idr_base;        /* index of the first closed GOP (idr_pic_id = (idr_base + n) & 1): lets a
                            GOP-sharded run reproduce the single-encoder stream */
    int rc_mode;         /* 0: fixed QP; 1: picture QP by spec/CODEC.md section 1a (qp = the base QP) */
    int intra_modes;     /* 0: Intra16x16 DC prediction only; 1: vertical / horizontal / DC chosen by SAD; 2: plane as well (spec/CODEC.md section 3) */
    int deblock;         /* 0: disable_deblocking_filter_idc = 1; 1: in-loop filter inside every slice (idc 2; spec/CODEC.md section 5a) */
    int subpel;          /* 0: integer-sample vectors; 1: half-sample refinement; 2: half then quarter (spec/CODEC.md section 2a) */
} orc_params;

typedef struct {
    int type;      /* nal_unit_type: 7 SPS, 8 PPS, 5 IDR slice, 1 non-IDR slice */
    int ref_idc;
    size_t offset; /* byte offset of the 4-byte length prefix inside out[] */
    size_t size;   /* bytes including the 4-byte length prefix */
} orc_nal;

orc_enc *orc_enc_open(const orc_params *p);
void orc_enc_close(orc_enc *e);

/* Encode one frame.  Returns total bytes written to out (all NALs, each with a
 * 4-byte big-endian length prefix, contiguous), or <0 on error. */
long orc_enc_encode(orc_enc *e, const uint8_t *y, int stride_y, const uint8_t *u, int stride_u,
                    const uint8_t *v, int stride_v, uint8_t *out, size_t out_cap, orc_nal *nals,
                    int *nal_count, int *is_idr);

/* Slice-parallel stand-in (tests of the multi-GPU band mode):
 * orc_enc_set_band restricts the encoder to slices [first, first+count) -- motion search,
 * transform and entropy coding of those slices only; parameter sets are emitted only by the band
 * that holds slice 0.  orc_enc_import_rows overwrites MB rows [row0, row1) of the reconstruction
 * of the last frame with a peer's rows (tight planes of the coded width) and rebuilds the padded
 * reference, which is what the exchange step of the band mode does on the GPU. */
int orc_enc_set_band(orc_enc *e, int slice_first, int slice_count);
int orc_enc_import_rows(orc_enc *e, int mb_row0, int mb_row1, const uint8_t *y, const uint8_t *u, const uint8_t *v);

/* Introspection of the last encoded frame (for kernel-level parity tests). */
int orc_enc_mb_width(const orc_enc *e);
int orc_enc_mb_height(const orc_enc *e);
const int8_t *orc_enc_mv(const orc_enc *e);        /* [mb][2] = dx,dy integer pel (the full search's result) */
const int8_t *orc_enc_mvq(const orc_enc *e);       /* [mb][2] the coded vector in quarter-sample units (4 * mv without refinement) */
const uint16_t *orc_enc_sad(const orc_enc *e);     /* [mb] best SAD (P frames) */
const int16_t *orc_enc_levels(const orc_enc *e);   /* [mb][27][16] */
const uint32_t *orc_enc_mb_bits(const orc_enc *e); /* [mb] bits of macroblock_layer (+skip_run) */
const uint8_t *orc_enc_recon(const orc_enc *e, int plane, int *stride); /* coded size */
const uint8_t *orc_enc_recon_unfiltered(const orc_enc *e, int plane);  /* the same before the in-loop filter (deblock = 1), tight planes */
/* first MB row of every slice, n_slices+1 entries */
const int *orc_enc_slice_rows(const orc_enc *e);

/* Stand-alone stages (same code the encoder uses), for unit tests. */
void orc_fdct4x4(const int16_t res[16], int16_t coef[16]);  /* raster in, raster out */
void orc_idct4x4(const int32_t d[16], int16_t res[16]);     /* raster in, (x+32)>>6 out */
int orc_quant(int coef, int qp, int raster_pos, int intra); /* AC/normal quantiser */
int orc_dequant(int level, int qp, int raster_pos);
int orc_me_rank(int dx, int dy);                            /* tie-break rank, 0..1088 */
/* picture QP under rate control (spec/CODEC.md section 1a); have_history: the previous picture was a P picture of the same GOP */
int orc_rc_qp(int base_qp, int is_idr, int have_history, uint64_t prev_sad_sum, int nmb);
int orc_enc_picture_qp(const orc_enc *e);                   /* QP of the last encoded picture */
/* luma prediction block at (xq, yq) quarter-sample units, 8.4.2.2.1 with clamped coordinates (orc_interp.h); dst is bw wide */
void orc_luma_interp(const uint8_t *ref, int stride, int w, int h, int xq, int yq, int bw, int bh, uint8_t *dst);
/* CAVLC residual block; coef[] in scan order; returns number of bits appended */
/* Intra4x4 (8.3.1; orc_intra4.h): predictor `mode` (0 .. 8) from explicit neighbours, -1 if the mode needs samples that are missing */
int orc_intra4_predict(const uint8_t top[8], const uint8_t left[4], int corner, int have_top, int have_left, int have_corner, int mode,
                       uint8_t out[16]);
int orc_intra4_topright_available(int bx, int by, int aT, int aTR);
int orc_cavlc_block_bits(const int16_t *coef, int max_coef, int nC, uint8_t *out, size_t cap);
/* emulation prevention: returns output size */
size_t orc_escape(const uint8_t *rbsp, size_t n, uint8_t *out);

/* ---- decoder (self-consistency only; shares this repo's reading of H.264) ---- */
typedef struct orc_dec orc_dec;
orc_dec *orc_dec_open(void);
void orc_dec_close(orc_dec *d);
/* Feed one access unit (length-prefixed NALs).  Returns 1 if a picture was
 * decoded, 0 if only parameter sets, <0 on a syntax error. */
int orc_dec_decode(orc_dec *d, const uint8_t *data, size_t size);
int orc_dec_width(const orc_dec *d);   /* coded size */
int orc_dec_height(const orc_dec *d);
int orc_dec_crop_width(const orc_dec *d);
int orc_dec_crop_height(const orc_dec *d);
const uint8_t *orc_dec_plane(const orc_dec *d, int plane, int *stride);
const char *orc_dec_error(const orc_dec *d);

#ifdef __cplusplus
}
#endif
#endif
