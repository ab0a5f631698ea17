/*
 * venc_internal.h -- device-side data layout shared by the kernels and the
 * launcher (venc_hip.hip).  Not part of the C-ABI (that is include/venc_hip.h).
 *
 * HBM layout (one encoder context, G = max items per step):
 *   input pool    [slots][ Y w*h | U w*h/4 | V w*h/4 ]      display size, tight I420
 *   recon         [8][G][ Y cw*ch | U | V ]                  coded size, a ring of eight: the intra pictures of the next unit are
 *                                                            coded beside the last P steps of this one (venc_hip.hip)
 *   mvsad         [G][nmb] u32   = sad<<16 | (dy&255)<<8 | (dx&255)
 *   levels        [G][nmb][27][16] i16   (block order documented in include/venc_hip.h)
 *   mbinfo        [G][nmb] 32 B  = nnz[24] (16 luma raster, 4 Cb, 4 Cr), cbp, flags
 *   mbbits        [G][nmb][VENC_MB_WORDS] u32  I pictures: the macroblock's MSB-first bit string (k_cavlc); P pictures: words 4 .. 15 of
 *                                              the rare entry beyond 128 bits, entry slot e at word 16 e (venc_cavlc_p.hip)
 *   mbnbits/mboff [G][nmb] u32
 *   refsum        [G][ch+25][cw+17 (padded)] u32 = S8(x,y) | S8(x+8,y)<<16 of the reference luma
 *   out headers (pinned host, device-mapped) [out_slots][header | slice table]
 *   out data    [out_slots][out_cap]: HBM staging, fetched to a pinned host pool by DMA (batched engines), or the
 *               device view of that host pool itself (one-item engines: lowest latency)
 */
#ifndef VENC_INTERNAL_H
#define VENC_INTERNAL_H
#include <stdint.h>

#include "venc_device.h"

#define VENC_MAX_ITEMS 64
#define VENC_MAX_SLICES 256
#define VENC_MB_COEFS 432   /* 27 blocks x 16 */
#define VENC_MB_WORDS 512   /* per-macroblock bit-string region: 8 units x 64 words (an I macroblock uses it as one string of <= 13312 bits) */
#define VENC_LANE_WORDS 16  /* per-block private capacity: 512 bits >= 16 + 16*28 */
#define VENC_ME_RANGE 16
#define VENC_ME_CANDS 1089

typedef struct {
    uint8_t nnz[24];
    uint8_t cbp;   /* coded_block_pattern, luma part: bits 0-3 = the four 8x8 blocks */
    uint8_t flags; /* bit 0: intra (Intra16x16 unless bit 3); bits 1-2: Intra16x16PredMode; bit 3: Intra4x4 (its sixteen modes sit in level slot 16) */
    uint8_t cbpc;  /* coded_block_pattern, chroma part (0, 1, 2): its own byte, the luma and chroma transform
                      blocks of a step run side by side and must not share one */
    uint8_t pad[5];
} venc_mbinfo;

typedef struct {
    uint32_t byte_off; /* offset of the slice RBSP inside the frame's output buffer (4-aligned) */
    uint32_t nbytes;   /* RBSP bytes incl. trailing bits */
    uint32_t nbits;    /* slice_data + header bits, without trailing */
    uint32_t pad;
} venc_slice_out;

/* header of one output slot in pinned host memory, followed by the slice table and the data */
typedef struct {
    uint32_t total_bytes; /* bytes used in data[] */
    uint32_t n_slices;
    uint32_t error;       /* non-zero: capacity overflow */
    uint32_t pad;
} venc_out_header;

#define VENC_STAMP_PAIRS 16384
#define VENC_STAMP_WORDS (8 + 2 * VENC_STAMP_PAIRS)
typedef struct {
    int w, h;        /* display */
    int cw, ch;      /* coded */
    int mbw, mbh, nmb;
    uint32_t mbw_magic; /* ceil(2^32 / mbw): mb / mbw = umulhi(mb, magic), exact while mb x mbw < 2^32 (0 stands for mbw = 1) */
    int n_slices;
    int slice_base, slice_rem; /* bands: the first slice_rem slices have slice_base+1 MB rows, the rest slice_base */
    int qp;
    int intra_plane; /* Intra16x16 plane prediction is a candidate (intra_modes = 2) */
    int subpel; /* 0: vectors in integer samples; 1, 2: P pictures refine them (k_subpel) and every vector of the step is in QUARTER samples */
    venc_qparams ql, qc;
} venc_geom;

typedef struct {
    int n_items;
    int item0;   /* first item of this launch (the search kernels: the items of a P step may go as two launches on two streams, venc_hip.hip TWO CHAINS) */
    int deblock; /* the in-loop filter is on: slice headers say so (k_deblock does the filtering) */
    int is_idr;
    int frame_num;
    int parity; /* recon buffer written this step (index into the ring of VENC_RECON buffers) */
    int ref;    /* recon buffer of the previous picture of the chain: the reference of a P step */
    /* band of the picture coded by this context (slice-parallel multi-GPU mode, venc_set_band);
     * the whole picture by default: slices [slice0, slice1) = MB rows [row0, row1) = macroblocks [mb0, mb1) */
    int slice0, slice1, row0, row1, mb0, mb1;
    int tile0, tile_split, tile1, tile_n; /* k_me_pool4: the launch's blocks 0 .. tile_n - 1 are the tiles [tile0, tile0 + tile_split) and [tile1, ...) (one run by default; two launches while an exchange is in flight) */
    int me_fast; /* k_me_pool: bounds tested as they are computed (VENC_ME_FAST=0: always the kept-bounds pass) */
    int me_stop; /* diagnostic (VENC_ME_STOP=n): k_me_tile returns after phase n; results are then meaningless */
    int in_slot[VENC_MAX_ITEMS];
    int out_slot[VENC_MAX_ITEMS];
    int idr_id[VENC_MAX_ITEMS];
    /* The entropy chain (k_skip / k_cavlc_p / k_cavlc, k_scan, k_pack) is launched once for up to four consecutive P steps
     * ("entropy batch", venc_hip.hip): its grid's item index then runs over (step, item) pairs, v = k * max_items + item,
     * and the per-item fields above and this one are indexed by v.  fnum[v] = frame_num of the step v belongs to. */
    uint8_t fnum[VENC_MAX_ITEMS];
    /* rate control (spec/CODEC.md section 1a; 0 = fixed QP, the quantiser constants of venc_geom apply) */
    int rc_mode;
    int qp_base;    /* pic_init_qp */
    int qp_idr;     /* QP of IDR pictures under rate control (host-known: base - 3, clipped) */
    int rc_history; /* P step: the previous picture of every chain was a P picture too (its SAD sum steers this one's QP) */
} venc_step;

#define VENC_COUNTER_STRIDE 32 /* words between per-item atomic counters: one cache line each */

typedef struct {
    const uint8_t *in_base;
    size_t in_frame_stride;
    uint8_t *recon; /* [2][G] frames */
    size_t recon_frame_bytes;
    int max_items;
    int vcap;        /* (step, item) pairs the entropy chain's per-item buffers hold: entropy batch x max_items */
    uint32_t *mvsad;
    const uint32_t *mvsad_prev; /* vectors of the previous step (temporal predictor of the motion search) */
    int16_t *levels;
    venc_mbinfo *mbinfo;
    uint32_t *mbbits;
    uint32_t *mbnbits;
    uint16_t *ubits;            /* [G][nmb][8] bits of the macroblock's syntax units (k_cavlc*), turned into running ends by k_scan */
    uint32_t *mboff;
    /* P pictures, round 4 (venc_cavlc_p.hip): a slice is cut into SEGMENTS of 256 macroblocks; segment (v, s, j) has index
     * (v * n_slices + s) * ent_spp + j.  Per segment: its syntax entries in bitstream order (mb << 5 | slot), their bit strings'
     * first four words, their bit offsets inside the segment (count + 1 of them), the segment's bits. */
    unsigned long long *codedmask; /* [G][n_slices][ent_wps] bit = macroblock (slice-relative index) is coded (not P_Skip), k_skip2 */
    uint32_t *ent_list;         /* [segments][ent_cap] mb << 5 | slot of the entries whose strings run beyond 128 bits (the others: not written) */
    uint32_t *ent_cls;          /* [segments][ent_cap] phase B's list of k_entropy_p beyond what its LDS holds */
    uint32_t *ent_cls2;         /* [segments][ent_cap] ... and the empty blocks behind those entries (two bits of nC class each, count << 16) */
    uint32_t *ent_count;        /* [segments] entries */
    uint32_t *ent_mb0;          /* [G][nmb] rank of the macroblock's first entry | its entries << 16 (debug / tests) */
    uint4 *ent_head;            /* [segments][ent_cap] */
    uint32_t *ent_off;          /* [segments][ent_cap + 4] */
    uint32_t *seg_bits;         /* [segments] */
    uint32_t ent_cap;           /* entries a segment can hold: 28 x min(256, macroblocks of the largest slice) */
    int ent_spp;                /* segments per slice (of the largest slice) */
    int ent_wps;                /* mask words per slice: 4 x ent_spp */
    uint32_t *slice_bits;       /* [G][VENC_MAX_SLICES] bits of every slice of the step (k_scan: per-slice blocks -> the block that builds the table) */
    uint32_t *scan_done;        /* [G][32] k_scan arrival counters, one cache line each; zero between launches */
    const int *slice_first_row; /* n_slices + 1 */
    const uint8_t *row_slice;   /* mbh */
    const uint16_t *me_rank;    /* [33*33] candidate (dyi*33+dxi) -> rank */
    const uint16_t *me_cand;    /* [1089] rank -> dyi<<8 | dxi */
    uint32_t *refsum;           /* [G][p8 rows][refsum_stride] packed 8x8 block sums of the reference (venc_me.hip) */
    int refsum_stride;          /* dwords per row */
    size_t refsum_item;         /* dwords per item */
    int *intra_rec;             /* [G][nmb][128] I pictures: block DCs + un-rounded edge samples (k_intra_fwd -> k_intra_dc) */
    int *intra_d0;              /* [G][nmb][32] I pictures: d0 + 64 pred + 32 per block (k_intra_dc -> k_intra_rec) */
    uint8_t *me_flag;           /* [G][nmb] 1 = k_me_tile left this macroblock to k_me_fix */
    uint32_t *fix_count;        /* [G] flagged macroblocks of the step.  Two sets alternate between P steps (the launcher sets both
                                   pointers): the search of a step zeroes the other set, which the next step's search fills */
    uint32_t *fix_count_next;
    uint32_t *fix_list;         /* [G][nmb] their indices */
    unsigned long long *stamps; /* [VENC_STAMP_WORDS] diagnostic builds (VENC_ME_STAMPS, VENC_ENT_STAMPS): 8 sums, then (start, end) clock pairs of blocks */
    uint8_t *me_cnt;            /* [G][nmb] candidates that passed the lower-bound test (saturating) */
    /* 4x4 luma blocks with a level, listed by the search's P5 for the extra blocks of the k_me_fix launch (venc_inter2.hip
     * luma_block_defer): one 64-entry segment per (tile, transforming wave), entry = lane | cx << 6 | cy << 12 */
    uint32_t *nz_ids;           /* [G][nz_item_regions][64] */
    uint4 *nz_rows;             /* [G][nz_item_regions][64][2] the listed block's four current and four predicted rows (packed samples) */
    uint32_t *nz_cnt;           /* [G][nz_item_regions] entries of each segment: written (also as 0) by every transforming wave of every P step */
    int nz_item_regions;        /* segments per item: 4 x tiles of the whole picture */
    /* rate control: three sets of per-item sums of the best SADs rotate between P steps -- the search of a step reads the
     * previous step's, adds to its own (k_me_fix corrects it for the macroblocks it redoes) and zeroes the next one's --
     * and the QP each (step, item) was coded at, in the step ring, for the kernels that run later (chroma, entropy) */
    const unsigned long long *sad_prev;
    unsigned long long *sad_cur, *sad_next;
    int *qp_cur;                /* [G] this step's slot of the QP ring (16 steps) */
    int *qp_next;               /* [G] the next step's slot: written by the last block of this step's k_me_fix from the final SAD sum */
    uint32_t *fix_done;         /* [G] k_me_fix blocks that have finished (zero between launches) */
    uint32_t *mvq;              /* sub-sample refinement: the step's vectors in QUARTER samples (SAD << 16 | qy << 8 | qx), what the
                                   transform and entropy kernels are handed as `mvsad`; the searches keep their integer vectors */
    uint8_t *hpel;              /* [item][3][cw * ch] sub-sample refinement: the half-sample planes b, h, j of the step's reference pictures (k_hpel) */
    const venc_qparams *qtab;   /* [52] quantiser constants of every QP (built by the host at open) */
    uint8_t *out_hdr;           /* device view of the pinned header pool: per slot venc_out_header + slice table */
    size_t out_hdr_stride;
    uint8_t *out_data;          /* per slot data[]: HBM staging buffer, or the device view of the pinned host pool */
    size_t out_data_stride;
    uint32_t out_cap;           /* capacity of data[] */
} venc_bufs;

#endif
