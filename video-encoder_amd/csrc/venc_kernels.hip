/*
 * venc_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the
 * per-macroblock encode loop.  No MFMA: nothing here is a dense contraction.
 *
 * This file: the shared device helpers, the entropy chain's common kernels and the first-generation kernels that
 * are kept as selectable variants and as on-GPU test oracles:
 *   k_me_dense  a4.1  exhaustive +-16 integer-pel 16x16 SAD search, LDS-staged window (VENC_ME=dense)
 *   k_inter     a4.2/3/5  MC prediction, 4x4 core transform, quantisation, reconstruction (P; VENC_INTER=legacy)
 *   k_intra     a4.2/3/5  Intra16x16-DC wavefront per slice (I; VENC_INTRA=legacy)
 *   k_cavlc     a4.4  zig-zag/run-level entropy coding of I pictures, one lane per 4x4 block
 *   k_scan      a4.4  I pictures: per-slice exclusive scan of the macroblocks' bit lengths
 *   k_pack      a4.4  I pictures: owner-gather bit packing into the output slot
 *   k_band_copy a4.5  slice-parallel mode: rows of a band <-> exchange chunk
 * The shipped path's other kernels: venc_me3.hip (k_me_pool), venc_me2.hip (k_me_fix), venc_inter2.hip
 * (k_inter_fused), venc_intra2.hip (k_intra_fwd / k_intra_dc / k_intra_rec), venc_cavlc_p.hip (P pictures: k_skip2, k_cavlc_p2, k_pack_p).
 *
 * Row references (a4.x) are SURVEY.md section 8a; the reference has no source
 * for any of these stages (they sit behind x264_encoder_encode(),
 * av_encode.c:970), so the arithmetic follows spec/CODEC.md.
 */
#include <hip/hip_runtime.h>

#include "venc_internal.h"

#define DEVFN __device__ __forceinline__

/* ------------------------------------------------------------------ plane addressing */
DEVFN const uint8_t *in_plane(const venc_bufs &B, const venc_geom &g, int slot, int plane, int &stride, int &pw,
                              int &ph) {
    const uint8_t *base = B.in_base + (size_t)slot * B.in_frame_stride;
    if (plane == 0) {
        stride = g.w;
        pw = g.w;
        ph = g.h;
        return base;
    }
    stride = g.w >> 1;
    pw = g.w >> 1;
    ph = g.h >> 1;
    size_t ysz = (size_t)g.w * g.h;
    return base + ysz + (plane == 2 ? (ysz >> 2) : 0);
}
DEVFN uint8_t *rec_plane(const venc_bufs &B, const venc_geom &g, int parity, int item, int plane, int &stride) {
    uint8_t *base = B.recon + (size_t)(parity * B.max_items + item) * B.recon_frame_bytes;
    size_t ysz = (size_t)g.cw * g.ch;
    if (plane == 0) {
        stride = g.cw;
        return base;
    }
    stride = g.cw >> 1;
    return base + ysz + (plane == 2 ? (ysz >> 2) : 0);
}
DEVFN int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
/* macroblock address -> (column, row) without an integer division (~25 instructions): one high multiply by ceil(2^32 / mbw) */
DEVFN void mb_xy(const venc_geom &g, int mb, int &mx, int &my) {
    my = g.mbw_magic ? (int)__umulhi((uint32_t)mb, g.mbw_magic) : mb;
    mx = mb - my * g.mbw;
}
/* first MB row of the slice that contains MB row my (bands are uniform: no table lookup, no memory round trip) */
DEVFN int slice_first_row_of(const venc_geom &g, int my) {
    const int big = g.slice_rem * (g.slice_base + 1);
    return my < big ? (my / (g.slice_base + 1)) * (g.slice_base + 1) : big + ((my - big) / g.slice_base) * g.slice_base;
}

/* luma4x4BlkIdx <-> (x,y) in 4x4 units */
DEVFN int blk_x(int b) { return (b & 1) | ((b >> 1) & 2); }
DEVFN int blk_y(int b) { return ((b >> 1) & 1) | ((b >> 2) & 2); }
DEVFN int blk_of(int x, int y) { return (x & 1) | ((y & 1) << 1) | ((x >> 1) << 2) | ((y >> 1) << 3); }

/* ---- half-wave (32-lane) reductions on the DPP network: 4 row_shr steps inside each row of 16
 * lanes, one row_bcast:15 into rows 1 and 3, then the totals are read from lanes 31 and 63.
 * About 8 VALU/SALU instructions and no LDS traffic (ds_bpermute butterflies cost ~5 LDS round trips). */
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
template <int CTRL, int ROWMASK>
DEVFN uint32_t dpp_fetch(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, ROWMASK, 0xf, false);
}
DEVFN uint32_t half_sum32(uint32_t v, int lane) {
    v += dpp_fetch<DPP_ROW_SHR(1), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(2), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(4), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(8), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_BCAST15, 0xa>(v, 0u);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return lane < 32 ? lo : hi;
}
DEVFN uint32_t half_min32(uint32_t v, int lane) {
    v = min(v, dpp_fetch<DPP_ROW_SHR(1), 0xf>(v, 0xffffffffu));
    v = min(v, dpp_fetch<DPP_ROW_SHR(2), 0xf>(v, 0xffffffffu));
    v = min(v, dpp_fetch<DPP_ROW_SHR(4), 0xf>(v, 0xffffffffu));
    v = min(v, dpp_fetch<DPP_ROW_SHR(8), 0xf>(v, 0xffffffffu));
    v = min(v, dpp_fetch<DPP_ROW_BCAST15, 0xa>(v, 0xffffffffu));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return lane < 32 ? lo : hi;
}

/* inclusive prefix sum inside each 32-lane half of a wave */
DEVFN uint32_t half_scan_incl32(uint32_t v) {
    v += dpp_fetch<DPP_ROW_SHR(1), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(2), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(4), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(8), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_BCAST15, 0xa>(v, 0u);
    return v;
}
/* inclusive prefix sum over the 64 lanes of a wave (row scans + row_bcast:15 / row_bcast:31) */
#define DPP_ROW_BCAST31 0x143
DEVFN uint32_t wave_scan_incl(uint32_t v) {
    v += dpp_fetch<DPP_ROW_SHR(1), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(2), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(4), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(8), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_BCAST15, 0xa>(v, 0u);
    v += dpp_fetch<DPP_ROW_BCAST31, 0xc>(v, 0u);
    return v;
}

#ifdef VENC_LEGACY_KERNELS /* first-generation kernels: test-only library */
#include "legacy/kernels_gen1_a.hip"
#endif

/* ================================================================== macroblock transform core
 * Lane roles inside a 32-lane macroblock team: l32 0..15 = luma4x4BlkIdx,
 * 16..19 = Cb block k, 20..23 = Cr block k, 24..31 spare.  Every lane of the
 * wave executes this function (the cross-lane DC exchanges use ds_bpermute).
 */
struct mb_lane_out {
    int lv[16]; /* levels, scan order (index 0 zero when the DC is coded separately) */
    int rec[16];
    int dclv;   /* separately coded DC level owned by this lane (I16 luma / chroma) */
    int nnz;
};

template <bool i16, bool intra>
DEVFN void mb_code(int lane, const int cur[16], const int pred[16], const venc_geom &g, mb_lane_out &o) {
    const int l32 = lane & 31, base = lane & 32;
    const bool luma = l32 < 16;
    const bool dc_sep = luma ? i16 : true;
    const venc_qparams &q = luma ? g.ql : g.qc;
    const int qmf0 = luma ? g.ql.mf[0] : g.qc.mf[0], qmf1 = luma ? g.ql.mf[1] : g.qc.mf[1],
              qmf2 = luma ? g.ql.mf[2] : g.qc.mf[2];
    const int qv0 = luma ? g.ql.v[0] : g.qc.v[0], qv1 = luma ? g.ql.v[1] : g.qc.v[1], qv2 = luma ? g.ql.v[2] : g.qc.v[2];
    const int qbits = luma ? g.ql.qbits : g.qc.qbits, per = luma ? g.ql.per : g.qc.per;
    const int f = intra ? (luma ? g.ql.f_intra : g.qc.f_intra) : (luma ? g.ql.f_inter : g.qc.f_inter);
    (void)q;

    int res[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) res[i] = cur[i] - pred[i];
    venc_fdct4(res, w);

    /* ---- separately coded DC: Intra16x16 luma (4x4 Hadamard over the 16 luma lanes) */
    int oy, ox;
    {
        int bx = blk_x(l32 & 15), by = blk_y(l32 & 15);
        oy = by;
        ox = bx;
    }
    int dcl_luma = 0, d0_luma = 0;
    if (i16) { /* compile-time: inter macroblocks carry no separate luma DC */
        int ydc = 0;
#pragma unroll
        for (int qi = 0; qi < 16; qi++) {
            int qy = qi >> 2, qx = qi & 3;
            int v = __shfl(w[0], base | blk_of(qx, qy));
            int neg = ((0xA6C0 >> (4 * oy + qy)) ^ (0xA6C0 >> (4 * ox + qx))) & 1;
            ydc += neg ? -v : v;
        }
        dcl_luma = venc_quant_luma_dc(ydc, &g.ql);
        int fdc = 0;
#pragma unroll
        for (int qi = 0; qi < 16; qi++) {
            int qy = qi >> 2, qx = qi & 3;
            int v = __shfl(dcl_luma, base | blk_of(qx, qy));
            int neg = ((0xA6C0 >> (4 * oy + qy)) ^ (0xA6C0 >> (4 * ox + qx))) & 1;
            fdc += neg ? -v : v;
        }
        d0_luma = venc_dequant_luma_dc(fdc, &g.ql);
    }

    /* ---- chroma DC: 2x2 over the four lanes of a plane */
    const int cbase = base | (l32 & 28), ck = l32 & 3;
    int yc = 0;
#pragma unroll
    for (int qi = 0; qi < 4; qi++) {
        int v = __shfl(w[0], cbase | qi);
        int neg = __popc(ck & qi) & 1;
        yc += neg ? -v : v;
    }
    int dcl_chroma = venc_quant_chroma_dc(yc, &g.qc, intra);
    int fc = 0;
#pragma unroll
    for (int qi = 0; qi < 4; qi++) {
        int v = __shfl(dcl_chroma, cbase | qi);
        int neg = __popc(ck & qi) & 1;
        fc += neg ? -v : v;
    }
    int d0_chroma = venc_dequant_chroma_dc(fc, &g.qc);

    /* ---- per-lane quantisation / dequantisation */
    int d[16];
    int nnz = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = venc_zigzag_ct(i);
        const int cls = venc_pos_class(r);
        const int mf = cls == 0 ? qmf0 : cls == 1 ? qmf1 : qmf2;
        const int vv = cls == 0 ? qv0 : cls == 1 ? qv1 : qv2;
        int l = venc_quant_mag(w[r], mf, f, qbits);
        if (i == 0 && dc_sep) l = 0;
        o.lv[i] = l;
        nnz += (l != 0);
        d[r] = VENC_MUL24(l, vv) << per;
    }
    if (dc_sep) d[0] = luma ? d0_luma : d0_chroma;
    o.dclv = luma ? dcl_luma : dcl_chroma;
    o.nnz = nnz;
    int rr[16];
    venc_idct4(d, rr);
#pragma unroll
    for (int i = 0; i < 16; i++) o.rec[i] = venc_clip_u8(pred[i] + rr[i]);
}

/* store one lane's outputs: levels, reconstruction, nnz; lane < 24 only */
DEVFN void mb_store(int l32, bool valid, const mb_lane_out &o, int16_t *lv_mb, venc_mbinfo *info, uint8_t *rec, int rstride,
                    int px, int py) {
    if (!valid || l32 >= 24) return;
    const bool luma = l32 < 16;
    const int slot = luma ? l32 : 19 + (l32 - 16);
    /* all-zero blocks are not written: k_cavlc reads a block's levels only when its nnz is non-zero
     * (most blocks of a P frame are empty, and the level record is the largest stream of the path) */
    if (o.nnz) {
        uint32_t p[8];
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = ((uint32_t)o.lv[2 * i] & 0xffffu) | ((uint32_t)o.lv[2 * i + 1] << 16);
        uint4 *dst = (uint4 *)(lv_mb + slot * 16);
        dst[0] = make_uint4(p[0], p[1], p[2], p[3]);
        dst[1] = make_uint4(p[4], p[5], p[6], p[7]);
    }
    const int nidx = luma ? blk_y(l32) * 4 + blk_x(l32) : l32;
    info->nnz[nidx] = (uint8_t)o.nnz;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t v = (uint32_t)o.rec[4 * y] | ((uint32_t)o.rec[4 * y + 1] << 8) | ((uint32_t)o.rec[4 * y + 2] << 16) |
                     ((uint32_t)o.rec[4 * y + 3] << 24);
        *(uint32_t *)(rec + (size_t)(py + y) * rstride + px) = v;
    }
}

#ifdef VENC_LEGACY_KERNELS /* first-generation kernels: test-only library */
#include "legacy/kernels_gen1_b.hip"
#endif

/* ================================================================== k_intra
 * Intra16x16 with DC prediction needs the reconstructed left and top
 * neighbours, so a slice is a 2-D dependency wavefront.  One block per slice;
 * half-wave r owns MB row r of the band and runs one macroblock behind half-wave r-1.
 * Only the neighbour *sums* are needed for DC prediction, so waves hand over
 * five integers per macroblock through a small LDS ring. */
#define INTRA_MAX_WAVES 16

/* four rows of a current 4x4 block as packed dwords (aligned loads inside the display area, clamped bytes on its edges) */
DEVFN void load_cur_rows(const uint8_t *cur, int cstride, int pw, int ph, int px, int py, bool aligned, uint32_t q[4]) {
    if (aligned && px + 4 <= pw && py + 4 <= ph) {
#pragma unroll
        for (int y = 0; y < 4; y++) q[y] = *(const uint32_t *)(cur + (size_t)(py + y) * cstride + px);
    } else {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint8_t *row = cur + (size_t)min(py + y, ph - 1) * cstride;
            q[y] = (uint32_t)row[min(px, pw - 1)] | ((uint32_t)row[min(px + 1, pw - 1)] << 8) |
                   ((uint32_t)row[min(px + 2, pw - 1)] << 16) | ((uint32_t)row[min(px + 3, pw - 1)] << 24);
        }
    }
}

#ifdef VENC_LEGACY_KERNELS /* first-generation kernels: test-only library */
#include "legacy/kernels_gen1_c.hip"
#endif

/* ================================================================== k_cavlc
 * Half-wave per macroblock, one lane per syntax element group, in syntax order:
 *   l32 0 header (slice header for the first MB of a slice, mb_skip_run,
 *         mb_type, mvd / intra modes, cbp, mb_qp_delta)
 *       1 Intra16x16 luma DC        2..17 luma blocks 0..15
 *       18,19 Cb,Cr DC              20..27 Cb,Cr AC blocks
 * Each lane codes into its own LDS strip; a half-wave prefix sum of the bit
 * counts gives the concatenation offsets; strips are OR-ed into the
 * macroblock's bit string in LDS and streamed out to HBM. */
#define CAVLC_MBS 8
/* A lane's 16 levels go to LDS as eight dwords and the CAVLC coder reads them back as int16 (c[first + i]).  Through plain
 * uint32_t / int16_t pointers that is a strict-aliasing violation the compiler may act on: in round 4's k_cavlc_p2 it moved the
 * int16 loads ABOVE the dword stores and the coder saw the previous entry's levels (escape-coded garbage, different from run to
 * run).  Stores through this type may alias anything. */
typedef uint32_t __attribute__((may_alias)) lds_u32;

/* byte idx (0..23) of a 32-byte macroblock record held in two uint4 */
DEVFN int info_byte(const uint4 &a, const uint4 &b, int idx) {
    const uint32_t w = idx < 4 ? a.x : idx < 8 ? a.y : idx < 12 ? a.z : idx < 16 ? a.w : idx < 20 ? b.x : b.y;
    return (int)((w >> (8 * (idx & 3))) & 255u);
}

DEVFN void load_levels(const int16_t *src, int c[16]) {
    const uint4 a = ((const uint4 *)src)[0], b = ((const uint4 *)src)[1];
    const uint32_t wds[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c[2 * i] = (int)(int16_t)(wds[i] & 0xffffu);
        c[2 * i + 1] = (int)(int16_t)(wds[i] >> 16);
    }
}

/* slice_header(), 7.3.3 -- written by the lane that codes the first macroblock of a slice */
template <class BS>
DEVFN void emit_slice_header(BS *bs, int mb, const venc_step &S, int item, const venc_bufs &B) {
    venc_bs_ue(bs, (uint32_t)mb);
    venc_bs_ue(bs, S.is_idr ? 7u : 5u);
    venc_bs_ue(bs, 0);
    venc_bs_put(bs, (uint32_t)S.fnum[item] & 15u, 4);
    if (S.is_idr) {
        venc_bs_ue(bs, (uint32_t)S.idr_id[item]);
        venc_bs_put(bs, 0, 2); /* no_output_of_prior_pics, long_term_reference */
    } else {
        venc_bs_put(bs, 0, 3); /* override, list modification, adaptive marking */
    }
    /* slice_qp_delta: 0 at fixed QP; under rate control the picture's QP (spec/CODEC.md section 1a) against pic_init_qp */
    venc_bs_se(bs, S.rc_mode ? (S.is_idr ? S.qp_idr : B.qp_cur[item]) - S.qp_base : 0);
    if (S.deblock) {
        venc_bs_ue(bs, 2); /* disable_deblocking_filter_idc: the in-loop filter inside the slice (spec/CODEC.md section 5a) */
        venc_bs_se(bs, 0); /* slice_alpha_c0_offset_div2 */
        venc_bs_se(bs, 0); /* slice_beta_offset_div2 */
    } else
        venc_bs_ue(bs, 1); /* disable_deblocking_filter_idc */
}

/* 16x16 motion vector prediction (8.4.1.3.1) of macroblock mb from its neighbours A (left), B (top), C (top right, or D top
 * left when C is missing); also reports whether A / B is missing or has a zero vector -- then the vector a decoder infers
 * for a P_Skip macroblock is zero instead (8.4.1.1) */
DEVFN void mv_predict(const venc_geom &g, const uint32_t *mvs, int mb, int mx, bool aL, bool aT, int &pxv, int &pyv, bool &skip_zero) {
    bool aA = aL, aB = aT, aC = aT && (mx + 1 < g.mbw);
    const bool aD = aT && mx > 0;
    int ax = 0, ay = 0, bxv = 0, byv = 0, cx = 0, cy = 0;
    if (aA) {
        uint32_t m = mvs[mb - 1];
        ax = (int)(int8_t)(m & 255u), ay = (int)(int8_t)((m >> 8) & 255u);
    }
    if (aB) {
        uint32_t m = mvs[mb - g.mbw];
        bxv = (int)(int8_t)(m & 255u), byv = (int)(int8_t)((m >> 8) & 255u);
    }
    if (aC) {
        uint32_t m = mvs[mb - g.mbw + 1];
        cx = (int)(int8_t)(m & 255u), cy = (int)(int8_t)((m >> 8) & 255u);
    } else if (aD) {
        uint32_t m = mvs[mb - g.mbw - 1];
        aC = true;
        cx = (int)(int8_t)(m & 255u), cy = (int)(int8_t)((m >> 8) & 255u);
    }
    skip_zero = !aL || !aT || (ax == 0 && ay == 0) || (bxv == 0 && byv == 0);
    if (!aB && !aC && aA)
        pxv = ax, pyv = ay;
    else if ((int)aA + (int)aB + (int)aC == 1)
        pxv = aA ? ax : aB ? bxv : cx, pyv = aA ? ay : aB ? byv : cy;
    else
        pxv = venc_median3(ax, bxv, cx), pyv = venc_median3(ay, byv, cy);
}

/* P_Skip (7.3.4): no residual and the vector a decoder would infer */
DEVFN bool mb_is_p_skip(const venc_geom &g, const uint32_t *mvs, int mb, int mx, bool aL, bool aT, int cbp) {
    if (cbp != 0) return false;
    int pxv, pyv;
    bool z;
    mv_predict(g, mvs, mb, mx, aL, aT, pxv, pyv, z);
    const uint32_t me = mvs[mb];
    const int dx = (int)(int8_t)(me & 255u), dy = (int)(int8_t)((me >> 8) & 255u);
    return z ? (dx == 0 && dy == 0) : (dx == pxv && dy == pyv);
}

/* mb_skip_run, mb_type, mvd, coded_block_pattern, mb_qp_delta of a P_L0_16x16 macroblock */
template <class BS>
DEVFN void emit_p_mb_header(BS *bs, const venc_geom &g, const uint32_t *mvs, int mb, int mx, int my, bool aL, bool aT,
                            int cbp, uint32_t skip_run) {
    (void)my;
    const uint32_t me = mvs[mb];
    const int dx = (int)(int8_t)(me & 255u), dy = (int)(int8_t)((me >> 8) & 255u);
    int pxv, pyv;
    bool z;
    mv_predict(g, mvs, mb, mx, aL, aT, pxv, pyv, z);
    venc_bs_ue(bs, skip_run); /* mb_skip_run: P_Skip macroblocks since the previous coded one */
    venc_bs_ue(bs, 0); /* mb_type P_L0_16x16 */
    venc_bs_se(bs, (dx - pxv) << (g.subpel ? 0 : 2)); /* mvd_l0 in quarter samples */
    venc_bs_se(bs, (dy - pyv) << (g.subpel ? 0 : 2));
    venc_bs_ue(bs, venc_cbp_inter_code[cbp]);
    if (cbp) venc_bs_se(bs, 0);
}

__global__ __launch_bounds__(256) void k_cavlc(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_priv[CAVLC_MBS * 32 * VENC_LANE_WORDS];
    __shared__ uint32_t s_mb[CAVLC_MBS][VENC_MB_WORDS + 2];
    __shared__ uint16_t s_tab[VENC_TAB_SIZE];
    const int item = blockIdx.z, tid = threadIdx.x;
    const int team = tid >> 5, lane = tid & 63, l32 = lane & 31, base = lane & 32;
    /* VLC tables: global loads first, LDS writes after the macroblock records below are in flight */
    uint16_t tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = tid + 256 * k;
        tv[k] = i >= VENC_TAB_SIZE      ? (uint16_t)0
              : i < VENC_TAB_TOKEN_CDC  ? venc_vlc_coeff_token[i]
              : i < VENC_TAB_TZ         ? venc_vlc_coeff_token_cdc[i - VENC_TAB_TOKEN_CDC]
              : i < VENC_TAB_TZ_CDC     ? venc_vlc_total_zeros[i - VENC_TAB_TZ]
              : i < VENC_TAB_RUN        ? venc_vlc_total_zeros_cdc[i - VENC_TAB_TZ_CDC]
                                        : venc_vlc_run_before[i - VENC_TAB_RUN];
    }
    const int mb_raw = S.mb0 + blockIdx.x * CAVLC_MBS + team;
    const bool valid = mb_raw < S.mb1;
    const int mb = valid ? mb_raw : S.mb0;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const int srow = slice_first_row_of(g, my);
    const bool aL = mx > 0, aT = my > srow;

    /* the macroblock's own 32-byte record and its left / top neighbours' (clamped), fetched whole
     * and unconditionally: one memory round trip instead of a chain of dependent byte loads */
    const venc_mbinfo *infos = B.mbinfo + (size_t)item * g.nmb;
    const uint4 *ip = (const uint4 *)(infos + mb);
    const uint4 *lp = (const uint4 *)(infos + (aL ? mb - 1 : mb));
    const uint4 *tp = (const uint4 *)(infos + (aT ? mb - g.mbw : mb));
    const uint4 o0 = ip[0], o1 = ip[1], l0 = lp[0], l1 = lp[1], t0 = tp[0], t1 = tp[1];
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (tid + 256 * k < VENC_TAB_SIZE) s_tab[tid + 256 * k] = tv[k];
    __syncthreads();
    const int cbpl = (int)(o1.z & 15u), cbpc = (int)((o1.z >> 16) & 3u), cbp = cbpl | (cbpc << 4);
    const bool i4 = (o1.z >> 11) & 1u;              /* flags bit 3: Intra4x4 (I_NxN) */
    const bool i16 = ((o1.z >> 8) & 1u) && !i4;     /* flags bit 0: intra */

    /* role decode */
    int slot = 0, first = 0, maxc = 16, nC = 0;
    bool coded = false;
    int own_idx = 0; /* index of this lane's block in nnz[] */
    if (l32 == 1) {
        slot = 16;
        coded = i16;
    } else if (l32 >= 2 && l32 < 18) {
        slot = l32 - 2;
        own_idx = blk_y(slot) * 4 + blk_x(slot);
        if (i16) {
            first = 1, maxc = 15;
            coded = cbpl != 0;
        } else
            coded = (cbpl >> (slot >> 2)) & 1;
    } else if (l32 == 18 || l32 == 19) {
        slot = 17 + (l32 - 18);
        maxc = 4;
        nC = -1;
        coded = cbpc > 0;
    } else if (l32 >= 20 && l32 < 28) {
        slot = 19 + (l32 - 20);
        own_idx = 16 + (l32 - 20);
        first = 1, maxc = 15;
        coded = cbpc == 2;
    }
    if (!valid) coded = false;
    /* nC from neighbouring TotalCoeff (9.2.1) */
    if (coded && nC == 0) {
        int ia, ib;      /* nnz indices of the left / top neighbour block */
        bool inA, inB;   /* neighbour block lies inside this macroblock */
        if (l32 < 18) {
            const int b = l32 == 1 ? 0 : l32 - 2;
            const int bx = blk_x(b), by = blk_y(b);
            inA = bx > 0, inB = by > 0;
            ia = inA ? by * 4 + bx - 1 : by * 4 + 3;
            ib = inB ? (by - 1) * 4 + bx : 12 + bx;
        } else {
            const int kk = (l32 - 20) & 3, nb = 16 + 4 * ((l32 - 20) >> 2);
            const int bx = kk & 1, by = kk >> 1;
            inA = bx > 0, inB = by > 0;
            ia = inA ? nb + 2 * by : nb + 2 * by + 1;
            ib = inB ? nb + bx : nb + 2 + bx;
        }
        const int nA = inA ? info_byte(o0, o1, ia) : info_byte(l0, l1, ia);
        const int nB = inB ? info_byte(o0, o1, ib) : info_byte(t0, t1, ib);
        nC = venc_nc(inA || aL, nA, inB || aT, nB);
    }
    venc_bitsink bs;
    venc_bs_init(&bs, s_priv + (team * 32 + l32) * VENC_LANE_WORDS);
    if (coded) {
        /* AC / luma blocks with nnz = 0 were not written by the transform kernels */
        const bool empty = ((l32 >= 2 && l32 < 18) || (l32 >= 20 && l32 < 28)) && info_byte(o0, o1, own_idx) == 0;
        /* level strips (8 dwords + 1 of padding per lane) live in the team's still unused bit-string buffer */
        lds_u32 *strip = (lds_u32 *)(&s_mb[team][0] + l32 * 9);
        uint32_t nz = 0;
        if (!empty) {
            const int16_t *src = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS + slot * 16;
            const uint4 a = ((const uint4 *)src)[0], bq = ((const uint4 *)src)[1];
            const uint32_t wds[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                strip[i] = wds[i];
                nz |= (uint32_t)((wds[i] & 0xffffu) != 0u) << (2 * i);
                nz |= (uint32_t)((wds[i] >> 16) != 0u) << (2 * i + 1);
            }
            nz = (nz >> first) & ((1u << maxc) - 1u);
        }
        venc_cavlc_block_mem(&bs, (const int16_t *)strip, nz, first, maxc, nC, s_tab);
    }
    if (l32 == 0 && valid) {
        if (mx == 0 && my == srow) emit_slice_header(&bs, mb, S, item, B);
        if (i16) {
            venc_bs_ue(&bs, (uint32_t)(1 + (int)((o1.z >> 9) & 3u) + 4 * cbpc + 12 * (cbpl != 0))); /* mb_type (table 7-11) carries Intra16x16PredMode: flags bits 1-2 */
            venc_bs_ue(&bs, 0);
            venc_bs_se(&bs, 0);
        } else if (i4) {
            /* I_NxN (7.3.5, 7.3.5.1): mb_type 0, sixteen prev_intra4x4_pred_mode_flag [+ rem_intra4x4_pred_mode] against
             * min(mode of the block to the left, mode of the block above) -- DC when either macroblock is outside the slice, a
             * macroblock that is not Intra4x4 counting as DC (8.3.1.1) --, intra_chroma_pred_mode, coded_block_pattern (Intra
             * column of table 9-4), mb_qp_delta iff cbp != 0.  The modes sit in level slot 16 of their macroblock. */
            const int16_t *lvb = B.levels + (size_t)item * g.nmb * VENC_MB_COEFS + 16 * 16;
            __attribute__((aligned(16))) int16_t own[16], lm[16], tm[16];
            const bool li4 = aL && ((l1.z >> 11) & 1u), ti4 = aT && ((t1.z >> 11) & 1u);
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                *(uint4 *)(own + 8 * h2) = ((const uint4 *)(lvb + (size_t)mb * VENC_MB_COEFS))[h2];
                *(uint4 *)(lm + 8 * h2) = ((const uint4 *)(lvb + (size_t)(li4 ? mb - 1 : mb) * VENC_MB_COEFS))[h2];
                *(uint4 *)(tm + 8 * h2) = ((const uint4 *)(lvb + (size_t)(ti4 ? mb - g.mbw : mb) * VENC_MB_COEFS))[h2];
            }
            venc_bs_ue(&bs, 0);
#pragma unroll
            for (int blk = 0; blk < 16; blk++) {
                const int bx = blk_x(blk), by = blk_y(blk);
                const int mA = bx > 0 ? own[blk_of(bx - 1, by)] : !aL ? -1 : li4 ? lm[blk_of(3, by)] : 2;
                const int mB = by > 0 ? own[blk_of(bx, by - 1)] : !aT ? -1 : ti4 ? tm[blk_of(bx, 3)] : 2;
                const int pm = (mA < 0 || mB < 0) ? 2 : min(mA, mB), m = own[blk];
                if (m == pm)
                    venc_bs_put(&bs, 1u, 1);
                else
                    venc_bs_put(&bs, (uint32_t)(m < pm ? m : m - 1), 4); /* flag 0 + three bits */
            }
            venc_bs_ue(&bs, 0);
            venc_bs_ue(&bs, venc_cbp_intra_code[cbp]);
            if (cbp) venc_bs_se(&bs, 0);
        } else {
            emit_p_mb_header(&bs, g, B.mvsad + (size_t)item * g.nmb, mb, mx, my, aL, aT, cbp, 0u); /* (P pictures are coded by k_cavlc_p) */
        }
    }
    const int nbits = venc_bs_flush(&bs);
    /* inclusive prefix over the 32 lanes of the team */
    int incl = (int)half_scan_incl32((uint32_t)nbits);
    int total = __shfl(incl, base | 31);
    if (total > 416 * 32) total = 416 * 32; /* cannot happen for QP >= 12; keeps LDS in bounds and the length inside 16 bits */
    const int off = incl - nbits;
    const int nw = (total + 31) >> 5;
    for (int j = l32; j < nw + 1; j += 32) s_mb[team][j] = 0;
    __syncthreads();
    {
        const uint32_t *pw = s_priv + (team * 32 + l32) * VENC_LANE_WORDS;
        const int myw = (nbits + 31) >> 5;
        for (int j = 0; j < myw; j++) {
            const int pos = off + 32 * j;
            if (pos >= 416 * 32) break;
            const uint32_t wd = pw[j];
            const int shb = pos & 31;
            atomicOr(&s_mb[team][pos >> 5], wd >> shb);
            if (shb) atomicOr(&s_mb[team][(pos >> 5) + 1], wd << (32 - shb));
        }
    }
    __syncthreads();
    if (valid) {
        uint32_t *dst = B.mbbits + ((size_t)item * g.nmb + mb) * VENC_MB_WORDS;
        for (int j = l32; j < nw; j += 32) dst[j] = s_mb[team][j];
        if (l32 == 0) B.mbnbits[(size_t)item * g.nmb + mb] = (uint32_t)total;
    }
}

/* ================================================================== k_scan
 * I pictures (P pictures need no scan kernel: venc_cavlc_p.hip).  One block per slice and item: exclusive scan of the
 * macroblocks' bit counts (k_cavlc's mbnbits -> mboff, the offsets inside the slice); the block that finishes last for an item
 * writes the item's slice table (byte offsets inside the frame's output buffer).  The waves of a block take contiguous runs of the
 * slice: a pass for the runs' totals, a block-wide exclusive sum of them, a second pass that scans and writes.
 * (The first version was one block per item with one wave per slice: 10 blocks per launch, 18 us at 1080p and
 * 60 us at 2160p, all of it latency, on a chain that had become nearly as long as the main chain.) */
#define SCAN_WAVES 16 /* (4 waves per block measured the same within noise: 143.8 against 145.3 Gpx/s at 1080p) */
__global__ __launch_bounds__(64 * SCAN_WAVES) void k_scan(venc_geom g, venc_bufs B, venc_step S, venc_slice_out *slice_dev) {
    __shared__ uint32_t s_tot[SCAN_WAVES];
    __shared__ uint32_t s_last;
    const int item = blockIdx.y, s = S.slice0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t *nbp = B.mbnbits + (size_t)item * g.nmb;
    uint32_t *offp = B.mboff + (size_t)item * g.nmb;
    const int first = B.slice_first_row[s] * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    const int per = ((last - first + SCAN_WAVES * 64 - 1) / (SCAN_WAVES * 64)) * 64; /* macroblocks per wave: whole groups of 64 */
    const int w0 = first + wv * per, w1 = min(w0 + per, last);
    /* pass 1: this wave's total */
    uint32_t acc = 0;
#pragma unroll 1
    for (int i = w0 + lane; i < w1; i += 64) acc += nbp[i];
    const uint32_t wsum = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(acc), 63);
    if (lane == 0) s_tot[wv] = wsum;
    __syncthreads();
    uint32_t running = 0, total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_WAVES; k++) {
        const uint32_t t = s_tot[k];
        running += k < wv ? t : 0u;
        total += t;
    }
    /* pass 2: scan and write (the lengths come out of L2 this time) */
#pragma unroll 1
    for (int c0 = w0; c0 < w1; c0 += 64) {
        const int i = c0 + lane;
        const uint32_t v = i < w1 ? nbp[i] : 0u;
        const uint32_t incl = wave_scan_incl(v);
        if (i < w1) offp[i] = running + incl - v;
        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    /* slice total -> memory; the last block of the item to get here builds the slice table */
    if (tid == 0) {
        B.slice_bits[(size_t)item * VENC_MAX_SLICES + s] = total;
        __threadfence(); /* agent-scope release: the total is visible before the arrival below */
        const uint32_t arrived = atomicAdd(&B.scan_done[item * VENC_COUNTER_STRIDE], 1u);
        s_last = arrived == (uint32_t)(S.slice1 - S.slice0 - 1);
        if (s_last) {
            __threadfence();
            B.scan_done[item * VENC_COUNTER_STRIDE] = 0; /* for the next step (same stream: ordered) */
            uint8_t *slot = B.out_hdr + (size_t)S.out_slot[item] * B.out_hdr_stride;
            venc_slice_out *host_tab = (venc_slice_out *)(slot + sizeof(venc_out_header));
            uint32_t byte_off = 0;
            for (int k = 0; k < g.n_slices; k++) {
                const bool mine = k >= S.slice0 && k < S.slice1; /* slices of other bands: empty entries */
                const uint32_t nbits = mine ? __hip_atomic_load(&B.slice_bits[(size_t)item * VENC_MAX_SLICES + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                const uint32_t nbytes = mine ? (nbits + 1 + 7) >> 3 : 0u;
                const venc_slice_out so = {byte_off, nbytes, nbits, 0};
                slice_dev[(size_t)item * g.n_slices + k] = so;
                host_tab[k] = so;
                byte_off += (nbytes + 3u) & ~3u;
            }
            const venc_out_header h = {byte_off, (uint32_t)g.n_slices, byte_off > B.out_cap ? 1u : 0u, 0};
            *(venc_out_header *)slot = h;
        }
    }
}

/* ================================================================== k_pack
 * Owner-gather: output word w of a slice (stream bits [32w, 32w+32)) is built by
 * the macroblock that contains bit 32w, reading on into the following
 * macroblocks' bit strings as needed.  No atomics, no pre-zeroed destination;
 * the destination is the HBM staging buffer of the output slot (or, in one-item engines, the device view of pinned host memory). */
DEVFN uint32_t fetch_bits(const uint32_t *wds, int pos, int n) { /* 1 <= n <= 32 */
    const int wi = pos >> 5, shb = pos & 31;
    const uint64_t two = ((uint64_t)wds[wi] << 32) | (uint64_t)((shb + n > 32) ? wds[wi + 1] : 0u);
    return (uint32_t)((two >> (64 - shb - n)) & (n == 32 ? 0xffffffffull : ((1ull << n) - 1ull)));
}

#define PACK_MBS 256
/* I pictures (P pictures: k_pack_p, venc_cavlc_p.hip).  One block per run of up to 256 consecutive macroblocks of a slice; consecutive
 * lanes build consecutive output words (coalesced stores).  The bit offsets and lengths of the run sit in LDS; a word finds its
 * first macroblock by binary search there. */
__global__ __launch_bounds__(256) void k_pack(venc_geom g, venc_bufs B, venc_step S, const venc_slice_out *slice_dev) {
    __shared__ uint32_t s_off[PACK_MBS];
    __shared__ uint32_t s_len[PACK_MBS];
    const int item = blockIdx.z, s = S.slice0 + blockIdx.y, tid = threadIdx.x;
    const int first = B.slice_first_row[s] * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    const int m0 = first + blockIdx.x * PACK_MBS;
    if (m0 >= last) return;
    const int n_here = min(PACK_MBS, last - m0);
    const venc_slice_out so = slice_dev[(size_t)item * g.n_slices + s];
    const uint32_t *lenp = B.mbnbits + (size_t)item * g.nmb;
    const uint32_t *offp = B.mboff + (size_t)item * g.nmb;
    const uint32_t *bits = B.mbbits + (size_t)item * g.nmb * VENC_MB_WORDS;
    uint32_t *out = (uint32_t *)(B.out_data + (size_t)S.out_slot[item] * B.out_data_stride + so.byte_off);
    if (tid < n_here) {
        s_off[tid] = offp[m0 + tid];
        s_len[tid] = lenp[m0 + tid];
    }
    __syncthreads();
    const uint32_t T = so.nbits;
    const uint32_t begin = s_off[0];
    const uint32_t end = (m0 + n_here == last) ? so.nbytes * 8u : s_off[n_here - 1] + s_len[n_here - 1];
    for (uint32_t w = ((begin + 31u) >> 5) + (uint32_t)tid; 32u * w < end; w += 256u) {
        if (so.byte_off + 4u * (w + 1u) > B.out_cap) break;
        uint32_t pos = 32u * w, val = 0;
        int lo = 0, hi = n_here - 1;
        while (lo < hi) { /* last macroblock of the run whose offset is <= pos */
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= pos)
                lo = mid;
            else
                hi = mid - 1;
        }
        int got = 0, m = lo;
        while (got < 32 && m0 + m < last) {
            uint32_t mo, ml;
            if (m < n_here)
                mo = s_off[m], ml = s_len[m];
            else
                mo = offp[m0 + m], ml = lenp[m0 + m]; /* the word runs on into the next block's macroblocks */
            const uint32_t local = pos - mo;
            if (local >= ml) { /* past this macroblock's last bit */
                m++;
                continue;
            }
            uint32_t take = ml - local;
            if (take > (uint32_t)(32 - got)) take = (uint32_t)(32 - got);
            const uint32_t v = fetch_bits(bits + (size_t)(m0 + m) * VENC_MB_WORDS, (int)local, (int)take);
            val = take == 32 ? v : ((val << take) | v);
            got += (int)take;
            pos += take;
        }
        if (got < 32) {
            if (pos == T) { /* rbsp_stop_one_bit */
                val = (val << 1) | 1u;
                got++;
            }
            val = got ? (val << (32 - got)) : 0u;
        }
        out[w] = __builtin_bswap32(val);
    }
}

/* ================================================================== k_band_copy
 * Slice-parallel multi-GPU mode (BASELINE config 5): moves the rows of a
 * band of the last reconstruction between the frame buffers and a contiguous exchange chunk
 * (the unit handed to RCCL).  A band is whole MB rows, so each plane contributes one contiguous
 * run of bytes (whole band) or two (the top and bottom halo rows only); the launcher describes
 * them as up to six segments.  grid = (blocks, segment, item); 16 bytes per lane per trip. */
typedef struct {
    uint32_t frame_off, chunk_off, bytes; /* all multiples of 16 */
} venc_band_seg;
typedef struct {
    int n_seg, to_chunk;
    size_t item_stride; /* bytes between the chunks of consecutive items */
    venc_band_seg seg[6];
} venc_band_job;

__global__ __launch_bounds__(256) void k_band_copy(venc_bufs B, int parity, venc_band_job J, uint8_t *chunk) {
    const int item = blockIdx.z;
    const venc_band_seg sg = J.seg[blockIdx.y];
    uint4 *f = (uint4 *)(B.recon + (size_t)(parity * B.max_items + item) * B.recon_frame_bytes + sg.frame_off);
    uint4 *c = (uint4 *)(chunk + (size_t)item * J.item_stride + sg.chunk_off);
    const uint32_t n = sg.bytes >> 4;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        if (J.to_chunk)
            c[i] = f[i];
        else
            f[i] = c[i];
    }
}
