/*
 * venc_inter2.hip -- P_L0_16x16 transform / quantisation / reconstruction with every lane busy
 * (SURVEY.md section 8a rows a4.2, a4.3, a4.5).
 *
 * The first version (k_inter in venc_kernels.hip) gave one half-wave to each macroblock: 16 luma
 * lanes, 8 chroma lanes, 8 idle, and the luma and chroma prediction paths ran one after the
 * other under divergence -- ~415 wave instructions per macroblock.  Here luma and chroma are
 * separate blocks of one launch (k_inter_fused) with homogeneous waves:
 *   luma blocks     16 lanes per macroblock, 4 macroblocks per wave   (lane = luma4x4BlkIdx)
 *   chroma blocks    8 lanes per macroblock, 8 macroblocks per wave   (lane = plane*4 + block)
 * Quantiser constants are wave-uniform (SGPRs), no per-lane selects.  The luma and chroma halves of
 * coded_block_pattern go to separate bytes of the macroblock record (they are written concurrently).
 */

/* Byte k of a packed dword as an operand, without unpacking it first (SDWA source selects; the compiler's own SDWA
 * peephole leaves these unpacks alone): cur - pred and pred + residual straight from the loaded dwords save 32 of the
 * ~450 instructions of a luma lane. */
template <int K>
DEVFN int byte_sub_byte(uint32_t a, uint32_t b) { /* a.byte[K] - b.byte[K] */
    int r;
    if (K == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    if (K == 1) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b));
    if (K == 2) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b));
    if (K == 3) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int K>
DEVFN int byte_sub_int(uint32_t a, int b) { /* a.byte[K] - b */
    int r;
    if (K == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (K == 1) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (K == 2) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (K == 3) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int K>
DEVFN int int_add_byte(int a, uint32_t b) { /* a + b.byte[K] */
    int r;
    if (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    if (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b));
    if (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b));
    if (K == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* quantisation, dequantisation and inverse transform of one 4x4 block: levels (scan order) and the residual rr */
template <bool DC_SEP>
DEVFN void blk_quant_res(const int w[16], const venc_qparams &q, int f, int d0, int lv[16], int rr[16], int &nnz) {
    int d[16];
    nnz = 0;
    /* sign(w) * ((|w| mf + f) >> s) without the magnitude: for w < 0 it equals (w mf + 2^s - 1 - f) >> s with an
     * arithmetic shift (-(x >> s) = (-x + 2^s - 1) >> s for x >= 0), so a level costs a select, one v_mad_i32_i24 and
     * a shift instead of six instructions.  |w| <= 9180 (4x4 transform of 9-bit residuals), mf <= 13107: inside 2^27.
     * Dequantisation: one 24-bit multiply by v << per (<= 29 << 8). */
    const int f_neg = (1 << q.qbits) - 1 - f;
    const int vs0 = q.v[0] << q.per, vs1 = q.v[1] << q.per, vs2 = q.v[2] << q.per;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = venc_zigzag_ct(i);
        const int cls = venc_pos_class(r);
        const int wv = w[r];
        int l = (VENC_MUL24(wv, q.mf[cls]) + (wv < 0 ? f_neg : f)) >> q.qbits;
        if (DC_SEP && i == 0) l = 0;
        lv[i] = l;
        nnz += (l != 0);
        d[r] = VENC_MUL24(l, cls == 0 ? vs0 : cls == 1 ? vs1 : vs2); /* |l| < 2^16 */
    }
    if (DC_SEP) d[0] = d0;
    venc_idct4(d, rr);
}
/* ZERO BLOCKS.  Most 4x4 blocks of a P picture quantise to nothing (a macroblock of the 1080p clip averages 0.2 residual
 * syntax units), and for those everything after the forward transform -- 16 quantisations, dequantisations, the inverse
 * transform, 16 clips: ~280 of the ~450 instructions of a block -- reproduces the prediction.  The test is exact, not a
 * SAD heuristic: level(w) = 0 <=> |w| <= zt[class] (the quantiser's own arithmetic, venc_qparams.zt_inter), i.e.
 * (unsigned)(w + zt) <= 2 zt; one add per coefficient, a running maximum per class.  A wave whose blocks are all zero takes
 * the short path (the decision must be wave-uniform to save anything). */
template <bool DC_SEP>
DEVFN bool blk_all_zero(const int w[16], const venc_qparams &q) {
    uint32_t m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int r = DC_SEP ? 1 : 0; r < 16; r++) {
        const int cls = venc_pos_class(r);
        const uint32_t u = (uint32_t)(w[r] + (cls == 0 ? q.zt_inter[0] : cls == 1 ? q.zt_inter[1] : q.zt_inter[2]));
        if (cls == 0) m0 = max(m0, u);
        if (cls == 1) m1 = max(m1, u);
        if (cls == 2) m2 = max(m2, u);
    }
    return m0 <= 2u * (uint32_t)q.zt_inter[0] && m1 <= 2u * (uint32_t)q.zt_inter[1] && m2 <= 2u * (uint32_t)q.zt_inter[2];
}

/* four rows of a current 4x4 block as packed dwords (aligned loads inside the display area, clamped bytes on its edges) */
DEVFN void load_cur4x4_pk(const uint8_t *cur, int cstride, int pw, int ph, int px, int py, bool aligned, uint32_t cq[4]) {
    if (aligned && px + 4 <= pw && py + 4 <= ph) {
#pragma unroll
        for (int y = 0; y < 4; y++) cq[y] = *(const uint32_t *)(cur + (size_t)(py + y) * cstride + px);
    } else {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint8_t *row = cur + (size_t)min(py + y, ph - 1) * cstride;
            cq[y] = (uint32_t)row[min(px, pw - 1)] | ((uint32_t)row[min(px + 1, pw - 1)] << 8) | ((uint32_t)row[min(px + 2, pw - 1)] << 16) |
                    ((uint32_t)row[min(px + 3, pw - 1)] << 24);
        }
    }
}

DEVFN void store_levels(int16_t *dst16, const int lv[16]) {
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = ((uint32_t)lv[2 * i] & 0xffffu) | ((uint32_t)lv[2 * i + 1] << 16);
    uint4 *dst = (uint4 *)dst16;
    dst[0] = make_uint4(p[0], p[1], p[2], p[3]);
    dst[1] = make_uint4(p[4], p[5], p[6], p[7]);
}
DEVFN void store_rec4x4(uint8_t *rec, int rstride, int px, int py, const int r[16]) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t v = (uint32_t)r[4 * y] | ((uint32_t)r[4 * y + 1] << 8) | ((uint32_t)r[4 * y + 2] << 16) | ((uint32_t)r[4 * y + 3] << 24);
        *(uint32_t *)(rec + (size_t)(py + y) * rstride + px) = v;
    }
}
DEVFN void unpack4(uint32_t v, int *o) {
    o[0] = (int)(v & 255u), o[1] = (int)((v >> 8) & 255u), o[2] = (int)((v >> 16) & 255u), o[3] = (int)(v >> 24);
}
/* current 4x4 block: aligned dword rows inside the display area, clamped bytes on its edges */

/* residual, transform, quantisation, reconstruction and stores of one luma 4x4 block per lane from its current rows cq[]
 * and predicted rows pq[] (four samples per dword): 16 lanes per macroblock, lane & 15 = luma4x4BlkIdx, lane & 48 = the
 * wave's group of 16; every lane of the wave calls this (ballots), `valid` guards the stores */
/* quantiser constants of this step's pictures: the encoder's fixed ones, or under rate control those of the QP the search
 * chose for (step, item) (B.qp_cur, written by the first tile of k_me_pool before any kernel that calls this runs) */
/* one row of the QP table into scalar registers: the row index is uniform, but loaded through a plain pointer the 13
 * constants land in vector registers (26 per pair of tables -- the fused kernels then spill around their transform) */
DEVFN venc_qparams qparams_uniform(const venc_qparams *row) {
    venc_qparams q;
    const int *src = (const int *)row;
    int *dst = (int *)&q;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(venc_qparams) / sizeof(int)); i++) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return q;
}
DEVFN void step_qparams(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, venc_qparams &ql, venc_qparams &qc) {
    if (S.rc_mode) {
        const int qp = __builtin_amdgcn_readfirstlane(B.qp_cur[item]); /* uniform: the tables load into scalar registers */
        ql = qparams_uniform(B.qtab + qp);
        qc = qparams_uniform(B.qtab + venc_chroma_qp(qp));
    } else {
        ql = g.ql;
        qc = g.qc;
    }
}

/* Tried in round 3 and dropped: coding only the few blocks of a wave that hold a level, sixteen lanes per
 * block with the inverse transform by wave shuffles (four blocks per pass, handed over through LDS).  ~170 fewer vector
 * instructions per transforming wave, bit-exact -- and 2 % SLOWER on the whole pass (same-box A/B 144.3 -> 141.6 Gpx/s; k_me_fix
 * 0.50 -> 0.59 ms per pass): the shuffles' latency lengthens the tail of every block, which is what bounds these kernels. */
DEVFN void luma_block_code(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int lane, int mb, bool valid, const uint32_t cq[4],
                           const uint32_t pq[4], const venc_qparams &ql) {
    const int blk = lane & 15, gbase = lane & 48;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const int px = mx * 16 + blk_x(blk) * 4, py = my * 16 + blk_y(blk) * 4;
    int wstride;
    uint8_t *rec = rec_plane(B, g, S.parity, item, 0, wstride);
    int res[16], w[16], lv[16], rr[16], nnz;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        res[4 * y] = byte_sub_byte<0>(cq[y], pq[y]);
        res[4 * y + 1] = byte_sub_byte<1>(cq[y], pq[y]);
        res[4 * y + 2] = byte_sub_byte<2>(cq[y], pq[y]);
        res[4 * y + 3] = byte_sub_byte<3>(cq[y], pq[y]);
    }
    venc_fdct4(res, w);
    venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
    const unsigned long long nzb = __ballot(valid && !blk_all_zero<false>(w, ql));
    if (nzb == 0ull) {
        /* every block of the wave quantises to zero: no levels, the reconstruction is the prediction */
        if (valid) {
            info->nnz[blk_y(blk) * 4 + blk_x(blk)] = 0;
#pragma unroll
            for (int y = 0; y < 4; y++) *(uint32_t *)(rec + (size_t)(py + y) * wstride + px) = pq[y];
            if (blk == 0) {
                info->cbp = 0;
                info->flags = 0;
            }
        }
        return;
    }
    blk_quant_res<false>(w, ql, ql.f_inter, 0, lv, rr, nnz);
#pragma unroll
    for (int y = 0; y < 4; y++) {
        rr[4 * y] = venc_clip_u8(int_add_byte<0>(rr[4 * y], pq[y]));
        rr[4 * y + 1] = venc_clip_u8(int_add_byte<1>(rr[4 * y + 1], pq[y]));
        rr[4 * y + 2] = venc_clip_u8(int_add_byte<2>(rr[4 * y + 2], pq[y]));
        rr[4 * y + 3] = venc_clip_u8(int_add_byte<3>(rr[4 * y + 3], pq[y]));
    }

    int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
    const unsigned long long bal = __ballot(nnz > 0);
    if (valid) {
        if (nnz) store_levels(lv_mb + blk * 16, lv);
        info->nnz[blk_y(blk) * 4 + blk_x(blk)] = (uint8_t)nnz;
        store_rec4x4(rec, wstride, px, py, rr);
        if (blk == 0) {
            const uint32_t h = (uint32_t)(bal >> gbase) & 0xffffu;
            int cbpl = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                if ((h >> (4 * qd)) & 15u) cbpl |= 1 << qd;
            info->cbp = (uint8_t)cbpl; /* the chroma part has its own byte (cbpc) */
            info->flags = 0;
        }
    }
}

/* DEFERRED BLOCKS (round 3, the shipped P5 of k_me_pool).  94-96 % of the clip's 4x4 blocks quantise to nothing, yet 80-94 % of
 * the transforming waves hold at least one block that does not, so nearly every wave used to walk the ~280 instructions of
 * quantisation ... reconstruction for three or four busy lanes.  Here a wave only decides: residual, forward transform and the
 * exact zero test for all its blocks; a block without a level is finished (its reconstruction is the prediction); the few blocks
 * WITH a level are listed -- entry = lane | cx << 6 | cy << 12 in the (tile, wave)'s own 64-entry segment, the count as a plain
 * store: no atomic, nothing the wave waits for -- and coded by extra blocks of the k_me_fix launch that follows on the same stream
 * (luma_nz_redo below), beside the exhaustive redo on a chip that launch leaves nearly idle.  Unlike the two in-kernel variants
 * tried before (second pass behind a barrier; sixteen lanes per listed block) nothing is appended to the tail of the search block. */
#define NZ_RPB 128 /* (tile, wave) segments per follow-up block of 256 threads (VENC_NZ_RPB): 16 / 64 blocks per 1080p / 2160p item; measured 32 / 64 / 128 */
DEVFN void luma_block_defer(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int lane, int mb, bool valid, const uint32_t cq[4],
                            const uint32_t pq[4], const venc_qparams &ql, uint32_t *seg, uint4 *seg_rows, uint32_t *seg_count, uint32_t mvbits) {
    const int blk = lane & 15;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const int px = mx * 16 + blk_x(blk) * 4, py = my * 16 + blk_y(blk) * 4;
    int wstride;
    uint8_t *rec = rec_plane(B, g, S.parity, item, 0, wstride);
    int res[16], w[16];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        res[4 * y] = byte_sub_byte<0>(cq[y], pq[y]);
        res[4 * y + 1] = byte_sub_byte<1>(cq[y], pq[y]);
        res[4 * y + 2] = byte_sub_byte<2>(cq[y], pq[y]);
        res[4 * y + 3] = byte_sub_byte<3>(cq[y], pq[y]);
    }
    venc_fdct4(res, w);
    venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
    const bool nz = valid && !blk_all_zero<false>(w, ql);
    const unsigned long long nzb = __ballot(nz);
    if (valid) {
        info->nnz[blk_y(blk) * 4 + blk_x(blk)] = 0; /* (a listed block's count and its bit of cbp come from the follow-up) */
        if (!nz) {
#pragma unroll
            for (int y = 0; y < 4; y++) *(uint32_t *)(rec + (size_t)(py + y) * wstride + px) = pq[y];
        }
        if (blk == 0) {
            info->cbp = 0;
            info->flags = 0;
        }
    }
    if (nz) { /* the entry and, beside it, the block's current and predicted rows: the follow-up reads them as two coalesced 16-byte pieces
                 instead of twelve scattered dwords per lane (scattered, its loads took 25 us per 2160p launch: one cache line per lane) */
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzb, 0u));
        seg[slot] = (uint32_t)lane | mvbits;
        seg_rows[2 * slot] = make_uint4(cq[0], cq[1], cq[2], cq[3]);
        seg_rows[2 * slot + 1] = make_uint4(pq[0], pq[1], pq[2], pq[3]);
    }
    if (lane == 0) *seg_count = (uint32_t)__popcll(nzb);
}

/* one listed block, any lane: its current and predicted rows as the search left them beside the entry, everything from the residual
 * to the stores; its bit of the macroblock's coded_block_pattern is OR-ed into the record's word (the search's P5 zeroed it; the
 * chroma byte of that word is written by a later kernel of the side chain) */
DEVFN void luma_block_redo(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int mx, int my, int blk, const uint4 cur4, const uint4 pred4,
                           const venc_qparams &ql) {
    const int mb = my * g.mbw + mx;
    const int px = mx * 16 + blk_x(blk) * 4, py = my * 16 + blk_y(blk) * 4;
    int wstride;
    uint8_t *rec = rec_plane(B, g, S.parity, item, 0, wstride);
    const uint32_t cq[4] = {cur4.x, cur4.y, cur4.z, cur4.w}, pq[4] = {pred4.x, pred4.y, pred4.z, pred4.w};
    int res[16], w[16], lv[16], rr[16], nnz;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        res[4 * y] = byte_sub_byte<0>(cq[y], pq[y]);
        res[4 * y + 1] = byte_sub_byte<1>(cq[y], pq[y]);
        res[4 * y + 2] = byte_sub_byte<2>(cq[y], pq[y]);
        res[4 * y + 3] = byte_sub_byte<3>(cq[y], pq[y]);
    }
    venc_fdct4(res, w);
    blk_quant_res<false>(w, ql, ql.f_inter, 0, lv, rr, nnz);
#pragma unroll
    for (int y = 0; y < 4; y++) {
        rr[4 * y] = venc_clip_u8(int_add_byte<0>(rr[4 * y], pq[y]));
        rr[4 * y + 1] = venc_clip_u8(int_add_byte<1>(rr[4 * y + 1], pq[y]));
        rr[4 * y + 2] = venc_clip_u8(int_add_byte<2>(rr[4 * y + 2], pq[y]));
        rr[4 * y + 3] = venc_clip_u8(int_add_byte<3>(rr[4 * y + 3], pq[y]));
    }
    venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
    if (nnz) {
        store_levels(B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS + blk * 16, lv);
        atomicOr((uint32_t *)&info->cbp, 1u << (blk >> 2)); /* (cbp is the low byte of a 4-aligned word: static_assert in venc_hip.hip) */
    }
    info->nnz[blk_y(blk) * 4 + blk_x(blk)] = (uint8_t)nnz;
    store_rec4x4(rec, wstride, px, py, rr);
}

/* luma transform of macroblock mb with vector (dx, dy), current and predicted samples fetched from memory */
DEVFN void inter_luma_mb(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int lane, int mb, bool valid, int dx, int dy,
                         const venc_qparams &ql) {
    const int blk = lane & 15;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const int px = mx * 16 + blk_x(blk) * 4, py = my * 16 + blk_y(blk) * 4;
    int cstride, pw, ph, rstride;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);

    uint32_t cq[4], pq[4]; /* current and predicted rows, four samples per dword */
    load_cur4x4_pk(cur, cstride, pw, ph, px, py, (g.w & 3) == 0, cq);
    const int x0 = px + dx, y0 = py + dy;
    if (x0 >= 0 && x0 + 8 <= g.cw && y0 >= 0 && y0 + 4 <= g.ch) { /* (integer vectors only: under sub-sample refinement k_subpel transforms luma itself) */
        const uint32_t sh = (uint32_t)(x0 & 3);
        const uint8_t *rp = ref + (size_t)y0 * rstride + (x0 & ~3);
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t *r32 = (const uint32_t *)(rp + (size_t)y * rstride);
            pq[y] = __builtin_amdgcn_alignbyte(r32[1], r32[0], sh);
        }
    } else {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint8_t *row = ref + (size_t)clampi(y0 + y, 0, g.ch - 1) * rstride;
            pq[y] = (uint32_t)row[clampi(x0, 0, g.cw - 1)] | ((uint32_t)row[clampi(x0 + 1, 0, g.cw - 1)] << 8) |
                    ((uint32_t)row[clampi(x0 + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(x0 + 3, 0, g.cw - 1)] << 24);
        }
    }
    luma_block_code(g, B, S, item, lane, mb, valid, cq, pq, ql);
}

/* chroma transform of macroblock mb: 8 lanes, lane & 3 = block, (lane >> 2) & 1 = plane, lane & 56 = group of 8 */
struct chroma_listed { /* a chroma block that holds an AC level, between the two passes of a k_inter_fused block (LDS) */
    uint4 cq, pq; /* current and predicted rows, four samples per dword */
    int d0;       /* its dequantised DC (the 2x2 transform spans the plane's four blocks: decided in the first pass) */
    uint32_t id;  /* mb << 3 | plane << 2 | block */
};
DEVFN void inter_chroma_mb(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int lane, int mb, bool valid, int dx, int dy,
                           const venc_qparams &qc, chroma_listed *cl, uint32_t *cl_count) {
    const int k = lane & 3, plane = 1 + ((lane >> 2) & 1), qbase = lane & ~3, gbase = lane & 56;
    int mx, my;
    mb_xy(g, mb, mx, my);
    const int px = mx * 8 + (k & 1) * 4, py = my * 8 + (k >> 1) * 4;
    int cstride, pw, ph, rstride, wstride;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], plane, cstride, pw, ph);
    const uint8_t *ref = rec_plane(B, g, S.ref, item, plane, rstride);
    uint8_t *rec = rec_plane(B, g, S.parity, item, plane, wstride);
    const int rw = g.cw >> 1, rh = g.ch >> 1;

    int p[16];
    uint32_t cq[4];
    load_cur4x4_pk(cur, cstride, pw, ph, px, py, (g.w & 7) == 0, cq);
    /* chroma MC, 8.4.2.2.2: integer luma mv -> eighth-sample chroma mv (frac 0 or 4); under sub-sample refinement the
     * vector arrives in quarter luma = eighth chroma samples */
    const int x0 = px + (g.subpel ? dx >> 3 : dx >> 1), y0 = py + (g.subpel ? dy >> 3 : dy >> 1);
    const bool fx = (dx & 1) != 0, fy = (dy & 1) != 0;
    /* rows of the reference as packed samples: ra[y] = samples x0..x0+3, rb[y] = x0+1..x0+4 */
    uint32_t ra[5], rb[5];
    if (x0 >= 0 && x0 + 8 <= rw && y0 >= 0 && y0 + 5 <= rh) {
        const uint32_t sh8 = (uint32_t)(x0 & 3) * 8u;
        const uint8_t *rp = ref + (size_t)y0 * rstride + (x0 & ~3);
#pragma unroll
        for (int y = 0; y < 5; y++) {
            const uint32_t *r32 = (const uint32_t *)(rp + (size_t)y * rstride);
            const uint64_t two = ((uint64_t)r32[1] << 32) | r32[0];
            ra[y] = (uint32_t)(two >> sh8);
            rb[y] = (uint32_t)(two >> (sh8 + 8u)); /* shift <= 32: the fifth sample is byte 3 of the upper dword at most */
        }
    } else {
#pragma unroll
        for (int y = 0; y < 5; y++) {
            const uint8_t *row = ref + (size_t)clampi(y0 + y, 0, rh - 1) * rstride;
            uint32_t s5[5];
#pragma unroll
            for (int x = 0; x < 5; x++) s5[x] = row[clampi(x0 + x, 0, rw - 1)];
            ra[y] = s5[0] | (s5[1] << 8) | (s5[2] << 16) | (s5[3] << 24);
            rb[y] = s5[1] | (s5[2] << 8) | (s5[3] << 16) | (s5[4] << 24);
        }
    }
    /* 8.4.2.2.2 with fractions in {0, 4}: ((8-xf)(8-yf) A + xf (8-yf) B + (8-xf) yf C + xf yf D + 32) >> 6 is A, (A+B+1)>>1,
     * (A+C+1)>>1 or (A+B+C+D+2)>>2 -- in every case (h(y) + (fy ? h(y+1) : h(y)) + 2) >> 2 with the horizontal pair sum
     * h = A + (fx ? B : A).  The pair sums come straight from the packed rows (one select per ROW between the row and the
     * row one sample on, then byte-select adds); one select, one three-operand add and a shift per sample remain.  No
     * quarter-rate multiplies, no unpacking. */
    if (g.subpel) {
        /* any of the 64 fractions: the bilinear formula as it stands (24-bit multiplies) */
        const int xf = dx & 7, yf = dy & 7;
        const int wa = (8 - xf) * (8 - yf), wb = xf * (8 - yf), wc = (8 - xf) * yf, wd = xf * yf;
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int A = (int)((ra[y] >> (8 * x)) & 255u), Bv = (int)((rb[y] >> (8 * x)) & 255u);
                const int Cv = (int)((ra[y + 1] >> (8 * x)) & 255u), D = (int)((rb[y + 1] >> (8 * x)) & 255u);
                p[4 * y + x] = (VENC_MUL24(wa, A) + VENC_MUL24(wb, Bv) + VENC_MUL24(wc, Cv) + VENC_MUL24(wd, D) + 32) >> 6;
            }
    } else {
    int hs[5][4];
#pragma unroll
    for (int y = 0; y < 5; y++) {
        const uint32_t rs = fx ? rb[y] : ra[y];
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(hs[y][0]) : "v"(ra[y]), "v"(rs));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(hs[y][1]) : "v"(ra[y]), "v"(rs));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(hs[y][2]) : "v"(ra[y]), "v"(rs));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(hs[y][3]) : "v"(ra[y]), "v"(rs));
    }
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) p[4 * y + x] = (hs[y][x] + (fy ? hs[y + 1][x] : hs[y][x]) + 2) >> 2;
    }

    int res[16], w[16];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        res[4 * y] = byte_sub_int<0>(cq[y], p[4 * y]);
        res[4 * y + 1] = byte_sub_int<1>(cq[y], p[4 * y + 1]);
        res[4 * y + 2] = byte_sub_int<2>(cq[y], p[4 * y + 2]);
        res[4 * y + 3] = byte_sub_int<3>(cq[y], p[4 * y + 3]);
    }
    venc_fdct4(res, w);
    /* 2x2 DC transform over the four lanes of the plane */
    int yc = 0;
#pragma unroll
    for (int qi = 0; qi < 4; qi++) {
        const int v = __shfl(w[0], qbase | qi);
        yc += (__popc(k & qi) & 1) ? -v : v;
    }
    const int dclv = venc_quant_chroma_dc(yc, &qc, 0);
    int fc = 0;
#pragma unroll
    for (int qi = 0; qi < 4; qi++) {
        const int v = __shfl(dclv, qbase | qi);
        fc += (__popc(k & qi) & 1) ? -v : v;
    }
    const int d0 = venc_dequant_chroma_dc(fc, &qc);
    /* DEFERRED FORM (round 4, as the luma transform's since round 3): this pass only DECIDES.  A block without an AC level --
     * 85-97 % of them -- is finished here: the inverse transform of a lone DC is (d0 + 32) >> 6 in every sample.  A block WITH one
     * is listed in LDS (its current rows, its prediction packed, its DC) and coded by the first lanes of the block after a barrier
     * (inter_chroma_listed).  Before, a wave took the short path only when all 64 of its blocks were empty, i.e. rarely where it
     * mattered: 6.0 M vector instructions per 1080p step.  The exact zero test is the quantiser's own arithmetic (blk_all_zero), so
     * listed <=> nnz > 0 and coded_block_pattern is known here. */
    const bool nz = valid && !blk_all_zero<true>(w, qc);
    const unsigned long long bal_ac = __ballot(nz), bal_dc = __ballot(valid && dclv != 0);
    int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
    venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
    const int dc0 = __shfl(dclv, qbase | 0), dc1 = __shfl(dclv, qbase | 1), dc2 = __shfl(dclv, qbase | 2), dc3 = __shfl(dclv, qbase | 3);
    const int pk = (plane - 1) * 4 + k;
    if (valid) {
        if (!nz) {
            const int r0 = (d0 + 32) >> 6;
            int rr[16];
#pragma unroll
            for (int i = 0; i < 16; i++) rr[i] = venc_clip_u8(p[i] + r0);
            info->nnz[16 + pk] = 0;
            store_rec4x4(rec, wstride, px, py, rr);
        }
        if (k == 0) {
            uint4 *dst = (uint4 *)(lv_mb + (16 + plane) * 16);
            dst[0] = make_uint4(((uint32_t)dc0 & 0xffffu) | ((uint32_t)dc1 << 16), ((uint32_t)dc2 & 0xffffu) | ((uint32_t)dc3 << 16), 0, 0);
        }
        if ((lane & 7) == 0) {
            const uint32_t hac = (uint32_t)(bal_ac >> gbase) & 0xffu, hdc = (uint32_t)(bal_dc >> gbase) & 0xffu;
            info->cbpc = (uint8_t)(hac ? 2 : hdc ? 1 : 0);
        }
    }
    if (nz) {
        const uint32_t at = atomicAdd(cl_count, 1u);
        chroma_listed e;
        e.cq = make_uint4(cq[0], cq[1], cq[2], cq[3]);
        uint32_t pp[4];
#pragma unroll
        for (int y = 0; y < 4; y++) pp[y] = (uint32_t)p[4 * y] | ((uint32_t)p[4 * y + 1] << 8) | ((uint32_t)p[4 * y + 2] << 16) | ((uint32_t)p[4 * y + 3] << 24);
        e.pq = make_uint4(pp[0], pp[1], pp[2], pp[3]);
        e.d0 = d0;
        e.id = ((uint32_t)mb << 3) | (uint32_t)pk;
        cl[at] = e;
    }
}

/* one listed chroma block, any lane: residual ... reconstruction, its levels and its count */
DEVFN void inter_chroma_listed(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, const chroma_listed &e, const venc_qparams &qc) {
    const int mb = (int)(e.id >> 3), pk = (int)(e.id & 7u), plane = 1 + (pk >> 2), k = pk & 3;
    int mx, my;
    mb_xy(g, mb, mx, my);
    const int px = mx * 8 + (k & 1) * 4, py = my * 8 + (k >> 1) * 4;
    int wstride;
    uint8_t *rec = rec_plane(B, g, S.parity, item, plane, wstride);
    const uint32_t cq[4] = {e.cq.x, e.cq.y, e.cq.z, e.cq.w}, pq[4] = {e.pq.x, e.pq.y, e.pq.z, e.pq.w};
    int res[16], w[16], lv[16], rr[16], nnz;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        res[4 * y] = byte_sub_byte<0>(cq[y], pq[y]);
        res[4 * y + 1] = byte_sub_byte<1>(cq[y], pq[y]);
        res[4 * y + 2] = byte_sub_byte<2>(cq[y], pq[y]);
        res[4 * y + 3] = byte_sub_byte<3>(cq[y], pq[y]);
    }
    venc_fdct4(res, w);
    blk_quant_res<true>(w, qc, qc.f_inter, e.d0, lv, rr, nnz);
#pragma unroll
    for (int y = 0; y < 4; y++) {
        rr[4 * y] = venc_clip_u8(int_add_byte<0>(rr[4 * y], pq[y]));
        rr[4 * y + 1] = venc_clip_u8(int_add_byte<1>(rr[4 * y + 1], pq[y]));
        rr[4 * y + 2] = venc_clip_u8(int_add_byte<2>(rr[4 * y + 2], pq[y]));
        rr[4 * y + 3] = venc_clip_u8(int_add_byte<3>(rr[4 * y + 3], pq[y]));
    }
    store_levels(B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS + (19 + pk) * 16, lv);
    (B.mbinfo + (size_t)item * g.nmb + mb)->nnz[16 + pk] = (uint8_t)nnz;
    store_rec4x4(rec, wstride, px, py, rr);
}

/* ================================================================== k_inter_fused
 * The luma and the chroma transform of a P step as independent blocks of ONE launch (two back-to-back
 * kernels cost a ramp-down and a ramp-up, and the chroma blocks need nothing from the luma blocks):
 * the first n_luma blocks take 16 macroblocks each (luma), the rest 32 macroblocks each (chroma).
 *
 * Tried and dropped: a third kind of block in this launch that searched the flagged macroblocks
 * exhaustively and transformed them, to hide k_me_fix's serial 13 us.  The search's registers
 * (or, capped, its spills in the shared prologue) cost the transform blocks more occupancy than
 * the hidden kernel was worth (74 us against 13 + 44 us).
 * Also tried: k_me_fix and a list-driven copy of this kernel on a third stream beside this launch
 * (which then skipped the flagged macroblocks).  Bit-exact, but the two cross-stream event hand-offs
 * per step cost more than the hidden 15 us: 1080p 5.63 -> 6.32 ms per clip, 2160p 17.6 -> 18.4 ms.
 * And a second attempt at the third kind of block with a register-light search (a quarter-wave per candidate, 16
 * candidates per round, ranks in LDS, 69 rounds; the fix blocks of all items first in dispatch order, one listed
 * macroblock per block).  It fits the 80 VGPRs, but on a chip saturated with transform waves the search chain takes
 * ~45 us (12 us as a kernel of its own on an empty chip), longer than the 37 us transform, and its prologue spills
 * cost every block ~5 %: 1080p +1.7 %, 2160p -1.3 % -- noise.  Dropped. */
/* n_luma == 0: chroma only -- the shipped P step: the luma transform runs inside the motion search (k_me_pool P5, k_me_fix),
 * this launch follows on the second stream beside the next step's search.  n_luma > 0: both (the legacy test library's
 * searches do not transform). */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_inter_fused(venc_geom g, venc_bufs B, venc_step S, int n_luma) {
    const int item = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int bx = blockIdx.x;
    const uint32_t *mvs = B.mvsad + (size_t)item * g.nmb;
    venc_qparams ql, qc;
    step_qparams(g, B, S, item, ql, qc);
    if (bx >= n_luma) {
        __shared__ chroma_listed s_cl[256];
        __shared__ uint32_t s_cl_count;
        if (tid == 0) s_cl_count = 0;
        __syncthreads();
        const int mb_raw = S.mb0 + (bx - n_luma) * 32 + (tid >> 3);
        const bool valid = mb_raw < S.mb1;
        const int mb = valid ? mb_raw : S.mb1 - 1;
        const uint32_t m = mvs[mb];
        inter_chroma_mb(g, B, S, item, lane, mb, valid, (int)(int8_t)(m & 255u), (int)(int8_t)((m >> 8) & 255u), qc, s_cl, &s_cl_count);
        __syncthreads();
        if ((uint32_t)tid < s_cl_count) inter_chroma_listed(g, B, S, item, s_cl[tid], qc);
    } else {
        const int mb_raw = S.mb0 + bx * 16 + (tid >> 4);
        const bool valid = mb_raw < S.mb1;
        const int mb = valid ? mb_raw : S.mb1 - 1;
        const uint32_t m = mvs[mb];
        inter_luma_mb(g, B, S, item, lane, mb, valid, (int)(int8_t)(m & 255u), (int)(int8_t)((m >> 8) & 255u), ql);
    }
}
