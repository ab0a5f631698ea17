/*
 * venc_cavlc_p.hip -- entropy coding of P pictures on SYNTAX ENTRIES.
 *
 * An ENTRY is the smallest piece of a P macroblock's syntax that can be coded on its own, because every CAVLC context (nC)
 * comes from the TotalCoeff values the transform kernels already stored, not from coded bits:
 *   slot 0        R: slice header (first macroblock of a slice) and mb_skip_run -- for a skipped macroblock that ends its slice
 *                 the slice's last mb_skip_run
 *   slot 1        H: mb_type, mvd_l0, coded_block_pattern, mb_qp_delta of a coded macroblock
 *   slot 2 .. 17  luma 4x4 block luma4x4BlkIdx = slot - 2 (present iff its 8x8 block's coded_block_pattern bit is set)
 *   slot 18, 19   Cb, Cr DC                 slot 20 .. 27  Cb, Cr AC blocks
 * Slots are numbered in bitstream order, so the entries of a slice sorted by (macroblock, slot) ARE its slice_data().
 *
 * History.  Round 1 gave a macroblock to one lane (a wave lasted as long as its heaviest macroblock, ~100 us), then eight
 * lanes (seven idle).  Rounds 2-3 cut a macroblock into seven UNITS (header + luma 8x8 block 0, three more 8x8 blocks, chroma
 * DC, Cb AC, Cr AC), listed the non-empty ones with one atomic per block and gave a lane to each: 77 us per batch of four
 * steps at 28 % vector utilisation -- five of six listed units are headers, the sixth carries up to four residual blocks and
 * every wave held one of those; each unit wrote a word here and there into its own 256-byte string of a 2 KB region per
 * macroblock, k_scan turned eight unit lengths per macroblock into offsets and k_pack read ONE 128-BYTE LINE PER UNIT to collect
 * a few words: 98 MB read to write 1.5 MB (profiles/pmc_traffic.json of round 3).
 *
 * Round 4 (this file), three kernels, each one block per SEGMENT = 256 macroblocks of one slice:
 *   k_skip2     one lane per macroblock, which holds its own, its left and its upper record: P_Skip decision, the mask of coded
 *               macroblocks (slice-relative words), the nC of all 24 blocks at once (packed bytes), and the macroblock's entries
 *               in bitstream order -- a block scan, no atomic.  What needs nothing else is FINISHED here: the H entry and the
 *               coeff_token of every block without a level (most entries: a coded 8x8 block holds one or two blocks with
 *               levels); R entries and blocks with levels are listed for k_cavlc_p2 with their nC
 *   k_cavlc_p2  one lane per listed entry: mb_skip_run from the mask / one residual block from its 32 bytes of levels.  An
 *               entry's string leaves as ONE 16-byte store (venc_bitsink4: four words in registers; the rare entry beyond
 *               128 bits spills into the macroblock's region of mbbits); the block ends with an exclusive scan of all the
 *               segment's lengths = the entries' bit offsets
 *   k_pack_p    every block sums the picture's segment totals itself (a few dozen numbers: no scan kernel), ORs its entries
 *               into an LDS window at their bit offsets and streams the window out; the block of a picture's first segment
 *               also writes the slice table
 * The first form of this round gave every entry a lane of one mixed kernel: 14.0 M vector instructions per batch of four
 * 1080p steps (half a motion search) -- a wave walked the header path AND the longest block of its 64 entries for what was
 * mostly one-bit tokens.  I pictures (26 dense blocks in every macroblock) stay on k_cavlc / k_scan / k_pack (venc_kernels.hip).
 */

#define ENT_SEG_MBS 256   /* macroblocks of one slice per segment */
#define ENT_PER_MB 28     /* entries a macroblock can have */
#define ENT_THREADS 256

/* segment index of (pair v, slice s, segment j of the slice) */
DEVFN size_t ent_seg(const venc_geom &g, const venc_bufs &B, int v, int s, int j) { return ((size_t)v * g.n_slices + s) * B.ent_spp + j; }
/* record slot of an entry's levels (27 per macroblock: luma 0 .. 15, Intra16x16 DC, Cb / Cr DC, Cb / Cr AC) */
DEVFN int ent_level_slot(int slot) { return slot < 18 ? slot - 2 : slot - 1; }

DEVFN uint32_t nonzero_bytes4(uint32_t w) { /* bit k = byte k of w is not zero */
    return (uint32_t)((w & 0xffu) != 0u) | ((uint32_t)((w & 0xff00u) != 0u) << 1) | ((uint32_t)((w & 0xff0000u) != 0u) << 2) | ((uint32_t)((w >> 24) != 0u) << 3);
}
/* nC (9.2.1) of four blocks at once: A / B = the TotalCoeff of their left / upper neighbours, one per byte (<= 16), mA / mB =
 * 0xff where that neighbour is available: (A + B + 1) >> 1 with both, the one that is there, or 0 */
DEVFN uint32_t nc_bytes4(uint32_t A, uint32_t Bv, uint32_t mA, uint32_t mB) {
    const uint32_t both = ((A + Bv + 0x01010101u) >> 1) & 0x7f7f7f7fu;
    return (both & mA & mB) | (A & mA & ~mB) | (Bv & ~mA & mB);
}

/* P_Skip (7.3.4, 8.4.1.1, 8.4.1.3.1) from values already loaded: the macroblock's vector, its neighbours' A (left), B (top), and C (top
 * right, or D top left when C is missing) -- what mb_is_p_skip (venc_kernels.hip) decides with loads that depend on one another */
DEVFN bool p_skip_vals(const venc_geom &g, int mx, bool aL, bool aT, int cbp, uint32_t me, uint32_t mA, uint32_t mB, uint32_t mCD) {
    const bool aA = aL, aB = aT, aC = aT && (mx + 1 < g.mbw || mx > 0);
    const int ax = aA ? (int)(int8_t)(mA & 255u) : 0, ay = aA ? (int)(int8_t)((mA >> 8) & 255u) : 0;
    const int bxv = aB ? (int)(int8_t)(mB & 255u) : 0, byv = aB ? (int)(int8_t)((mB >> 8) & 255u) : 0;
    const int cx = aC ? (int)(int8_t)(mCD & 255u) : 0, cy = aC ? (int)(int8_t)((mCD >> 8) & 255u) : 0;
    const bool z = !aL || !aT || (ax == 0 && ay == 0) || (bxv == 0 && byv == 0);
    int pxv, pyv;
    if (!aB && !aC && aA)
        pxv = ax, pyv = ay;
    else if ((int)aA + (int)aB + (int)aC == 1)
        pxv = aA ? ax : aB ? bxv : cx, pyv = aA ? ay : aB ? byv : cy;
    else
        pxv = venc_median3(ax, bxv, cx), pyv = venc_median3(ay, byv, cy);
    const int dx = (int)(int8_t)(me & 255u), dy = (int)(int8_t)((me >> 8) & 255u);
    return cbp == 0 && (z ? (dx == 0 && dy == 0) : (dx == pxv && dy == pyv));
}

/* ================================================================== k_entropy_p
 * grid = (segments per slice, slices of the band, (step, item) pairs); block = 256 lanes = the segment's macroblocks.
 *
 * PHASE A, one lane per macroblock, which holds its own, its left and its upper record.  P_Skip (7.3.4: no residual and the
 * vector a decoder infers, 8.4.1.1); the coded-macroblock mask as wave ballots (LDS); nC of the macroblock's 24 blocks at once
 * (packed bytes); its entries in bitstream order, their ranks from ONE block scan (no atomic).  Entries that need nothing else
 * are FINISHED here -- H, and the coeff_token of every block without a level (most entries: a coded 8x8 block holds one or two
 * blocks with levels) -- the others (R, blocks with levels) are listed in LDS with their nC.  Meanwhile the first wave looks
 * BACKWARDS from the segment's start for the last coded macroblock before it (64 macroblocks per step, nearly always one step):
 * the only thing a segment needs from its predecessors, and what lets one launch do what k_skip + k_cavlc_p did in two.
 * PHASE B, one lane per listed entry: mb_skip_run from the mask / one residual block from its 32 bytes of levels.
 * PHASE C: exclusive scan of all lengths (LDS) = the entries' bit offsets inside the segment.
 * An entry's string leaves as ONE 16-byte store (venc_bitsink4: four words in registers; the rare entry beyond 128 bits spills
 * into the macroblock's region of mbbits and says so in ent_list).
 * Dynamic LDS: 36 bytes per lane (the 16 levels of the block being coded + a dword against bank conflicts), 2 bytes x ENT_LEN_LDS
 * (lengths), 8 bytes x ENT_CLS_LDS (the list of phase B: record + the empty blocks behind the entry); longer segments continue in memory. */
#ifndef ENT_CLS_LDS
#define ENT_CLS_LDS 1024 /* entries of phase B's list kept in LDS */
#endif
#ifndef ENT_LEN_LDS
#define ENT_LEN_LDS 3072 /* entries whose lengths are kept in LDS (12 per macroblock; a segment of the 1080p clip has 600 .. 2600) */
#endif
/* (a block of this kernel holds the place of a motion-search block on its CU -- the search takes a CU's whole LDS -- so its LDS is
 * sized for ordinary pictures, 20 KB, not for the worst case, 40 KB; what does not fit goes through memory: same results) */
/* Diagnostic build only (-DVENC_ENT_STAMPS, tools/ent_stamps.py): the first wave of a block keeps the 100 MHz clock at eight points in registers
 * and leaves them as ONE record of eight words (ring of VENC_STAMP_PAIRS / 4 records behind B.stamps' eight sums; one atomic per block) */
#ifdef VENC_ENT_STAMPS
#define ENT_STAMP(i) do { t_st_[i] = wall_clock64(); } while (0)
#define ENT_STAMP_INIT unsigned long long t_st_[8]; const unsigned long long t_start_ = wall_clock64()
#define ENT_STAMP_END                                                                                                         \
    do {                                                                                                                      \
        if (threadIdx.x == 0) {                                                                                               \
            const unsigned long long n_ = atomicAdd(&B.stamps[0], 1ull);                                                      \
            if (n_ < VENC_STAMP_PAIRS / 4) {                                                                                  \
                unsigned long long *r_ = B.stamps + 8 + 8 * n_;                                                               \
                r_[0] = t_start_;                                                                                             \
                r_[1] = (t_st_[0] - t_start_) | ((unsigned long long)count << 40);                                            \
                for (int i_ = 1; i_ < 7; i_++) r_[1 + i_] = t_st_[i_] - t_st_[i_ - 1];                                        \
                r_[7] += t_st_[7] - t_st_[6]; /* (B and its barrier in one word: r_[7] holds 6 -> 7 on top of 5 -> 6) */      \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
#else
#define ENT_STAMP_END do { } while (0)
#define ENT_STAMP(i) do { } while (0)
#define ENT_STAMP_INIT do { } while (0)
#endif
#define ENT_LEN_PUT(rank, val)                                   \
    do {                                                         \
        if ((rank) < ENT_LEN_LDS) s_len[rank] = (uint16_t)(val); \
        else len_mem[rank] = (uint16_t)(val);                    \
    } while (0)
__global__ __launch_bounds__(ENT_THREADS) void k_entropy_p(venc_geom g, venc_bufs B, venc_step S) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint16_t s_tab[VENC_TAB_SIZE];
    __shared__ uint32_t s_wa[ENT_THREADS / 64], s_wr[ENT_THREADS / 64], s_wtot[2][ENT_THREADS / 64];
    __shared__ unsigned long long s_mask[ENT_SEG_MBS / 64];
    __shared__ uint32_t s_prev_run;
    __shared__ uint8_t s_cbp[48]; /* coded_block_pattern -> codeNum (table 9-4, Inter column): from memory it was a dependent round trip in the middle of phase A */
    const int v = blockIdx.z, s = S.slice0 + blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int srow = B.slice_first_row[s];
    const int first = srow * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    if (j * ENT_SEG_MBS >= last - first) return;
    ENT_STAMP_INIT;
    lds_u32 *strip = (lds_u32 *)(s_dyn + tid * 9); /* (written as dwords, read by the coder as int16: may_alias, see venc_kernels.hip) */
    uint16_t *s_len = (uint16_t *)(s_dyn + ENT_THREADS * 9);
    uint32_t *s_cls = s_dyn + ENT_THREADS * 9 + ENT_LEN_LDS / 2, *s_cls2 = s_cls + ENT_CLS_LDS; /* (s_cls2: the empty blocks behind a listed block) */
    const int mb_seg = first + j * ENT_SEG_MBS;
    const int mb = mb_seg + tid;
    const bool in = mb < last;
    const int mbc = in ? mb : first;
    int mx, my;
    mb_xy(g, mbc, mx, my);
    const bool aL = mx > 0, aT = my > srow;
    const venc_mbinfo *infos = B.mbinfo + (size_t)v * g.nmb;
    const uint32_t *mvs = B.mvsad + (size_t)v * g.nmb;
    /* ---------------------------------------------------------------- phase A */
    const uint4 *ip = (const uint4 *)(infos + mbc), *lp = (const uint4 *)(infos + (aL ? mbc - 1 : mbc)), *tp = (const uint4 *)(infos + (aT ? mbc - g.mbw : mbc));
    const uint4 o0 = ip[0], o1 = ip[1], l0 = lp[0], l1 = lp[1], t0 = tp[0], t1 = tp[1];
    const uint32_t me = mvs[mbc];
    uint32_t lb_rec = 0, lb_me = 0, lb_a = 0, lb_b = 0, lb_c = 0; /* the look-back's first step (first wave; see below) */
    if (tid < 64 && j > 0) {
        const int idx = mb_seg - 1 - lane, ic = idx >= first ? idx : first;
        int ix, iy;
        mb_xy(g, ic, ix, iy);
        const bool lL = ix > 0, lT = iy > srow;
        lb_rec = ((const uint32_t *)(infos + ic))[6];
        lb_me = mvs[ic];
        lb_a = mvs[lL ? ic - 1 : ic];
        lb_b = mvs[lT ? ic - g.mbw : ic];
        lb_c = mvs[lT && ix + 1 < g.mbw ? ic - g.mbw + 1 : lT && lL ? ic - g.mbw - 1 : ic];
    }
    int pxv, pyv;
    bool zero_inferred;
    mv_predict(g, mvs, mbc, mx, aL, aT, pxv, pyv, zero_inferred);
    for (int i = tid; i < VENC_TAB_SIZE; i += ENT_THREADS) { /* (the VLC tables of phase B: their loads fly with the records') */
        s_tab[i] = i < VENC_TAB_TOKEN_CDC  ? venc_vlc_coeff_token[i]
                 : i < VENC_TAB_TZ         ? venc_vlc_coeff_token_cdc[i - VENC_TAB_TOKEN_CDC]
                 : i < VENC_TAB_TZ_CDC     ? venc_vlc_total_zeros[i - VENC_TAB_TZ]
                 : i < VENC_TAB_RUN        ? venc_vlc_total_zeros_cdc[i - VENC_TAB_TZ_CDC]
                                           : venc_vlc_run_before[i - VENC_TAB_RUN];
    }
    if (tid >= 64 && tid < 64 + 48) s_cbp[tid - 64] = venc_cbp_inter_code[tid - 64];
    const int dx = (int)(int8_t)(me & 255u), dy = (int)(int8_t)((me >> 8) & 255u);
    const int cbpl = (int)(o1.z & 15u), cbpc = (int)((o1.z >> 16) & 3u), cbp = cbpl | (cbpc << 4);
    const bool skip = in && cbp == 0 && (zero_inferred ? (dx == 0 && dy == 0) : (dx == pxv && dy == pyv));
    const unsigned long long coded = __ballot(in && !skip);
    if (lane == 0) s_mask[tid >> 6] = coded;
    uint32_t em = 0, nm = 0, sm = 0; /* bit = slot: entries; those that are blocks with a level (the chroma DC blocks count as such: their record has no
                                      * count); those of them that hold exactly ONE level (three of four on ordinary content: phase B codes them apart) */
    if (in) {
        if (skip)
            em = (mb == last - 1 || mb == first) ? 1u : 0u; /* the slice's last mb_skip_run / its header */
        else {
            em = 3u;
#pragma unroll
            for (int q = 0; q < 4; q++) em |= ((cbpl >> q) & 1) ? 0xfu << (2 + 4 * q) : 0u;
            em |= cbpc > 0 ? 3u << 18 : 0u;
            em |= cbpc == 2 ? 0xffu << 20 : 0u;
            const uint32_t nzr = nonzero_bytes4(o0.x) | (nonzero_bytes4(o0.y) << 4) | (nonzero_bytes4(o0.z) << 8) | (nonzero_bytes4(o0.w) << 12); /* luma, raster order */
#pragma unroll
            for (int bk = 0; bk < 16; bk++) nm |= ((nzr >> (blk_y(bk) * 4 + blk_x(bk))) & 1u) << (2 + bk);
            nm |= 3u << 18;
            nm |= (nonzero_bytes4(o1.x) | (nonzero_bytes4(o1.y) << 4)) << 20;
            nm &= em;
            const uint32_t k1 = 0x01010101u;
            const uint32_t oner = (nonzero_bytes4(o0.x ^ k1) | (nonzero_bytes4(o0.y ^ k1) << 4) | (nonzero_bytes4(o0.z ^ k1) << 8) | (nonzero_bytes4(o0.w ^ k1) << 12)) ^ 0xffffu; /* TotalCoeff == 1, raster order */
#pragma unroll
            for (int bk = 0; bk < 16; bk++) sm |= ((oner >> (blk_y(bk) * 4 + blk_x(bk))) & 1u) << (2 + bk);
            sm |= ((nonzero_bytes4(o1.x ^ k1) | (nonzero_bytes4(o1.y ^ k1) << 4)) ^ 0xffu) << 20;
            sm &= nm & ~(3u << 18);
        }
    }
    /* ranks (bitstream order) and the places in phase B's list: one block scan of three counts */
    /* entries: R, H and the blocks with levels; the coeff_tokens of the blocks WITHOUT a level ride at the end of the entry before
     * them (at most three behind an H, seven behind a block: a coded 8x8 block or a coded chroma AC plane holds a block with levels) */
    const uint32_t n_all = (uint32_t)__popc(em & (3u | nm)), n_n = (uint32_t)__popc(nm), n_r = (em & 1u) | ((uint32_t)__popc(sm) << 16); /* (n_r: R | blocks with one level << 16) */
    const uint32_t ia = wave_scan_incl(n_all | (n_n << 16)), ir = wave_scan_incl(n_r);
    if (lane == 63) s_wa[tid >> 6] = ia, s_wr[tid >> 6] = ir;
    /* the first wave: P_Skip macroblocks directly before the segment (the run a mb_skip_run of this segment may continue).  The loads of its
     * first step -- nearly always the only one -- were issued with the records' above (lb_*): one memory round trip, where the
     * record -> coded_block_pattern -> neighbours' vectors chain behind the records' made the other three waves wait at the barrier
     * for a fifth of their lifetime (tools/ent_stamps.py) */
    if (tid < 64) {
        uint32_t prev = 0;
        if (j > 0) {
            int base = mb_seg;
            bool preloaded = true;
            for (;;) {
                const int idx = base - 1 - lane;
                const bool valid = idx >= first;
                const int ic = valid ? idx : first;
                int ix, iy;
                mb_xy(g, ic, ix, iy);
                bool sk;
                if (preloaded)
                    sk = p_skip_vals(g, ix, ix > 0, iy > srow, (int)(lb_rec & 15u) | (int)(((lb_rec >> 16) & 3u) << 4), lb_me, lb_a, lb_b, lb_c);
                else {
                    const uint32_t rec = ((const uint32_t *)(infos + ic))[6];
                    sk = mb_is_p_skip(g, mvs, ic, ix, ix > 0, iy > srow, (int)(rec & 15u) | (int)(((rec >> 16) & 3u) << 4));
                }
                preloaded = false;
                const unsigned long long cb = __ballot(valid && !sk), vb = __ballot(valid);
                if (cb) {
                    prev += (uint32_t)__builtin_ctzll(cb);
                    break;
                }
                prev += (uint32_t)__popcll(vb);
                if (vb != ~0ull) break; /* the slice's start */
                base -= 64;
            }
        }
        if (lane == 0) s_prev_run = prev;
    }
    ENT_STAMP(0);
    __syncthreads();
    ENT_STAMP(1);
    uint32_t xa = ia - (n_all | (n_n << 16)), xr = ir - n_r, ta = 0, tr = 0;
#pragma unroll
    for (int k = 0; k < ENT_THREADS / 64; k++) {
        const uint32_t a = s_wa[k], r = s_wr[k];
        xa += k < (tid >> 6) ? a : 0u;
        xr += k < (tid >> 6) ? r : 0u;
        ta += a;
        tr += r;
    }
    const size_t seg = ent_seg(g, B, v, s, j);
    /* phase B's list: the R entries, the blocks with ONE level, the other blocks with levels (a wave lasts as long as its longest block: apart,
     * the three blocks of four that hold one level -- noise just above the dead zone -- no longer wait for the fourth) */
    const uint32_t count = ta & 0xffffu, n_r_all = tr & 0xffffu, n_one = n_r_all + (tr >> 16), n_rn = n_r_all + (ta >> 16);
    uint4 *head = B.ent_head + seg * B.ent_cap;
    uint32_t *cls_mem = B.ent_cls + seg * B.ent_cap, *cls2_mem = B.ent_cls2 + seg * B.ent_cap, *list = B.ent_list + seg * B.ent_cap;
    uint16_t *len_mem = (uint16_t *)(B.ent_off + seg * (B.ent_cap + 4)) + 2 * (B.ent_cap + 4) - B.ent_cap - 8; /* (lengths beyond the LDS part park in the upper half of the segment's
                                                                                                                  offsets, which phase C fills from below: rank k's offset is written after every length <= k was read) */
    if (in) B.ent_mb0[(size_t)v * g.nmb + mb] = (xa & 0xffffu) | (n_all << 16); /* (debug / tests: the macroblock's entries are ranks [lo, lo + hi)) */
    if (em) {
        uint32_t rank = xa & 0xffffu, pos_r = xr & 0xffffu, pos_s = n_r_all + (xr >> 16), pos_m = n_one + (xa >> 16) - (xr >> 16);
        /* nC of the macroblock's 24 blocks: luma rows ncl[y] (byte = x), chroma planes ncc[plane] (byte = 2 y + x) */
        uint32_t ncl[4], ncc[2];
        {
            const uint32_t mLa = aL ? 0xffffffffu : 0xffffff00u, mTa = aT ? 0xffffffffu : 0u;
            const uint32_t own[4] = {o0.x, o0.y, o0.z, o0.w}, lft[4] = {l0.x, l0.y, l0.z, l0.w};
#pragma unroll
            for (int y = 0; y < 4; y++)
                ncl[y] = nc_bytes4(__builtin_amdgcn_alignbyte(own[y], lft[y], 3u), y ? own[y - 1] : t0.w, mLa, y ? 0xffffffffu : mTa);
            const uint32_t mLc = aL ? 0xffffffffu : 0xff00ff00u, mTc = aT ? 0xffffffffu : 0xffff0000u;
            const uint32_t oc[2] = {o1.x, o1.y}, lc[2] = {l1.x, l1.y}, tc[2] = {t1.x, t1.y};
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                ncc[pl] = nc_bytes4(__builtin_amdgcn_perm(oc[pl], lc[pl], 0x06030401u), __builtin_amdgcn_alignbyte(oc[pl], tc[pl], 2u), mLc, mTc);
        }
        /* The nC CLASS of every block (what selects its coeff_token table: nC < 2, < 4, < 8, >= 8), two bits per block in slot order
         * (block b = slot 2 + b at bits 2 b): classified four bytes at a time (n >= k  <=>  bit 7 of n + 128 - k; n <= 16), then put
         * into luma4x4BlkIdx order.  A block with levels takes its table from it (phase B needs no count), an empty block its token. */
        unsigned long long cw;
        {
            uint32_t c4[6];
            const uint32_t ncw[6] = {ncl[0], ncl[1], ncl[2], ncl[3], ncc[0], ncc[1]};
#pragma unroll
            for (int i = 0; i < 6; i++)
                c4[i] = (((ncw[i] + 0x7e7e7e7eu) >> 7) & 0x01010101u) + (((ncw[i] + 0x7c7c7c7cu) >> 7) & 0x01010101u) + (((ncw[i] + 0x78787878u) >> 7) & 0x01010101u);
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int bk = 0; bk < 16; bk++) lo |= ((c4[blk_y(bk)] >> (8 * blk_x(bk))) & 3u) << (2 * bk);
#pragma unroll
            for (int k = 0; k < 8; k++) hi |= ((c4[4 + (k >> 2)] >> (8 * (k & 3))) & 3u) << (4 + 2 * k); /* (blocks 16, 17 = the chroma DC blocks: no class) */
            cw = ((unsigned long long)hi << 32) | lo;
        }
#define ENT_TOKEN0(sink, c_) venc_bs_put(sink, (c_) == 3u ? 3u : (1u << ((c_) == 0u ? 1 : (c_) == 1u ? 2 : 4)) - 1u, (c_) == 0u ? 1 : (c_) == 1u ? 2 : (c_) == 2u ? 4 : 6) /* coeff_token of TotalCoeff 0 (table 9-5): 1, 11, 1111, 0000 11 */
        if (em & 1u) { /* R: left to phase B */
            const uint32_t rec = (rank << 18) | ((uint32_t)tid << 5);
            if (pos_r < ENT_CLS_LDS) s_cls[pos_r] = rec; else cls_mem[pos_r] = rec;
            rank++;
        }
        uint32_t blocks = em >> 2; /* bit b = block b (slot 2 + b) is an entry or rides on one */
        const uint32_t lev = nm >> 2; /* ... and holds levels */
        if (em & 2u) {
            /* H: mb_type P_L0_16x16, mvd_l0 in quarter samples, coded_block_pattern, mb_qp_delta (at most 49 bits), then the tokens of the
             * empty blocks in front of the first block with levels (at most three: 18 bits) */
            venc_bitsink4 hbs;
            venc_bs4_init(&hbs, nullptr);
            venc_bs_ue(&hbs, 0u);
            venc_bs_se(&hbs, (dx - pxv) << (g.subpel ? 0 : 2));
            venc_bs_se(&hbs, (dy - pyv) << (g.subpel ? 0 : 2));
            venc_bs_ue(&hbs, s_cbp[cbp]);
            if (cbp) venc_bs_se(&hbs, 0);
            uint32_t lead = lev ? blocks & ((1u << __builtin_ctz(lev)) - 1u) : blocks;
            blocks &= ~lead;
            while (lead) {
                const int bq = __builtin_ctz(lead);
                lead &= lead - 1u;
                ENT_TOKEN0(&hbs, (uint32_t)(cw >> (2 * bq)) & 3u);
            }
            ENT_LEN_PUT(rank, venc_bs_flush(&hbs));
            head[rank] = make_uint4(hbs.h0, hbs.h1, hbs.h2, hbs.h3);
            rank++;
        }
        /* the blocks: one with levels is listed for phase B (its table class beside it); the empty ones behind it ride on it: `tail` = their
         * classes, two bits each from bit 0, the count in bits 16-18 (at most seven: a coded 8x8 block / chroma AC plane holds levels).
         * The loop visits the blocks WITH levels only (a walk over every present block cost ~70 instructions per block, twenty-four
         * times in most waves): a block's riders are the present blocks below the next block with levels -- the rest of its own group
         * of four and the head of that next block's group, i.e. at most two runs of consecutive bits (groups in between are absent as
         * a whole; the eight chroma AC blocks are all present or all absent) -- and their classes two shifts of the class word. */
        uint32_t L = lev & blocks;
        while (L) {
            const int bq = __builtin_ctz(L);
            L &= L - 1u;
            const uint32_t above = blocks & (0xfffffffeu << bq);
            const uint32_t rm = L ? above & ((1u << __builtin_ctz(L)) - 1u) : above;
            const int s1 = __builtin_ctz(rm | 0x80000000u); /* (no rider: 31, and everything below comes out empty) */
            const int l1 = __builtin_ctz(~(rm >> s1));
            const uint32_t rm2 = rm & ~(((1u << l1) - 1u) << s1);
            const int s2 = __builtin_ctz(rm2 | 0x80000000u), l2 = __popc(rm2);
            const uint32_t f1 = (uint32_t)(cw >> (2 * s1)) & ((1u << (2 * l1)) - 1u), f2 = (uint32_t)(cw >> (2 * s2)) & ((1u << (2 * l2)) - 1u);
            const uint32_t tail = f1 | (f2 << (2 * l1)) | ((uint32_t)(l1 + l2) << 16);
            const uint32_t c4 = 4u * ((uint32_t)(cw >> (2 * bq)) & 3u);
            const uint32_t ncf = (bq == 16 || bq == 17) ? 0u : (0x9531u >> c4) & 15u; /* nC + 1 of the class: 1, 3, 5, 9; the chroma DC blocks: 0 */
            const uint32_t rec = (rank << 18) | (ncf << 13) | ((uint32_t)tid << 5) | (uint32_t)(bq + 2);
            const bool one = (sm >> (bq + 2)) & 1u;
            const uint32_t pos_n = one ? pos_s : pos_m;
            pos_s += one ? 1u : 0u;
            pos_m += one ? 0u : 1u;
            if (pos_n < ENT_CLS_LDS) {
                s_cls[pos_n] = rec;
                s_cls2[pos_n] = tail;
            } else {
                cls_mem[pos_n] = rec;
                cls2_mem[pos_n] = tail;
            }
            rank++;
        }
    }
    ENT_STAMP(2);
    __syncthreads();
    ENT_STAMP(3);
    /* ---------------------------------------------------------------- phase B */
    /* R: slice header, mb_skip_run */
    for (uint32_t k = tid; k < n_r_all; k += ENT_THREADS) {
        const uint32_t e = k < ENT_CLS_LDS ? s_cls[k] : cls_mem[k], rank = e >> 18;
        const int r = (int)((e >> 5) & 255u), mbk = mb_seg + r;
        venc_bitsink4 bs;
        venc_bs4_init(&bs, B.mbbits + ((size_t)v * g.nmb + mbk) * VENC_MB_WORDS);
        const bool skipped = !((s_mask[r >> 6] >> (r & 63)) & 1ull);
        if (mbk == first) emit_slice_header(&bs, mbk, S, v, B);
        if (!skipped || mbk == last - 1) {
            int w = r >> 6;
            unsigned long long m = s_mask[w] & ((1ull << (r & 63)) - 1ull); /* coded macroblocks of the segment below r in its word */
            uint32_t run;
            for (;;) {
                if (m) {
                    run = (uint32_t)(r - 1 - ((w << 6) + 63 - __builtin_clzll(m)));
                    break;
                }
                if (w == 0) {
                    run = (uint32_t)r + s_prev_run;
                    break;
                }
                w--;
                m = s_mask[w];
            }
            venc_bs_ue(&bs, skipped ? run + 1u : run); /* (a slice that ends in skipped macroblocks carries their count as a last mb_skip_run) */
        }
        const int bits = venc_bs_flush(&bs);
        ENT_LEN_PUT(rank, bits);
        head[rank] = make_uint4(bs.h0, bs.h1, bs.h2, bs.h3);
        if (bits > 128) list[rank] = (uint32_t)mbk << 5;
    }
    ENT_STAMP(4);
    /* blocks that hold a level (and the chroma DC blocks) */
    /* ... exactly ONE level (luma / chroma AC): no level strip in LDS, no loop -- the level's scan position from ORs of the record's eight
     * dwords, coeff_token (TotalCoeff 1, TrailingOnes 0 / 1), sign or level_prefix / suffix at suffixLength 0, total_zeros = the
     * position; no run_before.  At most 53 bits + 42 of the empty blocks behind it: assembled in two 64-bit words */
    for (uint32_t k = n_r_all + tid; k < n_one; k += ENT_THREADS) {
        const uint32_t e = k < ENT_CLS_LDS ? s_cls[k] : cls_mem[k], rank = e >> 18;
        const int slot = (int)(e & 31u), mbk = mb_seg + (int)((e >> 5) & 255u), nC = (int)((e >> 13) & 31u) - 1;
        const bool ac = slot >= 20;
        const int16_t *lv = B.levels + ((size_t)v * g.nmb + mbk) * VENC_MB_COEFS + ent_level_slot(slot) * 16;
        const uint4 a = ((const uint4 *)lv)[0], bq = ((const uint4 *)lv)[1];
        uint32_t tl = k < ENT_CLS_LDS ? s_cls2[k] : cls2_mem[k];
        const uint32_t d0 = ac ? a.x & 0xffff0000u : a.x;
        const uint32_t o01 = d0 | a.y, o23 = a.z | a.w, o45 = bq.x | bq.y, o67 = bq.z | bq.w;
        const uint32_t x = o01 | o23 | o45 | o67;
        const uint32_t kd = (uint32_t)((a.y | a.w | bq.y | bq.w) != 0u) | ((uint32_t)((o23 | o67) != 0u) << 1) | ((uint32_t)((o45 | o67) != 0u) << 2);
        const bool hi = (x >> 16) != 0u;
        const int val = hi ? (int)x >> 16 : (int)(int16_t)(x & 0xffffu);
        const uint32_t ci = 2u * kd + (hi ? 1u : 0u) - (ac ? 1u : 0u);
        const uint32_t alv = (uint32_t)(val < 0 ? -val : val), neg = val < 0 ? 1u : 0u;
        const bool t1 = alv == 1u;
        const uint32_t tok = s_tab[VENC_TAB_TOKEN + (nC < 2 ? 0 : nC < 4 ? 68 : nC < 8 ? 136 : 204) + 4 + (t1 ? 1 : 0)];
        const uint32_t code = 2u * alv - 4u + neg; /* levelCode - 2: the first level behind fewer than three trailing ones */
        const uint32_t lcode = t1 ? neg : code < 14u ? 1u : code < 30u ? 16u | (code - 14u) : 4096u | (code - 30u);
        const uint32_t llen = t1 ? 1u : code < 14u ? code + 1u : code < 30u ? 19u : 28u;
        const uint32_t tz = s_tab[VENC_TAB_TZ + ci];
        unsigned long long acc = (((unsigned long long)(tok & 0xffu) << llen) | lcode) << (tz >> 8) | (tz & 0xffu);
        const uint32_t n1 = (tok >> 8) + llen + (tz >> 8);
        unsigned long long tacc = 0;
        uint32_t n2 = 0;
        for (uint32_t n = tl >> 16; n; n--, tl >>= 2) { /* the empty blocks behind it: coeff_token of TotalCoeff 0 (1, 11, 1111, 0000 11) */
            const uint32_t c4 = 4u * (tl & 3u), len = (0x6421u >> c4) & 15u;
            tacc = (tacc << len) | ((0x3f31u >> c4) & 15u);
            n2 += len;
        }
        const unsigned long long tleft = n2 ? tacc << (64u - n2) : 0ull;
        const unsigned long long top = (acc << (64u - n1)) | (tleft >> n1), bot = tleft << (64u - n1);
        ENT_LEN_PUT(rank, n1 + n2);
        head[rank] = make_uint4((uint32_t)(top >> 32), (uint32_t)top, (uint32_t)(bot >> 32), (uint32_t)bot);
    }
    /* ... more (and the chroma DC blocks): the general coder */
    for (uint32_t k = n_one + tid; k < n_rn; k += ENT_THREADS) {
        const uint32_t e = k < ENT_CLS_LDS ? s_cls[k] : cls_mem[k], rank = e >> 18;
        const int slot = (int)(e & 31u), mbk = mb_seg + (int)((e >> 5) & 255u), nC = (int)((e >> 13) & 31u) - 1;
        const int firstc = slot >= 20 ? 1 : 0, maxc = slot >= 20 ? 15 : slot >= 18 ? 4 : 16;
        const int16_t *lv = B.levels + ((size_t)v * g.nmb + mbk) * VENC_MB_COEFS + ent_level_slot(slot) * 16;
        const uint4 a = ((const uint4 *)lv)[0];
        strip[0] = a.x, strip[1] = a.y, strip[2] = a.z, strip[3] = a.w;
        uint32_t wds[8] = {a.x, a.y, a.z, a.w, 0u, 0u, 0u, 0u};
        if (maxc != 4) { /* chroma DC records hold 4 levels: one 16-byte load is enough */
            const uint4 bq = ((const uint4 *)lv)[1];
            strip[4] = bq.x, strip[5] = bq.y, strip[6] = bq.z, strip[7] = bq.w;
            wds[4] = bq.x, wds[5] = bq.y, wds[6] = bq.z, wds[7] = bq.w;
        }
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            nz |= (uint32_t)((wds[i] & 0xffffu) != 0u) << (2 * i);
            nz |= (uint32_t)((wds[i] >> 16) != 0u) << (2 * i + 1);
        }
        nz = (nz >> firstc) & ((1u << maxc) - 1u);
        venc_bitsink4 bs;
        venc_bs4_init(&bs, B.mbbits + ((size_t)v * g.nmb + mbk) * VENC_MB_WORDS + slot * VENC_LANE_WORDS);
        venc_cavlc_block_mem(&bs, (const int16_t *)strip, nz, firstc, maxc, nC, s_tab);
        for (uint32_t t = k < ENT_CLS_LDS ? s_cls2[k] : cls2_mem[k], n = t >> 16; n; n--, t >>= 2) ENT_TOKEN0(&bs, t & 3u); /* the empty blocks behind it */
        const int bits = venc_bs_flush(&bs);
        ENT_LEN_PUT(rank, bits);
        head[rank] = make_uint4(bs.h0, bs.h1, bs.h2, bs.h3);
        if (bits > 128) list[rank] = ((uint32_t)mbk << 5) | (uint32_t)slot;
    }
    ENT_STAMP(5);
    __syncthreads();
    ENT_STAMP(6);
    /* ---------------------------------------------------------------- phase C: bit offsets = exclusive scan of the lengths in rank order */
    uint32_t *off = B.ent_off + seg * (B.ent_cap + 4);
    uint32_t base_bits = 0;
    int phase = 0;
    for (uint32_t k0 = 0; k0 < count; k0 += ENT_THREADS, phase ^= 1) {
        const uint32_t k = k0 + (uint32_t)tid;
        const uint32_t bits = k >= count ? 0u : k < ENT_LEN_LDS ? (uint32_t)s_len[k] : (uint32_t)len_mem[k];
        const uint32_t incl = wave_scan_incl(bits);
        if (lane == 63) s_wtot[phase][tid >> 6] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < ENT_THREADS / 64; w++) {
            const uint32_t t = s_wtot[phase][w];
            woff += w < (tid >> 6) ? t : 0u;
            tot += t;
        }
        if (k < count) off[k] = base_bits + woff + incl - bits;
        base_bits += tot;
    }
    if (tid == 0) {
        off[count] = base_bits;
        B.seg_bits[seg] = base_bits;
        B.ent_count[seg] = count;
    }
    ENT_STAMP(7);
    ENT_STAMP_END;
}

/* ================================================================== k_pack_p
 * grid as k_skip2.  Output word w of a slice (stream bits [32 w, 32 w + 32)) belongs to the segment that holds bit 32 w.  A
 * block zeroes an LDS window of its words, every lane ORs its entries' strings into it at their bit offsets (MSB first; strings
 * are zero beyond their length), the first wave adds what the block's last word needs from the following segments' first entries,
 * and the window leaves as coalesced stores.  The slice's last segment appends rbsp_stop_one_bit. */
#define PACK_P_WORDS 1024
/* ORs bits [o, o + len) of a slice (an entry's string: words 0 .. 3 in h, the rest at spill[4 ..]) into words [w_lo, w_hi) held at win[w - w_lo] */
DEVFN void ent_or_into(uint32_t *win, uint32_t w_lo, uint32_t w_hi, uint32_t o, uint32_t len, const uint4 &h, const uint32_t *spill) {
    const uint32_t nw = (len + 31u) >> 5, sh = o & 31u;
    uint32_t t = o >> 5;
    for (uint32_t i = 0; i < nw; i++, t++) {
        if (t + 1u < w_lo) continue;
        if (t >= w_hi) break;
        const uint32_t src = i == 0 ? h.x : i == 1 ? h.y : i == 2 ? h.z : i == 3 ? h.w : spill[i];
        if (t >= w_lo && (src >> sh)) atomicOr(&win[t - w_lo], src >> sh);
        if (sh && t + 1u >= w_lo && t + 1u < w_hi && (src << (32u - sh))) atomicOr(&win[t + 1u - w_lo], src << (32u - sh));
    }
}

__global__ __launch_bounds__(ENT_THREADS) void k_pack_p(venc_geom g, venc_bufs B, venc_step S, venc_slice_out *slice_dev) {
    __shared__ uint32_t s_win[PACK_P_WORDS];
    __shared__ uint32_t s_sl_bits[VENC_MAX_SLICES]; /* bits of the band's slices */
    __shared__ uint32_t s_sl_off[VENC_MAX_SLICES];  /* byte offset of every slice of the picture */
    __shared__ uint32_t s_segoff;                   /* bits of this slice's segments before this one */
    const int v = blockIdx.z, sb = blockIdx.y, s = S.slice0 + sb, j = blockIdx.x, tid = threadIdx.x;
    const int first = B.slice_first_row[s] * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    const int nseg = (last - first + ENT_SEG_MBS - 1) / ENT_SEG_MBS;
    const bool table_block = j == 0 && sb == 0; /* this block also writes the picture's slice table */
    if (j >= nseg) return;
    /* ---- every block sums the segment totals of the band's slices itself: slice lengths, byte offsets, its own bit offset */
    const int band_slices = S.slice1 - S.slice0;
    for (int q = tid; q < band_slices; q += ENT_THREADS) {
        const int sq = S.slice0 + q;
        const int nq = ((B.slice_first_row[sq + 1] - B.slice_first_row[sq]) * g.mbw + ENT_SEG_MBS - 1) / ENT_SEG_MBS;
        const uint32_t *sbq = B.seg_bits + ent_seg(g, B, v, sq, 0);
        uint32_t acc = 0;
        for (int jj = 0; jj < nq; jj++) {
            if (q == sb && jj == j) s_segoff = acc;
            acc += sbq[jj];
        }
        s_sl_bits[q] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t byte_off = 0;
        for (int k = 0; k < g.n_slices; k++) { /* slices of other bands: empty entries */
            const bool mine = k >= S.slice0 && k < S.slice1;
            const uint32_t nbytes = mine ? (s_sl_bits[k - S.slice0] + 1 + 7) >> 3 : 0u;
            s_sl_off[k] = byte_off;
            byte_off += (nbytes + 3u) & ~3u;
        }
        if (table_block) {
            uint8_t *slot = B.out_hdr + (size_t)S.out_slot[v] * B.out_hdr_stride;
            venc_slice_out *host_tab = (venc_slice_out *)(slot + sizeof(venc_out_header));
            for (int k = 0; k < g.n_slices; k++) {
                const bool mine = k >= S.slice0 && k < S.slice1;
                const uint32_t nbits = mine ? s_sl_bits[k - S.slice0] : 0u;
                const venc_slice_out so = {s_sl_off[k], mine ? (nbits + 1 + 7) >> 3 : 0u, nbits, 0};
                slice_dev[(size_t)v * g.n_slices + k] = so;
                host_tab[k] = so;
            }
            const venc_out_header h = {byte_off, (uint32_t)g.n_slices, byte_off > B.out_cap ? 1u : 0u, 0};
            *(venc_out_header *)slot = h;
        }
    }
    __syncthreads();
    const uint32_t T = s_sl_bits[sb], nbytes = (T + 1 + 7) >> 3, byte_off = s_sl_off[s];
    const size_t seg = ent_seg(g, B, v, s, j);
    const uint32_t count = min(B.ent_count[seg], B.ent_cap);
    const uint32_t *list = B.ent_list + seg * B.ent_cap;
    const uint4 *head = B.ent_head + seg * B.ent_cap;
    const uint32_t *off = B.ent_off + seg * (B.ent_cap + 4);
    const bool last_seg = j == nseg - 1;
    const uint32_t begin = s_segoff, seg_end = begin + B.seg_bits[seg];
    const uint32_t end = last_seg ? nbytes * 8u : seg_end; /* the words this block owns start in [begin, end) */
    uint32_t *out = (uint32_t *)(B.out_data + (size_t)S.out_slot[v] * B.out_data_stride + byte_off);
    const uint32_t *mbb = B.mbbits + (size_t)v * g.nmb * VENC_MB_WORDS;
    for (uint32_t w0 = (begin + 31u) >> 5; 32u * w0 < end; w0 += PACK_P_WORDS) {
        const uint32_t w1 = min(w0 + (uint32_t)PACK_P_WORDS, (end + 31u) >> 5); /* owned words of this window: [w0, w1) */
        __syncthreads();
        for (uint32_t q = tid; q < w1 - w0; q += ENT_THREADS) s_win[q] = 0;
        __syncthreads();
        for (uint32_t k = tid; k < count; k += ENT_THREADS) {
            const uint32_t o = begin + off[k], len = off[k + 1] - off[k];
            if (len == 0 || o + len <= 32u * w0 || o >= 32u * w1) continue;
            const uint32_t e = len > 128u ? list[k] : 0u; /* (only a string beyond its four head words says where the rest lies) */
            ent_or_into(s_win, w0, w1, o, len, head[k], mbb + (size_t)(e >> 5) * VENC_MB_WORDS + (e & 31u) * VENC_LANE_WORDS);
        }
        /* the window's last word may run on into the following segments (never past the slice: the last segment ends it) */
        if (!last_seg && 32u * w1 > seg_end && tid < 64) {
            uint32_t obase = seg_end;
            for (int jj = j + 1; jj < nseg && obase < 32u * w1; jj++) {
                const size_t sg = ent_seg(g, B, v, s, jj);
                const uint32_t cn = min(B.ent_count[sg], B.ent_cap);
                const uint32_t *of = B.ent_off + sg * (B.ent_cap + 4);
                for (uint32_t k = tid; k < cn; k += 64) { /* (an entry is at least one bit: the first 32 cover the word) */
                    const uint32_t o = obase + of[k], len = of[k + 1] - of[k];
                    if (__ballot(o < 32u * w1) == 0ull) break;
                    if (len == 0 || o >= 32u * w1) continue;
                    const uint32_t e = len > 128u ? B.ent_list[sg * B.ent_cap + k] : 0u;
                    ent_or_into(s_win, w0, w1, o, len, B.ent_head[sg * B.ent_cap + k], mbb + (size_t)(e >> 5) * VENC_MB_WORDS + (e & 31u) * VENC_LANE_WORDS);
                }
                obase += B.seg_bits[sg];
            }
        }
        /* rbsp_stop_one_bit: by whichever block owns the word that holds bit T (the last segment's when the slice ends on a word
         * boundary; an earlier one's when the segments behind it are short: a slice that ends in skipped macroblocks) */
        if (tid == 0 && (T >> 5) >= w0 && (T >> 5) < w1) atomicOr(&s_win[(T >> 5) - w0], 0x80000000u >> (T & 31u));
        __syncthreads();
        for (uint32_t q = tid; q < w1 - w0; q += ENT_THREADS) {
            if (byte_off + 4u * (w0 + q + 1u) > B.out_cap) break;
            out[w0 + q] = __builtin_bswap32(s_win[q]);
        }
    }
}

/* debug / tests (venc_debug_read selector 3): bits per macroblock of a P picture = the span of its entries' offsets */
__global__ __launch_bounds__(ENT_THREADS) void k_entry_mbbits(venc_geom g, venc_bufs B, venc_step S, int v) {
    const int s = S.slice0 + blockIdx.y, j = blockIdx.x;
    const int first = B.slice_first_row[s] * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    const int mb = first + j * ENT_SEG_MBS + (int)threadIdx.x;
    if (mb >= last) return;
    const uint32_t *off = B.ent_off + ent_seg(g, B, v, s, j) * (B.ent_cap + 4);
    const uint32_t m0 = B.ent_mb0[(size_t)v * g.nmb + mb], r0 = m0 & 0xffffu, n = m0 >> 16;
    B.mbnbits[(size_t)v * g.nmb + mb] = off[r0 + n] - off[r0];
}
