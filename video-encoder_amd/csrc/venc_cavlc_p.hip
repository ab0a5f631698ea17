/*
 * venc_cavlc_p.hip -- entropy coding of P pictures, ONE LANE PER SYNTAX ENTRY.
 *
 * An ENTRY is the smallest piece of a P macroblock's syntax that can be coded on its own, because every CAVLC context (nC)
 * comes from the TotalCoeff values the transform kernels already stored, not from coded bits:
 *   slot 0        the macroblock header: slice header (first macroblock of a slice), mb_skip_run, mb_type, mvd_l0,
 *                 coded_block_pattern, mb_qp_delta -- or, for a skipped macroblock that ends its slice, the last mb_skip_run
 *   slot 1 .. 16  luma 4x4 block luma4x4BlkIdx = slot - 1 (present iff its 8x8 block's coded_block_pattern bit is set)
 *   slot 17, 18   Cb, Cr DC                 slot 19 .. 26  Cb, Cr AC blocks
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
 * Round 4 (this file):
 *   k_skip2     one lane per macroblock, blocks of 256 macroblocks OF ONE SLICE ("segments"): P_Skip decision, the mask of
 *               coded macroblocks (slice-relative words), and the segment's entries in bitstream order -- a block scan, no atomic
 *   k_cavlc_p2  one lane per entry of a segment: the entry's bit string leaves as ONE 16-byte store (venc_bitsink4: four words
 *               in registers; the rare entry beyond 128 bits spills into the macroblock's region of mbbits), its bit offset
 *               inside the segment comes from a block scan of the lengths, the segment's total goes to seg_bits
 *   k_pack_p    one block per segment: every block sums the picture's segment totals itself (a few dozen numbers: no scan kernel),
 *               ORs its entries into an LDS window at their bit offsets and streams the window out; the block of a picture's
 *               first segment also writes the slice table
 * A lane's work is one header or ONE 4x4 block (most of them a coeff_token and a level or two), the gather reads 20 bytes per
 * entry from dense arrays, and k_scan has left the P chain.  I pictures (26 dense blocks in every macroblock) stay on k_cavlc /
 * k_scan / k_pack (venc_kernels.hip).
 */

#define ENT_SEG_MBS 256   /* macroblocks of one slice per segment */
#define ENT_PER_MB 27     /* entries a macroblock can have */
#define ENT_THREADS 256

/* segment index of (pair v, slice s, segment j of the slice) and the geometry that goes with it */
DEVFN size_t ent_seg(const venc_geom &g, const venc_bufs &B, int v, int s, int j) { return ((size_t)v * g.n_slices + s) * B.ent_spp + j; }

/* ================================================================== k_skip2
 * grid = (segments per slice, slices of the band, (step, item) pairs).  P_Skip (7.3.4): no residual and the vector a decoder
 * infers (mb_is_p_skip).  The coded-macroblock mask is a wave ballot per 64 macroblocks OF THE SLICE; k_cavlc_p2 derives
 * mb_skip_run from it by looking backwards for the previous coded macroblock -- a count-leading-zeros per word, and whole rows of
 * skipped macroblocks are a handful of words.
 * The segment's entries are listed twice: in bitstream order (ent_list[rank] = mb << 5 | slot: what rank means downstream) and
 * SORTED BY KIND for the coder (ent_cls: headers | blocks with a level | blocks without; element = rank << 13 | macroblock of the
 * segment << 5 | slot), so that a wave of k_cavlc_p2 holds one kind of work.  Unsorted (first form of this round) a wave walked the
 * header path AND the longest block of its 64 entries for what was mostly one-bit tokens of empty blocks: 14.0 M vector
 * instructions per batch of four 1080p steps, half a motion search. */
DEVFN uint32_t nonzero_bytes4(uint32_t w) { /* bit k = byte k of w is not zero */
    return (uint32_t)((w & 0xffu) != 0u) | ((uint32_t)((w & 0xff00u) != 0u) << 1) | ((uint32_t)((w & 0xff0000u) != 0u) << 2) | ((uint32_t)((w >> 24) != 0u) << 3);
}
__global__ __launch_bounds__(ENT_SEG_MBS) void k_skip2(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_wa[ENT_SEG_MBS / 64], s_wh[ENT_SEG_MBS / 64];
    const int v = blockIdx.z, s = S.slice0 + blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    const int srow = B.slice_first_row[s];
    const int first = srow * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    if (j * ENT_SEG_MBS >= last - first) return;
    const int rel = j * ENT_SEG_MBS + tid, mb = first + rel;
    const bool in = mb < last;
    const int mbc = in ? mb : first;
    const int my = mbc / g.mbw, mx = mbc - my * g.mbw;
    const uint4 *ip = (const uint4 *)(B.mbinfo + (size_t)v * g.nmb + mbc); /* the whole record: nnz[24], cbp, flags, cbpc */
    const uint4 o0 = ip[0], o1 = ip[1];
    const int cbpl = (int)(o1.z & 15u), cbpc = (int)((o1.z >> 16) & 3u);
    const bool skip = in && mb_is_p_skip(g, B.mvsad + (size_t)v * g.nmb, mbc, mx, mx > 0, my > srow, cbpl | (cbpc << 4));
    const unsigned long long coded = __ballot(in && !skip);
    if ((tid & 63) == 0) B.codedmask[((size_t)v * g.n_slices + s) * B.ent_wps + (rel >> 6)] = coded;
    uint32_t em = 0, nm = 0; /* bit = slot: entries; those that are blocks with a level (the chroma DC blocks count as such: their record has no count) */
    if (in) {
        if (skip)
            em = (mb == last - 1 || rel == 0) ? 1u : 0u; /* the slice's last mb_skip_run / its header */
        else {
            em = 1u;
#pragma unroll
            for (int q = 0; q < 4; q++) em |= ((cbpl >> q) & 1) ? 0xfu << (1 + 4 * q) : 0u;
            em |= cbpc > 0 ? 3u << 17 : 0u;
            em |= cbpc == 2 ? 0xffu << 19 : 0u;
            const uint32_t nzr = nonzero_bytes4(o0.x) | (nonzero_bytes4(o0.y) << 4) | (nonzero_bytes4(o0.z) << 8) | (nonzero_bytes4(o0.w) << 12); /* luma, raster order */
#pragma unroll
            for (int bk = 0; bk < 16; bk++) nm |= ((nzr >> (blk_y(bk) * 4 + blk_x(bk))) & 1u) << (1 + bk);
            nm |= 3u << 17;
            nm |= (nonzero_bytes4(o1.x) | (nonzero_bytes4(o1.y) << 4)) << 19;
            nm &= em;
        }
    }
    const uint32_t n_all = (uint32_t)__popc(em), n_n = (uint32_t)__popc(nm), n_h = em & 1u;
    const uint32_t ia = wave_scan_incl(n_all | (n_n << 16)), ih = wave_scan_incl(n_h);
    if ((tid & 63) == 63) s_wa[tid >> 6] = ia, s_wh[tid >> 6] = ih;
    __syncthreads();
    uint32_t xa = ia - (n_all | (n_n << 16)), xh = ih - n_h, ta = 0, th = 0;
#pragma unroll
    for (int k = 0; k < ENT_SEG_MBS / 64; k++) {
        const uint32_t a = s_wa[k], h = s_wh[k];
        xa += k < (tid >> 6) ? a : 0u;
        xh += k < (tid >> 6) ? h : 0u;
        ta += a;
        th += h;
    }
    const uint32_t total = ta & 0xffffu, tot_n = ta >> 16;
    uint32_t rank = xa & 0xffffu, pos_n = th + (xa >> 16), pos_h = xh, pos_e = th + tot_n + (rank - xh - (xa >> 16));
    const size_t seg = ent_seg(g, B, v, s, j);
    uint32_t *list = B.ent_list + seg * B.ent_cap, *cls = B.ent_cls + seg * B.ent_cap;
    while (em) {
        const int slot = __builtin_ctz(em);
        em &= em - 1u;
        const uint32_t rec = (rank << 13) | ((uint32_t)tid << 5) | (uint32_t)slot;
        list[rank++] = ((uint32_t)mb << 5) | (uint32_t)slot;
        if (slot == 0)
            cls[pos_h++] = rec;
        else if ((nm >> slot) & 1u)
            cls[pos_n++] = rec;
        else
            cls[pos_e++] = rec;
    }
    if (tid == 0) *(uint4 *)(B.ent_count + 4 * seg) = make_uint4(total, th, tot_n, 0u);
}

/* P_Skip macroblocks directly before the macroblock at slice-relative index rel, from the slice's coded-macroblock mask */
DEVFN uint32_t skip_run_before(const unsigned long long *mask, int rel) {
    int w = rel >> 6;
    unsigned long long m = mask[w] & ((1ull << (rel & 63)) - 1ull); /* coded macroblocks below rel in its word */
    for (;;) {
        if (m) return (uint32_t)(rel - 1 - ((w << 6) + 63 - __builtin_clzll(m)));
        if (w == 0) return (uint32_t)rel;
        w--;
        m = mask[w];
    }
}

/* a residual block's place in the records: level slot, scan range, the counts of its left / upper neighbours -> nC (9.2.1) */
DEVFN int ent_block_nc(const venc_geom &g, const venc_mbinfo *infos, int mb, int lslot, bool aL, bool aT, int &firstc, int &maxc) {
    firstc = 0, maxc = 16;
    if (lslot == 17 || lslot == 18) {
        maxc = 4;
        return -1;
    }
    int ia, ib;
    bool inA, inB;
    if (lslot < 16) {
        const int bx = blk_x(lslot), by = blk_y(lslot);
        inA = bx > 0, inB = by > 0;
        ia = inA ? by * 4 + bx - 1 : by * 4 + 3;
        ib = inB ? (by - 1) * 4 + bx : 12 + bx;
    } else {
        firstc = 1, maxc = 15;
        const int kk = (lslot - 19) & 3, nb = 16 + 4 * ((lslot - 19) >> 2);
        const int bx = kk & 1, by = kk >> 1;
        inA = bx > 0, inB = by > 0;
        ia = inA ? nb + 2 * by : nb + 2 * by + 1;
        ib = inB ? nb + bx : nb + 2 + bx;
    }
    const bool haveA = inA || aL, haveB = inB || aT;
    const int nA = haveA ? (int)infos[inA ? mb : mb - 1].nnz[ia] : 0;
    const int nB = haveB ? (int)infos[inB ? mb : mb - g.mbw].nnz[ib] : 0;
    return venc_nc(haveA, nA, haveB, nB);
}

/* ================================================================== k_cavlc_p2
 * grid as k_skip2; the waves of a block walk the segment's headers, then its blocks with a level, then its empty blocks (no
 * barrier between them), leave every entry's string at head[rank] and its length in LDS, and the block ends with an exclusive scan
 * of the lengths in rank order.  Dynamic LDS: 36 bytes per lane (the 16 levels of the block being coded + a dword against bank
 * conflicts), then 2 bytes per entry the segment can hold. */
__global__ __launch_bounds__(ENT_THREADS) void k_cavlc_p2(venc_geom g, venc_bufs B, venc_step S) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint16_t s_tab[VENC_TAB_SIZE];
    __shared__ uint32_t s_wtot[2][ENT_THREADS / 64];
    const int v = blockIdx.z, s = S.slice0 + blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    const int srow = B.slice_first_row[s];
    const int first = srow * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    if (j * ENT_SEG_MBS >= last - first) return;
    const size_t seg = ent_seg(g, B, v, s, j);
    const uint4 cc = *(const uint4 *)(B.ent_count + 4 * seg);
    const uint32_t count = min(cc.x, B.ent_cap), n_h = min(cc.y, count), n_hn = min(cc.y + cc.z, count);
    const uint32_t *cls = B.ent_cls + seg * B.ent_cap;
    uint4 *head = B.ent_head + seg * B.ent_cap;
    uint32_t *off = B.ent_off + seg * (B.ent_cap + 4);
    lds_u32 *strip = (lds_u32 *)(s_dyn + tid * 9); /* (written as dwords, read by the coder as int16: may_alias, see venc_kernels.hip) */
    uint16_t *s_len = (uint16_t *)(s_dyn + ENT_THREADS * 9);
    for (int i = tid; i < VENC_TAB_SIZE; i += ENT_THREADS) {
        s_tab[i] = i < VENC_TAB_TOKEN_CDC  ? venc_vlc_coeff_token[i]
                 : i < VENC_TAB_TZ         ? venc_vlc_coeff_token_cdc[i - VENC_TAB_TOKEN_CDC]
                 : i < VENC_TAB_TZ_CDC     ? venc_vlc_total_zeros[i - VENC_TAB_TZ]
                 : i < VENC_TAB_RUN        ? venc_vlc_total_zeros_cdc[i - VENC_TAB_TZ_CDC]
                                           : venc_vlc_run_before[i - VENC_TAB_RUN];
    }
    __syncthreads();
    const venc_mbinfo *infos = B.mbinfo + (size_t)v * g.nmb;
    const unsigned long long *mask = B.codedmask + ((size_t)v * g.n_slices + s) * B.ent_wps;
    const int mb_seg = first + j * ENT_SEG_MBS;
    /* ---- headers */
    for (uint32_t k = tid; k < n_h; k += ENT_THREADS) {
        const uint32_t e = cls[k], rank = e >> 13;
        const int mb = mb_seg + (int)((e >> 5) & 255u);
        const int my = mb / g.mbw, mx = mb - my * g.mbw;
        venc_bitsink4 bs;
        venc_bs4_init(&bs, B.mbbits + ((size_t)v * g.nmb + mb) * VENC_MB_WORDS);
        const uint32_t rec = ((const uint32_t *)(infos + mb))[6];
        const int cbp = (int)(rec & 15u) | (int)(((rec >> 16) & 3u) << 4);
        const int rel = mb - first;
        const bool skipped = !((mask[rel >> 6] >> (rel & 63)) & 1ull);
        if (rel == 0) emit_slice_header(&bs, mb, S, v, B);
        if (!skipped || mb == last - 1) {
            const uint32_t run = skip_run_before(mask, rel);
            if (skipped)
                venc_bs_ue(&bs, run + 1u); /* a slice that ends in skipped macroblocks carries their count as a last mb_skip_run */
            else
                emit_p_mb_header(&bs, g, B.mvsad + (size_t)v * g.nmb, mb, mx, my, mx > 0, my > srow, cbp, run);
        }
        s_len[rank] = (uint16_t)venc_bs_flush(&bs);
        head[rank] = make_uint4(bs.h0, bs.h1, bs.h2, bs.h3);
    }
    /* ---- blocks that hold a level (and the chroma DC blocks) */
    for (uint32_t k = n_h + tid; k < n_hn; k += ENT_THREADS) {
        const uint32_t e = cls[k], rank = e >> 13;
        const int slot = (int)(e & 31u), mb = mb_seg + (int)((e >> 5) & 255u);
        const int my = mb / g.mbw, mx = mb - my * g.mbw;
        const int lslot = slot <= 16 ? slot - 1 : slot;
        int firstc, maxc;
        const int nC = ent_block_nc(g, infos, mb, lslot, mx > 0, my > srow, firstc, maxc);
        const int16_t *lv = B.levels + ((size_t)v * g.nmb + mb) * VENC_MB_COEFS + lslot * 16;
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
        venc_bs4_init(&bs, B.mbbits + ((size_t)v * g.nmb + mb) * VENC_MB_WORDS + slot * VENC_LANE_WORDS);
        venc_cavlc_block_mem(&bs, (const int16_t *)strip, nz, firstc, maxc, nC, s_tab);
        s_len[rank] = (uint16_t)venc_bs_flush(&bs);
        head[rank] = make_uint4(bs.h0, bs.h1, bs.h2, bs.h3);
    }
    /* ---- blocks without a level inside a coded 8x8 block / coded chroma AC: coeff_token of TotalCoeff 0 for their nC */
    for (uint32_t k = n_hn + tid; k < count; k += ENT_THREADS) {
        const uint32_t e = cls[k], rank = e >> 13;
        const int slot = (int)(e & 31u), mb = mb_seg + (int)((e >> 5) & 255u);
        const int my = mb / g.mbw, mx = mb - my * g.mbw;
        int firstc, maxc;
        const int nC = ent_block_nc(g, infos, mb, slot <= 16 ? slot - 1 : slot, mx > 0, my > srow, firstc, maxc);
        const uint32_t tok = s_tab[VENC_TAB_TOKEN + (nC < 2 ? 0 : nC < 4 ? 68 : nC < 8 ? 136 : 204)];
        const uint32_t len = tok >> 8;
        s_len[rank] = (uint16_t)len;
        head[rank] = make_uint4((tok & 0xffu) << (32u - len), 0u, 0u, 0u);
    }
    __syncthreads();
    /* ---- bit offsets inside the segment: exclusive scan of the lengths in rank order */
    uint32_t base_bits = 0;
    int phase = 0;
    for (uint32_t k0 = 0; k0 < count; k0 += ENT_THREADS, phase ^= 1) {
        const uint32_t k = k0 + (uint32_t)tid;
        const uint32_t bits = k < count ? (uint32_t)s_len[k] : 0u;
        const uint32_t incl = wave_scan_incl(bits);
        if ((tid & 63) == 63) s_wtot[phase][tid >> 6] = incl;
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
    }
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
    const uint32_t count = min(B.ent_count[4 * seg], B.ent_cap);
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
            const uint32_t e = list[k];
            ent_or_into(s_win, w0, w1, o, len, head[k], mbb + (size_t)(e >> 5) * VENC_MB_WORDS + (e & 31u) * VENC_LANE_WORDS);
        }
        /* the window's last word may run on into the following segments (never past the slice: the last segment ends it) */
        if (!last_seg && 32u * w1 > seg_end && tid < 64) {
            uint32_t obase = seg_end;
            for (int jj = j + 1; jj < nseg && obase < 32u * w1; jj++) {
                const size_t sg = ent_seg(g, B, v, s, jj);
                const uint32_t cn = min(B.ent_count[4 * sg], B.ent_cap);
                const uint32_t *of = B.ent_off + sg * (B.ent_cap + 4);
                for (uint32_t k = tid; k < cn; k += 64) { /* (an entry is at least one bit: the first 32 cover the word) */
                    const uint32_t o = obase + of[k], len = of[k + 1] - of[k];
                    if (__ballot(o < 32u * w1) == 0ull) break;
                    if (len == 0 || o >= 32u * w1) continue;
                    const uint32_t e = B.ent_list[sg * B.ent_cap + k];
                    ent_or_into(s_win, w0, w1, o, len, B.ent_head[sg * B.ent_cap + k], mbb + (size_t)(e >> 5) * VENC_MB_WORDS + (e & 31u) * VENC_LANE_WORDS);
                }
                obase += B.seg_bits[sg];
            }
        }
        if (last_seg && tid == 0 && (T >> 5) >= w0 && (T >> 5) < w1) atomicOr(&s_win[(T >> 5) - w0], 0x80000000u >> (T & 31u)); /* rbsp_stop_one_bit */
        __syncthreads();
        for (uint32_t q = tid; q < w1 - w0; q += ENT_THREADS) {
            if (byte_off + 4u * (w0 + q + 1u) > B.out_cap) break;
            out[w0 + q] = __builtin_bswap32(s_win[q]);
        }
    }
}

/* debug / tests (venc_debug_read selector 3): bits per macroblock of a P picture = the sum of its entries' lengths */
__global__ __launch_bounds__(ENT_THREADS) void k_entry_mbbits(venc_geom g, venc_bufs B, venc_step S, int v) {
    const int s = S.slice0 + blockIdx.y, j = blockIdx.x;
    const int first = B.slice_first_row[s] * g.mbw, last = B.slice_first_row[s + 1] * g.mbw;
    if (j * ENT_SEG_MBS >= last - first) return;
    const size_t seg = ent_seg(g, B, v, s, j);
    const uint32_t count = min(B.ent_count[4 * seg], B.ent_cap);
    const uint32_t *off = B.ent_off + seg * (B.ent_cap + 4);
    for (uint32_t k = threadIdx.x; k < count; k += ENT_THREADS) atomicAdd(&B.mbnbits[(size_t)v * g.nmb + (B.ent_list[seg * B.ent_cap + k] >> 5)], off[k + 1] - off[k]);
}
