/*
 * venc_cavlc_p.hip -- entropy coding of P pictures, ONE LANE PER SYNTAX UNIT of a macroblock
 * (SURVEY.md section 8a row a4.4).
 *
 * A macroblock of a P picture is cut into seven units that can be coded independently, because every CAVLC
 * context (nC) comes from the TotalCoeff values the transform kernels already stored, not from coded bits:
 *   unit 0  slice header (first macroblock of a slice), mb_skip_run .. mb_qp_delta, luma 8x8 block 0
 *   unit 1-3  luma 8x8 blocks 1-3 (four 4x4 blocks each)        unit 4  Cb and Cr DC
 *   unit 5  Cb AC (four blocks)                                 unit 6  Cr AC
 * Each unit appends to its own 64-word bit string in the macroblock's region of mbbits; k_scan turns the seven
 * lengths into offsets and k_pack gathers the strings.
 * The first version gave a whole macroblock to one lane: a wave then took as long as the heaviest of its 64
 * macroblocks -- up to 26 block codings, each behind a dependent load of its levels, ~100 us -- and the entropy
 * chain became the longer of the two chains of a step.  A unit is at most four blocks.
 * The second gave every macroblock eight lanes: 45 us, but seven of eight lanes had nothing to code (a P macroblock
 * of the 1080p clip averages 17-40 bits), and those 10 000 mostly idle waves held 80 % of the chip's wave slots
 * beside the motion search of the next step.  Now k_skip lists the units that exist (header units of coded
 * macroblocks, residual units with a coded_block_pattern bit) and k_cavlc_p gives one lane to each list entry.
 * I pictures (26 dense blocks in every macroblock) stay on k_cavlc (one lane per block, LDS concatenation).
 */


/* ================================================================== k_skip
 * P_Skip decisions of a P picture (7.3.4), one lane per macroblock: no residual and the vector a decoder infers
 * (mb_is_p_skip).  The result is a bit mask of CODED macroblocks, one 64-bit word per 64 macroblocks of the picture
 * (wave ballot); k_cavlc_p derives mb_skip_run from it by looking backwards for the previous coded macroblock of the
 * slice -- a count-leading-zeros per word, and whole rows of skipped macroblocks are a handful of words.
 * The same lane lists the macroblock's non-empty syntax units (mb << 3 | unit) for k_cavlc_p -- a wave scan and one
 * atomic per wave; the order of the list is irrelevant, every unit writes to its own place -- and zeroes the
 * macroblock's eight unit lengths, which the coded units then overwrite. */
__global__ __launch_bounds__(256) void k_skip(venc_geom g, venc_bufs B, venc_step S) {
    const int item = blockIdx.z, tid = threadIdx.x;
    const int base = (S.mb0 & ~63) + blockIdx.x * 256; /* words are aligned to multiples of 64 macroblocks of the picture */
    const int mb = base + tid;
    const bool in = mb >= S.mb0 && mb < S.mb1;
    const int mbc = in ? mb : S.mb0;
    const int mx = mbc % g.mbw, my = mbc / g.mbw;
    const int srow = slice_first_row_of(g, my);
    const venc_mbinfo *infos = B.mbinfo + (size_t)item * g.nmb;
    const uint32_t rec = ((const uint32_t *)(infos + mbc))[6]; /* bytes 24..27: cbp, flags, cbpc, pad */
    const int cbp = (int)(rec & 15u) | (int)(((rec >> 16) & 3u) << 4);
    const bool skip = in && mb_is_p_skip(g, B.mvsad + (size_t)item * g.nmb, mbc, mx, mx > 0, my > srow, cbp);
    const unsigned long long coded = __ballot(in && !skip);
    if ((tid & 63) == 0 && (mb >> 6) < (g.nmb + 63) / 64) B.codedmask[(size_t)item * ((g.nmb + 63) / 64) + (mb >> 6)] = coded;
    /* units: 0 header (+ luma 8x8 block 0), 1-3 luma 8x8 blocks, 4 chroma DC, 5/6 Cb/Cr AC; a slice that ends in
     * skipped macroblocks codes their count in unit 0 of its last macroblock */
    uint32_t um = 0;
    if (in) {
        const int cbpl = cbp & 15, cbpc = cbp >> 4;
        const int slice_last = B.slice_first_row[B.row_slice[my] + 1] * g.mbw - 1;
        /* unit 0 of a skipped macroblock still exists when it carries the slice header or the slice's last mb_skip_run */
        um = skip ? ((mb == slice_last || (mx == 0 && my == srow)) ? 1u : 0u) : 1u | (uint32_t)(cbpl & 14) | (cbpc > 0 ? 16u : 0u) | (cbpc == 2 ? 96u : 0u);
        *(uint4 *)(B.ubits + ((size_t)item * g.nmb + mb) * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
    /* one atomic per block (the items' counters sit 128 bytes apart: atomics on one cache line serialise at ~11 ns each,
     * which made this kernel 58 us at 2160p when every wave pulled and the ten counters shared a line) */
    __shared__ uint32_t s_wtot[4], s_base;
    const uint32_t n = (uint32_t)__popc(um), incl = wave_scan_incl(n);
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        s_base = total ? atomicAdd(&B.unit_count[item * VENC_COUNTER_STRIDE], total) : 0u;
    }
    __syncthreads();
    uint32_t at = s_base + incl - n;
    for (int k = 0; k < (tid >> 6); k++) at += s_wtot[k];
    uint32_t *list = B.unit_list + (size_t)item * g.nmb * 7;
    while (um) {
        const int u = __builtin_ctz(um);
        um &= um - 1u;
        list[at++] = ((uint32_t)mb << 3) | (uint32_t)u;
    }
}

/* P_Skip macroblocks directly before macroblock mb in its slice (first macroblock: `first`), from the coded-macroblock mask */
DEVFN uint32_t skip_run_before(const unsigned long long *mask, int mb, int first) {
    int w = mb >> 6;
    unsigned long long m = mask[w] & ((1ull << (mb & 63)) - 1ull); /* coded macroblocks below mb in its word */
    for (;;) {
        if ((w << 6) < first) m &= ~((1ull << (first & 63)) - 1ull); /* (only the slice's first word can start before it) */
        if (m) return (uint32_t)(mb - 1 - ((w << 6) + 63 - __builtin_clzll(m)));
        if ((w << 6) <= first) return (uint32_t)(mb - first);
        w--;
        m = mask[w];
    }
}

/* Blocks of 1024 lanes (pictures up to ~2.5K x 10 items; 512 beyond, where the listed units need the whole chip
 * anyway): the listed units of a picture then occupy a hundred CUs instead of landing, a few waves each, on all
 * 256 -- any resident block of this kernel (62 VGPRs, LDS) costs the motion search of the next step, which runs
 * beside it, its third block on that CU.  1080p: k_me_pool beside the entropy chain 87 -> 78 us with 1024 lanes per
 * block, 85 us with 512; 2160p: no difference for the search, but 1024-lane blocks wait for CUs with room for their
 * 38 KB (k_cavlc_p 107 -> 245 us beside the search).  The per-lane strips are dynamic LDS: 36 bytes per lane. */
#define CAVLC_P_MAX_THREADS 1024
#define CAVLC_P_BLOCK_LANES (40 * 1024) /* lanes per item: blocks loop over longer lists */
__global__ __launch_bounds__(CAVLC_P_MAX_THREADS) void k_cavlc_p(venc_geom g, venc_bufs B, venc_step S) {
    extern __shared__ uint32_t s_lv[]; /* per lane: the 16 levels of the block being coded (+1 dword against bank conflicts) */
    const uint32_t nthreads = blockDim.x;
    __shared__ uint16_t s_tab[VENC_TAB_SIZE];
    const int item = blockIdx.z, tid = threadIdx.x;
    /* the other set of unit counters: read last by the previous P batch's k_cavlc_p, filled next by the next one's k_skip
     * (which may cover more (step, item) pairs than this launch: all of them are cleared) */
    if (blockIdx.x == 0 && item == 0 && tid < B.vcap) B.unit_count_next[tid * VENC_COUNTER_STRIDE] = 0;
    const uint32_t count = B.unit_count[item * VENC_COUNTER_STRIDE];
    const uint32_t *list = B.unit_list + (size_t)item * g.nmb * 7;
    if (blockIdx.x * nthreads >= count) return; /* whole block: nothing listed for it */
    for (int i = tid; i < VENC_TAB_SIZE; i += (int)nthreads) {
        s_tab[i] = i < VENC_TAB_TOKEN_CDC  ? venc_vlc_coeff_token[i]
                 : i < VENC_TAB_TZ         ? venc_vlc_coeff_token_cdc[i - VENC_TAB_TOKEN_CDC]
                 : i < VENC_TAB_TZ_CDC     ? venc_vlc_total_zeros[i - VENC_TAB_TZ]
                 : i < VENC_TAB_RUN        ? venc_vlc_total_zeros_cdc[i - VENC_TAB_TZ_CDC]
                                           : venc_vlc_run_before[i - VENC_TAB_RUN];
    }
    __syncthreads();
  for (uint32_t base_e = blockIdx.x * nthreads; base_e < count; base_e += gridDim.x * nthreads) {
    const bool valid = base_e + tid < count;
    const uint32_t entry = list[valid ? base_e + tid : base_e];
    const int unit = (int)(entry & 7u);
    const int mb = (int)(entry >> 3);
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const int srow = slice_first_row_of(g, my);
    const bool aL = mx > 0, aT = my > srow;
    const venc_mbinfo *infos = B.mbinfo + (size_t)item * g.nmb;
    const uint4 *ip = (const uint4 *)(infos + mb);
    const uint4 *lp = (const uint4 *)(infos + (aL ? mb - 1 : mb));
    const uint4 *tp = (const uint4 *)(infos + (aT ? mb - g.mbw : mb));
    const uint4 o0 = ip[0], o1 = ip[1], l0 = lp[0], l1 = lp[1], t0 = tp[0], t1 = tp[1];
    if (!valid) continue;

    const int cbpl = (int)(o1.z & 15u), cbpc = (int)((o1.z >> 16) & 3u), cbp = cbpl | (cbpc << 4);
    uint32_t *strip = s_lv + tid * 9;
    const int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
    uint16_t *ubits = B.ubits + ((size_t)item * g.nmb + mb) * 8;
    const unsigned long long *mask = B.codedmask + (size_t)item * ((g.nmb + 63) / 64);
    const bool skipped = !((mask[mb >> 6] >> (mb & 63)) & 1ull); /* k_skip */
    venc_bitsink bs;
    venc_bs_init(&bs, B.mbbits + ((size_t)item * g.nmb + mb) * VENC_MB_WORDS + unit * VENC_UNIT_WORDS);

    /* blocks of this unit: slots first_slot .. first_slot + n_slots - 1 of the level record */
    int first_slot = 0, n_slots = 0;
    if (unit < 4) {
        first_slot = 4 * unit;
        n_slots = (!skipped && ((cbpl >> unit) & 1)) ? 4 : 0;
    } else if (unit == 4) {
        first_slot = 17;
        n_slots = (!skipped && cbpc > 0) ? 2 : 0;
    } else if (unit < 7) {
        first_slot = unit == 5 ? 19 : 23;
        n_slots = (!skipped && cbpc == 2) ? 4 : 0;
    }
    if (unit == 0) {
        if (mx == 0 && my == srow) emit_slice_header(&bs, mb, S, item, B);
        const int slice_last = B.slice_first_row[B.row_slice[my] + 1] * g.mbw - 1;
        if (!skipped || mb == slice_last) {
            const uint32_t run = skip_run_before(mask, mb, srow * g.mbw);
            if (skipped)
                venc_bs_ue(&bs, run + 1u); /* a slice that ends in skipped macroblocks carries their count as a last mb_skip_run */
            else
                emit_p_mb_header(&bs, g, B.mvsad + (size_t)item * g.nmb, mb, mx, my, aL, aT, cbp, run);
        }
    }
    for (int k = 0; k < n_slots; k++) {
        const int slot = first_slot + k;
        int first = 0, maxc = 16, nC = 0, own_idx = 0;
        bool has_nnz_entry = true;
        if (slot < 16) {
            own_idx = blk_y(slot) * 4 + blk_x(slot);
        } else if (slot < 19) {
            maxc = 4;
            nC = -1;
            has_nnz_entry = false;
        } else {
            first = 1, maxc = 15;
            own_idx = 16 + (slot - 19);
        }
        if (nC == 0) {
            int ia, ib;
            bool inA, inB;
            if (slot < 16) {
                const int bx = blk_x(slot), by = blk_y(slot);
                inA = bx > 0, inB = by > 0;
                ia = inA ? by * 4 + bx - 1 : by * 4 + 3;
                ib = inB ? (by - 1) * 4 + bx : 12 + bx;
            } else {
                const int kk = (slot - 19) & 3, nb = 16 + 4 * ((slot - 19) >> 2);
                const int bx = kk & 1, by = kk >> 1;
                inA = bx > 0, inB = by > 0;
                ia = inA ? nb + 2 * by : nb + 2 * by + 1;
                ib = inB ? nb + bx : nb + 2 + bx;
            }
            const int nA = inA ? info_byte(o0, o1, ia) : info_byte(l0, l1, ia);
            const int nB = inB ? info_byte(o0, o1, ib) : info_byte(t0, t1, ib);
            nC = venc_nc(inA || aL, nA, inB || aT, nB);
        }
        uint32_t nz = 0;
        if (!(has_nnz_entry && info_byte(o0, o1, own_idx) == 0)) { /* empty blocks were never written */
            const uint4 a = ((const uint4 *)(lv_mb + slot * 16))[0];
            strip[0] = a.x, strip[1] = a.y, strip[2] = a.z, strip[3] = a.w;
            uint32_t wds[8] = {a.x, a.y, a.z, a.w, 0u, 0u, 0u, 0u};
            if (maxc != 4) { /* chroma DC records hold 4 levels: one 16-byte load is enough */
                const uint4 bq = ((const uint4 *)(lv_mb + slot * 16))[1];
                strip[4] = bq.x, strip[5] = bq.y, strip[6] = bq.z, strip[7] = bq.w;
                wds[4] = bq.x, wds[5] = bq.y, wds[6] = bq.z, wds[7] = bq.w;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                nz |= (uint32_t)((wds[i] & 0xffffu) != 0u) << (2 * i);
                nz |= (uint32_t)((wds[i] >> 16) != 0u) << (2 * i + 1);
            }
            nz = (nz >> first) & ((1u << maxc) - 1u);
        }
        venc_cavlc_block_mem(&bs, (const int16_t *)strip, nz, first, maxc, nC, s_tab);
    }
    ubits[unit] = (uint16_t)venc_bs_flush(&bs);
  }
}
