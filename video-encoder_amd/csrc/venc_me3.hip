/*
 * venc_me3.hip -- k_me_pool: the shipped motion search.
 *
 * Same search as k_me_tile (venc_me2.hip): exact +-16 full search of every 16x16 macroblock by
 * successive elimination on 8x8 block sums built inside the kernel; the result is the argmin of
 * (SAD << 11) | rank, bit for bit the exhaustive search's.  What changed is how the chip is used
 * (phase times of k_me_tile, tools/me_phase.py, 1080p x 10 items, 112 us per launch: window
 * staging 26, horizontal sums 6, vertical sums 13, lower bounds 21, start points + survivor scan
 * 18, survivor evaluation 29):
 *
 *  1. PERSISTENT BLOCKS WITH A REGISTER PREFETCH.  A block walks over many 8x2-macroblock tiles;
 *     right after a tile's window is in LDS it issues the global loads of the NEXT tile (reference
 *     window, current pixels, last frame's vectors) into registers, where they land while the
 *     sums, bounds and SADs of the present tile are computed.  No wave ever waits on HBM/L2.
 *     Blocks b, b+8, b+16 ... (one XCD under the observed round-robin placement -- speed only,
 *     never correctness) take consecutive tiles, so the halos they share meet in one L2.
 *  2. EDGE TILES TAKE THE SAME 16-BYTE LOADS.  Window pieces are 16 bytes at 16-byte-aligned
 *     columns and the coded width is a multiple of 16, so a piece lies entirely inside or entirely
 *     outside the picture: outside pieces are loaded from the clamped address and replaced by a
 *     broadcast of the edge byte.
 *  3. THE SURVIVORS OF A TILE ARE POOLED.  k_me_tile let each half-wave work through its own
 *     macroblock's survivors, so a tile waited for its worst macroblock (median 1 survivor, p99 25).
 *     Here all survivors of the 16 macroblocks go into one LDS pool and the 32 quarter-waves of the
 *     block take them round-robin; the running best of every macroblock lives in LDS (ds_min_u32 on
 *     the key), and a candidate whose bound already exceeds it is dropped unevaluated.  Evaluation
 *     order does not matter: a skipped candidate has SAD >= LB > best-so-far >= final best.
 *  4. A SAD IS TAKEN BY 16 LANES (one 16-pixel row each, reduced inside a DPP row), four candidates
 *     per wave instruction stream instead of two.
 *  5. Lower bounds are kept only in packed form (LB << 6 | cy): one v_lshl_or per candidate feeds
 *     both the argmin-of-bounds start point and the survivor test.
 */

#define T3_POOL (T2_MBX * T2_MBY * SEA_LIST) /* 1024 entries: (LB << 15) | (macroblock << 11) | candidate */
#define T3_CW 32                             /* current tile row stride in dwords: 128 pixels */

struct me3_pref {
    uint4 va, vb; /* window pieces tid and tid + 512 */
    uint2 cu;     /* 8 current pixels: row tid >> 4, piece tid & 15 */
    uint32_t pv;  /* last frame's vector of macroblock tid >> 5 */
};

DEVFN uint4 me3_window_piece(const uint8_t *ref, int rstride, const venc_geom &g, int x0, int y0, int i) {
    const int r = i / 10, q = i - r * 10;
    const int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
    const int x = x0 - 16 + 16 * q;
    return *(const uint4 *)(ref + (size_t)ry * rstride + clampi(x, 0, g.cw - 16)); /* me3_edge_fix() completes it */
}
/* a piece outside the picture was loaded from the clamped address: replace it by the edge byte (done when the
 * piece goes to LDS, so that nothing waits on the load while it is being prefetched) */
DEVFN uint4 me3_edge_fix(uint4 v, const venc_geom &g, int x0, int i) {
    const int q = i - (i / 10) * 10;
    const int x = x0 - 16 + 16 * q;
    if (x < 0) {
        const uint32_t e = (v.x & 255u) * 0x01010101u;
        v = make_uint4(e, e, e, e);
    } else if (x >= g.cw) {
        const uint32_t e = (v.w >> 24) * 0x01010101u;
        v = make_uint4(e, e, e, e);
    }
    return v;
}

DEVFN void me3_tile_of(int t, int tiles_x, int tiles_y, int &item, int &tx, int &ty) {
    const int per = tiles_x * tiles_y;
    item = t / per;
    const int r = t - item * per;
    ty = r / tiles_x;
    tx = r - ty * tiles_x;
}

/* Tile queues of a step: queue r holds the tiles [n*r/8, n*(r+1)/8) behind a head counter (one per 128-byte
 * line: eight heads in one line serialise like one word, ~11 ns per pull) that k_inter_luma re-zeroes.
 * me3_pull() issues the pull on the home queue and returns at once; me3_resolve() turns the ticket into a tile
 * much later (its latency hides behind a tile's arithmetic) and steals from the other queues when the home queue
 * has run dry, so that every queue drains and every block exits. */
#define ME3_CTR_STRIDE 32
DEVFN int me3_q_lo(int n_tiles, int r) { return (int)(((long long)n_tiles * r) >> 3); }
DEVFN uint32_t me3_pull(uint32_t *ctr, int home) { return atomicAdd(&ctr[home * ME3_CTR_STRIDE], 1u); }
DEVFN int me3_resolve(uint32_t *ctr, int home, int n_tiles, uint32_t ticket) {
    {
        const int lo = me3_q_lo(n_tiles, home), hi = me3_q_lo(n_tiles, home + 1);
        if (ticket < (uint32_t)(hi - lo)) return lo + (int)ticket;
    }
    for (int k = 1; k < 8; k++) {
        const int r = (home + k) & 7;
        const int lo = me3_q_lo(n_tiles, r), hi = me3_q_lo(n_tiles, r + 1);
        if (lo == hi) continue;
        const uint32_t i = atomicAdd(&ctr[r * ME3_CTR_STRIDE], 1u);
        if (i < (uint32_t)(hi - lo)) return lo + (int)i;
    }
    return n_tiles;
}

DEVFN void me3_prefetch(const venc_geom &g, const venc_bufs &B, const venc_step &S, int t, int tiles_x, int tiles_y, int tid,
                        me3_pref &p) {
    int item, tx, ty;
    me3_tile_of(t, tiles_x, tiles_y, item, tx, ty);
    const int x0 = tx * T2_MBX * 16, y0 = (S.row0 + ty * T2_MBY) * 16;
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.parity ^ 1, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    p.va = me3_window_piece(ref, rstride, g, x0, y0, tid);
    p.vb = make_uint4(0, 0, 0, 0);
    if (tid + 512 < T2_WR * 10) p.vb = me3_window_piece(ref, rstride, g, x0, y0, tid + 512);
    {
        const int cy_ = min(y0 + (tid >> 4), g.h - 1), cx_ = x0 + (tid & 15) * 8;
        const uint8_t *row = cur + (size_t)cy_ * cstride;
        if (((g.w & 7) == 0) && cx_ + 8 <= g.w) {
            p.cu = *(const uint2 *)(row + cx_);
        } else {
            p.cu.x = (uint32_t)row[min(cx_, g.w - 1)] | ((uint32_t)row[min(cx_ + 1, g.w - 1)] << 8) |
                     ((uint32_t)row[min(cx_ + 2, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 3, g.w - 1)] << 24);
            p.cu.y = (uint32_t)row[min(cx_ + 4, g.w - 1)] | ((uint32_t)row[min(cx_ + 5, g.w - 1)] << 8) |
                     ((uint32_t)row[min(cx_ + 6, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 7, g.w - 1)] << 24);
        }
    }
    {
        const int hw = tid >> 5;
        const int mx = tx * T2_MBX + (hw & (T2_MBX - 1)), my = S.row0 + ty * T2_MBY + (hw >> 3);
        p.pv = B.mvsad_prev[(size_t)item * g.nmb + (size_t)min(my, g.mbh - 1) * g.mbw + min(mx, g.mbw - 1)];
    }
}

/* SADs of four candidates per wave: quarter-wave q = lane >> 4 takes candidate (cx, cy) of macroblock
 * (col0, row0) -- its 16 lanes own one row each -- and the totals come back as the value of lane 15 of
 * every DPP row, broadcast to the row through readlane + select. */
DEVFN uint32_t me3_quarter_sad(const uint32_t *s_win, const uint32_t *s_cur, int col0, int row0, int lane, int cx, int cy) {
    const int r = lane & 15;
    const uint4 c = *(const uint4 *)(s_cur + (row0 + r) * T3_CW + (col0 >> 2));
    const int boff = col0 + cx;
    const uint32_t *p = s_win + (row0 + cy + r) * T2_WD + (boff >> 2);
    const uint32_t sh = (uint32_t)(boff & 3);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
    uint32_t a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d1, d0, sh), c.x, 0u);
    a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d2, d1, sh), c.y, a);
    a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d3, d2, sh), c.z, a);
    a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d4, d3, sh), c.w, a);
    a += dpp_fetch<DPP_ROW_SHR(1), 0xf>(a, 0u);
    a += dpp_fetch<DPP_ROW_SHR(2), 0xf>(a, 0u);
    a += dpp_fetch<DPP_ROW_SHR(4), 0xf>(a, 0u);
    a += dpp_fetch<DPP_ROW_SHR(8), 0xf>(a, 0u);
    return a; /* complete in lane 15 of the quarter */
}
DEVFN uint32_t me3_quarter_bcast(uint32_t a, int lane) {
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), q1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 31);
    const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47), q3 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    return lane < 16 ? q0 : lane < 32 ? q1 : lane < 48 ? q2 : q3;
}

/* PERSIST = true: persistent blocks pulling tiles from the queues, next tile prefetched (header, item 1).
 * PERSIST = false: one tile per block (tile = blockIdx.x), no queue, no prefetch registers. */
template <bool PERSIST>
__global__ __launch_bounds__(512) void k_me_pool(venc_geom g, venc_bufs B, venc_step S, int tiles_x, int tiles_y, int n_tiles) {
    __shared__ __attribute__((aligned(16))) uint32_t s_win[T2_WR * T2_WD];
    __shared__ __attribute__((aligned(16))) uint32_t s_cur[T2_MBY * 16 * T3_CW];
    __shared__ __attribute__((aligned(16))) uint2 s_h8[T2_WR * T2_HG];
    __shared__ __attribute__((aligned(16))) uint4 s_v8[T2_VR * T2_HG];
    /* the survivor lists and the pool live in the horizontal-sums buffer, which is dead after P2 (8 of its 19.5 KB):
     * 67.6 KB per block, so that two blocks leave room on a CU for the entropy chain's blocks running beside them */
    uint32_t *const s_pool = (uint32_t *)s_h8;
    uint32_t(*const s_list)[SEA_LIST] = (uint32_t(*)[SEA_LIST])((uint32_t *)s_h8 + T3_POOL);
    __shared__ int s_tq[2];
    __shared__ uint32_t s_best[T2_MBX * T2_MBY];
    __shared__ uint32_t s_total;
    __shared__ __attribute__((aligned(4))) uint16_t s_rank[VENC_ME_CANDS + 1];

    const int tid0 = threadIdx.x;
    const uint16_t *rank = s_rank;

    /* tiles are pulled from eight queues (one per group of blocks with equal blockIdx.x % 8: header, item 1);
     * a block whose home queue has run dry takes from the others, so every queue drains and every block exits */
    const int home = (int)(blockIdx.x & 7);
    int t = (int)blockIdx.x, t_next = n_tiles;
    if (PERSIST) {
        if (tid0 == 0) {
            const uint32_t k0 = me3_pull(B.me_ctr, home), k1 = me3_pull(B.me_ctr, home);
            s_tq[0] = me3_resolve(B.me_ctr, home, n_tiles, k0);
            s_tq[1] = me3_resolve(B.me_ctr, home, n_tiles, k1);
        }
        __syncthreads();
        t = s_tq[0], t_next = s_tq[1];
    }
    if (t >= n_tiles) return;

    {
        const uint32_t *rank32 = (const uint32_t *)B.me_rank; /* 1089 u16 = 545 dwords (table padded by the launcher) */
        ((uint32_t *)s_rank)[tid0] = rank32[tid0];
        if (tid0 + 512 < 545) ((uint32_t *)s_rank)[tid0 + 512] = rank32[tid0 + 512];
    }
    me3_pref pf;
    me3_prefetch(g, B, S, t, tiles_x, tiles_y, tid0, pf);

    for (;;) {
        /* The lane's indices are re-derived from an opaque copy of the thread id in every trip: hoisted out of
         * the tile loop, the dozens of per-lane LDS addresses of the phases below stayed live across all of
         * them and cost 40+ registers (and with them the fourth wave per SIMD). */
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, l32 = lane & 31, hbase = lane & 32, hw = tid >> 5;
        const int mbx = hw & (T2_MBX - 1), mby = hw >> 3;
        const int col0 = mbx * 16, row0 = mby * 16;
        int item, tx, ty;
        me3_tile_of(t, tiles_x, tiles_y, item, tx, ty);
        const int mx = tx * T2_MBX + mbx, my = S.row0 + ty * T2_MBY + mby;
        const bool mb_valid = mx < g.mbw && my < S.row1;
        const size_t mbidx = (size_t)item * g.nmb + (size_t)min(my, g.mbh - 1) * g.mbw + min(mx, g.mbw - 1);
        /* ---- P0: registers -> LDS ---- */
        {
            const int x0 = tx * T2_MBX * 16;
            const int ra = tid / 10, qa = tid - ra * 10;
            uint32_t *da = s_win + ra * T2_WD + 4 * qa;
            const uint4 va = me3_edge_fix(pf.va, g, x0, tid);
            *(uint2 *)da = make_uint2(va.x, va.y); /* row stride is 8-byte aligned only */
            *(uint2 *)(da + 2) = make_uint2(va.z, va.w);
            if (tid + 512 < T2_WR * 10) {
                const int i1 = tid + 512, rb = i1 / 10, qb = i1 - rb * 10;
                uint32_t *db = s_win + rb * T2_WD + 4 * qb;
                const uint4 vb = me3_edge_fix(pf.vb, g, x0, i1);
                *(uint2 *)db = make_uint2(vb.x, vb.y);
                *(uint2 *)(db + 2) = make_uint2(vb.z, vb.w);
            }
            if (tid < T2_WR) {
                s_win[tid * T2_WD + 40] = 0;
                s_win[tid * T2_WD + 41] = 0;
            }
            *(uint2 *)(s_cur + (tid >> 4) * T3_CW + (tid & 15) * 2) = pf.cu;
        }
        const uint32_t prev = pf.pv;
        uint32_t mycnt = 0; /* survivors of this lane's macroblock besides the start points */
        __syncthreads();
        /* the next tile's loads fly while this one is searched; the tile after it is pulled from the queue now and
         * read at the end of the trip (three barriers later) */
        const bool more = PERSIST && t_next < n_tiles;
        if (more) me3_prefetch(g, B, S, t_next, tiles_x, tiles_y, tid, pf);
        uint32_t ticket = 0; /* the tile after the next: pulled now, resolved before the last barrier of the trip */
        if (PERSIST && tid == 0 && more) ticket = me3_pull(B.me_ctr, home);
#define ME3_POST_TICKET()                                                                            \
    do {                                                                                             \
        if (PERSIST && tid == 0) s_tq[0] = more ? me3_resolve(B.me_ctr, home, n_tiles, ticket) : n_tiles; \
    } while (0)
        if (S.me_stop == 1) {
            ME3_POST_TICKET();
            goto next_tile_sync;
        }
        /* ---- P1: horizontal 8-sums, four columns per item (as k_me_tile) ---- */
        {
            const int r = tid >> 3, q0 = (tid & 7) * 5;
            const uint32_t *p = s_win + r * T2_WD + q0;
            uint32_t dwd[7];
#pragma unroll
            for (int k = 0; k < 7; k++) dwd[k] = p[k];
            uint2 h4[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const unsigned long long a = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)dwd[k + 1] << 32) | dwd[k], 0u, 0ull);
                h4[k] = make_uint2((uint32_t)a, (uint32_t)(a >> 32));
            }
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (q0 + k < T2_HG) s_h8[r * T2_HG + q0 + k] = pk4_add(h4[k], h4[k + 1]);
            if (tid == 0) s_total = 0;
        }
        __syncthreads();
        if (S.me_stop == 2) {
            ME3_POST_TICKET();
            goto next_tile_sync;
        }
        /* ---- P2: vertical 8-sums as vertical pairs (as k_me_tile) ---- */
        if (tid < 7 * T2_HG) {
            const int seg = tid / T2_HG, q = tid - seg * T2_HG;
            const int r0 = seg * 8;
            const uint2 *col = s_h8 + r0 * T2_HG + q;
            uint2 sr[16];
            uint2 acc = col[0];
#pragma unroll
            for (int k = 1; k < 8; k++) acc = pk4_add(acc, col[k * T2_HG]);
            sr[0] = acc;
#pragma unroll
            for (int k = 1; k < 16; k++) {
                if (r0 + k <= T2_VR + 7) acc = pk4_sub(pk4_add(acc, col[(k + 7) * T2_HG]), col[(k - 1) * T2_HG]);
                sr[k] = acc;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (r0 + k < T2_VR) {
                    const uint2 lo = sr[k], hi = sr[k + 8];
                    uint4 o;
                    o.x = (lo.x & 0xffffu) | (hi.x << 16);
                    o.y = (lo.x >> 16) | (hi.x & 0xffff0000u);
                    o.z = (lo.y & 0xffffu) | (hi.y << 16);
                    o.w = (lo.y >> 16) | (hi.y & 0xffff0000u);
                    s_v8[(r0 + k) * T2_HG + q] = o;
                }
            }
        }
        __syncthreads();
        if (S.me_stop == 3) {
            ME3_POST_TICKET();
            goto next_tile_sync;
        }
        {
            /* ---- block sums of the current macroblock's four 8x8 quadrants ---- */
            const uint2 cq = *(const uint2 *)(s_cur + (row0 + (l32 >> 1)) * T3_CW + (col0 >> 2) + (l32 & 1) * 2);
            uint32_t part = __builtin_amdgcn_sad_u8(cq.x, 0u, 0u);
            part = __builtin_amdgcn_sad_u8(cq.y, 0u, part);
            part += (uint32_t)__shfl_xor((int)part, 2);
            part += (uint32_t)__shfl_xor((int)part, 4);
            part += (uint32_t)__shfl_xor((int)part, 8);
            const uint32_t q00 = (uint32_t)__shfl((int)part, hbase | 0), q01 = (uint32_t)__shfl((int)part, hbase | 1);
            const uint32_t q10 = (uint32_t)__shfl((int)part, hbase | 16), q11 = (uint32_t)__shfl((int)part, hbase | 17);
            const uint32_t cleft = q00 | (q10 << 16), cright = q01 | (q11 << 16);

            /* ---- P3: packed lower bounds pk = LB << 6 | cy; lane = window column cx = l32 ---- */
            const uint32_t *v8 = (const uint32_t *)s_v8;
            uint32_t pk[33];
            {
                const uint32_t *colp = v8 + row0 * T2_SU + col0 + l32;
                /* three batches of 11 rows: at most 22 sums in flight (one batch of 33 would hold 66 registers) */
#pragma unroll
                for (int c0 = 0; c0 < 33; c0 += 11) {
#pragma unroll
                    for (int cy = c0; cy < c0 + 11; cy++) {
                        const uint32_t lb = __builtin_amdgcn_sad_u16(colp[cy * T2_SU + 8], cright,
                                                                     __builtin_amdgcn_sad_u16(colp[cy * T2_SU], cleft, 0u));
                        pk[cy] = (lb << 6) | (uint32_t)cy;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            uint32_t lbx0, lbx1 = 0xffffffffu; /* column cx = 32: lane l32 owns cy = l32, lane 0 also cy = 32 */
            {
                const uint32_t *colp = v8 + row0 * T2_SU + col0 + 32;
                lbx0 = __builtin_amdgcn_sad_u16(colp[l32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[l32 * T2_SU], cleft, 0u));
                if (l32 == 0) lbx1 = __builtin_amdgcn_sad_u16(colp[32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[32 * T2_SU], cleft, 0u));
            }
            /* candidate with the smallest bound (any of them on ties: start points only steer the pruning) */
            uint32_t m = pk[0];
#pragma unroll
            for (int cy = 1; cy < 33; cy += 2) m = min(min(m, pk[cy]), pk[cy + 1]);
            /* (LB << 12) | candidate index, as one comparable word */
            uint32_t lmin = ((m >> 6) << 12) | ((m & 63u) * 33u + (uint32_t)l32);
            lmin = min(lmin, (lbx0 << 12) | (uint32_t)(l32 * 33 + 32));
            if (l32 == 0) lmin = min(lmin, (lbx1 << 12) | (uint32_t)(32 * 33 + 32));
            const int ai = (int)(half_min32(lmin, lane) & 4095u);
            if (S.me_stop == 4) {
                if (l32 == 0 && mb_valid) B.mvsad[mbidx] = (uint32_t)ai << 16;
                ME3_POST_TICKET();
                goto next_tile_sync;
            }
            /* ---- P4a: four start points per macroblock, two quarter-wave passes:
             *      zero vector | last frame's vector,  smallest bound | the pair macroblock's last vector ---- */
            const int acy = ai / 33, acx = ai - acy * 33;
            const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
            const uint32_t oprev = (uint32_t)__shfl((int)prev, lane ^ 32);
            const int ocx = (int)(int8_t)(oprev & 255u) + 16, ocy = (int)(int8_t)((oprev >> 8) & 255u) + 16;
            const bool second = (lane & 16) != 0;
            const int ci_z = 16 * 33 + 16, ci_p = pcy * 33 + pcx, ci_o = ocy * 33 + ocx;
            const uint32_t s1 = me3_quarter_bcast(me3_quarter_sad(s_win, s_cur, col0, row0, lane, second ? pcx : 16, second ? pcy : 16), lane);
            const uint32_t s2 = me3_quarter_bcast(me3_quarter_sad(s_win, s_cur, col0, row0, lane, second ? ocx : acx, second ? ocy : acy), lane);
            uint32_t key = min((s1 << 11) | rank[second ? ci_p : ci_z], (s2 << 11) | rank[second ? ci_o : ai]);
            key = min(key, (uint32_t)__shfl_xor((int)key, 16)); /* both quarters of the macroblock */
            const uint32_t thr = key >> 11;
            if (l32 == 0) s_best[hw] = key;
            /* ---- survivors: bound <= best start SAD.  Compaction by wave ballots (no atomics): a row of 33 bounds
             * rarely holds a survivor, so the common case is one compare and one scalar branch per row ---- */
            {
                const uint32_t thrk = (thr << 6) | 63u;
                const uint32_t tag = (uint32_t)hw << 11;
                const uint32_t lt = (1u << l32) - 1u;
                uint32_t *list = s_list[hw];
                mycnt = 0;
#pragma unroll
                for (int cy = 0; cy < 33; cy++) {
                    const bool hit = mb_valid && pk[cy] <= thrk;
                    if (__ballot(hit) != 0ull) {
                        /* the rare path: the empty asm keeps its arithmetic inside the branch (hoisted, the 33 copies
                         * of it cost 60+ registers) */
                        uint32_t pv = pk[cy];
                        asm volatile("" : "+v"(pv));
                        const uint32_t ci = (pv & 63u) * 33u + (uint32_t)l32;
                        const bool keep = hit && ci != (uint32_t)ci_z && ci != (uint32_t)ci_p && ci != (uint32_t)ai && ci != (uint32_t)ci_o;
                        const unsigned long long kb = __ballot(keep);
                        const uint32_t hm = (uint32_t)(kb >> hbase);
                        const uint32_t pos = mycnt + (uint32_t)__popc(hm & lt);
                        if (keep && pos < SEA_LIST) list[pos] = ((pv >> 6) << 15) | tag | ci;
                        mycnt += (uint32_t)__popc(hm);
                    }
                }
                {   /* column cx = 32 */
                    const bool hit0 = mb_valid && lbx0 <= thr, hit1 = mb_valid && lbx1 <= thr;
                    if (__ballot(hit0 || hit1) != 0ull) {
                        const uint32_t ci0 = (uint32_t)(l32 * 33 + 32), ci1 = (uint32_t)(32 * 33 + 32);
                        const bool k0 = hit0 && ci0 != (uint32_t)ci_p && ci0 != (uint32_t)ai && ci0 != (uint32_t)ci_o;
                        const bool k1 = hit1 && ci1 != (uint32_t)ci_p && ci1 != (uint32_t)ai && ci1 != (uint32_t)ci_o;
                        const uint32_t hm0 = (uint32_t)(__ballot(k0) >> hbase), hm1 = (uint32_t)(__ballot(k1) >> hbase);
                        const uint32_t pos0 = mycnt + (uint32_t)__popc(hm0 & lt);
                        if (k0 && pos0 < SEA_LIST) list[pos0] = (lbx0 << 15) | tag | ci0;
                        mycnt += (uint32_t)__popc(hm0);
                        if (k1 && mycnt < SEA_LIST) list[mycnt] = (lbx1 << 15) | tag | ci1; /* lane 0 only */
                        mycnt += (uint32_t)__popc(hm1);
                    }
                }
                /* the macroblock's list joins the tile's pool (overflowing macroblocks go to k_me_fix instead) */
                const uint32_t n_copy = mycnt > SEA_LIST ? 0u : mycnt;
                uint32_t base = 0;
                if (l32 == 0 && n_copy) base = atomicAdd(&s_total, n_copy);
                base = (uint32_t)__shfl((int)base, hbase);
                for (uint32_t i = (uint32_t)l32; i < n_copy; i += 32u) s_pool[base + i] = list[i];
            }
        }
        ME3_POST_TICKET();
        __syncthreads();
        if (S.me_stop == 5) {
            if (l32 == 0 && mb_valid) B.mvsad[mbidx] = s_best[hw] << 5;
            goto next_tile_sync;
        }
        /* ---- P4b: the tile's survivors, round-robin over the block's 32 quarter-waves ---- */
        {
            const uint32_t total = s_total;
            for (uint32_t k = (uint32_t)(tid >> 4); k < ((total + 31u) & ~31u); k += 32u) {
                const uint32_t e = k < total ? s_pool[k] : 0xffffffffu;
                const int ehw = (int)((e >> 11) & 15u), ci = (int)(e & 2047u);
                bool need = k < total;
                if (need) need = (e >> 15) <= (s_best[ehw] >> 11);
                if (__ballot(need) == 0ull) continue;
                const int cy = need ? ci / 33 : 16, cx = need ? ci - (ci / 33) * 33 : 16;
                const int ecol = need ? (ehw & (T2_MBX - 1)) * 16 : 0, erow = need ? (ehw >> 3) * 16 : 0;
                const uint32_t sad = me3_quarter_sad(s_win, s_cur, ecol, erow, lane, cx, cy);
                if (need && (lane & 15) == 15) atomicMin(&s_best[ehw], (sad << 11) | rank[cy * 33 + cx]);
            }
        }
        __syncthreads();
        if (l32 == 0 && mb_valid) {
            const uint32_t best = s_best[hw], cnt = mycnt;
            const bool overflow = cnt > SEA_LIST;
            const uint32_t cand = B.me_cand[best & 2047u];
            const int dx = (int)(cand & 255u) - 16, dy = (int)(cand >> 8) - 16;
            B.mvsad[mbidx] = ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
            B.me_flag[mbidx] = overflow ? 1 : 0;
            if (overflow) B.fix_list[(size_t)item * g.nmb + atomicAdd(&B.fix_count[item], 1u)] = (uint32_t)(my * g.mbw + mx);
            B.me_cnt[mbidx] = (uint8_t)min(cnt, 255u); /* survivors of the bound test besides the start points (introspection) */
        }
        goto next_tile;
    next_tile_sync:
        __syncthreads();
    next_tile:
        if (!more) break;
        t = t_next;
        t_next = s_tq[0];
        /* s_win / s_cur are rewritten at the top: every wave is past its last read of them (the barrier
         * before the result phase), and s_cnt / s_best are next written after the coming P0 barrier */
    }
}
