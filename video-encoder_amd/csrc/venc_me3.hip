/*
 * venc_me3.hip -- k_me_pool: the motion search on three blocks per CU.  Since round 3 the shipped
 * search is k_me_pool4 (venc_me4.hip: the same search laid out for four blocks per CU); this kernel stays behind VENC_ME_BLOCKS=3 as
 * its A/B partner and as the carrier of round 2's inline transform (VENC_FUSE=1), and venc_me4.hip uses its helpers.
 *
 * Same search as k_me_tile (venc_me2.hip): exact +-16 full search of every 16x16 macroblock by
 * successive elimination on 8x8 block sums built inside the kernel; the result is the argmin of
 * (SAD << 11) | rank, bit for bit the exhaustive search's.  One block = one 8x2-macroblock tile,
 * one half-wave per macroblock for the bounds.  What changed against k_me_tile (whose phase times,
 * tools/me_phase.py, 1080p x 10 items, were: window staging 26 us, horizontal sums 6, vertical sums
 * 13, lower bounds 21, start points + survivor scan 18, survivor evaluation 29 = 112 us per launch):
 *
 *  1. THE SURVIVORS OF A TILE ARE POOLED.  k_me_tile let each half-wave work through its own
 *     macroblock's survivors, so a tile waited for its worst macroblock (median 1 survivor, p99 25).
 *     Here all survivors of the 16 macroblocks go into one LDS pool and the 32 quarter-waves of the
 *     block take them round-robin; the running best of every macroblock lives in LDS (ds_min_u32 on
 *     the key), and a candidate whose bound already exceeds it is dropped unevaluated.  Evaluation
 *     order does not matter: a skipped candidate has SAD >= LB > best-so-far >= final best.  (29 -> 5 us)
 *  2. A SAD IS TAKEN BY 16 LANES (one 16-pixel row each, reduced inside a DPP row): four candidates
 *     per wave instruction stream instead of two.  The current tile sits in LDS for that.
 *  3. THE SURVIVOR SCAN COMPACTS WITH WAVE BALLOTS: a row of bounds without a survivor costs one
 *     compare and one scalar branch; rows with survivors append to the pool with one LDS atomic per
 *     wave and row.  No per-macroblock list, no copy.
 *  4. EDGE TILES TAKE THE SAME 16-BYTE LOADS.  Window pieces are 16 bytes at 16-byte-aligned columns
 *     and the coded width is a multiple of 16, so a piece lies entirely inside or entirely outside
 *     the picture: outside pieces are loaded from the clamped address and replaced by a broadcast of
 *     the edge byte.
 *  5. THREE BLOCKS PER CU.  The vertical-sums buffer overlays the horizontal-sums buffer it is built
 *     from (all reads, a barrier, all writes): 51 KB of LDS per block instead of 68, and the register
 *     budget is held at 84, so six waves per SIMD hide the LDS and DPP latencies that four left bare.
 *
 *  6. THE BOUNDS ARE TESTED AS THEY ARE COMPUTED (round 2).  Start points first (last frame's vector, zero vector);
 *     their smaller SAD prunes the 1089 bounds on the fly, four rows per wave-uniform branch, nothing kept.  Only a
 *     wave whose start points were poor (> T3_RETRY survivors in one of its macroblocks) computes the bounds again,
 *     keeps them, takes the candidate with the smallest bound as a third start point and rebuilds its segment --
 *     which every wave did before.  843 -> 771 VALU instructions per wave; 2160p launch 284 -> 270 us.
 *
 * Tried and dropped (measured, same clip): (a) searching the macroblocks with too many survivors exhaustively
 * at the end of their own block, from the window already in LDS, instead of listing them for k_me_fix: such
 * macroblocks cluster, a tile with 16 of them became a 40 us straggler and the launch grew from 83 to 109 us;
 * (b) persistent blocks pulling tiles from per-XCD queues with a
 * register prefetch of the next tile.  Alone it was on par (96 us); beside the entropy chain that
 * runs on a second stream it lost 7 % of the whole step, because blocks that never retire keep the
 * other chain's blocks off the CUs they hold.
 */

#define T3_POOL 1024 /* survivor pool of a tile = 8 wave segments of 2 x T3_CAP; entry = (LB << 15) | (macroblock << 11) | candidate */
#define T3_CAP SEA_LIST /* survivors one macroblock may put into the pool (64); more: exhaustive search by k_me_fix */
#ifndef T3_RETRY
#define T3_RETRY 32 /* a wave one of whose macroblocks keeps more survivors than this takes the second pass (third start point) */
#endif
#define T3_CW 36 /* current tile row stride in dwords: 128 pixels + 16 bytes.  A SAD's 16 lanes read 16 bytes each of 16 consecutive rows
                    (ds_read_b128, banks of 64 dwords): at a stride of 32 dwords rows r and r + 2 share their banks -- an 8-way conflict,
                    28 extra LDS cycles per read, the 5.7 M conflict cycles per launch of the round-1 counters; at 36 the 16 rows
                    start at 16 different multiples of four banks */

/* window piece i (row i / 10, columns 16 * (i % 10) ...): 16 bytes from the clamped address; a piece outside the
 * picture is replaced by the edge byte */
DEVFN uint4 me3_window_piece(const uint8_t *ref, int rstride, const venc_geom &g, int x0, int y0, int i) {
    const int r = i / 10, q = i - r * 10;
    const int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
    const int x = x0 - 16 + 16 * q;
    uint4 v = *(const uint4 *)(ref + (size_t)ry * rstride + clampi(x, 0, g.cw - 16));
    if (x < 0) {
        const uint32_t e = (v.x & 255u) * 0x01010101u;
        v = make_uint4(e, e, e, e);
    } else if (x >= g.cw) {
        const uint32_t e = (v.w >> 24) * 0x01010101u;
        v = make_uint4(e, e, e, e);
    }
    return v;
}

/* SADs of four candidates per wave: quarter-wave q = lane >> 4 takes candidate (cx, cy) of macroblock
 * (col0, row0) -- its 16 lanes own one row each; the total is complete in lane 15 of the quarter. */
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
    return a;
}

/* RC: the rate-control variant (spec/CODEC.md section 1a) -- a template parameter, so that the fixed-QP kernel carries neither
 * its branches nor a second set of quantiser constants in scalar registers */
template <bool RC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_me_pool(venc_geom g, venc_bufs B, venc_step S, int tiles_x, int tiles_y, uint32_t tx_magic, int xcd_runs, int fuse_luma) {
    __shared__ __attribute__((aligned(16))) uint32_t s_win[T2_WR * T2_WD];
    __shared__ __attribute__((aligned(16))) uint32_t s_cur[T2_MBY * 16 * T3_CW];
    /* horizontal sums (P1 -> P2), then, after a barrier, the vertical pairs built from them (P2 -> P3) */
    __shared__ __attribute__((aligned(16))) uint4 s_hv[T2_VR * T2_HG];
    __shared__ uint32_t s_pool[T3_POOL];
    static_assert(8 * 2 * T3_CAP <= T3_POOL, "eight wave segments of two macroblocks each");
    /* running best per macroblock: ((SAD << 11 | rank) << 11) | candidate index -- ordered like the search's key (the
     * rank is unique), and the winner's displacement comes out of the same word: no rank -> candidate table, no global
     * load at the tail of the block */
    __shared__ __attribute__((aligned(8))) unsigned long long s_best[T2_MBX * T2_MBY];
    __shared__ uint32_t s_ovf[T2_MBX * T2_MBY];
    __shared__ __attribute__((aligned(16))) uint32_t s_wcnt[8]; /* entries in each wave's pool segment */
    __shared__ uint32_t s_prev[T2_MBX * T2_MBY];
    __shared__ int s_q[sizeof(venc_qparams) / sizeof(int)]; /* rate control: the picture's quantiser constants, fetched in P0 for P5 */
    __shared__ __attribute__((aligned(8))) uint32_t s_c8[T2_MBX * T2_MBY][2]; /* per macroblock: TL | BL << 16, TR | BR << 16 (8x8 sums of the current pixels) */
    __shared__ __attribute__((aligned(4))) uint16_t s_rank[VENC_ME_CANDS + 1];
    uint2 *const s_h8 = (uint2 *)s_hv; /* T2_WR * T2_HG uint2 = 19968 B of the 30576 */
    uint4 *const s_v8 = s_hv;
    static_assert(sizeof(uint2) * T2_WR * T2_HG <= sizeof(uint4) * T2_VR * T2_HG, "h8 must fit inside the v8 buffer");

    const int tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, hbase = lane & 32, hw = tid >> 5;
    const int mbx = hw & (T2_MBX - 1), mby = hw >> 3;
    const int col0 = mbx * 16, row0 = mby * 16;
    const uint16_t *rank = s_rank;
    /* Tile order.  Default: raster order over the picture (grid.x = tiles of a picture).  Workgroups are dealt
     * round-robin to the 8 XCDs (b and b + 8 share one, each XCD has its own 4 MB L2), so in raster order the tiles that
     * share a window halo -- left/right and upper/lower neighbours -- sit on different XCDs and every halo crosses the
     * fabric again: rocprofv3's FETCH_SIZE shows 2.5x the algorithmic bytes.  xcd_runs != 0 (VENC_ME_XCD=1) cuts the
     * raster order into runs of 64 tiles that go to the XCDs in rotation (grid.x is then a multiple of 8 and
     * blockIdx.x & 7 labels the XCD): the traffic falls to 1.03-1.14x the algorithmic bytes -- and the kernel gets 1-5 %
     * SLOWER at 1080p (neutral at 2160p), whether an XCD gets one contiguous eighth of the launch, strips of four tile
     * rows or runs of 64 tiles.  The kernel is bound by VALU and LDS, not by these fetches (which the 256 MB last-level
     * cache serves), and neighbours on one XCD pull the same L2 lines at the same time.  Hence off by default; kept for
     * memory-bound company.  Placement is never a correctness matter. */
    const int item = (int)blockIdx.z + S.item0;
    const int kx = (int)(blockIdx.x >> 3);
    const int tile = xcd_runs ? ((((kx >> 6) << 3) + (int)(blockIdx.x & 7u)) << 6) + (kx & 63) : (int)blockIdx.x;
    if (tile >= tiles_x * tiles_y) return;
    if (tile == 0 && threadIdx.x == 0) {
        B.fix_count_next[item] = 0; /* the other counter: the next P step's search fills it, nobody reads it now */
        if (RC) { /* rate control: the next step's SAD sum starts at zero; a first P picture of a GOP is coded at the base QP (every
                     other picture's QP was left in the ring by the last block of the previous step's k_me_fix) */
            if (!S.rc_history) B.qp_cur[item] = S.qp_base;
            B.sad_next[item] = 0ull;
        }
    }
    /* tx_magic = ceil(2^32 / tiles_x), exact for these sizes; 0 stands for tiles_x == 1 (2^32 does not fit) */
    const int ty = tx_magic ? (int)__umulhi((uint32_t)tile, tx_magic) : tile, tx = tile - ty * tiles_x;
    const int x0 = tx * T2_MBX * 16, y0 = (S.row0 + ty * T2_MBY) * 16;
    const int mx = tx * T2_MBX + mbx, my = S.row0 + ty * T2_MBY + mby;
    const bool mb_valid = mx < g.mbw && my < S.row1;
    const size_t mbidx = (size_t)item * g.nmb + (size_t)min(my, g.mbh - 1) * g.mbw + min(mx, g.mbw - 1);
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);

    /* ---- P0: every global load of the block is issued here, then the window, the current tile and the
     * tie-break ranks go to LDS ---- */
    const uint32_t *rank32 = (const uint32_t *)B.me_rank; /* 1089 u16 = 545 dwords (table padded by the launcher) */
    const uint32_t rk0 = rank32[tid], rk1 = tid + 512 < 545 ? rank32[tid + 512] : 0u;
    const uint32_t prev_early = B.mvsad_prev[mbidx]; /* issued with the other loads; parked in LDS until P4a */
    uint2 cu;
    {
        const int cy_ = min(y0 + (tid >> 4), g.h - 1), cx_ = x0 + (tid & 15) * 8;
        const uint8_t *row = cur + (size_t)cy_ * cstride;
        if (((g.w & 7) == 0) && cx_ + 8 <= g.w) {
            cu = *(const uint2 *)(row + cx_);
        } else {
            cu.x = (uint32_t)row[min(cx_, g.w - 1)] | ((uint32_t)row[min(cx_ + 1, g.w - 1)] << 8) |
                   ((uint32_t)row[min(cx_ + 2, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 3, g.w - 1)] << 24);
            cu.y = (uint32_t)row[min(cx_ + 4, g.w - 1)] | ((uint32_t)row[min(cx_ + 5, g.w - 1)] << 8) |
                   ((uint32_t)row[min(cx_ + 6, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 7, g.w - 1)] << 24);
        }
    }
    {
        const int ra = tid / 10, qa = tid - ra * 10;
        uint32_t *da = s_win + ra * T2_WD + 4 * qa;
        /* a tile whose window lies inside the coded picture (all but the rim: 87 % of the tiles at 1080p) needs neither clamps
         * nor edge replication: one address, one load per piece */
        const bool interior = x0 >= 16 && x0 + 144 <= g.cw && y0 >= 16 && y0 + 48 <= g.ch; /* block-uniform */
        if (interior) {
            const uint8_t *w0 = ref + (size_t)(y0 - 16) * rstride + (x0 - 16);
            const uint4 va = *(const uint4 *)(w0 + ra * rstride + 16 * qa);
            uint4 vb = make_uint4(0u, 0u, 0u, 0u);
            const bool second = tid + 512 < T2_WR * 10;
            const int i1 = tid + 512, rb = i1 / 10, qb = i1 - rb * 10;
            if (second) vb = *(const uint4 *)(w0 + rb * rstride + 16 * qb);
            *(uint2 *)da = make_uint2(va.x, va.y); /* row stride is 8-byte aligned only */
            *(uint2 *)(da + 2) = make_uint2(va.z, va.w);
            if (second) {
                uint32_t *db = s_win + rb * T2_WD + 4 * qb;
                *(uint2 *)db = make_uint2(vb.x, vb.y);
                *(uint2 *)(db + 2) = make_uint2(vb.z, vb.w);
            }
        } else {
        const uint4 va = me3_window_piece(ref, rstride, g, x0, y0, tid);
        *(uint2 *)da = make_uint2(va.x, va.y);
        *(uint2 *)(da + 2) = make_uint2(va.z, va.w);
        if (tid + 512 < T2_WR * 10) {
            const int i1 = tid + 512, rb = i1 / 10, qb = i1 - rb * 10;
            const uint4 vb = me3_window_piece(ref, rstride, g, x0, y0, i1);
            uint32_t *db = s_win + rb * T2_WD + 4 * qb;
            *(uint2 *)db = make_uint2(vb.x, vb.y);
            *(uint2 *)(db + 2) = make_uint2(vb.z, vb.w);
        }
        }
        if (tid < T2_WR) {
            s_win[tid * T2_WD + 40] = 0;
            s_win[tid * T2_WD + 41] = 0;
        }
        *(uint2 *)(s_cur + (tid >> 4) * T3_CW + (tid & 15) * 2) = cu;
        ((uint32_t *)s_rank)[tid] = rk0;
        if (tid + 512 < 545) ((uint32_t *)s_rank)[tid + 512] = rk1;
        if (l32 == 0) s_prev[hw] = prev_early;
        if (RC && tid < (int)(sizeof(venc_qparams) / sizeof(int))) {
            /* (two dependent loads -- the picture's QP, then its row of the table -- issued here, beside the window's, instead of at the
             * head of P5 where every block waited for them: the rate-control variant took 85 us alone where the fixed-QP one takes 74) */
            const int qp0 = S.rc_history ? B.qp_cur[item] : S.qp_base;
            s_q[tid] = ((const int *)(B.qtab + qp0))[tid];
        }
    }
    __syncthreads();
    if (S.me_stop == 1) return;
    /* ---- P1: horizontal 8-sums, four columns per item ---- */
    {   /* thread = (row, run of 5 column groups): 6 sliding-sum instructions give 5 groups of 8-sums */
        const int r = tid >> 3, q0 = (tid & 7) * 5;
        const uint32_t *p = s_win + r * T2_WD + q0;
        uint32_t dwd[7];
#pragma unroll
        for (int k = 0; k < 7; k++) dwd[k] = p[k]; /* q0 + 6 <= 41: inside the padded row */
        uint2 h4[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const unsigned long long a = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)dwd[k + 1] << 32) | dwd[k], 0u, 0ull);
            h4[k] = make_uint2((uint32_t)a, (uint32_t)(a >> 32));
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (q0 + k < T2_HG) s_h8[r * T2_HG + q0 + k] = pk4_add(h4[k], h4[k + 1]);
        /* the four 8x8 sums of every current macroblock, for the bounds of P3: thread = (macroblock, left / right
         * column of 8x8 blocks), 16 rows of 8 pixels each (done here, off P3's critical path, instead of seven
         * cross-lane exchanges per half-wave there) */
        if (tid < 2 * T2_MBX * T2_MBY) {
            const int m = tid >> 1, half = tid & 1;
            const uint32_t *cp = s_cur + ((m >> 3) * 16) * T3_CW + (m & (T2_MBX - 1)) * 4 + half * 2;
            uint32_t top = 0, bot = 0;
#pragma unroll
            for (int y = 0; y < 8; y++) {
                const uint2 a = *(const uint2 *)(cp + y * T3_CW), bq = *(const uint2 *)(cp + (y + 8) * T3_CW);
                top = __builtin_amdgcn_sad_u8(a.y, 0u, __builtin_amdgcn_sad_u8(a.x, 0u, top));
                bot = __builtin_amdgcn_sad_u8(bq.y, 0u, __builtin_amdgcn_sad_u8(bq.x, 0u, bot));
            }
            s_c8[m][half] = top | (bot << 16);
        }
    }
    __syncthreads();
    if (S.me_stop == 2) return;
    /* ---- P2: vertical 8-sums, stored as vertical pairs V(x,y) = S8(x,y) | S8(x,y+8) << 16 so that the bound of a
     * candidate needs two dword reads (left and right 8x8 columns) and no repacking.  6 segments of 9 rows x 78 column
     * pairs = 468 busy threads (the first layout, 7 x 8 rows x 39 groups of four columns, kept 273 threads busy and
     * three waves idle); a thread slides down 17 rows of block sums of its two columns.  The pairs overwrite the
     * horizontal sums: every thread holds its sums in registers before the barrier, the stores come after it. */
    {
        const bool act = tid < 6 * 2 * T2_HG;
        const int seg = act ? tid / (2 * T2_HG) : 0, q2 = act ? tid - seg * 2 * T2_HG : 0;
        const int r0 = seg * 9;
        uint32_t sr[17]; /* S8 of rows r0 .. r0 + 16, two columns as packed u16 */
        if (act) {
            const uint32_t *col = (const uint32_t *)s_h8 + r0 * 2 * T2_HG + q2;
            uint32_t acc = col[0];
#pragma unroll
            for (int k = 1; k < 8; k++) acc += col[k * 2 * T2_HG];
            sr[0] = acc;
#pragma unroll
            for (int k = 1; k < 17; k++) {
                if (r0 + k <= T2_VR + 7) acc = acc + col[(k + 7) * 2 * T2_HG] - col[(k - 1) * 2 * T2_HG]; /* h8 rows <= 63 */
                sr[k] = acc;
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (r0 + k < T2_VR) {
                    const uint32_t lo = sr[k], hi = sr[k + 8];
                    *(uint2 *)((uint32_t *)s_v8 + ((r0 + k) * T2_HG) * 4 + q2 * 2) =
                        make_uint2((lo & 0xffffu) | (hi << 16), (lo >> 16) | (hi & 0xffff0000u));
                }
            }
        }
    }
    __syncthreads();
    if (S.me_stop == 3) return;

    const uint2 c8 = *(const uint2 *)s_c8[hw];
    const uint32_t cleft = c8.x, cright = c8.y; /* (top, bottom) sums of the left / right 8x8 columns */

    const uint32_t *v8 = (const uint32_t *)s_v8;
    const uint32_t prev = s_prev[hw];
    /* ---- P3: two start points per macroblock in one quarter-wave pass: last frame's vector | the zero vector.  Their
     * smaller SAD is the threshold of the bound test (any real SAD is a valid threshold: start points only steer the
     * pruning, the result is the exact minimum either way). ---- */
    const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
    const int ci_p = pcy * 33 + pcx, ci_z = 16 * 33 + 16;
    uint32_t thr;
    {
        const bool second = (lane & 16) != 0;
        const uint32_t rank_p = rank[ci_p], rank_z = rank[ci_z]; /* issued before the SAD, used after it */
        const uint32_t a = me3_quarter_sad(s_win, s_cur, col0, row0, lane, second ? 16 : pcx, second ? 16 : pcy);
        /* quarter totals sit in lanes 15 / 31 / 47 / 63: quarters 0, 1 belong to the wave's first macroblock (last
         * frame's vector, zero vector), quarters 2, 3 to its second -- no cross-lane exchange beyond the readlanes */
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), q1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 31);
        const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47), q3 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
        const uint32_t sad_p = lane < 32 ? q0 : q2, sad_z = lane < 32 ? q1 : q3;
        thr = min(sad_p, sad_z);
        if (l32 == 0) {
            const unsigned long long kp = ((unsigned long long)((sad_p << 11) | rank_p) << 11) | (uint32_t)ci_p;
            const unsigned long long kz = ((unsigned long long)((sad_z << 11) | rank_z) << 11) | (uint32_t)ci_z;
            s_best[hw] = kp < kz ? kp : kz;
        }
    }
    if (S.me_stop == 4) { /* diagnostic: start points done, bounds not started */
        if (l32 == 0 && mb_valid) B.mvsad[mbidx] = thr << 16;
        return;
    }
    /* ---- P4a: lower bounds LB = |V_left - (TL,BL)| + |V_right - (TR,BR)| (two v_sad_u16 on packed u16 pairs; lane =
     * window column cx = l32, one value per vertical displacement) and the survivors: bound <= threshold, appended to
     * the tile's pool.  Every wave owns a segment of the pool (2 macroblocks x T3_CAP entries) and compacts into it with
     * wave ballots: no atomics, no pool-full case.
     *   fast pass: bounds are tested as they are computed, four rows at a time, against the start points' threshold --
     *     nothing is kept, most groups hold no survivor;
     *   second pass, only in a wave one of whose macroblocks kept more than T3_RETRY survivors (the start points were
     *     poor: new content, the first P picture of a GOP): the bounds are computed again and kept, the candidate with
     *     the SMALLEST bound is evaluated as a third start point, and the segment is rebuilt against the tighter
     *     threshold.
     * A macroblock with more than T3_CAP survivors after that is listed for the exhaustive search (k_me_fix). ---- */
    uint32_t mycnt = 0; /* survivors of this lane's macroblock besides the start points */
    {
        const uint32_t tag = (uint32_t)hw << 11;
        const uint32_t lt = (1u << l32) - 1u;
        uint32_t *seg = s_pool + (tid >> 6) * (2 * T3_CAP);
        uint32_t wcnt = 0; /* entries in this wave's segment (wave-uniform) */
        auto append = [&](bool keep, uint32_t entry) { /* wave-uniform call sites only */
            const unsigned long long kb = __ballot(keep);
            if (kb == 0ull) return;
            const uint32_t hm = (uint32_t)(kb >> hbase);
            const bool put = keep && mycnt + (uint32_t)__popc(hm & lt) < T3_CAP;
            const unsigned long long pb = __ballot(put);
            if (put) seg[wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(pb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pb, 0u))] = entry;
            wcnt += (uint32_t)__popcll(pb);
            mycnt += (uint32_t)__popc(hm);
        };
        const uint32_t *colp = v8 + row0 * T2_SU + col0 + l32;
        uint32_t lbx0, lbx1 = 0xffffffffu; /* column cx = 32: lane l32 owns cy = l32, lane 0 also cy = 32 */
        {
            const uint32_t *colx = v8 + row0 * T2_SU + col0 + 32;
            lbx0 = __builtin_amdgcn_sad_u16(colx[l32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colx[l32 * T2_SU], cleft, 0u));
            if (l32 == 0) lbx1 = __builtin_amdgcn_sad_u16(colx[32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colx[32 * T2_SU], cleft, 0u));
        }
        bool again = S.me_fast == 0;
        if (S.me_fast) {
            const int thrs = mb_valid ? (int)thr : -1; /* (bounds < 2^17: the signed compare never passes -1) */
#pragma unroll
            for (int g0 = 0; g0 < 33; g0 += 4) {
                uint32_t lb[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    lb[j] = g0 + j < 33 ? __builtin_amdgcn_sad_u16(colp[(g0 + j) * T2_SU + 8], cright,
                                                                   __builtin_amdgcn_sad_u16(colp[(g0 + j) * T2_SU], cleft, 0u))
                                        : 0x7fffffffu;
                const int gmin = (int)min(min(lb[0], lb[1]), min(lb[2], lb[3]));
                if (__ballot(gmin <= thrs) == 0ull) continue;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (g0 + j >= 33) break;
                    const bool hit = (int)lb[j] <= thrs;
                    if (__ballot(hit) != 0ull) {
                        const uint32_t ci = (uint32_t)((g0 + j) * 33) + (uint32_t)l32;
                        append(hit && ci != (uint32_t)ci_p && ci != (uint32_t)ci_z, (lb[j] << 15) | tag | ci);
                    }
                }
            }
            {   /* column cx = 32 */
                const bool hit0 = (int)lbx0 <= thrs, hit1 = l32 == 0 && (int)lbx1 <= thrs;
                if (__ballot(hit0 || hit1) != 0ull) {
                    const uint32_t ci0 = (uint32_t)(l32 * 33 + 32), ci1 = (uint32_t)(32 * 33 + 32);
                    append(hit0 && ci0 != (uint32_t)ci_p && ci0 != (uint32_t)ci_z, (lbx0 << 15) | tag | ci0);
                    append(hit1 && ci1 != (uint32_t)ci_p && ci1 != (uint32_t)ci_z, (lbx1 << 15) | tag | ci1);
                }
            }
            again = __ballot(mycnt > T3_RETRY) != 0ull;
        }
        if (again) {
            /* packed bounds pk = LB << 6 | cy, one register per vertical displacement */
            uint32_t pk[33];
            /* batches of 6 rows: at most 12 sums in flight (one batch of 33 would hold 66 registers and the sixth wave) */
#pragma unroll
            for (int c0 = 0; c0 < 33; c0 += 6) {
#pragma unroll
                for (int cy = c0; cy < (c0 + 6 < 33 ? c0 + 6 : 33); cy++) {
                    const uint32_t lb =
                        __builtin_amdgcn_sad_u16(colp[cy * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[cy * T2_SU], cleft, 0u));
                    pk[cy] = (lb << 6) | (uint32_t)cy;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            /* candidate with the smallest bound (any of them on ties) */
            int ai, acx, acy;
            {
                uint32_t m = pk[0];
#pragma unroll
                for (int cy = 1; cy < 33; cy += 2) m = min(min(m, pk[cy]), pk[cy + 1]);
                uint32_t lmin = ((m >> 6) << 12) | ((m & 63u) << 6) | (uint32_t)l32; /* LB << 12 | cy << 6 | cx */
                lmin = min(lmin, (lbx0 << 12) | ((uint32_t)l32 << 6) | 32u);
                if (l32 == 0) lmin = min(lmin, (lbx1 << 12) | (32u << 6) | 32u);
                const uint32_t w = half_min32(lmin, lane);
                acy = (int)((w >> 6) & 63u), acx = (int)(w & 63u);
                ai = acy * 33 + acx;
            }
            {
                const uint32_t rank_a = rank[ai];
                const uint32_t a = me3_quarter_sad(s_win, s_cur, col0, row0, lane, acx, acy); /* both quarters of a half: the same candidate */
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47);
                const uint32_t sad_a = lane < 32 ? q0 : q2;
                thr = min(thr, sad_a);
                if (l32 == 0) {
                    const unsigned long long ka = ((unsigned long long)((sad_a << 11) | rank_a) << 11) | (uint32_t)ai;
                    const unsigned long long cur_best = s_best[hw];
                    s_best[hw] = ka < cur_best ? ka : cur_best;
                }
            }
            wcnt = 0, mycnt = 0; /* the segment is rebuilt */
            const int thrk = mb_valid ? (int)((thr << 6) | 63u) : -1; /* (pk < 2^23: the signed compare never passes -1) */
#pragma unroll
            for (int g0 = 0; g0 < 33; g0 += 4) {
                int gmin = (int)pk[g0];
#pragma unroll
                for (int j = 1; j < 4; j++)
                    if (g0 + j < 33) gmin = min(gmin, (int)pk[g0 + j]);
                if (__ballot(gmin <= thrk) == 0ull) continue;
#pragma unroll
                for (int cy = g0; cy < (g0 + 4 < 33 ? g0 + 4 : 33); cy++) {
                    const bool hit = (int)pk[cy] <= thrk;
                    if (__ballot(hit) != 0ull) {
                        /* the rare path: the empty asm keeps its arithmetic inside the branch (hoisted, the 33 copies of
                         * it cost 60+ registers) */
                        uint32_t pv = pk[cy];
                        asm volatile("" : "+v"(pv));
                        const uint32_t ci = (pv & 63u) * 33u + (uint32_t)l32;
                        append(hit && ci != (uint32_t)ci_p && ci != (uint32_t)ci_z && ci != (uint32_t)ai, ((pv >> 6) << 15) | tag | ci);
                    }
                }
            }
            {   /* column cx = 32 */
                const bool hit0 = mb_valid && lbx0 <= thr, hit1 = mb_valid && lbx1 <= thr;
                if (__ballot(hit0 || hit1) != 0ull) {
                    const uint32_t ci0 = (uint32_t)(l32 * 33 + 32), ci1 = (uint32_t)(32 * 33 + 32);
                    append(hit0 && ci0 != (uint32_t)ci_p && ci0 != (uint32_t)ci_z && ci0 != (uint32_t)ai, (lbx0 << 15) | tag | ci0);
                    append(hit1 && ci1 != (uint32_t)ci_p && ci1 != (uint32_t)ci_z && ci1 != (uint32_t)ai, (lbx1 << 15) | tag | ci1);
                }
            }
        }
        if (l32 == 0) s_ovf[hw] = mycnt; /* > T3_CAP: the macroblock is left to k_me_fix */
        if (lane == 0) s_wcnt[tid >> 6] = wcnt;
    }
    __syncthreads();
    if (S.me_stop == 5) {
        if (l32 == 0 && mb_valid) B.mvsad[mbidx] = (uint32_t)(s_best[hw] >> 22) << 16; /* zero vector: nothing downstream reads out of range */
        return;
    }
    /* ---- P4b: the tile's survivors, round-robin over the block's 32 quarter-waves; survivor k of the tile is entry
     * k - (entries of the earlier segments) of the segment it falls into ---- */
    {
        const uint4 wa = *(const uint4 *)&s_wcnt[0], wb = *(const uint4 *)&s_wcnt[4];
        const uint32_t wc[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        uint32_t pre[9]; /* pre[j] = entries of the segments before j */
        pre[0] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) pre[j + 1] = pre[j] + wc[j];
        const uint32_t total = pre[8];
        for (uint32_t k = (uint32_t)(tid >> 4); k < ((total + 31u) & ~31u); k += 32u) {
            uint32_t start = 0, sbase = 0; /* branch-free: the last segment whose first entry is <= k */
#pragma unroll
            for (int j = 1; j < 8; j++) {
                const bool past = k >= pre[j];
                start = past ? pre[j] : start;
                sbase = past ? (uint32_t)(j * 2 * T3_CAP) : sbase;
            }
            const bool found = k < total;
            const uint32_t e = found ? s_pool[sbase + (k - start)] : 0xffffffffu;
            const int ehw = (int)((e >> 11) & 15u), ci = (int)(e & 2047u);
            bool need = found;
            if (need) need = s_ovf[ehw] <= (uint32_t)T3_CAP && (e >> 15) <= (uint32_t)(s_best[ehw] >> 22);
            if (__ballot(need) == 0ull) continue;
            const int cy = need ? ci / 33 : 16, cx = need ? ci - (ci / 33) * 33 : 16;
            const int ecol = need ? (ehw & (T2_MBX - 1)) * 16 : 0, erow = need ? (ehw >> 3) * 16 : 0;
            const uint32_t sad = me3_quarter_sad(s_win, s_cur, ecol, erow, lane, cx, cy);
            if (need && (lane & 15) == 15)
                atomicMin(&s_best[ehw], ((unsigned long long)((sad << 11) | rank[cy * 33 + cx]) << 11) | (uint32_t)ci);
        }
    }
    __syncthreads();
    /* the tile's results go out from ONE wave (wave 6, which has nothing to do in P5; lane = macroblock): vectors, the list
     * for k_me_fix, the survivor counts.  Every wave used to store its own two macroblocks' -- ~30 instructions per wave for
     * two busy lanes. */
    if ((threadIdx.x >> 6) == 6) {
        const int m = (int)(threadIdx.x & 63u);
        const int mx = tx * T2_MBX + (m & (T2_MBX - 1)), my = S.row0 + ty * T2_MBY + ((m >> 3) & (T2_MBY - 1));
        if (m < T2_MBX * T2_MBY && mx < g.mbw && my < S.row1) {
            const size_t mbidx = (size_t)item * g.nmb + (size_t)my * g.mbw + mx;
            const unsigned long long best = s_best[m];
            const uint32_t cnt = s_ovf[m];
            const uint32_t ci = (uint32_t)best & 2047u, bcy = ci / 33u;
            const int dx = (int)(ci - bcy * 33u) - 16, dy = (int)bcy - 16;
            B.mvsad[mbidx] = ((uint32_t)(best >> 22) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
            const bool overflow = cnt > T3_CAP;
            B.me_flag[mbidx] = overflow ? 1 : 0; /* k_me_fix searches it exhaustively */
            if (overflow) B.fix_list[(size_t)item * g.nmb + atomicAdd(&B.fix_count[item], 1u)] = (uint32_t)(my * g.mbw + mx);
            B.me_cnt[mbidx] = (uint8_t)min(cnt, 255u); /* introspection: survivors of the bound test besides the start points */
        }
    }
    if (S.me_stop == 6) return; /* diagnostic: the search is complete, the transform not started */
    /* ---- P5: the luma transform of the tile's macroblocks, straight from the LDS copies the search used: current samples
     * from s_cur, prediction from the window at the winning displacement -> residual, 4x4 transform, quantisation,
     * dequantisation, inverse transform, reconstruction, levels (venc_inter2.hip luma_block_code).  16 macroblocks x 16
     * blocks = the lanes of waves 0-3; macroblocks listed for k_me_fix are left to it.  The transform no longer re-reads
     * the two luma planes from memory and costs the main chain no launch of its own. ---- */
    /* the pictures' quantiser constants: fixed, or those of the QP chosen from the previous picture's SAD sum
     * (spec/CODEC.md section 1a; one table load -- the division of the rule is done once per picture, in k_me_fix) */
    venc_qparams ql5 = g.ql;
    if (RC) {
        ql5 = qparams_uniform((const venc_qparams *)s_q); /* (uniform reads: the constants land in scalar registers) */
        if (threadIdx.x >= 448) { /* wave 7 has nothing to do in P5: this tile's share of the picture's SAD sum, one atomic per block */
            const int l7 = threadIdx.x - 448;
            uint32_t sad = 0;
            if (l7 < T2_MBX * T2_MBY) {
                const int mx7 = tx * T2_MBX + (l7 & (T2_MBX - 1)), my7 = S.row0 + ty * T2_MBY + (l7 >> 3);
                if (mx7 < g.mbw && my7 < S.row1) sad = (uint32_t)(s_best[l7] >> 22);
            }
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(sad), 63);
            if (l7 == 0 && tot) atomicAdd(&B.sad_cur[item], (unsigned long long)tot);
        }
    }
    if (fuse_luma && threadIdx.x < 256) {
        const int t5 = threadIdx.x, m = t5 >> 4, blk = t5 & 15;
        const int mcol0 = (m & (T2_MBX - 1)) * 16, mrow0 = (m >> 3) * 16;
        const int mx5 = tx * T2_MBX + (m & (T2_MBX - 1)), my5 = S.row0 + ty * T2_MBY + (m >> 3);
        const bool ok = mx5 < g.mbw && my5 < S.row1 && s_ovf[m] <= (uint32_t)T3_CAP;
        const uint32_t ci = (uint32_t)s_best[m] & 2047u, bcy = ci / 33u, bcx = ci - bcy * 33u;
        const int bx = blk_x(blk) * 4, by = blk_y(blk) * 4;
        const int boff = mcol0 + (int)bcx + bx;
        const uint32_t sh = (uint32_t)(boff & 3);
        const uint32_t *wp = s_win + (mrow0 + (int)bcy + by) * T2_WD + (boff >> 2);
        const uint32_t *cp = s_cur + (mrow0 + by) * T3_CW + ((mcol0 + bx) >> 2);
        uint32_t cq[4], pq[4];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            cq[y] = cp[y * T3_CW];
            pq[y] = __builtin_amdgcn_alignbyte(wp[y * T2_WD + 1], wp[y * T2_WD], sh);
        }
        if (fuse_luma == 3) {
            /* the shipped path: decide only -- blocks without a level are finished here, the few with one are listed in this
             * (tile, wave)'s segment for the extra blocks of the k_me_fix launch (venc_inter2.hip luma_block_defer) */
            const size_t region = (size_t)item * B.nz_item_regions + (size_t)tile * 4 + (size_t)(t5 >> 6);
            luma_block_defer(g, B, S, item, t5 & 63, min(my5, g.mbh - 1) * g.mbw + min(mx5, g.mbw - 1), ok, cq, pq, ql5, B.nz_ids + region * 64, B.nz_rows + region * 128, B.nz_cnt + region, (bcx << 6) | (bcy << 12));
        } else
            luma_block_code(g, B, S, item, t5 & 63, min(my5, g.mbh - 1) * g.mbw + min(mx5, g.mbw - 1), ok, cq, pq, ql5);
    }
}
