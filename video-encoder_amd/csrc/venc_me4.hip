/*
 * venc_me4.hip -- k_me_pool4: k_me_pool (venc_me3.hip) re-laid for FOUR blocks per CU.
 *
 * The search is the one of venc_me3.hip, phase by phase and bit for bit: exact +-16 full search by successive elimination on
 * 8x8 block sums built inside the kernel, survivors pooled per tile, running best as a 64-bit LDS minimum, the tile's luma
 * decided at the end (P5, deferred form: venc_inter2.hip luma_block_defer).  What it was bound by is the lifetime of a block --
 * seven barriers, the memory latency of P0, a serial tail -- with three blocks (24 waves) on a CU: at two blocks per CU a launch
 * took 100 us, at three 84, i.e. T = 52 + 768 / waves per CU.  A fourth block needs <= 40 KB of LDS and <= 64 registers:
 *
 *  1. THE VERTICAL PAIRS ARE STORED ONCE.  venc_me3 keeps V(x, y) = S8(x, y) | S8(x, y + 8) << 16 for every row y = 0 .. 48, so
 *     that a candidate's two sums arrive as one dword -- and every sum sits in LDS twice (as the low half of V(y) and the high
 *     half of V(y - 8)): 30.6 KB.  Here only the rows with (y & 8) == 0 are kept: slot (y >> 4) * 8 + (y & 7) holds
 *     S8(y) | S8(y + 8) << 16, 32 slots = 20.0 KB.  A candidate row with (y & 8) != 0 takes its pair from two slots with one
 *     v_alignbit (high half of slot s, low half of slot s + 8).  Because a macroblock's rows are 16-aligned in the window,
 *     which case a displacement falls in is known at compile time, and the four displacements j, j + 8, j + 16, j + 24 share
 *     three slots: six dword reads for four candidate rows (venc_me3: eight), two extra vector instructions for two of them.
 *  2. P2 has less to do: four segments of sixteen rows x 78 column pairs, eight pair stores each (venc_me3: six segments of
 *     seventeen rows with nine stores).
 *  3. NO RANK TABLE (2.2 KB of LDS in venc_me3; read from memory here until round 4): the running best is a 32-bit key that
 *     carries the search's tie-break order itself (me4_key below).
 *  4. THE SECOND PASS KEEPS NO BOUNDS.  A wave whose start points were poor used to keep the 33 packed bounds of every lane
 *     (33 registers) to find the smallest one and to re-test them; here it finds the smallest in one pass of running minima
 *     and then simply runs the fast pass again against the tighter threshold -- the rare path pays twice for its bounds, the
 *     kernel holds 64 registers.
 *  5. P5 only in its deferred form (blocks with a level are listed for the k_me_fix launch), which is also what lets the
 *     transform live in 64 registers.
 *
 * 39.8 KB of LDS, 64 VGPRs: four blocks = 32 waves per CU.
 */

#define T4_VS 32 /* slots of vertical pairs: slot (y >> 4) * 8 + (y & 7) = S8(y) | S8(y + 8) << 16 for the rows with (y & 8) == 0 */

DEVFN uint32_t me4_bound(uint32_t l, uint32_t r, uint32_t cleft, uint32_t cright) {
    return __builtin_amdgcn_sad_u16(r, cright, __builtin_amdgcn_sad_u16(l, cleft, 0u));
}
DEVFN uint32_t me4_mid(uint32_t next, uint32_t cur) { /* high half of cur | low half of next << 16 */
    return __builtin_amdgcn_alignbit(next, cur, 16u);
}
/* THE RUNNING BEST IS ONE DWORD.  The search's order is (SAD, |dx| + |dy|, dy, dx) -- the rank table (venc_hip.hip) is the sort
 * index of its last three terms.  Given |dx| + |dy| and dy, dx is known up to its sign: SAD << 13 | (|dx| + |dy|) << 7 | cy << 1 |
 * (dx > 0) orders the candidates exactly as (SAD << 11 | rank) does, is unique, holds the displacement itself and fits 29 bits
 * (SAD <= 255 * 256 < 2^16): no rank load in front of a key (the one per evaluated survivor sat exposed at the tail of every
 * block), a 32-bit ds_min instead of a 64-bit one, no 64-bit shifts. */
DEVFN uint32_t me4_key(uint32_t sad, uint32_t cx, uint32_t cy) {
    const uint32_t l1 = __builtin_amdgcn_sad_u8(cx | (cy << 8), 0x1010u, 0u);
    return (sad << 13) | (l1 << 7) | (cy << 1) | (cx > 16u ? 1u : 0u);
}
DEVFN void me4_key_disp(uint32_t key, uint32_t &cx, uint32_t &cy) {
    cy = (key >> 1) & 63u;
    const uint32_t l1 = (key >> 7) & 63u;
    const uint32_t a = l1 - __builtin_amdgcn_sad_u8(cy, 16u, 0u);
    cx = (key & 1u) ? 16u + a : 16u - a;
}

template <bool RC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_me_pool4(venc_geom g, venc_bufs B, venc_step S, int tiles_x, int tiles_y, uint32_t tx_magic, int xcd_runs, int fuse_luma) {
    __shared__ __attribute__((aligned(16))) uint32_t s_win[T2_WR * T2_WD];
    __shared__ __attribute__((aligned(16))) uint32_t s_cur[T2_MBY * 16 * T3_CW];
    /* horizontal sums (P1 -> P2), then, after a barrier, the vertical pairs built from them (P2 -> P4a): the same 19968 bytes */
    __shared__ __attribute__((aligned(16))) uint4 s_hv[T4_VS * T2_HG];
    __shared__ uint32_t s_pool[T3_POOL];
    __shared__ uint32_t s_best[T2_MBX * T2_MBY]; /* me4_key of the best candidate so far */
    __shared__ uint32_t s_ovf[T2_MBX * T2_MBY];
    __shared__ __attribute__((aligned(16))) uint32_t s_wcnt[8];
    __shared__ uint32_t s_prev[T2_MBX * T2_MBY];
    __shared__ int s_q[sizeof(venc_qparams) / sizeof(int)]; /* rate control: the picture's quantiser constants, fetched in P0 for P5 */
    __shared__ __attribute__((aligned(8))) uint32_t s_c8[T2_MBX * T2_MBY][2];
    uint2 *const s_h8 = (uint2 *)s_hv;
    static_assert(sizeof(uint2) * T2_WR * T2_HG == sizeof(uint4) * T4_VS * T2_HG, "h8 and the pair slots share one buffer");
    static_assert(T2_MBY == 2 && T2_WR == 64, "slot arithmetic below: window rows 0 .. 63, macroblock rows at 0 and 16");

    const int tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, hbase = lane & 32, hw = tid >> 5;
    const int mbx = hw & (T2_MBX - 1), mby = hw >> 3;
    const int col0 = mbx * 16, row0 = mby * 16;
    const int item = (int)blockIdx.z + S.item0;
    const int kx = (int)(blockIdx.x >> 3);
    const int tile_l = xcd_runs ? ((((kx >> 6) << 3) + (int)(blockIdx.x & 7u)) << 6) + (kx & 63) : (int)blockIdx.x;
    if (tile_l >= S.tile_n) return;
    const int tile = tile_l < S.tile_split ? tile_l + S.tile0 : tile_l - S.tile_split + S.tile1; /* (one run of tiles, or the two edge rows of a band: venc_hip.hip) */
    if (tile == 0 && threadIdx.x == 0) {
        B.fix_count_next[item] = 0;
        if (RC) {
            if (!S.rc_history) B.qp_cur[item] = S.qp_base;
            B.sad_next[item] = 0ull;
        }
    }
    const int ty = tx_magic ? (int)__umulhi((uint32_t)tile, tx_magic) : tile, tx = tile - ty * tiles_x;
    const int x0 = tx * T2_MBX * 16, y0 = (S.row0 + ty * T2_MBY) * 16;
    const int mx = tx * T2_MBX + mbx, my = S.row0 + ty * T2_MBY + mby;
    const bool mb_valid = mx < g.mbw && my < S.row1;
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);

    /* ---- P0: window, current tile, last frame's vectors ---- */
    {
        const uint32_t prev_early = B.mvsad_prev[(size_t)item * g.nmb + (size_t)min(my, g.mbh - 1) * g.mbw + min(mx, g.mbw - 1)];
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
        const int ra = tid / 10, qa = tid - ra * 10;
        uint32_t *da = s_win + ra * T2_WD + 4 * qa;
        const bool interior = x0 >= 16 && x0 + 144 <= g.cw && y0 >= 16 && y0 + 48 <= g.ch; /* block-uniform */
        if (interior) {
            const uint8_t *w0 = ref + (size_t)(y0 - 16) * rstride + (x0 - 16);
            const uint4 va = *(const uint4 *)(w0 + ra * rstride + 16 * qa);
            uint4 vb = make_uint4(0u, 0u, 0u, 0u);
            const bool second = tid + 512 < T2_WR * 10;
            const int i1 = tid + 512, rb = i1 / 10, qb = i1 - rb * 10;
            if (second) vb = *(const uint4 *)(w0 + rb * rstride + 16 * qb);
            *(uint2 *)da = make_uint2(va.x, va.y);
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
    /* ---- P1: horizontal 8-sums (as venc_me3) ---- */
    {
        /* lane = window row, wave = five of the forty column groups: a half-wave's dword reads fall on sixteen banks twice over (rows 42
         * dwords apart) and its 8-byte stores on all of them; with eight threads along a row and the rows across the wave the reads were
         * four to a bank (60 % of this phase's LDS cycles were conflicts) */
        const int r = tid & 63, q0 = (tid >> 6) * 5;
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
    }
    __syncthreads();
    if (S.me_stop == 2) return;
    /* ---- P2: vertical 8-sums into the pair slots.  Thread = (segment of sixteen rows, two columns): it slides down the
     * sixteen block sums S8(16 seg + k), k = 0 .. 15, of its columns (packed u16 pair) and stores the eight pairs
     * S8(16 seg + j) | S8(16 seg + 8 + j) << 16 of each column into slots 8 seg + j.  The last segment has window rows for
     * k <= 8 only: its slots hold rows 48 .. 55 in their low halves (the row 56 beside row 48); the other high halves are
     * never read.  The slots overwrite the horizontal sums: all reads, a barrier, all writes. ---- */
    {
        const bool act = tid < 4 * 2 * T2_HG;
        const int seg = act ? tid / (2 * T2_HG) : 0, q2 = act ? tid - seg * 2 * T2_HG : 0;
        const int r0 = seg * 16;
        const int last = T2_WR - 1 - r0; /* h8 rows r0 + k exist for k <= last */
        uint32_t sr[16];
        if (act) {
            const uint32_t *col = (const uint32_t *)s_h8 + r0 * 2 * T2_HG + q2;
            uint32_t acc = col[0];
#pragma unroll
            for (int k = 1; k < 8; k++) acc += col[k * 2 * T2_HG];
            sr[0] = acc;
#pragma unroll
            for (int k = 1; k < 16; k++) {
                if (k + 7 <= last) acc = acc + col[(k + 7) * 2 * T2_HG] - col[(k - 1) * 2 * T2_HG];
                sr[k] = acc;
            }
        } else if (tid >= 512 - 2 * T2_MBX * T2_MBY) {
            /* the four 8x8 sums of every current macroblock, by half of the one wave this phase leaves idle (in P1, on wave 0, they
             * doubled that wave's instruction stream and the block waited for it at the barrier) */
            const int t = tid - (512 - 2 * T2_MBX * T2_MBY), m = t >> 1, half = t & 1;
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
        __syncthreads();
        if (act) {
            uint32_t *dst = (uint32_t *)s_hv + (seg * 8) * T2_SU + q2 * 2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t lo = sr[j], hi = sr[j + 8];
                *(uint2 *)(dst + j * T2_SU) = make_uint2(__builtin_amdgcn_perm(hi, lo, 0x05040100u), __builtin_amdgcn_perm(hi, lo, 0x07060302u)); /* lo.lo | hi.lo << 16, lo.hi | hi.hi << 16: one v_perm each */
            }
        }
    }
    __syncthreads();
    if (S.me_stop == 3) return;

    const uint2 c8 = *(const uint2 *)s_c8[hw];
    const uint32_t cleft = c8.x, cright = c8.y;
    const uint32_t *v8 = (const uint32_t *)s_hv;
    const uint32_t prev = s_prev[hw];
    /* ---- P3: two start points per macroblock (last frame's vector | the zero vector), as venc_me3 ---- */
    const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
    const int ci_p = (pcy << 6) | pcx, ci_z = (16 << 6) | 16; /* candidate = cy << 6 | cx in the pool's entries */
    uint32_t thr;
    {
        const bool second = (lane & 16) != 0;
        const uint32_t a = me3_quarter_sad(s_win, s_cur, col0, row0, lane, second ? 16 : pcx, second ? 16 : pcy);
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), q1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 31);
        const uint32_t q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47), q3 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
        const uint32_t sad_p = lane < 32 ? q0 : q2, sad_z = lane < 32 ? q1 : q3;
        thr = min(sad_p, sad_z);
        if (l32 == 0) s_best[hw] = min(me4_key(sad_p, (uint32_t)pcx, (uint32_t)pcy), (sad_z << 13) | (16u << 1));
    }
    if (S.me_stop == 4) {
        if (l32 == 0 && mb_valid) B.mvsad[(size_t)item * g.nmb + (size_t)my * g.mbw + mx] = thr << 16;
        return;
    }
    /* ---- P4a: lower bounds and survivors.  Lane = window column cx = l32; the displacements j, j + 8, j + 16, j + 24 (and 32
     * beside j = 0) of a macroblock share the three pair slots mby * 8 + j, + 8, + 16: direct pairs for j and j + 16, the
     * high half of one slot with the low half of the next for j + 8 and j + 24.  Tested as they are computed against the
     * start points' threshold, a group per wave-uniform branch; survivors are compacted with wave ballots into the wave's
     * own segment of the tile's pool.  A wave one of whose macroblocks keeps more than T3_RETRY survivors finds the
     * candidate with the smallest bound, evaluates it as a third start point and runs the pass again. ---- */
    uint32_t mycnt = 0;
    {
        const uint32_t tag = (uint32_t)hw << 12; /* entry = LB << 16 | macroblock << 12 | cy << 6 | cx (LB <= 4 * 64 * 255 < 2^16) */
        const uint32_t lt = (1u << l32) - 1u;
        uint32_t *seg = s_pool + (tid >> 6) * (2 * T3_CAP);
        uint32_t wcnt = 0;
        int ai = -1; /* the third start point of a second pass */
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
        const uint32_t *colp0 = v8 + (mby * 8) * T2_SU + col0 + l32;
        /* column cx = 32: lane l32 owns cy = l32, lane 0 also cy = 32 */
        uint32_t lbx0, lbx1 = 0x7fffffffu;
        {
            const uint32_t *colx = v8 + ((mby + (l32 >> 4)) * 8 + (l32 & 7)) * T2_SU + col0 + 32;
            const uint32_t l0 = colx[0], r0 = colx[8], l1 = colx[8 * T2_SU], r1 = colx[8 * T2_SU + 8];
            const bool mid = (l32 & 8) != 0;
            lbx0 = me4_bound(mid ? me4_mid(l1, l0) : l0, mid ? me4_mid(r1, r0) : r0, cleft, cright);
            if (l32 == 0) lbx1 = me4_bound(colx[16 * T2_SU], colx[16 * T2_SU + 8], cleft, cright);
        }
        bool first_pass = true;
#pragma unroll 1
        for (;;) {
            const int thrs = mb_valid ? (int)thr : -1; /* (bounds < 2^17: the signed compare never passes -1) */
            /* (an opaque copy of the column pointer per pass: seen as loop-invariant, the 33 bounds and their pool entries are hoisted
             * out of this loop and kept -- the 33 registers the second pass was rewritten to avoid) */
            uint32_t opq = 0; /* (an offset, not the pointer: through the asm a pointer loses its LDS address space and the reads turn flat) */
            asm volatile("" : "+v"(opq));
            const uint32_t *colp = colp0 + opq;
            uint32_t nx[6] = {colp[0], colp[8], colp[8 * T2_SU], colp[8 * T2_SU + 8], colp[16 * T2_SU], colp[16 * T2_SU + 8]};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                /* the next group's six dwords are on their way while this group is tested (a group's reads used to be issued behind the
                 * previous group's branch: every group began with an LDS round trip) */
                const uint32_t l0 = nx[0], r0 = nx[1], l1 = nx[2], r1 = nx[3], l2 = nx[4], r2 = nx[5];
                if (j < 7) {
                    nx[0] = colp[(j + 1) * T2_SU], nx[1] = colp[(j + 1) * T2_SU + 8];
                    nx[2] = colp[(j + 9) * T2_SU], nx[3] = colp[(j + 9) * T2_SU + 8];
                    nx[4] = colp[(j + 17) * T2_SU], nx[5] = colp[(j + 17) * T2_SU + 8];
                }
                uint32_t lb[5];
                lb[0] = me4_bound(l0, r0, cleft, cright);                           /* cy = j */
                lb[1] = me4_bound(me4_mid(l1, l0), me4_mid(r1, r0), cleft, cright); /* cy = j + 8 */
                lb[2] = me4_bound(l1, r1, cleft, cright);                           /* cy = j + 16 */
                lb[3] = me4_bound(me4_mid(l2, l1), me4_mid(r2, r1), cleft, cright); /* cy = j + 24 */
                lb[4] = j == 0 ? me4_bound(l2, r2, cleft, cright) : 0x7fffffffu;    /* cy = 32 */
                int gmin = (int)min(min(lb[0], lb[1]), min(lb[2], lb[3]));
                if (j == 0) gmin = min(gmin, (int)lb[4]);
                if (__ballot(gmin <= thrs) == 0ull) continue;
#pragma unroll
                for (int t = 0; t < (j == 0 ? 5 : 4); t++) {
                    const bool hit = (int)lb[t] <= thrs;
                    if (__ballot(hit) != 0ull) {
                        /* the rare path: the empty asm keeps its arithmetic inside the branch (the passes form a loop: hoisted as
                         * loop invariants, the 33 candidate indices, pool entries and start-point compares cost 60+ registers) */
                        uint32_t lx = (uint32_t)l32;
                        asm volatile("" : "+v"(lx));
                        const uint32_t ci = (uint32_t)((j + 8 * t) << 6) + lx;
                        append(hit && ci != (uint32_t)ci_p && ci != (uint32_t)ci_z && ci != (uint32_t)ai, (lb[t] << 16) | tag | ci);
                    }
                }
            }
            {   /* column cx = 32 */
                const bool hit0 = (int)lbx0 <= thrs, hit1 = l32 == 0 && (int)lbx1 <= thrs;
                if (__ballot(hit0 || hit1) != 0ull) {
                    uint32_t lx = (uint32_t)l32;
                    asm volatile("" : "+v"(lx));
                    const uint32_t ci0 = (lx << 6) + 32u, ci1 = (uint32_t)((32 << 6) + 32);
                    append(hit0 && ci0 != (uint32_t)ci_p && ci0 != (uint32_t)ci_z && ci0 != (uint32_t)ai, (lbx0 << 16) | tag | ci0);
                    append(hit1 && ci1 != (uint32_t)ci_p && ci1 != (uint32_t)ci_z && ci1 != (uint32_t)ai, (lbx1 << 16) | tag | ci1);
                }
            }
            if (!first_pass) break;
            first_pass = false;
            if (S.me_fast && __ballot(mycnt > T3_RETRY) == 0ull) break;
            /* second pass (rare: new content, the first P picture of a GOP; VENC_ME_FAST=0: every wave): the candidate with the
             * smallest bound (any of them on ties) becomes a third start point */
            uint32_t m = 0xffffffffu; /* LB << 12 | cy << 6 | cx, running minimum of this lane's column */
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t l0 = colp[j * T2_SU], r0 = colp[j * T2_SU + 8];
                const uint32_t l1 = colp[(j + 8) * T2_SU], r1 = colp[(j + 8) * T2_SU + 8];
                const uint32_t l2 = colp[(j + 16) * T2_SU], r2 = colp[(j + 16) * T2_SU + 8];
                m = min(m, (me4_bound(l0, r0, cleft, cright) << 12) | (uint32_t)(j << 6));
                m = min(m, (me4_bound(me4_mid(l1, l0), me4_mid(r1, r0), cleft, cright) << 12) | (uint32_t)((j + 8) << 6));
                m = min(m, (me4_bound(l1, r1, cleft, cright) << 12) | (uint32_t)((j + 16) << 6));
                m = min(m, (me4_bound(me4_mid(l2, l1), me4_mid(r2, r1), cleft, cright) << 12) | (uint32_t)((j + 24) << 6));
                if (j == 0) m = min(m, (me4_bound(l2, r2, cleft, cright) << 12) | (uint32_t)(32 << 6));
            }
            m |= (uint32_t)l32;
            m = min(m, (lbx0 << 12) | ((uint32_t)l32 << 6) | 32u);
            if (l32 == 0) m = min(m, (lbx1 << 12) | (32u << 6) | 32u);
            const uint32_t wmin = half_min32(m, lane);
            const int acy = (int)((wmin >> 6) & 63u), acx = (int)(wmin & 63u);
            ai = (int)(wmin & 4095u);
            {
                const uint32_t a = me3_quarter_sad(s_win, s_cur, col0, row0, lane, acx, acy); /* both quarters of a half: the same candidate */
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), q2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47);
                const uint32_t sad_a = lane < 32 ? q0 : q2;
                thr = min(thr, sad_a);
                if (l32 == 0) s_best[hw] = min(s_best[hw], me4_key(sad_a, (uint32_t)acx, (uint32_t)acy));
            }
            wcnt = 0, mycnt = 0; /* the segment is rebuilt */
        }
        {   /* (indices from an opaque copy of the thread id: kept across the passes, the address of s_ovf[hw] was spilled) */
            uint32_t tidc = threadIdx.x;
            asm volatile("" : "+v"(tidc));
            if ((tidc & 31u) == 0u) s_ovf[tidc >> 5] = mycnt; /* > T3_CAP: the macroblock is left to k_me_fix */
            if ((tidc & 63u) == 0u) s_wcnt[tidc >> 6] = wcnt;
        }
    }
    __syncthreads();
    if (S.me_stop == 5) {
        if (l32 == 0 && mb_valid) B.mvsad[(size_t)item * g.nmb + (size_t)my * g.mbw + mx] = (s_best[hw] >> 13) << 16;
        return;
    }
    /* ---- P4b: the tile's survivors, round-robin over the block's 32 quarter-waves (as venc_me3) ----
     * From here on the lane indices are re-derived from an opaque copy of the thread id: carried across P4a, four of them were
     * spilled (scratch stores and reloads of every thread: 42 MB of writes per 1080p launch in the PMC counters, and a memory round
     * trip at the head of every block's tail). */
    uint32_t tidb = threadIdx.x;
    asm volatile("" : "+v"(tidb));
    const int laneb = (int)(tidb & 63u);
    {
        const uint4 wa = *(const uint4 *)&s_wcnt[0], wb = *(const uint4 *)&s_wcnt[4];
        const uint32_t wc[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        uint32_t pre[9];
        pre[0] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) pre[j + 1] = pre[j] + wc[j];
        const uint32_t total = pre[8];
        for (uint32_t k = tidb >> 4; k < ((total + 31u) & ~31u); k += 32u) {
            uint32_t start = 0, sbase = 0;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                const bool past = k >= pre[j];
                start = past ? pre[j] : start;
                sbase = past ? (uint32_t)(j * 2 * T3_CAP) : sbase;
            }
            const bool found = k < total;
            const uint32_t e = found ? s_pool[sbase + (k - start)] : 0xffffffffu;
            const int ehw = (int)((e >> 12) & 15u);
            const uint32_t eov = s_ovf[ehw], ebest = s_best[ehw]; /* (both reads in flight together; a lane without an entry reads macroblock 15's) */
            const bool need = found && eov <= (uint32_t)T3_CAP && (e >> 16) <= (ebest >> 13);
            if (__ballot(need) == 0ull) continue;
            const int cy = need ? (int)((e >> 6) & 63u) : 16, cx = need ? (int)(e & 63u) : 16;
            const int ecol = need ? (ehw & (T2_MBX - 1)) * 16 : 0, erow = need ? (ehw >> 3) * 16 : 0;
            const uint32_t sad = me3_quarter_sad(s_win, s_cur, ecol, erow, laneb, cx, cy);
            if (need && (laneb & 15) == 15) atomicMin(&s_best[ehw], me4_key(sad, (uint32_t)cx, (uint32_t)cy));
        }
    }
    __syncthreads();
    /* the tile's results leave from one wave (lane = macroblock) */
    if ((tidb >> 6) == 6) {
        const int m = laneb;
        const int mxo = tx * T2_MBX + (m & (T2_MBX - 1)), myo = S.row0 + ty * T2_MBY + ((m >> 3) & (T2_MBY - 1));
        if (m < T2_MBX * T2_MBY && mxo < g.mbw && myo < S.row1) {
            const size_t mbo = (size_t)item * g.nmb + (size_t)myo * g.mbw + mxo;
            const uint32_t best = s_best[m];
            const uint32_t cnt = s_ovf[m];
            uint32_t bcx, bcy;
            me4_key_disp(best, bcx, bcy);
            const int dx = (int)bcx - 16, dy = (int)bcy - 16;
            B.mvsad[mbo] = ((best >> 13) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
            const bool overflow = cnt > T3_CAP;
            B.me_flag[mbo] = overflow ? 1 : 0;
            if (overflow) B.fix_list[(size_t)item * g.nmb + atomicAdd(&B.fix_count[item], 1u)] = (uint32_t)(myo * g.mbw + mxo);
            B.me_cnt[mbo] = (uint8_t)min(cnt, 255u);
        }
    }
    if (S.me_stop == 6) return;
    /* ---- P5: the tile's luma, decided from the LDS copies the search used (deferred form: venc_inter2.hip luma_block_defer) ---- */
    venc_qparams ql5 = g.ql;
    if (RC) {
        ql5 = qparams_uniform((const venc_qparams *)s_q); /* (uniform reads: the constants land in scalar registers) */
        if (tidb >= 448) { /* wave 7: this tile's share of the picture's SAD sum, one atomic per block */
            const int l7 = (int)tidb - 448;
            uint32_t sad = 0;
            if (l7 < T2_MBX * T2_MBY) {
                const int mx7 = tx * T2_MBX + (l7 & (T2_MBX - 1)), my7 = S.row0 + ty * T2_MBY + (l7 >> 3);
                if (mx7 < g.mbw && my7 < S.row1) sad = s_best[l7] >> 13;
            }
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(sad), 63);
            if (l7 == 0 && tot) atomicAdd(&B.sad_cur[item], (unsigned long long)tot);
        }
    }
    if (fuse_luma && tidb < 256) {
        const int t5 = (int)tidb, m = t5 >> 4, blk = t5 & 15;
        const int mcol0 = (m & (T2_MBX - 1)) * 16, mrow0 = (m >> 3) * 16;
        const int mx5 = tx * T2_MBX + (m & (T2_MBX - 1)), my5 = S.row0 + ty * T2_MBY + (m >> 3);
        const bool ok = mx5 < g.mbw && my5 < S.row1 && s_ovf[m] <= (uint32_t)T3_CAP;
        uint32_t bcx, bcy;
        me4_key_disp(s_best[m], bcx, bcy);
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
        const size_t region = (size_t)item * B.nz_item_regions + (size_t)tile * 4 + (size_t)(t5 >> 6);
        luma_block_defer(g, B, S, item, t5 & 63, min(my5, g.mbh - 1) * g.mbw + min(mx5, g.mbw - 1), ok, cq, pq, ql5, B.nz_ids + region * 64, B.nz_rows + region * 128, B.nz_cnt + region,
                         (bcx << 6) | (bcy << 12));
    }
}
