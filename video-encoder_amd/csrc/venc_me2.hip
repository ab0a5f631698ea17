/*
 * venc_me2.hip -- successive-elimination motion search with the block sums built
 * inside the kernel (no sums plane, no extra pass over the reference).
 *
 * Why: with the sums taken from a precomputed plane (venc_me.hip) every 8-macroblock
 * block staged 34 KB (4.2 KB per macroblock, 8x the algorithmic 512 B) and the kernel ran
 * at the L2->LDS fill rate.  Here one block owns an 8 x 2 macroblock tile, stages only the
 * pixels it needs (64 x 160 reference window + 32 x 128 current tile: 0.9 KB per
 * macroblock) and derives the 8x8 block sums of every window position from the LDS copy:
 *   P1  horizontal 8-sums with v_qsad_pk_u16_u8 against zero (4 sliding 4-byte sums per
 *       instruction, two of them added), packed u16x4
 *   P2  vertical 8-sums by a sliding window (packed u16 adds)
 *   P3  lower bounds LB(c) = sum_k |S8cur_k - S8ref_k(c)| for all 1089 candidates: every
 *       lane owns one displacement column, reads each sums row once, 2 v_sad_u16 per candidate
 *   P4  SAD only where LB <= best SAD so far (start: zero vector and last frame's vector)
 * The result equals the exhaustive search bit for bit (see venc_me.hip for the argument).
 * Macroblocks with more than SEA_LIST surviving candidates (flat content) are flagged and
 * redone by k_me_fix with the exhaustive half-wave search.
 */

#define SEA_LIST 64 /* survivors a macroblock may keep; more: exhaustive search by k_me_fix */
#define T2_MBX 8
#define T2_MBY 2
#define T2_WR (T2_MBY * 16 + 32) /* window rows: 64 */
#define T2_WD 42                 /* window row stride in dwords: 160 pixels + 8 bytes of zero padding */
#define T2_HG 39                 /* groups of 4 columns with horizontal sums: columns 0..155 (0..152 used) */
#define T2_SR (T2_MBY * 16 + 25) /* rows of block sums: 57 */
#define T2_SU (T2_HG * 4)        /* sums row stride in positions: 156 */
#define T2_VR (T2_MBY * 16 + 17) /* rows of vertical pairs (cy + row0 <= 48): 49 */

/* Diagnostic build only (-DVENC_ME_STAMPS, tools/me_stamps.py): per-phase wave-cycle sums of
 * k_me_tile into B.stamps; the shipped library contains no stamp. */
#ifdef VENC_ME_STAMPS
#define ME_STAMP(i)                                                                         \
    do {                                                                                    \
        const unsigned long long t_ = clock64();                                            \
        if ((threadIdx.x & 63) == 0) atomicAdd(&B.stamps[i], t_ - t_prev_);                 \
        t_prev_ = t_;                                                                       \
    } while (0)
#define ME_STAMP_INIT unsigned long long t_prev_ = clock64()
#else
#define ME_STAMP(i) do { } while (0)
#define ME_STAMP_INIT do { } while (0)
#endif

DEVFN uint2 pk4_add(uint2 a, uint2 b) { return make_uint2(a.x + b.x, a.y + b.y); }
DEVFN uint2 pk4_sub(uint2 a, uint2 b) { return make_uint2(a.x - b.x, a.y - b.y); }

#ifdef VENC_LEGACY_KERNELS /* first-generation kernels: test-only library */
#include "legacy/me_tile.hip"
#endif

/* Exhaustive redo of the macroblocks k_me_tile flagged (bound not selective enough), one
 * macroblock per block so that the few flagged ones (around occlusions; ~0.5 % on the synthetic
 * clip) run side by side instead of dragging whole tiles along.  The four waves of a block split
 * the 33 vertical displacements (9 each, the last wave's tail is masked); lane = horizontal
 * displacement.  k_me_tile appended the macroblock indices to fix_list / fix_count. */
#ifndef FIX_BLOCKS
#define FIX_BLOCKS 256
#endif
/* (FIX_BLOCKS:  per item: one listed macroblock per block in all but pathological pictures (a block that loops costs the step a second search latency) */
#define FIX_WD 13 /* window row: 48 bytes + 4 spare, in dwords */
#define FIX_NDY 9   /* vertical displacements per wave: 4 waves cover 36 >= 33.  (8 waves x 5: k_me_fix 17 -> 26 us per launch, 1080p 149 -> 143
                       Gpx/s; 16 x 3 spills at the 128-register cap of a 1024-thread block.  Every wave pays the 64 LDS reads of the
                       current block and the block-wide barriers whatever its share of the search.  Also tried: five waves, columns -16 .. 15 as
                       half-waves of five displacements each and the column +16 on a wave of its own -- 20 window rows and 320 SADs per
                       lane instead of 24 and 576, no idle lanes: 17 -> 28 us per launch, 1080p 151 -> 143 Gpx/s.) */
#define FIX_WAVES 4

/* The follow-up of luma_block_defer (venc_inter2.hip): the listed 4x4 blocks of one item, BALANCED over the item's nz_blocks
 * blocks of 256 threads.  Listed blocks cluster (an occlusion fills its segments, most segments are empty): dealt out segment
 * by segment, the block that drew the cluster looped sixteen times while the others idled -- the launch took 33 us whatever the
 * picture size.  So every block reads ALL segment counts of its item (rpt per thread, as uint4), the block scans them, and block
 * b takes entries [b, b + 1) x ceil(total / nz_blocks) of the item's concatenated lists: thread k of a round finds its entry's
 * thread-segment by binary search over the 256 partial sums and walks that segment's counts (bytes in LDS).
 *   s_part: 256 exclusive partial sums + the total; s_wsum: 4 wave totals; s_cnt8: 256 x rpt counts as bytes (dynamic LDS) */
DEVFN void luma_nz_redo(const venc_geom &g, const venc_bufs &B, const venc_step &S, int item, int nzblock, int nz_blocks, int regions, int rpt, int ntx,
                        uint32_t tx_magic, const venc_qparams &ql, uint32_t *s_part, uint32_t *s_wsum, uint32_t *s_cnt32) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t ibase = (size_t)item * B.nz_item_regions;
    const int rpt4 = rpt >> 2;
    uint32_t sum = 0;
    for (int q0 = 0; q0 < rpt4; q0 += 8) { /* eight quads per round, all loads of a round in flight together (one round up to 2160p) */
        uint4 cc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = tid * rpt + 4 * (q0 + u); /* (regions and rpt are multiples of four: a quad is inside or outside) */
            cc[u] = make_uint4(0u, 0u, 0u, 0u);
            if (q0 + u < rpt4 && r < regions) cc[u] = *(const uint4 *)(B.nz_cnt + ibase + r);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint4 c = cc[u];
            c.x = min(c.x, 64u), c.y = min(c.y, 64u), c.z = min(c.z, 64u), c.w = min(c.w, 64u);
            sum += c.x + c.y + c.z + c.w;
            if (q0 + u < rpt4) s_cnt32[tid * rpt4 + q0 + u] = c.x | (c.y << 8) | (c.z << 16) | (c.w << 24);
        }
    }
    const uint32_t incl = wave_scan_incl(sum);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < FIX_WAVES; w++) {
        const uint32_t ws = s_wsum[w];
        woff += w < wave ? ws : 0u;
        total += ws;
    }
    s_part[tid] = woff + incl - sum;
    if (tid == 0) s_part[256] = total;
    __syncthreads();
    const uint32_t chunk = (total + (uint32_t)nz_blocks - 1u) / (uint32_t)nz_blocks;
    const uint32_t lo = (uint32_t)nzblock * chunk, hi = min(total, lo + chunk);
    for (uint32_t k = lo + (uint32_t)tid; k < hi; k += 64 * FIX_WAVES) {
        uint32_t rem;
        const int r = venc_nz_locate(s_part, s_cnt32, rpt, k, &rem); /* (venc_lines.h; tests/test_device_functions_cpu.py) */
        const size_t slot = (ibase + r) * 64 + rem;
        const uint32_t e = B.nz_ids[slot];
        const uint4 cur4 = B.nz_rows[2 * slot], pred4 = B.nz_rows[2 * slot + 1];
        const int tile = r >> 2;
        const int m = (r & 3) * 4 + (int)((e >> 4) & 3u), blk = (int)(e & 15u);
        const int ty = tx_magic ? (int)__umulhi((uint32_t)tile, tx_magic) : tile, tx = tile - ty * ntx;
        const int mx = tx * T2_MBX + (m & (T2_MBX - 1)), my = S.row0 + ty * T2_MBY + (m >> 3);
        luma_block_redo(g, B, S, item, mx, my, blk, cur4, pred4, ql);
    }
}

template <bool RC> /* rate-control variant, as k_me_pool's */
__global__ __launch_bounds__(64 * FIX_WAVES) void k_me_fix(venc_geom g, venc_bufs B, venc_step S, int fuse_luma, int nz_blocks, int nz_regions, int nz_rpt, int ntx, uint32_t tx_magic) {
    __shared__ uint32_t s_win[(48 + 4) * FIX_WD];
    __shared__ uint32_t s_cur[16 * 4];
    __shared__ uint32_t s_best[FIX_WAVES];
    __shared__ int s_mv[2];
    /* Grid = (items, nz_blocks + FIX_BLOCKS): the item runs fastest in dispatch order, so that the few blocks with work -- the
     * first `count` redo blocks of every item, the follow-up blocks -- are all resident at once instead of queueing behind
     * the 256 redo blocks per item that only load their count and leave (three blocks of this kernel fit a CU). */
    const int item = (int)blockIdx.x + S.item0, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    __builtin_amdgcn_s_setprio(3); /* (two chains: this launch runs beside the other chain's search, whose eight waves per SIMD would leave its serial
                                      instruction streams an eighth of the issue slots) */
    /* the first nz_blocks rows of the grid (fuse_luma == 3): the 4x4 blocks the search's P5 listed instead of coding them
     * (venc_inter2.hip luma_block_defer) -- no part in the exhaustive redo or in the closing of the picture's SAD sum */
    if ((int)blockIdx.y < nz_blocks) {
        venc_qparams qln = g.ql, qcn = g.qc;
        if (RC) step_qparams(g, B, S, item, qln, qcn);
        extern __shared__ uint32_t s_dyn[]; /* 256 x nz_rpt bytes: the item's segment counts */
        luma_nz_redo(g, B, S, item, (int)blockIdx.y, nz_blocks, nz_regions, nz_rpt, ntx, tx_magic, qln, s_win, s_win + 260, s_dyn);
        return;
    }
    const uint32_t fb = blockIdx.y - (uint32_t)nz_blocks; /* redo block of this item */
    /* the block's first list entry is fetched together with the count (it is meaningful only below the count): one memory
     * round trip less on a kernel whose whole cost is latency */
    const uint32_t first_mb = B.fix_list[(size_t)item * g.nmb + min((int)fb, g.nmb - 1)];
    const uint32_t count = min(B.fix_count[item], (uint32_t)g.nmb);
    /* blocks with a listed macroblock (under rate control: block 0 too when nothing is listed: somebody has to close the sum) */
    const uint32_t reporting = RC ? max(1u, min(count, (uint32_t)FIX_BLOCKS)) : count;
    if (fb >= reporting) return;
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    venc_qparams qlf = g.ql, qcf = g.qc;
    if (RC) step_qparams(g, B, S, item, qlf, qcf);
    const int dy0 = wave * FIX_NDY;          /* first dy index of this wave */
    const int dxi = lane < 33 ? lane : 32;   /* lanes 33..63 repeat column 32 and are ignored */
    /* the ranks of this lane's candidates do not depend on the macroblock: fetched before anything else, used last */
    uint32_t rk[FIX_NDY];
#pragma unroll
    for (int k = 0; k < FIX_NDY; k++) rk[k] = (fb < count && dy0 + k <= 32 && lane < 33) ? (uint32_t)B.me_rank[(dy0 + k) * 33 + dxi] : 0x7ffu;
    for (uint32_t i = fb; i < count; i += FIX_BLOCKS) {
        const int mb = (int)(i == fb ? first_mb : B.fix_list[(size_t)item * g.nmb + i]);
        const int mx = mb % g.mbw, my = mb / g.mbw;
        const int x0 = mx * 16, y0 = my * 16;
        __syncthreads();
        if (tid < 52 * 3) { /* 52 rows x three 16-byte pieces; a piece is wholly inside or wholly outside the coded picture */
            const int r = tid / 3, q = tid - r * 3;
            const int x = x0 - 16 + 16 * q;
            uint4 v = *(const uint4 *)(ref + (size_t)clampi(y0 - 16 + r, 0, g.ch - 1) * rstride + clampi(x, 0, g.cw - 16));
            if (x < 0) {
                const uint32_t e = (v.x & 255u) * 0x01010101u;
                v = make_uint4(e, e, e, e);
            } else if (x >= g.cw) {
                const uint32_t e = (v.w >> 24) * 0x01010101u;
                v = make_uint4(e, e, e, e);
            }
            uint32_t *d = s_win + r * FIX_WD + 4 * q; /* rows are 52 bytes apart: dword stores */
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
            if (q == 0) s_win[r * FIX_WD + 12] = 0;
        } else if (tid >= 192 && tid < 256) {
            const int t = tid - 192, r = t >> 2, d = t & 3;
            const uint8_t *row = cur + (size_t)min(y0 + r, g.h - 1) * cstride;
            const int xb = x0 + 4 * d;
            s_cur[t] = (uint32_t)row[min(xb, g.w - 1)] | ((uint32_t)row[min(xb + 1, g.w - 1)] << 8) |
                       ((uint32_t)row[min(xb + 2, g.w - 1)] << 16) | ((uint32_t)row[min(xb + 3, g.w - 1)] << 24);
        }
        __syncthreads();
        uint32_t c[16][4];
#pragma unroll
        for (int y = 0; y < 16; y++)
#pragma unroll
            for (int j = 0; j < 4; j++) c[y][j] = s_cur[y * 4 + j];
        uint32_t acc[FIX_NDY];
#pragma unroll
        for (int k = 0; k < FIX_NDY; k++) acc[k] = 0;
        const int dw = dxi >> 2;
        const uint32_t sh = (uint32_t)(dxi & 3);
#pragma unroll
        for (int r = 0; r < FIX_NDY + 15; r++) {
            const uint32_t *row = s_win + min(dy0 + r, 51) * FIX_WD + dw;
            const uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3], d4 = row[4];
            const uint32_t r0 = __builtin_amdgcn_alignbyte(d1, d0, sh), r1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            const uint32_t r2 = __builtin_amdgcn_alignbyte(d3, d2, sh), r3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
#pragma unroll
            for (int y = 0; y < 16; y++) {
                const int k = r - y;
                if (k >= 0 && k < FIX_NDY) {
                    uint32_t a = acc[k];
                    a = __builtin_amdgcn_sad_u8(r0, c[y][0], a);
                    a = __builtin_amdgcn_sad_u8(r1, c[y][1], a);
                    a = __builtin_amdgcn_sad_u8(r2, c[y][2], a);
                    a = __builtin_amdgcn_sad_u8(r3, c[y][3], a);
                    acc[k] = a;
                }
            }
        }
        /* key = SAD << 11 | rank is unique per candidate: the thread that holds the block's minimum knows its own
         * displacement (no table lookup from the winning rank) */
        uint32_t mine = 0xffffffffu;
        int mine_k = 0;
#pragma unroll
        for (int k = 0; k < FIX_NDY; k++) {
            const uint32_t key = rk[k] == 0x7ffu ? 0xffffffffu : (acc[k] << 11) | rk[k];
            if (key < mine) mine = key, mine_k = k;
        }
        const uint32_t hm = half_min32(mine, lane);
        const uint32_t wm = min((uint32_t)__builtin_amdgcn_readlane((int)hm, 0), (uint32_t)__builtin_amdgcn_readlane((int)hm, 32));
        if (lane == 0) s_best[wave] = wm;
        __syncthreads();
        uint32_t best = s_best[0];
#pragma unroll
        for (int k = 1; k < FIX_WAVES; k++) best = min(best, s_best[k]);
        if (mine == best) {
            const int dx = dxi - 16, dy = dy0 + mine_k - 16;
            if (RC) { /* the picture's SAD sum holds what k_me_pool found for this macroblock: correct it */
                const uint32_t old = B.mvsad[(size_t)item * g.nmb + mb] >> 16;
                const unsigned long long before = atomicAdd(&B.sad_cur[item], (unsigned long long)(long long)((int)(best >> 11) - (int)old));
                asm volatile("" ::"v"((uint32_t)before)); /* (consumed: the wave waits for the atomic's return) */
            }
            B.mvsad[(size_t)item * g.nmb + mb] = ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
            s_mv[0] = dxi, s_mv[1] = dy0 + mine_k; /* window position of the winner */
        }
        if (fuse_luma) {
            /* the luma transform of this macroblock (k_me_pool left it out with the search), from the window and the
             * current block already in LDS: lanes 0-15 of wave 0, one 4x4 block each */
            __syncthreads();
            if (tid < 64) {
                const int blk = lane & 15, bx = blk_x(blk) * 4, by = blk_y(blk) * 4;
                const int boff = s_mv[0] + bx;
                const uint32_t sh = (uint32_t)(boff & 3);
                const uint32_t *wp = s_win + (s_mv[1] + by) * FIX_WD + (boff >> 2);
                uint32_t cq[4], pq[4];
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    cq[y] = s_cur[(by + y) * 4 + (bx >> 2)];
                    pq[y] = __builtin_amdgcn_alignbyte(wp[y * FIX_WD + 1], wp[y * FIX_WD], sh);
                }
                luma_block_code(g, B, S, item, lane, mb, lane < 16, cq, pq, qlf);
            }
        }
    }
    /* Rate control: the picture's SAD sum is final once every block of this launch has made its corrections.  The block that
     * reports in last derives the NEXT picture's QP from it (spec/CODEC.md section 1a: the one 64-bit division per picture
     * and item) and leaves it in the next step's slot of the QP ring, where that step's kernels find it. */
    if (RC) {
        /* every correction above was a RETURNING atomic whose result the thread consumed: it has been performed (device-wide, at
         * the memory side) before the thread got here, so the arrival below cannot overtake it.  (An agent-scope fence does the
         * same and costs 0.9 ms per 1080p pass: it writes back and invalidates the L2 the search has just filled -- k_me_fix
         * 0.54 -> 1.46 ms, `--crf` 129 -> 115 Gpx/s, same-box A/B gpurun_out/r03_fixx.) */
        __syncthreads();
        if (tid == 0) {
            if (atomicAdd(&B.fix_done[item], 1u) == reporting - 1u) {
                B.fix_done[item] = 0u; /* for the next launch (same stream: ordered) */
                const unsigned long long sum = atomicAdd(&B.sad_cur[item], 0ull);
                B.qp_next[item] = venc_rc_qp(S.qp_base, 0, 1, sum, g.nmb);
            }
        }
    }
}
