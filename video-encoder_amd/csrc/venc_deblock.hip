/*
 * venc_deblock.hip -- k_deblock: the in-loop deblocking filter (H.264 8.7; spec/CODEC.md section 5a; SURVEY.md
 * section 8 row f3).  Behind the reference's x264_encoder_encode (av_encode.c:970) this is libx264's loop filter, on in
 * every preset the reference can select (av_encode.c:365).
 *
 * The standard filters macroblocks in raster order, and an edge sees what the earlier edges left behind: the left edge
 * of a macroblock reads columns its left neighbour's HORIZONTAL edges have already changed, so "all vertical edges of
 * the picture, then all horizontal ones" is a different filter.  What is equivalent is a wavefront: macroblock (x, y)
 * may run once (x - 1, y) and (x + 1, y - 1) are done.  With disable_deblocking_filter_idc = 2 slices are independent,
 * so: one block per (slice, item), one wave per macroblock row of the slice (rows beyond the block's waves in further
 * rounds).  Rows advance on their own on LDS progress flags (no block-wide barrier).
 *
 * STRIPS (round 3).  The first version moved one macroblock at a time: records, samples and stores were three dependent
 * memory round trips per macroblock, 16 + 4 different 128-byte lines touched for 16 bytes each, and the nine rows of a
 * slice shared one CU's path to L2 -- 1.1 us per macroblock for a row alone, 3.7 us with nine rows in flight, whatever
 * the arithmetic did.  Now a wave works on a STRIP of DBK_S macroblocks of its row (4: the skew costs less than the second round of pieces saves):
 *   load      the records of the strip and of the row above it (one load instruction), the strip's 16 luma / 2 x 8
 *             chroma rows as whole 128- / 64-byte pieces -- issued before the wait for the row above, they depend on
 *             nobody -- and, after the wait, the four rows above; the four columns to the left are carried over in LDS
 *   strengths all at once: lane = (macroblock, direction, edge), four segments each, kept in a register
 *   filter    the strip's macroblocks one after the other in the LDS tile (the dependency the standard's raster order
 *             leaves inside a row), eight phases each -- vertical edges 0-3, horizontal edges 0-3, a phase only if one
 *             of its segments has a strength: lanes 0-15 a luma line across the edge, lanes 16-31 a chroma line
 *   store     the tile as whole pieces, then the row's progress: a row runs one strip plus one macroblock behind the row above
 * Memory instructions per macroblock fall eightfold, a wave's dependent round trips from three per macroblock to three
 * per strip; the price is a wavefront skew of two strips per row instead of two macroblocks.
 * Luma and chroma can run as separate launches (planes = 1, 2): under the fused search the chroma reconstruction of a
 * step is produced on the side chain, after the main chain has already filtered luma for the next search.
 */

#define DBK_MAX_WAVES 16
#define DBK_MAX_ROWS 512

__constant__ uint8_t c_dbk_alpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
                                        15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__constant__ uint8_t c_dbk_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                       6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__constant__ uint8_t c_dbk_tc0[52][4] = { /* bS = 1, 2, 3 (fourth column unused) */
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 1, 0}, {0, 0, 1, 0}, {0, 0, 1, 0}, {0, 0, 1, 0}, {0, 1, 1, 0},
    {0, 1, 1, 0}, {1, 1, 1, 0}, {1, 1, 1, 0}, {1, 1, 1, 0}, {1, 1, 1, 0}, {1, 1, 2, 0}, {1, 1, 2, 0}, {1, 1, 2, 0}, {1, 1, 2, 0}, {1, 2, 3, 0}, {1, 2, 3, 0},
    {2, 2, 3, 0}, {2, 2, 4, 0}, {2, 3, 4, 0}, {2, 3, 4, 0}, {3, 3, 5, 0}, {3, 4, 6, 0}, {3, 4, 6, 0}, {4, 5, 7, 0}, {4, 5, 8, 0}, {4, 6, 9, 0}, {5, 7, 10, 0},
    {6, 8, 11, 0}, {6, 8, 13, 0}, {7, 10, 14, 0}, {8, 11, 16, 0}, {9, 12, 18, 0}, {10, 13, 20, 0}, {11, 15, 23, 0}, {13, 17, 25, 0}};

#ifndef DBK_S
#define DBK_S 4 /* macroblocks per strip (4 or 8) */
#endif
#define DBK_YD (4 * DBK_S + 2) /* luma tile row in dwords: 4 columns to the left + 16 DBK_S + 4 spare bytes */
#define DBK_CD (2 * DBK_S + 2) /* chroma tile row in dwords: 4 + 8 DBK_S + 4 bytes */
#define DBK_K (DBK_S / 4)      /* rounds of 64 lanes for the strip's own rows: 16 rows x DBK_S 16-byte pieces (luma), 2 x 8 rows x DBK_S 8-byte pieces (chroma) */
struct dbk_strip {
    uint32_t y[20 * DBK_YD];        /* rows -4 .. 15; byte column 4 + x, x = -4 .. 16 DBK_S - 1 */
    uint32_t c[2][12 * DBK_CD];     /* Cb, Cr: rows -4 .. 7; byte column 4 + x, x = -4 .. 8 DBK_S - 1 */
    uint32_t info[2][DBK_S + 1][9]; /* records (8 dwords of venc_mbinfo + the vector word): [0] this row -- entry 0 the macroblock left of
                                       the strip, 1 .. DBK_S the strip --, [1] the row above (entries 1 .. DBK_S) */
};

/* the line filter: venc_lines.h (shared with k_deblock_ls and the host-check library) */
#define dbk_line venc_dbk_line

__global__ __launch_bounds__(64 * DBK_MAX_WAVES) void k_deblock(venc_geom g, venc_bufs B, venc_step S, int planes) {
    extern __shared__ __attribute__((aligned(16))) dbk_strip s_t[]; /* one tile per wave of the block (the launcher sizes it) */
    /* macroblocks done in each row of the slice.  Read and written as relaxed workgroup-scope ATOMICS: a `volatile` access
     * to this array (the first version's) is compiled to flat_load / flat_store sc0 sc1 -- every poll and every publication a
     * trip through the vector-memory path, 400 000 of them per launch -- where the atomic becomes a ds_read / ds_write */
    __shared__ int s_prog[DBK_MAX_ROWS];
    const int item = blockIdx.z, sl = S.slice0 + (int)blockIdx.x, tid = threadIdx.x;
    const int nw = blockDim.x >> 6, wv = tid >> 6, lane = tid & 63;
    const int r0 = B.slice_first_row[sl], r1 = B.slice_first_row[sl + 1], nrows = r1 - r0;
    for (int i = tid; i < min(nrows, DBK_MAX_ROWS); i += blockDim.x) s_prog[i] = 0;
    __syncthreads();
    dbk_strip &T = s_t[wv];
    const venc_mbinfo *infos = B.mbinfo + (size_t)item * g.nmb;
    const uint32_t *mvs = B.mvsad + (size_t)item * g.nmb;
    int ystride, cstride;
    uint8_t *py = rec_plane(B, g, S.parity, item, 0, ystride);
    uint8_t *pc0 = rec_plane(B, g, S.parity, item, 1, cstride), *pc1 = rec_plane(B, g, S.parity, item, 2, cstride);
    /* the picture's QP (one per picture): fixed, or the rate control's */
    const int qp = __builtin_amdgcn_readfirstlane(S.rc_mode ? (S.is_idr ? S.qp_idr : B.qp_cur[item]) : g.qp);
    const int qpc = venc_chroma_qp(qp);
    const int mvthr = g.subpel ? 4 : 1; /* "four quarter samples apart" in the units the vectors are kept in */
    const bool do_y = (planes & 1) != 0, do_c = (planes & 2) != 0;
    const bool lane_y = lane < 16, lane_c = lane >= 16 && lane < 32;
    const bool chroma = lane_c;
    const int alpha = chroma ? c_dbk_alpha[qpc] : c_dbk_alpha[qp], beta = chroma ? c_dbk_beta[qpc] : c_dbk_beta[qp];
    const uint32_t tcw = *(const uint32_t *)c_dbk_tc0[chroma ? qpc : qp];
    for (int rl = wv; rl < nrows; rl += nw) {
        const int my = r0 + rl;
        const bool aT = rl > 0;
        for (int x0 = 0; x0 < g.mbw; x0 += DBK_S) {
            const int n = min(DBK_S, g.mbw - x0); /* macroblocks of this strip (only a row's last strip can be short) */
            /* ---- what the last strip leaves behind: the record of its last macroblock, its last four columns ---- */
            uint32_t ly = 0, lc = 0;
            if (x0 > 0) {
                if (lane < 9) T.info[0][0][lane] = T.info[0][DBK_S][lane];
                if (do_y && lane < 16) ly = T.y[(4 + lane) * DBK_YD + 4 * DBK_S];
                if (do_c && lane < 16) lc = T.c[lane >> 3][(4 + (lane & 7)) * DBK_CD + 2 * DBK_S];
            }
            /* ---- loads that wait for nobody: the records (this row's and the row's above: nobody writes them here), the strip's own rows ---- */
            uint4 rec = make_uint4(0u, 0u, 0u, 0u);
            /* lanes 0 .. 4 S - 1: the two halves of a record (this row's strip, then the row's above); 4 S .. 6 S - 1: the vector words */
            const bool rec_lane = lane < 4 * DBK_S, mv_lane = !rec_lane && lane < 6 * DBK_S;
            const int rec_j = rec_lane ? (lane % (2 * DBK_S)) >> 1 : (lane - 4 * DBK_S) % DBK_S;
            const int rec_above = rec_lane ? lane / (2 * DBK_S) : (lane - 4 * DBK_S) / DBK_S;
            if (rec_lane || mv_lane) {
                const int m2 = my * g.mbw + x0 + min(rec_j, n - 1) - (rec_above && aT ? g.mbw : 0);
                if (rec_lane) rec = ((const uint4 *)(infos + m2))[lane & 1];
                else rec.x = mvs[m2];
            }
            uint4 yo[DBK_K];
            uint2 co[DBK_K];
#pragma unroll
            for (int k = 0; k < DBK_K; k++) {
                yo[k] = make_uint4(0u, 0u, 0u, 0u), co[k] = make_uint2(0u, 0u);
                const int p = lane + 64 * k; /* luma piece: row p / S (0 .. 15), macroblock p % S; chroma piece: plane p / 8 S, row (p / S) % 8, macroblock p % S */
                if (p % DBK_S < n) {
                    if (do_y) yo[k] = *(const uint4 *)(py + (size_t)(my * 16 + p / DBK_S) * ystride + (x0 + p % DBK_S) * 16);
                    if (do_c) co[k] = *(const uint2 *)((p / (8 * DBK_S) ? pc1 : pc0) + (size_t)(my * 8 + (p / DBK_S) % 8) * cstride + (x0 + p % DBK_S) * 8);
                }
            }
            /* ---- the row above is through the macroblock right of this strip (or finished) ---- */
            uint4 ya = make_uint4(0u, 0u, 0u, 0u);
            uint2 ca = make_uint2(0u, 0u);
            if (aT) {
                const int need = min(x0 + DBK_S + 1, g.mbw);
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_prog[rl - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
                __threadfence_block();
                if (lane % DBK_S < n) { /* luma: lanes 0 .. 4 S - 1 = (row, macroblock); chroma: lanes 0 .. 8 S - 1 = (plane, row, macroblock) */
                    if (do_y && lane < 4 * DBK_S) ya = *(const uint4 *)(py + (size_t)(my * 16 - 4 + lane / DBK_S) * ystride + (x0 + lane % DBK_S) * 16);
                    if (do_c && lane < 8 * DBK_S) ca = *(const uint2 *)((lane / (4 * DBK_S) ? pc1 : pc0) + (size_t)(my * 8 - 4 + (lane / DBK_S) % 4) * cstride + (x0 + lane % DBK_S) * 8);
                }
            }
            __builtin_amdgcn_wave_barrier();
            /* ---- into the tile ---- */
            if (rec_lane) {
                uint32_t *d = &T.info[rec_above][1 + rec_j][(lane & 1) * 4];
                d[0] = rec.x, d[1] = rec.y, d[2] = rec.z, d[3] = rec.w;
            } else if (mv_lane)
                T.info[rec_above][1 + rec_j][8] = rec.x;
            if (do_y) {
                if (lane < 16) T.y[(4 + lane) * DBK_YD] = ly;
#pragma unroll
                for (int k = 0; k < DBK_K; k++) {
                    const int p = lane + 64 * k;
                    uint32_t *d = &T.y[(4 + p / DBK_S) * DBK_YD + 1 + 4 * (p % DBK_S)];
                    d[0] = yo[k].x, d[1] = yo[k].y, d[2] = yo[k].z, d[3] = yo[k].w;
                }
                if (lane < 4 * DBK_S) {
                    uint32_t *d = &T.y[(lane / DBK_S) * DBK_YD + 1 + 4 * (lane % DBK_S)];
                    d[0] = ya.x, d[1] = ya.y, d[2] = ya.z, d[3] = ya.w;
                }
            }
            if (do_c) {
                if (lane < 16) T.c[lane >> 3][(4 + (lane & 7)) * DBK_CD] = lc;
#pragma unroll
                for (int k = 0; k < DBK_K; k++) {
                    const int p = lane + 64 * k;
                    uint32_t *d = &T.c[p / (8 * DBK_S)][(4 + (p / DBK_S) % 8) * DBK_CD + 1 + 2 * (p % DBK_S)];
                    d[0] = co[k].x, d[1] = co[k].y;
                }
                if (lane < 8 * DBK_S) {
                    uint32_t *d = &T.c[lane / (4 * DBK_S)][((lane / DBK_S) % 4) * DBK_CD + 1 + 2 * (lane % DBK_S)];
                    d[0] = ca.x, d[1] = ca.y;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            /* ---- strengths of the whole strip: lane = macroblock * 8 + direction * 4 + edge, four segments in one word ---- */
            uint32_t bsw = 0;
            {
                const int j = lane >> 3, dir = (lane >> 2) & 1, e = lane & 3;
                const bool aL = x0 + j > 0;
                if (lane < 8 * DBK_S && j < n && (e > 0 || (dir ? aT : aL))) {
                    const uint8_t *own = (const uint8_t *)T.info[0][1 + j];
                    const uint32_t *othw = e ? T.info[0][1 + j] : (dir ? T.info[1][1 + j] : T.info[0][j]);
                    const uint8_t *oth = (const uint8_t *)othw;
                    if ((own[25] | oth[25]) & 1) /* venc_mbinfo.flags: intra */
                        bsw = e ? 0x03030303u : 0x04040404u;
                    else {
                        uint32_t mvbs = 0;
                        if (e == 0) {
                            const uint32_t a = T.info[0][1 + j][8], b = othw[8];
                            const int ddx = (int)(int8_t)(a & 255u) - (int)(int8_t)(b & 255u), ddy = (int)(int8_t)((a >> 8) & 255u) - (int)(int8_t)((b >> 8) & 255u);
                            mvbs = (abs(ddx) >= mvthr || abs(ddy) >= mvthr) ? 1u : 0u;
                        }
#pragma unroll
                        for (int seg = 0; seg < 4; seg++) {
                            const int bq = dir ? 4 * e + seg : 4 * seg + e;
                            const int bp = e ? (dir ? 4 * (e - 1) + seg : 4 * seg + e - 1) : (dir ? 12 + seg : 4 * seg + 3);
                            bsw |= ((own[bq] | oth[bp]) ? 2u : mvbs) << (8 * seg);
                        }
                    }
                }
            }
            const unsigned long long anyb = __ballot(bsw != 0u); /* byte j: macroblock j has an edge to filter */
            /* ---- the strip's macroblocks, one after the other; of each only the phases that have a strength (byte j of the ballot
             * = macroblock j's phases: lane = j * 8 + phase) ---- */
            {
                const int k = chroma ? (lane & 7) : lane & 15; /* line along the edge */
                uint8_t *const tile = chroma ? (uint8_t *)T.c[(lane >> 3) & 1] : (uint8_t *)T.y;
                const int stride = chroma ? 4 * DBK_CD : 4 * DBK_YD, xstep = chroma ? 8 : 16, estep = chroma ? 2 : 4;
                uint8_t *const vbase = tile + (4 + k) * stride; /* a row's line across a vertical edge starts here + edge position */
                uint8_t *const hbase = tile + 4 + k;            /* a column's line across a horizontal edge */
                const bool on = (lane_y && do_y) || (lane_c && do_c);
                for (int j = 0; j < n; j++) {
                    uint32_t m = (uint32_t)(anyb >> (8 * j)) & 255u; /* (scalar) */
                    while (m) {
                        const int ph = __builtin_ctz(m);
                        m &= m - 1u;
                        const int e = ph & 3;
                        const uint32_t bs4 = (uint32_t)__builtin_amdgcn_readlane((int)bsw, j * 8 + ph);
                        if (on && !(chroma && (e & 1))) {
                            const int lbs = (int)((bs4 >> (8 * (chroma ? k >> 1 : k >> 2))) & 255u);
                            const int tc0 = (int)((tcw >> (8 * max(lbs - 1, 0))) & 255u);
                            int sm[8];
                            if (ph < 4) { /* vertical edge: p3 .. q3 are two aligned dwords of the row */
                                uint32_t *row = (uint32_t *)(vbase + xstep * j + estep * e);
                                const uint32_t w0 = row[0], w1 = row[1];
                                sm[0] = (int)(w0 & 255u), sm[1] = (int)((w0 >> 8) & 255u), sm[2] = (int)((w0 >> 16) & 255u), sm[3] = (int)(w0 >> 24);
                                sm[4] = (int)(w1 & 255u), sm[5] = (int)((w1 >> 8) & 255u), sm[6] = (int)((w1 >> 16) & 255u), sm[7] = (int)(w1 >> 24);
                                dbk_line(sm, lbs, alpha, beta, tc0, chroma);
                                row[0] = (uint32_t)sm[0] | ((uint32_t)sm[1] << 8) | ((uint32_t)sm[2] << 16) | ((uint32_t)sm[3] << 24);
                                row[1] = (uint32_t)sm[4] | ((uint32_t)sm[5] << 8) | ((uint32_t)sm[6] << 16) | ((uint32_t)sm[7] << 24);
                            } else { /* horizontal edge: eight rows of the column */
                                uint8_t *col = hbase + xstep * j + estep * e * stride;
#pragma unroll
                                for (int i2 = 0; i2 < 8; i2++) sm[i2] = col[i2 * stride];
                                dbk_line(sm, lbs, alpha, beta, tc0, chroma);
#pragma unroll
                                for (int i2 = 1; i2 < 7; i2++) col[i2 * stride] = (uint8_t)sm[i2];
                            }
                        }
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            /* ---- store: the tile as it was loaded -- the strip's rows, the rows above, the columns to its left (all of them final:
             * the row above finished with these columns before it published, the row below waits for this strip) ---- */
            if (anyb != 0ull) {
                if (do_y) {
#pragma unroll
                    for (int k = 0; k < DBK_K; k++) {
                        const int p = lane + 64 * k;
                        if (p % DBK_S < n) {
                            const uint32_t *d = &T.y[(4 + p / DBK_S) * DBK_YD + 1 + 4 * (p % DBK_S)];
                            *(uint4 *)(py + (size_t)(my * 16 + p / DBK_S) * ystride + (x0 + p % DBK_S) * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                        }
                    }
                    if (aT && lane < 4 * DBK_S && lane % DBK_S < n) {
                        const uint32_t *d = &T.y[(lane / DBK_S) * DBK_YD + 1 + 4 * (lane % DBK_S)];
                        *(uint4 *)(py + (size_t)(my * 16 - 4 + lane / DBK_S) * ystride + (x0 + lane % DBK_S) * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                    }
                    if (x0 > 0 && lane >= 32 && lane < 48) *(uint32_t *)(py + (size_t)(my * 16 + lane - 32) * ystride + x0 * 16 - 4) = T.y[(4 + lane - 32) * DBK_YD];
                }
                if (do_c) {
#pragma unroll
                    for (int k = 0; k < DBK_K; k++) {
                        const int p = lane + 64 * k;
                        if (p % DBK_S < n) {
                            const uint32_t *d = &T.c[p / (8 * DBK_S)][(4 + (p / DBK_S) % 8) * DBK_CD + 1 + 2 * (p % DBK_S)];
                            *(uint2 *)((p / (8 * DBK_S) ? pc1 : pc0) + (size_t)(my * 8 + (p / DBK_S) % 8) * cstride + (x0 + p % DBK_S) * 8) = make_uint2(d[0], d[1]);
                        }
                    }
                    if (aT && lane < 8 * DBK_S && lane % DBK_S < n) {
                        const uint32_t *d = &T.c[lane / (4 * DBK_S)][((lane / DBK_S) % 4) * DBK_CD + 1 + 2 * (lane % DBK_S)];
                        *(uint2 *)((lane / (4 * DBK_S) ? pc1 : pc0) + (size_t)(my * 8 - 4 + (lane / DBK_S) % 4) * cstride + (x0 + lane % DBK_S) * 8) = make_uint2(d[0], d[1]);
                    }
                    if (x0 > 0 && lane >= 48) {
                        const int p = (lane - 48) >> 3, r = lane & 7;
                        *(uint32_t *)((p ? pc1 : pc0) + (size_t)(my * 8 + r) * cstride + x0 * 8 - 4) = T.c[p][(4 + r) * DBK_CD];
                    }
                }
            }
            __threadfence_block(); /* the stores above before the flag below */
            if (lane == 0) __hip_atomic_store(&s_prog[rl], x0 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
