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
 * rounds), a row running at least two macroblocks behind the row above.  Rows advance on their own: a wave publishes
 * its progress in LDS after every macroblock and waits only for the row above (no block-wide barrier per step: most
 * macroblocks of a P picture have no edge to filter and cost a strength computation and a flag).
 *
 * One macroblock, one wave:
 *   strengths   lanes 0-31 = the 32 luma edge segments (2 directions x 4 edges x 4): 4 / 3 intra, 2 levels, 1 vectors
 *               apart, 0; nothing to do -> publish and go on
 *   tile        the macroblock with four columns to its left and four rows above, luma 20 x 20 and chroma 2 x 12 x 12,
 *               into the wave's LDS tile
 *   8 phases    vertical edges 0-3, horizontal edges 0-3, each only if one of its segments has a strength: lanes 0-15 a
 *               luma line across the edge, lanes 16-31 a chroma line (edges 0 and 2), samples from / to the tile
 *   store       what the macroblock may have changed: itself, three columns to the left, three rows above
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

struct dbk_tile {
    uint32_t y[20 * 5];     /* rows -4 .. 15, columns -4 .. 15 */
    uint32_t c[2][12 * 3];  /* Cb, Cr: rows -4 .. 7, columns -4 .. 7 */
    uint32_t bs[8];         /* strengths as bytes: [direction][edge][segment] */
    uint32_t info[3][9];    /* the records of this macroblock, its left and its upper neighbour: 8 dwords of venc_mbinfo + the vector word */
};

/* one line across an edge: s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 (chroma: p1 p0 q0 q1 in s[2..5]); 8.7.2.3 / 8.7.2.4 */
DEVFN void dbk_line(int s[8], int bs, int alpha, int beta, int tc0, bool chroma) {
    const int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!(bs && abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    const bool ap = !chroma && abs(p2 - p0) < beta, aq = !chroma && abs(q2 - q0) < beta;
    if (bs < 4) {
        const int tc = chroma ? tc0 + 1 : tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
        const int delta = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
        s[3] = venc_clip_u8(p0 + delta);
        s[4] = venc_clip_u8(q0 - delta);
        const int avg = (p0 + q0 + 1) >> 1;
        if (ap) s[2] = p1 + min(max((p2 + avg - (p1 << 1)) >> 1, -tc0), tc0);
        if (aq) s[5] = q1 + min(max((q2 + avg - (q1 << 1)) >> 1, -tc0), tc0);
    } else {
        const bool small = !chroma && abs(p0 - q0) < ((alpha >> 2) + 2);
        if (small && ap) {
            s[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
            s[2] = (p2 + p1 + p0 + q0 + 2) >> 2;
            s[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        } else
            s[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        if (small && aq) {
            s[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
            s[5] = (p0 + q0 + q1 + q2 + 2) >> 2;
            s[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        } else
            s[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}

__global__ __launch_bounds__(64 * DBK_MAX_WAVES) void k_deblock(venc_geom g, venc_bufs B, venc_step S, int planes) {
    __shared__ dbk_tile s_t[DBK_MAX_WAVES];
    __shared__ int s_prog[DBK_MAX_ROWS]; /* macroblocks done in each row of the slice */
    const int item = blockIdx.z, sl = S.slice0 + (int)blockIdx.x, tid = threadIdx.x;
    const int nw = blockDim.x >> 6, wv = tid >> 6, lane = tid & 63;
    const int r0 = B.slice_first_row[sl], r1 = B.slice_first_row[sl + 1], nrows = r1 - r0;
    for (int i = tid; i < min(nrows, DBK_MAX_ROWS); i += blockDim.x) s_prog[i] = 0;
    __syncthreads();
    dbk_tile &T = s_t[wv];
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
    /* Memory instructions: few of them, few lanes each, wide -- one load for the records of a macroblock and the one above
     * (the left one's is last round's own), rows as 16-byte / 8-byte pieces, the left neighbour's last four columns carried
     * over in LDS, the columns to the left / rows above stored only when the edge next to them was filtered.  (First
     * version: six byte-wise loads of 32 lanes for the strengths and three dword loads / stores of 64 + 16 + 16 lanes per
     * plane and macroblock; this shape: 443 -> 421 us per luma launch at 1080p x 10 -- the memory instructions are not what
     * makes nine rows on one CU 3.4 times slower per macroblock than one) */
    for (int rl = wv; rl < nrows; rl += nw) {
        const int my = r0 + rl;
        const bool aT = rl > 0;
        bool left_in_tile = false; /* the tile holds the left neighbour's filtered samples (it had an edge to filter) */
        for (int mx = 0; mx < g.mbw; mx++) {
            if (aT) { /* the row above is two macroblocks ahead (or finished) */
                const int need = min(mx + 2, g.mbw);
                while (__builtin_amdgcn_readfirstlane(((volatile int *)s_prog)[rl - 1]) < need) __builtin_amdgcn_s_sleep(1);
                __threadfence_block();
            }
            const int mb = my * g.mbw + mx;
            const bool aL = mx > 0;
            /* ---- records: ONE load, lanes 0-8 this macroblock (8 dwords + vector), lanes 9-17 the one above; the left one is
             * last round's own ---- */
            if (lane < 9) T.info[1][lane] = T.info[0][lane]; /* (garbage for mx = 0: never read, aL is false) */
            __builtin_amdgcn_wave_barrier();
            if (lane < 18) {
                const int which = lane >= 9, k = lane - 9 * which; /* k < 8: record dword, 8: the vector */
                const int m2 = which ? (aT ? mb - g.mbw : mb) : mb;
                const uint32_t v = k < 8 ? ((const uint32_t *)(infos + m2))[k] : mvs[m2];
                T.info[which ? 2 : 0][k] = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            /* ---- strengths: lane = direction * 16 + edge * 4 + segment ---- */
            int bs = 0;
            if (lane < 32) {
                const int dir = lane >> 4, e = (lane >> 2) & 3, seg = lane & 3;
                if (e > 0 || (dir ? aT : aL)) {
                    const int nb = e ? 0 : (dir ? 2 : 1); /* the record of the block on the p side */
                    const int bq = dir ? 4 * e + seg : 4 * seg + e;
                    const int bp = e ? (dir ? 4 * (e - 1) + seg : 4 * seg + e - 1) : (dir ? 12 + seg : 4 * seg + 3);
                    const uint8_t *own = (const uint8_t *)T.info[0], *oth = (const uint8_t *)T.info[nb];
                    if ((own[25] | oth[25]) & 1) /* venc_mbinfo.flags */
                        bs = e ? 3 : 4;
                    else if (own[bq] | oth[bp])
                        bs = 2;
                    else if (e == 0) {
                        const uint32_t a = T.info[0][8], b = T.info[nb][8];
                        const int ddx = (int)(int8_t)(a & 255u) - (int)(int8_t)(b & 255u), ddy = (int)(int8_t)((a >> 8) & 255u) - (int)(int8_t)((b >> 8) & 255u);
                        bs = (abs(ddx) >= mvthr || abs(ddy) >= mvthr) ? 1 : 0;
                    }
                }
            }
            const bool any = __ballot(bs != 0) != 0ull;
            if (any) {
                if (lane < 32) ((uint8_t *)T.bs)[lane] = (uint8_t)bs;
                /* ---- tile: rows as 16-byte (luma) / 8-byte (chroma) pieces; the columns to the left from the old tile when it
                 * holds them ---- */
                uint32_t ly = 0, lc = 0;
                if (left_in_tile) {
                    if (do_y && lane < 16) ly = T.y[(4 + lane) * 5 + 4];
                    if (do_c && lane < 16) lc = T.c[lane >> 3][(4 + (lane & 7)) * 3 + 2];
                } else if (aL) {
                    if (do_y && lane < 16) ly = *(const uint32_t *)(py + (size_t)(my * 16 + lane) * ystride + mx * 16 - 4);
                    if (do_c && lane < 16) lc = *(const uint32_t *)(((lane >> 3) ? pc1 : pc0) + (size_t)(my * 8 + (lane & 7)) * cstride + mx * 8 - 4);
                }
                __builtin_amdgcn_wave_barrier();
                if (do_y) {
                    if (lane < 16) T.y[(4 + lane) * 5] = ly;
                    if (lane < 20) { /* lanes 0-15: the macroblock's rows; 16-19: the four rows above */
                        const int r = lane < 16 ? lane : lane - 20; /* -4 .. 15 */
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (r >= 0 || aT) v = *(const uint4 *)(py + (size_t)(my * 16 + r) * ystride + mx * 16);
                        uint32_t *d = &T.y[(4 + r) * 5 + 1];
                        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
                    }
                }
                if (do_c) {
                    if (lane < 16) T.c[lane >> 3][(4 + (lane & 7)) * 3] = lc;
                    if (lane < 24) { /* lanes 0-15: (plane, row); 16-23: (plane, row above) */
                        const int p = lane < 16 ? lane >> 3 : (lane - 16) >> 2, r = lane < 16 ? (lane & 7) : ((lane - 16) & 3) - 4;
                        uint2 v = make_uint2(0u, 0u);
                        if (r >= 0 || aT) v = *(const uint2 *)((p ? pc1 : pc0) + (size_t)(my * 8 + r) * cstride + mx * 8);
                        uint32_t *d = &T.c[p][(4 + r) * 3 + 1];
                        d[0] = v.x, d[1] = v.y;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                /* ---- the eight phases ---- */
                for (int ph = 0; ph < 8; ph++) {
                    const int dir = ph >> 2, e = ph & 3;
                    const uint32_t bs4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.bs[ph]);
                    if (bs4 == 0u) continue; /* (wave-uniform) */
                    const bool act = (lane_y && do_y) || (lane_c && do_c && !(e & 1));
                    if (act) {
                        const int k = chroma ? (lane & 7) : lane; /* line along the edge */
                        const int lbs = (int)((bs4 >> (8 * (chroma ? k >> 1 : k >> 2))) & 255u);
                        uint8_t *base;
                        int stride, pos; /* tile row stride in bytes, position of the edge inside the tile */
                        if (chroma) base = (uint8_t *)T.c[(lane >> 3) & 1], stride = 12, pos = 4 + 2 * e;
                        else base = (uint8_t *)T.y, stride = 20, pos = 4 + 4 * e;
                        int sm[8];
                        if (dir == 0) {
                            uint8_t *row = base + (4 + k) * stride + pos - 4;
#pragma unroll
                            for (int i2 = 0; i2 < 8; i2++) sm[i2] = row[i2];
                            dbk_line(sm, lbs, alpha, beta, (int)((tcw >> (8 * max(lbs - 1, 0))) & 255u), chroma);
#pragma unroll
                            for (int i2 = 1; i2 < 7; i2++) row[i2] = (uint8_t)sm[i2];
                        } else {
                            uint8_t *col = base + (pos - 4) * stride + 4 + k;
#pragma unroll
                            for (int i2 = 0; i2 < 8; i2++) sm[i2] = col[i2 * stride];
                            dbk_line(sm, lbs, alpha, beta, (int)((tcw >> (8 * max(lbs - 1, 0))) & 255u), chroma);
#pragma unroll
                            for (int i2 = 1; i2 < 7; i2++) col[i2 * stride] = (uint8_t)sm[i2];
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
                /* ---- store: the macroblock and the rows above it as loaded; the columns left of it when its left edge was filtered ---- */
                const bool vleft = aL && T.bs[0] != 0u, htop = aT && T.bs[4] != 0u; /* (wave-uniform: LDS words) */
                if (do_y) {
                    if (lane < 20) {
                        const int r = lane < 16 ? lane : lane - 20;
                        if (r >= 0 || htop) {
                            const uint32_t *d = &T.y[(4 + r) * 5 + 1];
                            *(uint4 *)(py + (size_t)(my * 16 + r) * ystride + mx * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                        }
                    }
                    if (vleft && lane >= 32 && lane < 48) *(uint32_t *)(py + (size_t)(my * 16 + lane - 32) * ystride + mx * 16 - 4) = T.y[(4 + lane - 32) * 5];
                }
                if (do_c) {
                    if (lane < 24) {
                        const int p = lane < 16 ? lane >> 3 : (lane - 16) >> 2, r = lane < 16 ? (lane & 7) : ((lane - 16) & 3) - 4;
                        if (r >= 0 || htop) {
                            const uint32_t *d = &T.c[p][(4 + r) * 3 + 1];
                            *(uint2 *)((p ? pc1 : pc0) + (size_t)(my * 8 + r) * cstride + mx * 8) = make_uint2(d[0], d[1]);
                        }
                    }
                    if (vleft && lane >= 32 && lane < 48) {
                        const int p = (lane - 32) >> 3, r = lane & 7;
                        *(uint32_t *)((p ? pc1 : pc0) + (size_t)(my * 8 + r) * cstride + mx * 8 - 4) = T.c[p][(4 + r) * 3];
                    }
                }
            }
            left_in_tile = any;
            __threadfence_block(); /* the stores above before the flag below */
            if (lane == 0) ((volatile int *)s_prog)[rl] = mx + 1;
        }
    }
}
