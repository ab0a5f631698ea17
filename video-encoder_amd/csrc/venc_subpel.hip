/*
 * venc_subpel.hip -- k_subpel: sub-sample refinement of the full search's vectors (spec/CODEC.md section 2a;
 * SURVEY.md section 8 row f3).  Behind the reference's x264_encoder_encode (av_encode.c:970) this is the subpixel
 * motion estimation its presets (av_encode.c:365) ask libx264 for.
 *
 * One wave per macroblock, four macroblocks per block.  Around the full search's vector the wave builds, in its own
 * piece of LDS, the macroblock's corner of the DOUBLED sample grid of H.264 8.4.2.2.1:
 *   window  22 x 22 reference samples (rows / columns -3 .. 18 of the displaced block, coordinates clamped)
 *   b1      unclipped horizontal six-tap sums, 22 rows x 17 columns (column c: between samples c and c + 1, c = -1 .. 15)
 *   b       clip((b1 + 16) >> 5)          rows -1 .. 16
 *   h       vertical six-tap of the window, 17 rows (between r and r + 1, r = -1 .. 15) x 18 columns (-1 .. 16)
 *   j       vertical six-tap of b1, clip((sum + 512) >> 10), 17 x 17
 * Every candidate of both stages -- the eight half-sample neighbours of the vector, then the eight quarter-sample
 * neighbours of that stage's winner -- reads its prediction from these planes: a sample of the doubled grid, or the
 * rounded average of two (Table 8-12), four samples per lane (lane = row, quarter of a row), SAD by v_sad_u8 and a wave
 * sum.  The winner (minimum of (SAD << 4) | index, index 0 = the stage's centre) goes to B.mvq in QUARTER-sample units
 * -- the transform and entropy kernels of the step are handed that array as their vectors; the search's own integer
 * vectors stay where the next step's search reads its start points -- and its prediction to a 256-byte block of
 * B.pred_luma, which the luma transform (k_inter_fused, venc_inter2.hip) reads instead of the reference plane.
 */

#define SP_WS 24   /* window row stride in bytes: columns -3 .. 20 (22 used) */
#define SP_WR 22   /* window rows -3 .. 18 */
#define SP_PS 20   /* row stride of the b / h / j planes in bytes */
#define SP_B1S 18  /* row stride of b1 in int16 */

struct sp_lds {
    uint32_t win[SP_WR * SP_WS / 4 + 1]; /* (+1 dword: the unaligned four-sample read takes two dwords) */
    uint32_t bp[18 * SP_PS / 4 + 1];
    uint32_t hp[17 * SP_PS / 4 + 1];
    uint32_t jp[17 * SP_PS / 4 + 1];
    int16_t b1[SP_WR * SP_B1S];
};

DEVFN int sp_tap6(int a, int b, int c, int d, int e, int f) { return a + f - 5 * (b + e) + 20 * (c + d); }

/* four samples of the doubled grid for the lane's row y and columns x4 .. x4 + 3: half-sample coordinates (hx + 2 x, hy + 2 y)
 * relative to the displaced block; the plane follows from the parities of (hx, hy), which are wave-uniform */
DEVFN uint32_t sp_grid4(const sp_lds &L, int hx, int hy, int y, int x4) {
    const int xi = (hx >> 1) + x4, yi = (hy >> 1) + y;
    const uint8_t *base;
    int off;
    if (!(hx & 1) && !(hy & 1)) base = (const uint8_t *)L.win, off = (yi + 3) * SP_WS + xi + 3;
    else if ((hx & 1) && !(hy & 1)) base = (const uint8_t *)L.bp, off = (yi + 1) * SP_PS + xi + 1;
    else if (!(hx & 1)) base = (const uint8_t *)L.hp, off = (yi + 1) * SP_PS + xi + 1;
    else base = (const uint8_t *)L.jp, off = (yi + 1) * SP_PS + xi + 1;
    const uint32_t *d = (const uint32_t *)(base + (off & ~3));
    return __builtin_amdgcn_alignbyte(d[1], d[0], (uint32_t)(off & 3));
}
/* prediction at (qx, qy) quarter units from the displaced block, four samples at once: both even -- a sample of the doubled
 * grid; one odd -- the average of the two grid samples either side along that axis; both odd -- of the two corners of the
 * grid cell that are b / h samples (odd coordinate sum) */
DEVFN uint32_t sp_pred4(const sp_lds &L, int qx, int qy, int y, int x4) {
    const int ox = qx & 1, oy = qy & 1;
    const int hx0 = (qx - ox) >> 1, hy0 = (qy - oy) >> 1;
    if (!ox && !oy) return sp_grid4(L, hx0, hy0, y, x4);
    uint32_t a, b;
    if (ox && !oy) a = sp_grid4(L, hx0, hy0, y, x4), b = sp_grid4(L, hx0 + 1, hy0, y, x4);
    else if (!ox) a = sp_grid4(L, hx0, hy0, y, x4), b = sp_grid4(L, hx0, hy0 + 1, y, x4);
    else if ((hx0 + hy0) & 1) a = sp_grid4(L, hx0, hy0, y, x4), b = sp_grid4(L, hx0 + 1, hy0 + 1, y, x4);
    else a = sp_grid4(L, hx0 + 1, hy0, y, x4), b = sp_grid4(L, hx0, hy0 + 1, y, x4);
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); /* (a + b + 1) >> 1 in every byte */
}
DEVFN uint32_t sp_wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63); }

__global__ __launch_bounds__(256) void k_subpel(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ sp_lds s_l[4];
    const int item = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    sp_lds &L = s_l[wave];
    const int mb_raw = S.mb0 + (int)blockIdx.x * 4 + wave;
    const bool valid = mb_raw < S.mb1;
    const int mb = valid ? mb_raw : S.mb1 - 1;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const uint32_t m = B.mvsad[(size_t)item * g.nmb + mb];
    const int dx = (int)(int8_t)(m & 255u), dy = (int)(int8_t)((m >> 8) & 255u); /* the full search's vector, integer samples */
    const int x0 = mx * 16 + dx, y0 = my * 16 + dy;
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    /* the lane's share of the macroblock: row y, samples x4 .. x4 + 3 (display edges replicate, spec section 1) */
    const int y = lane >> 2, x4 = (lane & 3) * 4;
    uint32_t cur4;
    {
        const uint8_t *row = cur + (size_t)min(my * 16 + y, ph - 1) * cstride;
        const int px = mx * 16 + x4;
        cur4 = (uint32_t)row[min(px, pw - 1)] | ((uint32_t)row[min(px + 1, pw - 1)] << 8) | ((uint32_t)row[min(px + 2, pw - 1)] << 16) |
               ((uint32_t)row[min(px + 3, pw - 1)] << 24);
    }
    /* window: 22 rows x 24 columns, one dword of four clamped samples per lane and round */
    for (int i = lane; i < SP_WR * (SP_WS / 4); i += 64) {
        const int r = i / (SP_WS / 4), c = (i - r * (SP_WS / 4)) * 4;
        const uint8_t *row = ref + (size_t)clampi(y0 - 3 + r, 0, g.ch - 1) * rstride;
        const int xb = x0 - 3 + c;
        L.win[i] = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                   ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
    }
    __syncthreads();
    const uint8_t *w8 = (const uint8_t *)L.win;
    /* lanes as (row of three, column): 17-column planes and 18-column planes */
    const int r17 = lane / 17, c17 = lane - r17 * 17; /* lanes 0 .. 50 */
    const int r18 = lane / 18, c18 = lane - r18 * 18; /* lanes 0 .. 53 */
    /* b1[r][c] = taps over window columns c .. c + 5 of row r (c = 0 is the half sample left of the block's first column) */
    if (lane < 51) {
        for (int r = r17; r < SP_WR; r += 3) {
            const uint8_t *p = w8 + r * SP_WS + c17;
            L.b1[r * SP_B1S + c17] = (int16_t)sp_tap6(p[0], p[1], p[2], p[3], p[4], p[5]);
        }
    }
    /* h[rr][xx]: vertical taps over window rows rr .. rr + 5, window column xx + 2 */
    if (lane < 54) {
        uint8_t *h8 = (uint8_t *)L.hp;
        for (int rr = r18; rr < 17; rr += 3) {
            const uint8_t *p = w8 + rr * SP_WS + c18 + 2;
            const int v = sp_tap6(p[0], p[SP_WS], p[2 * SP_WS], p[3 * SP_WS], p[4 * SP_WS], p[5 * SP_WS]);
            h8[rr * SP_PS + c18] = (uint8_t)venc_clip_u8((v + 16) >> 5);
        }
    }
    __syncthreads();
    if (lane < 51) {
        uint8_t *b8 = (uint8_t *)L.bp, *j8 = (uint8_t *)L.jp;
        for (int yy = r17; yy < 18; yy += 3) /* b rows -1 .. 16 = b1 rows 2 .. 19 */
            b8[yy * SP_PS + c17] = (uint8_t)venc_clip_u8(((int)L.b1[(yy + 2) * SP_B1S + c17] + 16) >> 5);
        for (int rr = r17; rr < 17; rr += 3) {
            const int16_t *p = L.b1 + rr * SP_B1S + c17;
            const int v = sp_tap6(p[0], p[SP_B1S], p[2 * SP_B1S], p[3 * SP_B1S], p[4 * SP_B1S], p[5 * SP_B1S]);
            j8[rr * SP_PS + c17] = (uint8_t)venc_clip_u8((v + 512) >> 10);
        }
    }
    __syncthreads();
    /* the two stages (spec 2a): centre (cx, cy) in quarter units from the full search's vector */
    int cx = 0, cy = 0;
    uint32_t csad = m >> 16;
    for (int stage = 0; stage < g.subpel && stage < 2; stage++) {
        const int step = stage == 0 ? 2 : 1;
        uint32_t best = csad << 4;
        int bx = cx, by = cy, idx = 0;
        for (int ey = -1; ey <= 1; ey++)
            for (int ex = -1; ex <= 1; ex++) {
                if (!ex && !ey) continue;
                idx++;
                const int qx = cx + step * ex, qy = cy + step * ey;
                const uint32_t sad = sp_wave_sum(__builtin_amdgcn_sad_u8(sp_pred4(L, qx, qy, y, x4), cur4, 0u));
                const uint32_t key = (sad << 4) | (uint32_t)idx;
                if (key < best) best = key, bx = qx, by = qy;
            }
        cx = bx, cy = by, csad = best >> 4;
    }
    if (valid) {
        if (lane == 0) B.mvq[(size_t)item * g.nmb + mb] = (csad << 16) | (((uint32_t)(4 * dy + cy) & 255u) << 8) | ((uint32_t)(4 * dx + cx) & 255u);
        ((uint32_t *)(B.pred_luma + ((size_t)item * g.nmb + mb) * 256))[lane] = sp_pred4(L, cx, cy, y, x4);
    }
}
