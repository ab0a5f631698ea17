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

/* ================================================================== k_hpel
 * The half-sample planes b, h, j of a whole reference picture (H.264 8.4.2.2.1; what libx264 keeps per reference frame
 * as its "hpel" planes), once per P step: k_subpel then LOADS the 17 x 17 neighbourhood of a macroblock's vector from
 * them instead of filtering a 22 x 22 window itself (per macroblock that cost as many filter outputs as this kernel
 * spends per picture, at byte granularity in LDS: 170 us of a 220 us launch at 1080p x 10).
 * One wave = a strip of 256 samples x HP_ROWS rows; a lane owns four adjacent samples (one aligned dword, coalesced),
 * its neighbours' dwords arrive by lane shuffles; rows slide through registers (six rows of samples and of horizontal
 * sums).  The filters run on PACKED 16-bit pairs (v_pk_*): a row's dword splits into its even and odd samples, a pair
 * of outputs two samples apart takes its six taps from six such registers; b1 and h1 fit 16 bits (-2550 .. 10710); only
 * j -- the six-tap of six b1 -- needs 32.  Positions whose taps leave the picture use clamped coordinates, as the
 * standard does; k_subpel reads the planes only where the needed samples lie inside the picture. */
#ifndef HP_ROWS
#define HP_ROWS 16
#endif
typedef short hp_s2 __attribute__((ext_vector_type(2)));
DEVFN hp_s2 hp_even(uint32_t w) { return __builtin_bit_cast(hp_s2, w & 0x00ff00ffu); }
DEVFN hp_s2 hp_odd(uint32_t w) { return __builtin_bit_cast(hp_s2, (w >> 8) & 0x00ff00ffu); }
DEVFN hp_s2 hp_tap6(hp_s2 a, hp_s2 b, hp_s2 c, hp_s2 d, hp_s2 e, hp_s2 f) { return (a + f) + (c + d) * (short)20 - (b + e) * (short)5; }
/* clip((t + 16) >> 5) of two packed pairs (samples 0, 2 and 1, 3) -> four bytes */
DEVFN uint32_t hp_bytes(hp_s2 t02, hp_s2 t13) {
    const hp_s2 z = {0, 0}, m = {255, 255}, r = {16, 16};
    const hp_s2 v02 = __builtin_elementwise_min(__builtin_elementwise_max((t02 + r) >> 5, z), m);
    const hp_s2 v13 = __builtin_elementwise_min(__builtin_elementwise_max((t13 + r) >> 5, z), m);
    return __builtin_bit_cast(uint32_t, v02) | (__builtin_bit_cast(uint32_t, v13) << 8);
}

__global__ __launch_bounds__(256) void k_hpel(venc_geom g, venc_bufs B, venc_step S) {
    const int item = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = ((int)blockIdx.x * 64 + lane) * 4;
    const int yb = ((int)blockIdx.y * 4 + wave) * HP_ROWS;
    if (yb >= g.ch) return;
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const size_t plane = (size_t)g.cw * g.ch;
    uint8_t *hb = B.hpel + (size_t)item * 3 * plane, *hh = hb + plane, *hj = hh + plane;
    const bool in = x < g.cw; /* the coded width is a multiple of 16: a lane's dword is inside or outside as a whole */
    const int xc = in ? x : g.cw - 4;
    hp_s2 pe[6], po[6]; /* even / odd samples of the lane's dword, last six rows */
    int b1w[6][4];      /* horizontal sums of the last six rows */
    uint32_t bdw[4];    /* b of the last four rows, as bytes */
#pragma unroll
    for (int k = 0; k < HP_ROWS + 5; k++) {
        const int rc = clampi(yb - 2 + k, 0, g.ch - 1);
        const uint8_t *row = ref + (size_t)rc * rstride;
        const uint32_t d1 = *(const uint32_t *)(row + xc);
        uint32_t d0 = (uint32_t)__shfl_up((int)d1, 1), d2 = (uint32_t)__shfl_down((int)d1, 1);
        if (lane == 0) d0 = x > 0 ? *(const uint32_t *)(row + x - 4) : (d1 & 255u) * 0x01010101u;
        if (lane == 63 || x + 4 >= g.cw) d2 = x + 4 < g.cw ? *(const uint32_t *)(row + x + 4) : (d1 >> 24) * 0x01010101u;
        const uint32_t w2 = __builtin_amdgcn_alignbyte(d1, d0, 2), w6 = __builtin_amdgcn_alignbyte(d2, d1, 2); /* samples x-2..x+1, x+2..x+5 */
        const hp_s2 e2 = hp_even(w2), o2 = hp_odd(w2), e4 = hp_even(d1), o4 = hp_odd(d1), e6 = hp_even(w6), o6 = hp_odd(w6), e8 = hp_even(d2);
        const hp_s2 t02 = hp_tap6(e2, o2, e4, o4, e6, o6); /* b1 at x, x + 2 */
        const hp_s2 t13 = hp_tap6(o2, e4, o4, e6, o6, e8); /* b1 at x + 1, x + 3 */
        const int s = k % 6;
        pe[s] = e4, po[s] = o4;
        b1w[s][0] = t02.x, b1w[s][1] = t13.x, b1w[s][2] = t02.y, b1w[s][3] = t13.y;
        bdw[k % 4] = hp_bytes(t02, t13);
        if (k >= 5) {
            const int y = yb + k - 5; /* window rows y - 2 .. y + 3 = iterations k - 5 .. k */
            const int a = (k - 5) % 6, b = (k - 4) % 6, c = (k - 3) % 6, d = (k - 2) % 6, e = (k - 1) % 6, f = k % 6;
            const uint32_t hdw = hp_bytes(hp_tap6(pe[a], pe[b], pe[c], pe[d], pe[e], pe[f]), hp_tap6(po[a], po[b], po[c], po[d], po[e], po[f]));
            uint32_t jdw = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = (b1w[a][i] + b1w[f][i]) + 20 * (b1w[c][i] + b1w[d][i]) - 5 * (b1w[b][i] + b1w[e][i]);
                /* clamped BEFORE the shift: written as clip((t + 512) >> 10) per pair of samples, this compiler (ROCm 7.2, gfx950)
                 * forms v_ashr_pk_u8_i32, whose upper destination half keeps what the register held -- bits of an earlier sum
                 * then leak into the third sample's byte (tests/test_gpu_subpel.py, the planes test, caught it) */
                jdw |= ((uint32_t)min(max(t + 512, 0), (255 << 10) | 1023) >> 10) << (8 * i);
            }
            if (in && y < g.ch) {
                const size_t o = (size_t)y * g.cw + x;
                *(uint32_t *)(hb + o) = bdw[(k - 3) % 4]; /* row y = iteration k - 3 */
                *(uint32_t *)(hh + o) = hdw;
                *(uint32_t *)(hj + o) = jdw;
            }
        }
    }
}

/* ================================================================== k_subpel */
#define SP_PS 24   /* row stride of a plane piece in LDS, bytes: 18 samples + up to 3 of alignment slack */
#define SP_PD (18 * SP_PS / 4 + 1) /* dwords per plane piece (+1: the unaligned four-sample read takes two dwords) */
#define SP_WS 24   /* border path: window row stride in bytes (columns -3 .. 20, 22 used) */
#define SP_WR 22   /* border path: window rows -3 .. 18 */
#define SP_B1S 18  /* border path: row stride of b1 in int16 */
#ifndef SP_K
#define SP_K 4     /* macroblocks per wave, one after the other */
#endif

/* per wave: the macroblock's corner of the doubled sample grid as four plane pieces with one layout -- row = integer row
 * -1 .. 16 (h, j: -1 .. 15), byte column = integer column -1 .. 16 (b, j: -1 .. 15) + sh */
struct sp_lds {
    uint32_t pl[4][SP_PD];               /* 0: integer samples G, 1: b, 2: h, 3: j */
    uint32_t cur[SP_K][64];              /* the wave's current macroblocks: row y, dword x4 / 4 at [4 y + x4 / 4] */
    uint32_t pred[SP_K][64];             /* their predictions at the refined vectors, same layout (the luma transform's input) */
    uint32_t win[SP_WR * SP_WS / 4];     /* border path only */
    int16_t b1[SP_WR * SP_B1S];          /* border path only */
};

DEVFN int sp_tap6(int a, int b, int c, int d, int e, int f) { return a + f - 5 * (b + e) + 20 * (c + d); }

/* four samples of the doubled grid for the lane's row y and columns x4 .. x4 + 3: half-sample coordinates (hx + 2 x, hy + 2 y)
 * relative to the displaced block; the plane follows from the parities of (hx, hy), which are wave-uniform */
DEVFN uint32_t sp_grid4(const sp_lds &L, int sh, int hx, int hy, int y, int x4) {
    const uint8_t *base = (const uint8_t *)L.pl[(hx & 1) | ((hy & 1) << 1)];
    const int off = ((hy >> 1) + y + 1) * SP_PS + (hx >> 1) + x4 + 1 + sh;
    const uint32_t *d = (const uint32_t *)(base + (off & ~3));
    return __builtin_amdgcn_alignbyte(d[1], d[0], (uint32_t)(off & 3));
}
/* prediction at (qx, qy) quarter units from the displaced block, four samples at once: both even -- a sample of the doubled
 * grid; one odd -- the average of the two grid samples either side along that axis; both odd -- of the two corners of the
 * grid cell that are b / h samples (odd coordinate sum) */
DEVFN uint32_t sp_pred4(const sp_lds &L, int sh, int qx, int qy, int y, int x4) {
    const int ox = qx & 1, oy = qy & 1;
    const int hx0 = (qx - ox) >> 1, hy0 = (qy - oy) >> 1;
    if (!ox && !oy) return sp_grid4(L, sh, hx0, hy0, y, x4);
    uint32_t a, b;
    if (ox && !oy) a = sp_grid4(L, sh, hx0, hy0, y, x4), b = sp_grid4(L, sh, hx0 + 1, hy0, y, x4);
    else if (!ox) a = sp_grid4(L, sh, hx0, hy0, y, x4), b = sp_grid4(L, sh, hx0, hy0 + 1, y, x4);
    else if ((hx0 + hy0) & 1) a = sp_grid4(L, sh, hx0, hy0, y, x4), b = sp_grid4(L, sh, hx0 + 1, hy0 + 1, y, x4);
    else a = sp_grid4(L, sh, hx0 + 1, hy0, y, x4), b = sp_grid4(L, sh, hx0, hy0 + 1, y, x4);
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); /* (a + b + 1) >> 1 in every byte */
}

/* One refinement stage, ALL EIGHT neighbours at once: lane group c = lane >> 3 takes neighbour c (index c + 1 of the
 * stage), its lanes s = lane & 7 two rows each (2 s, 2 s + 1: eight dwords of prediction against eight of the current
 * macroblock, both from LDS).  A candidate's prediction is the rounded average of two grid samples A and B (A = B on the
 * doubled grid itself), chosen without a branch: with (hx0, hy0) the lower corner of the candidate's grid cell, (ox, oy)
 * the odd bits of its quarter position and par the parity of hx0 + hy0,
 *   A = (hx0 + [ox & oy & !par], hy0),  B = (hx0 + [ox & (!oy | par)], hy0 + oy)
 * -- the two samples either side along the odd axis, or for an odd / odd position the cell's two b / h corners.
 * Eight lanes sum their SADs on the DPP network (quad butterflies, half-row mirror); the minimum key (SAD << 4 | index)
 * over the groups comes back through a row mirror and four lane reads.  Returns the stage's best key below `best`
 * (else `best`).  QUARTER = false: positions on the doubled grid only (A alone). */
template <bool QUARTER>
DEVFN uint32_t sp_stage(const sp_lds &L, const uint32_t *curk, int sh, int cx, int cy, int step, int lane, uint32_t best) {
    const int c = lane >> 3, s2 = (lane & 7) * 2;
    const int kk = c < 4 ? c : c + 1; /* raster index of the neighbour in the 3 x 3 stencil, centre skipped */
    const int ey = kk / 3 - 1, ex = kk - (kk / 3) * 3 - 1;
    const int qx = cx + step * ex, qy = cy + step * ey;
    const int ox = qx & 1, oy = qy & 1;
    const int hx0 = (qx - ox) >> 1, hy0 = (qy - oy) >> 1, par = (hx0 + hy0) & 1;
    const int ax = hx0 + (ox & oy & (par ^ 1)), ay = hy0;
    const int bx = hx0 + (ox & ((oy ^ 1) | par)), by = hy0 + oy;
    /* byte address of the sample for row 0, column 0 in the wave's plane pieces */
    const int offa = (((ax & 1) | ((ay & 1) << 1)) * SP_PD * 4) + ((ay >> 1) + 1) * SP_PS + (ax >> 1) + 1 + sh;
    const int offb = (((bx & 1) | ((by & 1) << 1)) * SP_PD * 4) + ((by >> 1) + 1) * SP_PS + (bx >> 1) + 1 + sh;
    const uint8_t *base = (const uint8_t *)L.pl;
    uint32_t sad = 0;
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int xq = 0; xq < 4; xq++) {
            const int oa = offa + (s2 + rr) * SP_PS + 4 * xq;
            const uint32_t *da = (const uint32_t *)(base + (oa & ~3));
            uint32_t pr = __builtin_amdgcn_alignbyte(da[1], da[0], (uint32_t)(oa & 3));
            if (QUARTER) {
                const int ob = offb + (s2 + rr) * SP_PS + 4 * xq;
                const uint32_t *db = (const uint32_t *)(base + (ob & ~3));
                const uint32_t pb = __builtin_amdgcn_alignbyte(db[1], db[0], (uint32_t)(ob & 3));
                pr = (pr | pb) - (((pr ^ pb) >> 1) & 0x7f7f7f7fu);
            }
            sad = __builtin_amdgcn_sad_u8(pr, curk[(s2 + rr) * 4 + xq], sad);
        }
    sad += dpp_fetch<0xB1, 0xf>(sad, 0u);  /* quad_perm [1, 0, 3, 2] */
    sad += dpp_fetch<0x4E, 0xf>(sad, 0u);  /* quad_perm [2, 3, 0, 1] */
    sad += dpp_fetch<0x141, 0xf>(sad, 0u); /* row_half_mirror: the other quad of the eight */
    uint32_t key = (sad << 4) | (uint32_t)(c + 1);
    key = min(key, dpp_fetch<0x140, 0xf>(key, 0xffffffffu)); /* row_mirror: the row's other group */
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, 0), k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, 16);
    const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)key, 32), k3 = (uint32_t)__builtin_amdgcn_readlane((int)key, 48);
    return min(min(best, min(k0, k1)), min(k2, k3));
}

/* border path: the plane pieces of a macroblock whose neighbourhood leaves the picture, filtered here from a 22 x 22 window of
 * clamped reference samples (window rows / columns -3 .. 18; b1 = unclipped horizontal sums, 22 rows x 17 columns) */
DEVFN void sp_build_border(sp_lds &L, const venc_geom &g, const uint8_t *ref, int rstride, int x0, int y0, int lane) {
    for (int i = lane; i < SP_WR * (SP_WS / 4); i += 64) {
        const int r = i / (SP_WS / 4), c = (i - r * (SP_WS / 4)) * 4;
        const uint8_t *row = ref + (size_t)clampi(y0 - 3 + r, 0, g.ch - 1) * rstride;
        const int xb = x0 - 3 + c;
        L.win[i] = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                   ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the wave's own LDS stores are done (LDS ops of a wave retire in order) */
    __builtin_amdgcn_wave_barrier();
    const uint8_t *w8 = (const uint8_t *)L.win;
    uint8_t *g8 = (uint8_t *)L.pl[0], *b8 = (uint8_t *)L.pl[1], *h8 = (uint8_t *)L.pl[2], *j8 = (uint8_t *)L.pl[3];
    const int r17 = lane / 17, c17 = lane - r17 * 17; /* lanes 0 .. 50: (row of three, column) of a 17-column plane */
    const int r18 = lane / 18, c18 = lane - r18 * 18; /* lanes 0 .. 53: of an 18-column plane */
    if (lane < 51)
        for (int r = r17; r < SP_WR; r += 3) { /* b1[r][c] = taps over window columns c .. c + 5 */
            const uint8_t *p = w8 + r * SP_WS + c17;
            L.b1[r * SP_B1S + c17] = (int16_t)sp_tap6(p[0], p[1], p[2], p[3], p[4], p[5]);
        }
    if (lane < 54) {
        for (int rr = r18; rr < 18; rr += 3) g8[rr * SP_PS + c18] = w8[(rr + 2) * SP_WS + c18 + 2]; /* rows / columns -1 .. 16 */
        for (int rr = r18; rr < 17; rr += 3) { /* h: vertical taps over window rows rr .. rr + 5 */
            const uint8_t *p = w8 + rr * SP_WS + c18 + 2;
            h8[rr * SP_PS + c18] = (uint8_t)venc_clip_u8((sp_tap6(p[0], p[SP_WS], p[2 * SP_WS], p[3 * SP_WS], p[4 * SP_WS], p[5 * SP_WS]) + 16) >> 5);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane < 51) {
        for (int yy = r17; yy < 18; yy += 3) /* b rows -1 .. 16 = b1 rows 2 .. 19 */
            b8[yy * SP_PS + c17] = (uint8_t)venc_clip_u8(((int)L.b1[(yy + 2) * SP_B1S + c17] + 16) >> 5);
        for (int rr = r17; rr < 17; rr += 3) {
            const int16_t *p = L.b1 + rr * SP_B1S + c17;
            j8[rr * SP_PS + c17] = (uint8_t)venc_clip_u8((sp_tap6(p[0], p[SP_B1S], p[2 * SP_B1S], p[3 * SP_B1S], p[4 * SP_B1S], p[5 * SP_B1S]) + 512) >> 10);
        }
    }
}

/* One wave per macroblock (SP_K of them, one after the other), four waves per block, no block-wide synchronisation: a
 * wave's LDS piece is its own.  Around the full search's vector the wave holds the macroblock's corner of the DOUBLED
 * sample grid of H.264 8.4.2.2.1 as four plane pieces (integer samples, b, h, j; rows / columns -1 .. 16): loaded as
 * aligned dwords from the reference picture and k_hpel's planes when the neighbourhood lies inside the picture (all but
 * the picture's rim), filtered in place otherwise.  Every candidate of both stages -- the eight half-sample neighbours of
 * the vector, then the eight quarter-sample neighbours of that stage's winner -- reads its prediction from the pieces: a
 * sample of the doubled grid, or the rounded average of two (Table 8-12), four samples per lane (lane = row, quarter of
 * a row), SAD by v_sad_u8 and a wave sum.  The winner (minimum of (SAD << 4) | index, index 0 = the stage's centre) goes
 * to B.mvq in QUARTER-sample units -- the transform and entropy kernels of the step are handed that array as their
 * vectors; the search's own integer vectors stay where the next step's search reads its start points -- and its
 * prediction to a 256-byte block of B.pred_luma, which the luma transform (k_inter_fused) reads instead of the
 * reference plane. */
__global__ __launch_bounds__(256) void k_subpel(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ sp_lds s_l[4];
    const int item = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    sp_lds &L = s_l[wave];
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    const size_t plane = (size_t)g.cw * g.ch;
    const uint8_t *hpl = B.hpel + (size_t)item * 3 * plane;
    const int y = lane >> 2, x4 = (lane & 3) * 4; /* the lane's share of a macroblock: row y, samples x4 .. x4 + 3 */
    /* the interior path loads 4 planes x 18 rows x 6 aligned dwords per macroblock, seven per lane: where each of them comes
     * from and goes to depends on the lane alone -- worked out once per wave */
    const uint8_t *src[7];
    int dst[7];
#pragma unroll
    for (int it = 0; it < 7; it++) {
        const int i = min(lane + 64 * it, 4 * 18 * 6 - 1), p = i / 108, rem = i - p * 108, r = rem / 6, dw = rem - r * 6;
        src[it] = (p == 0 ? ref : hpl + (size_t)(p - 1) * plane) + (size_t)r * g.cw + 4 * dw;
        dst[it] = p * SP_PD + r * (SP_PS / 4) + dw;
    }
    /* the records and the current samples of the wave's macroblocks: loaded up front, so that a macroblock's round starts
     * with ONE level of loads (its plane pieces) instead of three in a row (record -> addresses -> pieces) */
    uint32_t mrec[SP_K];
#pragma unroll
    for (int kmb = 0; kmb < SP_K; kmb++) {
        const int mb_raw = S.mb0 + ((int)blockIdx.x * 4 + wave) * SP_K + kmb, mbc = min(mb_raw, S.mb1 - 1);
        mrec[kmb] = B.mvsad[(size_t)item * g.nmb + mbc];
        const int px = (mbc % g.mbw) * 16 + x4, py = (mbc / g.mbw) * 16 + y;
        uint32_t cur4; /* display edges replicate (spec section 1) */
        if ((g.w & 3) == 0 && px + 4 <= pw && py < ph)
            cur4 = *(const uint32_t *)(cur + (size_t)py * cstride + px);
        else {
            const uint8_t *row = cur + (size_t)min(py, ph - 1) * cstride;
            cur4 = (uint32_t)row[min(px, pw - 1)] | ((uint32_t)row[min(px + 1, pw - 1)] << 8) | ((uint32_t)row[min(px + 2, pw - 1)] << 16) |
                   ((uint32_t)row[min(px + 3, pw - 1)] << 24);
        }
        L.cur[kmb][lane] = cur4;
    }
#pragma unroll
    for (int kmb = 0; kmb < SP_K; kmb++) {
        const int mb_raw = S.mb0 + ((int)blockIdx.x * 4 + wave) * SP_K + kmb;
        if (mb_raw >= S.mb1) break; /* (wave-uniform) */
        const int mb = mb_raw, mx = mb % g.mbw, my = mb / g.mbw;
        /* (one macroblock per wave: its record is uniform -- said aloud, so that the stages' control flow stays on the
         * scalar unit) */
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)mrec[kmb]);
        const int dx = (int)(int8_t)(m & 255u), dy = (int)(int8_t)((m >> 8) & 255u); /* the full search's vector, integer samples */
        const int x0 = mx * 16 + dx, y0 = my * 16 + dy;
        int sh = 0;
        if (x0 >= 1 && x0 + 17 <= g.cw && y0 >= 1 && y0 + 17 <= g.ch) {
            /* rows y0 - 1 .. y0 + 16, columns from the aligned dword that holds x0 - 1: 6 dwords cover sh + 18 samples.
             * (h and j have no row 17 of their own: the picture's row y0 + 16 exists, its values are never read) */
            sh = (x0 - 1) & 3;
            const size_t o0 = (size_t)(y0 - 1) * g.cw + ((x0 - 1) & ~3);
            uint32_t *pl = &L.pl[0][0];
#pragma unroll
            for (int it = 0; it < 7; it++) pl[dst[it]] = *(const uint32_t *)(src[it] + o0); /* (the last round's spare lanes repeat the last dword) */
        } else
            sp_build_border(L, g, ref, rstride, x0, y0, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        /* the two stages (spec 2a): centre (cx, cy) in quarter units from the full search's vector */
        int cx = 0, cy = 0;
        uint32_t csad = m >> 16;
        for (int stage = 0; stage < g.subpel && stage < 2; stage++) {
            const int step = stage == 0 ? 2 : 1;
            const uint32_t best = stage == 0 ? sp_stage<false>(L, L.cur[kmb], sh, cx, cy, step, lane, csad << 4) : sp_stage<true>(L, L.cur[kmb], sh, cx, cy, step, lane, csad << 4);
            const int idx = (int)(best & 15u); /* 0: the centre stays */
            if (idx) {
                const int kk = idx <= 4 ? idx - 1 : idx;
                cx += step * (kk - (kk / 3) * 3 - 1), cy += step * (kk / 3 - 1);
            }
            csad = best >> 4;
        }
        if (lane == 0) B.mvq[(size_t)item * g.nmb + mb] = (csad << 16) | (((uint32_t)(4 * dy + cy) & 255u) << 8) | ((uint32_t)(4 * dx + cx) & 255u);
        L.pred[kmb][lane] = sp_pred4(L, sh, cx, cy, y, x4);
        __builtin_amdgcn_s_waitcnt(0xc07f); /* the pieces are rebuilt for the wave's next macroblock */
        __builtin_amdgcn_wave_barrier();
    }
    /* the luma transform of the wave's SP_K macroblocks, all at once: 16 lanes per macroblock, one 4x4 block each, current
     * and predicted rows from the wave's LDS copies (luma_block_code, venc_inter2.hip: residual ... reconstruction, levels) */
    static_assert(SP_K == 4, "four macroblocks x sixteen blocks = the lanes of a wave");
    {
        const int mbk = lane >> 4, blk = lane & 15;
        const int mb_raw = S.mb0 + ((int)blockIdx.x * 4 + wave) * SP_K + mbk;
        const bool valid = mb_raw < S.mb1;
        const int mb = valid ? mb_raw : S.mb1 - 1;
        venc_qparams ql, qc;
        step_qparams(g, B, S, item, ql, qc);
        uint32_t cq[4], pq[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cq[r] = L.cur[mbk][(blk_y(blk) * 4 + r) * 4 + blk_x(blk)];
            pq[r] = L.pred[mbk][(blk_y(blk) * 4 + r) * 4 + blk_x(blk)];
        }
        luma_block_code(g, B, S, item, lane, mb, valid, cq, pq, ql);
    }
}
