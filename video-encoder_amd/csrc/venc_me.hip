/*
 * venc_me.hip -- exact full-search motion estimation by successive elimination
 * ("levers that keep the identical argmin").
 *
 * The argmin of spec/CODEC.md section 2 -- min over all 1089 displacements of
 * (SAD << 11) | rank -- is found without computing most SADs:
 *
 *   k_refsum  once per reconstructed frame: P8(x,y) = S8(x,y) | S8(x+8,y) << 16 for every
 *             block position of the edge-replicated reference, S8 = sum of the 8x8 block at
 *             (x,y).  2 packed u16 per pixel position.
 *   k_me      per macroblock: LB(c) = sum over the four 8x8 sub-blocks of
 *             |S8cur_k - S8ref_k(c)|  <=  SAD(c)   (triangle inequality), 2 v_sad_u16 per
 *             candidate instead of 64 v_sad_u8.  SAD is evaluated only for candidates whose
 *             lower bound does not exceed the best SAD found so far, starting from (0,0) and the
 *             macroblock's vector in the previous frame.  A candidate that is skipped has
 *             SAD >= LB > best SAD, so it can neither win nor tie: the result is bit-identical
 *             to the exhaustive search (k_me_dense), which remains the in-kernel fallback for
 *             macroblocks where the bound does not discriminate (flat content).
 *
 * Measured on the synthetic 1080p clip (profiles/): a handful of SAD evaluations per
 * macroblock instead of 1089.
 */

#define RS_TW 64 /* k_refsum tile: outputs per block, x */
#define RS_TH 32

/* P8 plane geometry: positions x in [-16, cw], y in [-16, ch + 8] */
DEVFN int p8_width(const venc_geom &g) { return g.cw + 17; }
DEVFN int p8_height(const venc_geom &g) { return g.ch + 25; }

__global__ __launch_bounds__(256) void k_refsum(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint8_t s_px[RS_TH + 7][RS_TW + 8 + 7 + 1]; /* 39 x 80 */
    __shared__ uint16_t s_h8[RS_TH + 7][RS_TW + 8];        /* horizontal 8-sums, 72 per row */
    const int item = blockIdx.z, tid = threadIdx.x;
    const int ox = blockIdx.x * RS_TW - 16, oy = blockIdx.y * RS_TH - 16; /* picture coords of output (0,0) */
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.parity, item, 0, rstride); /* the frame just reconstructed */
    /* pixels needed: x in [ox, ox + 64 + 8 + 7), y in [oy, oy + 32 + 7) */
    for (int i = tid; i < (RS_TH + 7) * (RS_TW + 15); i += 256) {
        const int r = i / (RS_TW + 15), c = i - r * (RS_TW + 15);
        s_px[r][c] = ref[(size_t)clampi(oy + r, 0, g.ch - 1) * rstride + clampi(ox + c, 0, g.cw - 1)];
    }
    __syncthreads();
    for (int i = tid; i < (RS_TH + 7) * (RS_TW + 8); i += 256) {
        const int r = i / (RS_TW + 8), c = i - r * (RS_TW + 8);
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s += s_px[r][c + k];
        s_h8[r][c] = (uint16_t)s;
    }
    __syncthreads();
    const int pw = B.refsum_stride, W = p8_width(g), H = p8_height(g);
    uint32_t *dst = B.refsum + (size_t)item * B.refsum_item;
    for (int i = tid; i < RS_TH * RS_TW; i += 256) {
        const int r = i / RS_TW, c = i - r * RS_TW;
        const int X = blockIdx.x * RS_TW + c, Y = blockIdx.y * RS_TH + r; /* plane coords */
        if (X < W && Y < H) {
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a += s_h8[r + k][c];
                b += s_h8[r + k][c + 8];
            }
            dst[(size_t)Y * pw + X] = a | (b << 16);
        }
    }
}

#define SEA_PW (ME_TILE * 16 + 17) /* packed-sum window row: 145 dwords */
#define SEA_LIST 64               /* survivors evaluated one by one; more -> dense fallback */

/* SAD of the macroblock against the reference block at window position (cx, cy), computed by
 * the 32 lanes of a half-wave (8 pixels each); the result is identical in all 32 lanes. */
DEVFN uint32_t me_coop_sad(const uint32_t *s_win, int mbi, int l32, uint32_t c0, uint32_t c1, int cx, int cy) {
    const int row = l32 >> 1, hx = (l32 & 1) * 8;
    const int boff = mbi * 16 + cx + hx;
    const uint32_t *p = s_win + (cy + row) * ME_WDW + (boff >> 2);
    const uint32_t sh = (uint32_t)(boff & 3);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    uint32_t a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d1, d0, sh), c0, 0u);
    a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d2, d1, sh), c1, a);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) a += (uint32_t)__shfl_xor((int)a, m);
    return a;
}

__global__ __launch_bounds__(256) void k_me(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_win[48 * ME_WDW + 4]; /* +4: the realigning fetch may touch one dword past the last row */
    __shared__ uint32_t s_cur[16 * ME_TILE * 4];
    __shared__ uint32_t s_p8[41 * SEA_PW];
    __shared__ uint16_t s_rank[VENC_ME_CANDS];
    __shared__ uint32_t s_list[ME_TILE][SEA_LIST];
    __shared__ uint32_t s_cnt[ME_TILE];

    const int item = blockIdx.z, my = blockIdx.y, tx = blockIdx.x, tid = threadIdx.x;
    const int x0 = tx * ME_TILE * 16, y0 = my * 16;
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.parity ^ 1, item, 0, rstride);
    int cstride, pw, ph;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);

    for (int i = tid; i < VENC_ME_CANDS; i += 256) s_rank[i] = B.me_rank[i];
    if (tid < ME_TILE) s_cnt[tid] = 0;

    const bool xin = (x0 - 16 >= 0) && (x0 + ME_TILE * 16 + 16 <= g.cw);
    for (int i = tid; i < 48 * ME_WDW; i += 256) {
        int r = i / ME_WDW, d = i - r * ME_WDW;
        int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
        int xb = x0 - 16 + 4 * d;
        const uint8_t *row = ref + (size_t)ry * rstride;
        uint32_t v;
        if (xin) {
            v = *(const uint32_t *)(row + xb);
        } else {
            v = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
        }
        s_win[i] = v;
    }
    const bool cur_fast = ((g.w & 3) == 0) && (x0 + ME_TILE * 16 <= g.w);
    for (int i = tid; i < 16 * ME_TILE * 4; i += 256) {
        int r = i / (ME_TILE * 4), d = i - r * (ME_TILE * 4);
        int sy = min(y0 + r, g.h - 1);
        int xb = x0 + 4 * d;
        const uint8_t *row = cur + (size_t)sy * cstride;
        uint32_t v;
        if (cur_fast) {
            v = *(const uint32_t *)(row + xb);
        } else {
            v = (uint32_t)row[min(xb, g.w - 1)] | ((uint32_t)row[min(xb + 1, g.w - 1)] << 8) |
                ((uint32_t)row[min(xb + 2, g.w - 1)] << 16) | ((uint32_t)row[min(xb + 3, g.w - 1)] << 24);
        }
        s_cur[i] = v;
    }
    /* packed 8x8 block sums of the reference: plane row (y0 + r), plane column x0 + c  (plane origin = picture (-16,-16)) */
    {
        const uint32_t *p8 = B.refsum + (size_t)item * B.refsum_item;
        const int W = p8_width(g);
        for (int i = tid; i < 41 * SEA_PW; i += 256) {
            const int r = i / SEA_PW, c = i - r * SEA_PW;
            const int X = min(x0 + c, W - 1); /* columns beyond the plane belong to macroblocks beyond the frame */
            s_p8[i] = p8[(size_t)(y0 + r) * B.refsum_stride + X];
        }
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31, hbase = lane & 32;
    const int mbi = wave * 2 + half;
    const int mx = tx * ME_TILE + mbi;

    /* this lane's 8 pixels of the current macroblock (row l32/2, half l32&1) */
    const uint32_t c0 = s_cur[(l32 >> 1) * (ME_TILE * 4) + mbi * 4 + (l32 & 1) * 2];
    const uint32_t c1 = s_cur[(l32 >> 1) * (ME_TILE * 4) + mbi * 4 + (l32 & 1) * 2 + 1];
    /* 8x8 sums of the current macroblock: lane's 8 pixels belong to quadrant (row >= 8, half) */
    uint32_t part = __builtin_amdgcn_sad_u8(c0, 0u, 0u);
    part = __builtin_amdgcn_sad_u8(c1, 0u, part);
    /* sum over the 8 rows of a quadrant: lanes with the same (l32 & 1) and the same (l32 >> 4) */
    part += (uint32_t)__shfl_xor((int)part, 2);
    part += (uint32_t)__shfl_xor((int)part, 4);
    part += (uint32_t)__shfl_xor((int)part, 8);
    const uint32_t q00 = (uint32_t)__shfl((int)part, hbase | 0), q01 = (uint32_t)__shfl((int)part, hbase | 1);
    const uint32_t q10 = (uint32_t)__shfl((int)part, hbase | 16), q11 = (uint32_t)__shfl((int)part, hbase | 17);
    const uint32_t ctop = q00 | (q01 << 16), cbot = q10 | (q11 << 16);

    /* lower bounds: lane = window column cx = l32 (dx = cx - 16), register array over cy; column 32 separately */
    uint32_t lb[33];
    {
        const uint32_t *col = s_p8 + mbi * 16 + l32;
#pragma unroll
        for (int cy = 0; cy < 33; cy++) {
            uint32_t a = __builtin_amdgcn_sad_u16(col[cy * SEA_PW], ctop, 0u);
            lb[cy] = __builtin_amdgcn_sad_u16(col[(cy + 8) * SEA_PW], cbot, a);
        }
    }
    uint32_t lbx0, lbx1 = 0xffffffffu; /* column cx = 32: lane l32 owns cy = l32, lane 0 also cy = 32 */
    {
        const uint32_t *col = s_p8 + mbi * 16 + 32;
        uint32_t a = __builtin_amdgcn_sad_u16(col[l32 * SEA_PW], ctop, 0u);
        lbx0 = __builtin_amdgcn_sad_u16(col[(l32 + 8) * SEA_PW], cbot, a);
        if (l32 == 0) {
            a = __builtin_amdgcn_sad_u16(col[32 * SEA_PW], ctop, 0u);
            lbx1 = __builtin_amdgcn_sad_u16(col[40 * SEA_PW], cbot, a);
        }
    }

    /* starting points: zero vector and the macroblock's vector in the previous frame */
    const uint32_t prev = B.mvsad[(size_t)item * g.nmb + my * g.mbw + min(mx, g.mbw - 1)];
    const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
    uint32_t best = (me_coop_sad(s_win, mbi, l32, c0, c1, 16, 16) << 11) | s_rank[16 * 33 + 16];
    {
        const uint32_t k2 = (me_coop_sad(s_win, mbi, l32, c0, c1, pcx, pcy) << 11) | s_rank[pcy * 33 + pcx];
        best = min(best, k2);
    }
    /* collect the candidates whose bound does not exceed the best SAD so far */
    {
        const uint32_t thr = best >> 11;
#pragma unroll
        for (int cy = 0; cy < 33; cy++) {
            if (lb[cy] <= thr) {
                const uint32_t pos = atomicAdd(&s_cnt[mbi], 1u);
                if (pos < SEA_LIST) s_list[mbi][pos] = (lb[cy] << 12) | (uint32_t)(cy * 33 + l32);
            }
        }
        if (lbx0 <= thr) {
            const uint32_t pos = atomicAdd(&s_cnt[mbi], 1u);
            if (pos < SEA_LIST) s_list[mbi][pos] = (lbx0 << 12) | (uint32_t)(l32 * 33 + 32);
        }
        if (lbx1 <= thr) {
            const uint32_t pos = atomicAdd(&s_cnt[mbi], 1u);
            if (pos < SEA_LIST) s_list[mbi][pos] = (lbx1 << 12) | (uint32_t)(32 * 33 + 32);
        }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt[mbi];
    const uint32_t cnt_other = (uint32_t)__shfl((int)cnt, hbase ^ 32);
    if (cnt <= SEA_LIST && cnt_other <= SEA_LIST) {
        const uint32_t n = max(cnt, cnt_other);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t e = k < cnt ? s_list[mbi][k] : 0xffffffffu;
            const uint32_t lbv = e >> 12;
            const int ci = (int)(e & 4095u);
            /* both halves run the evaluation (shuffles need the whole wave); a half without work repeats (16,16) */
            const bool need = k < cnt && lbv <= (best >> 11);
            const int cy = need ? ci / 33 : 16, cx = need ? ci - (ci / 33) * 33 : 16;
            const uint32_t sad = me_coop_sad(s_win, mbi, l32, c0, c1, cx, cy);
            const uint32_t key = (sad << 11) | s_rank[cy * 33 + cx];
            best = min(best, key);
        }
    } else {
        best = me_dense_half(s_win, s_cur, s_rank, mbi, l32);
    }
    if (l32 == 0 && mx < g.mbw) {
        uint32_t cand = B.me_cand[best & 2047u];
        int dx = (int)(cand & 255u) - 16, dy = (int)(cand >> 8) - 16;
        B.mvsad[(size_t)item * g.nmb + my * g.mbw + mx] =
            ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
    }
}
