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

#define RS_TW 128 /* k_refsum tile: output columns per block */
#define RS_TH 32  /* output rows per block */
#define RS_PD 37  /* pixel row stride in dwords (36 used) */
#define RS_HG 34  /* groups of 4 columns carrying horizontal 8-sums: columns [0, 136) */

/* P8 plane geometry: positions x in [-16, cw], y in [-16, ch + 8] */
DEVFN int p8_width(const venc_geom &g) { return g.cw + 17; }
DEVFN int p8_height(const venc_geom &g) { return g.ch + 25; }

DEVFN uint2 pk_add_u16x4(uint2 a, uint2 b) { /* four independent u16 lanes, no carries across (sums < 2^16) */
    return make_uint2(a.x + b.x, a.y + b.y);
}
DEVFN uint2 pk_sub_u16x4(uint2 a, uint2 b) { /* every lane of a >= the same lane of b */
    return make_uint2(a.x - b.x, a.y - b.y);
}

/* Block sums of the frame just reconstructed.  Tile = 128 x 32 plane positions.
 *   A  pixels (edge-clamped) -> LDS as dwords
 *   B  horizontal 8-sums with v_qsad_pk_u16_u8 against zero: one instruction yields the four
 *      sliding 4-byte sums of a dword pair; two of them added give four 8-sums (packed u16x4)
 *   C  vertical 8-sums by a sliding window down 4-row segments (packed u16x4 adds)
 *   D  pair S8(x) with S8(x+8) and store 16 bytes per lane */
__global__ __launch_bounds__(256) void k_refsum(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_px[(RS_TH + 7) * RS_PD];
    __shared__ uint2 s_h8[(RS_TH + 7) * RS_HG];
    __shared__ uint2 s_s8[RS_TH * RS_HG];
    const int item = blockIdx.z, tid = threadIdx.x;
    const int ox = blockIdx.x * RS_TW - 16, oy = blockIdx.y * RS_TH - 16; /* picture coords of plane position (0,0) of the tile */
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.parity, item, 0, rstride);
    /* A: pixels x in [ox, ox + 144), y in [oy, oy + 39) */
    const bool xin = ox >= 0 && ox + 144 <= g.cw;
    for (int i = tid; i < (RS_TH + 7) * 36; i += 256) {
        const int r = i / 36, d = i - r * 36;
        const uint8_t *row = ref + (size_t)clampi(oy + r, 0, g.ch - 1) * rstride;
        const int xb = ox + 4 * d;
        uint32_t v;
        if (xin)
            v = *(const uint32_t *)(row + xb);
        else
            v = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
        s_px[r * RS_PD + d] = v;
    }
    __syncthreads();
    /* B: H8 for 34 groups x 39 rows */
    for (int i = tid; i < (RS_TH + 7) * RS_HG; i += 256) {
        const int r = i / RS_HG, gq = i - r * RS_HG;
        const uint32_t *p = s_px + r * RS_PD + gq;
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        const unsigned long long h4a = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d1 << 32) | d0, 0u, 0ull);
        const unsigned long long h4b = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d2 << 32) | d1, 0u, 0ull);
        s_h8[i] = pk_add_u16x4(make_uint2((uint32_t)h4a, (uint32_t)(h4a >> 32)), make_uint2((uint32_t)h4b, (uint32_t)(h4b >> 32)));
    }
    __syncthreads();
    /* C: S8 for 32 rows: 8 segments of 4 rows x 34 groups */
    for (int i = tid; i < 8 * RS_HG; i += 256) {
        const int seg = i / RS_HG, gq = i - seg * RS_HG;
        const uint2 *col = s_h8 + (seg * 4) * RS_HG + gq;
        uint2 acc = col[0];
#pragma unroll
        for (int k = 1; k < 8; k++) acc = pk_add_u16x4(acc, col[k * RS_HG]);
        s_s8[(seg * 4) * RS_HG + gq] = acc;
#pragma unroll
        for (int k = 1; k < 4; k++) {
            acc = pk_sub_u16x4(pk_add_u16x4(acc, col[(k + 7) * RS_HG]), col[(k - 1) * RS_HG]);
            s_s8[(seg * 4 + k) * RS_HG + gq] = acc;
        }
    }
    __syncthreads();
    /* D: P8(x) = S8(x) | S8(x+8) << 16, four positions (one group) per lane */
    const int W = p8_width(g), H = p8_height(g);
    uint32_t *dst = B.refsum + (size_t)item * B.refsum_item;
    for (int i = tid; i < RS_TH * 32; i += 256) {
        const int r = i >> 5, gq = i & 31;
        const int X = blockIdx.x * RS_TW + 4 * gq, Y = blockIdx.y * RS_TH + r;
        if (X < W && Y < H) { /* the plane rows are padded to a multiple of 16 dwords: a 4-dword store never leaves the row */
            const uint2 a = s_s8[r * RS_HG + gq], b = s_s8[r * RS_HG + gq + 2];
            uint4 o;
            o.x = (a.x & 0xffffu) | (b.x << 16);
            o.y = (a.x >> 16) | (b.x & 0xffff0000u);
            o.z = (a.y & 0xffffu) | (b.y << 16);
            o.w = (a.y >> 16) | (b.y & 0xffff0000u);
            *(uint4 *)(dst + (size_t)Y * B.refsum_stride + X) = o;
        }
    }
}

#define SEA_PW (ME_TILE * 16 + 20) /* packed-sum window row: 145 dwords used, padded to 16-byte multiples */

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
    __shared__ __attribute__((aligned(16))) uint32_t s_win[48 * ME_WDW + 4]; /* +4: the realigning fetch may touch one dword past the last row */
    __shared__ __attribute__((aligned(16))) uint32_t s_cur[16 * ME_TILE * 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_p8[41 * SEA_PW];
    __shared__ uint32_t s_list[ME_TILE][SEA_LIST];
    __shared__ uint32_t s_cnt[ME_TILE];

    const int item = blockIdx.z, my = blockIdx.y, tx = blockIdx.x, tid = threadIdx.x;
    const int x0 = tx * ME_TILE * 16, y0 = my * 16;
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    int cstride, pw, ph;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);

    if (tid < ME_TILE) s_cnt[tid] = 0;
    const uint16_t *rank = B.me_rank; /* read straight from global memory: only evaluated candidates need it */

    /* ---- staging, 16 bytes per lane wherever the tile lies inside the frame ---- */
    const bool xin = (x0 - 16 >= 0) && (x0 + ME_TILE * 16 + 16 <= g.cw);
    if (xin) { /* 48 rows x 10 uint4 */
        for (int i = tid; i < 480; i += 256) {
            const int r = i / 10, q = i - r * 10;
            const int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
            const uint4 v = *(const uint4 *)(ref + (size_t)ry * rstride + x0 - 16 + 16 * q);
            *(uint4 *)(s_win + r * ME_WDW + 4 * q) = v;
        }
    } else {
        for (int i = tid; i < 48 * ME_WDW; i += 256) {
            int r = i / ME_WDW, d = i - r * ME_WDW;
            int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
            int xb = x0 - 16 + 4 * d;
            const uint8_t *row = ref + (size_t)ry * rstride;
            s_win[i] = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                       ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) | ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
        }
    }
    const bool cur_fast = ((g.w & 15) == 0) && (x0 + ME_TILE * 16 <= g.w);
    if (cur_fast) { /* 16 rows x 8 uint4 */
        if (tid < 128) {
            const int r = tid >> 3, q = tid & 7;
            const uint4 v = *(const uint4 *)(cur + (size_t)min(y0 + r, g.h - 1) * cstride + x0 + 16 * q);
            *(uint4 *)(s_cur + r * (ME_TILE * 4) + 4 * q) = v;
        }
    } else {
        for (int i = tid; i < 16 * ME_TILE * 4; i += 256) {
            int r = i / (ME_TILE * 4), d = i - r * (ME_TILE * 4);
            int sy = min(y0 + r, g.h - 1);
            int xb = x0 + 4 * d;
            const uint8_t *row = cur + (size_t)sy * cstride;
            s_cur[i] = (uint32_t)row[min(xb, g.w - 1)] | ((uint32_t)row[min(xb + 1, g.w - 1)] << 8) |
                       ((uint32_t)row[min(xb + 2, g.w - 1)] << 16) | ((uint32_t)row[min(xb + 3, g.w - 1)] << 24);
        }
    }
    /* packed 8x8 block sums of the reference: plane row (y0 + r), plane column x0 + c (plane origin = picture (-16,-16)) */
    {
        const uint32_t *p8 = B.refsum + (size_t)item * B.refsum_item;
        if (x0 + SEA_PW <= B.refsum_stride) { /* 41 rows x 37 uint4 */
            for (int i = tid; i < 41 * 37; i += 256) {
                const int r = i / 37, q = i - r * 37;
                const uint4 v = *(const uint4 *)(p8 + (size_t)(y0 + r) * B.refsum_stride + x0 + 4 * q);
                *(uint4 *)(s_p8 + r * SEA_PW + 4 * q) = v;
            }
        } else {
            const int W = p8_width(g);
            for (int i = tid; i < 41 * 145; i += 256) {
                const int r = i / 145, c = i - r * 145;
                s_p8[r * SEA_PW + c] = p8[(size_t)(y0 + r) * B.refsum_stride + min(x0 + c, W - 1)];
            }
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
    const uint32_t prev = B.mvsad_prev[(size_t)item * g.nmb + my * g.mbw + min(mx, g.mbw - 1)];
    const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
    uint32_t best = (me_coop_sad(s_win, mbi, l32, c0, c1, 16, 16) << 11) | rank[16 * 33 + 16];
    {
        const uint32_t k2 = (me_coop_sad(s_win, mbi, l32, c0, c1, pcx, pcy) << 11) | rank[pcy * 33 + pcx];
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
            const uint32_t key = (sad << 11) | rank[cy * 33 + cx];
            best = min(best, key);
        }
    } else {
        best = me_dense_half(s_win, s_cur, rank, mbi, l32);
    }
    if (l32 == 0 && mx < g.mbw) {
        uint32_t cand = B.me_cand[best & 2047u];
        int dx = (int)(cand & 255u) - 16, dy = (int)(cand >> 8) - 16;
        B.mvsad[(size_t)item * g.nmb + my * g.mbw + mx] =
            ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
    }
}
