/*
 * legacy/me_tile.hip -- k_me_tile, the predecessor of k_me_pool
 * TEST-ONLY: compiled into libvenc_hip_legacy.so (-DVENC_LEGACY_KERNELS), the on-GPU oracles of tests/test_gpu_parity.py;
 * the product library does not contain a line of this file.  Included by venc_me2.hip.
 */
/* SAD of one macroblock at window position (col0 + cx, row0 + cy), 8 pixels per lane of a half-wave */
DEVFN uint32_t me2_coop_sad(const uint32_t *s_win, int col0, int row0, int lane, uint32_t c0, uint32_t c1, int cx, int cy) {
    const int l32 = lane & 31;
    const int row = l32 >> 1, hx = (l32 & 1) * 8;
    const int boff = col0 + cx + hx;
    const uint32_t *p = s_win + (row0 + cy + row) * T2_WD + (boff >> 2);
    const uint32_t sh = (uint32_t)(boff & 3);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    uint32_t a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d1, d0, sh), c0, 0u);
    a = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d2, d1, sh), c1, a);
    return half_sum32(a, lane);
}

__global__ __launch_bounds__(512) void k_me_tile(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ __attribute__((aligned(16))) uint32_t s_win[T2_WR * T2_WD];
    __shared__ __attribute__((aligned(16))) uint2 s_h8[T2_WR * T2_HG];
    __shared__ __attribute__((aligned(16))) uint4 s_v8[T2_VR * T2_HG];
    __shared__ uint32_t s_list[T2_MBX * T2_MBY][SEA_LIST];
    __shared__ uint32_t s_cnt[T2_MBX * T2_MBY];
    __shared__ __attribute__((aligned(4))) uint16_t s_rank[VENC_ME_CANDS + 1];

    const int item = blockIdx.z, tid = threadIdx.x;
    const int x0 = blockIdx.x * T2_MBX * 16, y0 = (S.row0 + blockIdx.y * T2_MBY) * 16;
    int rstride, cstride, pw, ph;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);
    /* every global load whose result is needed late is issued here, ahead of the window staging, so
     * that no memory round trip sits alone on the critical path: the tie-break ranks (written to LDS
     * with the window), this lane's current pixels, and last frame's vector of its macroblock */
    const uint32_t *rank32 = (const uint32_t *)B.me_rank; /* 1089 u16 = 545 dwords (table padded by the launcher) */
    const uint32_t rk0 = rank32[tid], rk1 = tid + 512 < 545 ? rank32[tid + 512] : 0u;
    const uint16_t *rank = s_rank;

    if (tid < T2_MBX * T2_MBY) s_cnt[tid] = 0;
    const int lane = tid & 63, l32 = lane & 31, hbase = lane & 32;
    const int hw = tid >> 5;
    const int mbx = hw & (T2_MBX - 1), mby = hw >> 3;
    const int mx = blockIdx.x * T2_MBX + mbx, my = S.row0 + blockIdx.y * T2_MBY + mby;
    const bool mb_valid = mx < g.mbw && my < S.row1;
    const int col0 = mbx * 16, row0 = mby * 16;
    /* this lane's 8 pixels of its macroblock (row l32/2, half l32&1): fetched now, used after the
     * block sums are built, so the load latency hides behind P0-P2 */
    uint32_t c0, c1;
    {
        const int cy_ = min(y0 + row0 + (l32 >> 1), g.h - 1), cx_ = x0 + col0 + (l32 & 1) * 8;
        const uint8_t *row = cur + (size_t)cy_ * cstride;
        if (((g.w & 7) == 0) && cx_ + 8 <= g.w) {
            const uint2 v = *(const uint2 *)(row + cx_);
            c0 = v.x, c1 = v.y;
        } else {
            c0 = (uint32_t)row[min(cx_, g.w - 1)] | ((uint32_t)row[min(cx_ + 1, g.w - 1)] << 8) |
                 ((uint32_t)row[min(cx_ + 2, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 3, g.w - 1)] << 24);
            c1 = (uint32_t)row[min(cx_ + 4, g.w - 1)] | ((uint32_t)row[min(cx_ + 5, g.w - 1)] << 8) |
                 ((uint32_t)row[min(cx_ + 6, g.w - 1)] << 16) | ((uint32_t)row[min(cx_ + 7, g.w - 1)] << 24);
        }
    }
    const size_t mbidx = (size_t)item * g.nmb + (size_t)min(my, g.mbh - 1) * g.mbw + min(mx, g.mbw - 1);
    const uint32_t prev = B.mvsad_prev[mbidx];
    ME_STAMP_INIT;
    /* ---- P0: reference window, rows y0-16 .. y0+47, columns x0-16 .. x0+143 (+8 zero bytes) ---- */
    const bool xin = (x0 - 16 >= 0) && (x0 + T2_MBX * 16 + 16 <= g.cw);
    if (xin) {
        /* 640 16-byte pieces for 512 threads: both loads of a thread are in flight before the first LDS write */
        const int i0 = tid, i1 = tid + 512;
        const int ra = i0 / 10, qa = i0 - ra * 10, rb = i1 / 10, qb = i1 - rb * 10;
        const bool two = i1 < T2_WR * 10;
        const uint4 va = *(const uint4 *)(ref + (size_t)clampi(y0 - 16 + ra, 0, g.ch - 1) * rstride + x0 - 16 + 16 * qa);
        uint4 vb = make_uint4(0, 0, 0, 0);
        if (two) vb = *(const uint4 *)(ref + (size_t)clampi(y0 - 16 + rb, 0, g.ch - 1) * rstride + x0 - 16 + 16 * qb);
        uint32_t *da = s_win + ra * T2_WD + 4 * qa;
        *(uint2 *)da = make_uint2(va.x, va.y); /* row stride is 8-byte aligned only */
        *(uint2 *)(da + 2) = make_uint2(va.z, va.w);
        if (two) {
            uint32_t *db = s_win + rb * T2_WD + 4 * qb;
            *(uint2 *)db = make_uint2(vb.x, vb.y);
            *(uint2 *)(db + 2) = make_uint2(vb.z, vb.w);
        }
    } else {
        for (int i = tid; i < T2_WR * 40; i += 512) {
            const int r = i / 40, d = i - r * 40;
            const int ry = clampi(y0 - 16 + r, 0, g.ch - 1);
            const int xb = x0 - 16 + 4 * d;
            const uint8_t *row = ref + (size_t)ry * rstride;
            s_win[r * T2_WD + d] = (uint32_t)row[clampi(xb, 0, g.cw - 1)] | ((uint32_t)row[clampi(xb + 1, 0, g.cw - 1)] << 8) |
                                   ((uint32_t)row[clampi(xb + 2, 0, g.cw - 1)] << 16) |
                                   ((uint32_t)row[clampi(xb + 3, 0, g.cw - 1)] << 24);
        }
    }
    if (tid < T2_WR) {
        s_win[tid * T2_WD + 40] = 0;
        s_win[tid * T2_WD + 41] = 0;
    }
    ((uint32_t *)s_rank)[tid] = rk0;
    if (tid + 512 < 545) ((uint32_t *)s_rank)[tid + 512] = rk1;
    __syncthreads();
    ME_STAMP(0);
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
    }
    __syncthreads();
    ME_STAMP(1);
    if (S.me_stop == 2) return;
    /* ---- P2: vertical 8-sums, stored as vertical pairs V(x,y) = S8(x,y) | S8(x,y+8) << 16 so that the
     * bound of a candidate needs two dword reads (left and right 8x8 columns) and no repacking.
     * 7 segments of 8 rows x 39 column groups; a thread slides down 16 rows of block sums. */
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
    ME_STAMP(2);
    if (S.me_stop == 3) return;
    /* ---- per macroblock: one half-wave ---- */
    ME_STAMP(3);
    uint32_t part = __builtin_amdgcn_sad_u8(c0, 0u, 0u);
    part = __builtin_amdgcn_sad_u8(c1, 0u, part);
    part += (uint32_t)__shfl_xor((int)part, 2);
    part += (uint32_t)__shfl_xor((int)part, 4);
    part += (uint32_t)__shfl_xor((int)part, 8);
    const uint32_t q00 = (uint32_t)__shfl((int)part, hbase | 0), q01 = (uint32_t)__shfl((int)part, hbase | 1);
    const uint32_t q10 = (uint32_t)__shfl((int)part, hbase | 16), q11 = (uint32_t)__shfl((int)part, hbase | 17);
    const uint32_t cleft = q00 | (q10 << 16), cright = q01 | (q11 << 16); /* (top, bottom) sums of the left / right 8x8 columns */

    /* ---- P3: lower bounds.  Lane = window column cx = l32, one register per vertical displacement:
     * LB = |V_left - (TL,BL)| + |V_right - (TR,BR)|, two v_sad_u16 on packed u16 pairs */
    const uint32_t *v8 = (const uint32_t *)s_v8;
    uint32_t lb[33];
    {
        const uint32_t *colp = v8 + row0 * T2_SU + col0 + l32;
#pragma unroll
        for (int cy = 0; cy < 33; cy++)
            lb[cy] = __builtin_amdgcn_sad_u16(colp[cy * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[cy * T2_SU], cleft, 0u));
    }
    uint32_t lbx0, lbx1 = 0xffffffffu; /* column cx = 32: lane l32 owns cy = l32, lane 0 also cy = 32 */
    {
        const uint32_t *colp = v8 + row0 * T2_SU + col0 + 32;
        lbx0 = __builtin_amdgcn_sad_u16(colp[l32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[l32 * T2_SU], cleft, 0u));
        if (l32 == 0) lbx1 = __builtin_amdgcn_sad_u16(colp[32 * T2_SU + 8], cright, __builtin_amdgcn_sad_u16(colp[32 * T2_SU], cleft, 0u));
    }

    ME_STAMP(4);
    /* ---- P4: starting points, survivors, evaluation ---- */
    const int pcx = (int)(int8_t)(prev & 255u) + 16, pcy = (int)(int8_t)((prev >> 8) & 255u) + 16;
    /* third starting point: the candidate with the smallest lower bound */
    uint32_t lmin = (lbx0 << 12) | (uint32_t)(l32 * 33 + 32);
    lmin = min(lmin, lbx1 == 0xffffffffu ? 0xffffffffu : ((lbx1 << 12) | (uint32_t)(32 * 33 + 32)));
#pragma unroll
    for (int cy = 0; cy < 33; cy++) lmin = min(lmin, (lb[cy] << 12) | (uint32_t)(cy * 33 + l32));
    const int ai = (int)(half_min32(lmin, lane) & 4095u);
    if (S.me_stop == 4) {
        if (l32 == 0 && mb_valid) B.mvsad[mbidx] = (uint32_t)ai << 16;
        return;
    }
    const int acy = ai / 33, acx = ai - acy * 33;
    const uint32_t sad_z = me2_coop_sad(s_win, col0, row0, lane, c0, c1, 16, 16);
    const uint32_t sad_p = me2_coop_sad(s_win, col0, row0, lane, c0, c1, pcx, pcy);
    const uint32_t sad_a = me2_coop_sad(s_win, col0, row0, lane, c0, c1, acx, acy);
    uint32_t best = min(min((sad_z << 11) | rank[16 * 33 + 16], (sad_p << 11) | rank[pcy * 33 + pcx]), (sad_a << 11) | rank[ai]);
    {
        const uint32_t thr = min(min(sad_z, sad_p), sad_a);
#pragma unroll
        for (int cy = 0; cy < 33; cy++) {
            if (lb[cy] <= thr) {
                const uint32_t pos = atomicAdd(&s_cnt[hw], 1u);
                if (pos < SEA_LIST) s_list[hw][pos] = (lb[cy] << 12) | (uint32_t)(cy * 33 + l32);
            }
        }
        if (lbx0 <= thr) {
            const uint32_t pos = atomicAdd(&s_cnt[hw], 1u);
            if (pos < SEA_LIST) s_list[hw][pos] = (lbx0 << 12) | (uint32_t)(l32 * 33 + 32);
        }
        if (lbx1 <= thr) {
            const uint32_t pos = atomicAdd(&s_cnt[hw], 1u);
            if (pos < SEA_LIST) s_list[hw][pos] = (lbx1 << 12) | (uint32_t)(32 * 33 + 32);
        }
    }
    ME_STAMP(5);
    __syncthreads();
    if (S.me_stop == 5) {
        if (l32 == 0 && mb_valid) B.mvsad[mbidx] = best << 16;
        return;
    }
    const uint32_t cnt = s_cnt[hw];
    const bool overflow = cnt > SEA_LIST;
    const uint32_t cnt_eff = overflow ? 0u : cnt;
    const uint32_t n = max(cnt_eff, (uint32_t)__shfl((int)cnt_eff, hbase ^ 32));
    uint32_t bsad = best >> 11; /* threshold follows the SADs directly, the rank lookup stays off the critical path */
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t e = k < cnt_eff ? s_list[hw][k] : 0xffffffffu;
        const int ci = (int)(e & 4095u);
        const bool need = k < cnt_eff && (e >> 12) <= bsad && ci != ai && ci != 16 * 33 + 16 && ci != pcy * 33 + pcx;
        if (__ballot(need) == 0ull) continue; /* neither macroblock of this wave needs the evaluation */
        const int cy = need ? ci / 33 : 16, cx = need ? ci - (ci / 33) * 33 : 16;
        const uint32_t sad = me2_coop_sad(s_win, col0, row0, lane, c0, c1, cx, cy);
        bsad = min(bsad, sad);
        best = min(best, (sad << 11) | rank[cy * 33 + cx]);
    }
    ME_STAMP(6);
    if (l32 == 0 && mb_valid) {
        const uint32_t cand = B.me_cand[best & 2047u];
        const int dx = (int)(cand & 255u) - 16, dy = (int)(cand >> 8) - 16;
        B.mvsad[mbidx] = ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
        B.me_flag[mbidx] = overflow ? 1 : 0;
        if (overflow) B.fix_list[(size_t)item * g.nmb + atomicAdd(&B.fix_count[item], 1u)] = (uint32_t)(my * g.mbw + mx);
        B.me_cnt[mbidx] = (uint8_t)min(cnt, 255u); /* survivors of the bound test (introspection) */
    }
}

