/*
 * venc_deblock2.hip -- k_deblock_ls: the in-loop deblocking filter as a LOCK-STEP diagonal wavefront (H.264 8.7;
 * spec/CODEC.md section 5a; behind av_encode.c:970 it is libx264's loop filter).
 *
 * What bounded k_deblock (venc_deblock.hip, kept as the test-only reference build's kernel): one wave per macroblock row, rows
 * handing over through progress flags -- and through MEMORY: a row waited for the flag of the row above, then loaded the four rows
 * above its strip from global memory, filtered, stored, drained its stores and only then published its own flag.  Every one of
 * the 30 + 2 x 8 strip steps of a 1080p slice carried two memory round trips and ran at the pace of nine waves sharing one CU:
 * 8.2 us per strip where a wave alone needs 3 -- 330 us per launch whatever the item count.
 *
 * Here the rows of a slice advance TOGETHER, one macroblock per step, each row two macroblocks behind the row above it
 * (macroblock (x, y) needs (x - 1, y) and (x + 1, y - 1): both were finished one step earlier).  Sixteen lanes own a row --
 * lane = line across the edge being filtered: a luma row / column of the macroblock, or a chroma line of either plane -- so a
 * wave carries four rows and a 1080p slice of nine rows is three waves.  Nobody waits for a flag: the step ends in a block
 * barrier.  Nothing is handed over through memory:
 *   - a row's macroblock x and its left neighbour x - 1 sit in a two-slot LDS tile of the row (slot = x & 1);
 *   - the four bottom rows of a finished macroblock are written straight into the tile of the row BELOW (as its rows -4 .. -1),
 *     which stores them after its own top-edge filter -- the row above never stores them, so no store order to keep;
 *   - samples and records of macroblock x + 2 are fetched while x is filtered (registers, two steps ahead), finished rows
 *     leave as whole 16-byte pieces a step after their macroblock (its last three columns wait for the next left edge).
 * A launch is mbw + 2 (rows - 1) + 2 steps: latency, not throughput -- and the same latency for every item count the CUs can
 * hold (one small block per slice and item).  What a step costs is the length of ONE wave's instruction stream (a wave alone on
 * its SIMD issues an instruction every ~8.5 cycles), so the stream is split between two kinds of waves
 * that run side by side and meet at the step's barrier:
 *   FILTER waves   the macroblock's phases on the LDS tile, and the hand-down of the bottom rows -- nothing else;
 *   HELPER waves   (same rows, one step ahead / behind) everything that is not filtering: samples and records of macroblock
 *                  x + 3 from memory into registers, those of x + 1 from registers into the tile (a ring of four slots per row),
 *                  the boundary strengths of x + 1 (eight words per macroblock in LDS, read by the filter wave as two 16-byte
 *                  pieces), and the finished macroblock x - 2 and the rows above x - 1 out to memory.
 * With everything in the filter waves (the first form of this kernel) a step was ~350 instructions long and the launch took as
 * long as the strip kernel's (340 us: 138 steps x 2.4 us against 46 strip steps x 7 us).
 *
 * Slices taller than the block's rows (4 x 16) run in passes; the first row of a later pass takes its rows above from memory.
 */

#define DL_MAX_WAVES 16 /* filter + helper waves of a block */
#define DL_ROWS (2 * DL_MAX_WAVES) /* rows of a pass: four per filter wave */

struct dl_row {                /* LDS state of one macroblock row */
    uint32_t y[4][20 * 4];     /* slot (x & 3), rows -4 .. 15 of 16 bytes */
    uint32_t c[4][2][12 * 2];  /* slot, plane, rows -4 .. 7 of 8 bytes */
    uint32_t info[2][12];      /* slot (x & 1): the macroblock's record (8 dwords) + its vector word (+ padding to 16-byte pieces) */
    uint32_t above[12];        /* record + vector word of the macroblock above the one whose strengths are being computed */
    uint32_t bs[2][12];        /* slot (x & 1): the strengths of the macroblock's eight phases, four segments per word; [8]: bit ph = phase ph has one */
};

struct dl_fetch { /* what a helper lane holds for the macroblock three steps ahead */
    uint4 y;      /* luma row k */
    uint2 c;      /* chroma: plane k >> 3, row k & 7 */
    uint4 rec;    /* lanes 0, 1: the halves of the record; 3, 4: the same of the macroblock above */
    uint32_t mv;  /* lane 2: the vector word; 5: that of the macroblock above (a register of its own: sharing rec.x, the two loads of a
                     fetch wrote one register under different masks and the second waited for the first to land) */
    uint4 ya;     /* first row of a later pass only: luma row k - 4 of the rows above (k < 4) */
    uint2 ca;     /* ... chroma */
};

/* one phase of one macroblock (PH 0-3: vertical edges, 4-7: horizontal edges); it runs if one of the wave's four macroblocks has a
 * strength on it.  s / sl: the slots of the macroblock and of its left neighbour; bs4: the four segment strengths of the phase */
template <bool CHROMA, int PH>
DEVFN void dl_phase(dl_row &T, int s, int sl, int k, bool act, uint32_t bs4, uint32_t wave_phases, int alpha, int beta, uint32_t tcw) {
    constexpr int e = PH & 3;
    if (CHROMA && (e & 1)) return;
    if (!(wave_phases & (1u << PH))) return; /* (scalar) none of the wave's four macroblocks has a strength on this phase */
    const int kk = CHROMA ? (k & 7) : k;
    uint32_t *const cs = CHROMA ? T.c[s][k >> 3] : T.y[s];
    uint32_t *const cl = CHROMA ? T.c[sl][k >> 3] : T.y[sl];
    constexpr int rd = CHROMA ? 2 : 4; /* dwords per tile row */
    const bool go = act && bs4 != 0u;
    const int lbs = (int)((bs4 >> (8 * (CHROMA ? kk >> 1 : kk >> 2))) & 255u);
    const int tc0 = (int)((tcw >> (8 * max(lbs - 1, 0))) & 255u);
    constexpr int ep = CHROMA ? e >> 1 : e; /* edge position in dwords (vertical) / in groups of four rows (horizontal) */
    if (__ballot(go && lbs == 4) == 0ull) {
        /* no strong edge in the wave (P pictures without intra macroblocks: always): the branch-free normal filter, six samples in,
         * four out */
        if (go) {
            if (PH < 4) {
                uint32_t *q = cs + (4 + kk) * rd + ep;
                uint32_t *p = ep ? q - 1 : cl + (4 + kk) * rd + rd - 1;
                const uint32_t w0 = *p, w1 = *q;
                int p1 = (int)((w0 >> 16) & 255u), p0 = (int)(w0 >> 24), q0 = (int)(w1 & 255u), q1 = (int)((w1 >> 8) & 255u);
                venc_dbk_line_normal(CHROMA, (int)((w0 >> 8) & 255u), &p1, &p0, &q0, &q1, (int)((w1 >> 16) & 255u), lbs, alpha, beta, tc0);
                *p = (w0 & 0xffffu) | ((uint32_t)p1 << 16) | ((uint32_t)p0 << 24);
                *q = (w1 & 0xffff0000u) | (uint32_t)q0 | ((uint32_t)q1 << 8);
            } else {
                uint8_t *col = (uint8_t *)cs + (4 * ep) * (4 * rd) + kk;
                int p2 = CHROMA ? 0 : col[1 * 4 * rd], p1 = col[2 * 4 * rd], p0 = col[3 * 4 * rd], q0 = col[4 * 4 * rd], q1 = col[5 * 4 * rd], q2 = CHROMA ? 0 : col[6 * 4 * rd];
                venc_dbk_line_normal(CHROMA, p2, &p1, &p0, &q0, &q1, q2, lbs, alpha, beta, tc0);
                if (!CHROMA) col[2 * 4 * rd] = (uint8_t)p1;
                col[3 * 4 * rd] = (uint8_t)p0;
                col[4 * 4 * rd] = (uint8_t)q0;
                if (!CHROMA) col[5 * 4 * rd] = (uint8_t)q1;
            }
        }
    } else if (go) {
        int sm[8];
        if (PH < 4) { /* vertical edge: p3 .. p0 and q0 .. q3 are two dwords of row kk (p in the left neighbour's slot for edge 0) */
            uint32_t *q = cs + (4 + kk) * rd + ep;
            uint32_t *p = ep ? q - 1 : cl + (4 + kk) * rd + rd - 1;
            const uint32_t w0 = *p, w1 = *q;
            sm[0] = (int)(w0 & 255u), sm[1] = (int)((w0 >> 8) & 255u), sm[2] = (int)((w0 >> 16) & 255u), sm[3] = (int)(w0 >> 24);
            sm[4] = (int)(w1 & 255u), sm[5] = (int)((w1 >> 8) & 255u), sm[6] = (int)((w1 >> 16) & 255u), sm[7] = (int)(w1 >> 24);
            dbk_line(sm, lbs, alpha, beta, tc0, CHROMA);
            *p = (uint32_t)sm[0] | ((uint32_t)sm[1] << 8) | ((uint32_t)sm[2] << 16) | ((uint32_t)sm[3] << 24);
            *q = (uint32_t)sm[4] | ((uint32_t)sm[5] << 8) | ((uint32_t)sm[6] << 16) | ((uint32_t)sm[7] << 24);
        } else { /* horizontal edge: eight rows of column kk, tile rows 4 ep .. 4 ep + 7 */
            uint8_t *col = (uint8_t *)cs + (4 * ep) * (4 * rd) + kk;
#pragma unroll
            for (int i = 0; i < 8; i++) sm[i] = col[i * 4 * rd];
            dbk_line(sm, lbs, alpha, beta, tc0, CHROMA);
#pragma unroll
            for (int i = 1; i < 7; i++) col[i * 4 * rd] = (uint8_t)sm[i];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}
template <bool CHROMA>
DEVFN void dl_phases(dl_row &T, int s, int sl, int k, bool act, const uint4 &b0, const uint4 &b1, uint32_t wp, int alpha, int beta, uint32_t tcw) {
    dl_phase<CHROMA, 0>(T, s, sl, k, act, b0.x, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 1>(T, s, sl, k, act, b0.y, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 2>(T, s, sl, k, act, b0.z, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 3>(T, s, sl, k, act, b0.w, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 4>(T, s, sl, k, act, b1.x, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 5>(T, s, sl, k, act, b1.y, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 6>(T, s, sl, k, act, b1.z, wp, alpha, beta, tcw);
    dl_phase<CHROMA, 7>(T, s, sl, k, act, b1.w, wp, alpha, beta, tcw);
}

template <int PLANES>
__global__ __launch_bounds__(64 * DL_MAX_WAVES) void k_deblock_ls(venc_geom g, venc_bufs B, venc_step S, int diag) {
    extern __shared__ __attribute__((aligned(16))) dl_row s_rows[]; /* one per row of a pass (the launcher sizes it: 4 x filter waves) */
    constexpr bool DO_Y = (PLANES & 1) != 0, DO_C = (PLANES & 2) != 0;
    const int item = blockIdx.z, sl = S.slice0 + (int)blockIdx.x, tid = threadIdx.x;
    const int nf = blockDim.x >> 7, wv = tid >> 6, lane = tid & 63; /* nf filter waves, then nf helper waves */
    const bool helper = wv >= nf;
    const int wf = helper ? wv - nf : wv;
    const int k = lane & 15, rw = lane >> 4;
    const int R = 4 * nf; /* rows of a pass */
    const int r0 = B.slice_first_row[sl], r1 = B.slice_first_row[sl + 1], nrows = r1 - r0;
    const venc_mbinfo *infos = B.mbinfo + (size_t)item * g.nmb;
    const uint32_t *mvs = B.mvsad + (size_t)item * g.nmb;
    int ystride, cstride;
    uint8_t *py = rec_plane(B, g, S.parity, item, 0, ystride);
    uint8_t *pc0 = rec_plane(B, g, S.parity, item, 1, cstride), *pc1 = rec_plane(B, g, S.parity, item, 2, cstride);
    uint8_t *pck = (k >> 3) ? pc1 : pc0; /* this lane's chroma plane */
    const int qp = __builtin_amdgcn_readfirstlane(S.rc_mode ? (S.is_idr ? S.qp_idr : B.qp_cur[item]) : g.qp);
    const int qpc = venc_chroma_qp(qp);
    const int mvthr = g.subpel ? 4 : 1;
    const int alpha_y = c_dbk_alpha[qp], beta_y = c_dbk_beta[qp], alpha_c = c_dbk_alpha[qpc], beta_c = c_dbk_beta[qpc];
    const uint32_t tcw_y = *(const uint32_t *)c_dbk_tc0[qp], tcw_c = *(const uint32_t *)c_dbk_tc0[qpc];
    const int mbw = g.mbw;

    for (int base = 0; base < nrows; base += R) { /* passes (one unless the slice is taller than the block's rows) */
        const int rp = min(R, nrows - base);      /* rows of this pass */
        const int rlp = 4 * wf + rw;              /* this lane's row inside the pass */
        const bool row_ok = rlp < rp;
        const int my = r0 + base + min(rlp, rp - 1);
        const bool aT = base + rlp > 0;             /* a row above inside the slice */
        const bool from_mem = rlp == 0 && base > 0; /* ... whose bottom rows come from memory (the last row of the previous pass stored them) */
        const bool below = rlp + 1 < rp;            /* a row below in this pass takes this row's bottom rows through LDS */
        dl_row &T = s_rows[min(rlp, rp - 1)];
        dl_row &TB = s_rows[min(rlp + 1, rp - 1)];

        /* this lane's rows of the pass (the macroblock's offset is all a step adds) */
        uint8_t *const yrow = py + (size_t)(my * 16 + k) * ystride, *const crow = pck + (size_t)(my * 8 + (k & 7)) * cstride;
        uint8_t *const yabv = py + (size_t)(my * 16 - 4 + (k & 3)) * ystride, *const cabv = pck + (size_t)(my * 8 - 4 + (k & 3)) * cstride; /* (used with aT only) */
        const int rec_part = k >= 3 ? k - 3 : k; /* lanes 0-2: this row's record halves and vector word, 3-5: the same of the row above */
        const int rec_m0 = my * mbw - (k >= 3 && aT ? mbw : 0);
        /* ---- helper: issue the loads of macroblock x of this row (no-op outside the row) ---- */
        auto fetch = [&](dl_fetch &f, int x) {
            if (!(row_ok && x >= 0 && x < mbw)) return;
            if (DO_Y) f.y = *(const uint4 *)(yrow + x * 16);
            if (DO_C) f.c = *(const uint2 *)(crow + x * 8);
            if (k < 6) {
                if (rec_part < 2) f.rec = ((const uint4 *)(infos + rec_m0 + x))[rec_part];
                else f.mv = mvs[rec_m0 + x];
            }
            if (from_mem) { /* (rare: slices taller than the block) the four rows above, as the previous pass left them */
                if (DO_Y && k < 4) f.ya = *(const uint4 *)(yabv + x * 16);
                if (DO_C && (k & 7) < 4) f.ca = *(const uint2 *)(cabv + x * 8);
            }
        };
        /* ---- helper, step t (the filter wave of the row is at macroblock x = t - 2 rlp) ---- */
        auto help = [&](dl_fetch &f, int t) {
            const int x = t - 2 * rlp;
            /* macroblock x + 1 (fetched two steps ago) into its slots; its strengths */
            const int x1 = x + 1;
            const bool in1 = row_ok && x1 >= 0 && x1 < mbw;
            const int s1 = x1 & 3, i1 = x1 & 1;
            if (in1) {
                if (DO_Y) *(uint4 *)&T.y[s1][(4 + k) * 4] = f.y;
                if (DO_C) *(uint2 *)&T.c[s1][k >> 3][(4 + (k & 7)) * 2] = f.c;
                if (k < 2) *(uint4 *)&T.info[i1][4 * k] = f.rec;
                else if (k == 2) T.info[i1][8] = f.mv;
                else if (k < 5) *(uint4 *)&T.above[4 * (k - 3)] = f.rec;
                else if (k == 5) T.above[8] = f.mv;
                if (from_mem) {
                    if (DO_Y && k < 4) *(uint4 *)&T.y[s1][k * 4] = f.ya;
                    if (DO_C && (k & 7) < 4) *(uint2 *)&T.c[s1][k >> 3][(k & 7) * 2] = f.ca;
                }
            }
            /* what became final in the last step leaves: macroblock x - 2 (its last columns met the left edge of x - 1) -- rows 0 .. 11,
             * and rows 12 .. 15 from the last row of the pass (the others handed them down); the rows above macroblock x - 1, which
             * met its top edge */
            const int x2 = x - 2;
            if (row_ok && x2 >= 0 && x2 < mbw) {
                const int s2 = x2 & 3;
                if (DO_Y && (k < 12 || !below)) *(uint4 *)(yrow + x2 * 16) = *(const uint4 *)&T.y[s2][(4 + k) * 4];
                if (DO_C && ((k & 7) < 4 || !below)) *(uint2 *)(crow + x2 * 8) = *(const uint2 *)&T.c[s2][k >> 3][(4 + (k & 7)) * 2];
            }
            const int xa = x - 1;
            if (row_ok && aT && xa >= 0 && xa < mbw) {
                const int sa = xa & 3;
                if (DO_Y && k < 4) *(uint4 *)(yabv + xa * 16) = *(const uint4 *)&T.y[sa][k * 4];
                if (DO_C && (k & 7) < 4) *(uint2 *)(cabv + xa * 8) = *(const uint2 *)&T.c[sa][k >> 3][(k & 7) * 2];
            }
            /* memory first -- the stores above, then the loads of the macroblock three steps ahead into the registers just emptied (in the
             * tile two steps from now): both have the rest of this step and the whole next one to complete */
            if (!(diag & 4)) fetch(f, x + 3); /* (VENC_DL_DIAG & 4: no loads) */
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            {   /* strengths: lane k < 8 = direction k >> 2, edge k & 3, four segments in one word (8.7.2.1 for this subset) */
                uint32_t bsw = 0;
                const int dir = (k >> 2) & 1, e = k & 3;
                if (!(diag & 2) && in1 && k < 8 && (e > 0 || (dir ? aT : x1 > 0))) { /* (VENC_DL_DIAG & 2: no strengths) */
                    const uint8_t *own = (const uint8_t *)T.info[i1];
                    const uint32_t *othw = e ? T.info[i1] : (dir ? T.above : T.info[i1 ^ 1]);
                    const uint8_t *oth = (const uint8_t *)othw;
                    if ((own[25] | oth[25]) & 1) /* venc_mbinfo.flags: intra */
                        bsw = e ? 0x03030303u : 0x04040404u;
                    else {
                        uint32_t mvbs = 0;
                        if (e == 0) {
                            const uint32_t a = T.info[i1][8], b = othw[8];
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
                if (in1 && k < 8) T.bs[i1][k] = bsw;
                /* which phases have a strength: bit k from lane k of the row (the filter wave skips the others on a scalar test) */
                const uint32_t rowbits = (uint32_t)(__ballot(bsw != 0u) >> (16 * rw)) & 255u;
                if (in1 && k == 8) T.bs[i1][8] = rowbits;
            }
        };
        /* ---- filter, step t ---- */
        auto filt = [&](int t) {
            const int x = t - 2 * rlp;
            const bool act = row_ok && x >= 0 && x < mbw;
            const int s = x & 3, sl4 = (x - 1) & 3;
            const uint4 b0 = *(const uint4 *)&T.bs[x & 1][0], b1 = *(const uint4 *)&T.bs[x & 1][4];
            const uint32_t pm = act ? T.bs[x & 1][8] : 0u;
            const uint32_t wp = (uint32_t)(__builtin_amdgcn_readlane((int)pm, 0) | __builtin_amdgcn_readlane((int)pm, 16) | __builtin_amdgcn_readlane((int)pm, 32) |
                                           __builtin_amdgcn_readlane((int)pm, 48)); /* the phases any of the wave's four macroblocks needs */
            if (!(diag & 1)) { /* (VENC_DL_DIAG & 1: no filtering -- wrong pictures, what is left is the step's fixed cost) */
            if (DO_Y) dl_phases<false>(T, s, sl4, k, act, b0, b1, wp, alpha_y, beta_y, tcw_y);
            if (DO_C) dl_phases<true>(T, s, sl4, k, act, b0, b1, wp, alpha_c, beta_c, tcw_c);
            }
            /* the bottom rows of macroblock x - 1 (their last columns have met this macroblock's left edge) go down: rows -4 .. -1 of
             * the same slot in the row below, which filters macroblock x - 1 in the next step */
            if (row_ok && below && x >= 1 && x <= mbw) {
                if (DO_Y && k >= 12) *(uint4 *)&TB.y[sl4][(k - 12) * 4] = *(const uint4 *)&T.y[sl4][(4 + k) * 4];
                if (DO_C && (k & 7) >= 4) *(uint2 *)&TB.c[sl4][k >> 3][((k & 7) - 4) * 2] = *(const uint2 *)&T.c[sl4][k >> 3][(4 + (k & 7)) * 2];
            }
        };

        dl_fetch fa, fb;
        fa.y = fb.y = fa.rec = fb.rec = fa.ya = fb.ya = make_uint4(0u, 0u, 0u, 0u), fa.c = fb.c = fa.ca = fb.ca = make_uint2(0u, 0u), fa.mv = fb.mv = 0u;
        if (helper) {
            fetch(fa, 0 - 2 * rlp); /* = the fetches of the steps -3 and -2 (rows other than the first fetch from inside the loop) */
            fetch(fb, 1 - 2 * rlp);
        }
        const int last = mbw + 1 + 2 * (rp - 1); /* the step in which the last row's helper stores the row's last macroblock */
        for (int t = -1; t <= last; t += 2) {
            if (helper) help(fa, t);
            else filt(t);
            __syncthreads();
            if (helper) help(fb, t + 1);
            else filt(t + 1);
            __syncthreads();
        }
        __threadfence_block(); /* the pass's stores before the next pass reads its last row's bottom rows */
        __syncthreads();
    }
}
