/*
 * legacy/kernels_gen1_c.hip -- first-generation kernels cut out of venc_kernels.hip (third block)
 * TEST-ONLY: compiled into libvenc_hip_legacy.so (-DVENC_LEGACY_KERNELS), the on-GPU oracles of tests/test_gpu_parity.py;
 * the product library does not contain a line of this file.  Included by venc_kernels.hip.
 */
__global__ __launch_bounds__(1024) void k_intra(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ int s_bot[2 * INTRA_MAX_WAVES][4][5];
    const int item = blockIdx.z, s = S.slice0 + blockIdx.x, tid = threadIdx.x;
    const int nw = blockDim.x >> 6;
    const int wv = tid >> 6, lane = tid & 63, l32 = lane & 31, hbase = lane & 32;
    /* a macroblock needs 24 lanes: each HALF of a wave owns a macroblock row of the band (row wv and row wv + nw),
     * so a block of 16 waves sweeps 32 rows in one pass (the 17-row slices of 2160p took two passes of 16 + 1) */
    const int rw = wv + nw * (lane >> 5);
    const int r0 = B.slice_first_row[s], r1 = B.slice_first_row[s + 1];
    const bool luma = l32 < 16;
    const int plane = luma ? 0 : 1 + (((l32 - 16) >> 2) & 1);
    const int k = l32 & 3;
    const int bx = luma ? blk_x(l32) * 4 : (k & 1) * 4, by = luma ? blk_y(l32) * 4 : (k >> 1) * 4;

    int cstride, pw, ph, wstride;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], plane, cstride, pw, ph);
    uint8_t *rec = rec_plane(B, g, S.parity, item, plane, wstride);

    for (int cb = r0; cb < r1; cb += 2 * nw) {
        const int nr = min(2 * nw, r1 - cb);
        const int my = cb + rw;
        const int steps = g.mbw + nr - 1;
        int leftY = 0, leftC0 = 0, leftC1 = 0; /* per lane: sums relevant to this lane's plane */
        /* the current block's pixels are fetched one macroblock ahead so that the load latency
         * hides behind the previous macroblock's arithmetic instead of sitting on the dependency chain */
        const bool cur_al = luma ? ((g.w & 3) == 0) : ((g.w & 7) == 0);
        const int pyl = (luma ? my * 16 : my * 8) + by;
        uint32_t cq[4] = {0, 0, 0, 0};
        if (rw < nr) load_cur_rows(cur, cstride, pw, ph, bx, pyl, cur_al, cq);
        for (int step = 0; step < steps; step++) {
            const int mx = step - rw;
            if (rw < nr && mx >= 0 && mx < g.mbw) {
                uint32_t nq[4] = {0, 0, 0, 0};
                if (mx + 1 < g.mbw) load_cur_rows(cur, cstride, pw, ph, (luma ? (mx + 1) * 16 : (mx + 1) * 8) + bx, pyl, cur_al, nq);
                const int mb = my * g.mbw + mx;
                const bool aL = mx > 0, aT = my > r0;
                /* top sums: [0] luma, [1],[2] Cb halves, [3],[4] Cr halves */
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                if (aT) {
                    if (rw > 0) {
                        const int *sb = s_bot[rw - 1][mx & 3];
                        t0 = sb[0], t1 = sb[1], t2 = sb[2], t3 = sb[3], t4 = sb[4];
                    } else {
                        /* row above belongs to the previous chunk of this slice: re-read its bottom lines */
                        int ys, us;
                        const uint8_t *ry = rec_plane(B, g, S.parity, item, 0, ys) + (size_t)(my * 16 - 1) * ys + mx * 16;
                        const uint8_t *ru = rec_plane(B, g, S.parity, item, 1, us) + (size_t)(my * 8 - 1) * us + mx * 8;
                        const uint8_t *rv = rec_plane(B, g, S.parity, item, 2, us) + (size_t)(my * 8 - 1) * us + mx * 8;
                        for (int x = 0; x < 16; x++) t0 += ry[x];
                        for (int x = 0; x < 4; x++) t1 += ru[x], t2 += ru[4 + x], t3 += rv[x], t4 += rv[4 + x];
                    }
                }
                int pred;
                if (luma) {
                    pred = (aL && aT) ? (leftY + t0 + 16) >> 5 : aL ? (leftY + 8) >> 4 : aT ? (t0 + 8) >> 4 : 128;
                } else {
                    const int sT = plane == 1 ? ((k & 1) ? t2 : t1) : ((k & 1) ? t4 : t3);
                    const int sL = (k >> 1) ? leftC1 : leftC0;
                    bool useT = aT, useL = aL;
                    if (k == 1 && aT) useL = false;
                    if (k == 2 && aL) useT = false;
                    pred = (useT && useL) ? (sT + sL + 4) >> 3 : useT ? (sT + 2) >> 2 : useL ? (sL + 2) >> 2 : 128;
                }
                const int px = (luma ? mx * 16 : mx * 8) + bx, py = (luma ? my * 16 : my * 8) + by;
                int c[16], p[16];
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        c[4 * y + x] = (int)((cq[y] >> (8 * x)) & 255u);
                        p[4 * y + x] = pred;
                    }
#pragma unroll
                for (int y = 0; y < 4; y++) cq[y] = nq[y];
                mb_lane_out o;
                mb_code<true, true>(lane, c, p, g, o);

                int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
                venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
                const bool st = true; /* both halves of the wave carry a macroblock */
                mb_store(l32, st, o, lv_mb, info, rec, wstride, px, py);
                /* luma DC record (slot 16, zig-zag order) and chroma DC records */
                if (st && luma) {
                    const int rp = blk_y(l32) * 4 + blk_x(l32);
                    int zi = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (venc_zigzag_ct(i) == rp) zi = i;
                    lv_mb[16 * 16 + zi] = (int16_t)o.dclv;
                }
                int dc0 = __shfl(o.dclv, hbase | (l32 & 28) | 0), dc1 = __shfl(o.dclv, hbase | (l32 & 28) | 1);
                int dc2 = __shfl(o.dclv, hbase | (l32 & 28) | 2), dc3 = __shfl(o.dclv, hbase | (l32 & 28) | 3);
                if (st && (l32 == 16 || l32 == 20)) {
                    uint4 *dst = (uint4 *)(lv_mb + (l32 == 16 ? 17 : 18) * 16);
                    dst[0] = make_uint4(((uint32_t)dc0 & 0xffffu) | ((uint32_t)dc1 << 16),
                                        ((uint32_t)dc2 & 0xffffu) | ((uint32_t)dc3 << 16), 0, 0);
                }
                const uint32_t hnz = (uint32_t)(__ballot(o.nnz > 0) >> hbase), hdc = (uint32_t)(__ballot(o.dclv != 0) >> hbase);
                if (l32 == 0) {
                    int cbpl = (hnz & 0xffffu) ? 15 : 0;
                    int cbpc = ((hnz >> 16) & 0xffu) ? 2 : ((hdc >> 16) & 0xffu) ? 1 : 0;
                    info->cbp = (uint8_t)cbpl;
                    info->cbpc = (uint8_t)cbpc;
                    info->flags = 1 | (2 << 1);
                    B.mvsad[(size_t)item * g.nmb + mb] = 0;
                }
                /* neighbour sums for the next macroblocks */
                const int pb = o.rec[12] + o.rec[13] + o.rec[14] + o.rec[15];
                const int pr = o.rec[3] + o.rec[7] + o.rec[11] + o.rec[15];
                const int botY = __shfl(pb, hbase | 10) + __shfl(pb, hbase | 11) + __shfl(pb, hbase | 14) + __shfl(pb, hbase | 15);
                const int rightY = __shfl(pr, hbase | 5) + __shfl(pr, hbase | 7) + __shfl(pr, hbase | 13) + __shfl(pr, hbase | 15);
                const int bcb0 = __shfl(pb, hbase | 18), bcb1 = __shfl(pb, hbase | 19), bcr0 = __shfl(pb, hbase | 22), bcr1 = __shfl(pb, hbase | 23);
                const int rcb0 = __shfl(pr, hbase | 17), rcb1 = __shfl(pr, hbase | 19), rcr0 = __shfl(pr, hbase | 21), rcr1 = __shfl(pr, hbase | 23);
                leftY = rightY;
                leftC0 = plane == 2 ? rcr0 : rcb0;
                leftC1 = plane == 2 ? rcr1 : rcb1;
                if (l32 == 0) {
                    int *sb = s_bot[rw][mx & 3];
                    sb[0] = botY, sb[1] = bcb0, sb[2] = bcb1, sb[3] = bcr0, sb[4] = bcr1;
                }
            }
            /* LDS-only barrier: __syncthreads() also waits for every global store and for the prefetch
             * (s_waitcnt vmcnt(0)), which put a memory round trip on each step of the dependency chain */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        __syncthreads(); /* the next pass re-reads this pass's last reconstructed row from memory */
    }
}

