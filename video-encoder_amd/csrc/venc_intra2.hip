/*
 * venc_intra2.hip -- I pictures (Intra16x16, DC prediction) as three kernels instead of one
 * dependency wavefront that carried the whole transform (k_intra, venc_kernels.hip: ~1400
 * instructions per macroblock step, ~4 us per step, 0.5 ms per I picture at 1080p).
 *
 * DC prediction makes the predictor a constant p per 4x4 block, and both transforms are linear:
 *
 *   forward:  T(cur - p) = T(cur) - p * T(1), and T(1) is 16 at position (0,0), zero elsewhere.
 *             Every AC coefficient -- hence every AC level, every nnz, the luma part of
 *             coded_block_pattern -- is independent of the neighbours.  Only W[0] = sum(cur) - 16 p
 *             depends on them.
 *   inverse:  the (0,0) coefficient d0 enters every output sample of venc_idct4 with weight 1 and
 *             passes none of its ">> 1" terms, so  rec = clip(p + ((h + d0 + 32) >> 6))
 *             = clip((h + d0 + 64 p + 32) >> 6),  h = the un-rounded inverse transform of the AC part.
 *
 * So the only work that has to walk the 2-D dependency wavefront is, per macroblock: the DC
 * predictor, the 4x4 (luma) and 2x2 (chroma) Hadamard of the 24 block DCs, their quantisation and
 * dequantisation, and the reconstruction of the 61 edge samples whose sums feed the neighbours.
 *
 *   k_intra_fwd   parallel over macroblocks: forward transform of the source, AC levels, nnz, the
 *                 un-rounded inverse of the AC part for the edge samples -> a 512-byte record per MB
 *   k_intra_dc    one block per slice, half-wave per MB row (as k_intra): the DC chain, ~150
 *                 instructions per step.  The compute waves issue no global loads: the remaining
 *                 waves of the block copy the records two steps ahead into an LDS ring, so no
 *                 s_waitcnt vmcnt ever sits on the chain, and the step barrier is LDS-only.
 *   k_intra_rec   parallel: inverse transform with d0 + 64 p in position 0, reconstruction.
 *
 * Results are bit-identical to k_intra (kept as VENC_INTRA=legacy and as the test oracle on the GPU).
 * Follows spec/CODEC.md sections "Intra prediction" and "Transform and quantisation".
 */
#define INTRA_CW_MAX 12       /* compute waves per block: 24 MB rows per pass */
#define INTRA_REC_DWORDS 128  /* per-macroblock record: [0..23] block DCs, [24 + 8 e ..] edge samples of edge block e */
#define INTRA_MAX_MBW 1024

/* edge blocks (right column or bottom row of their plane inside the macroblock) in lane order:
 * luma 5 7 10 11 13 14 15, Cb 1 2 3, Cr 1 2 3 */
DEVFN int intra_edge_slot(int l32) {
    if (l32 < 16) {
        const bool e = blk_x(l32) == 3 || blk_y(l32) == 3;
        /* rank among {5,7,10,11,13,14,15} */
        return e ? __popc(0xECA0u & ((1u << l32) - 1u)) : -1;
    }
    if (l32 < 24) {
        const int k = l32 & 3;
        return k ? 7 + 3 * ((l32 - 16) >> 2) + (k - 1) : -1;
    }
    return -1;
}

/* venc_idct4 without the final (x + 32) >> 6 */
DEVFN void intra_idct4_raw(const int d[16], int h[16]) {
    int f[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[4 * i] + d[4 * i + 2], e1 = d[4 * i] - d[4 * i + 2];
        int e2 = (d[4 * i + 1] >> 1) - d[4 * i + 3], e3 = d[4 * i + 1] + (d[4 * i + 3] >> 1);
        f[4 * i] = e0 + e3;
        f[4 * i + 1] = e1 + e2;
        f[4 * i + 2] = e1 - e2;
        f[4 * i + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = f[j] + f[8 + j], g1 = f[j] - f[8 + j];
        int g2 = (f[4 + j] >> 1) - f[12 + j], g3 = f[4 + j] + (f[12 + j] >> 1);
        h[j] = g0 + g3;
        h[4 + j] = g1 + g2;
        h[8 + j] = g1 - g2;
        h[12 + j] = g0 - g3;
    }
}

/* clip_u8(x >> 6), written as clamp-then-shift.  The natural form makes this compiler select
 * v_ashr_pk_u8_i32 (gfx950) for pairs of samples and then OR the next samples into bits 16..31 of its
 * result as if they were zero; on the hardware they keep the old contents of the destination register,
 * which is the first operand here -- wrong bytes whenever that sum is negative or >= 65536 (found by the
 * QP 51 noise case of tests/test_gpu_parity.py: one block of one picture). */
DEVFN int intra_px(int x) { return (x < 0 ? 0 : x > 16383 ? 16383 : x) >> 6; }

/* ================================================================== k_intra_fwd
 * Half-wave per macroblock (lane roles as mb_code: 0..15 luma block, 16..19 Cb, 20..23 Cr). */
__global__ __launch_bounds__(256) void k_intra_fwd(venc_geom g, venc_bufs B, venc_step S) {
    const int item = blockIdx.z, tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hbase = lane & 32;
    const int mb = S.mb0 + blockIdx.x * 8 + (tid >> 5);
    if (mb >= S.mb1) return;
    const int my = mb / g.mbw, mx = mb - my * g.mbw;
    const bool act = l32 < 24, luma = l32 < 16;
    const int plane = luma ? 0 : 1 + (((l32 - 16) >> 2) & 1);
    const int k = l32 & 3;
    const int bx = luma ? blk_x(l32) * 4 : (k & 1) * 4, by = luma ? blk_y(l32) * 4 : (k >> 1) * 4;
    int cstride, pw, ph;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], plane, cstride, pw, ph);
    const bool cur_al = luma ? ((g.w & 3) == 0) : ((g.w & 7) == 0);
    uint32_t cq[4] = {0, 0, 0, 0};
    if (act) load_cur_rows(cur, cstride, pw, ph, (luma ? mx * 16 : mx * 8) + bx, (luma ? my * 16 : my * 8) + by, cur_al, cq);
    int c[16], w[16];
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) c[4 * y + x] = (int)((cq[y] >> (8 * x)) & 255u);
    venc_fdct4(c, w);

    const int qmf0 = luma ? g.ql.mf[0] : g.qc.mf[0], qmf1 = luma ? g.ql.mf[1] : g.qc.mf[1], qmf2 = luma ? g.ql.mf[2] : g.qc.mf[2];
    const int qv0 = luma ? g.ql.v[0] : g.qc.v[0], qv1 = luma ? g.ql.v[1] : g.qc.v[1], qv2 = luma ? g.ql.v[2] : g.qc.v[2];
    const int qbits = luma ? g.ql.qbits : g.qc.qbits, per = luma ? g.ql.per : g.qc.per;
    const int f = luma ? g.ql.f_intra : g.qc.f_intra;
    int lv[16], d[16], nnz = 0;
    lv[0] = 0;
    d[0] = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const int r = venc_zigzag_ct(i);
        const int cls = venc_pos_class(r);
        const int mf = cls == 0 ? qmf0 : cls == 1 ? qmf1 : qmf2;
        const int vv = cls == 0 ? qv0 : cls == 1 ? qv1 : qv2;
        const int l = venc_quant_mag(w[r], mf, f, qbits);
        lv[i] = l;
        nnz += (l != 0);
        d[r] = VENC_MUL24(l, vv) << per;
    }
    int h[16];
    intra_idct4_raw(d, h);

    const size_t gmb = (size_t)item * g.nmb + mb;
    venc_mbinfo *info = B.mbinfo + gmb;
    if (act) {
        if (nnz) {
            uint32_t p[8];
#pragma unroll
            for (int i = 0; i < 8; i++) p[i] = ((uint32_t)lv[2 * i] & 0xffffu) | ((uint32_t)lv[2 * i + 1] << 16);
            uint4 *dst = (uint4 *)(B.levels + gmb * VENC_MB_COEFS + (luma ? l32 : 19 + (l32 - 16)) * 16);
            dst[0] = make_uint4(p[0], p[1], p[2], p[3]);
            dst[1] = make_uint4(p[4], p[5], p[6], p[7]);
        }
        info->nnz[luma ? blk_y(l32) * 4 + blk_x(l32) : l32] = (uint8_t)nnz;
        int *rec = B.intra_rec + gmb * INTRA_REC_DWORDS;
        rec[l32] = w[0];
        const int es = intra_edge_slot(l32);
        if (es >= 0) {
            int4 *e = (int4 *)(rec + 24 + 8 * es);
            e[0] = make_int4(h[3], h[7], h[11], h[15]);   /* right column */
            e[1] = make_int4(h[12], h[13], h[14], h[15]); /* bottom row */
        }
    }
    const uint32_t hnz = (uint32_t)(__ballot(act && nnz > 0) >> hbase);
    if (l32 == 0) {
        info->cbp = (uint8_t)((hnz & 0xffffu) ? 15 : 0);
        info->cbpc = (uint8_t)(((hnz >> 16) & 0xffu) ? 2 : 0); /* 1 (DC only) is decided by k_intra_dc / k_intra_rec */
        info->flags = 1 | (2 << 1); /* Intra16x16, Intra16x16PredMode 2 (DC): the only mode of this path */
        B.mvsad[gmb] = 0;
    }
}

/* ================================================================== k_intra_dc */
DEVFN void lds_barrier() {
    /* __syncthreads() would also wait for the global stores of the step (s_waitcnt vmcnt(0)) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define DPP_QUAD_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E /* quad_perm [2,3,0,1] */
#define DPP_ROW_ROR(n) (0x120 + (n))
/* natural-order Walsh-Hadamard butterflies over lane bits 0,1 (all lanes) and 2,3 (luma lanes):
 * returns the 2-bit result in v2 and the 4-bit result in v4.  With luma lane = luma4x4BlkIdx the
 * lane (x, y) ends up holding the coefficient (sig(x), sig(y)) of the standard's sequency-ordered
 * 4x4 Hadamard, sig = 0,3,1,2; applying the same butterflies to the levels brings block (x, y)
 * back to lane (x, y) because the matrix is symmetric. */
DEVFN void intra_wht(int v, int l32, int &v2, int &v4) {
    int o = (int)dpp_fetch<DPP_QUAD_XOR1, 0xf>((uint32_t)v, 0u);
    v = o + ((l32 & 1) ? -v : v);
    o = (int)dpp_fetch<DPP_QUAD_XOR2, 0xf>((uint32_t)v, 0u);
    v = o + ((l32 & 2) ? -v : v);
    v2 = v;
    const int oa = (int)dpp_fetch<DPP_ROW_ROR(4), 0xf>((uint32_t)v, 0u), ob = (int)dpp_fetch<DPP_ROW_ROR(12), 0xf>((uint32_t)v, 0u);
    v = (l32 & 4) ? oa - v : ob + v;
    o = (int)dpp_fetch<DPP_ROW_ROR(8), 0xf>((uint32_t)v, 0u);
    v = o + ((l32 & 8) ? -v : v);
    v4 = v;
}

/* Loader wave of k_intra_dc: the records of step t travel from memory straight into ring stage t & 3
 * (global_load_lds_dwordx4: no registers in between), issued three steps ahead.  A wave fetches NJ runs of 64 units
 * (16 bytes each) per step -- NJ is fixed per wave and pass, so "the records of step s + 1 have landed" is
 * s_waitcnt vmcnt(2 NJ): the fetches of steps s + 2 and s + 3 stay in flight.  Lanes without a record (rows beyond
 * the pass, macroblocks beyond the row) read a dummy address; their ring slots are never read.  The first version
 * fetched one step ahead into registers and waited with vmcnt(0): a step of the chain then took as long as a
 * global load, 0.9-1.3 us. */
#define INTRA_RING_STAGE_UINT4 (2 * INTRA_CW_MAX * 32)
template <int NJ>
DEVFN void intra_loader(const int *recs, uint4 *ring, int mbw, int cb, int nr, int steps, int lt0, int nload, int lane) {
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int u0 = lt0 + j * nload, u = u0 + lane, row = u >> 5, mxx = t - row;
            const bool ok = row < nr && mxx >= 0 && mxx < mbw;
            const uint4 *src = ok ? (const uint4 *)(recs + ((size_t)(cb + row) * mbw + mxx) * INTRA_REC_DWORDS) + (u & 31) : (const uint4 *)recs;
            uint4 *dst = ring + (t & 3) * INTRA_RING_STAGE_UINT4 + u0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    constexpr int n = 2 * NJ; /* s_waitcnt immediate: vmcnt in bits 3:0 (and 15:14), expcnt and lgkmcnt left at their maxima */
    fetch(0);
    fetch(1);
    fetch(2);
    __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8));
    __builtin_amdgcn_s_barrier();
    for (int step = 0; step < steps; step++) {
        fetch(step + 3);
        __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8));
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
}

__global__ __launch_bounds__(1024) void k_intra_dc(venc_geom g, venc_bufs B, venc_step S, int cw) {
    __shared__ int s_bot[2 * INTRA_CW_MAX][4][5];
    __shared__ int s_last[INTRA_MAX_MBW][5]; /* bottom sums of the last row of the previous pass */
    __shared__ uint4 s_ring[4][2 * INTRA_CW_MAX][INTRA_REC_DWORDS / 4];
    const int item = blockIdx.z, s = S.slice0 + blockIdx.x, tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63, l32 = lane & 31, hbase = lane & 32;
    const int r0 = B.slice_first_row[s], r1 = B.slice_first_row[s + 1];
    const int *recs = B.intra_rec + (size_t)item * g.nmb * INTRA_REC_DWORDS;

    if (wv >= cw) {
        /* ---- loader waves (intra_loader above) */
        const int nload = (int)blockDim.x - 64 * cw, lt0 = (tid - 64 * cw) & ~63;
        for (int cb = r0; cb < r1; cb += 2 * cw) {
            const int nr = min(2 * cw, r1 - cb), steps = g.mbw + nr - 1;
            /* fetches of this wave per step: its j-th run of 64 units starts at lt0 + j * nload (wave-uniform) */
            int nj = 0;
            for (int j = 0; j < 3; j++) nj += lt0 + j * nload < nr * 32 ? 1 : 0;
            uint4 *ring = &s_ring[0][0][0];
            if (nj == 0) {
                for (int k = 0; k < steps + 1; k++) __builtin_amdgcn_s_barrier();
            } else if (nj == 1)
                intra_loader<1>(recs, ring, g.mbw, cb, nr, steps, lt0, nload, lane);
            else if (nj == 2)
                intra_loader<2>(recs, ring, g.mbw, cb, nr, steps, lt0, nload, lane);
            else
                intra_loader<3>(recs, ring, g.mbw, cb, nr, steps, lt0, nload, lane);
            __syncthreads();
        }
        return;
    }

    /* ---- compute waves: each half owns a macroblock row of the pass */
    const int rw = wv + cw * (lane >> 5);
    const bool act = l32 < 24, luma = l32 < 16;
    const int plane = luma ? 0 : 1 + (((l32 - 16) >> 2) & 1);
    const int k = l32 & 3;
    const int es = intra_edge_slot(l32);
    const bool e_right = luma ? blk_x(l32) == 3 : (k & 1) != 0, e_bot = luma ? blk_y(l32) == 3 : (k >> 1) != 0;
    /* where this lane's luma DC level goes in the macroblock's DC record (zig-zag index of (sig(y), sig(x))) */
    int zi = 0;
    {
        const int sy = (0x9C >> (2 * blk_y(l32 & 15))) & 3, sx = (0x9C >> (2 * blk_x(l32 & 15))) & 3; /* sig = 0,3,1,2 */
        const int rp = sy * 4 + sx;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (venc_zigzag_ct(i) == rp) zi = i;
    }
    for (int cb = r0; cb < r1; cb += 2 * cw) {
        const int nr = min(2 * cw, r1 - cb), steps = g.mbw + nr - 1;
        const int my = cb + rw;
        int leftY = 0, leftC0 = 0, leftC1 = 0;
        lds_barrier();
        for (int step = 0; step < steps; step++) {
            const int mx = step - rw;
            if (rw < nr && mx >= 0 && mx < g.mbw) {
                const int *rg = (const int *)s_ring[step & 3][rw];
                const int wdc = rg[l32];
                int4 er = make_int4(0, 0, 0, 0), eb = make_int4(0, 0, 0, 0);
                if (es >= 0) {
                    er = *(const int4 *)(rg + 24 + 8 * es);
                    eb = *(const int4 *)(rg + 28 + 8 * es);
                }
                const int mb = my * g.mbw + mx;
                const bool aL = mx > 0, aT = my > r0;
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                if (aT) {
                    const int *sb = rw > 0 ? s_bot[rw - 1][mx & 3] : s_last[mx];
                    t0 = sb[0], t1 = sb[1], t2 = sb[2], t3 = sb[3], t4 = sb[4];
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
                /* DC of the residual, Hadamard, quantisation, and back */
                int y2, y4;
                intra_wht(act ? wdc - 16 * pred : 0, l32, y2, y4);
                const int dclv = luma ? venc_quant_luma_dc(y4, &g.ql) : venc_quant_chroma_dc(y2, &g.qc, 1);
                int f2, f4;
                intra_wht(dclv, l32, f2, f4);
                const int d0 = luma ? venc_dequant_luma_dc(f4, &g.ql) : venc_dequant_chroma_dc(f2, &g.qc);
                const int d0p = d0 + 64 * pred + 32;
                /* edge samples of the reconstruction -> sums for the right and lower neighbours */
                int pr = 0, pb = 0;
                if (e_right)
                    pr = intra_px(er.x + d0p) + intra_px(er.y + d0p) + intra_px(er.z + d0p) + intra_px(er.w + d0p);
                if (e_bot)
                    pb = intra_px(eb.x + d0p) + intra_px(eb.y + d0p) + intra_px(eb.z + d0p) + intra_px(eb.w + d0p);
                /* luma: both sums to every luma lane (rotations inside the row of 16 lanes); sums <= 4080 */
                uint32_t pk = luma ? ((uint32_t)pr << 16) | (uint32_t)pb : 0u;
                pk += dpp_fetch<DPP_ROW_ROR(8), 0xf>(pk, 0u);
                pk += dpp_fetch<DPP_ROW_ROR(4), 0xf>(pk, 0u);
                pk += dpp_fetch<DPP_ROW_ROR(2), 0xf>(pk, 0u);
                pk += dpp_fetch<DPP_ROW_ROR(1), 0xf>(pk, 0u);
                leftY = (int)(pk >> 16);
                /* chroma: the right-column sums of blocks 1 and 3 to the four lanes of the plane */
                leftC0 = (int)dpp_fetch<0x55, 0xf>((uint32_t)pr, 0u); /* quad_perm [1,1,1,1] */
                leftC1 = (int)dpp_fetch<0xFF, 0xf>((uint32_t)pr, 0u); /* quad_perm [3,3,3,3] */
                int *sb = s_bot[rw][mx & 3];
                if (l32 == 0) sb[0] = (int)(pk & 0xffffu);
                if (!luma && act && (k >> 1)) sb[1 + 2 * (plane - 1) + (k & 1)] = pb;
                if (rw == nr - 1 && cb + nr < r1) { /* a further pass follows: keep this row's bottom sums */
                    if (l32 == 0) s_last[mx][0] = (int)(pk & 0xffffu);
                    if (!luma && act && (k >> 1)) s_last[mx][1 + 2 * (plane - 1) + (k & 1)] = pb;
                }
                /* results (nobody in this kernel reads them back) */
                const size_t gmb = (size_t)item * g.nmb + mb;
                int16_t *lv_mb = B.levels + gmb * VENC_MB_COEFS;
                if (act) B.intra_d0[gmb * 32 + l32] = d0p;
                if (luma)
                    lv_mb[16 * 16 + zi] = (int16_t)dclv;
                else if (act) {
                    lv_mb[(plane == 1 ? 17 : 18) * 16 + k] = (int16_t)dclv;
                    if (k == 0) *(uint2 *)(lv_mb + (plane == 1 ? 17 : 18) * 16 + 4) = make_uint2(0u, 0u);
                }
                const uint32_t hdc = (uint32_t)(__ballot(act && !luma && dclv != 0) >> hbase);
                if (l32 == 0) B.mbinfo[gmb].pad[0] = hdc ? 1 : 0;
            }
            lds_barrier();
        }
        __syncthreads();
    }
}

/* ================================================================== k_intra_rec
 * Half-wave per macroblock: inverse transform of the AC levels with d0 + 64 p (+ 32, already
 * added) in position 0, reconstruction, and the final chroma coded_block_pattern. */
__global__ __launch_bounds__(256) void k_intra_rec(venc_geom g, venc_bufs B, venc_step S) {
    const int item = blockIdx.z, tid = threadIdx.x, lane = tid & 63, l32 = lane & 31;
    const int mb = S.mb0 + blockIdx.x * 8 + (tid >> 5);
    if (mb >= S.mb1) return;
    const int my = mb / g.mbw, mx = mb - my * g.mbw;
    const size_t gmb = (size_t)item * g.nmb + mb;
    venc_mbinfo *info = B.mbinfo + gmb;
    if (l32 == 0 && info->cbpc == 0 && info->pad[0]) info->cbpc = 1;
    if (l32 >= 24) return;
    const bool luma = l32 < 16;
    const int plane = luma ? 0 : 1 + (((l32 - 16) >> 2) & 1);
    const int k = l32 & 3;
    const int bx = luma ? blk_x(l32) * 4 : (k & 1) * 4, by = luma ? blk_y(l32) * 4 : (k >> 1) * 4;
    const int qv0 = luma ? g.ql.v[0] : g.qc.v[0], qv1 = luma ? g.ql.v[1] : g.qc.v[1], qv2 = luma ? g.ql.v[2] : g.qc.v[2];
    const int per = luma ? g.ql.per : g.qc.per;
    const int nnz = info->nnz[luma ? blk_y(l32) * 4 + blk_x(l32) : l32];
    const int d0p = B.intra_d0[gmb * 32 + l32];
    int d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = 0;
    if (nnz) {
        const uint4 *src = (const uint4 *)(B.levels + gmb * VENC_MB_COEFS + (luma ? l32 : 19 + (l32 - 16)) * 16);
        const uint4 a = src[0], b = src[1];
        const uint32_t p[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 1; i < 16; i++) {
            const int r = venc_zigzag_ct(i);
            const int cls = venc_pos_class(r);
            const int vv = cls == 0 ? qv0 : cls == 1 ? qv1 : qv2;
            const int l = (int)(int16_t)((i & 1) ? p[i >> 1] >> 16 : p[i >> 1] & 0xffffu);
            d[r] = VENC_MUL24(l, vv) << per;
        }
    }
    int h[16];
    intra_idct4_raw(d, h);
    int wstride;
    uint8_t *rec = rec_plane(B, g, S.parity, item, plane, wstride);
    const int px = (luma ? mx * 16 : mx * 8) + bx, py = (luma ? my * 16 : my * 8) + by;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t v = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) v |= (uint32_t)intra_px(h[4 * y + x] + d0p) << (8 * x);
        *(uint32_t *)(rec + (size_t)(py + y) * wstride + px) = v;
    }
}

/* ================================================================== Intra4x4 (spec/CODEC.md section 3, intra_modes = 3; H.264 8.3.1)
 * The trial of a macroblock as sixteen 4x4 blocks in decoding order, run by the 16 luma lanes of the macroblock's half-wave
 * with lane = SAMPLE of the block at hand (x = lane & 3, y = lane >> 2) -- the blocks are a serial chain (each predicts from
 * the reconstruction of the ones before it), so the parallelism is inside a block.  Per block:
 *   edge      the 13 neighbouring samples as one vector E' across the lanes: E'[0] = E'[1] = L3, L2, L1, L0, corner, T0 .. T7,
 *             E'[14] = T7 (the doubled ends make p[-1,3] and p[7,-1] repeat where 8.3.1.2.4 / .9 weight them by 3); lane i
 *             also forms F3[i] = (E'[i-1] + 2 E'[i] + E'[i+1] + 2) >> 2 and A2[i] = (E'[i] + E'[i+1] + 1) >> 1
 *   predict   every directional predictor of every sample is ONE of those 45 numbers (i4_tab: which, per mode and sample):
 *             a lane fetches its eight with wave shuffles; DC from the sums of T0..T3 and L0..L3
 *   choose    SAD of each usable mode over the 16 lanes, smallest wins, smaller mode number on ties
 *   code      residual, 4x4 transform (each lane one coefficient: shuffles along its row, then its column), quantisation of
 *             all 16 coefficients with the intra rounding, dequantisation, inverse transform, reconstruction into the tile
 * The tile (LDS, one per half-wave) holds the macroblock's reconstruction with the row above (and four samples beyond),
 * the column to the left and the corner, its source samples, the levels and counts of the trial. */
#define INTRA4_MAX_WAVES 8 /* waves per block of k_intra_modes<true>: 512 threads leave it 256 registers (at 1024 threads and 128 the trial spilled 90) */
struct i4_tile {
    uint32_t lv[16 * 8];  /* [block, luma4x4BlkIdx][16 levels in scan order, int16 pairs] */
    uint32_t src[16 * 4]; /* source samples, row y at bytes 16 y .. */
    uint32_t rec[17 * 6]; /* reconstruction: row r = -1 .. 15 at byte (r + 1) * 24, column c = -4 .. 19 at byte 4 + c */
    uint32_t nz[4];       /* bytes: non-zero levels of each block */
    uint32_t mode[4];     /* bytes: Intra4x4PredMode of each block */
    uint32_t pad[2];
};
#define I4_RS 24 /* bytes per row of i4_tile.rec */

/* (kind << 4) | index into E': kind 0 = E', 1 = F3, 2 = A2.  Mode m, sample (x, y); 8.3.1.2.1 - 8.3.1.2.9 with
 * P(x, -1) = E'[6 + x], P(-1, y) = E'[4 - y], P(-1, -1) = E'[5] */
DEVFN int i4_tab(int m, int x, int y) {
    switch (m) {
        case 0: return 6 + x;
        case 1: return 4 - y;
        case 3: return (1 << 4) | (7 + x + y);
        case 4: return (1 << 4) | (5 + x - y);
        case 5: {
            const int j = x - (y >> 1), z = 2 * x - y;
            if (z >= 0 && !(z & 1)) return (2 << 4) | (5 + j);
            if (z >= -1) return (1 << 4) | (5 + j);
            return (1 << 4) | (6 - y);
        }
        case 6: {
            const int j = y - (x >> 1), z = 2 * y - x;
            if (z >= 0 && !(z & 1)) return (2 << 4) | (4 - j);
            if (z >= -1) return (1 << 4) | (5 - j);
            return (1 << 4) | (4 + x);
        }
        case 7: {
            const int j = x + (y >> 1);
            return !(y & 1) ? (2 << 4) | (6 + j) : (1 << 4) | (7 + j);
        }
        case 8: {
            const int j = y + (x >> 1), z = x + 2 * y;
            if (z > 5) return 1;
            if (z == 5) return (1 << 4) | 1;
            return !(z & 1) ? (2 << 4) | (3 - j) : (1 << 4) | (3 - j);
        }
    }
    return 0;
}

/* one pass of the inverse transform's butterfly for the lane at position t (0 .. 3) of its row / column: inputs a b c d
 * in position order (8.5.12.2: e0 = a + c, e1 = a - c, e2 = (b >> 1) - d, e3 = b + (d >> 1); f = e0 + e3, e1 + e2, e1 - e2, e0 - e3) */
DEVFN int i4_inv1d(int t, int a, int b, int c, int d) {
    const int bb = (t == 1 || t == 2) ? b >> 1 : b, dd = (t == 0 || t == 3) ? d >> 1 : d;
    const int tb = t < 2 ? bb : -bb, tc = (t == 0 || t == 3) ? c : -c, td = (t & 1) ? -dd : dd;
    return a + tb + tc + td;
}
DEVFN int i4_fwd1d(int t, int a, int b, int c, int d) { /* Cf = [1 1 1 1; 2 1 -1 -2; 1 -1 -1 1; 1 -2 2 -1] */
    return t == 0 ? a + b + c + d : t == 1 ? 2 * a + b - c - 2 * d : t == 2 ? a - b - c + d : a - 2 * b + 2 * c - d;
}

/* sum over the 16 lanes of a DPP row, the same in each of them */
DEVFN uint32_t i4_row_sum(uint32_t v, int lane) {
    v += dpp_fetch<DPP_ROW_SHR(1), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(2), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(4), 0xf>(v, 0u);
    v += dpp_fetch<DPP_ROW_SHR(8), 0xf>(v, 0u);
    return (uint32_t)__shfl((int)v, (lane & 48) | 15);
}

/* The trial.  Called by all 32 lanes of the macroblock's half-wave; the tile's row -1 / column -1 / corner and its source
 * samples are in place.  A block waits only for its left, upper, upper-left and upper-right neighbours, so the sixteen blocks
 * go in TEN rounds -- round L = the blocks with bx + 2 by = L, at most two of them, one on lanes 0-15 of the half and one on
 * lanes 16-31 (the chroma lanes, idle here).  aL, aT, aTR: the macroblock to the left, above, above-right is in the slice
 * (the one above-left iff both aL and aT).  best16 = the SAD the trial has to beat: once the sum of the blocks done reaches
 * it in both halves of the wave, the rest is skipped.  Returns the sum of the sixteen blocks' SADs (or a partial sum
 * >= best16), the same in every lane of the half; reconstruction, levels, counts and modes are left in the tile. */
DEVFN uint32_t i4_trial(i4_tile &T, const venc_qparams &q, int lane, bool aL, bool aT, bool aTR, uint32_t best16) {
    const int grp = lane & 48, second = (lane >> 4) & 1, p = lane & 15, x = p & 3, y = p >> 2;
    uint8_t *const rec = (uint8_t *)T.rec;
    const uint8_t *const src = (const uint8_t *)T.src;
    /* where this lane's entry of E' sits relative to the block's top-left sample: lanes 0, 1 L3; 2 L2; 3 L1; 4 L0; 5 the corner;
     * 6 .. 13 T0 .. T7; 14, 15 T7 */
    const int e_row = p <= 4 ? (p == 0 ? 3 : 4 - p) : -1;
    const int e_col = p <= 5 ? -1 : min(p - 6, 7);
    int tab[9];
#pragma unroll
    for (int m = 0; m < 9; m++) tab[m] = i4_tab(m, x, y);
    const int cls = venc_pos_class(p);
    const int qm0 = __builtin_amdgcn_readfirstlane(q.mf[0]), qm1 = __builtin_amdgcn_readfirstlane(q.mf[1]), qm2 = __builtin_amdgcn_readfirstlane(q.mf[2]);
    const int qv0 = __builtin_amdgcn_readfirstlane(q.v[0]), qv1 = __builtin_amdgcn_readfirstlane(q.v[1]), qv2 = __builtin_amdgcn_readfirstlane(q.v[2]);
    const int mf = cls == 0 ? qm0 : cls == 1 ? qm1 : qm2, vv = (cls == 0 ? qv0 : cls == 1 ? qv1 : qv2) << q.per; /* (selects of opaque reads: no indexing into scratch) */
    int scan = 0; /* scan position of this lane's coefficient (raster p) */
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (venc_zigzag_ct(i) == p) scan = i;
    uint32_t total = 0; /* SADs of the blocks this lane's group has done */
    for (int L = 0; L < 10; L++) {
        /* the block of this lane's group in round L: (bx, by) with bx + 2 by = L; the first group takes the upper one */
        const int by0 = L <= 3 ? 0 : (L - 2) >> 1; /* smallest by with bx = L - 2 by <= 3 */
        const int by = by0 + second, bx = L - 2 * by;
        const bool act = bx >= 0 && by <= 3;
        const int k = act ? blk_of(bx, by) : 0;
        const int cbx = act ? bx : 0, cby = act ? by : 0;
        const bool top = cby > 0 || aT, left = cbx > 0 || aL;
        const bool corner = (cbx > 0 && cby > 0) || (cbx == 0 && cby > 0 && aL) || (cbx > 0 && cby == 0 && aT) || (cbx == 0 && cby == 0 && aL && aT);
        const bool topright = cby == 0 ? (cbx < 3 ? aT : aTR) : (cbx < 3 && blk_of(cbx + 1, cby - 1) < k);
        uint8_t *const o = rec + (4 * cby + 1) * I4_RS + 4 + 4 * cbx; /* the block's top-left sample in the tile */
        /* ---- the edge vector and the three numbers per lane ---- */
        const int ec = (!topright && e_col > 3) ? 3 : e_col;
        const uint32_t e = o[e_row * I4_RS + ec];
        const uint32_t ep = (uint32_t)__shfl((int)e, grp | ((p + 15) & 15)), en = (uint32_t)__shfl((int)e, grp | ((p + 1) & 15));
        const uint32_t pk = e | (((ep + 2u * e + en + 2u) >> 2) << 8) | (((e + en + 1u) >> 1) << 16);
        /* ---- the nine predictors of this lane's sample ---- */
        int pm[9];
#pragma unroll
        for (int m = 0; m < 9; m++) {
            if (m == 2) continue;
            const uint32_t v = (uint32_t)__shfl((int)pk, grp | (tab[m] & 15));
            pm[m] = (int)((v >> (8 * (tab[m] >> 4))) & 255u);
        }
        {
            const uint32_t part = ((p >= 6 && p <= 9) ? e : 0u) | ((p >= 1 && p <= 4) ? e << 16 : 0u);
            const uint32_t sums = i4_row_sum(part, lane);
            const int sT = (int)(sums & 0xffffu), sL = (int)(sums >> 16);
            pm[2] = (top && left) ? (sT + sL + 4) >> 3 : left ? (sL + 2) >> 2 : top ? (sT + 2) >> 2 : 128;
        }
        /* ---- SADs, the choice ---- */
        const int sv = (int)src[(4 * cby + y) * 16 + 4 * cbx + x];
        uint32_t d[9];
#pragma unroll
        for (int m = 0; m < 9; m++) d[m] = (uint32_t)abs(sv - pm[m]);
        const uint32_t s01 = i4_row_sum(d[0] | (d[1] << 16), lane), s23 = i4_row_sum(d[2] | (d[3] << 16), lane);
        const uint32_t s45 = i4_row_sum(d[4] | (d[5] << 16), lane), s67 = i4_row_sum(d[6] | (d[7] << 16), lane);
        const uint32_t s8 = i4_row_sum(d[8], lane);
        const bool tlc = top && left && corner;
        const uint32_t sad[9] = {top ? s01 & 0xffffu : 0xffffu, left ? s01 >> 16 : 0xffffu, s23 & 0xffffu, top ? s23 >> 16 : 0xffffu,
                                 tlc ? s45 & 0xffffu : 0xffffu, tlc ? s45 >> 16 : 0xffffu, tlc ? s67 & 0xffffu : 0xffffu,
                                 top ? s67 >> 16 : 0xffffu, left ? s8 : 0xffffu};
        uint32_t best = sad[0];
        int bm = 0, pred = pm[0];
#pragma unroll
        for (int m = 1; m < 9; m++)
            if (sad[m] < best) best = sad[m], bm = m, pred = pm[m];
        if (act) total += best;
        /* ---- residual -> coefficient of this lane (row pass, then column pass) ---- */
        const int r0 = sv - pred;
        const int ra = __shfl(r0, grp | (4 * y)), rb = __shfl(r0, grp | (4 * y + 1)), rc = __shfl(r0, grp | (4 * y + 2)), rd = __shfl(r0, grp | (4 * y + 3));
        const int h1 = i4_fwd1d(x, ra, rb, rc, rd);
        const int ca = __shfl(h1, grp | x), cb = __shfl(h1, grp | (4 + x)), cc = __shfl(h1, grp | (8 + x)), cd = __shfl(h1, grp | (12 + x));
        const int w = i4_fwd1d(y, ca, cb, cc, cd);
        const int lvl = venc_quant_mag(w, mf, q.f_intra, q.qbits);
        const uint32_t nzb = (uint32_t)(__ballot(act && lvl != 0) >> grp) & 0xffffu;
        if (act) ((int16_t *)T.lv)[k * 16 + scan] = (int16_t)lvl;
        /* ---- back: dequantisation, inverse transform (rows first, 8.5.12.2), reconstruction ---- */
        const int dq = VENC_MUL24(lvl, vv);
        const int ia = __shfl(dq, grp | (4 * y)), ib = __shfl(dq, grp | (4 * y + 1)), ic = __shfl(dq, grp | (4 * y + 2)), id = __shfl(dq, grp | (4 * y + 3));
        const int f1 = i4_inv1d(x, ia, ib, ic, id);
        const int ja = __shfl(f1, grp | x), jb = __shfl(f1, grp | (4 + x)), jc = __shfl(f1, grp | (8 + x)), jd = __shfl(f1, grp | (12 + x));
        const int rr = (i4_inv1d(y, ja, jb, jc, jd) + 32) >> 6;
        if (act) {
            o[y * I4_RS + x] = (uint8_t)venc_clip_u8(pred + rr);
            if (p == 0) {
                ((uint8_t *)T.nz)[k] = (uint8_t)__popc(nzb);
                ((uint8_t *)T.mode)[k] = (uint8_t)bm;
            }
        }
        /* (the next round reads the tile: LDS operations of a wave are ordered) */
        if (__ballot(total + (uint32_t)__shfl((int)total, lane ^ 16) < best16) == 0ull) break; /* neither macroblock of the wave can still win */
    }
    return total + (uint32_t)__shfl((int)total, lane ^ 16);
}

/* ================================================================== k_intra_modes
 * I pictures with Intra16x16 prediction chosen per macroblock among vertical (0), horizontal (1) and DC (2) by the SAD of
 * the predictor against the source (spec/CODEC.md section 3: smallest SAD, the smaller mode number on ties; vertical
 * needs the macroblock above, horizontal the one to the left, inside the slice).  Vertical and horizontal predictors are
 * not constants of the macroblock, so the linearity the three-kernel path above rests on (the predictor shifts
 * coefficient 0 only) is gone for 7 of a block's 16 coefficients; this kernel is the plain 2-D dependency wavefront
 * instead: one block per slice, half-wave r owns macroblock row r of the band and runs one macroblock behind half-wave
 * r - 1, a macroblock's transform on 24 lanes (mb_code), the reconstructed bottom ROW of every macroblock handed down
 * through a small LDS ring (16 luma + 8 + 8 chroma samples), its right COLUMN handed on in registers.  ~4.5 us per
 * wavefront step against 0.9 for the DC-only chain -- affordable because intra steps run ahead of their unit on a
 * stream of their own (venc_hip.hip, INTRA AHEAD): at 1080p the ~0.6 ms of an I step hide behind six P steps. */
/* I4: Intra4x4 is tried against the best 16x16 predictor (intra_modes = 3).  Its block 5 predicts from the macroblock
 * above-RIGHT, so a row then runs TWO macroblocks behind the row above (one otherwise). */
template <bool I4>
__global__ __launch_bounds__(I4 ? 64 * INTRA4_MAX_WAVES : 1024) void k_intra_modes(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_bot[2 * INTRA_MAX_WAVES][4][8]; /* per macroblock row, ring of 4: bottom row Y[16] Cb[8] Cr[8] */
    __shared__ __attribute__((aligned(16))) i4_tile s_i4[I4 ? 2 * INTRA_MAX_WAVES : 1];
    constexpr int SKEW = I4 ? 2 : 1;
    const int item = blockIdx.z, s = S.slice0 + blockIdx.x, tid = threadIdx.x;
    const int nw = blockDim.x >> 6;
    const int wv = tid >> 6, lane = tid & 63, l32 = lane & 31, hbase = lane & 32;
    const int rw = wv + nw * (lane >> 5); /* each half of a wave owns a macroblock row of the band */
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
        const int steps = g.mbw + SKEW * (nr - 1);
        uint32_t left4 = 0;                      /* luma lanes: the four samples left of this lane's block rows */
        uint32_t lcol[4] = {0, 0, 0, 0};         /* plane prediction: the whole column left of the macroblock, and the corner sample */
        uint32_t corner = 0;
        int leftY = 0, leftC0 = 0, leftC1 = 0;   /* sums of the left column (luma: all 16; chroma: upper / lower half of this lane's plane) */
        const bool cur_al = luma ? ((g.w & 3) == 0) : ((g.w & 7) == 0);
        const int pyl = (luma ? my * 16 : my * 8) + by;
        uint32_t cq[4] = {0, 0, 0, 0};
        if (rw < nr) load_cur_rows(cur, cstride, pw, ph, bx, pyl, cur_al, cq);
        for (int step = 0; step < steps; step++) {
            const int mx = step - SKEW * rw;
            if (rw < nr && mx >= 0 && mx < g.mbw) {
                uint32_t nq[4] = {0, 0, 0, 0};
                if (mx + 1 < g.mbw) load_cur_rows(cur, cstride, pw, ph, (luma ? (mx + 1) * 16 : (mx + 1) * 8) + bx, pyl, cur_al, nq);
                const int mb = my * g.mbw + mx;
                const bool aL = mx > 0, aT = my > r0;
                /* the row above: 8 dwords Y0..Y3 Cb0 Cb1 Cr0 Cr1 */
                uint32_t tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t tr4 = 0; /* (I4) the first four samples of the row above the macroblock to the right */
                const bool aTR = aT && mx + 1 < g.mbw;
                if (aT) {
                    if (rw > 0) {
                        const uint4 a = *(const uint4 *)&s_bot[rw - 1][mx & 3][0], c4 = *(const uint4 *)&s_bot[rw - 1][mx & 3][4];
                        tb[0] = a.x, tb[1] = a.y, tb[2] = a.z, tb[3] = a.w, tb[4] = c4.x, tb[5] = c4.y, tb[6] = c4.z, tb[7] = c4.w;
                        if (I4 && aTR) tr4 = s_bot[rw - 1][(mx + 1) & 3][0];
                    } else {
                        /* the row above belongs to the previous chunk of this slice: its bottom lines come from memory */
                        int ys, us;
                        const uint8_t *ry = rec_plane(B, g, S.parity, item, 0, ys) + (size_t)(my * 16 - 1) * ys + mx * 16;
                        const uint8_t *ru = rec_plane(B, g, S.parity, item, 1, us) + (size_t)(my * 8 - 1) * us + mx * 8;
                        const uint8_t *rv = rec_plane(B, g, S.parity, item, 2, us) + (size_t)(my * 8 - 1) * us + mx * 8;
                        const uint4 a = *(const uint4 *)ry;
                        if (I4 && aTR) tr4 = *(const uint32_t *)(ry + 16);
                        const uint2 cu = *(const uint2 *)ru, cv = *(const uint2 *)rv;
                        tb[0] = a.x, tb[1] = a.y, tb[2] = a.z, tb[3] = a.w, tb[4] = cu.x, tb[5] = cu.y, tb[6] = cv.x, tb[7] = cv.y;
                    }
                }
                const int t0 = (int)__builtin_amdgcn_sad_u8(tb[3], 0u, __builtin_amdgcn_sad_u8(tb[2], 0u, __builtin_amdgcn_sad_u8(tb[1], 0u, __builtin_amdgcn_sad_u8(tb[0], 0u, 0u))));
                const uint32_t top4 = bx == 0 ? tb[0] : bx == 4 ? tb[1] : bx == 8 ? tb[2] : tb[3]; /* (luma lanes) the four samples above this lane's block columns */
                int pred; /* the DC predictor of this lane's plane / block */
                if (luma) {
                    pred = (aL && aT) ? (leftY + t0 + 16) >> 5 : aL ? (leftY + 8) >> 4 : aT ? (t0 + 8) >> 4 : 128;
                } else {
                    const uint32_t tc = tb[4 + 2 * (plane - 1) + (k & 1)];
                    const int sT = (int)__builtin_amdgcn_sad_u8(tc, 0u, 0u);
                    const int sL = (k >> 1) ? leftC1 : leftC0;
                    bool useT = aT, useL = aL;
                    if (k == 1 && aT) useL = false;
                    if (k == 2 && aL) useT = false;
                    pred = (useT && useL) ? (sT + sL + 4) >> 3 : useT ? (sT + 2) >> 2 : useL ? (sL + 2) >> 2 : 128;
                }
                /* mode decision over the 16 luma lanes: SAD of the source against each predictor */
                uint32_t sV = 0, sH = 0, sD = 0;
                if (luma) {
                    const uint32_t d4 = (uint32_t)pred * 0x01010101u;
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        sV = __builtin_amdgcn_sad_u8(cq[y], top4, sV);
                        sH = __builtin_amdgcn_sad_u8(cq[y], ((left4 >> (8 * y)) & 255u) * 0x01010101u, sH);
                        sD = __builtin_amdgcn_sad_u8(cq[y], d4, sD);
                    }
                }
                /* plane prediction (8.3.3.4; intra_modes = 2): every luma lane holds the row above (tb), the column to the left
                 * (lcol) and the corner, so the two gradients need no exchange; the lane predicts its own 4x4 block */
                uint32_t sP = 0;
                int pl[16];
#pragma unroll
                for (int i = 0; i < 16; i++) pl[i] = 0;
                const bool try_plane = g.intra_plane && aL && aT; /* (uniform per half-wave row; lanes of both rows run the code) */
                if (g.intra_plane) {
                    int H = 0, V = 0;
#pragma unroll
                    for (int kk = 0; kk < 8; kk++) {
                        const int hi = 8 + kk, lo = 6 - kk;
                        const int th = (int)((tb[hi >> 2] >> (8 * (hi & 3))) & 255u), tl = lo >= 0 ? (int)((tb[lo >> 2] >> (8 * (lo & 3))) & 255u) : (int)corner;
                        const int lh = (int)((lcol[hi >> 2] >> (8 * (hi & 3))) & 255u), ll = lo >= 0 ? (int)((lcol[lo >> 2] >> (8 * (lo & 3))) & 255u) : (int)corner;
                        H += (kk + 1) * (th - tl), V += (kk + 1) * (lh - ll);
                    }
                    const int pa = 16 * ((int)(lcol[3] >> 24) + (int)(tb[3] >> 24)), pb = (5 * H + 32) >> 6, pc = (5 * V + 32) >> 6;
                    if (luma) {
#pragma unroll
                        for (int y = 0; y < 4; y++) {
                            uint32_t row = 0;
#pragma unroll
                            for (int x = 0; x < 4; x++) {
                                pl[4 * y + x] = venc_clip_u8((pa + pb * (bx + x - 7) + pc * (by + y - 7) + 16) >> 5);
                                row |= (uint32_t)pl[4 * y + x] << (8 * x);
                            }
                            sP = __builtin_amdgcn_sad_u8(cq[y], row, sP);
                        }
                    }
                    sP = half_sum32(sP, lane);
                }
                sV = half_sum32(sV, lane), sH = half_sum32(sH, lane), sD = half_sum32(sD, lane);
                int mode = 2;
                uint32_t best = sD;
                if (aL && sH <= best) best = sH, mode = 1;
                if (aT && sV <= best) best = sV, mode = 0;
                if (try_plane && sP < best) best = sP, mode = 3; /* (ties go to the smaller mode number) */
                /* Intra4x4 against it: the trial runs in the half-wave's tile; it wins iff its blocks' SADs sum to strictly less */
                bool use_i4 = false;
                i4_tile &T4 = s_i4[I4 ? (tid >> 5) : 0];
                if (I4) {
                    uint8_t *tr = (uint8_t *)T4.rec;
                    if (l32 < 4) T4.rec[1 + l32] = tb[l32];
                    if (l32 == 4) T4.rec[5] = tr4;
                    if (l32 == 5) tr[3] = (uint8_t)corner;
                    if (luma) {
                        tr[(l32 + 1) * I4_RS + 3] = (uint8_t)((lcol[l32 >> 2] >> (8 * (l32 & 3))) & 255u);
#pragma unroll
                        for (int y = 0; y < 4; y++) T4.src[(by + y) * 4 + (bx >> 2)] = cq[y];
                    }
                    const uint32_t tot4 = i4_trial(T4, g.ql, lane, aL, aT, aTR, best);
                    use_i4 = tot4 < best;
                }
                const int px = (luma ? mx * 16 : mx * 8) + bx, py = (luma ? my * 16 : my * 8) + by;
                int c[16], p[16];
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        c[4 * y + x] = (int)((cq[y] >> (8 * x)) & 255u);
                        p[4 * y + x] = !luma || mode == 2 ? pred : mode == 3 ? pl[4 * y + x] : mode == 0 ? (int)((top4 >> (8 * x)) & 255u) : (int)((left4 >> (8 * y)) & 255u);
                    }
#pragma unroll
                for (int y = 0; y < 4; y++) cq[y] = nq[y];
                mb_lane_out o;
                mb_code<true, true>(lane, c, p, g, o);
                if (I4 && use_i4 && luma) { /* the trial's results take the place of the 16x16 coding's (chroma is the same either way) */
                    o.nnz = (int)((const uint8_t *)T4.nz)[l32];
                    o.dclv = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t wd = T4.lv[l32 * 8 + i];
                        o.lv[2 * i] = (int)(int16_t)(wd & 0xffffu), o.lv[2 * i + 1] = (int)(int16_t)(wd >> 16);
                    }
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        const uint32_t wd = T4.rec[(by + y + 1) * 6 + 1 + (bx >> 2)];
                        o.rec[4 * y] = (int)(wd & 255u), o.rec[4 * y + 1] = (int)((wd >> 8) & 255u), o.rec[4 * y + 2] = (int)((wd >> 16) & 255u), o.rec[4 * y + 3] = (int)(wd >> 24);
                    }
                }

                int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
                venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
                mb_store(l32, true, o, lv_mb, info, rec, wstride, px, py);
                if (I4 && use_i4) { /* slot 16 holds no DC block: the sixteen Intra4x4PredModes, by luma4x4BlkIdx, for the entropy coder */
                    if (luma) lv_mb[16 * 16 + l32] = (int16_t)((const uint8_t *)T4.mode)[l32];
                } else if (luma) { /* luma DC record (slot 16, zig-zag order) */
                    const int rp = blk_y(l32) * 4 + blk_x(l32);
                    int zi = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (venc_zigzag_ct(i) == rp) zi = i;
                    lv_mb[16 * 16 + zi] = (int16_t)o.dclv;
                }
                const int dc0 = __shfl(o.dclv, hbase | (l32 & 28) | 0), dc1 = __shfl(o.dclv, hbase | (l32 & 28) | 1);
                const int dc2 = __shfl(o.dclv, hbase | (l32 & 28) | 2), dc3 = __shfl(o.dclv, hbase | (l32 & 28) | 3);
                if (l32 == 16 || l32 == 20) {
                    uint4 *dst = (uint4 *)(lv_mb + (l32 == 16 ? 17 : 18) * 16);
                    dst[0] = make_uint4(((uint32_t)dc0 & 0xffffu) | ((uint32_t)dc1 << 16), ((uint32_t)dc2 & 0xffffu) | ((uint32_t)dc3 << 16), 0, 0);
                }
                const uint32_t hnz = (uint32_t)(__ballot(o.nnz > 0) >> hbase), hdc = (uint32_t)(__ballot(o.dclv != 0) >> hbase);
                if (l32 == 0) {
                    int cbpl = (hnz & 0xffffu) ? 15 : 0;
                    if (I4 && use_i4) cbpl = ((hnz & 0xfu) ? 1 : 0) | ((hnz & 0xf0u) ? 2 : 0) | ((hnz & 0xf00u) ? 4 : 0) | ((hnz & 0xf000u) ? 8 : 0); /* (lane = luma4x4BlkIdx: four lanes per 8x8) */
                    info->cbp = (uint8_t)cbpl;
                    info->cbpc = (uint8_t)(((hnz >> 16) & 0xffu) ? 2 : ((hdc >> 16) & 0xffu) ? 1 : 0);
                    info->flags = (uint8_t)((I4 && use_i4) ? (1 | 8) : (1 | (mode << 1))); /* bit 0: intra; bits 1-2: Intra16x16PredMode; bit 3: Intra4x4 */
                    B.mvsad[(size_t)item * g.nmb + mb] = 0;
                }
                /* reconstructed edges for the neighbours: bottom row to the ring, right column to the next step */
                const uint32_t pbq = (uint32_t)o.rec[12] | ((uint32_t)o.rec[13] << 8) | ((uint32_t)o.rec[14] << 16) | ((uint32_t)o.rec[15] << 24);
                const uint32_t prq = (uint32_t)o.rec[3] | ((uint32_t)o.rec[7] << 8) | ((uint32_t)o.rec[11] << 16) | ((uint32_t)o.rec[15] << 24);
                if (luma ? by == 12 : (l32 < 24 && by == 4)) s_bot[rw][mx & 3][luma ? (bx >> 2) : 4 + 2 * (plane - 1) + (k & 1)] = pbq;
                const int pr = o.rec[3] + o.rec[7] + o.rec[11] + o.rec[15];
                const uint32_t c0 = (uint32_t)__shfl((int)prq, hbase | blk_of(3, 0)), c1 = (uint32_t)__shfl((int)prq, hbase | blk_of(3, 1));
                const uint32_t c2 = (uint32_t)__shfl((int)prq, hbase | blk_of(3, 2)), c3 = (uint32_t)__shfl((int)prq, hbase | blk_of(3, 3));
                left4 = by == 0 ? c0 : by == 4 ? c1 : by == 8 ? c2 : c3;
                lcol[0] = c0, lcol[1] = c1, lcol[2] = c2, lcol[3] = c3;
                corner = tb[3] >> 24; /* the last sample above this macroblock is the corner of the next */
                leftY = (int)__builtin_amdgcn_sad_u8(c3, 0u, __builtin_amdgcn_sad_u8(c2, 0u, __builtin_amdgcn_sad_u8(c1, 0u, __builtin_amdgcn_sad_u8(c0, 0u, 0u))));
                const int rcb0 = __shfl(pr, hbase | 17), rcb1 = __shfl(pr, hbase | 19), rcr0 = __shfl(pr, hbase | 21), rcr1 = __shfl(pr, hbase | 23);
                leftC0 = plane == 2 ? rcr0 : rcb0;
                leftC1 = plane == 2 ? rcr1 : rcb1;
            }
            /* LDS-only barrier (no wait for the global stores and prefetches of the step) */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        __syncthreads(); /* the next chunk re-reads this chunk's last reconstructed row from memory */
    }
}
