/*
 * legacy/kernels_gen1_b.hip -- first-generation kernels cut out of venc_kernels.hip (second block)
 * TEST-ONLY: compiled into libvenc_hip_legacy.so (-DVENC_LEGACY_KERNELS), the on-GPU oracles of tests/test_gpu_parity.py;
 * the product library does not contain a line of this file.  Included by venc_kernels.hip.
 */
/* ================================================================== k_inter
 * P_L0_16x16: half-wave per macroblock, 8 macroblocks per block. */
__global__ __launch_bounds__(256) void k_inter(venc_geom g, venc_bufs B, venc_step S) {
    const int item = blockIdx.z, tid = threadIdx.x;
    const int lane = tid & 63, l32 = lane & 31, base = lane & 32;
    const int mb_raw = S.mb0 + blockIdx.x * 8 + (tid >> 5);
    const bool valid = mb_raw < S.mb1;
    const int mb = valid ? mb_raw : S.mb1 - 1;
    const int mx = mb % g.mbw, my = mb / g.mbw;
    const bool luma = l32 < 16;
    const int plane = luma ? 0 : 1 + (((l32 - 16) >> 2) & 1);
    const int k = l32 & 3;
    const int bx = luma ? blk_x(l32) * 4 : (k & 1) * 4, by = luma ? blk_y(l32) * 4 : (k >> 1) * 4;
    const int px = (luma ? mx * 16 : mx * 8) + bx, py = (luma ? my * 16 : my * 8) + by;

    const uint32_t mvs = B.mvsad[(size_t)item * g.nmb + mb];
    const int dx = (int)(int8_t)(mvs & 255u), dy = (int)(int8_t)((mvs >> 8) & 255u);

    int cstride, pw, ph, rstride, wstride;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], plane, cstride, pw, ph);
    const uint8_t *ref = rec_plane(B, g, S.ref, item, plane, rstride);
    uint8_t *rec = rec_plane(B, g, S.parity, item, plane, wstride);
    const int rw = luma ? g.cw : g.cw >> 1, rh = luma ? g.ch : g.ch >> 1;

    int c[16], p[16];
    /* current block: aligned dword rows when the plane geometry allows it, clamped bytes at the display edges */
    const bool cur_fast = ((g.w & 7) == 0) && (px + 4 <= pw) && (py + 4 <= ph);
    if (cur_fast) {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t v = *(const uint32_t *)(cur + (size_t)(py + y) * cstride + px);
            c[4 * y] = (int)(v & 255u), c[4 * y + 1] = (int)((v >> 8) & 255u);
            c[4 * y + 2] = (int)((v >> 16) & 255u), c[4 * y + 3] = (int)(v >> 24);
        }
    } else {
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) c[4 * y + x] = cur[(size_t)min(py + y, ph - 1) * cstride + min(px + x, pw - 1)];
    }
    /* prediction: the reference block sits at an arbitrary byte offset; interior blocks fetch two
     * aligned dwords per row and realign with v_alignbyte, blocks touching the frame edge clamp per sample */
    const int xi = luma ? dx : dx >> 1, yi = luma ? dy : dy >> 1;
    const int x0 = px + xi, y0 = py + yi;
    const bool ref_fast = x0 >= 0 && x0 + 8 <= rw && y0 >= 0 && y0 + 5 <= rh;
    if (luma) {
        if (ref_fast) {
            const uint32_t sh = (uint32_t)(x0 & 3);
            const uint8_t *rp = ref + (size_t)y0 * rstride + (x0 & ~3);
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const uint32_t *rw32 = (const uint32_t *)(rp + (size_t)y * rstride);
                const uint32_t v = __builtin_amdgcn_alignbyte(rw32[1], rw32[0], sh);
                p[4 * y] = (int)(v & 255u), p[4 * y + 1] = (int)((v >> 8) & 255u);
                p[4 * y + 2] = (int)((v >> 16) & 255u), p[4 * y + 3] = (int)(v >> 24);
            }
        } else {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                    p[4 * y + x] = ref[(size_t)clampi(y0 + y, 0, rh - 1) * rstride + clampi(x0 + x, 0, rw - 1)];
        }
    } else {
        /* chroma MC, 8.4.2.2.2: integer luma mv -> eighth-sample chroma mv (frac 0 or 4) */
        const int xf = (dx & 1) * 4, yf = (dy & 1) * 4;
        int smp[5][5];
        if (ref_fast) {
            const uint32_t sh = (uint32_t)(x0 & 3);
            const uint8_t *rp = ref + (size_t)y0 * rstride + (x0 & ~3);
#pragma unroll
            for (int y = 0; y < 5; y++) {
                const uint32_t *rw32 = (const uint32_t *)(rp + (size_t)y * rstride);
                const uint32_t d0 = rw32[0], d1 = rw32[1];
                const uint32_t v = __builtin_amdgcn_alignbyte(d1, d0, sh);
                const uint32_t e = __builtin_amdgcn_alignbyte(0u, d1, sh); /* byte x0+4 in its low byte */
                smp[y][0] = (int)(v & 255u), smp[y][1] = (int)((v >> 8) & 255u);
                smp[y][2] = (int)((v >> 16) & 255u), smp[y][3] = (int)(v >> 24);
                smp[y][4] = (int)(e & 255u);
            }
        } else {
#pragma unroll
            for (int y = 0; y < 5; y++)
#pragma unroll
                for (int x = 0; x < 5; x++)
                    smp[y][x] = ref[(size_t)clampi(y0 + y, 0, rh - 1) * rstride + clampi(x0 + x, 0, rw - 1)];
        }
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                p[4 * y + x] = ((8 - xf) * (8 - yf) * smp[y][x] + xf * (8 - yf) * smp[y][x + 1] + (8 - xf) * yf * smp[y + 1][x] +
                                xf * yf * smp[y + 1][x + 1] + 32) >> 6;
    }
    mb_lane_out o;
    mb_code<false, false>(lane, c, p, g, o);

    int16_t *lv_mb = B.levels + ((size_t)item * g.nmb + mb) * VENC_MB_COEFS;
    venc_mbinfo *info = B.mbinfo + (size_t)item * g.nmb + mb;
    mb_store(l32, valid, o, lv_mb, info, rec, wstride, px, py);

    /* chroma DC records (slots 17, 18) and the empty luma-DC slot 16 */
    int dc0 = __shfl(o.dclv, base | (l32 & 28) | 0), dc1 = __shfl(o.dclv, base | (l32 & 28) | 1);
    int dc2 = __shfl(o.dclv, base | (l32 & 28) | 2), dc3 = __shfl(o.dclv, base | (l32 & 28) | 3);
    if (valid && (l32 == 16 || l32 == 20)) { /* slot 16 (Intra16x16 luma DC) is never read for inter macroblocks */
        uint4 *dst = (uint4 *)(lv_mb + (l32 == 16 ? 17 : 18) * 16);
        uint32_t a = ((uint32_t)dc0 & 0xffffu) | ((uint32_t)dc1 << 16);
        uint32_t b = ((uint32_t)dc2 & 0xffffu) | ((uint32_t)dc3 << 16);
        dst[0] = make_uint4(a, b, 0, 0);
    }
    /* coded_block_pattern from wave ballots */
    const unsigned long long bal_nz = __ballot(o.nnz > 0);
    const unsigned long long bal_dc = __ballot(o.dclv != 0);
    const uint32_t hnz = (uint32_t)(bal_nz >> base), hdc = (uint32_t)(bal_dc >> base);
    if (valid && l32 == 0) {
        int cbpl = 0;
#pragma unroll
        for (int qd = 0; qd < 4; qd++)
            if ((hnz >> (4 * qd)) & 15u) cbpl |= 1 << qd;
        int cbpc = ((hnz >> 16) & 0xffu) ? 2 : ((hdc >> 16) & 0xffu) ? 1 : 0;
        info->cbp = (uint8_t)cbpl;
        info->cbpc = (uint8_t)cbpc;
        info->flags = 0;
    }
}

