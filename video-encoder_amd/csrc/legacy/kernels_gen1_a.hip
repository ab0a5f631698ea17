/*
 * legacy/kernels_gen1_a.hip -- first-generation kernels cut out of venc_kernels.hip (first block)
 * TEST-ONLY: compiled into libvenc_hip_legacy.so (-DVENC_LEGACY_KERNELS), the on-GPU oracles of tests/test_gpu_parity.py;
 * the product library does not contain a line of this file.  Included by venc_kernels.hip.
 */
/* ================================================================== k_me
 * One block = 8 horizontally adjacent macroblocks of one MB row; one half-wave
 * (32 lanes) per macroblock.  Lane l of a half-wave owns displacement column
 * dx = l - 16 and keeps 33 accumulators (one per dy) in VGPRs.  Each of the 48
 * window rows is fetched from LDS once per lane (5 dwords + 4 v_alignbyte) and
 * reused for up to 16 (cur row, dy) pairs with v_sad_u8, so the inner loop is
 * ~88 % SAD instructions.  The dx = +16 column is swept separately.  The argmin
 * uses the total-order key (sad << 11) | rank (spec/CODEC.md section ME).
 */
#define ME_TILE 8
#define ME_WDW (ME_TILE * 4 + 8) /* window row in dwords: 160 bytes */

/* exhaustive search of one macroblock by one half-wave; returns the minimal key, identical in all 32 lanes */
DEVFN uint32_t me_dense_half(const uint32_t *s_win, const uint32_t *s_cur, const uint16_t *s_rank, int mbi, int l32) {
    uint32_t c[16][4];
#pragma unroll
    for (int y = 0; y < 16; y++)
#pragma unroll
        for (int j = 0; j < 4; j++) c[y][j] = s_cur[y * (ME_TILE * 4) + mbi * 4 + j];

    uint32_t acc[33];
#pragma unroll
    for (int i = 0; i < 33; i++) acc[i] = 0;

    const int boff = mbi * 16 + l32; /* window byte column of displacement dx = l32 - 16 */
    const int dw = boff >> 2;
    const uint32_t sh = (uint32_t)(boff & 3);
#pragma unroll
    for (int r = 0; r < 48; r++) {
        const uint32_t *row = s_win + r * ME_WDW + dw;
        uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3], d4 = row[4];
        uint32_t r0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        uint32_t r1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        uint32_t r2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
        uint32_t r3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
#pragma unroll
        for (int y = 0; y < 16; y++) {
            const int dyi = r - y; /* dy + 16 */
            if (dyi >= 0 && dyi <= 32) {
                uint32_t a = acc[dyi];
                a = __builtin_amdgcn_sad_u8(r0, c[y][0], a);
                a = __builtin_amdgcn_sad_u8(r1, c[y][1], a);
                a = __builtin_amdgcn_sad_u8(r2, c[y][2], a);
                a = __builtin_amdgcn_sad_u8(r3, c[y][3], a);
                acc[dyi] = a;
            }
        }
    }
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int dyi = 0; dyi < 33; dyi++) {
        uint32_t key = (acc[dyi] << 11) | s_rank[dyi * 33 + l32];
        best = min(best, key);
    }
    /* displacement column dx = +16 (window byte column mbi*16 + 32: dword aligned) */
    for (int dyi = l32; dyi < 33; dyi += 32) {
        uint32_t a = 0;
#pragma unroll
        for (int y = 0; y < 16; y++) {
            const uint32_t *row = s_win + (y + dyi) * ME_WDW + mbi * 4 + 8;
            a = __builtin_amdgcn_sad_u8(row[0], c[y][0], a);
            a = __builtin_amdgcn_sad_u8(row[1], c[y][1], a);
            a = __builtin_amdgcn_sad_u8(row[2], c[y][2], a);
            a = __builtin_amdgcn_sad_u8(row[3], c[y][3], a);
        }
        uint32_t key = (a << 11) | s_rank[dyi * 33 + 32];
        best = min(best, key);
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, m));
    return best;
}

__global__ __launch_bounds__(256) void k_me_dense(venc_geom g, venc_bufs B, venc_step S) {
    __shared__ uint32_t s_win[48 * ME_WDW];
    __shared__ uint32_t s_cur[16 * ME_TILE * 4];
    __shared__ uint16_t s_rank[VENC_ME_CANDS];

    const int item = blockIdx.z, my = S.row0 + blockIdx.y, tx = blockIdx.x, tid = threadIdx.x;
    const int x0 = tx * ME_TILE * 16, y0 = my * 16;
    int rstride;
    const uint8_t *ref = rec_plane(B, g, S.ref, item, 0, rstride);
    int cstride, pw, ph;
    const uint8_t *cur = in_plane(B, g, S.in_slot[item], 0, cstride, pw, ph);

    for (int i = tid; i < VENC_ME_CANDS; i += 256) s_rank[i] = B.me_rank[i];

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
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int mbi = wave * 2 + half;
    const int mx = tx * ME_TILE + mbi;

    const uint32_t best = me_dense_half(s_win, s_cur, s_rank, mbi, l32);
    if (l32 == 0 && mx < g.mbw) {
        uint32_t cand = B.me_cand[best & 2047u];
        int dx = (int)(cand & 255u) - 16, dy = (int)(cand >> 8) - 16;
        B.mvsad[(size_t)item * g.nmb + my * g.mbw + mx] =
            ((best >> 11) << 16) | (((uint32_t)dy & 255u) << 8) | ((uint32_t)dx & 255u);
    }
}

