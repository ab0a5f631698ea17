/*
 * venc_synth.h -- deterministic, integer-only synthetic YUV420p source
 * (BASELINE.md section 3).  Counter-based (every pixel is
 * a pure function of plane, x, y, frame, seed) so the same code runs per pixel
 * on the host (venc_synth_frame) and on the GPU (k_synth) and produces
 * identical bytes on every machine.  Content: 8x8-px value-noise texture
 * panning (+3,+1) px/frame, four textured rectangles bouncing at speeds up to
 * 13 px/frame (inside the +-16 search range), +-2 LSB per-pixel noise.
 */
#ifndef VENC_SYNTH_H
#define VENC_SYNTH_H
#include <stdint.h>
#if defined(__HIPCC__)
#define VENC_SYN __host__ __device__ __forceinline__
#else
#define VENC_SYN static inline
#endif

VENC_SYN uint32_t venc_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
/* value noise, cell = 2^cl pixels, range 16..235 */
VENC_SYN int venc_tex(uint32_t seed, uint32_t X, uint32_t Y, int cl) {
    const uint32_t cs = 1u << cl, i = X >> cl, j = Y >> cl, fx = X & (cs - 1), fy = Y & (cs - 1);
    const uint32_t k = seed * 0x9E3779B1U;
    const uint32_t l00 = 16 + venc_hash32(k + (i & 1023) + ((j & 1023) << 10)) % 220;
    const uint32_t l10 = 16 + venc_hash32(k + ((i + 1) & 1023) + ((j & 1023) << 10)) % 220;
    const uint32_t l01 = 16 + venc_hash32(k + (i & 1023) + (((j + 1) & 1023) << 10)) % 220;
    const uint32_t l11 = 16 + venc_hash32(k + ((i + 1) & 1023) + (((j + 1) & 1023) << 10)) % 220;
    const uint32_t v = (cs - fx) * (cs - fy) * l00 + fx * (cs - fy) * l10 + (cs - fx) * fy * l01 + fx * fy * l11;
    return (int)((v + (1u << (2 * cl - 1))) >> (2 * cl));
}
VENC_SYN uint32_t venc_tri(uint32_t p, uint32_t r) { /* triangle wave in [0, r] */
    if (r == 0) return 0;
    const uint32_t q = p % (2 * r);
    return q < r ? q : 2 * r - q;
}
/* plane 0: (x,y) luma coordinates; planes 1,2: chroma coordinates */
VENC_SYN uint8_t venc_synth_pixel(int plane, int x, int y, int t, int w, int h, uint32_t seed) {
    const uint32_t lx = plane ? 2u * (uint32_t)x : (uint32_t)x, ly = plane ? 2u * (uint32_t)y : (uint32_t)y;
    const uint32_t ut = (uint32_t)t;
    int v = venc_tex(seed + (uint32_t)plane, lx + 3u * ut, ly + ut, plane ? 4 : 3);
    const uint32_t rw = (uint32_t)(w / 30 > 16 ? w / 30 : 16), rh = (uint32_t)(h * 2 / 45 > 12 ? h * 2 / 45 : 12);
    const uint32_t rx = (uint32_t)w > rw ? (uint32_t)w - rw : 0u, ry = (uint32_t)h > rh ? (uint32_t)h - rh : 0u;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t vx = k == 0 ? 5u : k == 1 ? 11u : k == 2 ? 7u : 13u;
        const uint32_t vy = k == 0 ? 2u : k == 1 ? 7u : k == 2 ? 5u : 3u;
        const uint32_t p0x = rx ? venc_hash32(seed + 100u + k) % rx : 0u, p0y = ry ? venc_hash32(seed + 200u + k) % ry : 0u;
        const uint32_t px = venc_tri(p0x + vx * ut, rx), py = venc_tri(p0y + vy * ut, ry);
        if (lx >= px && lx < px + rw && ly >= py && ly < py + rh)
            v = venc_tex(seed + 10u + 3u * k + (uint32_t)plane, lx - px + 97u * k + 31u, ly - py + 61u * k + 17u, plane ? 4 : 3);
    }
    const uint32_t idx = (uint32_t)y * (uint32_t)(plane ? w / 2 : w) + (uint32_t)x;
    const uint32_t n = venc_hash32((seed ^ 0x9E3779B9U) + ut * 0x85EBCA6BU + (uint32_t)plane * 0xC2B2AE35U + idx);
    v += plane ? (int)(n % 3u) - 1 : (int)(n % 5u) - 2;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}
#endif
