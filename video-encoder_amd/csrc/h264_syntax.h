/*
 * h264_syntax.h -- host-side (C99) bitstream packaging of the engine: parameter
 * sets, NAL headers, emulation prevention and the 4-byte length prefixes the
 * reference's muxer expects (av_encode.c:635-640, :692-705, :722-727).
 * SURVEY.md section 8a row a4.6.
 */
#ifndef H264_SYNTAX_H
#define H264_SYNTAX_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int width, height; /* display size */
    int qp;
    int fps_num, fps_den;
    int sar_w, sar_h;
} h264_seq;

int h264_level_idc(int mbs, int fps_num, int fps_den);
/* RBSP writers; return bytes written (0 if cap is too small) */
size_t h264_write_sps(const h264_seq *s, uint8_t *rbsp, size_t cap);
size_t h264_write_pps(const h264_seq *s, uint8_t *rbsp, size_t cap);
/* worst-case size of a wrapped NAL for n RBSP bytes */
static inline size_t h264_nal_bound(size_t n) { return 4 + 1 + n + n / 2 + 4; }
/* Writes [prefix(4)] [nal header] [escaped rbsp] to dst, returns total bytes.
 * annexb = 0: prefix is the big-endian NAL size; 1: 00 00 00 01 start code. */
size_t h264_wrap_nal(uint8_t *dst, int annexb, int ref_idc, int type, const uint8_t *rbsp, size_t n);

#endif
