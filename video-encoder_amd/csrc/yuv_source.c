/*
 * yuv_source.c -- raw I420 / YUV4MPEG2 frame source (include/venc_source.h; SURVEY.md section 8f
 * row f2; replaces the reference's pull-and-copy at av_encode.c:525-560).  Plain C99.
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/venc_source.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct venc_source {
    FILE *f;
    int y4m;
    int width, height;
    int fps_num, fps_den, sar_num, sar_den;
    long frames_read;
    long frame_count; /* from the file size, -1 if unknown */
};

static int parse_ratio(const char *s, int *a, int *b) {
    int x = 0, y = 0;
    if (sscanf(s, "%d:%d", &x, &y) != 2 || x < 0 || y < 0) return -1;
    *a = x, *b = y;
    return 0;
}

/* reads one header line (terminated by '\n', at most cap - 1 bytes); returns its length or -1 */
static int read_line(FILE *f, char *buf, int cap) {
    int n = 0;
    for (;;) {
        const int c = fgetc(f);
        if (c == EOF) return n ? -2 : -1;
        if (c == '\n') break;
        if (n + 1 >= cap) return -2;
        buf[n++] = (char)c;
    }
    buf[n] = 0;
    return n;
}

venc_source *venc_source_open(const char *filename, int width, int height) {
    if (!filename) return NULL;
    venc_source *s = (venc_source *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->f = fopen(filename, "rb");
    if (!s->f) {
        fprintf(stderr, "source: cannot open %s\n", filename);
        free(s);
        return NULL;
    }
    s->fps_num = 30, s->fps_den = 1, s->sar_num = 1, s->sar_den = 1;
    if (width > 0 && height > 0) {
        s->width = width, s->height = height;
    } else {
        char line[256];
        const int n = read_line(s->f, line, (int)sizeof(line));
        if (n < 10 || strncmp(line, "YUV4MPEG2", 9) != 0) {
            fprintf(stderr, "source: %s is not a YUV4MPEG2 stream (give a size for headerless I420)\n", filename);
            goto fail;
        }
        s->y4m = 1;
        for (char *tok = strtok(line + 9, " "); tok; tok = strtok(NULL, " ")) {
            switch (tok[0]) {
                case 'W': s->width = atoi(tok + 1); break;
                case 'H': s->height = atoi(tok + 1); break;
                case 'F':
                    if (parse_ratio(tok + 1, &s->fps_num, &s->fps_den) != 0 || s->fps_num == 0 || s->fps_den == 0) s->fps_num = 30, s->fps_den = 1;
                    break;
                case 'A':
                    if (parse_ratio(tok + 1, &s->sar_num, &s->sar_den) != 0 || s->sar_num == 0 || s->sar_den == 0) s->sar_num = 1, s->sar_den = 1;
                    break;
                case 'I':
                    if (tok[1] != 'p' && tok[1] != '?') {
                        fprintf(stderr, "source: interlaced YUV4MPEG2 (%s) is not supported\n", tok);
                        goto fail;
                    }
                    break;
                case 'C':
                    if (strcmp(tok, "C420") && strcmp(tok, "C420jpeg") && strcmp(tok, "C420mpeg2") && strcmp(tok, "C420paldv")) {
                        fprintf(stderr, "source: colour space %s is not 8-bit 4:2:0\n", tok + 1);
                        goto fail;
                    }
                    break;
                default: break; /* X... comments */
            }
        }
    }
    if (s->width <= 0 || s->height <= 0 || (s->width & 1) || (s->height & 1)) {
        fprintf(stderr, "source: unsupported picture size %dx%d (both must be even)\n", s->width, s->height);
        goto fail;
    }
    {
        /* frames = payload bytes / (frame bytes + the 6 bytes of a plain "FRAME\n" line): exact for files written without
         * per-frame parameters, an estimate otherwise (it only feeds a progress percentage) */
        s->frame_count = -1;
        const long here = ftell(s->f);
        if (here >= 0 && fseek(s->f, 0, SEEK_END) == 0) {
            const long end = ftell(s->f);
            const long per = (long)s->width * s->height * 3 / 2 + (s->y4m ? 6 : 0);
            if (end >= here && per > 0) s->frame_count = (end - here) / per;
            if (fseek(s->f, here, SEEK_SET) != 0) goto fail;
        }
    }
    return s;
fail:
    fclose(s->f);
    free(s);
    return NULL;
}

void venc_source_close(venc_source *s) {
    if (!s) return;
    fclose(s->f);
    free(s);
}
long venc_source_frame_count(const venc_source *s) { return s->frame_count; }
int venc_source_width(const venc_source *s) { return s->width; }
int venc_source_height(const venc_source *s) { return s->height; }
void venc_source_rate(const venc_source *s, int *n, int *d) {
    if (n) *n = s->fps_num;
    if (d) *d = s->fps_den;
}
void venc_source_aspect(const venc_source *s, int *n, int *d) {
    if (n) *n = s->sar_num;
    if (d) *d = s->sar_den;
}

/* rows of `w` bytes into a plane with `stride` bytes per row (the reference's sws_scale call is this copy) */
static int read_plane(FILE *f, uint8_t *dst, int stride, int w, int h) {
    if (stride == w) return fread(dst, 1, (size_t)w * (size_t)h, f) == (size_t)w * (size_t)h ? 0 : -1;
    for (int y = 0; y < h; y++)
        if (fread(dst + (size_t)y * (size_t)stride, 1, (size_t)w, f) != (size_t)w) return -1;
    return 0;
}

int venc_source_read(venc_source *s, x264_picture_t *pic) {
    if (!s || !pic || pic->img.i_plane < 3 || !pic->img.plane[0] || pic->img.i_stride[0] < s->width ||
        pic->img.i_stride[1] < s->width / 2 || pic->img.i_stride[2] < s->width / 2)
        return -1;
    if (s->y4m) {
        char line[128];
        const int n = read_line(s->f, line, (int)sizeof(line));
        if (n == -1) return 0; /* clean end of stream */
        if (n < 5 || strncmp(line, "FRAME", 5) != 0) {
            fprintf(stderr, "source: frame %ld: missing FRAME header\n", s->frames_read);
            return -1;
        }
    } else {
        const int c = fgetc(s->f);
        if (c == EOF) return 0;
        ungetc(c, s->f);
    }
    if (read_plane(s->f, pic->img.plane[0], pic->img.i_stride[0], s->width, s->height) != 0 ||
        read_plane(s->f, pic->img.plane[1], pic->img.i_stride[1], s->width / 2, s->height / 2) != 0 ||
        read_plane(s->f, pic->img.plane[2], pic->img.i_stride[2], s->width / 2, s->height / 2) != 0) {
        fprintf(stderr, "source: frame %ld is truncated\n", s->frames_read);
        return -1;
    }
    s->frames_read++;
    return 1;
}
