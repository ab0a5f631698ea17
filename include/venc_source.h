/*
 * venc_source.h -- raw YUV420p / YUV4MPEG2 frame source: the step right before the encode path in
 * the reference program.  The reference pulls decoded, filtered frames
 * from libav* and copies them into the encoder's input picture with a same-size sws_scale used purely
 * as a strided copy (av_encode.c:427-430, :525-560, :545-547); demuxing, decoding and filtering are
 * out of scope, so frames come from a file that already holds 8-bit 4:2:0 pictures:
 *   - YUV4MPEG2 (.y4m): size, frame rate and pixel aspect ratio come from the stream header;
 *     colour spaces C420, C420jpeg, C420mpeg2, C420paldv (all 8-bit 4:2:0 planar; anything else is
 *     refused, as is interlacing other than Ip / I?);
 *   - headerless I420: the caller gives the size.
 * venc_source_read() fills the three planes of an x264_picture_t obtained from x264_picture_alloc
 * (pinned host memory, av_encode.c:415) honouring its strides, so the next x264_encoder_encode() is
 * a single DMA per plane.  Host C, no GPU.
 */
#ifndef VENC_SOURCE_H
#define VENC_SOURCE_H
#include <stdint.h>

#include "x264.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct venc_source venc_source;

/* width / height: picture size of a headerless I420 file; pass 0, 0 for YUV4MPEG2.  NULL on failure
 * (message on stderr). */
venc_source *venc_source_open(const char *filename, int width, int height);
void venc_source_close(venc_source *s);
int venc_source_width(const venc_source *s);
int venc_source_height(const venc_source *s);
/* frame rate and pixel aspect ratio (30:1 and 1:1 when the file does not say) */
void venc_source_rate(const venc_source *s, int *fps_num, int *fps_den);
void venc_source_aspect(const venc_source *s, int *sar_num, int *sar_den);
/* frames in the file, from its size (the reference takes the duration of its progress line from the container,
 * av_encode.c:1032-1069); -1 when the size is unknown (a pipe) */
long venc_source_frame_count(const venc_source *s);
/* 1: a frame was read into pic's planes; 0: end of stream; < 0: truncated or malformed input */
int venc_source_read(venc_source *s, x264_picture_t *pic);

#ifdef __cplusplus
}
#endif
#endif
