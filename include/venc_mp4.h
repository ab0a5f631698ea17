/*
 * venc_mp4.h -- minimal ISO-BMFF ("MP4") writer for one avc1 video track: the step right after the
 * encode path in the reference program, which does it with libmp4v2
 * (pinned to an svn revision the build downloads: av_encode.c Makefile:16, not available offline).
 * Host C, no GPU, no third-party library.
 *
 * The three entry points mirror the reference's own functions one to one, so that its main loop
 * could call them unchanged:
 *
 *   venc_mp4_open                 enc_mp4_open                 av_encode.c:614-664
 *       video track only (the audio track, FAAC, is out of scope for good: SURVEY.md section 2);
 *       time scale = time_base num * den (:638), 4-byte NAL length fields (:640), pixel aspect
 *       ratio box (:646).
 *   venc_mp4_write_video_sample   enc_mp4_write_video_sample   av_encode.c:670-751
 *       the first SPS configures the track from its bytes 5..7 (:692-715); SPS and PPS go to the
 *       avcC box without their 4-byte prefix (:720, :725); filler NALs are dropped (:729); the
 *       first NAL of any other type and everything after it become ONE sample, length prefixes
 *       included, sized by pointer subtraction from nals[0].p_payload (:732-746).
 *   venc_mp4_mux_video            enc_mp4_mux_video            av_encode.c:762-834
 *       holds every frame back until the next one arrives, because a sample's duration is the
 *       difference of the two decode times (:775); composition offset = pts - dts (:776); a call
 *       with payload_size = 0 flushes the last frame with duration 1 (:817).
 *   venc_mp4_close                MP4Close                     av_encode.c:1104
 *
 * One documented deviation: the reference hands decode_delta (in time_base units) to the container
 * as is, although its track time scale is num * den ticks per second -- correct only for num = 1 (its
 * comment at :633 says the pts "only has to be multiplied with the numerator", the code never does).
 * Here durations and offsets ARE multiplied by num; for num = 1 the files agree.
 *
 * File layout: ftyp, mdat (64-bit size, samples appended as they come), moov at the end
 * (mvhd, trak: tkhd, mdia: mdhd, hdlr, minf: vmhd, dinf, stbl: stsd[avc1: avcC, pasp], stts, ctts
 * (only if an offset is non-zero), stss, stsc, stsz, co64).  All functions return 0 on success and a
 * negative value on error (message on stderr, like the reference's mp4 functions).
 */
#ifndef VENC_MP4_H
#define VENC_MP4_H
#include <stddef.h>
#include <stdint.h>

#include "x264.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct venc_mp4 venc_mp4;

venc_mp4 *venc_mp4_open(const char *filename, int width, int height, int time_base_num, int time_base_den, int sar_num, int sar_den);
int venc_mp4_write_video_sample(venc_mp4 *m, const x264_nal_t *nals, int nal_count, size_t payload_size, int is_sync_sample,
                                int64_t decode_delta, int64_t composition_offset);
int venc_mp4_mux_video(venc_mp4 *m, const x264_nal_t *nals, int nal_count, int payload_size, const x264_picture_t *pic_out);
int venc_mp4_close(venc_mp4 *m);
/* samples written so far (tests, progress output) */
long venc_mp4_sample_count(const venc_mp4 *m);

#ifdef __cplusplus
}
#endif
#endif
