/*
 * x264.h -- source-compatible SUBSET of the libx264 public API, limited to the
 * names the reference program touches, implemented by libx264_venc.so on top of
 * the MI355X engine (include/venc_hip.h).
 *
 * This header is written for this repository; it is NOT libx264's header and is
 * not binary compatible with a real libx264.so (struct layouts differ; the real
 * header is GPL and absent from the build image).  Each declaration cites the
 * reference call site it serves (file av_encode.c of arkanis/video-encoder):
 *
 *   x264_param_default_preset   :384      x264_param_apply_profile  :403
 *   x264_encoder_open           :408      x264_picture_alloc        :415
 *   x264_encoder_encode         :970, :1078
 *   x264_encoder_delayed_frames :1076     x264_picture_clean        :442
 *   x264_encoder_close          :443
 *
 * Fields read/written by the reference: x264_param_t {i_width, i_height,
 * b_annexb, i_fps_num, i_fps_den, vui.i_sar_width, vui.i_sar_height,
 * rc.i_rc_method, rc.f_rf_constant} (:389-401); x264_picture_t {i_type, i_pts,
 * i_dts, b_keyframe, img.plane[], img.i_stride[]} (:543-547, :775-783);
 * x264_nal_t {i_type, i_payload, p_payload} (:682, :703-705, :722).
 *
 * Documented deviations (spec/CODEC.md): the encoder is an
 * H.264 Constrained-Baseline subset (CAVLC, one reference, no B pictures).
 * X264_RC_CQP codes every picture at rc.i_qp_constant; X264_RC_CRF -- what the
 * reference sets (av_encode.c:400-401) -- is a frame-level approximation decided
 * on the GPU (spec/CODEC.md section 1a: IDR pictures at round(f) - 3, P pictures
 * at round(f) +- up to 6 steered by the previous picture's SAD sum; base clamped
 * to 12..51), not libx264's adaptive rate control.  Preset names select the
 * optional coding tools as libx264's do in spirit: every preset but "ultrafast"
 * switches on quarter-sample motion refinement (hip.i_subpel = 2), the in-loop
 * deblocking filter (hip.i_deblock = 1) and, for I pictures, Intra16x16 V / H / DC /
 * plane prediction with Intra4x4 tried against it (hip.i_intra_modes = 3);
 * "ultrafast" leaves them off (Intra16x16 with plane prediction only).  Of the tunes
 * only "zerolatency" changes behaviour (no output delay).  Unknown names fail
 * (av_encode.c:384-386).
 */
#ifndef X264_VENC_X264_H
#define X264_VENC_X264_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X264_BUILD 0 /* not a libx264 build */

typedef struct x264_t x264_t;

enum nal_unit_type_e {
    NAL_UNKNOWN = 0,
    NAL_SLICE = 1,
    NAL_SLICE_IDR = 5,
    NAL_SEI = 6,
    NAL_SPS = 7,
    NAL_PPS = 8,
    NAL_AUD = 9,
    NAL_FILLER = 12
};
enum nal_priority_e { NAL_PRIORITY_DISPOSABLE = 0, NAL_PRIORITY_LOW = 1, NAL_PRIORITY_HIGH = 2, NAL_PRIORITY_HIGHEST = 3 };

typedef struct x264_nal_t {
    int i_ref_idc;
    int i_type;        /* nal_unit_type_e */
    int i_payload;     /* bytes at p_payload, INCLUDING the 4-byte length prefix / start code */
    uint8_t *p_payload; /* all NALs of one frame are contiguous, starting at nals[0].p_payload */
} x264_nal_t;

#define X264_CSP_I420 0x0001
#define X264_TYPE_AUTO 0x0000
#define X264_TYPE_IDR 0x0001
#define X264_TYPE_I 0x0002
#define X264_TYPE_P 0x0003
#define X264_RC_CQP 0
#define X264_RC_CRF 1
#define X264_RC_ABR 2

typedef struct x264_param_t {
    int i_width, i_height;
    int i_csp;
    int b_annexb;        /* 0: 4-byte big-endian length prefixes (what the reference asks for, :392); 1: start codes */
    int i_fps_num, i_fps_den;
    int i_keyint_max;    /* closed GOP length, default 30 */
    int i_slice_count;   /* MB-row bands per frame, default 8 (independent of the GPU count) */
    int i_bframe;        /* always 0 */
    int i_frame_reference; /* always 1 */
    int b_zerolatency;   /* set by tune "zerolatency": encode returns the frame it was given */
    struct {
        int i_sar_width, i_sar_height;
    } vui;
    struct {
        int i_rc_method;
        int i_qp_constant;   /* X264_RC_CQP */
        float f_rf_constant; /* X264_RC_CRF: QP = round(f) */
    } rc;
    struct {                /* engine extension, not part of libx264 */
        int i_device;       /* HIP device ordinal */
        /* Closed GOPs encoded in lock-step ("items"): 1 = frame-by-frame streaming; 0 (x264_param_default_preset; tune zerolatency:
         * 1) = chosen by x264_encoder_open: about 245 000 macroblocks per step and never fewer than 15 items (30 at 1080p, 15 at
         * 2160p: what fills the chip best, measured), or with the in-loop filter or the Intra4x4 trial on -- whose kernels launch one
         * block per slice and item -- at least enough that items x slices reaches the chip's 256 CUs (32 with the default 8 slices).
         * WHAT IT COSTS THE CALLER.  Delay: x264_encoder_encode returns 0 (av_encode.c:971-974) for the first 1.5 x i_gop_batch x
         * i_keyint_max pictures (1350 / 1440 at 1080p with a key-frame interval of 30: a unit is launched when its last picture is in, and
         * half a unit of further pictures covers its time on the GPU) and the rest arrives through the x264_encoder_delayed_frames loop
         * (:1076-1083).  A caller that wants its first frame earlier sets i_gop_batch itself (10 items: 450 pictures, 12 % slower).
         * Memory: the encoder pools twice that many pictures in HBM -- per picture its I420 input (1.5 B/px) and an output slot of
         * 400 bytes per macroblock (the conformance bound), 6.4 MB at 1080p, 25.4 MB at 2160p -- plus ~22 KB per macroblock and item
         * of working buffers: 3.9 / 11.5 / 12.3 GB at 1080p with 10 / 30 / 32 items, 15.5 / 23 / 49 GB at 2160p with 10 / 15 / 32.  Whatever is asked for here or chosen,
         * x264_encoder_open lowers it until that fits into half of the device's free memory (a long i_keyint_max -- libx264's own
         * default is 250 -- ends with fewer items, never with a failed open).  Host memory: four steps' worth of pinned
         * output buffers of that slot size (4 x items, at least 8: 390 MB at 1080p with 30 items) whatever the pool, and the packaged NAL
         * payloads of the pooled pictures (their real size). */
        int i_gop_batch;
        int i_gop_base;     /* index of this encoder's first GOP in the whole stream (GOP-sharded multi-GPU runs) */
        /* slice-parallel multi-GPU mode (BASELINE config 5): this encoder codes only
         * the slices [i_slice_first, i_slice_first + i_slice_count) of every picture (0, 0 = all) and calls
         * pf_exchange(p_exchange_user, engine, n_items) after every picture that a P picture follows, so
         * that the bands' owners can swap reconstructed rows (venc_band_export / RCCL / venc_band_import,
         * include/venc_hip.h).  The NAL array of a frame then holds this band's slices only (parameter
         * sets come from the band with slice 0); concatenating the bands' payloads in band order gives the
         * single-encoder frame byte for byte. */
        int i_slice_first, i_slice_count;
        int (*pf_exchange)(void *user, void *engine, int n_items);
        void *p_exchange_user;
        /* The same exchange done by the engine itself over RCCL (include/venc_hip.h, "RCCL exchange"), for C hosts:
         * i_rccl_world > 0 makes x264_encoder_open join a communicator of that many ranks as rank i_rccl_rank with
         * the 128-byte id rank 0 obtained from venc_rccl_unique_id(); the band follows from world and rank (slices
         * dealt in contiguous runs; i_slice_first / i_slice_count are ignored) and every exchange is one call on the
         * engine's stream: i_halo_rows = 16 (neighbours' halos, grouped ncclSend / ncclRecv) or 0 (whole bands,
         * ncclAllGather).  pf_exchange, when set, takes precedence. */
        int i_rccl_world, i_rccl_rank, i_halo_rows;
        unsigned char rccl_id[128];
        /* Intra prediction of I pictures: 3 (x264_param_default_preset's default, like libx264's presets) = Intra4x4 (nine modes per
         * block) tried against the best Intra16x16 predictor; 2 = Intra16x16 vertical / horizontal / DC / plane chosen per macroblock by
         * SAD ("ultrafast"); 1 = without plane; 0 = DC only (the codec definition of round 1, what bench.py measures) */
        int i_intra_modes;
        /* sub-sample refinement of the motion vectors (libx264's analyse.i_subpel_refine, which the reference leaves to its
         * preset, av_encode.c:365): 0 = integer-sample vectors (what bench.py measures), 1 = half-sample, 2 = half- then
         * quarter-sample stage (x264_param_default_preset's default for every preset but "ultrafast", as in libx264).
         * Slice-band encoders run with 0. */
        int i_subpel;
        /* in-loop deblocking filter (libx264's b_deblocking_filter, on in its defaults and in every preset): 1 = H.264 8.7
         * inside every slice (disable_deblocking_filter_idc = 2: slices, and with them GPU bands, stay independent;
         * x264_param_default_preset's default), 0 = off (what bench.py measures) */
        int i_deblock;
        /* Input pictures the caller rotates through.  1 (default) = the reference's shape: ONE pic_in, rewritten right after
         * every x264_encoder_encode (av_encode.c:543-547, :970), so the call returns only when its H2D copy is done.
         * n >= 2 = the caller fills n pictures in turn and promises not to touch a picture until n - 1 further calls have
         * returned: the call then waits for the copy issued n - 1 calls earlier, and the host's work between two calls
         * overlaps the copy in flight. */
        int i_input_depth;
    } hip;
    char psz_preset[16], psz_tune[16], psz_profile[16];
} x264_param_t;

typedef struct x264_image_t {
    int i_csp;
    int i_plane;
    int i_stride[4];
    uint8_t *plane[4];
} x264_image_t;

typedef struct x264_picture_t {
    int i_type;
    int i_qpplus1;
    int b_keyframe;
    int64_t i_pts;
    int64_t i_dts;
    x264_image_t img;
    void *opaque;
} x264_picture_t;

/* 0 on success, non-zero for an unknown preset or tune name (:384-386) */
int x264_param_default_preset(x264_param_t *param, const char *preset, const char *tune);
/* accepts NULL (:105, :403); only "baseline" (or NULL) is honoured, other known names are accepted and ignored */
int x264_param_apply_profile(x264_param_t *param, const char *profile);
/* NULL on failure (:408-412), including "no HIP device": there is no CPU fallback.  Side effect: the calling thread -- it launches the
 * encoder's kernels from here on -- and the threads the encoder creates are restricted to the CPUs of the card's NUMA node
 * (venc_numa_bind, venc_hip.h; VENC_NUMA_BIND=0 leaves the caller's affinity alone) */
x264_t *x264_encoder_open(x264_param_t *param);
/* 0 on success; planes live in pinned host memory so ingest is a single DMA (:415, :545-547) */
int x264_picture_alloc(x264_picture_t *pic, int i_csp, int i_width, int i_height);
void x264_picture_clean(x264_picture_t *pic);
/* > 0: total bytes of all NALs of the returned frame; 0: nothing yet; < 0: error.
 * pic_in == NULL flushes (:1078).  *pp_nal and the payload stay valid until the next call. */
int x264_encoder_encode(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_in, x264_picture_t *pic_out);
int x264_encoder_delayed_frames(x264_t *h);
void x264_encoder_close(x264_t *h);

/* ---- engine extension: frames already resident in HBM (bench, multi-stream servers).
 * d_frames: consecutive tight I420 frames in device memory.  For every encoded
 * frame, in display order, sink(user, nals, nal_count, payload_bytes, pic_out)
 * is called exactly as the reference's muxer would be (av_encode.c:972).
 * Returns the number of frames delivered, or < 0. */
typedef int (*x264_venc_sink)(void *user, const x264_nal_t *nals, int nal_count, int payload_bytes,
                              const x264_picture_t *pic_out);
int x264_venc_encode_resident(x264_t *h, const void *d_frames, uint64_t frame_stride, int n_frames, int64_t first_pts,
                              x264_venc_sink sink, void *user);

/* The same over `repeats` back-to-back passes of the resident clip, as ONE stream of clip_frames * repeats
 * frames (stream frame f reads clip frame f % clip_frames): the steady state of a server that encodes batch
 * after batch.  Consecutive scheduling units overlap -- the GPU works on unit k + 1 while the host packages
 * unit k and hands it to the sink -- which separate calls cannot do (each returns only after its last frame
 * has been delivered).  With clip_frames a multiple of the GOP length the bitstream equals that of `repeats`
 * separate calls. */
int x264_venc_encode_resident_loop(x264_t *h, const void *d_frames, uint64_t frame_stride, int clip_frames, int repeats,
                                   int64_t first_pts, x264_venc_sink sink, void *user);

/* ready-made sink: frame/byte counters + FNV-1a digest of every payload byte.  pass_frames != 0: the time
 * (CLOCK_MONOTONIC, ns) at which the last frame of every pass_frames frames was delivered, for the first
 * X264_VENC_TALLY_PASSES passes -- how a benchmark reads the spread between the passes of ONE pipelined stream */
#define X264_VENC_TALLY_PASSES 256
typedef struct {
    uint64_t frames, bytes, keyframes, fnv1a;
    uint64_t pass_frames, passes;
    uint64_t pass_end_ns[X264_VENC_TALLY_PASSES];
} x264_venc_tally;
int x264_venc_tally_sink(void *user, const x264_nal_t *nals, int nal_count, int payload_bytes, const x264_picture_t *pic_out);

/* The reference's own loop over host frames, in C, for measurements (what bench.py's "pcie_inclusive" leg times): for every
 * frame of `repeats` passes over clip_frames tight I420 frames in (pinned) host memory -- pic_in's planes point at the
 * frame, i_type / i_pts as av_encode.c:543-544, x264_encoder_encode (av_encode.c:970), the sink on every output -- then
 * the flush loop of av_encode.c:1076-1083.  Nothing but the eight public entry points is used.  Returns the frames
 * delivered, or < 0. */
int x264_venc_encode_host_frames(x264_t *h, const uint8_t *frames, uint64_t frame_stride, int width, int height, int clip_frames,
                                 int repeats, x264_venc_sink sink, void *user);
/* the venc_ctx* behind an encoder (for venc_profile / venc_debug_read) */
void *x264_venc_engine(x264_t *h);
/* closed GOPs the encoder advances in lock-step: hip.i_gop_batch, or what 0 resolved to, after the cap by device memory */
int x264_venc_gop_batch(x264_t *h);

#ifdef __cplusplus
}
#endif
#endif
