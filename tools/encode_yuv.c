/*
 * encode_yuv.c -- C driver that replays the reference program's video path
 * against libx264_venc.so, call for call:
 *
 *   open      enc_x264_open            av_encode.c:378-438 (called at :895)
 *   ingest    pic_in.i_type/i_pts + plane copy   :543-547
 *   encode    x264_encoder_encode      :970-975
 *   mux       enc_mp4_mux_video / enc_mp4_write_video_sample   :762-834, :670-751
 *   flush     delayed-frames loop      :1076-1083, final mux flush :1087-1088
 *   close     enc_x264_close           :440-444 (called at :1120)
 *
 * The reference's demux/decode/filter stages (libav*) and its MP4 writer
 * (libmp4v2) are out of scope; frames come from a raw
 * I420 file or the built-in synthetic source, and the "muxer" below keeps the
 * reference's one-frame hold-back logic and checks every invariant the real
 * one relies on before writing the samples as an
 * Annex-B elementary stream (or nothing with -o -).
 *
 * With -m the samples also go through the ISO-BMFF writer of this repository (include/venc_mp4.h,
 * the reference's enc_mp4_open / enc_mp4_mux_video / MP4Close) into a playable .mp4; -i takes a
 * YUV4MPEG2 file (size, rate and aspect from its header) or, with -s, headerless I420
 * (include/venc_source.h, the reference's pull-and-copy at av_encode.c:525-560).
 *
 * usage: encode_yuv [-s WxH] [-n frames] [-q qp] [-S slices] [-k keyint] [-g gop_batch] [-t tune]
 *                   [-i in.y4m|in.yuv] [-o out.h264] [-m out.mp4] [-f fps]
 *
 * The reference's own command line (av_encode.c:89-196) is accepted too, by
 * its long option names -- its short ones (-s -d -l -f) collide with the switches above:
 *        encode_yuv [--silent] [--debug] [--frame-limit N] [--preset P] [--tune T] [--quality Q]
 *                   [--profile P] [--video-stream N] [--audio-stream N] [--filters F] input output
 * input: .y4m (or headerless I420 with -s); output: .mp4 -> the MP4 writer, anything else -> Annex-B.
 * --quality is the reference's CRF number (default 20, av_encode.c:104), mapped to the fixed QP by the
 * shim; --frame-limit is enforced (the reference parses it and then ignores it); stream selection and
 * filters belong to the libav* stages that are out of scope: accepted and ignored with a note.  Unless
 * --silent (default when stdin is not a terminal, as in the reference) a progress line in the
 * reference's format is printed every half second (av_encode.c:1032-1069, without the audio half).
 *
 * Built a second time as tools/av_encode (-DAV_ENCODE_CLI): the reference's command line and nothing else, parsed exactly
 * as av_encode.c:110-187 parses it -- getopt_long with the option string "sdv:a:l:f:", so the SHORT options work too
 * (-s silent, -d debug, -v / -a stream indices, -l frame limit, -f filters), then "input output"; the option dump of
 * av_encode.c:188-193, the exit codes of main (1: bad command line, ...), and the progress line.  The driver's own
 * switches (-n -q -S -k -g -t -i -o -m) exist in encode_yuv only; in av_encode the stream constants they set keep the
 * defaults of x264_param_default_preset (8 slices, GOP 30) and can be overridden through VENC_SLICES / VENC_KEYINT /
 * VENC_GOP_BATCH in the environment.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "venc_hip.h"
#include "venc_mp4.h"
#include "venc_source.h"
#include "x264.h"

typedef struct {
    x264_t *encoder;
    x264_picture_t pic_in, pic_out;
    x264_nal_t *nals;
    int nal_count;
    int payload_size;
} x264_context_t; /* av_encode.c:369-376 without the swscale handle */

static bool enc_x264_open(int width, int height, int fps_num, int fps_den, int sar_num, int sar_den, const char *preset, const char *tune,
                          int quality, const char *profile, int slices, int keyint, int gop_batch, x264_context_t *x) {
    x264_param_t params;
    if (x264_param_default_preset(&params, preset, tune) != 0) {
        fprintf(stderr, "x264: failed to set preset %s and tune %s\n", preset, tune);
        return false;
    }
    params.i_width = width;
    params.i_height = height;
    params.b_annexb = false;
    params.i_fps_num = fps_num;
    params.i_fps_den = fps_den;
    params.vui.i_sar_width = sar_num;
    params.vui.i_sar_height = sar_den;
    params.rc.i_rc_method = X264_RC_CRF;
    params.rc.f_rf_constant = (float)quality;
    params.i_slice_count = slices;
    params.i_keyint_max = keyint;
    if (gop_batch > 0) params.hip.i_gop_batch = gop_batch; /* (0: what the preset / tune chose: 10 GOPs in lock-step, 1 under zerolatency) */
    /* the preset's coding tools can be pinned for measurements: VENC_SUBPEL=0/1/2 (sub-sample refinement; preset default 2,
     * ultrafast 0), VENC_INTRA_MODES=0/1/2/3 (Intra16x16 DC only / V, H, DC / plane as well / Intra4x4 tried against that; preset default 3, ultrafast 2),
     * VENC_DEBLOCK=0/1 (in-loop filter; preset default 1, ultrafast 0) */
    if (getenv("VENC_SUBPEL")) params.hip.i_subpel = atoi(getenv("VENC_SUBPEL"));
    if (getenv("VENC_INTRA_MODES")) params.hip.i_intra_modes = atoi(getenv("VENC_INTRA_MODES"));
    if (getenv("VENC_DEBLOCK")) params.hip.i_deblock = atoi(getenv("VENC_DEBLOCK"));
    if (x264_param_apply_profile(&params, profile) != 0) {
        fprintf(stderr, "x264: failed to apply profile %s\n", profile);
        return false;
    }
    x->encoder = x264_encoder_open(&params);
    if (x->encoder == NULL) {
        fprintf(stderr, "x264: failed to initialize encoder\n");
        return false;
    }
    if (x264_picture_alloc(&x->pic_in, X264_CSP_I420, width, height) != 0) {
        fprintf(stderr, "x264: could not allocate input picture\n");
        return false;
    }
    x->pic_out.i_pts = 0;
    return true;
}

static void enc_x264_close(x264_context_t *x) {
    x264_picture_clean(&x->pic_in);
    x264_encoder_close(x->encoder);
}

/* ---- muxer stand-in with the reference's structure and its assumptions made explicit ---- */
typedef struct {
    uint8_t *payload_data;
    size_t payload_size;
    x264_nal_t *nal_data;
    size_t nal_count;
    x264_picture_t pic;
} x264_frame_t;

static FILE *g_out = NULL;
static venc_mp4 *g_mp4 = NULL;
static long g_samples = 0, g_sync_samples = 0, g_bytes = 0, g_violations = 0;
static bool g_track_configured = false;

#define REQUIRE(cond, msg)                                   \
    do {                                                     \
        if (!(cond)) {                                       \
            fprintf(stderr, "boundary violation: %s\n", msg); \
            g_violations++;                                  \
        }                                                    \
    } while (0)

static void write_sample(x264_nal_t *nals, int nal_count, size_t payload_size, bool is_sync, int64_t decode_delta,
                         int64_t composition_offset) {
    static const uint8_t startcode[4] = {0, 0, 0, 1};
    REQUIRE(decode_delta > 0, "dts must increase strictly (sample duration = delta dts)");
    REQUIRE(composition_offset >= 0, "pts - dts must be >= 0");
    for (int i = 0; i < nal_count; i++) {
        x264_nal_t *n = &nals[i];
        const uint32_t len = ((uint32_t)n->p_payload[0] << 24) | ((uint32_t)n->p_payload[1] << 16) | ((uint32_t)n->p_payload[2] << 8) | n->p_payload[3];
        REQUIRE((int)len + 4 == n->i_payload, "i_payload must include the 4-byte length prefix");
        REQUIRE((n->p_payload[4] & 31) == n->i_type, "NAL header type must match i_type");
        switch (n->i_type) {
            case NAL_SPS:
                if (!g_track_configured) {
                    REQUIRE(n->p_payload[5] == 66, "SPS byte 5 = profile_idc");
                    REQUIRE(n->p_payload[7] >= 10 && n->p_payload[7] <= 52, "SPS byte 7 = level_idc");
                    g_track_configured = true;
                }
                /* fallthrough: parameter sets go to the elementary stream too */
            case NAL_PPS:
                if (g_out) {
                    fwrite(startcode, 1, 4, g_out);
                    fwrite(n->p_payload + 4, 1, (size_t)n->i_payload - 4, g_out);
                }
                break;
            case NAL_FILLER:
                break;
            default: {
                int remaining = nal_count - i;
                uint8_t *start = nals[i].p_payload;
                long size = (long)payload_size - (long)(start - nals[0].p_payload);
                long sum = 0;
                for (int k = i; k < nal_count; k++) {
                    REQUIRE(nals[k].p_payload == start + sum, "NAL payloads must be contiguous");
                    REQUIRE(nals[k].i_type != NAL_SPS && nals[k].i_type != NAL_PPS, "parameter sets must precede slices");
                    if (g_out) {
                        fwrite(startcode, 1, 4, g_out);
                        fwrite(nals[k].p_payload + 4, 1, (size_t)nals[k].i_payload - 4, g_out);
                    }
                    sum += nals[k].i_payload;
                }
                REQUIRE(sum == size, "sample size computed by pointer subtraction must cover exactly the remaining NALs");
                g_samples++;
                g_sync_samples += is_sync;
                g_bytes += size;
                i += remaining;
            } break;
        }
    }
}

static bool mux_video(x264_context_t *x) {
    static x264_frame_t prev = {NULL, 0, NULL, 0, {0}};
    /* the real container, when asked for: same call, same arguments as the reference's enc_mp4_mux_video */
    if (g_mp4 && venc_mp4_mux_video(g_mp4, x->nals, x->nal_count, x->payload_size, &x->pic_out) != 0) return false;
    if (x->payload_size > 0) {
        if (prev.payload_size > 0)
            write_sample(prev.nal_data, (int)prev.nal_count, prev.payload_size, prev.pic.b_keyframe,
                         x->pic_out.i_dts - prev.pic.i_dts, prev.pic.i_pts - prev.pic.i_dts);
        free(prev.payload_data);
        free(prev.nal_data);
        prev.payload_data = (uint8_t *)malloc((size_t)x->payload_size);
        prev.nal_data = (x264_nal_t *)malloc((size_t)x->nal_count * sizeof(x264_nal_t));
        if (!prev.payload_data || !prev.nal_data) return false;
        memcpy(prev.payload_data, x->nals[0].p_payload, (size_t)x->payload_size);
        prev.payload_size = (size_t)x->payload_size;
        prev.pic = x->pic_out;
        prev.nal_count = (size_t)x->nal_count;
        for (int i = 0; i < x->nal_count; i++) {
            prev.nal_data[i] = x->nals[i];
            prev.nal_data[i].p_payload = prev.payload_data + (x->nals[i].p_payload - x->nals[0].p_payload);
        }
    } else if (prev.payload_size > 0) {
        write_sample(prev.nal_data, (int)prev.nal_count, prev.payload_size, prev.pic.b_keyframe, 1, prev.pic.i_pts - prev.pic.i_dts);
        free(prev.payload_data);
        free(prev.nal_data);
        prev.payload_data = NULL;
        prev.nal_data = NULL;
        prev.payload_size = 0;
    }
    return true;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    int w = 0, h = 0, frames = 30, qp = 26, slices = 8, keyint = 30, gop_batch = 0, fps = 30, fps_den = 1, sar_num = 1, sar_den = 1;
    int fps_given = 0;
    const char *in = NULL, *out = NULL, *mp4 = NULL, *tune = "film", *preset = "medium", *profile = NULL;
    /* the reference's default (av_encode.c:92); VENC_ASSUME_TTY=1 stands in for a terminal where none can be had (test boxes without pty devices) */
    bool silent = !isatty(STDIN_FILENO) && !getenv("VENC_ASSUME_TTY"), debug_show = false, ref_cli = false;
    long long frame_limit = -1;
    float quality = -1.0f;
    int video_stream_index = -1, audio_stream_index = -1;
    const char *video_filter = NULL;
#ifdef AV_ENCODE_CLI
    /* the reference's parse_cli_options (av_encode.c:89-196), option for option */
    {
        static const struct option long_opts[] = {{"silent", no_argument, NULL, 's'},          {"debug", no_argument, NULL, 'd'},
                                                  {"video-stream", required_argument, NULL, 'v'}, {"audio-stream", required_argument, NULL, 'a'},
                                                  {"frame-limit", required_argument, NULL, 'l'}, {"filters", required_argument, NULL, 'f'},
                                                  {"preset", required_argument, NULL, 1},        {"tune", required_argument, NULL, 2},
                                                  {"quality", required_argument, NULL, 3},       {"profile", required_argument, NULL, 4},
                                                  {NULL, 0, NULL, 0}};
        int idx = 0, opt;
        quality = 20.0f; /* av_encode.c:104 */
        while ((opt = getopt_long(argc, argv, "sdv:a:l:f:", long_opts, &idx)) != -1) {
            switch (opt) {
                case 's': silent = true; break;
                case 'd': debug_show = true; break;
                case 'v': video_stream_index = (int)strtol(optarg, NULL, 10); break;
                case 'a': audio_stream_index = (int)strtol(optarg, NULL, 10); break;
                case 'l': frame_limit = strtoll(optarg, NULL, 10); break;
                case 'f': video_filter = optarg; break;
                case 1: preset = optarg; break;
                case 2: tune = optarg; break;
                case 3: quality = strtof(optarg, NULL); break;
                case 4: profile = optarg; break;
                default: return 1; /* getopt_long has printed the message (av_encode.c:160-163, main's return 1) */
            }
        }
        if (optind < argc)
            in = argv[optind++];
        else {
            fprintf(stderr, "no input file specified!\n");
            return 1;
        }
        if (optind < argc) {
            const char *o = argv[optind++];
            const size_t n = strlen(o);
            if (n > 4 && !strcmp(o + n - 4, ".mp4")) mp4 = o; else out = o;
        } else {
            fprintf(stderr, "no output file specified!\n");
            return 1;
        }
        printf("silent: %d \ndebug: %d \ninput_file: %s \noutput_file: %s \nvideo_stream_index: %d \naudio_stream_index: %d \nframe_limit: %ld \nvideo_filter: %s \npreset: %s \ntune: %s \nquality: %f \nprofile: %s\n",
               silent, debug_show, in, mp4 ? mp4 : out, video_stream_index, audio_stream_index, (long)frame_limit, video_filter ? video_filter : "(null)", preset,
               tune, (double)quality, profile ? profile : "(null)");
        if (video_stream_index >= 0 || audio_stream_index >= 0 || video_filter)
            fprintf(stderr, "note: stream selection and filters belong to the demux/decode/filter stages, which are out of scope: ignored\n");
        if (getenv("VENC_SLICES")) slices = atoi(getenv("VENC_SLICES"));
        if (getenv("VENC_KEYINT")) keyint = atoi(getenv("VENC_KEYINT"));
        if (getenv("VENC_GOP_BATCH")) gop_batch = atoi(getenv("VENC_GOP_BATCH"));
        frames = 0x7fffffff;
        argc = 1; /* nothing left for the driver's own switches */
    }
#else
    (void)video_stream_index, (void)audio_stream_index, (void)video_filter;
    /* first pass: the reference's long options (av_encode.c:110-124); they are removed from argv */
    {
        static const struct option long_opts[] = {{"silent", no_argument, NULL, 's'},          {"debug", no_argument, NULL, 'd'},
                                                  {"video-stream", required_argument, NULL, 'v'}, {"audio-stream", required_argument, NULL, 'a'},
                                                  {"frame-limit", required_argument, NULL, 'l'}, {"filters", required_argument, NULL, 'F'},
                                                  {"preset", required_argument, NULL, 1},        {"tune", required_argument, NULL, 2},
                                                  {"quality", required_argument, NULL, 3},       {"profile", required_argument, NULL, 4},
                                                  {NULL, 0, NULL, 0}};
        char *rest[64];
        int nrest = 0;
        rest[nrest++] = argv[0];
        for (int i = 1; i < argc && nrest < 63; i++) {
            const char *a = argv[i];
            if (strncmp(a, "--", 2) != 0) {
                rest[nrest++] = argv[i];
                continue;
            }
            const struct option *o = NULL;
            for (const struct option *q = long_opts; q->name; q++)
                if (!strcmp(a + 2, q->name)) o = q;
            if (!o) {
                fprintf(stderr, "%s: unrecognized option '%s'\n", argv[0], a);
                return 1;
            }
            const char *val = NULL;
            if (o->has_arg == required_argument) {
                if (i + 1 >= argc) {
                    fprintf(stderr, "%s: option '%s' requires an argument\n", argv[0], a);
                    return 1;
                }
                val = argv[++i];
            }
            ref_cli = true;
            switch (o->val) {
                case 's': silent = true; break;
                case 'd': debug_show = true; break;
                case 'l': frame_limit = strtoll(val, NULL, 10); break;
                case 'v':
                case 'a':
                case 'F': fprintf(stderr, "note: --%s belongs to the demux/decode/filter stages, which are out of scope: ignored\n", o->name); break;
                case 1: preset = val; break;
                case 2: tune = val; break;
                case 3: quality = strtof(val, NULL); break;
                case 4: profile = val; break;
            }
        }
        for (int i = 0; i < nrest; i++) argv[i] = rest[i];
        argc = nrest;
    }
    /* the reference takes "input output" as positional arguments (av_encode.c:172-187) */
    if (argc >= 3 && argv[argc - 2][0] != '-' && argv[argc - 1][0] != '-' && (argc == 3 || argv[argc - 3][0] != '-' || ref_cli)) {
        bool pair = true;
        if (argc > 3 && argv[argc - 3][0] == '-' && strlen(argv[argc - 3]) == 2 && strchr("snqSkgftiom", argv[argc - 3][1])) pair = false;
        if (pair) {
            in = argv[argc - 2];
            const char *o = argv[argc - 1];
            const size_t n = strlen(o);
            if (n > 4 && !strcmp(o + n - 4, ".mp4")) mp4 = o; else out = o;
            argc -= 2;
            ref_cli = true;
        }
    }
    if (ref_cli && quality < 0) quality = 20.0f; /* av_encode.c:104 */
    if (ref_cli) frames = 0x7fffffff;             /* the whole input, or --frame-limit */
#endif
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i], *v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "-s") && v && sscanf(v, "%dx%d", &w, &h) == 2) i++;
        else if (!strcmp(a, "-n") && v) frames = atoi(v), i++;
        else if (!strcmp(a, "-q") && v) qp = atoi(v), i++;
        else if (!strcmp(a, "-S") && v) slices = atoi(v), i++;
        else if (!strcmp(a, "-k") && v) keyint = atoi(v), i++;
        else if (!strcmp(a, "-g") && v) gop_batch = atoi(v), i++;
        else if (!strcmp(a, "-f") && v) fps = atoi(v), fps_given = 1, i++;
        else if (!strcmp(a, "-t") && v) tune = v, i++;
        else if (!strcmp(a, "-i") && v) in = v, i++;
        else if (!strcmp(a, "-o") && v) out = v, i++;
        else if (!strcmp(a, "-m") && v) mp4 = v, i++;
        else {
            fprintf(stderr, "usage: %s [-s WxH] [-n frames] [-q qp] [-S slices] [-k keyint] [-g gop_batch] [-t tune] [-f fps] [-i in.y4m|in.yuv] [-o out.h264] [-m out.mp4]\n", argv[0]);
            return 1;
        }
    }
    venc_source *src = NULL;
    if (in) {
        if (!(src = venc_source_open(in, w, h))) return 2; /* -s given: headerless I420; else YUV4MPEG2 */
        w = venc_source_width(src), h = venc_source_height(src);
        if (!fps_given) venc_source_rate(src, &fps, &fps_den); /* the stream header's rate (30:1 for headerless I420) unless -f */
        venc_source_aspect(src, &sar_num, &sar_den);
    }
    if (!w || !h) w = 352, h = 288;
    if (out && strcmp(out, "-") && !(g_out = fopen(out, "wb"))) {
        perror(out);
        return 2;
    }
    x264_context_t x264;
    memset(&x264, 0, sizeof(x264));
#ifdef AV_ENCODE_CLI
    if (quality >= 0) qp = (int)quality; /* the reference hands its float option to an int parameter (av_encode.c:79, :380, :895): truncated */
#else
    if (quality >= 0) qp = (int)(quality + 0.5f);
#endif
    if (frame_limit >= 0 && frame_limit < frames) frames = (int)frame_limit;
    if (ref_cli)
        printf("silent: %d \ndebug: %d \ninput_file: %s \noutput_file: %s \nframe_limit: %lld \npreset: %s \ntune: %s \nquality: %f \nprofile: %s\n", silent,
               debug_show, in ? in : "(synthetic)", mp4 ? mp4 : out ? out : "(none)", frame_limit, preset, tune, (double)qp, profile ? profile : "(null)");
    if (!enc_x264_open(w, h, fps, fps_den, sar_num, sar_den, preset, tune, qp, profile, slices, keyint, gop_batch, &x264)) return 6;
    /* track time base = 1 / frame rate, as the reference's codec context has it (av_encode.c:638) */
    if (mp4 && !(g_mp4 = venc_mp4_open(mp4, w, h, fps_den, fps, sar_num, sar_den))) return 2;

    const double t0 = now_s();
    double last_progress = t0;
    int64_t encoded_video_pts = 0;
    /* progress line of av_encode.c:1032-1069, video half: position, percentage of the input's duration, estimated time
     * left = (duration left / duration encoded) x time spent; every half second (VENC_PROGRESS_INTERVAL overrides: tests) */
    const double progress_every = getenv("VENC_PROGRESS_INTERVAL") ? atof(getenv("VENC_PROGRESS_INTERVAL")) : 0.5;
    long total_frames = src ? venc_source_frame_count(src) : frames;
    if (total_frames > frames) total_frames = frames;
    const double duration_sec = total_frames > 0 ? (double)total_frames * fps_den / fps : 0.0;
    for (int f = 0; f < frames; f++) {
        if (!silent && now_s() - last_progress > progress_every) {
            const double secs = (double)encoded_video_pts * fps_den / fps, spent = now_s() - t0;
            const long long s = (long long)secs;
            const double left = secs > 0 && duration_sec > secs ? (duration_sec - secs) / secs * spent : 0.0;
            const long long l = (long long)left;
            printf("\rvideo: %lld:%02lld:%02lld (%.1lf%%) - time left: %lld:%02lld:%02lld", s / 3600, (s / 60) % 60, s % 60,
                   duration_sec > 0 ? secs / duration_sec * 100 : 0.0, l / 3600, (l / 60) % 60, l % 60);
            if (debug_show) printf("\n"); else fflush(stdout);
            last_progress = now_s();
        }
        x264.pic_in.i_type = X264_TYPE_AUTO;
        x264.pic_in.i_pts = f;
        if (src) {
            if (venc_source_read(src, &x264.pic_in) != 1) break; /* end of stream (or a truncated frame: reported on stderr) */
        } else {
            venc_synth_frame(w, h, f, 0x1234ABCDu, x264.pic_in.img.plane[0], x264.pic_in.img.plane[1], x264.pic_in.img.plane[2]);
        }
        x264.payload_size = x264_encoder_encode(x264.encoder, &x264.nals, &x264.nal_count, &x264.pic_in, &x264.pic_out);
        if (x264.payload_size > 0)
            mux_video(&x264);
        else if (x264.payload_size < 0)
            fprintf(stderr, "x264: encoder error\n");
        encoded_video_pts = x264.pic_out.i_pts;
    }
    while (x264_encoder_delayed_frames(x264.encoder) > 0) {
        x264.payload_size = x264_encoder_encode(x264.encoder, &x264.nals, &x264.nal_count, NULL, &x264.pic_out);
        if (x264.payload_size > 0)
            mux_video(&x264);
        else if (x264.payload_size < 0)
            fprintf(stderr, "x264: encoder error");
    }
    x264.payload_size = 0;
    mux_video(&x264);
    const double dt = now_s() - t0;
    enc_x264_close(&x264);
    if (g_out) fclose(g_out);
    if (g_mp4 && venc_mp4_close(g_mp4) != 0) g_violations++;
    if (!silent) printf("\n");
    if (src) venc_source_close(src);
    printf("{\"frames\": %ld, \"sync_samples\": %ld, \"bytes\": %ld, \"last_pts\": %lld, \"seconds\": %.4f, \"mpixels_per_s\": %.2f, \"violations\": %ld}\n",
           g_samples, g_sync_samples, g_bytes, (long long)encoded_video_pts, dt, g_samples * (double)w * h / dt / 1e6, g_violations);
    return g_violations ? 3 : 0;
}
