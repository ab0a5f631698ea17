/*
 * x264_shim.c -- C99 host side of the drop-in boundary: the subset of the x264
 * API the reference calls (include/x264.h), implemented on the thin C-ABI of the
 * HIP engine (include/venc_hip.h).  "Host code stays in C."
 *
 * What the reference's caller relies on (all honoured):
 *  1. all NAL payloads of a frame are one contiguous buffer starting at
 *     nals[0].p_payload (av_encode.c:802, :810-811, :738);
 *  2. every NAL starts with a 4-byte length prefix that i_payload includes
 *     (:722, :727) -- b_annexb = 0 (:392);
 *  3. SPS bytes 5..7 = profile_idc, constraint flags, level_idc (:692-705);
 *  4. parameter sets precede slice NALs (:732-746);
 *  5. i_dts = i_pts, strictly increasing (no B-frames) (:775-776);
 *  6. b_keyframe set on IDR frames (:783);
 *  encode may return 0 while the pipeline fills and is drained through
 *  x264_encoder_delayed_frames (:971-974, :1076-1083).
 *
 * Pipeline: frames are queued into an input pool in HBM.  With gop_batch = 1
 * every call launches one lock-step frame and returns the previous one
 * (zerolatency: the same one).  With gop_batch = G > 1 the shim waits until G
 * closed GOPs are queued and advances them together, one kernel sequence per
 * frame index, so each launch covers G x (macroblocks per frame) work items.
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/x264.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/venc_hip.h"
#include "h264_syntax.h"

#define SHIM_MAX_NALS (2 + 256)
#define SHIM_MAX_HELPERS 6
#define SHIM_HEAVY_STEP_BYTES ((size_t)128 << 10) /* a step's frames go to the helpers from this many payload bytes (VENC_PACK_HEAVY_KB): a unit's I pictures above all (5.6 MB at 1080p x 10 GOPs), and P steps of ten or more GOPs (270 KB there) */

typedef struct {
    int64_t pts;
    int is_idr;
    int out_slot;
    int ticket;
    volatile int assembled; /* NALs of this frame are packaged in payload[] (gop_batch > 1: set by the packaging thread under mtx) */
    uint8_t *payload;   /* grown on demand, reused by the next frame that maps to this ring entry */
    size_t payload_cap;
    int payload_size;
    int n_nals;
    x264_nal_t *nals;   /* nal_cap entries */
} shim_frame;

struct x264_t {
    x264_param_t param;
    venc_ctx *ctx;
    int qp, keyint, gop_batch, unit; /* unit = frames per scheduling unit (1 or G*keyint) */
    int input_depth;                 /* hip.i_input_depth: pictures the caller rotates through (1: the reference's one pic_in) */
    int delay;                       /* frames between submission and delivery */
    int pool;                        /* in/out slots */
    int64_t n_in, n_out, n_launched;
    int64_t n_asm; /* frames [0, n_asm) were packaged unit by unit, in completion order */
    int64_t gop_counter;
    shim_frame *ring; /* pool entries, indexed by frame number % pool */
    h264_seq seq;
    int nal_cap;
    int external_input; /* venc_set_input is active (resident mode) */
    double trace_t0;    /* VENC_TRACE: start of the resident call (0 = no trace) */
    double enc_ms[5];   /* VENC_TRACE: x264_encoder_encode's host time by part: upload, wait for the copy, launches, delivery; [4] = calls */
    int enc_trace;
    int64_t ext_clip;   /* resident mode: frames in the resident clip; stream frame f reads clip frame f % ext_clip */
    int band_first, band_count; /* slices coded by this encoder (slice-parallel mode), 0/0 = all */
    int rccl;                   /* the engine exchanges bands itself over RCCL (hip.i_rccl_world > 0) */
    /* gop_batch > 1: NAL packaging (the wait for the GPU, emulation prevention, length prefixes: ~1.9 ms per 300-frame
     * 1080p unit) runs on a thread of its own, in the order the GPU completes frames, beside the caller's thread, which
     * launches units and hands frames to the sink in display order.  The reference's caller stays single-threaded: it
     * sees frames only through x264_encoder_encode / the sink, on its own thread. */
    pthread_t worker;
    int worker_on;
    pthread_mutex_t mtx;
    pthread_cond_t cv_work, cv_done;
    volatile int64_t w_launched; /* copy of n_launched for the worker (under mtx) */
    volatile int w_stop, w_err;
    /* ... and the frames of a HEAVY step (the unit's I pictures: 5.6 MB per 1080p unit, 1.1 ms on one thread, during which the
     * packaging thread fell a dozen P steps behind the GPU and the unit's last frames reached the caller late) are shared with
     * helper threads that sleep otherwise: the packaging thread lists the step's frames as a job, takes frames from it itself
     * and waits for the last one.  Everything about a job is read and written under jmtx.  Same box, six runs each, 1080p x 10
     * GOPs: packaging thread alone 180.2 Gpx/s (178.6 - 182.9), + 2 helpers on the I pictures 190.4 (190.2 - 191.2), + P steps
     * shared too 191.6; 2160p the same with and without (244): gpurun_out/r04_s3f. */
    pthread_t helper[SHIM_MAX_HELPERS];
    int n_helpers;
    pthread_mutex_t jmtx;
    pthread_cond_t jcv_work, jcv_done;
    int64_t j_frames[64]; /* (at most 64 items advance in lock-step) */
    int j_n, j_next, j_done, j_err, j_gen, j_stop;
    size_t heavy_bytes;
};

/* slice-parallel mode: let the bands' owners swap reconstructed rows before the next P picture */
static int exchange_after_step(x264_t *h, int n_items) {
    if (!h->band_count) return 0;
    if (h->param.hip.pf_exchange) return h->param.hip.pf_exchange(h->param.hip.p_exchange_user, h->ctx, n_items) != 0 ? -1 : 0;
    if (h->rccl) return venc_band_exchange_rccl(h->ctx, n_items, h->param.hip.i_halo_rows) != 0 ? -1 : 0;
    return 0; /* a single band covering every slice needs no peer */
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static void *packaging_thread(void *arg);
static void *packaging_helper(void *arg);

static const char *const k_presets[] = {"ultrafast", "superfast", "veryfast", "faster", "fast",
                                        "medium",    "slow",      "slower",   "veryslow", "placebo", NULL};
static const char *const k_tunes[] = {"film", "animation",  "grain",       "stillimage",
                                      "psnr", "ssim",       "fastdecode",  "zerolatency", NULL};
static const char *const k_profiles[] = {"baseline", "main", "high", "high10", "high422", "high444", NULL};

static int name_in(const char *s, const char *const *list) {
    for (; *list; list++)
        if (!strcmp(s, *list)) return 1;
    return 0;
}

int x264_param_default_preset(x264_param_t *param, const char *preset, const char *tune) {
    if (!param) return -1;
    memset(param, 0, sizeof(*param));
    param->i_csp = X264_CSP_I420;
    param->b_annexb = 1; /* libx264's default; the reference overrides it to 0 (av_encode.c:392) */
    param->i_fps_num = 25;
    param->i_fps_den = 1;
    param->i_keyint_max = 30;
    param->i_slice_count = 8;
    param->i_frame_reference = 1;
    param->rc.i_rc_method = X264_RC_CQP;
    param->rc.i_qp_constant = 26;
    param->rc.f_rf_constant = 23.0f;
    param->hip.i_device = 0;
    param->hip.i_gop_batch = 0; /* closed GOPs coded in lock-step: 0 = chosen when the encoder is opened (shim_auto_gop_batch; tune zerolatency: 1,
                                   below).  A caller shaped like the reference -- a file transcoder that drains the encoder through
                                   x264_encoder_delayed_frames (av_encode.c:1076) -- wants the chip filled, not the first frame early; libx264's
                                   own presets delay output by their lookahead too */
    param->hip.i_intra_modes = preset && strcmp(preset, "ultrafast") == 0 ? 2 : 3; /* (libx264: ultrafast analyses no i4x4 partitions) */
    param->hip.i_subpel = preset && strcmp(preset, "ultrafast") == 0 ? 0 : 2;
    param->hip.i_deblock = preset && strcmp(preset, "ultrafast") == 0 ? 0 : 1; /* (libx264: ultrafast sets no-deblock) */
    if (preset) {
        if (!name_in(preset, k_presets)) return -1;
        snprintf(param->psz_preset, sizeof(param->psz_preset), "%s", preset);
    }
    if (tune) {
        /* libx264 accepts several tunes separated by , . or space */
        char buf[64];
        snprintf(buf, sizeof(buf), "%s", tune);
        for (char *tok = strtok(buf, ",./-+ "); tok; tok = strtok(NULL, ",./-+ ")) {
            if (!name_in(tok, k_tunes)) return -1;
            if (!strcmp(tok, "zerolatency")) param->b_zerolatency = 1, param->hip.i_gop_batch = 1;
        }
        snprintf(param->psz_tune, sizeof(param->psz_tune), "%.15s", tune);
    }
    return 0;
}

int x264_param_apply_profile(x264_param_t *param, const char *profile) {
    if (!param) return -1;
    if (!profile) return 0; /* the reference's default (av_encode.c:105) */
    if (!name_in(profile, k_profiles)) return -1;
    snprintf(param->psz_profile, sizeof(param->psz_profile), "%s", profile);
    return 0; /* the engine always emits Constrained Baseline */
}

int x264_picture_alloc(x264_picture_t *pic, int i_csp, int i_width, int i_height) {
    if (!pic || i_csp != X264_CSP_I420 || i_width <= 0 || i_height <= 0 || (i_width & 1) || (i_height & 1)) return -1;
    memset(pic, 0, sizeof(*pic));
    const size_t ysz = (size_t)i_width * i_height;
    uint8_t *buf = (uint8_t *)venc_host_alloc(ysz * 3 / 2);
    if (!buf) return -1;
    pic->i_type = X264_TYPE_AUTO;
    pic->img.i_csp = i_csp;
    pic->img.i_plane = 3;
    pic->img.plane[0] = buf;
    pic->img.plane[1] = buf + ysz;
    pic->img.plane[2] = buf + ysz + ysz / 4;
    pic->img.i_stride[0] = i_width;
    pic->img.i_stride[1] = i_width / 2;
    pic->img.i_stride[2] = i_width / 2;
    return 0;
}

void x264_picture_clean(x264_picture_t *pic) {
    if (!pic) return;
    venc_host_free(pic->img.plane[0]);
    memset(pic, 0, sizeof(*pic));
}

/* Closed GOPs advanced in lock-step when the caller left the choice to the encoder (hip.i_gop_batch = 0).  What fills the chip:
 * the search and the transforms launch items x macroblocks-per-picture lanes -- ten 1080p pictures are 5100 search blocks on 1024
 * block slots --, but the WAVEFRONT tools (in-loop filter, Intra4x4 trial) launch one block per slice and item and take as long
 * for 80 blocks as for 256: with either on, items x slices should reach the chip's 256 CUs. */
/* ... and how many items fill it best depends on the picture (same box, resident 1080p clip, side chain on 128 CUs: 10 items 188 Gpx/s,
 * 16: 204, 24: 215, 28: 218, 32: 218; 2160p: 6 items 226, 8: 241, 10: 241, 12: 249, 14: 252, 16: 251, 20: 218 -- gpurun_out/r04_s3h, r04_s3i):
 * launches, the waits between the chains and the host's per-step work are paid per step, whatever it holds.  About 245 000 macroblocks
 * per step, never fewer than 15 items: 30 at 1080p, 15 at 2160p.  The price is delay (a unit and a half: 1350 pictures at 1080p) and
 * pool memory (include/x264.h), both capped below by what the device holds. */
static int shim_auto_gop_batch(const x264_param_t *p, int n_slices) {
    const long nmb = (long)((p->i_width + 15) / 16) * (long)((p->i_height + 15) / 16);
    int g = (int)(244800 / (nmb > 0 ? nmb : 1));
    if (g < 15) g = 15;
    if (p->hip.i_deblock || p->hip.i_intra_modes >= 3) {
        const int gw = (256 + n_slices - 1) / n_slices;
        if (g < gw) g = gw;
    }
    return g > VENC_MAX_ITEMS ? VENC_MAX_ITEMS : g;
}
/* HBM an encoder of gop_batch items takes: per pooled frame (2 x gop_batch x keyint of them) its input picture and its output
 * staging slot (the conformance bound of 400 bytes per macroblock), per item the engine's rings (reconstructions, levels, records,
 * entropy buffers: ~22 KB per macroblock) */
static double shim_device_bytes(int w, int h, int keyint, int gop_batch, int subpel) {
    const double nmb = (double)((w + 15) / 16) * (double)((h + 15) / 16);
    const double per_frame = 1.5 * w * h + 400.0 * nmb + 8192.0;
    const double per_item = nmb * (22.0 * 1024.0 + (subpel ? 3.0 * 256.0 : 0.0));
    return 2.0 * gop_batch * keyint * per_frame + gop_batch * per_item;
}

x264_t *x264_encoder_open(x264_param_t *param) {
    if (!param || param->i_width <= 0 || param->i_height <= 0) return NULL;
    x264_t *h = (x264_t *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->param = *param;
    int qp = param->rc.i_rc_method == X264_RC_CQP ? param->rc.i_qp_constant : (int)lroundf(param->rc.f_rf_constant);
    if (qp < 12) qp = 12;
    if (qp > 51) qp = 51;
    h->qp = qp;
    h->keyint = param->i_keyint_max > 0 ? param->i_keyint_max : 30;
    h->gop_counter = param->hip.i_gop_base > 0 ? param->hip.i_gop_base : 0;
    {
        const int ns = param->i_slice_count > 0 ? param->i_slice_count : 1;
        h->gop_batch = param->hip.i_gop_batch > 0 ? param->hip.i_gop_batch : param->b_zerolatency ? 1 : shim_auto_gop_batch(param, ns);
        if (h->gop_batch > VENC_MAX_ITEMS) h->gop_batch = VENC_MAX_ITEMS;
        /* The unit (gop_batch x keyint frames) is the encoder's delay and half its frame pool: cap it by what the device holds.  Without
         * this a long keyint -- libx264's own default is 250 -- asked for tens of GB and open or the first upload failed (round 3).
         * At most half of the device's free memory; never below one GOP at a time. */
        size_t free_b = 0, total_b = 0;
        if (h->gop_batch > 1 && venc_device_memory(param->hip.i_device, &free_b, &total_b) == 0 && free_b > 0) {
            const int asked = h->gop_batch;
            double budget = 0.5 * (double)free_b;
            if (getenv("VENC_POOL_BUDGET_MB") && atof(getenv("VENC_POOL_BUDGET_MB")) * 1e6 < budget) budget = atof(getenv("VENC_POOL_BUDGET_MB")) * 1e6; /* (tests: the cap without the memory) */
            while (h->gop_batch > 1 && shim_device_bytes(param->i_width, param->i_height, h->keyint, h->gop_batch, param->hip.i_subpel) > budget) h->gop_batch--;
            if (h->gop_batch < asked && getenv("VENC_DEBUG"))
                fprintf(stderr, "x264 [venc]: gop_batch %d -> %d (%.1f GB of device memory free)\n", asked, h->gop_batch, (double)free_b / 1e9);
        }
    }
    h->input_depth = param->hip.i_input_depth > 1 ? param->hip.i_input_depth : 1;
    h->enc_trace = getenv("VENC_TRACE") != NULL;
    if (h->gop_batch > VENC_MAX_ITEMS) h->gop_batch = VENC_MAX_ITEMS;
    if (h->gop_batch > 1) {
        h->unit = h->gop_batch * h->keyint;
        /* A picture is handed back a unit AND A HALF after it came in.  A unit is launched when its last picture has arrived, and its first
         * GOP is complete only with its last step: with a delay of exactly one unit the call that owed the unit's first picture waited out the
         * whole unit on the GPU -- 36 ms per 960-picture unit with every preset tool, 35 of the 110 us per call of the reference's own
         * sequence (VENC_TRACE) -- while the copy engine, which bounds that sequence, stood still.  Half a unit of further pictures (38 ms of
         * ingest at 1080p) covers it; the pool of two units holds delay + 1 pictures as before. */
        h->delay = h->unit + h->unit / 2;
        h->pool = 2 * h->unit;
    } else {
        h->unit = 1;
        h->delay = param->b_zerolatency ? 0 : 1;
        h->pool = 4;
    }
    venc_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.width = param->i_width;
    cfg.height = param->i_height;
    cfg.qp = qp;
    cfg.n_slices = param->i_slice_count > 0 ? param->i_slice_count : 1;
    cfg.max_items = h->gop_batch;
    cfg.in_slots = h->pool;
    cfg.out_slots = h->pool;
    cfg.device = param->hip.i_device;
    cfg.intra_modes = param->hip.i_intra_modes < 0 ? 0 : param->hip.i_intra_modes > 3 ? 3 : param->hip.i_intra_modes;
    cfg.deblock = param->hip.i_deblock != 0;
    /* sub-sample refinement: whole-picture encoders only (the six-tap filter reaches three rows past a band's halo) */
    cfg.subpel = (param->hip.i_slice_count <= 0 && param->hip.i_slice_first <= 0 && param->hip.i_rccl_world <= 0) ? param->hip.i_subpel : 0;
    /* X264_RC_CRF (what the reference asks for, av_encode.c:400-401): the rate factor is the base QP and every picture's QP
     * follows spec/CODEC.md section 1a -- a frame-level approximation of libx264's CRF, decided on the GPU.  Slice-band
     * encoders keep the fixed QP (the rule needs the whole picture's SAD sum). */
    cfg.rc_mode = param->rc.i_rc_method == X264_RC_CRF && param->hip.i_slice_count <= 0 && param->hip.i_slice_first <= 0 && param->hip.i_rccl_world <= 0;
    if (venc_open(&cfg, &h->ctx) != 0) {
        fprintf(stderr, "x264 [venc]: %s\n", venc_last_error());
        free(h);
        return NULL;
    }
    if (h->input_depth > 1) (void)venc_set_copy_streams(h->ctx, 2); /* two pictures in flight: their copies on two streams in turn */
    h->ring = (shim_frame *)calloc((size_t)h->pool, sizeof(shim_frame));
    h->seq.width = param->i_width;
    h->seq.height = param->i_height;
    h->seq.qp = qp;
    h->seq.fps_num = param->i_fps_num;
    h->seq.fps_den = param->i_fps_den;
    h->seq.sar_w = param->vui.i_sar_width;
    h->seq.sar_h = param->vui.i_sar_height;
    h->nal_cap = 2 + venc_n_slices(h->ctx);
    if (param->hip.i_rccl_world > 0 && !param->hip.pf_exchange) {
        /* C-level slice-parallel mode: the engine joins the communicator and takes the band of its rank */
        const int ns = venc_n_slices(h->ctx), world = param->hip.i_rccl_world, rank = param->hip.i_rccl_rank;
        if (venc_rccl_init_rank(h->ctx, param->hip.rccl_id, world, rank) != 0) {
            fprintf(stderr, "x264 [venc]: %s\n", venc_last_error());
            x264_encoder_close(h);
            return NULL;
        }
        h->rccl = 1;
        h->band_first = rank * (ns / world) + (rank < ns % world ? rank : ns % world);
        h->band_count = ns / world + (rank < ns % world ? 1 : 0);
    } else if (param->hip.i_slice_count > 0 || param->hip.i_slice_first > 0) {
        const int ns = venc_n_slices(h->ctx);
        h->band_first = param->hip.i_slice_first;
        h->band_count = param->hip.i_slice_count > 0 ? param->hip.i_slice_count : ns - param->hip.i_slice_first;
        if (h->band_count < ns && !param->hip.pf_exchange) {
            fprintf(stderr, "x264 [venc]: a slice band needs hip.pf_exchange (its P pictures read the other bands' rows)\n");
            x264_encoder_close(h);
            return NULL;
        }
        if (venc_set_band(h->ctx, h->band_first, h->band_count) != 0) {
            fprintf(stderr, "x264 [venc]: %s\n", venc_last_error());
            x264_encoder_close(h);
            return NULL;
        }
    }
    if (!h->ring) {
        x264_encoder_close(h);
        return NULL;
    }
    for (int i = 0; i < h->pool; i++) {
        h->ring[i].nals = (x264_nal_t *)calloc((size_t)h->nal_cap, sizeof(x264_nal_t));
        if (!h->ring[i].nals) {
            x264_encoder_close(h);
            return NULL;
        }
    }
    /* a packaging thread for GOP-batched encoders (VENC_HOST_THREADS=1 keeps everything on the caller's thread) */
    if (h->gop_batch > 1 && !(getenv("VENC_HOST_THREADS") && atoi(getenv("VENC_HOST_THREADS")) <= 1)) {
        pthread_mutex_init(&h->mtx, NULL);
        pthread_cond_init(&h->cv_work, NULL);
        pthread_cond_init(&h->cv_done, NULL);
        h->worker_on = 1;
        if (pthread_create(&h->worker, NULL, packaging_thread, h) != 0) {
            h->worker_on = 0;
            pthread_mutex_destroy(&h->mtx);
            pthread_cond_destroy(&h->cv_work);
            pthread_cond_destroy(&h->cv_done);
        } else {
            /* helpers for the heavy steps: VENC_HOST_THREADS = packaging threads in all (default 3: the thread above + 2 helpers) */
            int want = getenv("VENC_HOST_THREADS") ? atoi(getenv("VENC_HOST_THREADS")) - 2 : 2;
            if (want > SHIM_MAX_HELPERS) want = SHIM_MAX_HELPERS;
            if (want > h->gop_batch - 1) want = h->gop_batch - 1;
            h->heavy_bytes = getenv("VENC_PACK_HEAVY_KB") ? (size_t)atol(getenv("VENC_PACK_HEAVY_KB")) << 10 : SHIM_HEAVY_STEP_BYTES;
            pthread_mutex_init(&h->jmtx, NULL);
            pthread_cond_init(&h->jcv_work, NULL);
            pthread_cond_init(&h->jcv_done, NULL);
            for (int i = 0; i < want; i++) {
                if (pthread_create(&h->helper[h->n_helpers], NULL, packaging_helper, h) != 0) break;
                h->n_helpers++;
            }
        }
    }
    return h;
}

void x264_encoder_close(x264_t *h) {
    if (!h) return;
    if (h->enc_trace && h->enc_ms[4] > 0)
        fprintf(stderr, "venc trace: x264_encoder_encode, %.0f calls with a picture: per call upload %.1f us, wait for the copy %.1f us, launches %.1f us, delivery %.1f us\n",
                h->enc_ms[4], 1e3 * h->enc_ms[0] / h->enc_ms[4], 1e3 * h->enc_ms[1] / h->enc_ms[4], 1e3 * h->enc_ms[2] / h->enc_ms[4], 1e3 * h->enc_ms[3] / h->enc_ms[4]);
    if (h->worker_on) {
        pthread_mutex_lock(&h->mtx);
        h->w_stop = 1;
        pthread_cond_broadcast(&h->cv_work);
        pthread_mutex_unlock(&h->mtx);
        pthread_join(h->worker, NULL);
        pthread_mutex_lock(&h->jmtx);
        h->j_stop = 1;
        pthread_cond_broadcast(&h->jcv_work);
        pthread_mutex_unlock(&h->jmtx);
        for (int i = 0; i < h->n_helpers; i++) pthread_join(h->helper[i], NULL);
        h->n_helpers = 0;
        pthread_mutex_destroy(&h->jmtx);
        pthread_cond_destroy(&h->jcv_work);
        pthread_cond_destroy(&h->jcv_done);
        pthread_mutex_destroy(&h->mtx);
        pthread_cond_destroy(&h->cv_work);
        pthread_cond_destroy(&h->cv_done);
        h->worker_on = 0;
    }
    venc_close(h->ctx);
    if (h->ring)
        for (int i = 0; i < h->pool; i++) {
            free(h->ring[i].payload);
            free(h->ring[i].nals);
        }
    free(h->ring);
    free(h);
}

int x264_encoder_delayed_frames(x264_t *h) { return h ? (int)(h->n_in - h->n_out) : 0; }
/* closed GOPs this encoder advances in lock-step (what hip.i_gop_batch = 0 resolved to, after the memory cap) */
int x264_venc_gop_batch(x264_t *h) { return h ? h->gop_batch : 0; }

/* launch every complete scheduling unit that has not been launched yet;
 * `flush` also launches a trailing partial unit */
static int launch_pending(x264_t *h, int flush) {
    while (h->n_launched < h->n_in) {
        const int64_t avail = h->n_in - h->n_launched;
        if (avail < h->unit && !flush) break;
        const int n = (int)(avail < h->unit ? avail : h->unit);
        if (h->gop_batch == 1) {
            const int64_t f = h->n_launched;
            shim_frame *fr = &h->ring[f % h->pool];
            int in_slot = h->external_input ? (int)(f % h->ext_clip) : (int)(f % h->pool), out_slot = (int)(f % h->pool);
            int idr_id = (int)(h->gop_counter & 1);
            if (venc_encode_step(h->ctx, 1, &in_slot, &out_slot, fr->is_idr, (int)(f % h->keyint) & 15, &idr_id) != 0) return -1;
            if (fr->is_idr) h->gop_counter++;
            fr->ticket = venc_mark(h->ctx);
            if (fr->ticket < 0) return -1;
            /* frame f+1, if it comes, is a P picture unless it opens a GOP (every band's owner decides alike) */
            if ((f + 1) % h->keyint != 0 && exchange_after_step(h, 1) != 0) return -1;
        } else {
            const int gops = (n + h->keyint - 1) / h->keyint, rem = n % h->keyint;
            for (int t = 0; t < h->keyint && t < n; t++) {
                int in_slots[VENC_MAX_ITEMS], out_slots[VENC_MAX_ITEMS], idr_ids[VENC_MAX_ITEMS];
                int items = 0;
                for (int g = 0; g < gops; g++) {
                    if (g == gops - 1 && rem && t >= rem) break;
                    const int64_t f = h->n_launched + (int64_t)g * h->keyint + t;
                    in_slots[items] = h->external_input ? (int)(f % h->ext_clip) : (int)(f % h->pool);
                    out_slots[items] = (int)(f % h->pool);
                    idr_ids[items] = (int)((h->gop_counter + g) & 1);
                    items++;
                }
                if (!items) break;
                if (venc_encode_step(h->ctx, items, in_slots, out_slots, t == 0, t & 15, idr_ids) != 0) return -1;
                const int ticket = venc_mark(h->ctx);
                if (ticket < 0) return -1;
                for (int g = 0; g < items; g++) h->ring[(h->n_launched + (int64_t)g * h->keyint + t) % h->pool].ticket = ticket;
                if (t + 1 < h->keyint && t + 1 < n && exchange_after_step(h, items) != 0) return -1;
            }
            h->gop_counter += gops;
        }
        h->n_launched += n;
        if (h->worker_on) {
            if (venc_flush(h->ctx) != 0) return -1; /* every ticket of the unit is now recorded on the GPU's streams */
            pthread_mutex_lock(&h->mtx);
            h->w_launched = h->n_launched;
            pthread_cond_signal(&h->cv_work);
            pthread_mutex_unlock(&h->mtx);
        }
    }
    return 0;
}

/* wait for frame f on the GPU and package its slice RBSPs into NALs (host C, a4.6) */
static int assemble_frame_opt(x264_t *h, int64_t f, int step_done) {
    shim_frame *fr = &h->ring[f % h->pool];
    if (fr->assembled) return 0;
    /* the packaging thread only waits on tickets whose work has been flushed to the GPU (launch_pending ends every unit
     * with venc_flush): venc_wait_issued touches no engine state.  (step_done: the caller has waited for the frame's step and
     * fetched its bytes.) */
    if (!step_done && (h->worker_on ? venc_wait_issued(h->ctx, fr->ticket) : venc_wait(h->ctx, fr->ticket)) != 0) return -1;
    venc_frame vf;
    if (venc_get_frame(h->ctx, fr->out_slot, &vf) != 0) return -1;
    const size_t need = 256 + (size_t)vf.n_slices * 16 + (size_t)vf.total_bytes * 3 / 2;
    if (need > fr->payload_cap) {
        free(fr->payload);
        fr->payload_cap = need + need / 4;
        fr->payload = (uint8_t *)malloc(fr->payload_cap);
        if (!fr->payload) {
            fr->payload_cap = 0;
            return -1;
        }
    }
    size_t o = 0;
    int n = 0;
    const int annexb = h->param.b_annexb;
    if (fr->is_idr && h->band_first == 0) {
        uint8_t rbsp[96];
        size_t k = h264_write_sps(&h->seq, rbsp, sizeof(rbsp));
        size_t sz = h264_wrap_nal(fr->payload + o, annexb, NAL_PRIORITY_HIGHEST, NAL_SPS, rbsp, k);
        fr->nals[n++] = (x264_nal_t){NAL_PRIORITY_HIGHEST, NAL_SPS, (int)sz, fr->payload + o};
        o += sz;
        k = h264_write_pps(&h->seq, rbsp, sizeof(rbsp));
        sz = h264_wrap_nal(fr->payload + o, annexb, NAL_PRIORITY_HIGHEST, NAL_PPS, rbsp, k);
        fr->nals[n++] = (x264_nal_t){NAL_PRIORITY_HIGHEST, NAL_PPS, (int)sz, fr->payload + o};
        o += sz;
    }
    const int ref_idc = fr->is_idr ? NAL_PRIORITY_HIGHEST : NAL_PRIORITY_HIGH;
    const int type = fr->is_idr ? NAL_SLICE_IDR : NAL_SLICE;
    for (uint32_t s = 0; s < vf.n_slices && n < h->nal_cap; s++) {
        if (h->band_count && vf.slices[s].nbytes == 0) continue; /* another band's slice */
        const size_t sz = h264_wrap_nal(fr->payload + o, annexb, ref_idc, type, vf.data + vf.slices[s].byte_off, vf.slices[s].nbytes);
        fr->nals[n++] = (x264_nal_t){ref_idc, type, (int)sz, fr->payload + o};
        o += sz;
    }
    fr->n_nals = n;
    fr->payload_size = (int)o;
    if (h->worker_on) {
        pthread_mutex_lock(&h->mtx);
        fr->assembled = 1;
        pthread_cond_broadcast(&h->cv_done);
        pthread_mutex_unlock(&h->mtx);
    } else
        fr->assembled = 1;
    return 0;
}

static int assemble_frame(x264_t *h, int64_t f) { return assemble_frame_opt(h, f, 0); }

/* a job's frames, taken one at a time by whoever is free (the packaging thread and its helpers) */
static void pack_job_share(x264_t *h, int gen) {
    for (;;) {
        int64_t f = -1;
        pthread_mutex_lock(&h->jmtx);
        if (h->j_gen == gen && h->j_next < h->j_n) f = h->j_frames[h->j_next++];
        pthread_mutex_unlock(&h->jmtx);
        if (f < 0) return;
        const int err = assemble_frame_opt(h, f, 1);
        pthread_mutex_lock(&h->jmtx);
        if (err) h->j_err = 1;
        if (++h->j_done == h->j_n) pthread_cond_broadcast(&h->jcv_done);
        pthread_mutex_unlock(&h->jmtx);
    }
}

static void *packaging_helper(void *arg) {
    x264_t *h = (x264_t *)arg;
    (void)venc_bind_thread(h->ctx);
    int seen = 0;
    for (;;) {
        pthread_mutex_lock(&h->jmtx);
        while (!h->j_stop && h->j_gen == seen) pthread_cond_wait(&h->jcv_work, &h->jmtx);
        const int stop = h->j_stop;
        seen = h->j_gen;
        pthread_mutex_unlock(&h->jmtx);
        if (stop) break;
        pack_job_share(h, seen);
    }
    return NULL;
}

/* the frames base + g * keyint + t, g < gops, of a step that is complete and on the host: alone, or -- a heavy step -- with the helpers */
static int assemble_step(x264_t *h, int64_t base, int t, int gops, int n) {
    int64_t fs[64];
    int cnt = 0;
    for (int g = 0; g < gops && cnt < 64; g++) {
        const int64_t k = (int64_t)g * h->keyint + t;
        if (k < n) fs[cnt++] = base + k;
    }
    if (!cnt) return 0;
    int heavy = 0;
    if (h->worker_on && h->n_helpers > 0 && cnt >= 2) {
        venc_frame vf;
        const shim_frame *f0 = &h->ring[fs[0] % h->pool];
        heavy = venc_get_frame(h->ctx, f0->out_slot, &vf) == 0 && (size_t)vf.total_bytes * (size_t)cnt >= h->heavy_bytes;
    }
    if (!heavy) {
        for (int i = 0; i < cnt; i++)
            if (assemble_frame_opt(h, fs[i], h->worker_on) != 0) return -1;
        return 0;
    }
    pthread_mutex_lock(&h->jmtx);
    memcpy(h->j_frames, fs, (size_t)cnt * sizeof(fs[0]));
    h->j_n = cnt, h->j_next = 0, h->j_done = 0, h->j_err = 0;
    const int gen = ++h->j_gen;
    pthread_cond_broadcast(&h->jcv_work);
    pthread_mutex_unlock(&h->jmtx);
    pack_job_share(h, gen);
    pthread_mutex_lock(&h->jmtx);
    while (h->j_done < h->j_n) pthread_cond_wait(&h->jcv_done, &h->jmtx);
    const int err = h->j_err;
    pthread_mutex_unlock(&h->jmtx);
    return err ? -1 : 0;
}

static int assemble_next_unit(x264_t *h);

/* packaging thread: units in launch order, frames of a unit in completion order */
static void *packaging_thread(void *arg) {
    x264_t *h = (x264_t *)arg;
    (void)venc_bind_thread(h->ctx);
    for (;;) {
        pthread_mutex_lock(&h->mtx);
        while (!h->w_stop && h->n_asm >= h->w_launched) pthread_cond_wait(&h->cv_work, &h->mtx);
        const int stop = h->w_stop;
        pthread_mutex_unlock(&h->mtx);
        if (stop) break;
        if (assemble_next_unit(h) != 0) {
            fprintf(stderr, "x264 [venc]: packaging a unit failed: %s\n", venc_last_error()); /* (the reference logs and carries on, av_encode.c:971) */
            pthread_mutex_lock(&h->mtx);
            h->w_err = 1;
            pthread_cond_broadcast(&h->cv_done);
            pthread_mutex_unlock(&h->mtx);
            break;
        }
    }
    return NULL;
}

/* caller's thread: block until frame f is packaged (or the packaging thread failed) */
static int wait_assembled(x264_t *h, int64_t f) {
    shim_frame *fr = &h->ring[f % h->pool];
    int err;
    pthread_mutex_lock(&h->mtx);
    while (!fr->assembled && !h->w_err) pthread_cond_wait(&h->cv_done, &h->mtx);
    err = h->w_err;
    pthread_mutex_unlock(&h->mtx);
    return err ? -1 : 0;
}

/* package the oldest launched unit in the order its frames complete on the GPU
 * (frame index t of every GOP after step t), so the host works while later steps run */
static int assemble_next_unit(x264_t *h) {
    const int64_t launched = h->worker_on ? h->w_launched : h->n_launched;
    if (h->n_asm >= launched) return 0;
    const int64_t base = h->n_asm;
    const int64_t left = launched - base;
    const int n = (int)(left < h->unit ? left : h->unit);
    if (h->gop_batch == 1) {
        if (assemble_frame(h, base) != 0) return -1;
    } else {
        const int gops = (n + h->keyint - 1) / h->keyint;
        for (int t = 0; t < h->keyint; t++) {
            /* the frames of step t (one per GOP, keyint output slots apart): wait for the step, bring their bytes to the
             * host with one strided copy (engines that stage their output in HBM), then package them */
            int cnt = 0;
            for (int g = 0; g < gops; g++)
                if ((int64_t)g * h->keyint + t < n) cnt++;
            if (cnt > 0) {
                shim_frame *f0 = &h->ring[(base + t) % h->pool];
                if ((h->worker_on ? venc_wait_issued(h->ctx, f0->ticket) : venc_wait(h->ctx, f0->ticket)) != 0) return -1;
                /* (a unit that follows a flushed partial unit does not start on a multiple of the unit and may wrap around the pool: the
                 * strided fetch goes as far as the slots run on, the frames behind the wrap as a second one) */
                int run = 0;
                while (run < cnt && (int64_t)f0->out_slot + (int64_t)run * h->keyint < h->pool) run++;
                if (venc_fetch(h->ctx, f0->out_slot, h->keyint, run) != 0) return -1;
                if (run < cnt && venc_fetch(h->ctx, (int)((base + t + (int64_t)run * h->keyint) % h->pool), h->keyint, cnt - run) != 0) return -1;
            }
            if (assemble_step(h, base, t, gops, n) != 0) return -1;
            if (h->trace_t0 > 0 && (t % 5 == 0 || t == h->keyint - 1))
                fprintf(stderr, "venc trace:     step %d of unit at frame %lld packaged at +%.3f ms\n", t, (long long)base, now_ms() - h->trace_t0);
        }
    }
    if (h->worker_on) {
        pthread_mutex_lock(&h->mtx);
        h->n_asm += n;
        pthread_cond_broadcast(&h->cv_done);
        pthread_mutex_unlock(&h->mtx);
    } else
        h->n_asm += n;
    return 0;
}

/* hand frame n_out to the caller */
static int deliver(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out) {
    shim_frame *fr = &h->ring[h->n_out % h->pool];
    if ((h->worker_on ? wait_assembled(h, h->n_out) : assemble_frame(h, h->n_out)) != 0) return -1;
    if (pic_out) {
        pic_out->i_type = fr->is_idr ? X264_TYPE_IDR : X264_TYPE_P;
        pic_out->b_keyframe = fr->is_idr;
        pic_out->i_pts = fr->pts;
        pic_out->i_dts = fr->pts;
    }
    *pp_nal = fr->nals;
    *pi_nal = fr->n_nals;
    h->n_out++;
    if (!h->worker_on && h->n_asm < h->n_out) h->n_asm = h->n_out;
    return fr->payload_size;
}

/* negative returns of the encode call say where they come from when VENC_DEBUG is set (the reference logs and carries on, av_encode.c:971) */
static int shim_fail(int where) {
    if (getenv("VENC_DEBUG")) fprintf(stderr, "x264 [venc]: x264_encoder_encode failed at check %d: %s\n", where, venc_last_error());
    return -1;
}
int x264_encoder_encode(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_in, x264_picture_t *pic_out) {
    if (!h || !pp_nal || !pi_nal) return shim_fail(1);
    *pp_nal = NULL;
    *pi_nal = 0;
    if (pic_in) {
        if (h->external_input) return shim_fail(2);
        if (h->n_in - h->n_out >= h->pool) return shim_fail(3); /* cannot happen: delay < pool */
        const int64_t f = h->n_in;
        shim_frame *fr = &h->ring[f % h->pool];
        fr->pts = pic_in->i_pts;
        fr->is_idr = (f % h->keyint) == 0;
        fr->out_slot = (int)(f % h->pool);
        fr->ticket = -1;
        fr->assembled = 0;
        const double tt0 = h->enc_trace ? now_ms() : 0;
        if (venc_upload(h->ctx, (int)(f % h->pool), pic_in->img.plane[0], pic_in->img.i_stride[0], pic_in->img.plane[1],
                        pic_in->img.i_stride[1], pic_in->img.plane[2], pic_in->img.i_stride[2]) != 0)
            return shim_fail(4);
        const double tt1 = h->enc_trace ? now_ms() : 0;
        /* WHILE THE COPY FLIES (round 4): the unit's kernel launches, if this picture completes one (they wait for the pictures' copy
         * events on the GPU), and the hand-over of the output frame that is due.  Round 3 waited for the copy first and did both
         * behind it: ~6 us of every ~81 us call with the reference's one pic_in, whose copy (71.6 us alone at 1080p) cannot start
         * before the call and must end before it returns. */
        h->n_in++;
        int ret = 0;
        if (launch_pending(h, 0) != 0) ret = shim_fail(7);
        const double tt2 = h->enc_trace ? now_ms() : 0;
        if (ret == 0 && h->n_in - h->n_out > h->delay) ret = deliver(h, pp_nal, pi_nal, pic_out);
        const double tt3 = h->enc_trace ? now_ms() : 0;
        /* the caller rewrites pic_in right after this call (av_encode.c:545): wait for the ingest copy -- on its own stream,
         * so this does not wait for the kernels of earlier frames, which keep running beside the copy of the next frame.  On every
         * path, also a failed one: the picture must not change under a copy in flight. */
        if (h->input_depth > 1) { /* hip.i_input_depth: the picture handed over input_depth - 1 calls ago is the next one the caller fills */
            if (f >= h->input_depth - 1 && venc_upload_sync_slot(h->ctx, (int)((f - (h->input_depth - 1)) % h->pool)) != 0) return shim_fail(5);
        } else if (venc_upload_sync(h->ctx) != 0)
            return shim_fail(6);
        if (h->enc_trace) {
            const double tt4 = now_ms();
            h->enc_ms[0] += tt1 - tt0, h->enc_ms[1] += tt4 - tt3, h->enc_ms[2] += tt2 - tt1, h->enc_ms[3] += tt3 - tt2, h->enc_ms[4] += 1;
        }
        return ret;
    } else {
        if (h->n_in == h->n_out) return 0;
        if (launch_pending(h, 1) != 0) return shim_fail(8);
    }
    return deliver(h, pp_nal, pi_nal, pic_out);
}

/* restart the frame numbering (resident calls; after a failed call).  The packaging thread is idle or is made idle first:
 * it owns n_asm while units are in flight. */
static void reset_counters(x264_t *h) {
    if (h->worker_on) {
        pthread_mutex_lock(&h->mtx);
        while (h->n_asm < h->w_launched && !h->w_err) pthread_cond_wait(&h->cv_done, &h->mtx);
        h->n_in = h->n_out = h->n_launched = h->n_asm = 0;
        h->w_launched = 0;
        pthread_mutex_unlock(&h->mtx);
    } else
        h->n_in = h->n_out = h->n_launched = h->n_asm = 0;
}

int x264_venc_encode_resident(x264_t *h, const void *d_frames, uint64_t frame_stride, int n_frames, int64_t first_pts,
                              x264_venc_sink sink, void *user) {
    return x264_venc_encode_resident_loop(h, d_frames, frame_stride, n_frames, 1, first_pts, sink, user);
}

static int resident_loop(x264_t *h, int n_frames, int64_t first_pts, x264_venc_sink sink, void *user);

int x264_venc_encode_resident_loop(x264_t *h, const void *d_frames, uint64_t frame_stride, int clip_frames, int repeats,
                                   int64_t first_pts, x264_venc_sink sink, void *user) {
    if (!h || !d_frames || clip_frames < 1 || repeats < 1 || h->n_in != h->n_out) return -1;
    if ((int64_t)clip_frames * repeats > 0x7fffffff) return -1;
    if (venc_set_input(h->ctx, d_frames, (size_t)frame_stride, clip_frames) != 0) return -1;
    h->external_input = 1;
    h->ext_clip = clip_frames;
    const int r = resident_loop(h, clip_frames * repeats, first_pts, sink, user);
    h->external_input = 0; /* on every path: after an error the encoder takes host pictures again */
    if (r < 0) {
        /* frames still inside belong to the failed call: drop them, so that n_in == n_out holds for the next call */
        (void)venc_sync(h->ctx);
        reset_counters(h);
    }
    return r;
}

static int resident_loop(x264_t *h, int n_frames, int64_t first_pts, x264_venc_sink sink, void *user) {
    /* frame numbering restarts so that resident frame f is input slot f */
    reset_counters(h);
    if (h->w_err) return -1;
    int delivered = 0, f = 0;
    x264_nal_t *nals;
    int nn;
    x264_picture_t po;
    const int trace = getenv("VENC_TRACE") != NULL; /* host-side phase times to stderr */
    double t_launch = 0, t_asm = 0, t_sink = 0, t0 = now_ms(), ta;
    h->trace_t0 = trace ? t0 : 0;
    while (f < n_frames || h->n_out < h->n_in) {
        ta = now_ms();
        /* keep the GPU queue full: register frames while output slots are free */
        while (f < n_frames && h->n_in - h->n_out < h->pool) {
            shim_frame *fr = &h->ring[h->n_in % h->pool];
            fr->pts = first_pts + f;
            fr->is_idr = (f % h->keyint) == 0;
            fr->out_slot = (int)(h->n_in % h->pool);
            fr->ticket = -1;
            fr->assembled = 0;
            h->n_in++;
            f++;
            if (launch_pending(h, 0) != 0) return -1;
        }
        if (f >= n_frames && launch_pending(h, 1) != 0) return -1;
        t_launch += now_ms() - ta;
        ta = now_ms();
        /* package the oldest unit while the GPU works on the next one (the packaging thread does, when there is one), then
         * hand it over in display order */
        int64_t upto;
        if (h->worker_on) {
            upto = h->n_out + h->unit < h->n_launched ? h->n_out + h->unit : h->n_launched;
        } else {
            if (assemble_next_unit(h) != 0) return -1;
            upto = h->n_asm;
        }
        t_asm += now_ms() - ta;
        ta = now_ms();
        while (h->n_out < upto) {
            const int bytes = deliver(h, &nals, &nn, &po);
            if (bytes < 0) return -1;
            if (sink && sink(user, nals, nn, bytes, &po) != 0) return -1;
            delivered++;
        }
        t_sink += now_ms() - ta;
        if (trace)
            fprintf(stderr, "venc trace:   +%.3f ms: %d frames delivered, %lld launched\n", now_ms() - t0, delivered, (long long)h->n_launched);
    }
    if (trace)
        fprintf(stderr, "venc trace: %d frames, total %.3f ms = launch %.3f + package(wait+escape) %.3f + sink %.3f\n", delivered,
                now_ms() - t0, t_launch, t_asm, t_sink);
    return delivered;
}

/* Built-in sink for throughput runs: counts frames and bytes and folds every
 * payload byte into a 64-bit FNV-1a digest (user points at x264_venc_tally). */
int x264_venc_tally_sink(void *user, const x264_nal_t *nals, int nal_count, int payload_bytes, const x264_picture_t *pic_out) {
    x264_venc_tally *t = (x264_venc_tally *)user;
    (void)pic_out;
    if (!t || nal_count < 1) return -1;
    /* FNV-1a folded over 8-byte words (tail bytes one by one): a digest that costs about a memcpy,
     * like the payload copy the reference's muxer makes (av_encode.c:802) */
    uint64_t hsh = t->fnv1a ? t->fnv1a : 1469598103934665603ull;
    const uint8_t *p = nals[0].p_payload;
    int i = 0;
    /* four independent multiply chains over 32-byte stripes, folded at the end of the frame */
    uint64_t h0 = hsh, h1 = hsh ^ 0x9E3779B97F4A7C15ull, h2 = hsh ^ 0xC2B2AE3D27D4EB4Full, h3 = hsh ^ 0x165667B19E3779F9ull;
    for (; i + 32 <= payload_bytes; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        h0 = (h0 ^ w[0]) * 1099511628211ull;
        h1 = (h1 ^ w[1]) * 1099511628211ull;
        h2 = (h2 ^ w[2]) * 1099511628211ull;
        h3 = (h3 ^ w[3]) * 1099511628211ull;
    }
    hsh = (((h0 * 31 + h1) * 31 + h2) * 31 + h3);
    for (; i < payload_bytes; i++) hsh = (hsh ^ p[i]) * 1099511628211ull;
    t->fnv1a = hsh;
    t->frames++;
    t->bytes += (uint64_t)payload_bytes;
    t->keyframes += (uint64_t)(pic_out && pic_out->b_keyframe);
    if (t->pass_frames && t->frames % t->pass_frames == 0 && t->passes < X264_VENC_TALLY_PASSES) {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        t->pass_end_ns[t->passes++] = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
    }
    return 0;
}

/* the reference's loop (av_encode.c:940-980, :1076-1083) over frames in host memory, through the public entry points only */
int x264_venc_encode_host_frames(x264_t *h, const uint8_t *frames, uint64_t frame_stride, int width, int height, int clip_frames,
                                 int repeats, x264_venc_sink sink, void *user) {
    if (!h || !frames || clip_frames < 1 || repeats < 1 || width <= 0 || height <= 0) return -1;
    x264_picture_t pic_in, pic_out;
    memset(&pic_in, 0, sizeof(pic_in));
    memset(&pic_out, 0, sizeof(pic_out));
    pic_in.img.i_csp = X264_CSP_I420;
    pic_in.img.i_plane = 3;
    pic_in.img.i_stride[0] = width;
    pic_in.img.i_stride[1] = pic_in.img.i_stride[2] = width / 2;
    const size_t ysz = (size_t)width * height;
    x264_nal_t *nals;
    int nal_count, delivered = 0;
    for (int64_t k = 0; k < (int64_t)clip_frames * repeats; k++) {
        uint8_t *base = (uint8_t *)frames + (size_t)(k % clip_frames) * frame_stride;
        pic_in.i_type = X264_TYPE_AUTO;
        pic_in.i_pts = k;
        pic_in.img.plane[0] = base;
        pic_in.img.plane[1] = base + ysz;
        pic_in.img.plane[2] = base + ysz + ysz / 4;
        const int payload = x264_encoder_encode(h, &nals, &nal_count, &pic_in, &pic_out);
        if (payload < 0) return -1;
        if (payload > 0) {
            if (sink && sink(user, nals, nal_count, payload, &pic_out) != 0) return -1;
            delivered++;
        }
    }
    while (x264_encoder_delayed_frames(h) > 0) {
        const int payload = x264_encoder_encode(h, &nals, &nal_count, NULL, &pic_out);
        if (payload < 0) return -1;
        if (payload > 0) {
            if (sink && sink(user, nals, nal_count, payload, &pic_out) != 0) return -1;
            delivered++;
        }
    }
    return delivered;
}

void *x264_venc_engine(x264_t *h) { return h ? h->ctx : NULL; }
