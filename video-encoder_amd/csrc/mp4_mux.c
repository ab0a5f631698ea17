/*
 * mp4_mux.c -- minimal ISO-BMFF writer for one avc1 track (include/venc_mp4.h; SURVEY.md section 8f
 * row f1).  Structure and behaviour follow the reference's MP4 functions (av_encode.c:614-834), the
 * box syntax follows ISO/IEC 14496-12 / -15 as commonly documented.  Plain C99, no dependencies.
 */
#define _POSIX_C_SOURCE 200809L
#define _FILE_OFFSET_BITS 64
#include "../../include/venc_mp4.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t *p;
    size_t n, cap;
    int failed;
} buf_t;

static void b_need(buf_t *b, size_t k) {
    if (b->failed || b->n + k <= b->cap) return;
    size_t cap = b->cap ? b->cap : 1024;
    while (cap < b->n + k) cap *= 2;
    uint8_t *q = (uint8_t *)realloc(b->p, cap);
    if (!q) {
        b->failed = 1;
        return;
    }
    b->p = q;
    b->cap = cap;
}
static void b_put(buf_t *b, const void *src, size_t k) {
    b_need(b, k);
    if (b->failed) return;
    memcpy(b->p + b->n, src, k);
    b->n += k;
}
static void b_u8(buf_t *b, unsigned v) {
    const uint8_t x = (uint8_t)v;
    b_put(b, &x, 1);
}
static void b_u16(buf_t *b, unsigned v) {
    const uint8_t x[2] = {(uint8_t)(v >> 8), (uint8_t)v};
    b_put(b, x, 2);
}
static void b_u32(buf_t *b, uint32_t v) {
    const uint8_t x[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    b_put(b, x, 4);
}
static void b_u64(buf_t *b, uint64_t v) {
    b_u32(b, (uint32_t)(v >> 32));
    b_u32(b, (uint32_t)v);
}
static void b_zero(buf_t *b, size_t k) {
    b_need(b, k);
    if (b->failed) return;
    memset(b->p + b->n, 0, k);
    b->n += k;
}
/* box = size(4) type(4) ...; returns the position of the size field for box_end() */
static size_t box_begin(buf_t *b, const char *type) {
    const size_t at = b->n;
    b_u32(b, 0);
    b_put(b, type, 4);
    return at;
}
static size_t fullbox_begin(buf_t *b, const char *type, unsigned version, uint32_t flags) {
    const size_t at = box_begin(b, type);
    b_u32(b, ((uint32_t)version << 24) | (flags & 0xffffffu));
    return at;
}
static void box_end(buf_t *b, size_t at) {
    if (b->failed) return;
    const uint32_t sz = (uint32_t)(b->n - at);
    b->p[at] = (uint8_t)(sz >> 24), b->p[at + 1] = (uint8_t)(sz >> 16), b->p[at + 2] = (uint8_t)(sz >> 8), b->p[at + 3] = (uint8_t)sz;
}

typedef struct {
    uint32_t size;
    uint32_t duration;   /* ticks of the media time scale */
    int32_t cts_offset;  /* ticks */
    uint8_t sync;
} sample_t;

typedef struct {
    uint8_t *data;
    size_t size;
} blob_t;

/* one held-back frame of venc_mp4_mux_video (x264_frame_t, av_encode.c:754-760) */
typedef struct {
    uint8_t *payload_data;
    size_t payload_size;
    x264_nal_t *nal_data;
    size_t nal_count;
    x264_picture_t pic;
} held_frame_t;

struct venc_mp4 {
    FILE *f;
    int width, height;
    int tb_num, tb_den;
    uint32_t timescale; /* time_base num * den (av_encode.c:638) */
    int sar_num, sar_den;
    int configured; /* the first SPS has set profile / compatibility / level (av_encode.c:675, :692-715) */
    uint8_t profile_idc, profile_compat, level_idc;
    blob_t *sps, *pps;
    int n_sps, n_pps;
    sample_t *samples;
    long n_samples, cap_samples;
    uint64_t mdat_pos;   /* file offset of the mdat box */
    uint64_t mdat_bytes; /* sample bytes written */
    held_frame_t prev;
};

static int add_blob(blob_t **list, int *n, const uint8_t *data, size_t size) {
    for (int i = 0; i < *n; i++) /* x264 repeats its parameter sets with every keyframe: store each distinct one once */
        if ((*list)[i].size == size && !memcmp((*list)[i].data, data, size)) return 0;
    blob_t *q = (blob_t *)realloc(*list, (size_t)(*n + 1) * sizeof(blob_t));
    if (!q) return -1;
    *list = q;
    q[*n].data = (uint8_t *)malloc(size ? size : 1);
    if (!q[*n].data) return -1;
    memcpy(q[*n].data, data, size);
    q[*n].size = size;
    (*n)++;
    return 0;
}

venc_mp4 *venc_mp4_open(const char *filename, int width, int height, int time_base_num, int time_base_den, int sar_num, int sar_den) {
    if (!filename || width <= 0 || height <= 0 || time_base_num <= 0 || time_base_den <= 0) {
        fprintf(stderr, "mp4: bad track parameters\n");
        return NULL;
    }
    venc_mp4 *m = (venc_mp4 *)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->f = fopen(filename, "wb");
    if (!m->f) {
        fprintf(stderr, "mp4: failed to create mp4 file %s\n", filename);
        free(m);
        return NULL;
    }
    m->width = width, m->height = height;
    m->tb_num = time_base_num, m->tb_den = time_base_den;
    m->timescale = (uint32_t)time_base_num * (uint32_t)time_base_den;
    m->sar_num = sar_num > 0 ? sar_num : 1, m->sar_den = sar_den > 0 ? sar_den : 1;
    /* ftyp, then an mdat box with a 64-bit size that venc_mp4_close() patches */
    buf_t b = {0};
    const size_t at = box_begin(&b, "ftyp");
    b_put(&b, "isom", 4);
    b_u32(&b, 0x200);
    b_put(&b, "isom", 4);
    b_put(&b, "iso2", 4);
    b_put(&b, "avc1", 4);
    b_put(&b, "mp41", 4);
    box_end(&b, at);
    m->mdat_pos = b.n;
    b_u32(&b, 1); /* size = 1: the real size follows as a 64-bit field */
    b_put(&b, "mdat", 4);
    b_u64(&b, 16);
    const int ok = !b.failed && fwrite(b.p, 1, b.n, m->f) == b.n;
    free(b.p);
    if (!ok) {
        fprintf(stderr, "mp4: write error\n");
        fclose(m->f);
        free(m);
        return NULL;
    }
    return m;
}

static int push_sample(venc_mp4 *m, uint32_t size, int64_t duration, int64_t offset, int sync) {
    if (m->n_samples == m->cap_samples) {
        const long cap = m->cap_samples ? 2 * m->cap_samples : 256;
        sample_t *q = (sample_t *)realloc(m->samples, (size_t)cap * sizeof(sample_t));
        if (!q) return -1;
        m->samples = q;
        m->cap_samples = cap;
    }
    sample_t *s = &m->samples[m->n_samples++];
    s->size = size;
    s->duration = (uint32_t)(duration * m->tb_num); /* see the deviation note in venc_mp4.h */
    s->cts_offset = (int32_t)(offset * m->tb_num);
    s->sync = (uint8_t)(sync != 0);
    return 0;
}

int venc_mp4_write_video_sample(venc_mp4 *m, const x264_nal_t *nals, int nal_count, size_t payload_size, int is_sync_sample,
                                int64_t decode_delta, int64_t composition_offset) {
    if (!m || !nals || nal_count < 1) return -1;
    for (int i = 0; i < nal_count; i++) {
        const x264_nal_t *n = &nals[i];
        if (n->i_payload < 5) {
            fprintf(stderr, "mp4: NAL %d shorter than its length prefix and header\n", i);
            return -1;
        }
        switch (n->i_type) {
            case NAL_SPS:
                /* the first sequence parameter set configures the track: byte 4 is the NAL header, bytes 5..7 are
                 * profile_idc, the constraint flags and level_idc (av_encode.c:692-715) */
                if (!m->configured) {
                    if (n->i_payload < 8) return -1;
                    m->profile_idc = n->p_payload[5];
                    m->profile_compat = n->p_payload[6];
                    m->level_idc = n->p_payload[7];
                    m->configured = 1;
                }
                if (add_blob(&m->sps, &m->n_sps, n->p_payload + 4, (size_t)n->i_payload - 4) != 0) return -1; /* av_encode.c:720 */
                break;
            case NAL_PPS:
                if (add_blob(&m->pps, &m->n_pps, n->p_payload + 4, (size_t)n->i_payload - 4) != 0) return -1; /* av_encode.c:725 */
                break;
            case NAL_FILLER:
                break; /* av_encode.c:729 */
            default: {
                /* everything else is data of the video track: this NAL and all that follow are one sample
                 * (av_encode.c:732-746), length prefixes included (the track declares 4-byte length fields) */
                const uint8_t *start = n->p_payload;
                const ptrdiff_t skipped = start - nals[0].p_payload;
                if (skipped < 0 || (size_t)skipped >= payload_size) {
                    fprintf(stderr, "mp4: NAL payloads are not one contiguous buffer\n");
                    return -1;
                }
                const size_t size = payload_size - (size_t)skipped;
                if (size > 0xffffffffu || fwrite(start, 1, size, m->f) != size) {
                    fprintf(stderr, "mp4: failed to write sample (NAL %d)\n", i);
                    return -1;
                }
                m->mdat_bytes += size;
                if (push_sample(m, (uint32_t)size, decode_delta, composition_offset, is_sync_sample) != 0) return -1;
                i = nal_count; /* the remaining NALs went into the sample */
            } break;
        }
    }
    return 0;
}

static void held_free(held_frame_t *h) {
    free(h->payload_data);
    free(h->nal_data);
    memset(h, 0, sizeof(*h));
}

int venc_mp4_mux_video(venc_mp4 *m, const x264_nal_t *nals, int nal_count, int payload_size, const x264_picture_t *pic_out) {
    if (!m) return -1;
    held_frame_t *prev = &m->prev;
    if (payload_size > 0) {
        if (!nals || nal_count < 1 || !pic_out) return -1;
        /* with a frame already held back its duration is known now (av_encode.c:771-783) */
        if (prev->payload_size > 0) {
            const int64_t decode_delta = pic_out->i_dts - prev->pic.i_dts;
            const int64_t composition_offset = prev->pic.i_pts - prev->pic.i_dts;
            if (decode_delta <= 0 || composition_offset < 0) {
                fprintf(stderr, "mp4: decode times must increase and must not exceed the presentation time\n");
                return -1;
            }
            if (venc_mp4_write_video_sample(m, prev->nal_data, (int)prev->nal_count, prev->payload_size, prev->pic.b_keyframe, decode_delta,
                                            composition_offset) != 0)
                return -1;
        }
        /* hold the new frame back: deep copy, NAL pointers rebased (av_encode.c:789-812) */
        held_free(prev);
        prev->payload_data = (uint8_t *)malloc((size_t)payload_size);
        prev->nal_data = (x264_nal_t *)malloc((size_t)nal_count * sizeof(x264_nal_t));
        if (!prev->payload_data || !prev->nal_data) {
            fprintf(stderr, "mp4: failed to allocate buffers for the held frame\n");
            held_free(prev);
            return -1;
        }
        memcpy(prev->payload_data, nals[0].p_payload, (size_t)payload_size);
        prev->payload_size = (size_t)payload_size;
        prev->pic = *pic_out;
        prev->nal_count = (size_t)nal_count;
        for (int i = 0; i < nal_count; i++) {
            prev->nal_data[i] = nals[i];
            prev->nal_data[i].p_payload = prev->payload_data + (nals[i].p_payload - nals[0].p_payload);
        }
    } else if (prev->payload_size > 0) {
        /* flush: the last frame may have any duration, the reference gives it 1 (av_encode.c:813-822) */
        const int rc = venc_mp4_write_video_sample(m, prev->nal_data, (int)prev->nal_count, prev->payload_size, prev->pic.b_keyframe, 1,
                                                   prev->pic.i_pts - prev->pic.i_dts);
        held_free(prev);
        return rc;
    }
    return 0;
}

long venc_mp4_sample_count(const venc_mp4 *m) { return m ? m->n_samples : 0; }

static void write_moov(venc_mp4 *m, buf_t *b) {
    uint64_t duration = 0;
    int any_offset = 0;
    long n_sync = 0;
    for (long i = 0; i < m->n_samples; i++) {
        duration += m->samples[i].duration;
        any_offset |= m->samples[i].cts_offset != 0;
        n_sync += m->samples[i].sync;
    }
    static const uint32_t unity[9] = {0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000};
    const size_t moov = box_begin(b, "moov");
    {
        const size_t at = fullbox_begin(b, "mvhd", 0, 0);
        b_u32(b, 0), b_u32(b, 0);         /* creation, modification */
        b_u32(b, m->timescale);           /* the movie uses the track's time scale */
        b_u32(b, (uint32_t)duration);
        b_u32(b, 0x00010000);             /* rate 1.0 */
        b_u16(b, 0x0100);                 /* volume */
        b_zero(b, 10);
        for (int i = 0; i < 9; i++) b_u32(b, unity[i]);
        b_zero(b, 24);
        b_u32(b, 2);                      /* next_track_ID */
        box_end(b, at);
    }
    const size_t trak = box_begin(b, "trak");
    {
        const size_t at = fullbox_begin(b, "tkhd", 0, 7); /* enabled, in movie, in preview */
        b_u32(b, 0), b_u32(b, 0);
        b_u32(b, 1); /* track_ID */
        b_u32(b, 0);
        b_u32(b, (uint32_t)duration);
        b_zero(b, 8);
        b_u16(b, 0), b_u16(b, 0), b_u16(b, 0), b_u16(b, 0); /* layer, alternate group, volume, reserved */
        for (int i = 0; i < 9; i++) b_u32(b, unity[i]);
        /* presentation size in 16.16: the width scaled by the pixel aspect ratio */
        b_u32(b, (uint32_t)(((uint64_t)m->width * (uint64_t)m->sar_num << 16) / (uint64_t)m->sar_den));
        b_u32(b, (uint32_t)m->height << 16);
        box_end(b, at);
    }
    const size_t mdia = box_begin(b, "mdia");
    {
        size_t at = fullbox_begin(b, "mdhd", 0, 0);
        b_u32(b, 0), b_u32(b, 0);
        b_u32(b, m->timescale);
        b_u32(b, (uint32_t)duration);
        b_u16(b, 0x55c4); /* language: und */
        b_u16(b, 0);
        box_end(b, at);
        at = fullbox_begin(b, "hdlr", 0, 0);
        b_u32(b, 0);
        b_put(b, "vide", 4);
        b_zero(b, 12);
        b_put(b, "VideoHandler", 13);
        box_end(b, at);
    }
    const size_t minf = box_begin(b, "minf");
    {
        size_t at = fullbox_begin(b, "vmhd", 0, 1);
        b_zero(b, 8);
        box_end(b, at);
        const size_t dinf = box_begin(b, "dinf");
        at = fullbox_begin(b, "dref", 0, 0);
        b_u32(b, 1);
        const size_t url = fullbox_begin(b, "url ", 0, 1); /* media data in the same file */
        box_end(b, url);
        box_end(b, at);
        box_end(b, dinf);
    }
    const size_t stbl = box_begin(b, "stbl");
    {
        size_t at = fullbox_begin(b, "stsd", 0, 0);
        b_u32(b, 1);
        const size_t avc1 = box_begin(b, "avc1");
        b_zero(b, 6);
        b_u16(b, 1); /* data_reference_index */
        b_zero(b, 16);
        b_u16(b, (unsigned)m->width), b_u16(b, (unsigned)m->height);
        b_u32(b, 0x00480000), b_u32(b, 0x00480000); /* 72 dpi */
        b_u32(b, 0);
        b_u16(b, 1); /* frame_count */
        b_zero(b, 32); /* compressorname */
        b_u16(b, 0x0018);
        b_u16(b, 0xffff);
        {
            const size_t avcc = box_begin(b, "avcC");
            b_u8(b, 1);
            b_u8(b, m->profile_idc), b_u8(b, m->profile_compat), b_u8(b, m->level_idc); /* av_encode.c:710-715 */
            b_u8(b, 0xfc | 3); /* lengthSizeMinusOne = 3: 4-byte NAL lengths (av_encode.c:640) */
            b_u8(b, 0xe0 | (unsigned)(m->n_sps & 31));
            for (int i = 0; i < (m->n_sps & 31); i++) b_u16(b, (unsigned)m->sps[i].size), b_put(b, m->sps[i].data, m->sps[i].size);
            b_u8(b, (unsigned)(m->n_pps & 255));
            for (int i = 0; i < (m->n_pps & 255); i++) b_u16(b, (unsigned)m->pps[i].size), b_put(b, m->pps[i].data, m->pps[i].size);
            box_end(b, avcc);
            const size_t pasp = box_begin(b, "pasp"); /* MP4AddPixelAspectRatio, av_encode.c:646 */
            b_u32(b, (uint32_t)m->sar_num), b_u32(b, (uint32_t)m->sar_den);
            box_end(b, pasp);
        }
        box_end(b, avc1);
        box_end(b, at);

        at = fullbox_begin(b, "stts", 0, 0); /* run-length coded sample durations */
        const size_t cnt_at = b->n;
        b_u32(b, 0);
        uint32_t runs = 0;
        for (long i = 0; i < m->n_samples;) {
            long j = i;
            while (j < m->n_samples && m->samples[j].duration == m->samples[i].duration) j++;
            b_u32(b, (uint32_t)(j - i)), b_u32(b, m->samples[i].duration);
            runs++;
            i = j;
        }
        if (!b->failed) b->p[cnt_at] = (uint8_t)(runs >> 24), b->p[cnt_at + 1] = (uint8_t)(runs >> 16), b->p[cnt_at + 2] = (uint8_t)(runs >> 8), b->p[cnt_at + 3] = (uint8_t)runs;
        box_end(b, at);

        if (any_offset) {
            at = fullbox_begin(b, "ctts", 0, 0);
            const size_t c_at = b->n;
            b_u32(b, 0);
            uint32_t cr = 0;
            for (long i = 0; i < m->n_samples;) {
                long j = i;
                while (j < m->n_samples && m->samples[j].cts_offset == m->samples[i].cts_offset) j++;
                b_u32(b, (uint32_t)(j - i)), b_u32(b, (uint32_t)m->samples[i].cts_offset);
                cr++;
                i = j;
            }
            if (!b->failed) b->p[c_at] = (uint8_t)(cr >> 24), b->p[c_at + 1] = (uint8_t)(cr >> 16), b->p[c_at + 2] = (uint8_t)(cr >> 8), b->p[c_at + 3] = (uint8_t)cr;
            box_end(b, at);
        }

        at = fullbox_begin(b, "stss", 0, 0); /* sync samples = keyframes (av_encode.c:783) */
        b_u32(b, (uint32_t)n_sync);
        for (long i = 0; i < m->n_samples; i++)
            if (m->samples[i].sync) b_u32(b, (uint32_t)(i + 1));
        box_end(b, at);

        at = fullbox_begin(b, "stsc", 0, 0); /* all samples sit back to back in one chunk */
        b_u32(b, m->n_samples ? 1u : 0u);
        if (m->n_samples) b_u32(b, 1), b_u32(b, (uint32_t)m->n_samples), b_u32(b, 1);
        box_end(b, at);

        at = fullbox_begin(b, "stsz", 0, 0);
        b_u32(b, 0);
        b_u32(b, (uint32_t)m->n_samples);
        for (long i = 0; i < m->n_samples; i++) b_u32(b, m->samples[i].size);
        box_end(b, at);

        at = fullbox_begin(b, "co64", 0, 0);
        b_u32(b, m->n_samples ? 1u : 0u);
        if (m->n_samples) b_u64(b, m->mdat_pos + 16);
        box_end(b, at);
    }
    box_end(b, stbl);
    box_end(b, minf);
    box_end(b, mdia);
    box_end(b, trak);
    box_end(b, moov);
}

int venc_mp4_close(venc_mp4 *m) {
    if (!m) return -1;
    int rc = 0;
    if (m->prev.payload_size > 0) rc = venc_mp4_mux_video(m, NULL, 0, 0, NULL); /* a caller that forgot the flush call */
    buf_t b = {0};
    write_moov(m, &b);
    if (b.failed || fwrite(b.p, 1, b.n, m->f) != b.n) rc = -1;
    free(b.p);
    /* patch the 64-bit size of the mdat box */
    const uint64_t total = 16 + m->mdat_bytes;
    uint8_t sz[8];
    for (int i = 0; i < 8; i++) sz[i] = (uint8_t)(total >> (56 - 8 * i));
    if (fseeko(m->f, (off_t)(m->mdat_pos + 8), SEEK_SET) != 0 || fwrite(sz, 1, 8, m->f) != 8) rc = -1;
    if (fclose(m->f) != 0) rc = -1;
    if (rc != 0) fprintf(stderr, "mp4: failed to finish the file\n");
    for (int i = 0; i < m->n_sps; i++) free(m->sps[i].data);
    for (int i = 0; i < m->n_pps; i++) free(m->pps[i].data);
    free(m->sps);
    free(m->pps);
    free(m->samples);
    held_free(&m->prev);
    free(m);
    return rc;
}
