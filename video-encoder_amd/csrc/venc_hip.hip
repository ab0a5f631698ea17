/*
 * venc_hip.hip -- C-ABI launcher of the gfx950 macroblock-loop engine
 * (declared in include/venc_hip.h).  Owns device memory, the HIP stream, the
 * pinned output pool, and the per-step kernel sequence.  No PyTorch, no
 * third-party code: hipcc + libamdhip64 only.
 *
 * Reference boundary served: x264_encoder_encode() at av_encode.c:970 -- the C
 * shim (x264_shim.c) turns each call into venc_upload + venc_encode_step and
 * wraps the slice RBSPs produced here into the NAL array the reference's muxer
 * consumes (av_encode.c:670-834).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> /* types only: the library is bound at run time (venc_rccl.hip) */

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <dirent.h>
#include <sched.h>

#include "../../include/venc_hip.h"
#include "venc_lines.h"
#include "venc_kernels.hip"
static_assert(offsetof(venc_mbinfo, cbp) % 4 == 0 && sizeof(venc_mbinfo) == 32, "luma_block_redo ORs a macroblock's luma cbp bits into the aligned word that starts at cbp");
#include "venc_inter2.hip"
#include "venc_me2.hip"
#ifdef VENC_LEGACY_KERNELS
#include "legacy/venc_me.hip"
#endif
#include "venc_me3.hip"
#include "venc_me4.hip"
#include "venc_subpel.hip"
#include "venc_deblock.hip"
#include "venc_deblock2.hip"
#include "venc_cavlc_p.hip"
#include "venc_intra2.hip"
#include "venc_synth.h"

static thread_local char g_err[512] = "";
#define SET_ERR(...) snprintf(g_err, sizeof(g_err), __VA_ARGS__)
#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            SET_ERR("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (getenv("VENC_DEBUG")) fprintf(stderr, "venc: %s\n", g_err);                   \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

#define VENC_EVENTS_MIN 256 /* completion tickets: one event per step; the ring holds max(this, out_slots + 2) of them, and a step
                              occupies at least one output slot until its frames are delivered, so a live ticket is never recycled */
/* Motion vectors, levels and macroblock records live in a ring of VENC_RING step slots: the entropy chain of step t
 * (second stream) reads slot t % RING while the main chain runs ahead.  Slot t % RING was last read by the entropy
 * chain of step t - RING, so the main stream has to wait for that chain only now and then: every VENC_RING_WAIT
 * steps it waits for the entropy chain of step t + WAIT - 1 - RING, which covers the next WAIT steps.  (With two
 * slots the wait came every step and cost the main stream ~5 us each time -- 3 % of a 1080p step.) */
#define VENC_RING 8
#define VENC_RING_WAIT 4
/* ENTROPY BATCH.  Nothing on the GPU waits for a step's bitstream, so the entropy chain need not run once per step: the
 * launcher collects up to VENC_EBATCH consecutive P steps (ring slots of one aligned group of four, same item count) and
 * launches k_skip / k_cavlc_p / k_scan / k_pack ONCE for all of them, with the grid's item index running over (step, item)
 * pairs.  The chain's kernels are latency bound (a listed unit is one lane's serial bit emission; k_scan and k_pack are
 * dependent round trips): four steps per launch cost little more than one, and the chain stops being co-critical with the
 * main chain (round 1: 3.8 ms against 4.2 ms per 1080p pass).  A wait for a ticket of a step whose batch is still open
 * flushes the batch first, so streaming callers (one frame per call) see batches of one. */
#define VENC_EBATCH 4
/* INTRA AHEAD.  The I pictures of a scheduling unit depend on nothing before them (closed GOPs), and their kernel chain is
 * the slowest of the pass for what it computes: k_intra_dc walks a dependency wavefront on a few CUs for ~110 us.  With a
 * ring of VENC_RECON reconstruction buffers the I step is launched on a THIRD stream as soon as the buffer it will write is
 * idle -- the P step VENC_RECON - 1 before it has finished with it -- and runs beside the last P steps of the previous unit instead
 * of between the units.  The main stream picks the chain up again with the first P step (one event). */
#define VENC_QP_RING 16
#define VENC_RECON 8 /* (the buffer step N writes was read last by step N - 7: at least four steps of overlap with batches of four) */
#define VENC_BHIST 8

struct venc_ctx {
    venc_config cfg;
    venc_geom g;
    venc_bufs b;
    hipStream_t stream;  /* main chain: ingest, k_me, k_me_fix, k_inter / k_intra */
    /* TWO CHAINS (round 4).  The items of a P step are independent prediction chains, and the launch between two searches -- k_me_fix:
     * the exhaustive redo and the follow-up of the deferred transform, a few hundred blocks, 13-17 us of dependent latency -- leaves
     * the chip nearly idle on the critical path.  With two_chain the items' first half runs search -> fix -> search ... on `stream`,
     * the second half the same on `stream_b`: one half's fix runs beside the other half's search.  Everything else that follows the
     * main chain (side chain, intra steps running ahead, ring waits) waits for both.  Eligible steps: whole pictures, the deferred
     * transform, no in-loop filter / refinement / exchange on the main chain; any other step joins stream_b into `stream` first. */
    hipStream_t stream_b;
    hipEvent_t ev_b;
    int two_chain;  /* VENC_CHAINS=2 */
    int b_active;   /* stream_b carries work the main stream has not waited for */
    int b_behind;   /* the main stream ran something stream_b's next search depends on (a step that was not split, a band import, an intra
                       step on the main stream): the next split step makes stream_b wait for the main stream first */
    int last_n_a;   /* items of the main stream's half in the last split step (another split: the chains resynchronise) */
    hipEvent_t ev_a;
    hipStream_t stream2; /* entropy chain of the same step: k_cavlc, k_scan, k_pack -- overlaps the next step's main chain */
    hipStream_t stream_copy; /* host -> HBM ingest (venc_upload): overlaps both kernel chains; the main chain waits for a slot's copy
                                event only when a step reads that slot */
    hipStream_t stream_copy2; /* odd slots' copies when the caller keeps two pictures in flight (venc_set_copy_streams) */
    int n_copy_streams;
    hipEvent_t *ev_copy;     /* [in_slots] copy of the slot's frame complete */
    uint8_t *copy_pending;   /* [in_slots] ev_copy[slot] recorded and not yet waited for by the main chain */
    int last_copy_slot;      /* slot of the most recent venc_upload (-1: none): what venc_upload_sync waits for */
    int prev_copy_slot;      /* the one before (the other copy stream's last when two are in use) */
    hipEvent_t ev_main[2];            /* alternating: main chains of an entropy batch done */
    int ev_main_next;
    hipEvent_t ev_luma[2];            /* alternating: search + luma transform of a P step done (the chroma launch on stream 2 waits) */
    int ev_luma_next;
    hipEvent_t ev_chroma;             /* the last chroma launch on stream 2 */
    int chroma_pending;               /* ev_chroma recorded and not yet waited for by the main stream */
    int fuse_luma;                    /* 3 (default): luma transform inside k_me_pool / k_me_fix, the 4x4 blocks that hold a level left to extra blocks of the
                                         k_me_fix launch, chroma on stream 2; 1: every block coded inside the search (round 2's shape);
                                         VENC_FUSE=0: k_inter_fused on the main chain */
    int fix_phase;                    /* which set of fix counters the next P step fills */
    /* rate control (cfg.rc_mode; spec/CODEC.md section 1a) */
    venc_geom g_idr;                  /* geometry with the quantiser constants of IDR pictures (base - 3) */
    unsigned long long *d_sad_sum;    /* [3][G] per-item sums of best SADs, rotating between P steps */
    int *d_qp_ring;                   /* [VENC_QP_RING][G] QP each (step, item) was coded at; slot = step % 16: written one step ahead, read by the
                                         entropy batch up to a few steps later -- twice the step ring, so that the early write never meets a late read */
    venc_qparams *d_qtab;             /* [52] */
    uint32_t *d_fix_done;             /* [G] */
    int rc_phase;                     /* set the next P step adds to */
    int last_step_p;                  /* the previous step was a P step */
    hipEvent_t ev_side[VENC_RING];    /* per ring slot: the entropy kernels that read the slot's motion vectors, levels and records are done */
    long long side_step[VENC_RING];  /* step whose entropy batch recorded ev_side[slot] (-1: none) */
    /* the open entropy batch: steps whose main chain is enqueued and whose entropy chain is not */
    int eb_max;                       /* steps per batch: min(VENC_EBATCH, VENC_MAX_ITEMS / max_items), env VENC_ENTROPY_BATCH */
    int eb_n;                         /* steps collected */
    int eb_first_slot;                /* ring slot of the first */
    long long eb_first_step;
    venc_step eb_S;                   /* per (step, item) fields of the batch */
    venc_step eb_step[VENC_EBATCH];   /* the steps themselves: their chroma transforms are launched with the batch, one after the other */
    int eb_chroma;                    /* the steps of the batch still owe their chroma transform */
    int eb_tickets[VENC_EBATCH * 2];  /* venc_mark() tickets issued while the batch was open: recorded at the flush */
    int eb_nt;
    int overlap;
    hipEvent_t *ev; /* [n_events] */
    int n_events;
    int ev_done; /* newest ticket known to be complete (-1: none) */
    int ev_next;
    int parity;           /* reconstruction buffer the next step writes; rec_last: the one the last step wrote (the next reference) */
    int rec_last;
    hipStream_t stream3;  /* intra steps running ahead */
    /* EXCHANGE STREAM (slice bands): between venc_band_xbegin and venc_band_xend the band's export, the
     * collective and the import of the peers' rows run on stream_x, behind everything the main stream has done so far.  The next P step's search
     * then goes as TWO launches: the tile rows whose 64-row windows stay inside the own band at once, beside the exchange; the first /
     * last tile row of a band that has a neighbour above / below after ev_x_done.  Everything else that reads the reconstruction
     * follows those on the main stream.  (VENC_XOVERLAP=0: the exchange stays on the main stream, as until round 3.) */
    hipStream_t stream_x;
    hipEvent_t ev_x_main, ev_x_done;
    int x_overlap;          /* VENC_XOVERLAP (default 1) */
    int x_active;           /* between xbegin and xend: band copies go to stream_x */
    int x_pending;          /* ev_x_done recorded and not yet waited for by the main stream */
    long long x_split_steps; /* P steps whose search ran as interior + edge launches */
    hipEvent_t ev_intra;  /* the last intra step on stream3 done */
    int intra_pending;    /* ev_intra recorded and not yet waited for by the main stream */
    int intra_async_last; /* the open entropy batch holds an intra step that ran on stream3 */
    int intra_ahead;      /* 1 (default): intra steps on stream3 (VENC_INTRA_AHEAD=0: on the main stream) */
    /* the last entropy batches: the step each ended with, an event behind its main chains (stream 1) and one behind its
     * chroma transforms (stream 2): what an intra step running ahead waits for */
    struct {
        long long last_step;
        hipEvent_t ev_main, ev_chroma, ev_main_b;
        int has_chroma, has_b;
    } bh[VENC_BHIST];
    int bh_next;
    long long step_no; /* steps issued so far; ring slot of a step = step_no % VENC_RING */
    uint8_t *d_in;        /* internal input pool (may be null) */
    int in_frames;        /* frames addressable through b.in_base */
    uint8_t *h_hdr;       /* pinned, device-mapped: per out slot venc_out_header + slice table (k_scan writes it over PCIe: tens of bytes) */
    uint8_t *h_data;      /* pinned host pool of the slots' data[] */
    uint8_t *d_data;      /* staged engines: HBM buffer k_pack writes; venc_fetch copies the used bytes to h_data by DMA */
    int staged;           /* max_items > 1 (VENC_STAGED=0/1 overrides): k_pack's stores to pinned host memory wait on PCIe and, worse,
                             slow every other kernel's memory traffic while they drain (the search beside an I picture's k_pack ran at
                             a quarter of its speed); HBM stores do neither, and the copy engine moves the bytes without a CU */
    uint8_t *fetched;     /* [out_slots] staged engines: the slot's data is in h_data */
    /* Staged engines keep the bytes on the host only from the fetch of a step to the packaging of its frames, so the pinned pool is a
     * small ROTATING set of host buffers, not one per output slot: round 3 pinned out_slots x out_cap (400 bytes per macroblock:
     * 1.96 GB for the preset's 600 slots at 1080p, 7.8 GB at 2160p, tens of GB for a long keyint) where 40 buffers do.  A frame's bytes
     * stay valid on the host until host_slots further frames have been fetched.  Only venc_fetch / venc_get_frame touch these
     * (one thread: the packaging thread, or the caller's in engines without one). */
    size_t lds_per_block; /* hipDeviceProp_t.sharedMemPerBlock */
    int host_slots;       /* buffers in h_data (direct engines: out_slots, slot = buffer) */
    int *host_of;         /* [out_slots] buffer that holds the slot's bytes (-1: none) */
    int *host_owner;      /* [host_slots] slot whose bytes the buffer holds (-1: none) */
    int host_next;
    hipStream_t stream_d2h;
    venc_slice_out *d_slice;
    int *d_slice_rows;
    uint8_t *d_row_slice;
    uint16_t *d_rank, *d_cand;
    int intra_waves;
    int last_batch_p; /* the last entropy batch coded P pictures (venc_debug_read derives their per-macroblock bits from the entries) */
    venc_step last_batch_S;
    int me_xcd;       /* XCD-aware tile order of the search (VENC_ME_XCD; default for pictures beyond 1080p from 200 000 macroblocks per step, see venc_open) */
    int intra_cw;     /* compute waves of k_intra_dc (the other waves of its 16 stage the records) */
    int intra_legacy; /* VENC_INTRA=legacy: the one-kernel wavefront k_intra */
    int max_slice_mbs;
    int steps_done;
    int band_s0, band_s1; /* slices coded by this context (venc_set_band); whole picture by default */
    int slice_rows[VENC_MAX_SLICES + 1];
    int me_stop, me_fast;
    int inter_legacy; /* VENC_INTER=legacy: the first, half-wave-per-macroblock k_inter */
    int me_mode;  /* 0 pool (default): k_me_pool, successive elimination with in-kernel sums, pooled survivors, three
                     blocks per CU; 3 tile: its predecessor k_me_tile; 1 plane: sums from k_refsum; 2 dense: exhaustive
                     search.  Env VENC_ME=pool|tile|plane|dense (VENC_ME_DENSE=1 = dense). */
    /* slice-parallel mode over RCCL (venc_rccl.hip) */
    ncclComm_t rc_comm;
    int rc_world, rc_rank, rc_items, rc_halo, rc_always, rc_poison;
    void *rc_send, *rc_recv; /* own chunk; one chunk per rank (all-gather layout) */
    size_t rc_item_stride;
    int *rc_flag;
    long long rc_exchanges;
    /* optional per-kernel HIP-event timing (venc_profile) */
    int prof_on;
    int prof_stride, prof_seen[16]; /* event pairs around every prof_stride-th launch of a selected kernel */
    unsigned prof_mask; /* kernels that get event pairs (bit = kernel id); every pair costs the stream a few microseconds */
    hipEvent_t *prof_ev; /* pairs */
    int *prof_kid;
    int prof_n, prof_cap, prof_open;
};

enum { KID_ME = 0, KID_INTER, KID_INTRA, KID_CAVLC, KID_SCAN, KID_PACK, KID_REFSUM, KID_MEFIX, KID_INTER_C, KID_SUBPEL, KID_HPEL, KID_DEBLOCK, KID_COUNT };
/* "k_inter" = k_inter_fused (luma + chroma transforms) */
static const char *const k_kernel_names[KID_COUNT] = {"k_me", "k_inter", "k_intra", "k_cavlc", "k_scan", "k_pack", "k_refsum", "k_me_fix", "k_inter_chroma", "k_subpel", "k_hpel", "k_deblock"};

static void prof_begin(venc_ctx *c, int kid, hipStream_t st) {
    c->prof_open = 0;
    if (!c->prof_on || !((c->prof_mask >> kid) & 1u) || c->prof_n >= c->prof_cap) return;
    if (c->prof_seen[kid]++ % c->prof_stride) return;
    c->prof_open = 1;
    c->prof_kid[c->prof_n] = kid;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n], st);
}
static void prof_end(venc_ctx *c, hipStream_t st) {
    if (!c->prof_open) return;
    c->prof_open = 0;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n + 1], st);
    c->prof_n++;
}

/* host copies of the small quantiser tables (device copies live in venc_tables.h) */
static const int h_mf[6][3] = {{13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554},
                               {9362, 3647, 5825},  {8192, 3355, 5243},  {7282, 2893, 4559}};
static const int h_v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
static const int h_qpc_hi[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

static void fill_qparams(venc_qparams *q, int qp) {
    q->per = qp / 6;
    q->qbits = 15 + qp / 6;
    for (int k = 0; k < 3; k++) {
        q->mf[k] = h_mf[qp % 6][k];
        q->v[k] = h_v[qp % 6][k];
    }
    q->f_intra = (1 << q->qbits) / 3;
    q->f_inter = (1 << q->qbits) / 6;
    for (int k = 0; k < 3; k++) q->zt_inter[k] = ((1 << q->qbits) - q->f_inter - 1) / q->mf[k];
}

static int entropy_flush(venc_ctx *c);
/* the in-loop filter of a step's pictures: one block per (slice, item) -- k_deblock_ls (venc_deblock2.hip): the slice's rows in lock-step,
 * four rows per filter wave + as many helper waves; VENC_DBK=strip: k_deblock (venc_deblock.hip), a wave per macroblock row */
static void launch_deblock(venc_ctx *c, hipStream_t st, const venc_geom &g, const venc_bufs &b, const venc_step &S, int n_items, int planes) {
    int max_rows = 1;
    for (int s = S.slice0; s < S.slice1; s++) max_rows = c->slice_rows[s + 1] - c->slice_rows[s] > max_rows ? c->slice_rows[s + 1] - c->slice_rows[s] : max_rows;
    static const int cap = getenv("VENC_DBK_WAVES") ? atoi(getenv("VENC_DBK_WAVES")) : DBK_MAX_WAVES; /* (diagnostic: fewer rows at a time) */
    const int wmax = cap >= 1 && cap < DBK_MAX_WAVES ? cap : DBK_MAX_WAVES;
    const int waves = max_rows < wmax ? max_rows : wmax;
    /* k_deblock_ls (venc_deblock2.hip): the rows of a slice in lock-step, four rows per wave; VENC_DBK=strip: the first kernel
     * (one wave per row, progress flags) */
    static const bool strip = getenv("VENC_DBK") && !strcmp(getenv("VENC_DBK"), "strip");
    prof_begin(c, KID_DEBLOCK, st);
    if (strip)
        hipLaunchKernelGGL(k_deblock, dim3(S.slice1 - S.slice0, 1, n_items), dim3(64 * waves), (size_t)waves * sizeof(dbk_strip), st, g, b, S, planes);
    else {
        int lw = (max_rows + 3) / 4 < DL_MAX_WAVES / 2 ? (max_rows + 3) / 4 : DL_MAX_WAVES / 2; /* filter waves; as many helper waves */
        /* (4 x lw row rings of 2288 bytes: eight filter waves take 73 KB, which a gfx950 CU has (160 KB) and a 64 KB part has not; the kernel
         * runs taller slices in passes anyway, so fewer rows at a time is only slower -- never a launch that silently does not happen) */
        while (lw > 1 && (size_t)4 * lw * sizeof(dl_row) > c->lds_per_block) lw--;
        const dim3 grid(S.slice1 - S.slice0, 1, n_items), block(2 * 64 * lw);
        const size_t lds = (size_t)4 * lw * sizeof(dl_row);
        static const int diag = getenv("VENC_DL_DIAG") ? atoi(getenv("VENC_DL_DIAG")) : 0; /* (diagnostic: parts of a step left out, pictures then wrong) */
        if (planes == 1) hipLaunchKernelGGL(k_deblock_ls<1>, grid, block, lds, st, g, b, S, diag);
        else if (planes == 2) hipLaunchKernelGGL(k_deblock_ls<2>, grid, block, lds, st, g, b, S, diag);
        else hipLaunchKernelGGL(k_deblock_ls<3>, grid, block, lds, st, g, b, S, diag);
    }
    prof_end(c, st);
}
/* the chroma transform of the last P step runs on the second stream: anything on the main stream that touches chroma
 * reconstructions (intra pictures, band export / import) waits for it */
/* two chains: the main stream waits for what the second chain still has in flight */
static hipError_t main_joins_b(venc_ctx *c) {
    if (!c->b_active) return hipSuccess;
    c->b_active = 0;
    hipError_t e = hipEventRecord(c->ev_b, c->stream_b);
    return e != hipSuccess ? e : hipStreamWaitEvent(c->stream, c->ev_b, 0);
}
static hipError_t main_waits_for_chroma(venc_ctx *c) {
    if (!c->chroma_pending) return hipSuccess;
    c->chroma_pending = 0;
    return hipStreamWaitEvent(c->stream, c->ev_chroma, 0);
}

extern "C" int venc_abi_version(void) { return VENC_ABI_VERSION; }
extern "C" const char *venc_last_error(void) { return g_err; }
/* free and total memory of a device (the host shim sizes its frame pools by it) */
extern "C" int venc_device_memory(int device, size_t *free_bytes, size_t *total_bytes) {
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return -1;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}
extern "C" int venc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


/* ---- host threads beside their GPU.  A two-socket host reaches the card through ONE socket; a thread that launches kernels, rings
 * doorbells and polls pinned memory from the other socket pays the inter-socket hop on every one of those (measured on the round-4
 * boxes, 2 x EPYC 9575F, same box, medians of four runs: 1080p 171.5 Gpx/s unbound (162 - 173: a run keeps the socket it started
 * on), 179.2 bound to the card's node, 173.0 bound to the other one).  venc_numa_bind(device, scope): the card's NUMA node from
 * sysfs (PCI bus id -> numa_node -> cpulist), intersected with the CPUs the caller may use; scope 0 = the calling thread (threads it
 * creates later inherit), 1 = every thread of the process (the HIP runtime's included).  Returns the node, or -1 when there is
 * nothing to do (one node, no sysfs, an empty intersection).  venc_open calls it with scope 0 unless VENC_NUMA_BIND=0
 * (VENC_NUMA_BIND=2: scope 1). */
static int read_small_file(const char *path, char *buf, size_t cap) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return (int)n;
}
extern "C" int venc_numa_bind(int device, int scope) {
    char bdf[64] = {0}, path[160], buf[4096];
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return -1;
    for (char *p = bdf; *p; p++)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a'); /* sysfs spells bus ids in lower case */
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    if (read_small_file(path, buf, sizeof(buf)) <= 0) return -1;
    const int node = atoi(buf);
    if (node < 0) return -1;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    if (read_small_file(path, buf, sizeof(buf)) <= 0) return -1;
    cpu_set_t want, have;
    CPU_ZERO(&want);
    for (char *p = buf; *p;) { /* "0-63,128-191" */
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long k = a; k <= b && k < CPU_SETSIZE; k++) CPU_SET((int)k, &want);
        p = *e == ',' ? e + 1 : e;
        if (*e != ',' ) break;
    }
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return -1;
    int n_have = 0, n_both = 0;
    for (int k = 0; k < CPU_SETSIZE; k++) {
        if (CPU_ISSET(k, &have)) n_have++;
        if (CPU_ISSET(k, &have) && CPU_ISSET(k, &want)) n_both++; else CPU_CLR(k, &want);
    }
    if (n_both == 0 || n_both == n_have) return n_both ? node : -1; /* nothing allowed on that node / already there */
    if (scope == 0) return sched_setaffinity(0, sizeof(want), &want) == 0 ? node : -1;
    DIR *d = opendir("/proc/self/task");
    if (!d) return sched_setaffinity(0, sizeof(want), &want) == 0 ? node : -1;
    int ok = 0;
    for (struct dirent *e; (e = readdir(d)) != NULL;) {
        const int tid = atoi(e->d_name);
        if (tid > 0 && sched_setaffinity(tid, sizeof(want), &want) == 0) ok++;
    }
    closedir(d);
    return ok ? node : -1;
}

static void rccl_release(venc_ctx *c);

extern "C" void venc_close(venc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream && c->stream2 && c->b.mbbits) (void)entropy_flush(c);
    if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
    if (c->stream_x) (void)hipStreamSynchronize(c->stream_x);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rccl_release(c);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream_copy2) (void)hipStreamSynchronize(c->stream_copy2);
    if (c->stream_copy) (void)hipStreamSynchronize(c->stream_copy);
    if (c->ev)
        for (int i = 0; i < c->n_events; i++)
            if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    free(c->ev);
    if (c->ev_copy)
        for (int i = 0; i < c->cfg.in_slots; i++)
            if (c->ev_copy[i]) (void)hipEventDestroy(c->ev_copy[i]);
    free(c->ev_copy);
    free(c->copy_pending);
    if (c->stream_copy) (void)hipStreamDestroy(c->stream_copy);
    if (c->stream_copy2) (void)hipStreamDestroy(c->stream_copy2);
    for (int i = 0; i < 2; i++) {
        if (c->ev_main[i]) (void)hipEventDestroy(c->ev_main[i]);
        if (c->ev_luma[i]) (void)hipEventDestroy(c->ev_luma[i]);
    }
    if (c->ev_chroma) (void)hipEventDestroy(c->ev_chroma);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->ev_intra) (void)hipEventDestroy(c->ev_intra);
    if (c->ev_x_main) (void)hipEventDestroy(c->ev_x_main);
    if (c->ev_x_done) (void)hipEventDestroy(c->ev_x_done);
    if (c->stream_x) (void)hipStreamDestroy(c->stream_x);
    for (int i = 0; i < VENC_BHIST; i++) {
        if (c->bh[i].ev_main) (void)hipEventDestroy(c->bh[i].ev_main);
        if (c->bh[i].ev_chroma) (void)hipEventDestroy(c->bh[i].ev_chroma);
        if (c->bh[i].ev_main_b) (void)hipEventDestroy(c->bh[i].ev_main_b);
    }
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    for (int i = 0; i < VENC_RING; i++)
        if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    (void)hipFree(c->d_in);
    (void)hipFree(c->b.recon);
    (void)hipFree(c->b.mvsad);
    (void)hipFree(c->b.levels);
    (void)hipFree(c->b.mbinfo);
    (void)hipFree(c->b.mbbits);
    (void)hipFree(c->b.mbnbits);
    (void)hipFree(c->b.mboff);
    (void)hipFree(c->b.codedmask);
    (void)hipFree(c->b.ent_list);
    (void)hipFree(c->b.ent_cls);
    (void)hipFree(c->b.ent_cls2);
    (void)hipFree(c->b.ent_mb0);
    (void)hipFree(c->b.ent_count);
    (void)hipFree(c->b.ent_head);
    (void)hipFree(c->b.ent_off);
    (void)hipFree(c->b.seg_bits);
    (void)hipFree(c->b.slice_bits);
    (void)hipFree(c->b.scan_done);
    (void)hipFree(c->b.refsum);
    (void)hipFree(c->b.me_flag);
    (void)hipFree(c->b.me_cnt);
    (void)hipFree(c->b.hpel);
    (void)hipFree(c->b.mvq);
    (void)hipFree(c->b.stamps);
    (void)hipFree(c->b.fix_count);
    (void)hipFree(c->d_sad_sum);
    (void)hipFree(c->d_qp_ring);
    (void)hipFree(c->d_qtab);
    (void)hipFree(c->d_fix_done);
    (void)hipFree(c->b.intra_rec);
    (void)hipFree(c->b.intra_d0);
    (void)hipFree(c->b.fix_list);
    (void)hipFree(c->b.nz_ids);
    (void)hipFree(c->b.nz_cnt);
    (void)hipFree(c->b.nz_rows);
    (void)hipFree(c->d_slice);
    (void)hipFree(c->d_slice_rows);
    (void)hipFree(c->d_row_slice);
    (void)hipFree(c->d_rank);
    (void)hipFree(c->d_cand);
    if (c->h_hdr) (void)hipHostFree(c->h_hdr);
    if (c->h_data) (void)hipHostFree(c->h_data);
    (void)hipFree(c->d_data);
    free(c->fetched);
    free(c->host_of);
    free(c->host_owner);
    if (c->stream_d2h) (void)hipStreamDestroy(c->stream_d2h);
    if (c->prof_ev)
        for (int i = 0; i < 2 * c->prof_cap; i++)
            if (c->prof_ev[i]) (void)hipEventDestroy(c->prof_ev[i]);
    free(c->prof_ev);
    free(c->prof_kid);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    free(c);
}

extern "C" int venc_open(const venc_config *cfg, venc_ctx **out) {
    if (!cfg || !out) {
        SET_ERR("venc_open: null argument");
        return -1;
    }
    *out = NULL;
    if (cfg->width < 16 || cfg->height < 16 || (cfg->width & 1) || (cfg->height & 1) || cfg->width > 16384 ||
        cfg->height > 16384) {
        SET_ERR("venc_open: unsupported size %dx%d", cfg->width, cfg->height);
        return -1;
    }
    if (cfg->qp < 12 || cfg->qp > 51) {
        SET_ERR("venc_open: qp %d outside 12..51", cfg->qp);
        return -1;
    }
    if (cfg->max_items < 1 || cfg->max_items > VENC_MAX_ITEMS || cfg->out_slots < 1 || cfg->in_slots < 0) {
        SET_ERR("venc_open: bad pool configuration");
        return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        SET_ERR("venc_open: no HIP device available (the HIP path has no CPU fallback)");
        return -2;
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        SET_ERR("venc_open: device %d out of range (%d devices)", cfg->device, ndev);
        return -1;
    }
    HIP_OK(hipSetDevice(cfg->device));
    {   /* the calling thread -- which launches every kernel of this context -- and the threads created from here on: beside the card */
        const int nb = getenv("VENC_NUMA_BIND") ? atoi(getenv("VENC_NUMA_BIND")) : 1;
        if (nb > 0) (void)venc_numa_bind(cfg->device, nb >= 2 ? 1 : 0);
    }
    venc_ctx *c = (venc_ctx *)calloc(1, sizeof(venc_ctx));
    if (!c) return -1;
    c->cfg = *cfg;
    venc_geom &g = c->g;
    g.w = cfg->width;
    g.h = cfg->height;
    g.mbw = (g.w + 15) / 16;
    g.mbh = (g.h + 15) / 16;
    g.cw = g.mbw * 16;
    g.ch = g.mbh * 16;
    g.nmb = g.mbw * g.mbh;
    g.mbw_magic = g.mbw > 1 ? (uint32_t)((0x100000000ull + (uint64_t)g.mbw - 1) / (uint64_t)g.mbw) : 0u;
    g.n_slices = cfg->n_slices < 1 ? 1 : cfg->n_slices;
    if (g.n_slices > g.mbh) g.n_slices = g.mbh;
    if (g.n_slices > VENC_MAX_SLICES) g.n_slices = VENC_MAX_SLICES;
    g.qp = cfg->qp;
    fill_qparams(&g.ql, cfg->qp);
    fill_qparams(&g.qc, cfg->qp < 30 ? cfg->qp : h_qpc_hi[cfg->qp - 30]);
    c->cfg.n_slices = g.n_slices;
    g.subpel = cfg->subpel < 0 ? 0 : cfg->subpel > 2 ? 2 : cfg->subpel;
    g.intra_plane = cfg->intra_modes >= 2;
    c->g_idr = g;
    g.slice_base = g.mbh / g.n_slices;
    g.slice_rem = g.mbh % g.n_slices;

#define OPEN_OK(call)               \
    do {                            \
        hipError_t e_ = (call);     \
        if (e_ != hipSuccess) {     \
            SET_ERR("%s failed: %s", #call, hipGetErrorString(e_)); \
            venc_close(c);          \
            return -1;              \
        }                           \
    } while (0)

    /* VENC_PRIO=1 (experiment): the main chain's stream at the highest priority, the side chain's at the lowest */
    static const int prio = getenv("VENC_PRIO") ? atoi(getenv("VENC_PRIO")) : 0;
    int prio_lo = 0, prio_hi = 0;
    OPEN_OK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    if (prio)
        OPEN_OK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi));
    else
        OPEN_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        /* VENC_ENTROPY_CUS=n: the entropy chain's stream may only use n of the chip's CUs (queue CU mask; the mask bits
         * interleave the XCDs, so the first n bits are n/8 CUs of every XCD).  The chain is latency bound and its
         * blocks otherwise land on every CU, where their LDS and registers cost the motion search its third block. */
        /* Default (round 4, same-box medians of three rounds, gpurun_out/r04_cus2*): 192 of the 256 CUs from 60 000 macroblocks per
         * step (1080p x 10 items: 166.5 Gpx/s unmasked, 163.6 on 128, 175.6 on 192, 176.9 on 224), 160 from 200 000 (2160p x 10: 208.5
         * unmasked, 220.0 on 128, 219.6 on 160, 217.4 on 192).  In round 3, with a side chain a quarter heavier, a mask LOST at 1080p;
         * the chain is light enough now to live on three quarters of the chip, and the search keeps a quarter to itself.  Smaller
         * steps (streaming engines) stay unmasked.  VENC_ENTROPY_CUS=0 / =n overrides. */
        const char *e = getenv("VENC_ENTROPY_CUS");
        const long long step_mbs = (long long)g.nmb * cfg->max_items;
        /* Third session of round 4 (gpurun_out/r04_s3g ... r04_s3i; the search 5 % shorter, the host's packaging no longer in the way): 1080p x 10
         * items 185.9 on 224, 187.5 on 192, 191.0 on 160; x 24 items 193.4 on 96, 214.6 on 128, 213.1 on 160, 209.4 on 192; 2160p x 12 items
         * 238.6 on 96, 249.5 on 128, 248.4 on 160: 160 CUs from 60 000 macroblocks per step, 128 from 150 000. */
        const int ncu = e ? atoi(e) : step_mbs >= 150000 ? 128 : step_mbs >= 60000 ? 160 : 0;
        hipDeviceProp_t prop;
        OPEN_OK(hipGetDeviceProperties(&prop, cfg->device));
        c->lds_per_block = prop.sharedMemPerBlock;
        if (ncu >= 8 && ncu < prop.multiProcessorCount) {
            uint32_t mask[16] = {0};
            for (int i = 0; i < ncu && i < 512; i++) mask[i >> 5] |= 1u << (i & 31);
            OPEN_OK(hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)((prop.multiProcessorCount + 31) / 32), mask));
        } else if (prio)
            OPEN_OK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_lo));
        else
            OPEN_OK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    }
    OPEN_OK(hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
    OPEN_OK(hipStreamCreateWithFlags(&c->stream_copy2, hipStreamNonBlocking));
    c->n_events = cfg->out_slots + 2 > VENC_EVENTS_MIN ? cfg->out_slots + 2 : VENC_EVENTS_MIN;
    c->ev = (hipEvent_t *)calloc((size_t)c->n_events, sizeof(hipEvent_t));
    c->ev_copy = (hipEvent_t *)calloc((size_t)(cfg->in_slots > 0 ? cfg->in_slots : 1), sizeof(hipEvent_t));
    c->copy_pending = (uint8_t *)calloc((size_t)(cfg->in_slots > 0 ? cfg->in_slots : 1), 1);
    c->last_copy_slot = -1;
    c->prev_copy_slot = -1;
    c->n_copy_streams = 1;
    if (!c->ev || !c->ev_copy || !c->copy_pending) {
        SET_ERR("venc_open: out of host memory");
        venc_close(c);
        return -1;
    }
    for (int i = 0; i < c->n_events; i++) OPEN_OK(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
    for (int i = 0; i < cfg->in_slots; i++) OPEN_OK(hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        OPEN_OK(hipEventCreateWithFlags(&c->ev_main[i], hipEventDisableTiming));
        OPEN_OK(hipEventCreateWithFlags(&c->ev_luma[i], hipEventDisableTiming));
    }
    OPEN_OK(hipEventCreateWithFlags(&c->ev_chroma, hipEventDisableTiming));
    OPEN_OK(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    OPEN_OK(hipEventCreateWithFlags(&c->ev_intra, hipEventDisableTiming));
    OPEN_OK(hipStreamCreateWithFlags(&c->stream_x, hipStreamNonBlocking));
    OPEN_OK(hipEventCreateWithFlags(&c->ev_x_main, hipEventDisableTiming));
    OPEN_OK(hipEventCreateWithFlags(&c->ev_x_done, hipEventDisableTiming));
    c->x_overlap = getenv("VENC_XOVERLAP") ? atoi(getenv("VENC_XOVERLAP")) != 0 : 1;
    for (int i = 0; i < VENC_BHIST; i++) {
        OPEN_OK(hipEventCreateWithFlags(&c->bh[i].ev_main, hipEventDisableTiming));
        OPEN_OK(hipEventCreateWithFlags(&c->bh[i].ev_chroma, hipEventDisableTiming));
        OPEN_OK(hipEventCreateWithFlags(&c->bh[i].ev_main_b, hipEventDisableTiming));
        c->bh[i].last_step = -1;
    }
    OPEN_OK(hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
    OPEN_OK(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
    OPEN_OK(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
    c->b_behind = 1;
    c->two_chain = getenv("VENC_CHAINS") ? atoi(getenv("VENC_CHAINS")) == 2 : 0;
    c->intra_ahead = !(getenv("VENC_INTRA_AHEAD") && atoi(getenv("VENC_INTRA_AHEAD")) == 0);
    c->rec_last = VENC_RECON - 1;
    /* 0: k_inter_fused (luma + chroma) on the main chain; 1 (default): luma transform inside the search kernels, chroma on stream 2;
     * 2: chroma inside the search kernels too (waves 4-5 of P5): bit-exact, measured SLOWER -- the chroma samples come from memory
     * and their latency lengthens every block's tail while it holds a third of a CU's LDS (1080p 149 -> 141 Gpx/s, search alone
     * 85 -> 104 us; 2160p 196 -> 179) */
    c->fuse_luma = getenv("VENC_FUSE") ? atoi(getenv("VENC_FUSE")) : 3;
    if (c->fuse_luma < 0 || c->fuse_luma > 3 || c->fuse_luma == 2) c->fuse_luma = 3; /* (2, the chroma transform inside the search kernels too, was measured slower in round 2 and left the sources in round 3) */
    for (int i = 0; i < VENC_RING; i++) {
        OPEN_OK(hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming));
        c->side_step[i] = -1;
    }
    c->overlap = !(getenv("VENC_NO_OVERLAP") && atoi(getenv("VENC_NO_OVERLAP")) != 0);
    c->ev_done = -1;

    /* slice bands: the first (mbh % n) bands get one extra MB row */
    int rows[VENC_MAX_SLICES + 1];
    uint8_t *row_slice = (uint8_t *)malloc((size_t)g.mbh);
    int maxrows = 0;
    {
        int base = g.mbh / g.n_slices, rem = g.mbh % g.n_slices, r = 0;
        for (int s = 0; s < g.n_slices; s++) {
            rows[s] = r;
            int n = base + (s < rem ? 1 : 0);
            for (int k = 0; k < n; k++) row_slice[r + k] = (uint8_t)s;
            if (n > maxrows) maxrows = n;
            r += n;
        }
        rows[g.n_slices] = r;
    }
    memcpy(c->slice_rows, rows, sizeof(int) * (g.n_slices + 1));
    c->band_s0 = 0;
    c->band_s1 = g.n_slices;
    c->intra_cw = (maxrows + 1) / 2 < INTRA_CW_MAX ? (maxrows + 1) / 2 : INTRA_CW_MAX; /* compute waves of k_intra_dc */
    {
        const char *e = getenv("VENC_INTRA");
        c->intra_legacy = e && !strcmp(e, "legacy");
    }
#ifndef VENC_LEGACY_KERNELS
    {
        /* the first-generation kernels (VENC_ME=tile|plane|dense, VENC_INTER=legacy, VENC_INTRA=legacy) are on-GPU test
         * oracles: they live in libvenc_hip_legacy.so (tests/conftest.py loads it), not in the product library */
        const char *m = getenv("VENC_ME"), *i1 = getenv("VENC_INTER"), *i2 = getenv("VENC_INTRA");
        if ((m && strcmp(m, "pool")) || (getenv("VENC_ME_DENSE") && atoi(getenv("VENC_ME_DENSE")) != 0) || (i1 && !strcmp(i1, "legacy")) ||
            (i2 && !strcmp(i2, "legacy"))) {
            SET_ERR("venc_open: the first-generation kernels selected by VENC_ME / VENC_INTER / VENC_INTRA are built into "
                    "libvenc_hip_legacy.so (test-only), not into this library");
            venc_close(c);
            return -1;
        }
    }
#endif
    c->intra_waves = (maxrows + 1) / 2 < INTRA_MAX_WAVES ? (maxrows + 1) / 2 : INTRA_MAX_WAVES; /* two MB rows per wave */
    c->max_slice_mbs = maxrows * g.mbw;
    OPEN_OK(hipMalloc(&c->d_slice_rows, sizeof(int) * (g.n_slices + 1)));
    OPEN_OK(hipMemcpy(c->d_slice_rows, rows, sizeof(int) * (g.n_slices + 1), hipMemcpyHostToDevice));
    OPEN_OK(hipMalloc(&c->d_row_slice, (size_t)g.mbh));
    OPEN_OK(hipMemcpy(c->d_row_slice, row_slice, (size_t)g.mbh, hipMemcpyHostToDevice));
    free(row_slice);

    /* ME tie-break order: all 33x33 displacements sorted by (|dx|+|dy|, dy, dx) */
    {
        uint16_t rank[VENC_ME_CANDS + 1], cand[VENC_ME_CANDS]; /* rank padded to whole dwords (kernels stage it as 545 dwords) */
        rank[VENC_ME_CANDS] = 0;
        int r = 0;
        for (int l1 = 0; l1 <= 2 * VENC_ME_RANGE; l1++)
            for (int dy = -VENC_ME_RANGE; dy <= VENC_ME_RANGE; dy++)
                for (int dx = -VENC_ME_RANGE; dx <= VENC_ME_RANGE; dx++)
                    if (abs(dx) + abs(dy) == l1) {
                        rank[(dy + 16) * 33 + dx + 16] = (uint16_t)r;
                        cand[r] = (uint16_t)(((dy + 16) << 8) | (dx + 16));
                        r++;
                    }
        OPEN_OK(hipMalloc(&c->d_rank, sizeof(rank)));
        OPEN_OK(hipMemcpy(c->d_rank, rank, sizeof(rank), hipMemcpyHostToDevice));
        OPEN_OK(hipMalloc(&c->d_cand, sizeof(cand)));
        OPEN_OK(hipMemcpy(c->d_cand, cand, sizeof(cand), hipMemcpyHostToDevice));
    }

    venc_bufs &b = c->b;
    const size_t G = (size_t)cfg->max_items, nmb = (size_t)g.nmb;
    b.max_items = cfg->max_items;
    {
        /* steps per entropy batch: a power of two that divides the ring, (step, item) pairs within VENC_MAX_ITEMS */
        const char *e = getenv("VENC_ENTROPY_BATCH");
        int want = e ? atoi(e) : VENC_EBATCH;
        if (want < 1) want = 1;
        if (want > VENC_EBATCH) want = VENC_EBATCH;
        c->eb_max = 1;
        while (c->eb_max * 2 <= want && c->eb_max * 2 * cfg->max_items <= VENC_MAX_ITEMS) c->eb_max *= 2;
        b.vcap = c->eb_max * cfg->max_items;
    }
    const size_t GV = (size_t)b.vcap; /* the entropy chain's own buffers hold a whole batch */
    b.in_frame_stride = (size_t)g.w * g.h * 3 / 2;
    /* the internal input pool is allocated on first use (venc_upload / venc_input_slot_ptr):
     * resident-input users (venc_set_input) never pay for it */
    b.recon_frame_bytes = (size_t)g.cw * g.ch * 3 / 2;
    OPEN_OK(hipMalloc(&b.recon, VENC_RECON * G * b.recon_frame_bytes));
    OPEN_OK(hipMemset(b.recon, 128, VENC_RECON * G * b.recon_frame_bytes));
    /* mvsad, levels and mbinfo: a ring of VENC_RING step slots (see VENC_RING) */
    OPEN_OK(hipMalloc(&b.mvsad, VENC_RING * G * nmb * 4));
    OPEN_OK(hipMemset(b.mvsad, 0, VENC_RING * G * nmb * 4));
    OPEN_OK(hipMalloc(&b.levels, VENC_RING * G * nmb * VENC_MB_COEFS * 2));
    OPEN_OK(hipMalloc(&b.mbinfo, VENC_RING * G * nmb * sizeof(venc_mbinfo)));
    OPEN_OK(hipMemset(b.mbinfo, 0, VENC_RING * G * nmb * sizeof(venc_mbinfo)));
    OPEN_OK(hipMalloc(&b.mbbits, GV * nmb * VENC_MB_WORDS * 4));
    OPEN_OK(hipMalloc(&b.mbnbits, GV * nmb * 4));
    OPEN_OK(hipMalloc(&b.mboff, GV * nmb * 4));
    {
        /* P pictures: segments of 256 macroblocks of a slice (venc_cavlc_p.hip) */
        const int slice_mbs = maxrows * g.mbw;
        b.ent_spp = (slice_mbs + ENT_SEG_MBS - 1) / ENT_SEG_MBS;
        b.ent_wps = 4 * b.ent_spp;
        b.ent_cap = (uint32_t)(ENT_PER_MB * (slice_mbs < ENT_SEG_MBS ? slice_mbs : ENT_SEG_MBS));
        const size_t nseg = GV * g.n_slices * b.ent_spp;
        OPEN_OK(hipMalloc(&b.codedmask, GV * g.n_slices * b.ent_wps * 8));
        OPEN_OK(hipMalloc(&b.ent_list, nseg * b.ent_cap * 4));
        OPEN_OK(hipMalloc(&b.ent_cls, nseg * b.ent_cap * 4));
        OPEN_OK(hipMalloc(&b.ent_cls2, nseg * b.ent_cap * 4));
        OPEN_OK(hipMalloc(&b.ent_mb0, GV * nmb * 4));
        OPEN_OK(hipMalloc(&b.ent_count, nseg * 16));
        OPEN_OK(hipMemset(b.ent_count, 0, nseg * 16));
        OPEN_OK(hipMalloc(&b.ent_head, nseg * b.ent_cap * sizeof(uint4)));
        OPEN_OK(hipMalloc(&b.ent_off, nseg * (b.ent_cap + 4) * 4));
        OPEN_OK(hipMalloc(&b.seg_bits, nseg * 4));
        OPEN_OK(hipMemset(b.seg_bits, 0, nseg * 4));
    }
    OPEN_OK(hipMalloc(&b.slice_bits, GV * VENC_MAX_SLICES * 4));
    OPEN_OK(hipMalloc(&b.scan_done, GV * VENC_COUNTER_STRIDE * 4));
    OPEN_OK(hipMemset(b.scan_done, 0, GV * VENC_COUNTER_STRIDE * 4));
    b.refsum_stride = (g.cw + 17 + 15) & ~15;
    b.refsum_item = (size_t)b.refsum_stride * (g.ch + 25);
#ifdef VENC_LEGACY_KERNELS
    OPEN_OK(hipMalloc(&b.refsum, G * b.refsum_item * 4));
    OPEN_OK(hipMemset(b.refsum, 0, G * b.refsum_item * 4));
#endif
    {
        const char *m = getenv("VENC_ME");
        c->me_mode = 0;
        if (m && !strcmp(m, "plane")) c->me_mode = 1;
        if (m && !strcmp(m, "dense")) c->me_mode = 2;
        if (m && !strcmp(m, "tile")) c->me_mode = 3;
        if (getenv("VENC_ME_DENSE") && atoi(getenv("VENC_ME_DENSE")) != 0) c->me_mode = 2;
    }
    c->me_stop = getenv("VENC_ME_STOP") ? atoi(getenv("VENC_ME_STOP")) : 0;
    c->me_fast = getenv("VENC_ME_FAST") ? atoi(getenv("VENC_ME_FAST")) : 1;
    /* XCD-aware tile order of the search (runs of 64 tiles dealt to the eight XCDs in rotation: a tile's neighbours share an L2).  Default from
     * 200 000 macroblocks per step (2160p x 10 items: 223.4 - 225.6 -> 228.3 - 231.6 Gpx/s in three same-box A/Bs of round 4; at 1080p x 10 the
     * launch gets 5 - 8 % longer and the pass does not change); whole pictures only (a band's search may go as two launches) */
    /* (by the PICTURE's size as well as the step's: 1080p x 20 items 207.6 Gpx/s in raster order, 195.6 in XCD order; 2160p x 10 items 240.5 / 244.3) */
    c->me_xcd = getenv("VENC_ME_XCD") ? atoi(getenv("VENC_ME_XCD")) != 0 : g.nmb >= 20000 && (long long)g.nmb * cfg->max_items >= 200000;
    c->inter_legacy = getenv("VENC_INTER") && !strcmp(getenv("VENC_INTER"), "legacy");
    if (cfg->deblock && (c->me_mode != 0 || c->inter_legacy || (getenv("VENC_INTRA") && !strcmp(getenv("VENC_INTRA"), "legacy")))) {
        SET_ERR("venc_open: the in-loop filter runs behind the shipped kernels only (VENC_ME / VENC_INTER / VENC_INTRA select test kernels)");
        venc_close(c);
        return -1;
    }
    if (cfg->deblock && g.slice_base + 1 > DBK_MAX_ROWS) { /* (the rows' progress flags live in the block's LDS) */
        SET_ERR("venc_open: the in-loop filter handles slices of up to %d macroblock rows (this picture: %d per slice; use more slices)", DBK_MAX_ROWS, g.slice_base + 1);
        venc_close(c);
        return -1;
    }
    if (g.subpel && (c->me_mode != 0 || c->inter_legacy)) {
        SET_ERR("venc_open: sub-sample refinement runs behind the shipped search and transform only (VENC_ME / VENC_INTER select test kernels)");
        venc_close(c);
        return -1;
    }
    OPEN_OK(hipMalloc(&b.me_flag, G * nmb));
    OPEN_OK(hipMemset(b.me_flag, 0, G * nmb));
    if (cfg->rc_mode) {
        if (c->me_mode != 0 || c->inter_legacy || c->intra_legacy) {
            SET_ERR("venc_open: rate control needs the shipped kernels (not the legacy test variants)");
            venc_close(c);
            return -1;
        }
        int qi = cfg->qp - 3;
        qi = qi < 12 ? 12 : qi;
        c->g_idr = g;
        c->g_idr.qp = qi;
        fill_qparams(&c->g_idr.ql, qi);
        fill_qparams(&c->g_idr.qc, qi < 30 ? qi : h_qpc_hi[qi - 30]);
        OPEN_OK(hipMalloc(&c->d_sad_sum, 3 * G * sizeof(unsigned long long)));
        OPEN_OK(hipMemset(c->d_sad_sum, 0, 3 * G * sizeof(unsigned long long)));
        OPEN_OK(hipMalloc(&c->d_qp_ring, VENC_QP_RING * G * sizeof(int)));
        OPEN_OK(hipMemset(c->d_qp_ring, 0, VENC_QP_RING * G * sizeof(int)));
        OPEN_OK(hipMalloc(&c->d_fix_done, G * sizeof(uint32_t)));
        OPEN_OK(hipMemset(c->d_fix_done, 0, G * sizeof(uint32_t)));
        venc_qparams qt[52];
        memset(qt, 0, sizeof(qt));
        for (int q = 0; q < 52; q++) fill_qparams(&qt[q], q);
        OPEN_OK(hipMalloc(&c->d_qtab, sizeof(qt)));
        OPEN_OK(hipMemcpy(c->d_qtab, qt, sizeof(qt), hipMemcpyHostToDevice));
        b.qtab = c->d_qtab;
        b.fix_done = c->d_fix_done;
    }
    OPEN_OK(hipMalloc(&b.fix_count, 2 * G * 4));
    OPEN_OK(hipMalloc(&b.intra_rec, G * nmb * INTRA_REC_DWORDS * 4));
    OPEN_OK(hipMalloc(&b.intra_d0, G * nmb * 32 * 4));
    OPEN_OK(hipMemset(b.fix_count, 0, 2 * G * 4));
    OPEN_OK(hipMalloc(&b.fix_list, G * nmb * 4));
    /* the segments of 4x4 luma blocks left to the k_me_fix launch: four transforming waves per 8 x 2-macroblock tile of the search */
    b.nz_item_regions = 4 * ((g.mbw + T2_MBX - 1) / T2_MBX) * ((g.mbh + T2_MBY - 1) / T2_MBY);
    OPEN_OK(hipMalloc(&b.nz_ids, (size_t)G * b.nz_item_regions * 64 * 4));
    OPEN_OK(hipMalloc(&b.nz_rows, (size_t)G * b.nz_item_regions * 64 * 32));
    OPEN_OK(hipMalloc(&b.nz_cnt, (size_t)G * b.nz_item_regions * 4));
    OPEN_OK(hipMemset(b.nz_cnt, 0, (size_t)G * b.nz_item_regions * 4));
    OPEN_OK(hipMalloc(&b.stamps, VENC_STAMP_WORDS * sizeof(unsigned long long)));
    OPEN_OK(hipMemset(b.stamps, 0, VENC_STAMP_WORDS * sizeof(unsigned long long)));
    if (g.subpel) {
        OPEN_OK(hipMalloc(&b.hpel, G * 3 * (size_t)g.cw * g.ch + 64)); /* (+64: k_subpel's aligned row pieces end up to 6 bytes past a row) */
        OPEN_OK(hipMalloc(&b.mvq, (size_t)VENC_RING * G * nmb * 4));
        OPEN_OK(hipMemset(b.mvq, 0, (size_t)VENC_RING * G * nmb * 4));
    }
    OPEN_OK(hipMalloc(&b.me_cnt, G * nmb));
    OPEN_OK(hipMemset(b.me_cnt, 0, G * nmb));
    OPEN_OK(hipMalloc(&c->d_slice, GV * g.n_slices * sizeof(venc_slice_out)));
    b.slice_first_row = c->d_slice_rows;
    b.row_slice = c->d_row_slice;
    b.me_rank = c->d_rank;
    b.me_cand = c->d_cand;

    /* output pool: headers + slice tables in pinned, device-mapped memory; data in HBM (staged) or pinned host memory */
    b.out_hdr_stride = (sizeof(venc_out_header) + (size_t)g.n_slices * sizeof(venc_slice_out) + 63) & ~(size_t)63;
    /* Capacity per frame.  H.264 Annex A.3.1 caps macroblock_layer() at 3200 bits = 400 bytes per macroblock for the
     * Baseline family; a macroblock above that is not a conforming stream whatever the buffer holds (libx264 re-codes it as
     * I_PCM, a mode this subset does not have).  QP < 40 gets exactly that bound = 1.56 bytes per pixel.  Uniform white
     * noise, the densest content there is, measures with the oracle (352x288, bytes per pixel, largest macroblock in bits):
     * QP 12: 1.89 / 4179, 16: 1.65 / 3615, 20: 1.40 / 3086, 26: 1.09 / 2389, 30: 0.93 / 2008, 40: 0.58 / 1296, 51: 0.23 / 545
     * -- so from QP ~19 up every content fits, below it noise leaves the conformance limit and the slot with it.  QP >= 40:
     * 256 bytes per macroblock (twice what noise needs there).  + slice headers and the per-slice 4-byte alignment.
     * Overflow is reported through header.error -> venc_get_frame() = -3, never silent. */
    b.out_cap = (uint32_t)((size_t)g.nmb * (cfg->qp < 40 ? 400 : 256) + (size_t)g.n_slices * 64 + 4096);
    b.out_data_stride = ((size_t)b.out_cap + 255) & ~(size_t)255;
    {
        const char *e = getenv("VENC_STAGED");
        c->staged = e ? atoi(e) != 0 : cfg->max_items > 1;
    }
    OPEN_OK(hipHostMalloc((void **)&c->h_hdr, b.out_hdr_stride * cfg->out_slots, hipHostMallocMapped));
    memset(c->h_hdr, 0, b.out_hdr_stride * cfg->out_slots);
    c->host_slots = cfg->out_slots;
    if (c->staged) {
        const int want = 4 * cfg->max_items > 8 ? 4 * cfg->max_items : 8;
        c->host_slots = cfg->out_slots < want ? cfg->out_slots : want;
    }
    OPEN_OK(hipHostMalloc((void **)&c->h_data, b.out_data_stride * c->host_slots, c->staged ? hipHostMallocDefault : hipHostMallocMapped));
    void *dview = NULL;
    OPEN_OK(hipHostGetDevicePointer(&dview, c->h_hdr, 0));
    b.out_hdr = (uint8_t *)dview;
    if (c->staged) {
        OPEN_OK(hipMalloc(&c->d_data, b.out_data_stride * cfg->out_slots));
        b.out_data = c->d_data;
        c->fetched = (uint8_t *)calloc((size_t)cfg->out_slots, 1);
        c->host_of = (int *)malloc(sizeof(int) * (size_t)cfg->out_slots);
        c->host_owner = (int *)malloc(sizeof(int) * (size_t)c->host_slots);
        if (c->host_of) for (int i = 0; i < cfg->out_slots; i++) c->host_of[i] = -1;
        if (c->host_owner) for (int i = 0; i < c->host_slots; i++) c->host_owner[i] = -1;
        if (!c->fetched || !c->host_of || !c->host_owner) {
            SET_ERR("venc_open: out of host memory");
            venc_close(c);
            return -1;
        }
        OPEN_OK(hipStreamCreateWithFlags(&c->stream_d2h, hipStreamNonBlocking));
    } else {
        OPEN_OK(hipHostGetDevicePointer(&dview, c->h_data, 0));
        b.out_data = (uint8_t *)dview;
    }
    OPEN_OK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

extern "C" int venc_mb_width(const venc_ctx *c) { return c->g.mbw; }
extern "C" int venc_mb_height(const venc_ctx *c) { return c->g.mbh; }
extern "C" int venc_n_slices(const venc_ctx *c) { return c->g.n_slices; }
extern "C" size_t venc_frame_bytes(const venc_ctx *c) { return c->b.in_frame_stride; }

extern "C" void *venc_host_alloc(size_t bytes) {
    void *p = NULL;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return NULL;
    return p;
}
extern "C" void venc_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

static int ensure_input_pool(venc_ctx *c) {
    if (c->d_in) return 0;
    if (c->cfg.in_slots < 1) {
        SET_ERR("no internal input pool configured (in_slots = 0)");
        return -1;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    const size_t stride = (size_t)c->g.w * c->g.h * 3 / 2;
    HIP_OK(hipMalloc(&c->d_in, stride * c->cfg.in_slots));
    c->b.in_base = c->d_in;
    c->b.in_frame_stride = stride;
    c->in_frames = c->cfg.in_slots;
    return 0;
}

extern "C" void *venc_input_slot_ptr(venc_ctx *c, int slot) {
    if (slot < 0 || slot >= c->cfg.in_slots || ensure_input_pool(c) != 0) return NULL;
    return c->d_in + (size_t)slot * c->b.in_frame_stride;
}

extern "C" int venc_set_input(venc_ctx *c, const void *d_frames, size_t frame_stride, int n_frames) {
    if (!d_frames || n_frames < 1 || frame_stride < (size_t)c->g.w * c->g.h * 3 / 2) {
        SET_ERR("venc_set_input: bad arguments");
        return -1;
    }
    if ((((uintptr_t)d_frames) | frame_stride) & 15) {
        SET_ERR("venc_set_input: d_frames and frame_stride must be multiples of 16 bytes");
        return -1;
    }
    c->b.in_base = (const uint8_t *)d_frames;
    c->b.in_frame_stride = frame_stride;
    c->in_frames = n_frames;
    return 0;
}

extern "C" int venc_upload(venc_ctx *c, int slot, const uint8_t *y, int sy, const uint8_t *u, int su, const uint8_t *v,
                           int sv) {
    if (slot < 0 || slot >= c->cfg.in_slots) {
        SET_ERR("venc_upload: slot %d not in the internal pool", slot);
        return -1;
    }
    if (ensure_input_pool(c) != 0) return -1;
    if (c->b.in_base != c->d_in) { /* switch back from a borrowed buffer */
        c->b.in_base = c->d_in;
        c->b.in_frame_stride = (size_t)c->g.w * c->g.h * 3 / 2;
        c->in_frames = c->cfg.in_slots;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    const int w = c->g.w, h = c->g.h;
    uint8_t *d = c->d_in + (size_t)slot * c->b.in_frame_stride;
    /* Ingest runs on its own stream: the copy of frame f + 1 overlaps the kernels of frame f (and earlier), and the caller
     * that must get its picture back (venc_upload_sync) waits for the copy alone, not for the kernel chain.  The pool slot is
     * free to overwrite: the caller reuses a slot only after the frame that last used it was delivered (its step complete).
     * Tight planes in one pinned allocation (what x264_picture_alloc hands out) go as ONE copy. */
    /* one stream, or two taken in turn (venc_set_copy_streams: callers that keep two pictures in flight): a lone 3.1 MB copy per
     * call tops out at 43 GB/s whatever waits for it, two in flight on two streams carry 52 (tools/h2d_ubench.hip) */
    hipStream_t sc = c->n_copy_streams > 1 && (slot & 1) ? c->stream_copy2 : c->stream_copy;
    const size_t ysz = (size_t)w * h;
    if (sy == w && su == w / 2 && sv == w / 2 && u == y + ysz && v == u + ysz / 4) {
        HIP_OK(hipMemcpyAsync(d, y, ysz * 3 / 2, hipMemcpyHostToDevice, sc));
    } else {
        HIP_OK(hipMemcpy2DAsync(d, (size_t)w, y, (size_t)sy, (size_t)w, (size_t)h, hipMemcpyHostToDevice, sc));
        HIP_OK(hipMemcpy2DAsync(d + ysz, (size_t)(w / 2), u, (size_t)su, (size_t)(w / 2), (size_t)(h / 2), hipMemcpyHostToDevice, sc));
        HIP_OK(hipMemcpy2DAsync(d + ysz * 5 / 4, (size_t)(w / 2), v, (size_t)sv, (size_t)(w / 2), (size_t)(h / 2), hipMemcpyHostToDevice, sc));
    }
    HIP_OK(hipEventRecord(c->ev_copy[slot], sc));
    c->copy_pending[slot] = 1;
    c->prev_copy_slot = c->last_copy_slot;
    c->last_copy_slot = slot;
    return 0;
}
/* the source planes of every venc_upload so far may be reused (waits for the copies only, never for kernels) */
/* A frame's copy takes tens of microseconds (3.1 MB at 1080p: ~57 us at 55 GB/s) and the caller sits in the call that waits
 * for it: poll the event for a while before handing the thread to the runtime's blocking wait, whose wake-up costs about as
 * much as the copy itself (x264_encoder_encode from pinned host frames: 35 -> see DESIGN.md section 2 GB/s of H2D). */
static int wait_copy_event(hipEvent_t ev) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) {
            SET_ERR("hipEventQuery failed: %s", hipGetErrorString(e));
            return -1;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 2000000ll) break; /* 2 ms: something else holds the copy up */
    }
    HIP_OK(hipEventSynchronize(ev));
    return 0;
}
extern "C" int venc_upload_sync(venc_ctx *c) {
    if (c->last_copy_slot < 0) return 0;
    if (c->n_copy_streams > 1 && c->prev_copy_slot >= 0 && wait_copy_event(c->ev_copy[c->prev_copy_slot]) != 0) return -1; /* the other stream's last */
    return wait_copy_event(c->ev_copy[c->last_copy_slot]); /* per stream the last copy covers the earlier ones */
}
/* copies of consecutive venc_upload calls on two streams in turn (n = 2) or all on one (n = 1, the default) */
extern "C" int venc_set_copy_streams(venc_ctx *c, int n) {
    if (!c || n < 1 || n > 2) return -1;
    c->n_copy_streams = n;
    return 0;
}
extern "C" int venc_upload_sync_slot(venc_ctx *c, int slot) {
    if (slot < 0 || slot >= c->cfg.in_slots || !c->ev_copy) return 0; /* nothing was ever copied into it */
    return wait_copy_event(c->ev_copy[slot]);
}

extern "C" int venc_encode_step(venc_ctx *c, int n_items, const int *in_slots, const int *out_slots, int is_idr,
                                int frame_num, const int *idr_ids) {
    if (n_items < 1 || n_items > c->cfg.max_items || !in_slots || !out_slots) {
        SET_ERR("venc_encode_step: n_items %d outside 1..%d", n_items, c->cfg.max_items);
        return -1;
    }
    if (!c->b.in_base) {
        SET_ERR("venc_encode_step: no input pool (in_slots = 0 and venc_set_input not called)");
        return -1;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    venc_step S;
    memset(&S, 0, sizeof(S));
    S.n_items = n_items;
    S.is_idr = is_idr ? 1 : 0;
    S.frame_num = frame_num;
    S.parity = c->parity;
    S.ref = c->rec_last;
    S.rc_mode = c->cfg.rc_mode ? 1 : 0;
    S.qp_base = c->cfg.qp;
    S.qp_idr = c->g_idr.qp;
    S.rc_history = !is_idr && c->last_step_p;
    S.me_stop = c->me_stop;
    S.deblock = c->cfg.deblock ? 1 : 0;
    S.me_fast = c->me_fast;
    S.slice0 = c->band_s0;
    S.slice1 = c->band_s1;
    S.row0 = c->slice_rows[c->band_s0];
    S.row1 = c->slice_rows[c->band_s1];
    S.mb0 = S.row0 * c->g.mbw;
    S.mb1 = S.row1 * c->g.mbw;
    const int band_rows = S.row1 - S.row0, band_mbs = S.mb1 - S.mb0, band_slices = S.slice1 - S.slice0;
    /* intra steps run ahead on a third stream (VENC_RECON) unless the engine codes a band or the legacy kernel */
    const bool intra_async = is_idr && c->intra_ahead && c->overlap && !c->intra_legacy && c->me_mode == 0 && !c->rc_comm && c->band_s0 == 0 && c->band_s1 == c->g.n_slices;
    /* two chains (see venc_ctx): this P step's items [0, n_a) on the main stream, [n_a, n_items) on stream_b */
    static const int me_blocks_env = getenv("VENC_ME_BLOCKS") ? atoi(getenv("VENC_ME_BLOCKS")) : 4;
    const bool two = c->two_chain && !is_idr && c->overlap && n_items >= 2 && c->me_mode == 0 && !c->inter_legacy && c->fuse_luma == 3 && me_blocks_env != 3 && !c->g.subpel &&
                     !c->cfg.deblock && !c->rc_comm && c->band_s0 == 0 && c->band_s1 == c->g.n_slices && !c->me_xcd;
    const int n_a = two ? (n_items + 1) / 2 : n_items;
    if (c->x_pending && (is_idr || two)) { /* an exchange in flight on stream_x: a P step takes it in at its search (below), any other step here */
        HIP_OK(hipStreamWaitEvent(c->stream, c->ev_x_done, 0));
        c->x_pending = 0;
    }
    if (!two) {
        HIP_OK(main_joins_b(c));
        c->b_behind = 1;
    } else if (c->b_behind || c->last_n_a != n_a) { /* both ways: the main stream has what stream_b did, stream_b waits for what the main stream has enqueued */
        HIP_OK(main_joins_b(c));
        HIP_OK(hipEventRecord(c->ev_a, c->stream));
        HIP_OK(hipStreamWaitEvent(c->stream_b, c->ev_a, 0));
        c->b_behind = 0;
        c->last_n_a = n_a;
    }
    for (int i = 0; i < n_items; i++) {
        if (in_slots[i] < 0 || in_slots[i] >= c->in_frames || out_slots[i] < 0 || out_slots[i] >= c->cfg.out_slots) {
            SET_ERR("venc_encode_step: slot out of range (item %d: in %d, out %d)", i, in_slots[i], out_slots[i]);
            return -1;
        }
        S.in_slot[i] = in_slots[i];
        S.out_slot[i] = out_slots[i];
        if (c->staged) c->fetched[out_slots[i]] = 0;
        if (c->d_in && c->b.in_base == c->d_in && c->copy_pending[in_slots[i]]) { /* the frame's ingest copy precedes the step */
            HIP_OK(hipStreamWaitEvent(intra_async ? c->stream3 : i >= n_a ? c->stream_b : c->stream, c->ev_copy[in_slots[i]], 0));
            c->copy_pending[in_slots[i]] = 0;
        }
        S.idr_id[i] = idr_ids ? idr_ids[i] : 0;
    }
    const venc_geom &g = c->g;
    const int p = c->parity;
    const size_t GN = (size_t)c->b.max_items * g.nmb;
    venc_bufs b = c->b;
    const int rs = (int)(c->step_no % VENC_RING), rs_prev = (int)((c->step_no + VENC_RING - 1) % VENC_RING);
    b.mvsad = c->b.mvsad + (size_t)rs * GN;
    b.mvsad_prev = c->b.mvsad + (size_t)rs_prev * GN;
    /* The first P step of a GOP has no previous P picture: the I step left zeros in its slot, the zero vector is a poor start
     * point for anything that moves, and every wave of the search took its second pass (124 us alone, 223 us beside the I
     * pictures' entropy kernels, where a steady step takes 74 / 94).  Two slots back lies the last P step of the PREVIOUS unit
     * (the ring is zeroed at open; the items there are the GOPs one unit earlier in the stream): its vectors are only a hint --
     * start points steer the pruning, never the result -- but for a camera that keeps moving they are a good one. */
    static const int gop_hint = getenv("VENC_GOP_HINT") ? atoi(getenv("VENC_GOP_HINT")) : 1;
    if (gop_hint && !is_idr && !c->last_step_p && c->step_no >= 2) b.mvsad_prev = c->b.mvsad + (size_t)((c->step_no + VENC_RING - 2) % VENC_RING) * GN;
    if (g.subpel) b.mvq = c->b.mvq + (size_t)rs * GN;
    b.levels = c->b.levels + (size_t)rs * GN * VENC_MB_COEFS;
    b.mbinfo = c->b.mbinfo + (size_t)rs * GN;
    if (c->cfg.rc_mode) {
        const size_t G_ = (size_t)c->b.max_items;
        b.qp_cur = c->d_qp_ring + (size_t)(c->step_no % VENC_QP_RING) * G_;
        b.qp_next = c->d_qp_ring + (size_t)((c->step_no + 1) % VENC_QP_RING) * G_;
        if (!is_idr) {
            b.sad_prev = c->d_sad_sum + (size_t)((c->rc_phase + 2) % 3) * G_;
            b.sad_cur = c->d_sad_sum + (size_t)c->rc_phase * G_;
            b.sad_next = c->d_sad_sum + (size_t)((c->rc_phase + 1) % 3) * G_;
            c->rc_phase = (c->rc_phase + 1) % 3;
        }
    }
    hipStream_t s1 = c->stream;
    bool chroma_deferred = false;
    /* an open entropy batch takes this step only if it continues it: same kind of picture (P), same items, same band,
     * the next ring slot of the same aligned group */
    if (c->eb_n > 0 && !(!is_idr && !c->eb_S.is_idr && n_items == c->cfg.max_items && c->eb_S.slice0 == S.slice0 && c->eb_S.slice1 == S.slice1 &&
                         rs == c->eb_first_slot + c->eb_n && rs / c->eb_max == c->eb_first_slot / c->eb_max && c->eb_n < c->eb_max))
        if (entropy_flush(c) != 0) return -1;
    if (c->overlap && c->step_no % VENC_RING_WAIT == 0) {
        /* the next WAIT steps overwrite the ring slots of steps step_no - RING .. step_no - RING + WAIT - 1: the entropy
         * batch that read the last of them has been launched (batches never span a group of WAIT slots) and covers the others */
        const long long need = c->step_no + VENC_RING_WAIT - 1 - VENC_RING;
        if (need >= 0 && c->side_step[need % VENC_RING] == need) {
            HIP_OK(hipStreamWaitEvent(s1, c->ev_side[need % VENC_RING], 0));
            if (c->two_chain) HIP_OK(hipStreamWaitEvent(c->stream_b, c->ev_side[need % VENC_RING], 0)); /* (the wait covers the next WAIT steps, whichever stream runs them) */
        }
    }
    /* the first P step after intra steps that ran ahead joins their stream */
    if (!is_idr && c->intra_pending) {
        HIP_OK(hipStreamWaitEvent(s1, c->ev_intra, 0));
        if (two) HIP_OK(hipStreamWaitEvent(c->stream_b, c->ev_intra, 0));
        c->intra_pending = 0;
    }
    if (is_idr) {
        if (intra_async) {
            /* The buffer this step writes (ring index step % VENC_RECON) was written by step - VENC_RECON and read last by
             * step - VENC_RECON + 1 (its search and transforms on stream 1, its chroma transform on stream 2): wait for the
             * EARLIEST entropy batch that ended at or after that step -- both its events; batches complete in order -- and,
             * for the ring slot of levels and records, for the entropy chain that last read it.  Nothing else orders this
             * step: it reads input pictures only.  (No such batch in the history: those steps do not exist.) */
            s1 = c->stream3;
            int best = -1;
            const long long last_reader = c->step_no - VENC_RECON + 1;
            for (int i = 0; i < VENC_BHIST; i++)
                if (c->bh[i].last_step >= 0 && c->bh[i].last_step >= last_reader && (best < 0 || c->bh[i].last_step < c->bh[best].last_step)) best = i;
            if (best >= 0) {
                HIP_OK(hipStreamWaitEvent(s1, c->bh[best].ev_main, 0));
                if (c->bh[best].has_b) HIP_OK(hipStreamWaitEvent(s1, c->bh[best].ev_main_b, 0));
                if (c->bh[best].has_chroma) HIP_OK(hipStreamWaitEvent(s1, c->bh[best].ev_chroma, 0));
            }
            const long long need = c->step_no - VENC_RING;
            if (need >= 0 && c->side_step[rs] == need) HIP_OK(hipStreamWaitEvent(s1, c->ev_side[rs], 0));
        } else {
            if (c->intra_pending) { /* (a synchronous intra step behind asynchronous ones: band mode switched on in between) */
                HIP_OK(hipStreamWaitEvent(s1, c->ev_intra, 0));
                c->intra_pending = 0;
            }
            HIP_OK(main_waits_for_chroma(c)); /* the intra kernels overwrite chroma planes the last P step's chroma launch may still read */
        }
        prof_begin(c, KID_INTRA, s1);
#ifdef VENC_LEGACY_KERNELS
        if (c->intra_legacy)
            hipLaunchKernelGGL(k_intra, dim3(band_slices, 1, n_items), dim3(64 * c->intra_waves), 0, s1, g, b, S);
        else
#endif
        if (c->cfg.intra_modes) {
            /* Intra16x16 vertical / horizontal / DC by SAD: the plain dependency wavefront (venc_intra2.hip k_intra_modes) */
            const venc_geom &gi = c->cfg.rc_mode ? c->g_idr : g;
            if (c->cfg.intra_modes >= 3)
                hipLaunchKernelGGL(k_intra_modes<true>, dim3(band_slices, 1, n_items), dim3(64 * (c->intra_waves < INTRA4_MAX_WAVES ? c->intra_waves : INTRA4_MAX_WAVES)), 0, s1, gi, b, S);
            else
                hipLaunchKernelGGL(k_intra_modes<false>, dim3(band_slices, 1, n_items), dim3(64 * c->intra_waves), 0, s1, gi, b, S);
        } else {
            /* DC prediction only: transform off the dependency chain, DC chain, reconstruction (venc_intra2.hip) */
            const venc_geom &gi = c->cfg.rc_mode ? c->g_idr : g; /* IDR pictures under rate control: their own quantiser constants */
            hipLaunchKernelGGL(k_intra_fwd, dim3((band_mbs + 7) / 8, 1, n_items), dim3(256), 0, s1, gi, b, S);
            hipLaunchKernelGGL(k_intra_dc, dim3(band_slices, 1, n_items), dim3(1024), 0, s1, gi, b, S, c->intra_cw);
            hipLaunchKernelGGL(k_intra_rec, dim3((band_mbs + 7) / 8, 1, n_items), dim3(256), 0, s1, gi, b, S);
        }
        prof_end(c, s1);
        if (c->cfg.deblock) launch_deblock(c, s1, c->cfg.rc_mode ? c->g_idr : g, b, S, n_items, 3);
        if (intra_async) {
            HIP_OK(hipEventRecord(c->ev_intra, s1));
            c->intra_pending = 1;
            s1 = c->stream;
        }
    } else {
        /* Two sets of fix counters alternate between P steps: this step's search fills one and zeroes the other (the
         * legacy searches do not: a memset node instead). */
        b.fix_count = c->b.fix_count + (size_t)c->fix_phase * c->b.max_items;
        b.fix_count_next = c->b.fix_count + (size_t)(c->fix_phase ^ 1) * c->b.max_items;
        c->fix_phase ^= 1;
        const bool fuse = c->fuse_luma && c->me_mode == 0 && !c->inter_legacy && !g.subpel; /* (refinement sits between search and transform) */
        const int fuse_arg = !fuse ? 0 : c->fuse_luma; /* 3: luma in the search kernels, blocks with a level behind them; 1: all of luma inside */
        int nz_regions = 0, nz_ntx = 1;
        uint32_t nz_magic = 0;
        if (c->me_mode != 0) HIP_OK(hipMemsetAsync(b.fix_count_next, 0, (size_t)c->b.max_items * 4, s1));
        if (c->x_pending && c->me_mode != 0) { /* (the first-generation searches do not split) */
            HIP_OK(hipStreamWaitEvent(s1, c->ev_x_done, 0));
            c->x_pending = 0;
        }
        prof_begin(c, KID_ME, s1);
#ifdef VENC_LEGACY_KERNELS
        if (c->me_mode == 2)
            hipLaunchKernelGGL(k_me_dense, dim3((g.mbw + ME_TILE - 1) / ME_TILE, band_rows, n_items), dim3(256), 0, s1, g, b, S);
        else if (c->me_mode == 1)
            hipLaunchKernelGGL(k_me, dim3((g.mbw + ME_TILE - 1) / ME_TILE, g.mbh, n_items), dim3(256), 0, s1, g, b, S);
        else if (c->me_mode == 3)
            hipLaunchKernelGGL(k_me_tile, dim3((g.mbw + T2_MBX - 1) / T2_MBX, (band_rows + T2_MBY - 1) / T2_MBY, n_items), dim3(512), 0,
                               s1, g, b, S);
        else
#endif
        {
            const int ntx = (g.mbw + T2_MBX - 1) / T2_MBX, nty = (band_rows + T2_MBY - 1) / T2_MBY;
            /* raster order, or (VENC_ME_XCD=1) runs of 64 tiles dealt to the XCDs in rotation: grid.x is then the picture's
             * tile count padded to whole rounds of eight runs (see k_me_pool) */
            const int runs = (ntx * nty + 63) / 64;
            const uint32_t tx_magic = ntx > 1 ? (uint32_t)((0x100000000ull + (uint64_t)ntx - 1) / (uint64_t)ntx) : 0u;
            const int me_xcd = c->me_xcd && !c->rc_comm && c->band_s0 == 0 && c->band_s1 == c->g.n_slices;
            const int gx = me_xcd ? 8 * 64 * ((runs + 7) / 8) : ntx * nty;
            /* (VENC_ME_LDS_PAD=bytes of unused dynamic LDS per block: an experiment that holds the search at two blocks per CU) */
            static const int lds_pad = getenv("VENC_ME_LDS_PAD") ? atoi(getenv("VENC_ME_LDS_PAD")) : 0;
            /* k_me_pool4 (venc_me4.hip): the same search laid out for four blocks per CU; it transforms in the deferred form only */
            /* VENC_ME_BLOCKS=3 selects k_me_pool (three blocks per CU); default: k_me_pool4.  Same box, ten items per launch
             * (gpurun_out/r03_defer2, r03_cus): 1080p 150.6 -> 158.2 Gpx/s; 2160p 206.4 -> 206.9, and 211.8 with the side chain on
             * 192 CUs (the default mask at that size): with every register and the whole LDS of a CU in the search's hands the side
             * chain waits for whole CUs wherever it may run. */
            static const int me_blocks = getenv("VENC_ME_BLOCKS") ? atoi(getenv("VENC_ME_BLOCKS")) : 4;
            const bool pool4 = me_blocks != 3 && (fuse_arg == 0 || fuse_arg == 3);
            /* An exchange in flight on stream_x (slice bands): the tile rows whose windows (tile rows +- 16 luma rows) stay inside the own band
             * go first, beside the exchange; the band's first / last tile row -- where a neighbour's rows are read -- after it.  A launch
             * takes its tiles from two runs: [tile0, tile0 + tile_split) and [tile1, ...), tile_n in all. */
            S.tile0 = S.tile1 = 0;
            S.tile_split = S.tile_n = ntx * nty;
            const bool has_up = S.row0 > 0, has_dn = S.row1 < g.mbh;
            const bool x_split = c->x_pending && pool4 && !two && !me_xcd && (has_up || has_dn) && nty >= 3;
            if (c->x_pending && !x_split) {
                HIP_OK(hipStreamWaitEvent(s1, c->ev_x_done, 0));
                c->x_pending = 0;
            }
            for (int part = 0; part < (x_split ? 2 : 1); part++) {
                int gxp = gx;
                if (x_split) {
                    const int lo = has_up ? 1 : 0, hi = nty - (has_dn ? 1 : 0);
                    if (part == 0) {
                        S.tile0 = lo * ntx, S.tile_split = S.tile_n = (hi - lo) * ntx, S.tile1 = 0;
                    } else {
                        HIP_OK(hipStreamWaitEvent(s1, c->ev_x_done, 0));
                        c->x_pending = 0;
                        c->x_split_steps++;
                        S.tile0 = 0, S.tile_split = has_up ? ntx : 0, S.tile1 = (nty - 1) * ntx, S.tile_n = ((has_up ? 1 : 0) + (has_dn ? 1 : 0)) * ntx;
                    }
                    gxp = S.tile_n;
                }
            for (int half = 0; half < (two ? 2 : 1); half++) {
                venc_step Sh = S;
                Sh.item0 = half ? n_a : 0;
                const int nh = two ? (half ? n_items - n_a : n_a) : n_items;
                hipStream_t sh = half ? c->stream_b : s1;
                if (half) prof_end(c, s1); /* (the timed launches are the main stream's: half the items) */
                if (pool4 && c->cfg.rc_mode)
                    hipLaunchKernelGGL(k_me_pool4<true>, dim3(gxp, 1, nh), dim3(512), (size_t)lds_pad, sh, g, b, Sh, ntx, nty, tx_magic, me_xcd, fuse_arg);
                else if (pool4)
                    hipLaunchKernelGGL(k_me_pool4<false>, dim3(gxp, 1, nh), dim3(512), (size_t)lds_pad, sh, g, b, Sh, ntx, nty, tx_magic, me_xcd, fuse_arg);
                else if (c->cfg.rc_mode)
                    hipLaunchKernelGGL(k_me_pool<true>, dim3(gx, 1, nh), dim3(512), (size_t)lds_pad, sh, g, b, Sh, ntx, nty, tx_magic, me_xcd, fuse_arg);
                else
                    hipLaunchKernelGGL(k_me_pool<false>, dim3(gx, 1, nh), dim3(512), (size_t)lds_pad, sh, g, b, Sh, ntx, nty, tx_magic, me_xcd, fuse_arg);
            }
            }
            S.tile0 = S.tile1 = 0;
            S.tile_split = S.tile_n = ntx * nty;
            if (fuse_arg == 3) nz_regions = 4 * ntx * nty, nz_ntx = ntx, nz_magic = tx_magic;
        }
        prof_end(c, s1);
        const bool flags_mbs = c->me_mode == 0 || c->me_mode == 3; /* these searches list macroblocks for the exhaustive search */
        if (flags_mbs) {
            prof_begin(c, KID_MEFIX, s1);
            /* the blocks that code the 4x4 blocks P5 listed, first in dispatch order: per item one per NZ_RPB segments, each takes an
             * equal share of the item's entries; every thread of such a block reads nz_rpt segment counts (a multiple of four) */
            /* (about sixteen follow-up blocks per item: NZ_RPB = 128 segments per block up to 1080p; at 2160p -- 8100 segments -- 256 or 512 per
             * block shorten the launch from 26.4 to 24.0 us and the pass by 1 %, 64 or 1024 lengthen it to 31 - 32: r04_rpb) */
            static const int nz_rpb_env = getenv("VENC_NZ_RPB") ? atoi(getenv("VENC_NZ_RPB")) : 0;
            const int nz_rpb = nz_rpb_env > 0 ? nz_rpb_env : (nz_regions + 15) / 16 > NZ_RPB ? (((nz_regions + 15) / 16 + 3) & ~3) : NZ_RPB;
            const int nz_blocks = (nz_regions + nz_rpb - 1) / nz_rpb;
            const int nz_rpt = (((nz_regions + 64 * FIX_WAVES - 1) / (64 * FIX_WAVES)) + 3) & ~3;
            const size_t fix_lds = nz_blocks ? (size_t)64 * FIX_WAVES * nz_rpt : 0;
            for (int half = 0; half < (two ? 2 : 1); half++) {
                venc_step Sh = S;
                Sh.item0 = half ? n_a : 0;
                const int nh = two ? (half ? n_items - n_a : n_a) : n_items;
                hipStream_t sh = half ? c->stream_b : s1;
                if (half) prof_end(c, s1);
                if (c->cfg.rc_mode)
                    hipLaunchKernelGGL(k_me_fix<true>, dim3(nh, nz_blocks + FIX_BLOCKS, 1), dim3(64 * FIX_WAVES), fix_lds, sh, g, b, Sh, fuse_arg, nz_blocks, nz_regions, nz_rpt, nz_ntx, nz_magic);
                else
                    hipLaunchKernelGGL(k_me_fix<false>, dim3(nh, nz_blocks + FIX_BLOCKS, 1), dim3(64 * FIX_WAVES), fix_lds, sh, g, b, Sh, fuse_arg, nz_blocks, nz_regions, nz_rpt, nz_ntx, nz_magic);
            }
            if (two) c->b_active = 1;
            prof_end(c, s1);
        }
#ifdef VENC_LEGACY_KERNELS
        if (c->inter_legacy) {
            prof_begin(c, KID_INTER, s1);
            hipLaunchKernelGGL(k_inter, dim3((band_mbs + 7) / 8, 1, n_items), dim3(256), 0, s1, g, b, S);
            prof_end(c, s1);
        } else
#endif
        if (g.subpel && c->me_mode == 0 && !c->inter_legacy) {
            /* sub-sample refinement (venc_subpel.hip): the reference's half-sample planes, then per macroblock the two
             * stages and the luma transform at the refined vector; vectors -> B.mvq in quarter samples.  The chroma
             * transform follows with the entropy batch on the second stream, as in the fused path. */
            prof_begin(c, KID_HPEL, s1);
            hipLaunchKernelGGL(k_hpel, dim3((g.cw / 4 + 63) / 64, (g.ch / HP_ROWS + 3) / 4, n_items), dim3(256), 0, s1, g, b, S);
            prof_end(c, s1);
            prof_begin(c, KID_SUBPEL, s1);
            hipLaunchKernelGGL(k_subpel, dim3((band_mbs + 4 * SP_K - 1) / (4 * SP_K), 1, n_items), dim3(256), 0, s1, g, b, S);
            prof_end(c, s1);
            chroma_deferred = true;
            if (c->cfg.deblock) { /* luma here, for the next search; chroma behind its transform on the second stream */
                venc_bufs bt = b;
                bt.mvsad = b.mvq;
                launch_deblock(c, s1, g, bt, S, n_items, 1);
            }
        } else if (!fuse) {
            /* luma transform and chroma transform: independent blocks of one launch */
            const int n_luma = (band_mbs + 15) / 16, n_chroma = (band_mbs + 31) / 32;
            prof_begin(c, KID_INTER, s1);
            hipLaunchKernelGGL(k_inter_fused, dim3(n_luma + n_chroma, 1, n_items), dim3(256), 0, s1, g, b, S, n_luma);
            prof_end(c, s1);
            if (c->cfg.deblock) launch_deblock(c, s1, g, b, S, n_items, 3);
        } else {
            /* The main chain ends here: search + luma transform, the flagged macroblocks' redo.  The next step's search
             * needs nothing else (it reads luma only).  The chroma transforms of a batch of steps follow on the second
             * stream, one after the other (chroma reconstructions chain from step to step), ahead of the batch's entropy
             * chain and beside the next searches: ONE event hand-off per batch, none between the kernels of the main chain. */
            chroma_deferred = true;
            /* in-loop filter: luma here (the next search reads it); chroma behind its transform on the second stream */
            if (c->cfg.deblock) launch_deblock(c, s1, g, b, S, n_items, chroma_deferred ? 1 : 3);
        }
    }
#ifdef VENC_LEGACY_KERNELS
    if (c->me_mode == 1) { /* block sums of the frame just reconstructed = reference of the next P step */
        prof_begin(c, KID_REFSUM, s1);
        hipLaunchKernelGGL(k_refsum, dim3((g.cw + 17 + RS_TW - 1) / RS_TW, (g.ch + 25 + RS_TH - 1) / RS_TH, n_items), dim3(256), 0,
                           s1, g, b, S);
        prof_end(c, s1);
    }
#endif
    HIP_OK(hipGetLastError());
    /* the step joins the entropy batch as (step, item) pairs k * max_items + i */
    if (c->eb_n == 0) {
        c->eb_S = S;
        c->eb_first_slot = rs;
        c->eb_first_step = c->step_no;
    }
    {
        const int k = c->eb_n, G = c->cfg.max_items;
        for (int i = 0; i < n_items; i++) {
            c->eb_S.out_slot[k * G + i] = S.out_slot[i];
            c->eb_S.idr_id[k * G + i] = S.idr_id[i];
            c->eb_S.fnum[k * G + i] = (uint8_t)(frame_num & 15);
        }
        c->eb_step[k] = S;
        if (k == 0) c->eb_chroma = 0;
        if (chroma_deferred) c->eb_chroma = 1;
        c->intra_async_last = intra_async;
        c->eb_n++;
    }
    c->rec_last = c->parity;
    c->parity = (c->parity + 1) % VENC_RECON;
    c->last_step_p = !is_idr;
    c->step_no++;
    c->steps_done++;
    /* closed: I pictures (their own kernel), ragged steps, the last slot of the group, a full batch */
    if (is_idr || n_items != c->cfg.max_items || c->eb_n == c->eb_max || (rs + 1) % c->eb_max == 0) return entropy_flush(c);
    return 0;
}

/* launches the entropy chain of the open batch (second stream, behind the batch's main chains) */
static int entropy_flush(venc_ctx *c) {
    if (c->eb_n == 0) return 0;
    HIP_OK(hipSetDevice(c->cfg.device));
    const venc_geom &g = c->g;
    const int G = c->cfg.max_items;
    const size_t GN = (size_t)c->b.max_items * g.nmb;
    venc_step S = c->eb_S;
    const int n_steps = c->eb_n, rs0 = c->eb_first_slot;
    const int nv = n_steps == 1 ? S.n_items : n_steps * G; /* (step, item) pairs; a batch of several steps has max_items each */
    S.n_items = nv;
    const int is_idr = S.is_idr;
    const int band_mbs = S.mb1 - S.mb0, band_slices = S.slice1 - S.slice0;
    venc_bufs b = c->b;
    b.mvsad = (g.subpel && !S.is_idr ? c->b.mvq : c->b.mvsad) + (size_t)rs0 * GN; /* (refined vectors: quarter samples) */
    b.levels = c->b.levels + (size_t)rs0 * GN * VENC_MB_COEFS;
    b.mbinfo = c->b.mbinfo + (size_t)rs0 * GN;
    if (c->cfg.rc_mode) b.qp_cur = c->d_qp_ring + (size_t)(c->eb_first_step % VENC_QP_RING) * c->b.max_items;
    hipStream_t s1 = c->stream, s2 = c->overlap ? c->stream2 : c->stream;
    const int hi = c->bh_next;
    if (c->overlap) {
        if (is_idr && c->intra_async_last) {
            HIP_OK(hipStreamWaitEvent(s2, c->ev_intra, 0)); /* the intra step ran on stream3 */
            c->bh[hi].last_step = -1;                       /* (not a batch an intra step running ahead could wait for) */
        } else {
            HIP_OK(hipEventRecord(c->bh[hi].ev_main, s1));
            HIP_OK(hipStreamWaitEvent(s2, c->bh[hi].ev_main, 0));
            c->bh[hi].has_b = c->b_active;
            if (c->b_active) { /* (the second chain stays ahead of the main stream: only the side chain waits for it here) */
                HIP_OK(hipEventRecord(c->bh[hi].ev_main_b, c->stream_b));
                HIP_OK(hipStreamWaitEvent(s2, c->bh[hi].ev_main_b, 0));
            }
            c->bh[hi].last_step = c->eb_first_step + n_steps - 1;
            c->bh[hi].has_chroma = 0;
            c->bh_next = (hi + 1) % VENC_BHIST;
        }
    }
    if (c->eb_chroma) {
        for (int k = 0; k < n_steps; k++) {
            venc_bufs bk = c->b;
            const int rsk = rs0 + k;
            bk.mvsad = (g.subpel ? c->b.mvq : c->b.mvsad) + (size_t)rsk * GN; /* (refined vectors: quarter samples) */
            bk.levels = c->b.levels + (size_t)rsk * GN * VENC_MB_COEFS;
            bk.mbinfo = c->b.mbinfo + (size_t)rsk * GN;
            if (c->cfg.rc_mode) bk.qp_cur = c->d_qp_ring + (size_t)((c->eb_first_step + k) % VENC_QP_RING) * c->b.max_items;
            prof_begin(c, KID_INTER, s2);
            hipLaunchKernelGGL(k_inter_fused, dim3((band_mbs + 31) / 32, 1, c->eb_step[k].n_items), dim3(256), 0, s2, g, bk, c->eb_step[k], 0);
            prof_end(c, s2);
            if (c->cfg.deblock) launch_deblock(c, s2, g, bk, c->eb_step[k], c->eb_step[k].n_items, 2); /* (the next step's chroma transform predicts from the filtered planes) */
        }
        if (c->overlap) {
            HIP_OK(hipEventRecord(c->ev_chroma, s2));
            c->chroma_pending = 1;
            if (c->bh[hi].last_step >= 0) {
                HIP_OK(hipEventRecord(c->bh[hi].ev_chroma, s2));
                c->bh[hi].has_chroma = 1;
            }
        }
        c->eb_chroma = 0;
    }
    prof_begin(c, KID_CAVLC, s2);
    if (is_idr)
        hipLaunchKernelGGL(k_cavlc, dim3((band_mbs + CAVLC_MBS - 1) / CAVLC_MBS, 1, nv), dim3(256), 0, s2, g, b, S);
    else {
        /* P pictures (venc_cavlc_p.hip): one block per segment of 256 macroblocks of a slice in both kernels */
        const dim3 eg(c->b.ent_spp, band_slices, nv);
        hipLaunchKernelGGL(k_entropy_p, eg, dim3(ENT_THREADS), (size_t)ENT_THREADS * 36 + (size_t)ENT_LEN_LDS * 2 + (size_t)ENT_CLS_LDS * 8, s2, g, b, S);
    }
    prof_end(c, s2);
    if (c->overlap) {
        /* the ring slots of the batch are free again once these kernels are done (k_scan and the pack kernels read the chain's own buffers) */
        for (int k = 0; k < n_steps; k++) {
            HIP_OK(hipEventRecord(c->ev_side[rs0 + k], s2));
            c->side_step[rs0 + k] = c->eb_first_step + k;
        }
    }
    if (is_idr) {
        prof_begin(c, KID_SCAN, s2);
        hipLaunchKernelGGL(k_scan, dim3(band_slices, nv), dim3(64 * SCAN_WAVES), 0, s2, g, b, S, c->d_slice);
        prof_end(c, s2);
        prof_begin(c, KID_PACK, s2);
        hipLaunchKernelGGL(k_pack, dim3((c->max_slice_mbs + PACK_MBS - 1) / PACK_MBS, band_slices, nv), dim3(256), 0, s2, g, b, S,
                           (const venc_slice_out *)c->d_slice);
        prof_end(c, s2);
    } else {
        prof_begin(c, KID_PACK, s2);
        hipLaunchKernelGGL(k_pack_p, dim3(c->b.ent_spp, band_slices, nv), dim3(ENT_THREADS), 0, s2, g, b, S, c->d_slice);
        prof_end(c, s2);
    }
    c->last_batch_p = !is_idr;
    c->last_batch_S = S;
    HIP_OK(hipGetLastError());
    /* tickets handed out while the batch was open complete here */
    for (int i = 0; i < c->eb_nt; i++) HIP_OK(hipEventRecord(c->ev[c->eb_tickets[i] % c->n_events], s2));
    c->eb_nt = 0;
    c->eb_n = 0;
    return 0;
}

/* ---------------------------------------------------------------- slice-parallel multi-GPU mode */
extern "C" int venc_set_band(venc_ctx *c, int slice_first, int slice_count) {
    if (slice_first == 0 && slice_count == 0) slice_count = c->g.n_slices;
    if (slice_first < 0 || slice_count < 1 || slice_first + slice_count > c->g.n_slices) {
        SET_ERR("venc_set_band: slices [%d, %d) outside 0..%d", slice_first, slice_first + slice_count, c->g.n_slices);
        return -1;
    }
    if (c->g.subpel && slice_count != c->g.n_slices) {
        SET_ERR("venc_set_band: sub-sample refinement needs the whole picture (a band's halo of 16 rows is three short of the six-tap filter's reach)");
        return -1;
    }
    if (c->cfg.rc_mode && slice_count != c->g.n_slices) {
        SET_ERR("venc_set_band: rate control needs the whole picture (its SAD sum spans all slices)");
        return -1;
    }
    if ((c->me_mode == 1 || c->inter_legacy) && slice_count != c->g.n_slices) {
        SET_ERR("venc_set_band: the diagnostic kernels (VENC_ME=plane, VENC_INTER=legacy) code whole pictures only");
        return -1;
    }
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(main_joins_b(c));
    c->band_s0 = slice_first;
    c->band_s1 = slice_first + slice_count;
    return 0;
}

extern "C" int venc_band_rows(const venc_ctx *c, int slice_first, int slice_count, int *mb_row0, int *mb_row1) {
    if (slice_first < 0 || slice_count < 1 || slice_first + slice_count > c->g.n_slices) {
        SET_ERR("venc_band_rows: slices [%d, %d) outside 0..%d", slice_first, slice_first + slice_count, c->g.n_slices);
        return -1;
    }
    if (mb_row0) *mb_row0 = c->slice_rows[slice_first];
    if (mb_row1) *mb_row1 = c->slice_rows[slice_first + slice_count];
    return 0;
}

/* luma rows carried per band: the whole band, or (halo_rows > 0) its first and last halo_rows rows */
static int band_job(const venc_ctx *c, int row0, int row1, int halo_rows, venc_band_job *J) {
    const venc_geom &g = c->g;
    if (halo_rows < 0 || (halo_rows & 15)) return -1;
    const int ly0 = row0 * 16, ly1 = row1 * 16;
    const uint32_t ysz = (uint32_t)g.cw * g.ch;
    memset(J, 0, sizeof(*J));
    uint32_t off = 0;
    for (int plane = 0; plane < 3; plane++) {
        const int sh = plane ? 1 : 0;
        const uint32_t pw = (uint32_t)(g.cw >> sh), pbase = plane == 0 ? 0u : plane == 1 ? ysz : ysz + ysz / 4;
        const int y0 = ly0 >> sh, y1 = ly1 >> sh, hr = halo_rows >> sh;
        if (halo_rows == 0 || 2 * hr >= y1 - y0) { /* the whole band */
            J->seg[J->n_seg++] = (venc_band_seg){pbase + (uint32_t)y0 * pw, off, (uint32_t)(y1 - y0) * pw};
            off += (uint32_t)(y1 - y0) * pw;
        } else {
            J->seg[J->n_seg++] = (venc_band_seg){pbase + (uint32_t)y0 * pw, off, (uint32_t)hr * pw};
            off += (uint32_t)hr * pw;
            J->seg[J->n_seg++] = (venc_band_seg){pbase + (uint32_t)(y1 - hr) * pw, off, (uint32_t)hr * pw};
            off += (uint32_t)hr * pw;
        }
    }
    return (int)off;
}

extern "C" size_t venc_band_chunk_bytes(const venc_ctx *c, int mb_rows, int halo_rows) {
    venc_band_job J;
    if (mb_rows < 1 || mb_rows > c->g.mbh) return 0;
    const int n = band_job(c, 0, mb_rows, halo_rows, &J);
    return n < 0 ? 0 : (size_t)n;
}

/* what a band copy on the main stream (or the start of an exchange on stream_x) follows: the open entropy batch (its chroma transforms are
 * launched with it), the second chain, the chroma transforms, an intra step that ran ahead */
static int band_main_ready(venc_ctx *c) {
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(main_joins_b(c));
    c->b_behind = 1;
    HIP_OK(main_waits_for_chroma(c));
    if (c->intra_pending) { /* (band engines code their intra steps on the main stream; an engine may change mode) */
        HIP_OK(hipStreamWaitEvent(c->stream, c->ev_intra, 0));
        c->intra_pending = 0;
    }
    return 0;
}

static int band_copy(venc_ctx *c, int n_items, void *d_chunk, size_t item_stride, int row0, int row1, int halo_rows, int to_chunk) {
    venc_band_job J;
    const int bytes = band_job(c, row0, row1, halo_rows, &J);
    if (n_items < 1 || n_items > c->cfg.max_items || !d_chunk || bytes < 0 || (item_stride & 15) || item_stride < (size_t)bytes ||
        (((uintptr_t)d_chunk) & 15)) {
        SET_ERR("venc_band_%s: bad arguments (items %d, stride %zu, chunk %d bytes, halo %d)", to_chunk ? "export" : "import", n_items,
                item_stride, bytes, halo_rows);
        return -1;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    if (!c->x_active && band_main_ready(c) != 0) return -1;
    J.to_chunk = to_chunk;
    J.item_stride = item_stride;
    uint32_t big = 0;
    for (int i = 0; i < J.n_seg; i++) big = J.seg[i].bytes > big ? J.seg[i].bytes : big;
    int blocks = (int)((big / 16 + 1023) / 1024); /* four trips per lane */
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_band_copy, dim3(blocks, J.n_seg, n_items), dim3(256), 0, c->x_active ? c->stream_x : c->stream, c->b, c->rec_last, J, (uint8_t *)d_chunk);
    HIP_OK(hipGetLastError());
    return bytes;
}

/* the exchange stream (see venc_ctx): between these two calls venc_band_export / venc_band_import are enqueued on it */
extern "C" int venc_band_xbegin(venc_ctx *c) {
    if (!c || c->x_active) {
        SET_ERR("venc_band_xbegin: %s", c ? "already open" : "no context");
        return -1;
    }
    if (!c->x_overlap) return 0; /* (everything stays on the main stream) */
    HIP_OK(hipSetDevice(c->cfg.device));
    if (band_main_ready(c) != 0) return -1;
    if (c->x_pending) { /* (an exchange nobody consumed: the main stream takes it in before the next one starts) */
        HIP_OK(hipStreamWaitEvent(c->stream, c->ev_x_done, 0));
        c->x_pending = 0;
    }
    HIP_OK(hipEventRecord(c->ev_x_main, c->stream));
    HIP_OK(hipStreamWaitEvent(c->stream_x, c->ev_x_main, 0));
    c->x_active = 1;
    return 0;
}
extern "C" int venc_band_xend(venc_ctx *c) {
    if (!c) return -1;
    if (!c->x_active) return 0;
    HIP_OK(hipSetDevice(c->cfg.device));
    HIP_OK(hipEventRecord(c->ev_x_done, c->stream_x));
    c->x_active = 0;
    c->x_pending = 1;
    return 0;
}
/* the stream band copies and the collective of an open exchange belong on (the main stream without one) */
extern "C" void *venc_band_stream(venc_ctx *c) { return c ? (void *)(c->x_active ? c->stream_x : c->stream) : NULL; }
extern "C" long long venc_band_split_steps(const venc_ctx *c) { return c ? c->x_split_steps : 0; }

extern "C" int venc_band_export(venc_ctx *c, int n_items, void *d_chunk, size_t item_stride, int halo_rows) {
    return band_copy(c, n_items, d_chunk, item_stride, c->slice_rows[c->band_s0], c->slice_rows[c->band_s1], halo_rows, 1);
}

extern "C" int venc_band_import(venc_ctx *c, int n_items, const void *d_chunk, size_t item_stride, int slice_first, int slice_count,
                                int halo_rows) {
    int r0, r1;
    if (venc_band_rows(c, slice_first, slice_count, &r0, &r1) != 0) return -1;
    return band_copy(c, n_items, (void *)d_chunk, item_stride, r0, r1, halo_rows, 0);
}

extern "C" void *venc_stream_handle(venc_ctx *c) { return (void *)c->stream; }
/* launches the search of a full P step is split into (1, or 2 with two chains): what a timed k_me launch covers */
extern "C" int venc_search_split(const venc_ctx *c) { return c->two_chain ? 2 : 1; }

extern "C" int venc_mark(venc_ctx *c) {
    const int t = c->ev_next;
    if (c->eb_n > 0) {
        /* the entropy chain of the steps issued so far is not launched yet: the ticket completes with the batch */
        if (c->eb_nt >= (int)(sizeof(c->eb_tickets) / sizeof(c->eb_tickets[0])) && entropy_flush(c) != 0) return -1;
    }
    if (c->eb_n > 0)
        c->eb_tickets[c->eb_nt++] = t;
    else
        HIP_OK(hipEventRecord(c->ev[t % c->n_events], c->overlap ? c->stream2 : c->stream));
    c->ev_next++;
    return t;
}
/* launches whatever the engine has deferred (the open entropy batch): after this everything issued completes without
 * further calls.  venc_wait / venc_sync / venc_debug_read do it themselves; a caller that polls needs it. */
extern "C" int venc_flush(venc_ctx *c) { return entropy_flush(c); }
extern "C" int venc_wait(venc_ctx *c, int ticket) {
    if (ticket < 0 || ticket >= c->ev_next || c->ev_next - ticket > c->n_events) {
        SET_ERR("venc_wait: stale ticket %d", ticket);
        return -1;
    }
    /* tickets complete in order (one stream records them): the frames of a step share a ticket, and a later ticket
     * that has been waited for covers the earlier ones -- no second trip into the runtime for them */
    if (ticket <= c->ev_done) return 0;
    if (c->eb_nt > 0 && ticket >= c->eb_tickets[0] && entropy_flush(c) != 0) return -1; /* its batch is still open */
    HIP_OK(hipEventSynchronize(c->ev[ticket % c->n_events]));
    c->ev_done = ticket;
    return 0;
}
/* Thread-safe subset for a second host thread (the shim's packaging thread): bind the device once, then wait for tickets
 * whose work has already been handed to the GPU (venc_flush / venc_wait was called after the ticket was issued).  Touches
 * no engine state. */
extern "C" int venc_bind_thread(venc_ctx *c) {
    HIP_OK(hipSetDevice(c->cfg.device));
    return 0;
}
extern "C" int venc_wait_issued(venc_ctx *c, int ticket) {
    if (ticket < 0) {
        SET_ERR("venc_wait_issued: bad ticket %d", ticket);
        return -1;
    }
    HIP_OK(hipEventSynchronize(c->ev[ticket % c->n_events]));
    return 0;
}
extern "C" int venc_sync(venc_ctx *c) {
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(hipStreamSynchronize(c->stream_x));
    c->x_pending = 0;
    HIP_OK(hipStreamSynchronize(c->stream_copy));
    HIP_OK(hipStreamSynchronize(c->stream3));
    HIP_OK(hipStreamSynchronize(c->stream_b));
    c->b_active = 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipStreamSynchronize(c->stream2));
    return 0;
}

/* 1 (default): the entropy chain of a step runs on a second stream beside the next step's main
 * chain; 0: every kernel on one stream (isolated per-kernel timings for the roofline report) */
extern "C" int venc_set_overlap(venc_ctx *c, int enable) {
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(hipStreamSynchronize(c->stream3));
    HIP_OK(hipStreamSynchronize(c->stream_b));
    c->b_active = 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipStreamSynchronize(c->stream2));
    c->chroma_pending = 0;
    c->intra_pending = 0;
    c->overlap = enable ? 1 : 0;
    for (int i = 0; i < VENC_RING; i++) c->side_step[i] = -1; /* both streams are idle: nothing to wait for */
    return 0;
}

/* Staged engines: copy the encoded bytes of `count` output slots first_slot, first_slot + slot_step, ... (frames whose step
 * has completed -- the caller waited for their ticket) from the HBM staging buffer to the pinned host pool: ONE strided DMA
 * copy of the widest frame's byte count per slot, on a stream of its own, then a wait.  No-op for slots already fetched and
 * in direct engines.  Safe on the packaging thread beside venc_encode_step on another thread (a slot is reused only after
 * its frame was delivered). */
extern "C" int venc_fetch(venc_ctx *c, int first_slot, int slot_step, int count) {
    if (!c->staged) return 0;
    if (count < 1 || first_slot < 0 || slot_step < 1 || first_slot + (count - 1) * slot_step >= c->cfg.out_slots) {
        SET_ERR("venc_fetch: bad slot range (%d, step %d, count %d)", first_slot, slot_step, count);
        return -1;
    }
    if (count > c->host_slots) { /* (more frames than host buffers: in runs that fit) */
        for (int i = 0; i < count; i += c->host_slots) {
            const int n = count - i < c->host_slots ? count - i : c->host_slots;
            if (venc_fetch(c, first_slot + i * slot_step, slot_step, n) != 0) return -1;
        }
        return 0;
    }
    uint32_t widest = 0;
    int need = 0;
    for (int i = 0; i < count; i++) {
        const int slot = first_slot + i * slot_step;
        const venc_out_header *h = (const venc_out_header *)(c->h_hdr + (size_t)slot * c->b.out_hdr_stride);
        if (c->fetched[slot] || h->error) continue;
        need++;
        if (h->total_bytes > widest) widest = h->total_bytes;
    }
    if (!need) return 0;
    if (widest > c->b.out_cap) widest = c->b.out_cap;
    /* `count` consecutive host buffers, taken in rotation; whatever they held is no longer on the host */
    const int h0 = c->host_next + count <= c->host_slots ? c->host_next : 0;
    c->host_next = (h0 + count) % c->host_slots;
    for (int i = 0; i < count; i++) {
        const int slot = first_slot + i * slot_step, old = c->host_owner[h0 + i];
        if (old >= 0 && c->host_of[old] == h0 + i) c->host_of[old] = -1, c->fetched[old] = 0;
        if (c->host_of[slot] >= 0 && c->host_owner[c->host_of[slot]] == slot) c->host_owner[c->host_of[slot]] = -1; /* (fetched again: its earlier copy is stale) */
        c->host_owner[h0 + i] = slot;
        c->host_of[slot] = h0 + i;
    }
    if (widest) {
        const size_t src = (size_t)first_slot * c->b.out_data_stride, spitch = (size_t)slot_step * c->b.out_data_stride;
        uint8_t *dst = c->h_data + (size_t)h0 * c->b.out_data_stride;
        if (count == 1)
            HIP_OK(hipMemcpyAsync(dst, c->d_data + src, widest, hipMemcpyDeviceToHost, c->stream_d2h));
        else
            HIP_OK(hipMemcpy2DAsync(dst, c->b.out_data_stride, c->d_data + src, spitch, widest, (size_t)count, hipMemcpyDeviceToHost, c->stream_d2h));
        HIP_OK(hipStreamSynchronize(c->stream_d2h));
    }
    for (int i = 0; i < count; i++) c->fetched[first_slot + i * slot_step] = 1;
    return 0;
}

extern "C" int venc_get_frame(venc_ctx *c, int slot, venc_frame *out) {
    if (slot < 0 || slot >= c->cfg.out_slots || !out) {
        SET_ERR("venc_get_frame: bad slot %d", slot);
        return -1;
    }
    const uint8_t *p = c->h_hdr + (size_t)slot * c->b.out_hdr_stride;
    const venc_out_header *h = (const venc_out_header *)p;
    if (c->staged && !c->fetched[slot] && !h->error && venc_fetch(c, slot, 1, 1) != 0) return -1;
    out->data = c->h_data + (size_t)(c->staged ? (c->host_of[slot] >= 0 ? c->host_of[slot] : 0) : slot) * c->b.out_data_stride;
    out->slices = (const venc_slice *)(p + sizeof(venc_out_header));
    out->n_slices = h->n_slices;
    out->total_bytes = h->total_bytes;
    out->error = h->error;
    if (h->error) {
        SET_ERR("venc_get_frame: output capacity exceeded in slot %d", slot);
        return -3;
    }
    return 0;
}

extern "C" int venc_debug_read(venc_ctx *c, int what, int item, void *dst, size_t dst_bytes) {
    if (item < 0 || item >= c->cfg.max_items || !dst) {
        SET_ERR("venc_debug_read: bad item");
        return -1;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(hipStreamSynchronize(c->stream_b));
    c->b_active = 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipStreamSynchronize(c->stream2));
    const venc_geom &g = c->g;
    const size_t nmb = (size_t)g.nmb;
    const void *src = NULL;
    size_t n = 0;
    HIP_OK(hipStreamSynchronize(c->stream3));
    const int last_parity = c->rec_last;
    const size_t lp_off = (size_t)((c->step_no + VENC_RING - 1) % VENC_RING) * c->b.max_items; /* ring slot of the last step */
    const uint8_t *rec = c->b.recon + (size_t)(last_parity * c->b.max_items + item) * c->b.recon_frame_bytes;
    switch (what) {
        case 0: src = c->b.mvsad + (lp_off + item) * nmb, n = nmb * 4; break; /* the search's vectors, integer samples */
        case 14: /* the coded vectors in quarter samples (sub-sample refinement) */
            if (!c->b.mvq) {
                SET_ERR("venc_debug_read: no refined vectors (subpel = 0)");
                return -1;
            }
            src = c->b.mvq + (lp_off + item) * nmb, n = nmb * 4;
            break;
        case 15: /* the half-sample planes b, h, j of the last P step's reference picture */
            if (!c->b.hpel) {
                SET_ERR("venc_debug_read: no half-sample planes (subpel = 0)");
                return -1;
            }
            src = c->b.hpel + (size_t)item * 3 * g.cw * g.ch, n = (size_t)3 * g.cw * g.ch;
            break;
        case 1: src = c->b.levels + (lp_off + item) * nmb * VENC_MB_COEFS, n = nmb * VENC_MB_COEFS * 2; break;
        case 2: src = c->b.mbinfo + (lp_off + item) * nmb, n = nmb * sizeof(venc_mbinfo); break;
        case 3:
            if (c->last_batch_p) { /* P pictures: the sum of each macroblock's entry lengths (the entropy chain keeps no per-macroblock count) */
                HIP_OK(hipMemset(c->b.mbnbits + item * nmb, 0, nmb * 4));
                hipLaunchKernelGGL(k_entry_mbbits, dim3(c->b.ent_spp, c->last_batch_S.slice1 - c->last_batch_S.slice0, 1), dim3(ENT_THREADS), 0, 0, g, c->b, c->last_batch_S, item);
                HIP_OK(hipDeviceSynchronize());
            }
            src = c->b.mbnbits + item * nmb, n = nmb * 4;
            break;
        case 16: src = c->b.ent_count + (size_t)item * g.n_slices * c->b.ent_spp, n = (size_t)g.n_slices * c->b.ent_spp * 4; break; /* P pictures: entries per segment */
        case 18: src = c->b.ent_list + (size_t)item * g.n_slices * c->b.ent_spp * c->b.ent_cap, n = (size_t)g.n_slices * c->b.ent_spp * c->b.ent_cap * 4; break;
        case 19: src = c->b.ent_off + (size_t)item * g.n_slices * c->b.ent_spp * (c->b.ent_cap + 4), n = (size_t)g.n_slices * c->b.ent_spp * (c->b.ent_cap + 4) * 4; break;
        case 20: src = c->b.ent_cls + (size_t)item * g.n_slices * c->b.ent_spp * c->b.ent_cap, n = (size_t)g.n_slices * c->b.ent_spp * c->b.ent_cap * 4; break;
        case 17: src = c->b.seg_bits + (size_t)item * g.n_slices * c->b.ent_spp, n = (size_t)g.n_slices * c->b.ent_spp * 4; break;  /* ... and bits */
        case 4: src = rec, n = (size_t)g.cw * g.ch; break;
        case 5: src = rec + (size_t)g.cw * g.ch, n = (size_t)g.cw * g.ch / 4; break;
        case 6: src = rec + (size_t)g.cw * g.ch * 5 / 4, n = (size_t)g.cw * g.ch / 4; break;
        case 7: src = c->b.mboff + item * nmb, n = nmb * 4; break;
        case 9: src = c->b.me_cnt + item * nmb, n = nmb; break;
        case 11: src = c->b.stamps, n = VENC_STAMP_WORDS * sizeof(unsigned long long); break;
        case 10: src = c->b.me_flag + item * nmb, n = nmb; break;
        case 12: src = c->b.intra_d0 + item * nmb * 32, n = nmb * 32 * 4; break;
        case 13: src = c->b.intra_rec + item * nmb * INTRA_REC_DWORDS, n = nmb * INTRA_REC_DWORDS * 4; break;
        case 8:
            if (!c->b.refsum) {
                SET_ERR("venc_debug_read: the block-sums plane exists in the legacy test library only");
                return -1;
            }
            src = c->b.refsum + item * c->b.refsum_item, n = c->b.refsum_item * 4;
            break;
        default: SET_ERR("venc_debug_read: unknown selector %d", what); return -1;
    }
    if (dst_bytes < n) {
        SET_ERR("venc_debug_read: destination too small (%zu < %zu)", dst_bytes, n);
        return -1;
    }
    HIP_OK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    if (what == 1) {
        /* the kernels do not write blocks that hold no level (and k_cavlc never reads them):
         * present the record with those blocks zeroed, as the level record is defined */
        venc_mbinfo *mi = (venc_mbinfo *)malloc(nmb * sizeof(venc_mbinfo));
        if (!mi) return -1;
        HIP_OK(hipMemcpy(mi, c->b.mbinfo + (lp_off + item) * nmb, nmb * sizeof(venc_mbinfo), hipMemcpyDeviceToHost));
        int16_t *lv = (int16_t *)dst;
        static const int blk_raster[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
        for (size_t m = 0; m < nmb; m++) {
            int16_t *r = lv + m * VENC_MB_COEFS;
            const int i16 = (mi[m].flags & 1) && !(mi[m].flags & 8), cbpc = mi[m].cbpc; /* (Intra4x4: slot 16 holds the modes, not levels) */
            for (int b = 0; b < 16; b++)
                if (mi[m].nnz[blk_raster[b]] == 0) memset(r + b * 16, 0, 32);
            if (!i16) memset(r + 16 * 16, 0, 32);
            for (int k = 0; k < 8; k++)
                if (mi[m].nnz[16 + k] == 0) memset(r + (19 + k) * 16, 0, 32);
            memset(r + 17 * 16 + 4, 0, 24);
            memset(r + 18 * 16 + 4, 0, 24);
            (void)cbpc;
        }
        free(mi);
    }
    return 0;
}

/* ------------------------------------------------------------------ per-kernel timing */
extern "C" int venc_profile(venc_ctx *c, int enable, int capacity) {
    HIP_OK(hipSetDevice(c->cfg.device));
    if (enable && !c->prof_ev) {
        c->prof_cap = capacity > 0 ? capacity : 4096;
        c->prof_ev = (hipEvent_t *)calloc((size_t)2 * c->prof_cap, sizeof(hipEvent_t));
        c->prof_kid = (int *)calloc((size_t)c->prof_cap, sizeof(int));
        if (!c->prof_ev || !c->prof_kid) return -1;
        for (int i = 0; i < 2 * c->prof_cap; i++) HIP_OK(hipEventCreate(&c->prof_ev[i]));
    }
    c->prof_on = enable ? 1 : 0;
    c->prof_mask = 0xffffffffu;
    c->prof_stride = 1;
    memset(c->prof_seen, 0, sizeof(c->prof_seen));
    c->prof_n = 0;
    return 0;
}

extern "C" int venc_profile_select(venc_ctx *c, unsigned kernel_mask, int stride) {
    c->prof_mask = kernel_mask;
    c->prof_stride = stride > 0 ? stride : 1;
    memset(c->prof_seen, 0, sizeof(c->prof_seen));
    return 0;
}
extern "C" int venc_profile_kernels(void) { return KID_COUNT; }
extern "C" const char *venc_profile_name(int kid) { return kid >= 0 && kid < KID_COUNT ? k_kernel_names[kid] : ""; }
/* sums the recorded launches per kernel (after a stream sync) and resets the log */
extern "C" int venc_profile_read(venc_ctx *c, double *ms_sum, int *launches) {
    if (entropy_flush(c) != 0) return -1;
    HIP_OK(hipStreamSynchronize(c->stream_b));
    c->b_active = 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipStreamSynchronize(c->stream2));
    for (int k = 0; k < KID_COUNT; k++) ms_sum[k] = 0.0, launches[k] = 0;
    for (int i = 0; i < c->prof_n; i++) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        ms_sum[c->prof_kid[i]] += ms;
        launches[c->prof_kid[i]]++;
    }
    const int dropped = c->prof_on && c->prof_n >= c->prof_cap;
    c->prof_n = 0;
    return dropped ? 1 : 0;
}

/* ------------------------------------------------------------------ synthetic source */
extern "C" void venc_synth_frame(int width, int height, int frame_index, uint32_t seed, uint8_t *y, uint8_t *u,
                                 uint8_t *v) {
    for (int r = 0; r < height; r++)
        for (int x = 0; x < width; x++) y[(size_t)r * width + x] = venc_synth_pixel(0, x, r, frame_index, width, height, seed);
    const int cw = width / 2, chh = height / 2;
    for (int r = 0; r < chh; r++)
        for (int x = 0; x < cw; x++) {
            u[(size_t)r * cw + x] = venc_synth_pixel(1, x, r, frame_index, width, height, seed);
            v[(size_t)r * cw + x] = venc_synth_pixel(2, x, r, frame_index, width, height, seed);
        }
}

__global__ void k_synth(uint8_t *frames, size_t frame_stride, int w, int h, int first_frame, uint32_t seed) {
    const int f = blockIdx.z;
    uint8_t *base = frames + (size_t)f * frame_stride;
    const size_t ysz = (size_t)w * h;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = ysz * 3 / 2;
    if (i >= total) return;
    int plane, x, y;
    if (i < ysz) {
        plane = 0, x = (int)(i % w), y = (int)(i / w);
    } else {
        size_t j = i - ysz;
        plane = j < ysz / 4 ? 1 : 2;
        if (plane == 2) j -= ysz / 4;
        x = (int)(j % (w / 2)), y = (int)(j / (w / 2));
    }
    base[i] = venc_synth_pixel(plane, x, y, first_frame + f, w, h, seed);
}

/* fills n_frames consecutive I420 frames in device memory with the synthetic source */
extern "C" int venc_synth_device(void *d_frames, size_t frame_stride, int width, int height, int first_frame,
                                 int n_frames, uint32_t seed) {
    if (!d_frames || n_frames < 1) {
        SET_ERR("venc_synth_device: bad arguments");
        return -1;
    }
    const size_t total = (size_t)width * height * 3 / 2;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {
        const int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
        hipLaunchKernelGGL(k_synth, dim3((unsigned)((total + 255) / 256), 1, nf), dim3(256), 0, 0,
                           (uint8_t *)d_frames + (size_t)f0 * frame_stride, frame_stride, width, height, first_frame + f0,
                           seed);
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    return 0;
}

#include "venc_rccl.hip"
