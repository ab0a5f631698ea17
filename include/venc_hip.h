/*
 * venc_hip.h -- thin C-ABI of the MI355X (gfx950) macroblock-loop engine.
 *
 * This is the device-facing half of the drop-in boundary.  The reference's only
 * entry into video coding is x264_encoder_encode() (av_encode.c:970, flush twin
 * av_encode.c:1078) plus open/close (av_encode.c:408, :443); include/x264.h
 * re-declares exactly that surface and csrc/x264_shim.c implements it in C on
 * top of the functions below ("host code stays in C and calls HIP through a
 * thin C-ABI").  Plain pointers and sizes only; no C++ or torch types.
 *
 * Model: an encoder context owns
 *   - an input pool of I420 frame slots in HBM (display size, tight planes), or
 *     borrows a caller-provided device buffer of the same layout;
 *   - up to max_items independent prediction chains ("items": streams or closed
 *     GOPs) that advance in lock-step, one frame per step;
 *   - an output pool in pinned host memory that the pack kernel writes directly.
 * One venc_encode_step() launches, for all items at once (grid z = item),
 *   main chain (stream 1)     P: k_me_pool -> k_me_fix -> k_inter_fused    I: k_intra_fwd -> k_intra_dc -> k_intra_rec
 *   entropy chain (stream 2)  P: k_skip -> k_cavlc_p -> k_scan -> k_pack    I: k_cavlc -> k_scan -> k_pack
 * the entropy chain possibly deferred and launched once for several P steps.  Nothing blocks until venc_wait().
 *
 * All functions return 0 on success, a negative value on error (never abort,
 * never print unless VENC_DEBUG is set) -- the reference's convention for the
 * encoder (av_encode.c:971-974).
 */
#ifndef VENC_HIP_H
#define VENC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VENC_ABI_VERSION 1
#define VENC_MAX_ITEMS 64

typedef struct venc_ctx venc_ctx;

typedef struct {
    int width, height; /* display size, both even */
    int qp;            /* fixed QP 12..51 */
    int n_slices;      /* MB-row bands per frame (clamped to 1..mb_height, <= 256) */
    int max_items;     /* independent chains advanced per step, 1..VENC_MAX_ITEMS */
    int in_slots;      /* input pool size in frames (0: caller will use venc_set_input) */
    int out_slots;     /* output pool size in frames */
    int device;        /* HIP device ordinal */
    int rc_mode;       /* 0: every picture at qp; 1: picture QP chosen on the GPU from the previous picture's motion search
                          (spec/CODEC.md section 1a: the frame-level approximation of libx264's CRF; qp = the rate factor,
                          IDR pictures qp - 3, P pictures qp +- up to 6).  Whole-picture engines only. */
    int intra_modes;   /* 0: Intra16x16 with DC prediction only (the three-kernel I path); 3: as 2 plus Intra4x4 tried against the best 16x16 predictor
                          (nine modes per block by SAD, spec/CODEC.md section 3); 2: as 1 plus plane prediction; 1: vertical / horizontal / DC chosen
                          per macroblock by SAD (spec/CODEC.md section 3; one wavefront kernel, ~3x the I-step time, hidden
                          behind the previous unit's P steps) */
    int deblock;       /* 0: no in-loop filter (disable_deblocking_filter_idc = 1); 1: H.264 8.7 inside every slice (idc 2,
                          spec/CODEC.md section 5a): a slice-wise wavefront kernel after every picture's reconstruction */
    int subpel;        /* 0: integer-sample motion vectors (the search + transform run fused); 1: half-sample refinement; 2: half-
                          then quarter-sample (spec/CODEC.md section 2a: H.264 8.4.2.2.1 interpolation, exhaustive 8-neighbour
                          stages; P steps then run search, refinement and transform as separate kernels).  Whole-picture
                          engines only. */
} venc_config;

/* Per-slice record of an encoded frame (raw RBSP: slice header + slice data +
 * trailing bits, no NAL header, no emulation prevention, no length prefix). */
typedef struct {
    uint32_t byte_off; /* offset inside data */
    uint32_t nbytes;
    uint32_t nbits;
    uint32_t pad;
} venc_slice;

typedef struct {
    const uint8_t *data;      /* pinned host memory; engines with max_items > 1 stage their output in HBM and keep a frame's bytes on the host
                                 until max(8, 4 x max_items) further frames have been fetched -- copy or package them before that */
    const venc_slice *slices; /* n_slices entries */
    uint32_t n_slices;
    uint32_t total_bytes;
    uint32_t error;
} venc_frame;

int venc_abi_version(void);
const char *venc_last_error(void);
int venc_device_count(void);
int venc_device_memory(int device, size_t *free_bytes, size_t *total_bytes); /* 0 = ok */
/* Host threads beside their GPU: restricts the calling thread (scope 0; threads it creates later inherit) or every thread of the
 * process (scope 1) to the CPUs of the card's NUMA node (sysfs: PCI bus id -> numa_node -> cpulist, intersected with the CPUs the
 * caller may use).  Returns the node, or -1 when there was nothing to do.  venc_open() calls it with scope 0 for the thread that opens
 * an engine -- the one that will launch its kernels -- unless VENC_NUMA_BIND=0 (2: scope 1).  On a two-socket host an unbound run
 * keeps whichever socket it started on: 162 - 173 Gpx/s at 1080p against 177 - 183 bound (NOTEBOOK.md section 10).  A launcher that
 * starts one process per GPU does the same with numactl / taskset before the process starts. */
int venc_numa_bind(int device, int scope);

int venc_open(const venc_config *cfg, venc_ctx **out);
void venc_close(venc_ctx *c);

/* geometry */
int venc_mb_width(const venc_ctx *c);
int venc_mb_height(const venc_ctx *c);
int venc_n_slices(const venc_ctx *c);
size_t venc_frame_bytes(const venc_ctx *c); /* w*h*3/2: size of one input slot */

/* Input.  venc_upload copies host planes into pool slot `slot` (async on the
 * copy stream); the source may be reused once venc_upload_sync() returns
 * (venc_upload_sync_slot: once the copy into that slot is done -- callers that
 * keep several pictures in flight).  Both wait for copies only, never for kernels.
 * venc_set_input borrows caller-owned device memory laid out as consecutive
 * I420 frames (frame_stride bytes apart, pointer and stride multiples of 16)
 * instead of the internal pool. */
int venc_upload(venc_ctx *c, int slot, const uint8_t *y, int stride_y, const uint8_t *u, int stride_u,
                const uint8_t *v, int stride_v);
int venc_upload_sync(venc_ctx *c);
int venc_upload_sync_slot(venc_ctx *c, int slot);
int venc_set_copy_streams(venc_ctx *c, int n); /* 1 (default) or 2: consecutive venc_upload calls' copies on two streams in turn */
int venc_set_input(venc_ctx *c, const void *d_frames, size_t frame_stride, int n_frames);
void *venc_input_slot_ptr(venc_ctx *c, int slot); /* device pointer of an internal pool slot */
/* pinned host allocation helpers used by x264_picture_alloc (av_encode.c:415) */
void *venc_host_alloc(size_t bytes);
void venc_host_free(void *p);

/* One lock-step frame for n_items chains.  in_slots[i] / out_slots[i]: input
 * frame slot and output slot of item i.  is_idr: all items code an IDR picture
 * (no reference), else a P picture predicted from the item's previous
 * reconstruction.  idr_ids[i]: idr_pic_id (used when is_idr). */
int venc_encode_step(venc_ctx *c, int n_items, const int *in_slots, const int *out_slots, int is_idr,
                     int frame_num, const int *idr_ids);

/* ---- Slice-parallel multi-GPU mode (BASELINE config 5) ----
 * One context per GPU codes a band of the picture: `slice_count` consecutive slices starting at
 * `slice_first` (0, 0 = the whole picture again).  Slices never predict across their borders, so a
 * band's bitstream is independent of the other bands; only the motion search and the motion
 * compensation of the NEXT P picture read reconstructed rows of the neighbouring bands (+-16 luma
 * rows).  After a step that is followed by a P picture the host therefore runs one exchange:
 *   venc_band_export()  packs this context's band of the reconstruction just written (all rows, or
 *                       with halo_rows = 16 only its first and last 16 luma / 8 chroma rows) into a
 *                       contiguous device chunk, [item][Y rows | U rows | V rows];
 *   RCCL all-gather / broadcast of the chunks (torch.distributed in video-encoder_amd/dist.py,
 *                       on the stream returned by venc_stream_handle());
 *   venc_band_import()  unpacks a peer's chunk (the band `slice_first, slice_count` it codes) into
 *                       this context's reconstruction.
 * Both are enqueued on the context's main stream: ordered after the step's transform kernels and
 * before the next step's motion search, no host synchronisation.  The output slot's slice table
 * keeps n_slices entries; slices outside the band have nbytes = 0.
 * venc_band_export/import return the chunk bytes per item (> 0) or a negative error. */
int venc_set_band(venc_ctx *c, int slice_first, int slice_count);
int venc_band_rows(const venc_ctx *c, int slice_first, int slice_count, int *mb_row0, int *mb_row1);
size_t venc_band_chunk_bytes(const venc_ctx *c, int mb_rows, int halo_rows);
int venc_band_export(venc_ctx *c, int n_items, void *d_chunk, size_t item_stride, int halo_rows);
int venc_band_import(venc_ctx *c, int n_items, const void *d_chunk, size_t item_stride, int slice_first, int slice_count,
                     int halo_rows);
void *venc_stream_handle(venc_ctx *c); /* hipStream_t of the main kernel chain */
/* The exchange off the main chain.  Between venc_band_xbegin() and venc_band_xend() the export, the transfer (on venc_band_stream())
 * and the imports are enqueued on the context's EXCHANGE stream, behind everything the main stream has been given so far.  The next P
 * step then launches its motion search twice: the tile rows whose search windows stay inside the own band at once -- beside the
 * exchange --, the band's first / last tile row (the only ones that read a neighbour's rows) once the exchange has finished;
 * whatever else reads the reconstruction follows those on the main stream.  Without the pair (or with VENC_XOVERLAP=0) everything
 * stays on the main stream as before.  venc_band_exchange_rccl() uses the pair itself.  venc_band_split_steps(): P steps whose
 * search ran as two launches (tests, benchmarks).  Unmeasured on hardware with more than one GPU. */
int venc_band_xbegin(venc_ctx *c);
int venc_band_xend(venc_ctx *c);
void *venc_band_stream(venc_ctx *c); /* hipStream_t band copies and the transfer belong on right now */
long long venc_band_split_steps(const venc_ctx *c);

/* ---- RCCL exchange: the three steps above as ONE call, in C, on RCCL over xGMI (csrc/venc_rccl.hip) ----
 * One communicator per context.  One process per GPU: rank 0 calls venc_rccl_unique_id, any out-of-band channel carries
 * the 128 bytes to the peers, every rank calls venc_rccl_init_rank(ctx, id, world, rank).  One process driving all GPUs
 * (a C host shaped like the reference, av_encode.c:840): venc_rccl_init_all(ctxs, n) = ncclCommInitAll.  Both set the
 * context's band: the n_slices slices are dealt to the ranks in contiguous runs, earlier ranks take the remainder.
 * venc_band_exchange_rccl(ctx, n_items, halo_rows) is then called after every step that a P picture follows:
 * halo_rows = 16 sends the band's two 16-row halos to the two neighbours (grouped ncclSend / ncclRecv), halo_rows = 0
 * all-gathers whole bands; everything is enqueued on the main stream, nothing blocks.  A single process calls it for each
 * of its contexts between venc_rccl_group_begin() and venc_rccl_group_end(): inside that group RCCL only records the
 * transfers, so each such call leaves the unpacking of the peers' rows pending and venc_rccl_group_end() enqueues it,
 * behind the collective, on every context's main stream -- the next step may be launched once it has returned.
 * venc_rccl_pending() = exchanges of the calling thread whose unpacking is still pending.  venc_rccl_agree(ctx, flag) is a
 * host-synchronous MAX of `flag` over the ranks (use it at the end of a pass: a rank that failed makes all ranks stop).
 * RCCL is loaded at run time (librccl.so.1; VENC_RCCL_LIB overrides): engines that never call these do not load it. */
#define VENC_RCCL_ID_BYTES 128
int venc_rccl_unique_id(uint8_t id[VENC_RCCL_ID_BYTES]);
int venc_rccl_init_rank(venc_ctx *c, const uint8_t id[VENC_RCCL_ID_BYTES], int world, int rank);
int venc_rccl_init_all(venc_ctx **ctxs, int n);
int venc_rccl_world(const venc_ctx *c);
int venc_rccl_rank(const venc_ctx *c);
int venc_rccl_comm_count(const venc_ctx *c); /* ncclCommCount of the context's communicator (0: none) */
int venc_band_exchange_rccl(venc_ctx *c, int n_items, int halo_rows);
int venc_rccl_group_begin(void);
int venc_rccl_group_end(void);
int venc_rccl_agree(venc_ctx *c, int local_flag);
long long venc_rccl_exchanges(const venc_ctx *c);
int venc_rccl_pending(void);
void venc_rccl_always(venc_ctx *c, int on); /* tests: run the collective even in a world of one */
void venc_rccl_poison(venc_ctx *c, int on); /* tests: fill the receive buffer with 0xA5 before every collective, so that rows unpacked too early show */

/* Completion tracking: venc_mark() returns a ticket for everything issued so
 * far; venc_wait(ticket) blocks the host until that point.  The engine may defer
 * the entropy chain of up to four consecutive P steps and launch it once for all
 * of them (one launch covers 4 x max_items pictures); venc_wait on a ticket of a
 * deferred step launches it first, venc_flush() does so without waiting. */
int venc_mark(venc_ctx *c);
int venc_wait(venc_ctx *c, int ticket);
int venc_flush(venc_ctx *c);
/* For a second host thread (packaging beside launching): venc_bind_thread once on that thread; venc_wait_issued blocks
 * until a ticket completes, provided venc_flush (or venc_wait) ran after the ticket was issued.  Both are safe beside
 * venc_encode_step / venc_mark / venc_flush on another thread; every other function belongs to one thread at a time. */
int venc_bind_thread(venc_ctx *c);
int venc_wait_issued(venc_ctx *c, int ticket);
int venc_sync(venc_ctx *c);
/* 1 (default): the entropy kernels of a step (k_cavlc, k_scan, k_pack) run on a second HIP
 * stream beside the next step's motion/transform kernels; 0: one stream, no concurrency. */
int venc_set_overlap(venc_ctx *c, int enable);

/* Output of out slot `slot` (valid after the producing step completed).  Engines with max_items > 1 stage the encoded
 * bytes in HBM; venc_get_frame brings a slot's bytes to pinned host memory when nobody has yet (one DMA copy and a
 * wait per slot), venc_fetch does it for the `count` slots first_slot, first_slot + slot_step, ... with a single strided
 * copy -- the slots of one lock-step frame of `count` GOPs, say. */
int venc_get_frame(venc_ctx *c, int slot, venc_frame *out);
int venc_fetch(venc_ctx *c, int first_slot, int slot_step, int count);

/* Introspection for kernel-level parity tests: copies device intermediates of
 * chain `item` (state after the last step) to host memory.
 *   what 0: mvsad   u32[nmb]      sad<<16 | (dy&255)<<8 | (dx&255)
 *        1: levels  i16[nmb][27][16]
 *        2: mbinfo  u8[nmb][32]   nnz[24], cbp, flags, pad
 *        3: mbnbits u32[nmb]
 *        4/5/6: reconstructed Y/U/V plane (coded size) of the last step
 *        7: mboff   u32[nmb]
 *        8: packed 8x8 block sums of the last reconstruction, u32[ch+25][(cw+32) & ~15]
 *           = S8(x-16,y-16) | S8(x-8,y-16) << 16 (motion-search lower bounds)
 * Level record per macroblock (27 blocks x 16 levels, scan order):
 *   0..15 luma 4x4 blocks in luma4x4BlkIdx order; 16 Intra16x16 luma DC;
 *   17/18 Cb/Cr DC (4 levels); 19..22 Cb AC; 23..26 Cr AC (levels at 1..15). */
int venc_debug_read(venc_ctx *c, int what, int item, void *dst, size_t dst_bytes);

/* Per-kernel timing with HIP events on the engine's own stream (bench.py's
 * roofline leg).  venc_profile(c, 1, capacity) starts logging one event pair per
 * kernel launch; venc_profile_read() synchronises, returns per-kernel summed
 * milliseconds and launch counts (arrays of venc_profile_kernels() entries,
 * names via venc_profile_name) and clears the log.  Returns 1 if the log
 * overflowed its capacity. */
int venc_profile(venc_ctx *c, int enable, int capacity);
/* restrict the event pairs to the kernels whose id bit is set (default: all) and to every stride-th
 * launch of each (default: 1); every pair costs the stream a few microseconds, so the timed region of
 * bench.py samples the dominant kernel only */
int venc_profile_select(venc_ctx *c, unsigned kernel_mask, int stride);
/* launches the motion search of a full P step goes as: 1, or 2 with two chains (VENC_CHAINS=2: half the items per launch, the
 * second half on a stream of its own) -- what one timed "k_me" launch covers */
int venc_search_split(const venc_ctx *c);
int venc_profile_kernels(void);
const char *venc_profile_name(int kernel_id);
int venc_profile_read(venc_ctx *c, double *ms_sum, int *launches);

/* Deterministic synthetic YUV420p source, host side. */
void venc_synth_frame(int width, int height, int frame_index, uint32_t seed, uint8_t *y, uint8_t *u, uint8_t *v);
/* same source, generated on the GPU into n_frames consecutive I420 frames in device memory */
int venc_synth_device(void *d_frames, size_t frame_stride, int width, int height, int first_frame, int n_frames,
                      uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
