/*
 * venc_rccl.hip -- the slice-parallel exchange in C on RCCL (BASELINE config 5), part of the C-ABI
 * (include/venc_hip.h, "RCCL exchange"), so that a C host -- which is what the reference is (av_encode.c:840: one C
 * program, one thread, av_encode.c:940-1070) -- can run the band mode without Python in the per-picture path.
 *
 * One communicator per context (one context per GPU):
 *   venc_rccl_init_rank   one process per GPU: rank 0 makes a unique id (venc_rccl_unique_id), any out-of-band channel
 *                         carries its 128 bytes to the peers (torch.distributed in bench.py, a file in a C launcher)
 *   venc_rccl_init_all    one process driving all GPUs: ncclCommInitAll over the contexts' devices
 * Per picture that a P picture follows, on the engine's main stream, no host synchronisation:
 *   venc_band_exchange_rccl   pack own band (k_band_copy) -> halo mode: ONE grouped ncclSend + ncclRecv per neighbour
 *                             (the chunk = first and last 16 luma / 8 chroma rows of the band, 184 KB at 2160p; xGMI is
 *                             point-to-point, the neighbours' links carry it; the cost is latency, not bandwidth)
 *                             whole-band mode: ncclAllGather of equal-size chunks (every rank ends with the whole
 *                             reference picture: the "broadcast of reconstructed frames" of the north star)
 *                             -> unpack the peers' chunks into the reconstruction.
 *   venc_rccl_agree           host-synchronous MAX over the ranks of an error flag: called at coarse points (end of a
 *                             pass), so that a rank that failed makes every rank stop instead of leaving its peers in
 *                             a collective.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1, RTLD_LOCAL): the engine has no link-time dependency on it -- a
 * single-GPU encoder never loads a 570 MB library -- and a host process that carries another RCCL build (PyTorch's
 * wheel bundles its own) cannot capture these calls by symbol interposition: the streams and pointers handed over
 * below belong to the HIP runtime this library is linked to, and so must the RCCL that receives them.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>

struct venc_rccl_api {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static venc_rccl_api g_nccl;

static int rccl_load(void) {
    if (g_nccl.lib) return 0;
    const char *names[] = {getenv("VENC_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void *h = NULL;
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !h; i++)
        if (names[i] && names[i][0]) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        SET_ERR("venc_rccl: cannot load librccl.so.1 (%s)", dlerror());
        return -1;
    }
#define RCCL_SYM(field, name)                                   \
    do {                                                        \
        *(void **)(&g_nccl.field) = dlsym(h, name);             \
        if (!g_nccl.field) {                                    \
            SET_ERR("venc_rccl: %s missing in librccl", name);  \
            dlclose(h);                                         \
            return -1;                                          \
        }                                                       \
    } while (0)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(Send, "ncclSend");
    RCCL_SYM(Recv, "ncclRecv");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    g_nccl.lib = h;
    return 0;
}

#define NCCL_OK(call)                                                                       \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            SET_ERR("%s failed: %s (%s:%d)", #call, g_nccl.GetErrorString(r_), __FILE__, __LINE__); \
            if (getenv("VENC_DEBUG")) fprintf(stderr, "venc: %s\n", g_nccl.GetErrorString(r_)); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

/* slices of `rank` when n_slices are dealt to `world` ranks: contiguous, earlier ranks take the remainder
 * (the rule of video-encoder_amd/dist.py slice_band) */
static void rccl_band_of(int n_slices, int world, int rank, int *first, int *count) {
    const int base = n_slices / world, rem = n_slices % world;
    *first = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
}

extern "C" int venc_rccl_unique_id(uint8_t id[VENC_RCCL_ID_BYTES]) {
    static_assert(VENC_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id || rccl_load() != 0) return -1;
    ncclUniqueId u;
    NCCL_OK(g_nccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

/* exchange buffers: own chunk + one per rank (all-gather layout: rank r's chunk at r * rank_bytes) */
static int rccl_buffers(venc_ctx *c, int n_items, int halo_rows) {
    int biggest = 0;
    for (int r = 0; r < c->rc_world; r++) {
        int f, n;
        rccl_band_of(c->g.n_slices, c->rc_world, r, &f, &n);
        const int rows = c->slice_rows[f + n] - c->slice_rows[f];
        if (rows > biggest) biggest = rows;
    }
    const size_t chunk = (venc_band_chunk_bytes(c, biggest, halo_rows) + 255) & ~(size_t)255;
    if (chunk == 0) {
        SET_ERR("venc_rccl: bad halo_rows %d", halo_rows);
        return -1;
    }
    if (c->rc_send && chunk == c->rc_item_stride && n_items <= c->rc_items && halo_rows == c->rc_halo) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipStreamSynchronize(c->stream_x));
    (void)hipFree(c->rc_send);
    (void)hipFree(c->rc_recv);
    c->rc_send = c->rc_recv = NULL;
    c->rc_item_stride = chunk;
    c->rc_items = n_items > c->cfg.max_items ? n_items : c->cfg.max_items;
    c->rc_halo = halo_rows;
    HIP_OK(hipMalloc(&c->rc_send, chunk * c->rc_items));
    HIP_OK(hipMalloc(&c->rc_recv, chunk * c->rc_items * c->rc_world));
    HIP_OK(hipMemset(c->rc_send, 0, chunk * c->rc_items));
    return 0;
}

/* checked BEFORE a communicator is created: ncclCommInitRank blocks until all ranks of the world have called it */
static int rccl_world_fits(const venc_ctx *c, int world) {
    if (world > c->g.n_slices) {
        SET_ERR("venc_rccl: %d ranks but only %d slices (the slice count is a stream constant, not the GPU count)", world, c->g.n_slices);
        return -1;
    }
    return 0;
}

static int rccl_attach(venc_ctx *c, ncclComm_t comm, int world, int rank) {
    c->rc_comm = comm;
    c->rc_world = world;
    c->rc_rank = rank;
    int f, n;
    rccl_band_of(c->g.n_slices, world, rank, &f, &n);
    return venc_set_band(c, f, n);
}

extern "C" int venc_rccl_init_rank(venc_ctx *c, const uint8_t id[VENC_RCCL_ID_BYTES], int world, int rank) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world || c->rc_comm) {
        SET_ERR("venc_rccl_init_rank: bad arguments (world %d, rank %d)", world, rank);
        return -1;
    }
    if (rccl_world_fits(c, world) != 0 || rccl_load() != 0) return -1;
    HIP_OK(hipSetDevice(c->cfg.device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    NCCL_OK(g_nccl.CommInitRank(&comm, world, u, rank));
    return rccl_attach(c, comm, world, rank);
}

extern "C" int venc_rccl_init_all(venc_ctx **ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) {
        SET_ERR("venc_rccl_init_all: bad arguments");
        return -1;
    }
    if (rccl_load() != 0) return -1;
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || ctxs[i]->rc_comm) {
            SET_ERR("venc_rccl_init_all: context %d is null or already has a communicator", i);
            return -1;
        }
        if (rccl_world_fits(ctxs[i], n) != 0) return -1;
        devs[i] = ctxs[i]->cfg.device;
    }
    NCCL_OK(g_nccl.CommInitAll(comms, n, devs));
    for (int i = 0; i < n; i++)
        if (rccl_attach(ctxs[i], comms[i], n, i) != 0) return -1;
    return 0;
}

/* the communicator's own count of its ranks (ncclCommCount): what a benchmark quotes as evidence that the exchange ran
 * on a world of that size */
extern "C" int venc_rccl_comm_count(const venc_ctx *c) {
    if (!c || !c->rc_comm) return 0;
    int n = 0;
    NCCL_OK(g_nccl.CommCount(c->rc_comm, &n));
    return n;
}
extern "C" int venc_rccl_world(const venc_ctx *c) { return c && c->rc_comm ? c->rc_world : 0; }
extern "C" int venc_rccl_rank(const venc_ctx *c) { return c && c->rc_comm ? c->rc_rank : -1; }

/* Single-process hosts (venc_rccl_init_all) bracket the per-context exchange calls of one picture with
 * venc_rccl_group_begin / venc_rccl_group_end.  Inside that outer group RCCL only RECORDS the sends, receives and
 * all-gathers; they reach the contexts' streams at the outermost ncclGroupEnd.  Unpacking the peers' chunks must follow
 * them in stream order, so an exchange called inside the group leaves its import half pending and
 * venc_rccl_group_end runs it after the group has been submitted (round 2 launched the unpack kernels at once: stream
 * order was export, unpack of the PREVIOUS picture's chunks, collective).  The group is a per-thread notion in RCCL,
 * and so is this list. */
static thread_local int g_group_depth = 0;
static thread_local struct { venc_ctx *c; int n_items, halo_rows; } g_pending[64];
static thread_local int g_n_pending = 0;

static int rccl_import_peers(venc_ctx *c, int n_items, int halo_rows) {
    const size_t stride = c->rc_item_stride, rank_bytes = stride * (size_t)c->rc_items;
    uint8_t *recv = (uint8_t *)c->rc_recv;
    const int lo = c->rc_rank - 1, hi = c->rc_rank + 1;
    HIP_OK(hipSetDevice(c->cfg.device));
    for (int peer = 0; peer < c->rc_world; peer++) {
        if (peer == c->rc_rank || (halo_rows > 0 && peer != lo && peer != hi)) continue;
        int f, n;
        rccl_band_of(c->g.n_slices, c->rc_world, peer, &f, &n);
        if (venc_band_import(c, n_items, recv + (size_t)peer * rank_bytes, stride, f, n, halo_rows) < 0) return -1;
    }
    return 0;
}

/* one exchange of the reconstruction just written, on the main stream */
extern "C" int venc_band_exchange_rccl(venc_ctx *c, int n_items, int halo_rows) {
    if (!c || !c->rc_comm) {
        SET_ERR("venc_band_exchange_rccl: no communicator (venc_rccl_init_rank / venc_rccl_init_all first)");
        return -1;
    }
    if (c->rc_world == 1 && !c->rc_always) return 0; /* the band is the picture */
    HIP_OK(hipSetDevice(c->cfg.device));
    if (rccl_buffers(c, n_items, halo_rows) != 0) return -1;
    const size_t stride = c->rc_item_stride, rank_bytes = stride * (size_t)c->rc_items;
    uint8_t *send = (uint8_t *)c->rc_send, *recv = (uint8_t *)c->rc_recv;
    /* pack, collective and unpack on the exchange stream (venc_hip.hip EXCHANGE STREAM): the next P step's search starts on the main stream
     * with the tile rows that read no neighbour's row and takes the exchange in before the band's first / last tile row */
    if (venc_band_xbegin(c) != 0) return -1;
    const hipStream_t xs = (hipStream_t)venc_band_stream(c);
    if (venc_band_export(c, n_items, send, stride, halo_rows) < 0) {
        (void)venc_band_xend(c);
        return -1;
    }
    if (c->rc_poison) /* tests: a chunk consumed before its collective has run shows up in the reconstruction */
        HIP_OK(hipMemsetAsync(recv, 0xA5, rank_bytes * (size_t)c->rc_world, xs));
    const int lo = c->rc_rank - 1, hi = c->rc_rank + 1;
    if (halo_rows > 0) {
        NCCL_OK(g_nccl.GroupStart());
        for (int k = 0; k < 2; k++) {
            const int peer = k ? hi : lo;
            if (peer < 0 || peer >= c->rc_world) continue;
            NCCL_OK(g_nccl.Send(send, stride * (size_t)n_items, ncclUint8, peer, c->rc_comm, xs));
            NCCL_OK(g_nccl.Recv(recv + (size_t)peer * rank_bytes, stride * (size_t)n_items, ncclUint8, peer, c->rc_comm, xs));
        }
        NCCL_OK(g_nccl.GroupEnd());
    } else {
        NCCL_OK(g_nccl.AllGather(send, recv, rank_bytes, ncclUint8, c->rc_comm, xs));
    }
    c->rc_exchanges++;
    if (g_group_depth > 0) { /* the collective is only recorded so far: unpack at venc_rccl_group_end */
        if (g_n_pending >= 64) {
            SET_ERR("venc_band_exchange_rccl: more than 64 exchanges inside one venc_rccl_group_begin / _end");
            return -1;
        }
        g_pending[g_n_pending].c = c;
        g_pending[g_n_pending].n_items = n_items;
        g_pending[g_n_pending].halo_rows = halo_rows;
        g_n_pending++;
        return 0;
    }
    if (rccl_import_peers(c, n_items, halo_rows) != 0) return -1;
    return venc_band_xend(c);
}

/* single-process multi-GPU hosts bracket the per-context exchange calls of one picture with these; the peers' rows are
 * in place on every context's main stream once venc_rccl_group_end has returned */
extern "C" int venc_rccl_group_begin(void) {
    if (rccl_load() != 0) return -1;
    NCCL_OK(g_nccl.GroupStart());
    g_group_depth++;
    return 0;
}
extern "C" int venc_rccl_group_end(void) {
    if (rccl_load() != 0) return -1;
    if (g_group_depth <= 0) {
        SET_ERR("venc_rccl_group_end without venc_rccl_group_begin");
        return -1;
    }
    const ncclResult_t r = g_nccl.GroupEnd();
    if (--g_group_depth > 0 && r == ncclSuccess) return 0; /* an inner pair: nothing has been submitted yet */
    const int n = g_n_pending;
    g_n_pending = 0;
    if (r != ncclSuccess) {
        SET_ERR("ncclGroupEnd failed: %s", g_nccl.GetErrorString(r));
        return -1;
    }
    int rc = 0;
    for (int i = 0; i < n; i++)
        if (rccl_import_peers(g_pending[i].c, g_pending[i].n_items, g_pending[i].halo_rows) != 0 || venc_band_xend(g_pending[i].c) != 0) rc = -1;
    return rc;
}

extern "C" int venc_rccl_pending(void) { return g_n_pending; }

extern "C" int venc_rccl_agree(venc_ctx *c, int local_flag) {
    if (!c || !c->rc_comm) {
        SET_ERR("venc_rccl_agree: no communicator");
        return -1;
    }
    HIP_OK(hipSetDevice(c->cfg.device));
    if (!c->rc_flag) HIP_OK(hipMalloc(&c->rc_flag, 2 * sizeof(int)));
    int v = local_flag;
    HIP_OK(hipMemcpyAsync(c->rc_flag, &v, sizeof(int), hipMemcpyHostToDevice, c->stream));
    NCCL_OK(g_nccl.AllReduce(c->rc_flag, c->rc_flag + 1, 1, ncclInt32, ncclMax, c->rc_comm, c->stream));
    HIP_OK(hipMemcpyAsync(&v, c->rc_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return v;
}

extern "C" long long venc_rccl_exchanges(const venc_ctx *c) { return c ? c->rc_exchanges : 0; }
/* tests: run export, the collective and the (empty) import loop even in a world of one */
extern "C" void venc_rccl_always(venc_ctx *c, int on) {
    if (c) c->rc_always = on;
}
/* tests: fill the receive buffer with 0xA5 on the main stream before every collective */
extern "C" void venc_rccl_poison(venc_ctx *c, int on) {
    if (c) c->rc_poison = on;
}

static void rccl_release(venc_ctx *c) {
    if (c->rc_comm && g_nccl.lib) (void)g_nccl.CommDestroy(c->rc_comm);
    c->rc_comm = NULL;
    (void)hipFree(c->rc_send);
    (void)hipFree(c->rc_recv);
    (void)hipFree(c->rc_flag);
    c->rc_send = c->rc_recv = NULL;
    c->rc_flag = NULL;
}
