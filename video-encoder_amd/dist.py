"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in CPU tests).

The path shards without a data-path collective in two ways (SURVEY.md section
8e): independent streams (BASELINE config 4, what bench.py --gpus N runs) and
closed GOPs of one stream.  Closed GOPs never reference each other, so rank r
encodes a contiguous range of GOPs with its encoder's gop_base set to the index
of its first GOP (idr_pic_id parity) and the per-rank bitstreams concatenate, in
rank order, to exactly the single-encoder stream.  Collectives are used only for
control: barrier, max-over-ranks timing, and gathering the encoded bytes.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend="nccl", device=None):
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def gop_shard(n_frames, keyint, world, rank):
    """(first_frame, n_frames, gop_base) of `rank`: contiguous GOP ranges, earlier ranks take the remainder."""
    n_gops = (n_frames + keyint - 1) // keyint
    base, rem = divmod(n_gops, world)
    g0 = rank * base + min(rank, rem)
    g1 = g0 + base + (1 if rank < rem else 0)
    f0, f1 = min(g0 * keyint, n_frames), min(g1 * keyint, n_frames)
    return f0, f1 - f0, g0


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_in_rank_order(payload):
    """Concatenate every rank's bytes in rank order on all ranks (control-plane only)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bytes(payload)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, bytes(payload))
    return b"".join(parts)


# ----------------------------------------------------------------------------- slice-parallel mode
# BASELINE config 5 / SURVEY.md section 8e: ONE stream, every picture cut into the same n_slices
# MB-row bands whatever the GPU count, rank r codes a contiguous run of slices.  The only data the
# ranks owe each other is the reconstruction of their bands before the next P picture (its motion
# search and compensation read +-16 luma rows across band borders): one exchange per picture.

def slice_band(n_slices, world, rank):
    """(slice_first, slice_count) of `rank`: contiguous slices, earlier ranks take the remainder."""
    if world > n_slices:
        raise ValueError("slice-parallel mode needs n_slices (%d) >= ranks (%d): the slice count is a "
                         "stream constant, not the GPU count" % (n_slices, world))
    base, rem = divmod(n_slices, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


class EnginePort:
    """Chunk access for BandExchange on a GPU engine: device chunks, pack/unpack kernels on the engine's
    main stream (include/venc_hip.h venc_band_export / venc_band_import)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self._stream = None
        self._stream_handle = None

    def alloc(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)

    def rows_of(self, engine, slice_first, slice_count):
        return engine.band_rows(slice_first, slice_count)

    def chunk_bytes(self, engine, mb_rows, halo_rows):
        return engine.band_chunk_bytes(mb_rows, halo_rows)

    def export(self, engine, n_items, chunk, item_stride, halo_rows):
        engine.band_export(n_items, chunk.data_ptr(), item_stride, halo_rows)

    def import_(self, engine, n_items, chunk, item_stride, slice_first, slice_count, halo_rows):
        engine.band_import(n_items, chunk.data_ptr(), item_stride, slice_first, slice_count, halo_rows)

    def xbegin(self, engine):
        """export / import until xend() go to the engine's exchange stream; the next P step launches the tile rows that read no
        neighbour's row beside them and the band's edge rows behind them (include/venc_hip.h venc_band_xbegin)"""
        engine.band_xbegin()

    def xend(self, engine):
        engine.band_xend()

    def stream(self, engine):
        """The stream band copies belong on right now (the exchange stream between xbegin and xend, else the main stream) as a
        torch stream: collectives issued under it are ordered with the engine's kernels (ProcessGroupNCCL syncs its own stream
        with the current one on both sides)."""
        h = engine.band_stream()
        if self._stream is None or self._stream_handle != h:
            self._stream = self.torch.cuda.ExternalStream(h, device=self.device)
            self._stream_handle = h
        return self.torch.cuda.stream(self._stream)

    def sync(self, engine):
        engine.sync()


class BandExchange:
    """Callable handed to the encoder as its exchange hook (x264_param_t.hip.pf_exchange):
    export own band -> all-gather of equal-size chunks -> import the peers' bands.

    halo_rows = 0 ships whole bands (every rank ends up with the full reference picture, the
    "broadcast of reconstructed reference frames" of the north star); halo_rows = 16 ships only the
    first and last 16 luma / 8 chroma rows of each band and imports just the two neighbours' --
    all the next P picture can read.  Backend "nccl" (RCCL) gathers device chunks on the engine's
    stream with no host synchronisation; any other backend (gloo rehearsals) stages through host
    memory."""

    def __init__(self, port, n_slices, world, rank, halo_rows=0, group=None):
        self.port, self.n_slices, self.world, self.rank = port, n_slices, world, rank
        self.halo_rows, self.group = halo_rows, group
        self.bands = [slice_band(n_slices, world, r) for r in range(world)]
        self.exchanges = 0
        self.always = False  # tests: run export + gather even in a world of one
        self._send = self._recv = None
        self._stride = self._items = 0

    def peers(self):
        if self.halo_rows:
            return [r for r in (self.rank - 1, self.rank + 1) if 0 <= r < self.world]
        return [r for r in range(self.world) if r != self.rank]

    def _buffers(self, engine, n_items):
        if self._send is None or n_items > self._items:
            rows = [self.port.rows_of(engine, *b) for b in self.bands]
            biggest = max(r1 - r0 for r0, r1 in rows)
            self._stride = (self.port.chunk_bytes(engine, biggest, self.halo_rows) + 255) & ~255
            self._items = n_items
            self._send = self.port.alloc(self._stride * n_items)
            self._recv = self.port.alloc(self._stride * n_items * self.world)
        return self._send, self._recv

    def __call__(self, engine, n_items):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not self.always:
            return
        send, recv = self._buffers(engine, n_items)
        per_rank = self._stride * self._items
        xbegin, xend = getattr(self.port, "xbegin", None), getattr(self.port, "xend", None)
        if xbegin:
            xbegin(engine)
        self.port.export(engine, n_items, send, self._stride, self.halo_rows)
        on_device = send.is_cuda and dist.get_backend(self.group) == "nccl"
        if on_device:
            with self.port.stream(engine):
                dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            self.port.sync(engine)
            parts = [torch.empty(per_rank, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, send.cpu(), group=self.group)
            recv.copy_(torch.cat(parts))
            if recv.is_cuda:
                torch.cuda.synchronize(recv.device)  # the copy ran on torch's stream, the import runs on the engine's
        for r in self.peers():
            s0, cnt = self.bands[r]
            self.port.import_(engine, n_items, recv[r * per_rank:(r + 1) * per_rank], self._stride, s0, cnt,
                              self.halo_rows)
        if xend:
            xend(engine)
        self.exchanges += 1


def interleave_bands(per_rank_frames):
    """per_rank_frames[r][f] = payload bytes of frame f from band owner r (band order = rank order);
    returns the per-frame payloads of the whole picture."""
    n = len(per_rank_frames[0])
    assert all(len(p) == n for p in per_rank_frames)
    return [b"".join(p[f] for p in per_rank_frames) for f in range(n)]


def gather_frames_to_root(frames, group=None, root=0):
    """Every rank's per-frame payload list on `root` only (None elsewhere): the bitstream is assembled where it is
    written out, the other ranks receive nothing."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [list(frames)]
    me = dist.get_rank(group)
    parts = [None] * dist.get_world_size(group) if me == root else None
    dist.gather_object([bytes(f) for f in frames], parts, dst=root, group=group)
    return parts


def gather_frames(frames, group=None):
    """All ranks' per-frame payload lists on rank 0 (control plane: bitstream bytes, not pixels)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [list(frames)]
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, [bytes(f) for f in frames], group=group)
    return parts
