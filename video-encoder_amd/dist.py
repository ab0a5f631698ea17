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
