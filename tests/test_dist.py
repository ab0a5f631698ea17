"""Multi-rank path on CPU: world_size-2 gloo run of the GOP-sharding logic
(video-encoder_amd/dist.py).  Each rank encodes its contiguous range of closed
GOPs -- here with the CPU oracle standing in for the GPU encoder, which has the
same gop_base contract -- and the rank-ordered concatenation must equal the
single-encoder stream byte for byte."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gop_shard_partition():
    d = importlib.import_module("video-encoder_amd.dist")
    for n_frames, keyint, world in ((300, 30, 8), (300, 30, 3), (95, 30, 2), (10, 30, 4), (300, 30, 1)):
        cover = []
        for r in range(world):
            f0, n, g0 = d.gop_shard(n_frames, keyint, world, r)
            assert f0 == min(g0 * keyint, n_frames) and n >= 0
            cover += list(range(f0, f0 + n))
        assert cover == list(range(n_frames))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import orc_py
    from conftest import noise_frames
    d = importlib.import_module("video-encoder_amd.dist")
    d.init_from_env("gloo")
    w, h, n_frames, keyint = 64, 48, 14, 4
    clip = noise_frames(w, h, n_frames, seed=11)
    f0, n, g0 = d.gop_shard(n_frames, keyint, world, rank)
    enc = orc_py.Encoder(w, h, qp=30, n_slices=2, keyint=keyint, idr_base=g0)
    mine = b"".join(enc.encode(*clip[f]).data for f in range(f0, f0 + n))
    whole = d.gather_in_rank_order(mine)
    slowest = d.max_over_ranks(1.0 + rank)
    if rank == 0:
        ref = orc_py.Encoder(w, h, qp=30, n_slices=2, keyint=keyint)
        single = b"".join(ref.encode(*clip[f]).data for f in range(n_frames))
        q.put((whole == single, slowest, len(whole)))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gop_sharding_reproduces_the_single_stream():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, slowest, n = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same and n > 1000
    assert slowest == 2.0
