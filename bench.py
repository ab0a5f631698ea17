#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s encoded at fixed QP on synthetic YUV420p.

One "step" = one pass of the hot path over the whole clip of BASELINE.json
config 2 (1920x1080, 300 frames, IPPP closed GOP 30, QP 26, 8 slices, 16x16
full search +-16), with the frames already resident in HBM, through the
reference's own entry points (include/x264.h; the resident-frames extension
x264_venc_encode_resident drives the same open/encode/flush path as
av_encode.c:970/:1076-1083 and hands every frame's NAL array to a sink the way
the reference hands it to its muxer).  Timed region = kernels + write-out of the
slice RBSPs into pinned host memory + host-side NAL packaging (emulation
prevention, length prefixes) of every frame.

The metric names "1080p & 4K" and "bit-exact vs CPU ref", so the ONE JSON line of the default run carries, besides the
1080p headline:
  "2160p"           the same measurement on BASELINE config 3 (value, ms_per_step, roofline)
  "bit_exact"       SHA-256 of the HIP bitstream of one untimed pass against the SHA-256 of the CPU oracle's bitstream
                    of the same frames (all 300 at both sizes); a mismatch exits non-zero
  "reference_defaults"  what the reference's own open sequence delivers: preset medium / film, X264_RC_CRF, every preset tool, one
                    pic_in from host memory per call (av_encode.c:384-401, :543-547, :970); summary also in config
  "pcie_inclusive"  host frames (pinned memory) through x264_encoder_encode -- the call the reference makes
                    (av_encode.c:970), with its H2D copy per frame -- never `value`
  "cpu_baseline"    the CPU oracle on the box's host cores: all cores (<= 16 OpenMP threads) on the whole clip, and a
                    one-thread line on one GOP; `cores` = threads used, `host_cores` = os.cpu_count()
  "roofline"        dominant kernel (k_me) against the HBM-read roofline, from HIP events recorded on the engine's
                    stream inside the timed region
The CPU oracle is this repository's own C restatement (oracle/): NOT libx264; parity with libx264 is unpinned.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), each rank
encodes its own clip (independent streams / closed GOPs shard with no data-path
collective -- SURVEY.md section 8e, BASELINE config 4); value = pixels of all
ranks / max-over-ranks time; scaling = "weak".

--gpus N > 1 without RANK / WORLD_SIZE in the environment: bench.py starts the N rank processes itself (before anything in
the parent touches a GPU) and relays rank 0's line; it never reports n_gpus 1 for a --gpus N request.  An N > 1 streams
run also carries "slices_2160p": BASELINE config 5 (strong scaling) measured by a second set of rank processes that every
rank starts BEFORE it touches its GPU and waits for under a time limit -- a fault or hang of the RCCL exchange, which
has never met a second rank on hardware, cannot take the config-4 number with it.

--mode slices (BASELINE config 5, e.g. --workload 2160p): ONE stream, every picture cut into the
same 8 slices whatever N is, rank r codes a contiguous band of slices and the ranks swap
reconstructed rows once per picture; the bands' payloads are gathered on rank 0 and
interleaved into the single-encoder bitstream inside the timed region; value = pixels of the
one clip / max-over-ranks time; scaling = "strong".
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # BASELINE.json configs[1] -- the configuration the metric is quoted on
    "1080p": dict(w=1920, h=1080, frames=300, keyint=30, qp=26, slices=8,
                  name="1080p30 YUV420p, 300 frames, IPPP closed GOP 30, QP 26, 8 slices, 16x16 full search +-16"),
    # BASELINE.json configs[2]
    "2160p": dict(w=3840, h=2160, frames=300, keyint=30, qp=26, slices=8,
                  name="2160p30 YUV420p, 300 frames, IPPP closed GOP 30, QP 26, 8 slices, 16x16 full search +-16"),
    "cif": dict(w=352, h=288, frames=300, keyint=30, qp=26, slices=8, name="CIF smoke workload"),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
ME_EVENT_STRIDE = 7             # timed region: HIP-event pairs around every 7th k_me launch (each pair stalls the stream ~8 us: 1 in 4 cost 3.6 % of a 1080p pass).  Odd on purpose: the side chain works in batches of four steps, and one launch in eight always met the same phase of that cycle (rocprofv3 over all launches read 94 us where the sampled events read 100)
ME_ALGO_BYTES_PER_PX = 2.0      # k_me reads cur Y (1.0) + ref Y (1.0) per P-frame pixel
ORACLE_NOTE = "CPU oracle = this repo's own C restatement (oracle/), not libx264; parity with libx264 unpinned"


def host_frames(wl, host):
    """[(y, u, v)] views of frames copied back from the resident clip (host: uint8 array [n][w*h*3/2])."""
    w, h = wl["w"], wl["h"]
    return [(f[:w * h].reshape(h, w), f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), f[w * h * 5 // 4:].reshape(h // 2, w // 2)) for f in host]


ORACLE_OPTS = {}  # the coding options of this run that are not part of the workload (--crf, --intra-modes, --subpel)


def oracle_run(wl, clip, threads, gop_workers=1):
    """The CPU oracle over `clip` (list of (y, u, v)): (seconds, sha256 of the concatenated payloads, bytes).
    gop_workers > 1: closed GOPs are independent, so that many encoder instances (each slice-parallel on `threads` OpenMP
    threads, its idr_base = the index of its GOP) work side by side from a thread pool (ctypes releases the GIL); their
    streams concatenate, in GOP order, to exactly the one-encoder stream (tests/test_dist.py)."""
    import orc_py
    o = ORACLE_OPTS
    keyint = wl["keyint"]

    def make(idr_base):
        return orc_py.Encoder(wl["w"], wl["h"], qp=int(round(o["crf"])) if o.get("crf") is not None else wl["qp"], n_slices=wl["slices"], keyint=keyint,
                              fps=(30, 1), sar=(1, 1), threads=threads, idr_base=idr_base, rc=o.get("crf") is not None,
                              intra_modes=int(o.get("intra_modes") or 0), subpel=int(o.get("subpel", 0)), deblock=bool(o.get("deblock")))

    def gop(gi):
        enc = make(gi)
        out = [enc.encode(y, u, v).data for y, u, v in clip[gi * keyint:(gi + 1) * keyint]]
        enc.close()
        return out
    n_gops = (len(clip) + keyint - 1) // keyint
    md, nbytes = hashlib.sha256(), 0
    t0 = time.perf_counter()
    if gop_workers > 1 and n_gops > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=gop_workers) as ex:
            parts = list(ex.map(gop, range(n_gops)))
    else:  # one encoder over the whole clip
        enc = make(0)
        parts = [[enc.encode(y, u, v).data for y, u, v in clip]]
        enc.close()
    dt = time.perf_counter() - t0
    for part in parts:
        for d in part:
            md.update(d)
            nbytes += len(d)
    return dt, md.hexdigest(), nbytes


def usable_cores():
    """Host cores this process may actually use: the box reports every core of the node (os.cpu_count()), its share can be
    smaller (CPU affinity, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_shape(n_gops):
    """(threads per encoder, encoders side by side) for the all-cores CPU baseline: slices are the unit inside an encoder
    (8 per picture), closed GOPs the unit across encoders."""
    cores = usable_cores()
    threads = min(8, cores)
    return threads, max(1, min(n_gops, cores // threads))


def hip_digest(enc, clip_ptr, fb, frames, first):
    """SHA-256 over the payloads of the first `first` frames of one untimed pass (Python sink), and their byte count."""
    md, cnt = hashlib.sha256(), [0, 0]

    def sink(payload, nals, pts, key):
        if cnt[0] < first:
            md.update(payload)
            cnt[1] += len(payload)
        cnt[0] += 1
    enc.encode_resident(clip_ptr, fb, frames, sink)
    return md.hexdigest(), cnt[1]


def reference_defaults(venc, torch, orc_check, wl, clip, fb, device, passes=2):
    """What a caller shaped like the reference gets (missing #3): x264_param_default_preset("medium", "film"),
    X264_RC_CRF at the reference's default quality 20 (av_encode.c:102-104, :384-401), nothing else set but size / frame rate / SAR
    / b_annexb, ONE pic_in filled from host memory per call (av_encode.c:543-547) through x264_encoder_encode (:970) and the flush
    loop (:1076-1083) -- every coding tool the presets switch on, the number of GOPs in lock-step the encoder chooses itself.
    Bit-exactness: SHA-256 of the first GOPs' payloads against the oracle coding the same frames with the same tools."""
    w, h, frames = wl["w"], wl["h"], wl["frames"]
    H, L = venc.hip_lib(), venc.x264_lib()
    nbytes = frames * fb
    hp = H.venc_host_alloc(nbytes)
    if not hp:
        return {"error": "pinned host allocation of %d bytes failed" % nbytes}
    try:
        host = torch.frombuffer((C.c_uint8 * nbytes).from_address(hp), dtype=torch.uint8)
        host.copy_(clip.cpu())
        enc = venc.X264Encoder(w, h, fps=(30, 1), sar=(1, 1), preset="medium", tune="film", crf=20, device=device, reference_defaults=True)
        p = enc.param
        L.x264_venc_gop_batch.argtypes = [C.c_void_p]
        items = L.x264_venc_gop_batch(enc.h)
        keyint, slices = int(p.i_keyint_max), int(p.i_slice_count)
        sink_fn = C.cast(L.x264_venc_tally_sink, C.c_void_p)
        L.x264_venc_encode_host_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

        def run(n_passes):
            t = venc.X264Tally()
            r = L.x264_venc_encode_host_frames(enc.h, C.c_void_p(hp), fb, w, h, frames, n_passes, sink_fn, C.cast(C.pointer(t), C.c_void_p))
            if r < 0:
                raise SystemExit("reference_defaults: x264_venc_encode_host_frames failed: %s" % H.venc_last_error().decode())
            return r, int(t.bytes)
        # several clips per unit when the encoder batches more GOPs than one clip holds: enough passes to fill two units
        unit_passes = max(1, -(-items * keyint // frames))
        run(unit_passes)  # warm-up: pools, payload buffers
        torch.cuda.synchronize()
        n_timed = max(passes, 2 * unit_passes)
        t0 = time.perf_counter()
        nf, nb = run(n_timed)
        dt = time.perf_counter() - t0
        # the same over a job ten units long (a film, not a clip: the fill of the first unit and the drain of the last are a tenth of it)
        n_long = 10 * unit_passes
        t0l = time.perf_counter()
        nfl, _ = run(n_long)
        dtl = time.perf_counter() - t0l
        # the same encoder with the clip resident in HBM (no copy per call): what the kernels of the preset's tools deliver
        enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=unit_passes)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr = enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=n_timed)
        dtr = time.perf_counter() - t1
        enc.close()
        out = {"value": round(nf * w * h / dt / 1e6, 2), "unit": "Mpixels/s", "ms_per_clip": round(dt / n_timed * 1e3, 3), "h2d_GBps": round(nf * fb / dt / 1e9, 2),
               "resident_mpx_s": round(int(tr.frames) * w * h / dtr / 1e6, 2),
               "long_job": {"frames": nfl, "mpx_s": round(nfl * w * h / dtl / 1e6, 2), "h2d_GBps": round(nfl * fb / dtl / 1e9, 2),
                            "what": "the same call sequence over %d passes (ten units of %d GOPs): fill and drain are a tenth of the job instead of %d %%" % (n_long, items, round(100.0 * unit_passes / n_timed))},
               "bytes_per_clip": nb // n_timed, "gop_batch_chosen": items, "keyint": keyint, "slices": slices,
               "tools": {"rate_control": "CRF 20 (spec/CODEC.md 1a)", "intra_modes": int(p.hip.i_intra_modes), "subpel": int(p.hip.i_subpel), "deblock": int(p.hip.i_deblock)},
               "what": "x264_param_default_preset(medium, film) + X264_RC_CRF 20, one pic_in from pinned host memory per x264_encoder_encode call + the flush loop "
                       "(av_encode.c:384-401, :543-547, :970, :1076-1083); %d passes of the %d-frame clip as one stream, pipeline fill and drain inside the timing" % (n_timed, frames)}
        if orc_check:
            # the first two GOPs through the same entry points (Python loop), against the oracle with the same tools
            n_chk = min(frames, 2 * keyint)
            e2 = venc.X264Encoder(w, h, fps=(30, 1), sar=(1, 1), preset="medium", tune="film", crf=20, device=device, reference_defaults=True)
            hf = host.numpy().reshape(frames, fb)
            got = []
            for t in range(n_chk):
                f = hf[t]
                r = e2.encode(f[:w * h].reshape(h, w), f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), f[w * h * 5 // 4:].reshape(h // 2, w // 2), t)
                if r is not None:
                    got.append(r)
            got += e2.flush()
            e2.close()
            import orc_py
            o = orc_py.Encoder(w, h, qp=20, n_slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=min(8, usable_cores()), rc=True,
                               intra_modes=int(p.hip.i_intra_modes), subpel=int(p.hip.i_subpel), deblock=bool(p.hip.i_deblock))
            ma, mb = hashlib.sha256(), hashlib.sha256()
            for t in range(n_chk):
                f = hf[t]
                mb.update(o.encode(f[:w * h].reshape(h, w), f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), f[w * h * 5 // 4:].reshape(h // 2, w // 2)).data)
                ma.update(got[t].payload)
            o.close()
            out["bit_exact"] = {"frames": n_chk, "hip_sha256": ma.hexdigest(), "oracle_sha256": mb.hexdigest(), "equal": ma.hexdigest() == mb.hexdigest(), "note": ORACLE_NOTE}
        return out
    finally:
        H.venc_host_free(hp)


def pcie_inclusive(venc, torch, wl, clip, fb, gop_batch, device, passes=3, input_depth=1):
    """Host pictures through x264_encoder_encode (av_encode.c:970): the clip sits in pinned host memory, pic_in's plane
    pointers walk through it (no fread, no host copy), every call uploads one frame (copy stream) and returns once the
    picture may be reused; then the reference's flush loop (av_encode.c:1076-1083).  Whole job: pipeline fill and drain
    included."""
    w, h, frames, keyint = wl["w"], wl["h"], wl["frames"], wl["keyint"]
    H, L = venc.hip_lib(), venc.x264_lib()
    nbytes = frames * fb
    hp = H.venc_host_alloc(nbytes)
    if not hp:
        return {"error": "pinned host allocation of %d bytes failed" % nbytes}
    try:
        host = torch.frombuffer((C.c_uint8 * nbytes).from_address(hp), dtype=torch.uint8)
        host.copy_(clip.cpu())
        enc = venc.X264Encoder(w, h, qp=wl["qp"], slices=wl["slices"], keyint=keyint, fps=(30, 1), sar=(1, 1), gop_batch=gop_batch, device=device,
                               input_depth=input_depth)
        sink_fn = C.cast(L.x264_venc_tally_sink, C.c_void_p)
        L.x264_venc_encode_host_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

        def run(n_passes):
            # the reference's loop in C (x264_venc_encode_host_frames: pic_in's planes walk through the pinned clip, x264_encoder_encode
            # per frame, then the flush loop), so that the harness's own per-call cost does not sit between two copies
            t = venc.X264Tally()
            r = L.x264_venc_encode_host_frames(enc.h, C.c_void_p(hp), fb, w, h, frames, n_passes, sink_fn, C.cast(C.pointer(t), C.c_void_p))
            if r < 0:
                raise SystemExit("pcie_inclusive: x264_venc_encode_host_frames failed: %s" % H.venc_last_error().decode())
            return r, int(t.bytes)
        run(1)  # warm-up: pools, payload buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nf, nb = run(passes)
        dt = time.perf_counter() - t0
        t0l = time.perf_counter()
        nfl, _ = run(10 * passes)  # a job ten times as long: the drain of the last unit is a thirtieth of it
        dtl = time.perf_counter() - t0l
        enc.close()
        assert nf == passes * frames, (nf, passes * frames)
        return {"value": round(nf * w * h / dt / 1e6, 2), "unit": "Mpixels/s", "ms_per_clip": round(dt / passes * 1e3, 3),
                "bytes_per_clip": nb // passes, "h2d_GBps": round(nf * fb / dt / 1e9, 2),
                "long_job": {"frames": nfl, "mpx_s": round(nfl * w * h / dtl / 1e6, 2), "h2d_GBps": round(nfl * fb / dtl / 1e9, 2)},
                "input_depth": input_depth,
                "what": "%d passes of the %d-frame %dx%d clip from pinned host memory through x264_encoder_encode + the flush loop "
                        "(one H2D copy per call on the copy stream; %s; gop_batch %d; pipeline fill and drain inside the timing)" % (
                            passes, frames, w, h, "the call returns when its own picture may be rewritten -- the reference's one pic_in (av_encode.c:545)"
                            if input_depth == 1 else "hip.i_input_depth = %d: the caller rotates %d pictures, the call waits for the copy issued %d call(s) earlier" % (
                                input_depth, input_depth, input_depth - 1), gop_batch)}
    finally:
        H.venc_host_free(hp)



# ----------------------------------------------------------------------------- launching ranks
ELASTIC_VARS = ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
                "GROUP_WORLD_SIZE", "LOCAL_WORLD_SIZE")


def rank_env(rank, world, port, base=None):
    """Environment of one rank process started by bench.py itself (env:// rendezvous on 127.0.0.1, rank 0 hosts the store)."""
    env = dict(base if base is not None else os.environ)
    for k in ELASTIC_VARS:
        env.pop(k, None)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def self_launch(argv, world):
    """--gpus N > 1 and no WORLD_SIZE: the parent starts N rank processes of this script and touches no GPU itself.
    Rank 0 inherits stdout (its one JSON line is this command's output); returns the worst exit code."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 2000))
    procs = []
    for r in range(world):
        env = rank_env(r, world, port)
        env["BENCH_LAUNCHED_BY"] = "bench.py self-launch"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all of them: a rank that exits early with an error leaves its peers blocked in the rendezvous or a barrier until
    # torch's multi-minute timeout -- end the others at once and report the rank that failed (fresh processes; nothing is re-executed)
    rc, failed = 0, None
    live = list(range(world))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.remove(r)
            if code != 0 and rc == 0:
                rc, failed = code, r
                for o in live:
                    procs[o].terminate()
        if live:
            time.sleep(0.05)
    if rc:
        sys.stderr.write("bench: rank %d failed (exit code %d); the other ranks were stopped\n" % (failed, rc))
    return 1 if rc else 0


def slices_leg(args, rank, world):
    """BASELINE config 5 beside an N > 1 streams run: every rank starts ONE child (this script, --mode slices --workload
    2160p, control plane on gloo so that the engine's RCCL is the only RCCL that makes a communicator there) before it has
    touched its GPU, and waits for it under a time limit.  The children rendezvous among themselves on their own port.
    Rank 0 returns the child's JSON line as a dict (or {"error": ...}); the other ranks return None."""
    import signal
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29500")) + 23
    env = rank_env(rank, world, port)
    env["LOCAL_RANK"] = os.environ.get("LOCAL_RANK", str(rank))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--mode", "slices", "--workload", "2160p", "--leg-child",
           "--steps", str(args.slices_steps), "--warmup", "1", "--cpu-frames", "0", "--no-extras", "--halo", str(args.halo),
           "--backend", args.backend, "--control", "gloo", "--gop-batch", str(args.gop_batch)]
    if args.device >= 0:
        cmd += ["--device", str(args.device)]
    if args.dry_run:
        cmd += ["--dry-run"]
    t0 = time.perf_counter()
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                          stderr=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, start_new_session=True)
    try:
        out, err = pr.communicate(timeout=args.slices_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)  # its own session: nothing of ours is in that group
        except OSError:
            pass
        pr.wait()
        return {"error": "the slices leg did not finish within %d s and was stopped" % args.slices_timeout} if rank == 0 else None
    if rank != 0:
        return None
    lines = [ln for ln in (out or b"").decode(errors="replace").splitlines() if ln.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"error": "the slices leg failed (exit code %s): %s" % (pr.returncode, (err or b"").decode(errors="replace")[-400:])}
    d = json.loads(lines[-1])
    d["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return d


class quiet_stdout:
    """stdout carries ONE JSON line: whatever a library prints there while the process group forms (gloo announces its
    connections on std::cout) goes to stderr instead."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def dry_run(args, rank, world, slices):
    """CPU rehearsal of the launch logic (tests/test_dist.py): rendezvous, barrier, max-over-ranks, one JSON line.  No GPU."""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with quiet_stdout():
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == float(world)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": "Mpixels/s encoded at fixed QP, 1080p & 4K YUV420p; bit-exact vs CPU ref", "value": 0.0, "unit": "Mpixels/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "mode": args.mode,
               "launched_by": os.environ.get("BENCH_LAUNCHED_BY", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "environment")}
        if slices is not None:
            out["slices_2160p"] = slices
        print(json.dumps(out), flush=True)


def measure(args, torch, dist, venc, wl_key, steps, warmup, rank, world, local_rank, want_digest_frames=0):
    """One workload on this rank: the timed region of the contract + the untimed per-kernel passes.  Returns the fields
    of the JSON line that depend on the workload (rank 0) or None."""
    wl = WORKLOADS[wl_key]
    w, h, frames, keyint = wl["w"], wl["h"], wl["frames"], wl["keyint"]
    # closed GOPs advanced in lock-step: the GOPs of one clip by default; with the in-loop filter or the Intra4x4 trial on -- kernels that
    # walk a dependency wavefront inside every slice, one block per (slice, item), and fill the chip only through the item count -- the
    # GOPs of three consecutive passes of the stream
    # 0 = the encoder's own choice (x264_encoder_open: by picture size, include/x264.h; 30 at 1080p, 15 at 2160p, 32 with the wavefront tools);
    # slices mode keeps one clip's GOPs per call (every call is one pass, and the exchange callback is per call)
    n_gops = (frames + keyint - 1) // keyint
    gop_batch = args.gop_batch or (n_gops if args.mode == "slices" else 0)
    fb = w * h * 3 // 2
    slices_mode = args.mode == "slices"

    # synthetic clip, generated on the GPU straight into HBM (same bytes as the host generator)
    clip = torch.empty(frames * fb, dtype=torch.uint8, device="cuda")
    if venc.hip_lib().venc_synth_device(clip.data_ptr(), fb, w, h, 0, frames, 0x1234ABCD + (0 if slices_mode else rank)) != 0:
        raise SystemExit("venc_synth_device failed")
    torch.cuda.synchronize()

    band_kw, xc, dmod = {}, None, None
    c_exchange = slices_mode and args.backend == "nccl"
    if slices_mode:
        dmod = importlib.import_module("video-encoder_amd.dist")
        s0, cnt = dmod.slice_band(wl["slices"], world, rank)
        if c_exchange:
            # the exchange runs inside the engine, in C on RCCL (include/venc_hip.h "RCCL exchange"): Python only carries
            # rank 0's 128-byte communicator id to the peers, once
            ident = [venc.rccl_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(ident, src=0)
            band_kw = dict(rccl=(world, rank, ident[0], args.halo))
        else:  # gloo rehearsals: host-staged exchange through torch.distributed
            xc = dmod.BandExchange(dmod.EnginePort(local_rank), wl["slices"], world, rank, halo_rows=args.halo)
            band_kw = dict(slice_first=s0, slice_count=cnt, exchange=xc)
    enc = venc.X264Encoder(w, h, qp=wl["qp"], slices=wl["slices"], keyint=keyint, fps=(30, 1), sar=(1, 1),
                           gop_batch=gop_batch, device=local_rank, intra_modes=args.intra_modes, crf=args.crf, subpel=args.subpel, deblock=args.deblock, **band_kw)
    gop_batch = enc.gop_batch()  # what 0 resolved to (after the cap by device memory)
    # passes of the clip that make whole units: the untimed per-kernel passes below run this many, so that their launches hold as many items as the timed ones
    unit_passes = 1 if slices_mode else (gop_batch // math.gcd(gop_batch, n_gops))
    if not args.no_profile:
        enc.profile(True, capacity=8 * frames * max(2, steps, warmup))
        # timed region: HIP-event pairs around the dominant kernel only (each pair stalls the stream for a
        # few microseconds; pairs around all nine kernels of a step slowed the step by ~20 %).  The other
        # kernels are timed in the untimed passes below.
        enc.profile_select(["k_me"], stride=ME_EVENT_STRIDE)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    stream_info = {}

    def one_pass():
        """streams: C tally sink.  slices: collect this band's payloads, gather them on rank 0 and
        interleave them into the whole-picture bitstream (part of the job, so inside the timing)."""
        if not slices_mode or world == 1:
            return enc.encode_resident_tally(clip.data_ptr(), fb, frames)
        mine = []
        enc.encode_resident(clip.data_ptr(), fb, frames, sink=lambda payload, nals, pts, key: mine.append(payload))
        if c_exchange and venc.hip_lib().venc_rccl_agree(enc.L.x264_venc_engine(enc.h), 0) != 0:
            raise SystemExit("bench: a rank reported a failure (venc_rccl_agree)")  # every rank stops together
        parts = dmod.gather_frames_to_root(mine)  # rank 0 only: the bitstream is assembled where it is written out
        if rank == 0:
            whole = dmod.interleave_bands(parts)
            stream_info["frames"] = whole
        return None

    # K steps = K passes over the resident clip.  streams mode hands them to the engine as ONE stream of K x frames
    # (x264_venc_encode_resident_loop): the GPU works on pass k + 1 while the host packages pass k and hands it to the
    # sink, as a server encoding batch after batch does; --serial-steps makes K separate calls, each of which returns only
    # after its last frame was delivered.  slices mode: separate calls (the exchange callback is per call).
    pipelined = (not slices_mode or world == 1) and not args.serial_steps and steps > 1
    tally = None
    if pipelined and warmup > 0:
        # the warm-up goes through the same call; at least two passes, so that both halves of the engine's
        # double-buffered output ring (and their host payload buffers) have been touched before the clock starts
        enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=max(warmup, 2))
    else:
        for _ in range(warmup):
            tally = one_pass()
    if not args.no_profile:
        enc.profile_read()
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        tally = enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=steps)
    else:
        for _ in range(steps):
            tally = one_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spread = None
    if pipelined and tally is not None and int(tally.passes) >= 3:
        # the K passes are ONE pipelined stream: a pass "ends" when its last frame reaches the sink (CLOCK_MONOTONIC stamps
        # taken by the C tally sink); the first interval contains the pipeline's fill and is left out of the spread
        # (a unit of gop_batch GOPs delivers unit_passes passes at once: the intervals are taken between the ends of whole units)
        ends = [int(tally.pass_end_ns[i]) for i in range(int(tally.passes))][unit_passes - 1::unit_passes]
        per = sorted((b - a) * 1e-9 for a, b in zip(ends[:-1], ends[1:])) or [dt]
        rate = lambda sec: round(unit_passes * frames * w * h / sec / 1e6, 1)
        spread = {"passes": len(per), "min": rate(per[-1]), "median": rate(per[len(per) // 2]), "max": rate(per[0]), "unit": "Mpixels/s",
                  "what": "rate of each timed unit (%d passes of the clip: the GOPs the encoder advances in lock-step) but the first, from the delivery time of its last frame to the sink "
                          "(one pipelined stream: neighbouring units overlap, so single units can read above the whole-run value)" % unit_passes}
    bytes_per_step = int(tally.bytes) // (steps if pipelined else 1) if tally else None
    bytes_timed = int(tally.bytes) if tally else None
    prof, dropped = ({}, False) if args.no_profile else enc.profile_read()
    barrier()
    if pipelined and not args.no_check_pass:
        tally = one_pass()  # untimed: the digest of ONE pass, comparable across runs whatever --steps was
        if bytes_timed != steps * int(tally.bytes):  # every pass of the timed stream must have produced this pass's bytes
            raise SystemExit("bench: the %d timed passes produced %d bytes, %d x one pass = %d" % (
                steps, bytes_timed, steps, steps * int(tally.bytes)))
        if not args.no_profile:
            enc.profile_read()
        barrier()
    # Untimed extra pass with the two kernel chains serialised on one stream: per-kernel durations
    # without the concurrency that the timed region uses (a kernel sharing the chip with another
    # chain's kernels takes longer although it does the same work).
    prof_iso, prof_all = {}, {}
    if not args.no_profile and rank == 0:
        if not slices_mode or world == 1:  # (every rank would have to join the exchanges of an extra pass)
            enc.profile_select(None, stride=1)
            enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=unit_passes)
            prof_all, _ = enc.profile_read()
            enc.set_overlap(False)
            enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=unit_passes)
            prof_iso, _ = enc.profile_read()
            enc.set_overlap(True)
            # per pass of the clip (the passes above were unit_passes long)
            prof_all = {k: (v[0] / unit_passes, v[1]) for k, v in prof_all.items()}
            prof_iso = {k: (v[0] / unit_passes, v[1]) for k, v in prof_iso.items()}
    barrier()
    digest, host = None, None
    if want_digest_frames and rank == 0 and not slices_mode:
        if not args.no_profile:
            enc.profile(False)
        digest = hip_digest(enc, clip.data_ptr(), fb, frames, want_digest_frames)
        host = clip[:want_digest_frames * fb].cpu().numpy().reshape(want_digest_frames, fb)  # the oracle's input: the very same bytes
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    res = None
    if rank == 0:
        px = (1 if slices_mode else world) * steps * frames * w * h
        res = {
            "value": round(px / dt / 1e6, 2),
            "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "warmup": warmup, "per_pass": spread,
            "config": {"workload": wl["name"], "frames_per_step": frames, "gop_batch": gop_batch,
                       "parallelism": ("%d slice bands of one stream, %s exchange of reconstructed rows per picture (%s)" % (
                           world, "halo (%d luma rows per edge)" % args.halo if args.halo else "whole-band", args.backend))
                       if slices_mode else ("1 stream per GPU, no collective" if world > 1 else "single GPU"),
                       "steps_pipelined": pipelined,
                       "intra_modes": ["DC", "V/H/DC", "V/H/DC/plane", "Intra4x4 | Intra16x16 V/H/DC/plane"][args.intra_modes], "subpel": ["none", "half-sample", "quarter-sample"][args.subpel], "deblock": bool(args.deblock), "rate_control": ("crf %g" % args.crf) if args.crf is not None else "fixed QP",
                       "bytes_per_step": bytes_per_step,
                       "bitstream_fnv1a": "%016x" % tally.fnv1a if tally else None,
                       "cpu_oracle_note": ORACLE_NOTE},
        }
        if digest:
            res["hip_sha256"], res["hip_sha256_frames"], res["hip_sha256_bytes"] = digest[0], want_digest_frames, digest[1]
        if slices_mode:
            eng = enc.L.x264_venc_engine(enc.h)
            res["config"]["exchange"] = "C, RCCL (venc_band_exchange_rccl)" if c_exchange else "Python, torch.distributed %s (rehearsal)" % args.backend
            n_x = venc.hip_lib().venc_rccl_exchanges(eng) if c_exchange else xc.exchanges
            res["config"]["exchanges_per_step"] = int(n_x) // max(1, steps + warmup)
            # one exchange call serves every item (closed GOP) of a lock-step step: per picture = calls x items per call / pictures
            n_gops = (frames + keyint - 1) // keyint
            res["config"]["exchange_calls_per_step"] = res["config"]["exchanges_per_step"]
            res["config"]["exchanges_per_picture"] = round(res["config"]["exchanges_per_step"] * min(gop_batch, n_gops) / float(frames), 4)
            if c_exchange:
                res["config"]["rccl_ranks"] = int(venc.hip_lib().venc_rccl_comm_count(eng))  # ncclCommCount of the engine's communicator
                # P steps whose search ran as interior tile rows (beside the exchange, on its own stream) + edge tile rows (behind it)
                res["config"]["search_split_steps_per_step"] = int(venc.hip_lib().venc_band_split_steps(eng)) // max(1, steps + warmup)
            if world > 1:
                whole = stream_info.get("frames", [])
                res["config"]["bytes_per_step"] = sum(len(f) for f in whole)
                res["config"]["bitstream_sha256"] = hashlib.sha256(b"".join(whole)).hexdigest()
                res["config"].pop("bitstream_fnv1a", None)
        if prof:
            me_ms, me_n = prof.get("k_me", (0.0, 0))
            if me_n:
                # P launches of one pass over the clip: every unit of gop_batch GOPs advances keyint - 1 P pictures in lock-step
                # (the timed stream holds steps x n_gops GOPs when the passes are pipelined into one stream, n_gops per call otherwise)
                stream_gops = n_gops * (steps if pipelined else 1)
                launches_per_stream = ((stream_gops + gop_batch - 1) // gop_batch) * (keyint - 1)
                px_per_launch = stream_gops * (keyint - 1) * w * h / launches_per_stream
                if slices_mode:  # rank 0 searches only its band
                    r0, r1 = venc.Engine.view(enc.L.x264_venc_engine(enc.h)).band_rows(*dmod.slice_band(wl["slices"], world, 0))
                    px_per_launch *= (r1 - r0) * 16.0 / h
                px_per_launch /= venc.hip_lib().venc_search_split(enc.L.x264_venc_engine(enc.h))  # two chains: a timed launch searches half the items
                algo_bytes = ME_ALGO_BYTES_PER_PX * px_per_launch
                avg_s = me_ms / me_n * 1e-3
                achieved = algo_bytes / avg_s / 1e9
                # kept counter values (PMC passes cannot run inside this process): quoted only while the device sources are
                # the ones they were collected on
                traffic, traffic_source, valu = None, "no profiles/pmc_traffic.json", None
                src_now = venc.kernel_sources_sha16()
                for fname, key in (("pmc_traffic.json", "traffic"), ("sq_counters.json", "valu")):
                    path = os.path.join(ROOT, "profiles", fname)
                    if not os.path.exists(path):
                        continue
                    try:
                        doc = json.load(open(path))
                    except Exception:
                        continue
                    tag = doc.get("source") or {}
                    fresh = tag.get("kernel_sources_sha16") == src_now
                    where = "profiles/%s collected %s at commit %s (kernel sources %s)" % (fname, tag.get("collected"), tag.get("git_head"), tag.get("kernel_sources_sha16"))
                    if key == "traffic":
                        traffic = doc.get(wl_key, {}).get("k_me_hbm_bytes_per_launch") if fresh else None
                        traffic_source = where if fresh else "dropped: %s is older than the kernels now built (%s)" % (where, src_now)
                    elif fresh and doc.get(wl_key, {}).get("k_me_pool"):
                        kk = doc[wl_key]["k_me_pool"]
                        valu = {"bound": "valu issue", "frac": kk.get("valu_issue_frac"), "valu_wave_instructions_per_launch": kk.get("valu_wave_instructions_per_launch"),
                                "isolated_launch_us": kk.get("avg_us"), "lds_busy_frac": kk.get("lds_busy_frac"), "source": where,
                                "note": "the bound that is active: SQ_INSTS_VALU x 4.3 cycles (tools/valu_ubench.hip) over the SIMD-cycles of an isolated launch"}
                res["roofline"] = {"bound": "hbm", "kernel": "k_me_pool = exact +-16 full search + luma 4x4 transform / quantisation / reconstruction, fused (the ME+DCT kernel)",
                                   "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                                   "algorithmic_bytes_per_launch": int(algo_bytes),
                                   "avg_launch_us": round(avg_s * 1e6, 2), "launches": me_n, "launches_sampled_one_in": ME_EVENT_STRIDE,
                                   "search_positions_per_s": round(1089.0 * 256.0 * px_per_launch / 256.0 / avg_s, 1),
                                   "timed_region_runs_two_kernel_chains_concurrently": True,
                                   "events_dropped": dropped,
                                   "note": "HBM is the bound this line is priced against; the kernel itself is VALU/LDS bound (DESIGN.md 4.1). "
                                           "Algorithmic bytes = 2.0 B per P-frame pixel read (cur Y + ref Y); since round 2 the same launch also does the luma "
                                           "transform that round 1 ran as a separate kernel, so its duration is not comparable with round 1's search-only k_me_pool"}
                if valu:
                    res["roofline_valu"] = valu
                # the whole P path of SURVEY.md section 8d (3.0 B/px read: cur Y+U+V 1.5, ref Y 1.0, ref U+V 0.5) over one main-chain period
                if prof_all:
                    chain_ms = sum(prof_all.get(k, (0.0, 0))[0] for k in ("k_me", "k_me_fix", "k_inter"))
                    n_p = prof_all.get("k_me", (0.0, 0))[1]
                    if n_p:
                        per = chain_ms * unit_passes / n_p * 1e-3
                        res["roofline"]["p_path"] = {"algorithmic_bytes_per_step": int(1.5 * algo_bytes), "kernels": "k_me_pool + k_me_fix + chroma k_inter_fused",
                                                     "sum_of_kernel_us": round(per * 1e6, 2), "achieved": round(1.5 * algo_bytes / per / 1e9, 2),
                                                     "frac": round(1.5 * algo_bytes / per / 1e9 / HBM_PEAK_GBS, 5)}
                iso_ms, iso_n = prof_iso.get("k_me", (0.0, 0))
                if iso_n:
                    iso_s = iso_ms * unit_passes / iso_n * 1e-3  # (the milliseconds are per pass, the launch count is the run's)
                    iso_px = unit_passes * n_gops * (keyint - 1) * w * h / iso_n
                    iso_bytes = ME_ALGO_BYTES_PER_PX * iso_px
                    res["roofline"]["isolated_avg_launch_us"] = round(iso_s * 1e6, 2)
                    res["roofline"]["isolated_achieved"] = round(iso_bytes / iso_s / 1e9, 2)
                    res["roofline"]["isolated_frac"] = round(iso_bytes / iso_s / 1e9 / HBM_PEAK_GBS, 5)
                    res["roofline"]["isolated_algorithmic_bytes_per_launch"] = int(iso_bytes)
            if prof_all:  # untimed pass, both chains concurrent as in the timed region, event pairs around every kernel
                res["kernel_ms_per_step"] = {k: round(v[0], 3) for k, v in prof_all.items()}
                res["kernel_ms_per_step"]["sum"] = round(sum(v[0] for v in prof_all.values()), 3)
            else:
                res["kernel_ms_per_step"] = {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}
            if prof_iso:
                res["kernel_ms_per_step_isolated"] = {k: round(v[0], 3) for k, v in prof_iso.items()}
    enc.close()
    if slices_mode and world > 1 and args.leg_child:
        # the same clip through ONE whole-picture encoder on rank 0 (untimed): the slice-parallel bitstream must equal it
        if rank == 0:
            solo = venc.X264Encoder(w, h, qp=wl["qp"], slices=wl["slices"], keyint=keyint, fps=(30, 1), sar=(1, 1), gop_batch=gop_batch,
                                    device=local_rank, intra_modes=args.intra_modes, crf=args.crf, subpel=args.subpel, deblock=args.deblock)
            sha1, nb1 = hip_digest(solo, clip.data_ptr(), fb, frames, frames)
            solo.close()
            res["config"]["one_rank_sha256"] = sha1
            res["config"]["equals_one_rank_bitstream"] = sha1 == res["config"].get("bitstream_sha256") and nb1 == res["config"].get("bytes_per_step")
        barrier()
    extras = {}
    if rank == 0 and world == 1 and not slices_mode and args.pcie and wl_key == args.workload:
        extras["reference_defaults"] = reference_defaults(venc, torch, args.cpu_frames > 0, wl, clip, fb, local_rank)
        # (ten GOPs in lock-step, as in every earlier round: these legs are bound by the copy engine, more items only lengthen the fill and
        # the drain of their 900-frame job; the encoder's own choice under the reference's call sequence is the reference_defaults leg)
        extras["pcie_inclusive"] = pcie_inclusive(venc, torch, wl, clip, fb, n_gops, local_rank)
        extras["pcie_inclusive_rotating_pictures"] = pcie_inclusive(venc, torch, wl, clip, fb, n_gops, local_rank, input_depth=2)
    del clip
    torch.cuda.empty_cache()
    return res, extras, host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed passes over the clip (default 30: ~125 ms timed at 1080p; the pipeline's fill and drain, ~2.5 ms, are inside the timing)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--gop-batch", type=int, default=0, help="closed GOPs advanced in lock-step (0 = the encoder's choice: x264_encoder_open, include/x264.h)")
    ap.add_argument("--no-check-pass", action="store_true", help="skip the untimed one-pass digest check behind the timed passes (counter runs: only whole units in the process)")
    ap.add_argument("--cpu-frames", type=int, default=300, help="frames of the all-cores CPU baseline sample, which is also the bit-exact check (0 = skip both); 300 = the whole clip, ~12 s on 16 cores")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only: no 2160p object, no PCIe-inclusive leg (profiling runs)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--serial-steps", action="store_true", help="one engine call per step (no overlap between consecutive steps)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--mode", default="streams", choices=("streams", "slices"),
                    help="streams: one clip per GPU, no collective (default); slices: one clip, slice bands across GPUs")
    ap.add_argument("--halo", type=int, default=16, help="slices mode: luma rows exchanged per band edge (0 = whole bands)")
    ap.add_argument("--deblock", action="store_true", help="in-loop deblocking filter inside every slice (spec/CODEC.md section 5a); default off: not a headline configuration")
    ap.add_argument("--subpel", type=int, default=0, choices=[0, 1, 2], help="sub-sample refinement of the motion vectors (spec/CODEC.md section 2a): 0 none (default, the definition the metric is quoted on), 1 half-sample, 2 half- then quarter-sample; not a headline configuration")
    ap.add_argument("--intra-modes", type=int, nargs="?", const=1, default=0, choices=[0, 1, 2, 3], help="intra prediction of I pictures: 0 Intra16x16 DC only (default, the definition the metric's digests are frozen on), 1 vertical / horizontal / DC by SAD, 2 plane as well, 3 Intra4x4 tried against that (the C API's preset default); not a headline configuration")
    ap.add_argument("--crf", type=float, default=None, help="X264_RC_CRF with this rate factor instead of fixed QP (spec/CODEC.md section 1a); not a headline configuration")
    ap.add_argument("--device", type=int, default=-1, help="force this GPU ordinal for every rank (rehearsal on a 1-GPU box)")
    ap.add_argument("--control", default=None, choices=("nccl", "gloo"), help="torch.distributed backend of the control plane (barrier, max-over-ranks time, the RCCL id); default: --backend")
    ap.add_argument("--no-slices-leg", action="store_true", help="N > 1 streams run: do not measure BASELINE config 5 beside it")
    ap.add_argument("--slices-timeout", type=int, default=240, help="seconds the slices leg of an N > 1 run may take before it is stopped and reported as an error")
    ap.add_argument("--slices-steps", type=int, default=3, help="timed passes of the slices leg")
    ap.add_argument("--leg-child", action="store_true", help=argparse.SUPPRESS)  # this process is a rank of the slices leg started by slices_leg()
    ap.add_argument("--dry-run", action="store_true", help="launch logic only (rendezvous, barrier, one JSON line): no GPU, gloo; the CPU test of --gpus N")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: start the N ranks here, before anything touches a GPU
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    args.control = args.control or ("gloo" if args.dry_run else args.backend)
    slices = None
    if world > 1 and args.mode == "streams" and not args.no_slices_leg and not args.leg_child:
        try:
            slices = slices_leg(args, rank, world)  # before this process has touched its GPU
        except Exception as e:  # the config-4 number must not depend on the config-5 leg in any way
            slices = {"error": "the slices leg could not be run: %r" % (e,)} if rank == 0 else None
    if args.dry_run:
        return dry_run(args, rank, world, slices)

    import torch
    if args.device >= 0:
        local_rank = args.device
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench: rank %d wants GPU %d, this node has %d" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with quiet_stdout():
            if args.control == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.control, rank=rank, world_size=world)
                dist.barrier()

    venc = importlib.import_module("video-encoder_amd")
    # one process per GPU, on the CPUs of its card's NUMA node (what numactl does in a launcher; include/venc_hip.h venc_numa_bind):
    # on a two-socket host an unbound process keeps the socket it happened to start on, and the far one costs ~5 %
    numa_node = venc.hip_lib().venc_numa_bind(local_rank, 1) if os.environ.get("VENC_NUMA_BIND", "1") != "0" else -1
    ORACLE_OPTS.update(crf=args.crf, intra_modes=args.intra_modes, subpel=args.subpel, deblock=args.deblock)
    slices_mode = args.mode == "slices"
    solo = world == 1 and not slices_mode
    extras_on = solo and not args.no_extras and args.workload == "1080p"
    args.pcie = solo and not args.no_extras
    check = solo and args.cpu_frames > 0
    wl = WORKLOADS[args.workload]

    res, extras, host = measure(args, torch, dist, venc, args.workload, args.steps, args.warmup, rank, world, local_rank,
                          want_digest_frames=min(args.cpu_frames, wl["frames"]) if check else 0)
    out = None
    if rank == 0:
        out = {
            "metric": "Mpixels/s encoded at fixed QP, 1080p & 4K YUV420p; bit-exact vs CPU ref",
            "value": res["value"],
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if slices_mode else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": res["config"],
        }
        out["config"]["host_numa_node"] = numa_node  # -1: not bound (one node, no sysfs, or VENC_NUMA_BIND=0)
        for k in ("per_pass", "roofline", "roofline_valu", "kernel_ms_per_step", "kernel_ms_per_step_isolated"):
            if res.get(k) is not None:
                out[k] = res[k]
        out.update(extras)
        if "value" in (extras.get("pcie_inclusive") or {}):
            # BASELINE.md section 3 words the metric from frame ingest to the last bitstream byte on the host: that number, through the
            # reference's own call (av_encode.c:970) with its H2D copy per frame, rides in `config` beside the HBM-resident `value`
            pi = extras["pcie_inclusive"]
            out["config"]["ingest_inclusive_mpx_s"] = pi["value"]
            out["config"]["h2d_GBps"] = pi["h2d_GBps"]
            out["config"]["ingest_inclusive_long_job_mpx_s"] = (pi.get("long_job") or {}).get("mpx_s")  # ten times the frames: the last unit's drain is a thirtieth
            pr = extras.get("pcie_inclusive_rotating_pictures") or {}
            if "value" in pr:
                out["config"]["ingest_inclusive_rotating_pictures_mpx_s"] = pr["value"]
                out["config"]["h2d_GBps_rotating_pictures"] = pr["h2d_GBps"]
                out["config"]["ingest_inclusive_rotating_pictures_long_job_mpx_s"] = (pr.get("long_job") or {}).get("mpx_s")
        rd = extras.get("reference_defaults") or {}
        if "value" in rd:
            # what the reference's own open sequence delivers (preset medium / film, CRF, every preset tool, one pic_in from host memory)
            out["config"]["reference_defaults"] = {"mpx_s": rd["value"], "long_job_mpx_s": (rd.get("long_job") or {}).get("mpx_s"), "resident_mpx_s": rd.get("resident_mpx_s"), "h2d_GBps": rd["h2d_GBps"], "gop_batch_chosen": rd["gop_batch_chosen"], "tools": rd["tools"],
                                                   "bit_exact_first_gops": (rd.get("bit_exact") or {}).get("equal")}
        if res.get("per_pass") is not None:
            out["config"]["per_pass_mpx_s"] = res["per_pass"]  # min / median / max over the timed passes (box-to-box and within-lease spread: +-5 %)
        if world > 1:
            out["config"]["ranks_started_by"] = os.environ.get("BENCH_LAUNCHED_BY", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "environment")
        if slices is not None:
            # BASELINE config 5 beside the config-4 number: one 2160p stream in 8 slices, slice bands across the same N GPUs
            if "error" in slices:
                out["slices_2160p"] = slices
            else:
                c5 = slices.get("config", {})
                out["slices_2160p"] = {"value": slices.get("value"), "unit": "Mpixels/s", "ms_per_step": slices.get("ms_per_step"), "steps": slices.get("steps"),
                                       "n_gpus": slices.get("n_gpus"), "scaling": "strong", "parallelism": c5.get("parallelism"), "exchange": c5.get("exchange"),
                                       "exchanges_per_step": c5.get("exchanges_per_step"), "exchanges_per_picture": c5.get("exchanges_per_picture"),
                                       "rccl_ranks": c5.get("rccl_ranks"), "search_split_steps_per_step": c5.get("search_split_steps_per_step"), "bitstream_sha256": c5.get("bitstream_sha256"),
                                       "one_rank_sha256": c5.get("one_rank_sha256"), "equals_one_rank_bitstream": c5.get("equals_one_rank_bitstream"),
                                       "leg_wall_s": slices.get("leg_wall_s"), "roofline": slices.get("roofline"),
                                       "note": "measured by a second set of rank processes before the streams run (bench.py slices_leg); control plane gloo, data path RCCL inside the engine"}
    bit_exact = {}
    if extras_on:
        # BASELINE config 3 in the same invocation: the metric names both sizes.  Fewer steps (a pass is ~4x longer).
        args.pcie = False
        wl2 = WORKLOADS["2160p"]
        n2 = min(args.cpu_frames, wl2["frames"]) if check else 0  # all 300 frames by default
        r2, _, host2 = measure(args, torch, dist, venc, "2160p", max(3, args.steps // 2), 2, rank, world, local_rank, want_digest_frames=n2)
        out["2160p"] = {k: r2[k] for k in ("value", "ms_per_step", "steps", "warmup", "per_pass", "roofline", "roofline_valu", "kernel_ms_per_step", "config") if r2.get(k) is not None}
        out["2160p"]["unit"] = "Mpixels/s"
        if check:
            th2, gw2 = cpu_shape((n2 + wl2["keyint"] - 1) // wl2["keyint"])
            dt2, sha2, nb2 = oracle_run(wl2, host_frames(wl2, host2), th2, gw2)
            del host2
            bit_exact["2160p"] = {"frames": n2, "hip_sha256": r2.get("hip_sha256"), "oracle_sha256": sha2, "bytes": nb2,
                                  "equal": r2.get("hip_sha256") == sha2 and r2.get("hip_sha256_bytes") == nb2,
                                  "oracle_Mpx_s": round(n2 * wl2["w"] * wl2["h"] / dt2 / 1e6, 2), "oracle_threads": th2 * gw2}
    if check and rank == 0:
        # all-cores CPU baseline on a bounded sample of the same workload = the oracle side of the bit-exact check
        n = min(args.cpu_frames, wl["frames"])
        kf = (n + wl["keyint"] - 1) // wl["keyint"]
        th, gw = cpu_shape(kf)
        threads = th * gw
        clip_h = host_frames(wl, host)
        dt_c, sha_c, nb_c = oracle_run(wl, clip_h, th, gw)
        n1 = min(wl["keyint"], n)  # one-thread line: one closed GOP
        dt_1, _, _ = oracle_run(wl, clip_h[:n1], 1)
        out["cpu_baseline"] = {
            "value": round(n * wl["w"] * wl["h"] / dt_c / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
            "kind": "port",
            "sample": "first %d frames (%d I + %d P, closed GOPs of %d) of the same %dx%d clip, %d encoders side by side (one per closed GOP) x %d OpenMP threads "
                      "(one per slice) = %d threads: all the independence this stream has (%d GOPs x %d slices), %.1f s" % (
                n, kf, n - kf, wl["keyint"], wl["w"], wl["h"], gw, th, threads, kf, wl["slices"], dt_c),
            "one_thread": {"value": round(n1 * wl["w"] * wl["h"] / dt_1 / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                           "sample": "first %d frames (one closed GOP) of the same clip, 1 thread, %.1f s" % (n1, dt_1)},
            "note": ORACLE_NOTE}
        bit_exact[args.workload] = {"frames": n, "hip_sha256": res.get("hip_sha256"), "oracle_sha256": sha_c, "bytes": nb_c,
                                    "equal": res.get("hip_sha256") == sha_c and res.get("hip_sha256_bytes") == nb_c}
    if rank == 0:
        if bit_exact:
            out["bit_exact"] = all(v["equal"] for v in bit_exact.values())
            out["bit_exact_detail"] = bit_exact
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and bit_exact and not out["bit_exact"]:
        raise SystemExit("bench: the HIP bitstream differs from the CPU oracle's: %s" % json.dumps(bit_exact))


if __name__ == "__main__":
    main()
