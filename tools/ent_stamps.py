#!/usr/bin/env python3
"""Phase shares of k_entropy_p from in-kernel 100 MHz clock stamps (diagnostic build only).
Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVENC_ENT_STAMPS \\
              -o video-encoder_amd/csrc/libvenc_hip_stamps.so video-encoder_amd/csrc/venc_hip.hip -ldl
Run:    VENC_HIP_LIB=libvenc_hip_stamps.so [VENC_NO_OVERLAP=1] python tools/ent_stamps.py [1080p|2160p]
One pass of the bench's resident clip (ten GOPs in lock-step); sums over all waves of all k_entropy_p launches."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
w, h = (3840, 2160) if len(sys.argv) > 1 and sys.argv[1] == "2160p" else (1920, 1080)
frames, fb = 300, w * h * 3 // 2
clip = torch.empty(frames * fb, dtype=torch.uint8, device="cuda")
assert venc.hip_lib().venc_synth_device(clip.data_ptr(), fb, w, h, 0, frames, 0x1234ABCD) == 0
torch.cuda.synchronize()
enc = venc.X264Encoder(w, h, qp=26, slices=8, keyint=30, fps=(30, 1), sar=(1, 1), gop_batch=10)
enc.encode_resident_tally(clip.data_ptr(), fb, frames, repeats=1)
torch.cuda.synchronize()
eng = venc.Engine.view(venc.x264_lib().x264_venc_engine(enc.h))
names = ["A1 records, P_Skip, masks, ranks (+ look-back)", "   barrier", "A2 nC, H strings, lists", "   barrier", "B  R entries",
         "B  blocks with levels (one, then more)", "   barrier", "C  offsets"]
raw = eng.debug(11)
n = int(min(raw[0], 16384 // 4))
rec = raw[8:8 + 8 * n].reshape(n, 8).astype(np.int64)
start, ents = rec[:, 0], rec[:, 1] >> 40
d = rec[:, 1:].copy(); d[:, 0] &= (1 << 40) - 1
life = d.sum(axis=1)
names = ["A1 records, P_Skip, masks, ranks (+ look-back)", "   barrier", "A2 nC, H strings, lists", "   barrier", "B  R entries",
         "B  blocks with levels (one level, then more)", "   barrier + C offsets"]
print("%d blocks (first wave of each); 100 MHz ticks = 10 ns" % n)
for k, nm in enumerate(names):
    print("%-48s mean %7.1f us  p50 %7.1f  p90 %7.1f  max %7.1f   %5.1f %%" % (nm, d[:, k].mean() / 100, np.percentile(d[:, k], 50) / 100, np.percentile(d[:, k], 90) / 100, d[:, k].max() / 100, 100 * d[:, k].sum() / life.sum()))
print("block lifetime: mean %.1f us p10 %.1f p50 %.1f p90 %.1f max %.1f; entries p50 %d p90 %d max %d; corr(lifetime, entries) %.2f" %
      (life.mean() / 100, np.percentile(life, 10) / 100, np.percentile(life, 50) / 100, np.percentile(life, 90) / 100, life.max() / 100, np.percentile(ents, 50), np.percentile(ents, 90), ents.max(), np.corrcoef(life, ents)[0, 1]))
order = np.argsort(start); s0 = start[order] - start.min(); e0 = s0 + life[order]
# launches: a start more than 30 us after the previous block's start opens a new one
cuts = [0] + list(np.where(np.diff(s0) > 3000)[0] + 1) + [n]
for a, b in zip(cuts[:-1], cuts[1:]):
    if b - a < 8: continue
    print("launch: %4d blocks, starts spread over %6.1f us, last end %6.1f us after the first start; lifetime p50 %5.1f max %5.1f" %
          (b - a, (s0[b - 1] - s0[a]) / 100, (e0[a:b].max() - s0[a]) / 100, np.percentile(e0[a:b] - s0[a:b], 50) / 100, (e0[a:b] - s0[a:b]).max() / 100))
enc.close()
