"""In-loop deblocking on the GPU (k_deblock, a slice-wise wavefront) against the oracle (spec/CODEC.md section 5a;
SURVEY.md section 8f row f3).  PARITY AND CONFORMANCE UNPINNED as in tests/test_oracle_deblock.py, which pins the oracle's
filter against a second restatement."""
import numpy as np
import pytest

from conftest import noise_frames, split_nals, unescape

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,qp,slices,subpel,n,kind", [
    (176, 144, 30, 2, 0, 4, "noise"),
    (176, 144, 38, 3, 2, 4, "noise"),
    (178, 146, 26, 1, 0, 3, "synth"),      # one slice of ten rows, ragged edges
    (352, 288, 34, 4, 1, 4, "synth"),
    (64, 48, 44, 1, 0, 3, "noise"),
    (1920, 1080, 30, 8, 0, 3, "synth"),
    (640, 1104, 36, 2, 0, 2, "noise"),     # 35-row slices: more rows than waves in a block
    (16, 16, 40, 1, 2, 3, "noise"),        # one macroblock: internal edges only, every sub-sample window on the rim
    (1920, 1080, 34, 1, 0, 2, "synth"),    # one slice of 68 rows: five rounds of the block's waves
    (3840, 2160, 30, 8, 2, 2, "synth"),    # BASELINE config 3 geometry with refinement and filter
])
def test_deblocked_pictures_match_oracle_stage_by_stage(venc, orc, w, h, qp, slices, subpel, n, kind):
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=30, threads=8, subpel=subpel, deblock=True)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, subpel=subpel, deblock=True)
    frames = noise_frames(w, h, n, seed=5) if kind == "noise" else [venc.synth_frame(w, h, t, 9) for t in range(n)]
    changed = 0
    for t, (y, u, v) in enumerate(frames):
        ref = enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
        eng.sync()
        rec_o = enc.recon()
        for p in range(3):
            got = eng.debug(4 + p)
            bad = np.argwhere(got != rec_o[p])
            assert len(bad) == 0, "recon plane %d frame %d: first mismatch at (y, x) = %s gpu %d oracle %d (%d differ)" % (
                p, t, bad[0], got[tuple(bad[0])], rec_o[p][tuple(bad[0])], len(bad))
        lv_g, lv_o = eng.debug(1), enc.levels()
        assert np.array_equal(lv_g, lv_o), "levels frame %d" % t
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert eng.get_slices(t % 2) == sl_o, "slices frame %d" % t
        changed += int(sum(np.count_nonzero(a != b) for a, b in zip(enc.recon_unfiltered(), rec_o)))
    assert changed > 0
    eng.close()


def test_deblocking_with_gop_batches_and_every_other_tool(venc, orc):
    """lock-step GOPs, CRF, Intra16x16 V / H / DC, quarter-sample vectors and the in-loop filter through x264_encoder_encode"""
    from test_gpu_parity import _x264_stream
    from subpel_clip import subpel_clip
    w, h, n, keyint = 176, 144, 14, 4
    frames = subpel_clip(w, h, n, factor=4, sx=3, sy=2)
    got = _x264_stream(venc, frames, w, h, slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), crf=30, gop_batch=3, intra_modes=True, subpel=2, deblock=True)
    enc = orc.Encoder(w, h, qp=30, n_slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True, intra_modes=True, subpel=2, deblock=True)
    dec = orc.Decoder()
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        assert got[t].payload == ref.data, "frame %d" % t
        assert dec.decode(got[t].payload) == 1
        for a, b in zip(dec.planes(), enc.recon()):
            assert np.array_equal(a, b)
