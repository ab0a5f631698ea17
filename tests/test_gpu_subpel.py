"""Sub-sample refinement on the GPU (k_subpel + the refined-vector paths of the transform and entropy kernels) against the
oracle (spec/CODEC.md section 2a).  PARITY UNPINNED as everywhere: the oracle is this repo's
restatement (its interpolation is pinned against an independent numpy restatement in tests/test_oracle_subpel.py)."""
import numpy as np
import pytest

from conftest import split_nals, unescape
from subpel_clip import subpel_clip

pytestmark = pytest.mark.gpu


def _clip(venc, kind, w, h, n):
    if kind == "quarter":
        return subpel_clip(w, h, n, factor=4, sx=5, sy=-3)
    if kind == "half":
        return subpel_clip(w, h, n, factor=2, sx=3, sy=1)
    return [venc.synth_frame(w, h, t, 7) for t in range(n)]  # integer pan + moving objects + noise


@pytest.mark.parametrize("w,h,qp,slices,subpel,kind,n", [
    (176, 144, 26, 2, 1, "half", 4),
    (176, 144, 26, 2, 2, "quarter", 4),
    (178, 146, 30, 3, 2, "quarter", 3),      # clamped ingest, ragged macroblocks: windows reach over every border
    (352, 288, 22, 4, 2, "synth", 4),        # vectors up to +-16 at the picture edges
    (64, 48, 34, 1, 2, "half", 3),
    (1920, 1080, 26, 8, 2, "synth", 3),
    (16, 16, 30, 1, 2, "half", 3),          # one macroblock: every window leaves the picture
    (7680, 4320, 32, 8, 2, "synth", 2),     # the largest geometry of the parity tests
])
def test_refined_p_pictures_match_oracle_stage_by_stage(venc, orc, w, h, qp, slices, subpel, kind, n):
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=30, threads=8, subpel=subpel)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, subpel=subpel)
    frac = 0
    for t, (y, u, v) in enumerate(_clip(venc, kind, w, h, n)):
        ref = enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
        eng.sync()
        if t:
            mv_i, _ = eng.mv_sad()
            assert np.array_equal(mv_i, enc.mv()), "full-search vectors frame %d" % t
            mvq, sad = eng.mvq_sad()
            bad = np.argwhere((mvq != enc.mvq()).any(axis=1))
            assert len(bad) == 0, "refined vectors frame %d: mb %d gpu %s oracle %s (%d differ)" % (
                t, bad[0][0], mvq[bad[0][0]], enc.mvq()[bad[0][0]], len(bad))
            assert np.array_equal(sad, enc.sad()), "refined SADs frame %d" % t
            frac += int(((mvq & 3) != 0).any(axis=1).sum())
        lv_g, lv_o = eng.debug(1), enc.levels()
        if not np.array_equal(lv_g, lv_o):
            bad = np.argwhere(lv_g != lv_o)
            raise AssertionError("levels frame %d: first mismatch mb,blk,idx=%s gpu=%d orc=%d (%d diffs)" % (
                t, bad[0], lv_g[tuple(bad[0])], lv_o[tuple(bad[0])], len(bad)))
        rec_o = enc.recon()
        for p in range(3):
            assert np.array_equal(eng.debug(4 + p), rec_o[p]), "recon plane %d frame %d" % (p, t)
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert eng.get_slices(t % 2) == sl_o, "slices frame %d" % t
    if kind != "synth":
        assert frac > 0
    eng.close()


def test_half_sample_planes_match_the_numpy_restatement(venc, orc):
    """k_hpel's b, h, j planes of a whole reference picture against tests/test_oracle_subpel.py's numpy planes (clamped taps at
    every border), at a width that leaves the last 256-sample strip partly empty"""
    from test_oracle_subpel import _numpy_planes
    w, h = 304, 208
    enc = orc.Encoder(w, h, qp=28, n_slices=2, keyint=30, threads=8, subpel=1)
    eng = venc.Engine(w, h, qp=28, n_slices=2, max_items=1, in_slots=2, out_slots=2, subpel=1)
    for t in range(2):
        y, u, v = venc.synth_frame(w, h, t, 3)
        if t == 1:
            ref = enc.recon()[0]  # the I picture's reconstruction = the P step's reference
        enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
        eng.sync()
    got = eng.debug(15)
    _, b, hh, j = _numpy_planes(ref)
    H, W = ref.shape
    for k, (name, want) in enumerate((("b", b), ("h", hh), ("j", j))):
        want = want[4:4 + H, 4:4 + W]
        bad = np.argwhere(got[k] != want)
        assert len(bad) == 0, "%s plane: first mismatch at (y, x) = %s: %d vs %d (%d differ)" % (name, bad[0], got[k][tuple(bad[0])], want[tuple(bad[0])], len(bad))
    eng.close()


def test_refinement_with_gop_batches_rate_control_and_intra_modes(venc, orc):
    """everything the C API's presets switch on at once, through x264_encoder_encode: CRF, Intra16x16 V / H / DC, quarter-sample
    vectors, three GOPs per step"""
    from test_gpu_parity import _x264_stream
    w, h, n, keyint = 176, 144, 14, 4
    frames = subpel_clip(w, h, n, factor=4, sx=3, sy=2)
    got = _x264_stream(venc, frames, w, h, slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), crf=25, gop_batch=3, intra_modes=True, subpel=2)
    enc = orc.Encoder(w, h, qp=25, n_slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True, intra_modes=True, subpel=2)
    dec = orc.Decoder()
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        assert got[t].payload == ref.data, "frame %d" % t
        assert dec.decode(got[t].payload) == 1 and np.array_equal(dec.planes()[0], enc.recon()[0])


def test_refinement_is_refused_in_band_mode_and_absent_by_default(venc):
    eng = venc.Engine(64, 64, qp=26, n_slices=4, subpel=1)
    with pytest.raises(RuntimeError, match="sub-sample"):
        eng.set_band(0, 2)
    eng.close()
    eng = venc.Engine(64, 64, qp=26, n_slices=4)
    with pytest.raises(RuntimeError, match="no refined vectors"):
        eng.debug(14)
    eng.close()
