"""An independent decoder against the oracle ENCODER (tests/pydec.py: numpy only, shares no header, table or helper with oracle/).

The oracle's own decoder shares orc_tables.h / orc_interp.h / orc_deblock.h / orc_intra4.h with its encoder, so their round trips
are blind to a misreading in those headers; this second reading of the standard -- bitstream syntax, CAVLC tables, mv prediction,
P_Skip inference, interpolation, Intra16x16 / Intra4x4 prediction, transforms, loop filter -- is not.  Every picture of every
golden case (incl. cif_all_tools: CRF, Intra4x4, quarter-sample vectors, in-loop filter) and of a stream under rate control must
come out sample for sample as the encoder reconstructed it.  Conformance stays UNPINNED in the strict sense (no published stream
in the image, same authorship); what this removes is the shared-code blind spot."""
import json
import os

import numpy as np
import pytest

import pydec

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(HERE, "golden.json")))


def _check_stream(orc_enc, frames, tag):
    dec = pydec.Decoder()
    for t, f in enumerate(frames):
        fr = orc_enc.encode(*f)
        got = dec.decode(fr.data)
        want = orc_enc.recon()
        for p in range(3):
            w = want[p][:, :got[p].shape[1]]
            assert got[p].shape == w.shape, (got[p].shape, w.shape)
            if not np.array_equal(got[p], w):
                bad = np.argwhere(got[p] != w)
                raise AssertionError("%s frame %d plane %d: %d samples differ, first at (y, x) = %s: decoder %d, encoder %d" % (
                    tag, t, p, len(bad), tuple(bad[0]), got[p][tuple(bad[0])], w[tuple(bad[0])]))
    return dec


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_independent_decoder_reproduces_the_encoders_reconstruction(venc, orc, case):
    enc = orc.Encoder(case["w"], case["h"], qp=case["qp"], n_slices=case["slices"], keyint=case["keyint"], fps=(30, 1), sar=(1, 1), **case.get("tools", {}))
    frames = [venc.synth_frame(case["w"], case["h"], t, case["seed"]) for t in range(case["frames"])]
    dec = _check_stream(enc, frames, case["name"])
    assert dec.size() == (case["w"], case["h"])  # frame cropping of the SPS


def test_independent_decoder_on_tools_one_at_a_time(venc, orc):
    """each optional tool alone, so that a failure names it: Intra16x16 with plane, Intra4x4, half / quarter samples, the filter, CRF"""
    w, h, n = 176, 144, 4
    frames = [venc.synth_frame(w, h, t, 11) for t in range(n)]
    for tools in (dict(intra_modes=2), dict(intra_modes=3), dict(subpel=1), dict(subpel=2), dict(deblock=True), dict(rc=True),
                  dict(intra_modes=3, subpel=2, deblock=True, rc=True)):
        enc = orc.Encoder(w, h, qp=28, n_slices=2, keyint=3, fps=(30, 1), sar=(1, 1), threads=4, **tools)
        _check_stream(enc, frames, str(tools))


def test_independent_decoder_on_noise_and_quantiser_extremes(orc):
    """dense residuals (escape-coded levels, every coeff_token class) at the finest quantiser, sparse ones at the coarsest"""
    from conftest import noise_frames
    for qp in (12, 51):
        enc = orc.Encoder(64, 48, qp=qp, n_slices=2, keyint=2, fps=(30, 1), sar=(1, 1), intra_modes=3, subpel=2, deblock=True)
        _check_stream(enc, noise_frames(64, 48, 3, seed=qp), "noise qp %d" % qp)


def test_vlc_tables_are_complete_prefix_codes():
    """the typed-in tables: prefix-free (checked when the module loads) and complete -- Kraft sum 1 -- wherever the standard's code is:
    every total_zeros and run_before table but the two that reserve their all-zero word (tzVlcIndex 1: 511/512; zerosLeft > 6: 2047/2048)"""
    from fractions import Fraction
    kraft = lambda lens: sum(Fraction(1, 1 << l) for l in lens)
    assert [kraft(t) for t in pydec._TZ_LEN] == [Fraction(511, 512)] + [1] * 14
    assert [kraft(t) for t in pydec._TZC_LEN] == [1, 1, 1]
    assert [kraft(t) for t in pydec._RUN_LEN] == [1] * 6 + [Fraction(2047, 2048)]
    assert sorted(pydec._CBP_INTRA) == list(range(48)) and sorted(pydec._CBP_INTER) == list(range(48))
