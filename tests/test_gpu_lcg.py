"""SURVEY.md section 8d's content (the LCG generator of BASELINE.md section 2's CPU probe) on the device: the GPU generator
makes the frames of the numpy twin, the GPU encoder the reference encoder's packets, the GPU decoder the reference
decoder's planes — digests made by the reference's own lib/RTjpeg.c (tests/golden/make_lcg_golden.py)."""
import json
import os

import numpy as np
import pytest

import rtjlib as R
from pkg import P, ROOT

G = json.load(open(os.path.join(ROOT, "tests", "golden", "lcg_golden.json")))


def test_twin_matches_the_golden_frames():
    for c in G["frames"]:
        if c["w"] * c["h"] <= 640 * 368:
            assert R.digest(R.synth_frame_lcg(c["w"], c["h"], c["n"], amp=c["amp"])) == c["frame"]


@pytest.mark.gpu
def test_generator_encoder_decoder_on_lcg_content():
    dev = P.MiRtj()
    for c in G["frames"]:
        w, h, n, fsz = c["w"], c["h"], c["n"], c["w"] * c["h"] * 3 // 2
        d_fr = dev.synth_lcg(w, h, n, 1, seed=12345, amp=c["amp"])  # frame n of the one sequence
        dev.sync()
        assert R.digest(dev.d2h(d_fr, fsz)) == c["frame"], c
        d_st, po, pl = dev.encode(w, h, c["Q"], 1, d_fr, align=1)
        dev.sync()
        pkt = dev.d2h(d_st, int(pl[0]), offset=int(po[0]))
        assert pkt.size == c["packet_bytes"] and R.digest(pkt) == c["packet"], c
        got = np.zeros(fsz, np.uint8)
        dev.decode(pkt, got)
        assert R.digest(got) == c["planes"], c
        dev.free(d_fr)
        dev.free(d_st)
    # frames made in two passes continue the one sequence
    w, h, fsz = 320, 240, 320 * 240 * 3 // 2
    d = dev.synth_lcg(w, h, 0, 8, amp=8)
    dev.sync()
    whole = dev.d2h(d, 8 * fsz)
    assert R.digest(whole[7 * fsz:]) == [c for c in G["frames"] if c["w"] == 320 and c["n"] == 7][0]["frame"]
    dev.free(d)
    dev.close()
