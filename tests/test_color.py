"""Colour stage: RTjpeg_yuv420rgb32/bgr32/rgb24/bgr24/rgb16.  The oracle is pinned
to the reference's own functions where oracle/_ref exists; the HIP kernel is checked against the
oracle (bit-exact) on the GPU box."""
import numpy as np
import pytest

import rtjlib as R
from pkg import P


def random_planes(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)  # full range: exercises both clamps


@pytest.mark.skipif(not R.have_reference(), reason="oracle/_ref not built")
@pytest.mark.parametrize("fmt", range(5))
def test_oracle_matches_reference_helpers(fmt):
    for (w, h) in ((16, 16), (64, 48), (320, 240)):
        planes = random_planes(w, h, fmt * 7 + w)
        pitch = w * R.RGB_BPP[fmt] + 16
        a = np.full(h * pitch, 0xA5, np.uint8)
        b = a.copy()
        R.reference_to_rgb(fmt, w, h, planes, a, pitch)
        R.oracle_to_rgb(fmt, w, h, planes, b, pitch)
        assert np.array_equal(a, b), (fmt, w, h)
        if fmt < 2:  # the fourth byte of every pixel is untouched
            assert (a.reshape(h, pitch)[:, 3:w * 4:4] == 0xA5).all()


def test_oracle_colour_anchors():
    # black, white, and saturated chroma against hand-computed values of the reference's constants
    w = h = 16
    for (yv, cb, cr, rgb) in [(16, 128, 128, (0, 0, 0)), (235, 128, 128, (254, 254, 254)),
                              (128, 255, 128, (130, 80, 255)), (128, 128, 0, (0, 234, 130))]:
        planes = np.concatenate([np.full(w * h, yv, np.uint8), np.full(w * h // 4, cb, np.uint8),
                                 np.full(w * h // 4, cr, np.uint8)])
        out = np.zeros(h * w * 3, np.uint8)
        R.oracle_to_rgb(2, w, h, planes, out, w * 3)
        assert tuple(int(x) for x in out[:3]) == rgb, (yv, cb, cr, out[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", range(5))
def test_hip_colour_stage_matches_oracle(fmt):
    dev = P.MiRtj()
    for (w, h, n) in ((16, 16, 1), (320, 240, 3), (1920, 1088, 2)):
        fsz = w * h * 3 // 2
        in_stride = (fsz + 255) // 256 * 256
        pitch = (w * R.RGB_BPP[fmt] + 15) // 16 * 16 + 32
        out_stride = pitch * h
        host_in = np.zeros(in_stride * n, np.uint8)
        frames = [random_planes(w, h, 100 * fmt + i + w) for i in range(n)]
        for i, f in enumerate(frames):
            host_in[i * in_stride:i * in_stride + fsz] = f
        d_in, d_out = dev.alloc(host_in.size), dev.alloc(out_stride * n)
        dev.h2d(d_in, host_in)
        dev.memset(d_out, 0x3C, out_stride * n)
        dev.to_rgb(fmt, w, h, n, d_in, in_stride, d_out, pitch, out_stride)
        dev.sync()
        got = dev.d2h(d_out, out_stride * n)
        for i, f in enumerate(frames):
            want = np.full(out_stride, 0x3C, np.uint8)
            R.oracle_to_rgb(fmt, w, h, f, want, pitch)
            assert np.array_equal(got[i * out_stride:(i + 1) * out_stride], want), (fmt, w, h, i)
        dev.free(d_in)
        dev.free(d_out)
    with pytest.raises(P.MiRtjError, match="aligned|bad argument"):
        dev.to_rgb(fmt, 24, 16, 1, 0, 0, 0, 0, 0)
    dev.close()
