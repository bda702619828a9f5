"""The C plugin wrapper (gmerlin-avdecoder_amd/csrc/video_rtjpeg_mi355x.c — the replacement for the
reference's lib/video_rtjpeg.c) driven the way lib/video.c drives a bgav_video_decoder_t, by
tests/harness/plugin_harness.c.  Host code is C end to end; Python only prepares packets and
compares planes with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import rtjlib as R
from pkg import ROOT

HARNESS = HARNESS_COPY = os.path.join(ROOT, "gmerlin-avdecoder_amd", "lib", "plugin_harness")


def build_harness():
    subprocess.run(["make", "-C", os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")], check=True,
                   capture_output=True)
    assert os.path.exists(HARNESS)


def write_packets(path, pkts):
    with open(path, "wb") as fh:
        for p in pkts:
            fh.write(struct.pack("<I", p.size))
            fh.write(p.tobytes())


def test_wrapper_builds_and_registers_rtj0(tmp_path):
    build_harness()
    src = open(os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc", "video_rtjpeg_mi355x.c")).read()
    # the symbols the reference's build expects from this translation unit
    assert "void bgav_init_video_decoders_rtjpeg(void)" in src
    assert "BGAV_MK_FOURCC('R', 'T', 'J', '0')" in src
    out = subprocess.run(["nm", "-D", "--defined-only", HARNESS], capture_output=True, text=True).stdout + \
        subprocess.run(["nm", HARNESS], capture_output=True, text=True).stdout
    assert "bgav_init_video_decoders_rtjpeg" in out


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_without_a_gpu_the_probe_declines_and_nothing_decodes(tmp_path):
    build_harness()
    w, h = 64, 48
    pk = tmp_path / "p.bin"
    write_packets(pk, [R.OracleEncoder(w, h, 200).encode(R.synth_frame(w, h, 0))])
    r = subprocess.run([HARNESS, str(pk), str(w), str(h), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "no video decoder accepted" in r.stderr


def expected_stream(pkts, w, h, iw, ih, skip_every):
    dec = R.OracleDecoder()
    priv = np.zeros(w * h * 3 // 2, np.uint8)
    frames, k = [], 0
    for i, p in enumerate(pkts):
        k += 1
        if skip_every and k % skip_every == 0:
            continue  # decode_rtjpeg consumes the packet and decodes nothing (lib/video_rtjpeg.c:75-79)
        dec.decode(p, priv)
        y = priv[: w * h].reshape(h, w)[:ih, :iw]
        u = priv[w * h: w * h * 5 // 4].reshape(h // 2, w // 2)[: (ih + 1) // 2, : (iw + 1) // 2]
        v = priv[w * h * 5 // 4:].reshape(h // 2, w // 2)[: (ih + 1) // 2, : (iw + 1) // 2]
        frames.append((np.concatenate([y.ravel(), u.ravel(), v.ravel()]), 1000 + 40 * i))
    return frames


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,iw,ih,key_rate,skip_every,nocopy",
                         [(320, 240, 320, 240, 0, 0, 0), (320, 240, 314, 234, 4, 0, 0), (1920, 1088, 1920, 1080, 0, 0, 0),
                          (160, 128, 160, 121, 3, 3, 0), (320, 240, 314, 234, 4, 0, 1), (1920, 1088, 1920, 1080, 0, 0, 1)])
def test_wrapper_decodes_like_the_reference_wrapper(tmp_path, w, h, iw, ih, key_rate, skip_every, nocopy):
    build_harness()
    HARNESS = HARNESS_COPY + ("_nocopy" if nocopy else "")
    enc = R.OracleEncoder(w, h, 220, key_rate, 2, 2)
    n = 3 if w >= 1920 else 7
    pkts = [enc.encode(R.synth_frame(w, h, i // 2, seed=21, amp=4)) for i in range(n)]
    pk, out = tmp_path / "p.bin", tmp_path / "o.bin"
    write_packets(pk, pkts)
    r = subprocess.run([HARNESS, str(pk), str(iw), str(ih), str(out)] + ([str(skip_every)] if skip_every else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "format RTjpeg" in r.stderr and f"frame {w}x{h}" in r.stderr
    want = expected_stream(pkts, w, h, iw, ih, skip_every)
    fsz = iw * ih + 2 * ((iw + 1) // 2) * ((ih + 1) // 2)
    raw = np.fromfile(out, dtype=np.uint8)
    assert raw.size == len(want) * (fsz + 8)
    for i, (planes, pts) in enumerate(want):
        rec = raw[i * (fsz + 8):(i + 1) * (fsz + 8)]
        assert np.array_equal(rec[:fsz], planes), i
        assert struct.unpack("<q", rec[fsz:].tobytes())[0] == pts
