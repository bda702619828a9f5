"""GPU parity of the DV25 525/60 decoder (libmi_dv.so through its C ABI) against oracle/dv_oracle.c, bit for bit.
PARITY UNPINNED (no DV pixel decoder in the reference: lib/dvframe.c:663-676 hands the frame to libavcodec): the
oracle is this repository's statement of the format."""
import importlib

import numpy as np
import pytest

import dvlib as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    dv = importlib.import_module("gmerlin-avdecoder_amd.dv")
    d = dv.MiDv(0)
    yield d
    d.close()


def same(dev, frames):
    frames = np.ascontiguousarray(frames).reshape(-1, D.FRAME_BYTES)
    got = dev.decode_frames(frames)
    for i, f in enumerate(frames):
        want = D.decode(f)
        if not np.array_equal(got[i], want):
            bad = np.flatnonzero(got[i] != want)
            raise AssertionError(f"frame {i}: {bad.size} bytes differ, first at {bad[0]} (got {got[i][bad[0]]}, want {want[bad[0]]})")


@pytest.mark.parametrize("amp,flags", [(0, 0), (2, 3), (8, 3), (12, 3), (24, 1), (40, 3), (90, 2)])
def test_encoded_pictures_decode_like_the_oracle(dev, amp, flags):
    """flat to very noisy content: few to many blocks that overflow into the macroblock's and the segment's space;
    both transform modes; classes by block content"""
    same(dev, [D.encode(D.synth(n, 3 + amp, amp), flags) for n in range(3)])


def test_every_class_quantisation_number_and_mode(dev):
    """the header bits of an encoded frame rewritten at random: the words then mean other coefficients, both decoders
    must still agree (class 3's doubling, all 16 quantisation numbers, 2-4-8 blocks everywhere)"""
    rng = np.random.default_rng(5)
    frames = []
    for n in range(4):
        dif = D.encode(D.synth(n, 21, 10), 3).copy()
        for seq in range(10):
            for v in range(135):
                o = D.video_block_offset(seq, v)
                dif[o + 3] = rng.integers(0, 256)  # STA | QNO
                for a in D.AREA_OFF:
                    dif[o + a + 1] = (dif[o + a + 1] & 0x8F) | (rng.integers(0, 8) << 4)  # mode, class
        frames.append(dif)
    same(dev, frames)


def test_arbitrary_bytes_decode_like_the_oracle(dev):
    """no frame is refused: random bytes are long runs of escapes, cut-off words in every pass, runs past the last
    coefficient, blocks that never finish"""
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (6, D.FRAME_BYTES), dtype=np.uint8)
    frames[1] = 0
    frames[2] = 0xFF
    frames[3, ::2] = 0x7F
    same(dev, frames)


def test_biased_bits_decode_like_the_oracle(dev):
    """bytes with mostly-zero / mostly-one bits: short words (many coefficients per block) and escapes (few)"""
    rng = np.random.default_rng(12)
    frames = []
    for pr in (0.1, 0.3, 0.7, 0.9):
        bits = (rng.random(D.FRAME_BYTES * 8) < pr).astype(np.uint8)
        frames.append(np.packbits(bits))
    same(dev, frames)


def test_a_batch_of_frames_and_the_one_frame_entry_point(dev):
    frames = [D.encode(D.synth(n, 2, 6 + n % 5), 3) for n in range(24)]
    same(dev, frames)
    # mi_dv_decode_frame: the caller's planes with the caller's strides
    planes = dev.decode_frame(frames[5], strides=(768, 200, 192))
    want = D.decode(frames[5])
    y = planes[0].reshape(480, 768)[:, :720]
    cb = planes[1].reshape(480, 200)[:, :180]
    cr = planes[2].reshape(480, 192)[:, :180]
    assert np.array_equal(y.ravel(), want[:720 * 480])
    assert np.array_equal(cb.ravel(), want[720 * 480:720 * 480 + 180 * 480])
    assert np.array_equal(cr.ravel(), want[720 * 480 + 180 * 480:])


def test_a_frame_of_another_system_is_refused_by_the_one_frame_entry_point(dev):
    dv = importlib.import_module("gmerlin-avdecoder_amd.dv")
    f = D.encode(D.synth(0, 1, 4), 0).copy()
    f[3] |= 0x80  # DSF: 625/50
    with pytest.raises(dv.MiDvError):
        dev.decode_frame(f)


def test_dv_through_the_plugin_seam(tmp_path):
    """csrc/video_dv_mi355x.c registered in front of the RTjpeg decoder, driven by the harness the way lib/video.c drives
    a bgav_video_decoder_t: fourcc 'dvc ', packets = DIF frames (lib/dvframe.c:663-676), pictures into the caller's
    strided planes; a frame of another system ends the stream with a log line"""
    import os
    import struct
    import subprocess
    from pkg import ROOT
    subprocess.run(["make", "-C", os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "gmerlin-avdecoder_amd", "lib", "plugin_harness_dv")
    frames = [D.encode(D.synth(n, 4, 6), 3) for n in range(5)]
    bad = frames[3].copy()
    bad[3] |= 0x80
    pk, out = tmp_path / "p.bin", tmp_path / "o.bin"
    with open(pk, "wb") as f:
        for fr in frames[:3] + [bad] + frames[4:]:
            f.write(struct.pack("<I", fr.size))
            f.write(fr.tobytes())
    r = subprocess.run([exe, str(pk), "720", "480", str(out), "fourcc=dvc "], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "DV video decoder (MI355X)" in r.stderr and "format DV" in r.stderr
    assert "not a 525/60" in r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    rec = D.PICTURE_BYTES + 8
    assert raw.size == 3 * rec  # the pictures before the damaged frame
    for i in range(3):
        assert np.array_equal(raw[i * rec:i * rec + D.PICTURE_BYTES], D.decode(frames[i])), i
    # a 625/50-sized stream is not this decoder's: the probe declines and (here) nothing else takes the fourcc
    r = subprocess.run([exe, str(pk), "720", "576", str(out), "fourcc=dvcp"], capture_output=True, text=True)
    assert r.returncode == 3
