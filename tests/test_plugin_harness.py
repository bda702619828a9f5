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


# --------------------------------------------------------------------------- the read-ahead (pipelined) flavour
PIPE = HARNESS_COPY + "_pipe"


def run_pipe(tmp_path, pkts, iw, ih, *opts, depth=None, extra_env=None):
    pk, out = tmp_path / "p.bin", tmp_path / "o.bin"
    write_packets(pk, pkts)
    env = dict(os.environ)
    if depth:
        env["MI_RTJ_DEPTH"] = str(depth)
    env.update(extra_env or {})
    r = subprocess.run([PIPE, str(pk), str(iw), str(ih), str(out)] + list(opts), capture_output=True, text=True, env=env)
    fsz = iw * ih + 2 * ((iw + 1) // 2) * ((ih + 1) // 2)
    raw = np.fromfile(out, dtype=np.uint8) if os.path.exists(out) else np.zeros(0, np.uint8)
    recs = [(raw[i * (fsz + 8): i * (fsz + 8) + fsz], struct.unpack("<q", raw[i * (fsz + 8) + fsz:(i + 1) * (fsz + 8)].tobytes())[0])
            for i in range(raw.size // (fsz + 8))]
    return r, recs


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,iw,ih,key_rate,depth", [(320, 240, 320, 240, 0, None), (320, 240, 314, 234, 4, 2),
                                                      (1920, 1088, 1920, 1080, 5, 3), (160, 128, 160, 121, 3, 16)])
def test_pipelined_decoder_hands_out_the_same_pictures_in_order(tmp_path, w, h, iw, ih, key_rate, depth):
    """packets in flight, pictures out in stream order with their own time stamps; streams with unchanged blocks keep
    their history although several packets are on the device at once"""
    build_harness()
    enc = R.OracleEncoder(w, h, 220, key_rate, 2, 2)
    n = 5 if w >= 1920 else 23
    pkts = [enc.encode(R.synth_frame(w, h, i // 2, seed=21, amp=4)) for i in range(n)]
    r, recs = run_pipe(tmp_path, pkts, iw, ih, depth=depth)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, iw, ih, 0)
    assert len(recs) == len(want)
    for i, ((got, pts), (planes, wpts)) in enumerate(zip(recs, want)):
        assert np.array_equal(got, planes), i
        assert pts == wpts


@pytest.mark.gpu
@pytest.mark.parametrize("idx_group,out_group,depth", [(1, 1, 3), (1, 2, 4), (2, 1, 6), (2, 2, 6), (4, 2, 6), (4, 4, 8), (2, 4, 5)])
@pytest.mark.parametrize("seek", [False, True])
def test_index_and_copy_groups_of_any_size_hand_out_the_same_pictures(tmp_path, idx_group, out_group, depth, seek):
    """the packets of a session are indexed in groups (one launch of the index kernels per group, k_decode packet by
    packet: unchanged blocks come from the predecessor's picture) and their pictures copied out in groups; every
    combination gives the pictures of the packet-by-packet session, also when the stream length is no multiple of the
    group size, and after a seek that leaves the ring in the middle of a group"""
    build_harness()
    w, h = 320, 240
    # (with a seek: an intra-only stream, so that what comes after the seek does not depend on what was dropped)
    enc = R.OracleEncoder(w, h, 220) if seek else R.OracleEncoder(w, h, 220, 4, 2, 2)
    pkts = [enc.encode(R.synth_frame(w, h, i // 2, seed=33, amp=5)) for i in range(23)]
    env = {"MI_RTJ_IDX_GROUP": str(idx_group), "MI_RTJ_OUT_GROUP": str(out_group)}
    opts = ["seek=7:12"] if seek else []  # after 7 pictures the demultiplexer repositions to packet 12
    r, recs = run_pipe(tmp_path, pkts, w, h, *opts, depth=depth, extra_env=env)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    if seek:
        want = want[:7] + want[12:]  # pictures 0..6, then the stream from packet 12 on
    assert len(recs) == len(want)
    for i, ((got, pts), (planes, wpts)) in enumerate(zip(recs, want)):
        assert np.array_equal(got, planes), i
        assert pts == wpts


@pytest.mark.gpu
def test_pipelined_decoder_resync_mid_stream_drops_what_was_read_ahead(tmp_path):
    """a seek after 5 pictures to packet 14 (intra-only stream): pictures 0-4, then 14 onwards — nothing of the
    packets that were in flight when .resync came"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=5, amp=10)) for i in range(20)]
    r, recs = run_pipe(tmp_path, pkts, w, h, "seek=5:14", depth=6)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    order = list(range(5)) + list(range(14, 20))
    assert [pts for _, pts in recs] == [want[i][1] for i in order]
    for (got, _), i in zip(recs, order):
        assert np.array_equal(got, want[i][0]), i


@pytest.mark.gpu
def test_pipelined_decoder_drops_stale_pictures_after_a_packet_skip(tmp_path):
    """bgav_video_skipto on an intra-only stream skips packets at the source and sets s->out_time (lib/video.c:596-612);
    pictures of packets read ahead before that must not come out"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=6, amp=10)) for i in range(20)]
    # after 3 pictures skip to t = 1000 + 40 * 12 + 1: packets whose pts + 40 <= t go, i.e. 0..11
    r, recs = run_pipe(tmp_path, pkts, w, h, f"skippkts=3:{1000 + 40 * 12 + 1}", depth=6)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    order = list(range(3)) + list(range(12, 20))
    assert [pts for _, pts in recs] == [want[i][1] for i in order]
    for (got, _), i in zip(recs, order):
        assert np.array_equal(got, want[i][0]), i


@pytest.mark.gpu
def test_pipelined_decoder_skipto_decodes_every_packet_and_drops_pictures(tmp_path):
    """.skipto on a stream with unchanged blocks: every packet still goes through the decoder (the history must be
    right), pictures before the target are dropped"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 210, 6, 3, 3)
    pkts = [enc.encode(R.synth_frame(w, h, i // 3, seed=7, amp=3)) for i in range(18)]
    assert any((p[12:] == 255).any() for p in pkts)
    r, recs = run_pipe(tmp_path, pkts, w, h, f"skipto=2:{1000 + 40 * 9 + 5}", depth=4)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    order = [0, 1] + list(range(9, 18))
    assert [pts for _, pts in recs] == [want[i][1] for i in order]
    for (got, _), i in zip(recs, order):
        assert np.array_equal(got, want[i][0]), i


@pytest.mark.gpu
def test_pipelined_decoder_refuses_a_packet_of_another_size(tmp_path):
    """a packet whose header announces another picture size than the stream's is a damaged packet: the pictures
    before it come out, then the stream ends (EOF + a log line), nothing is allocated for the claimed size"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=8, amp=10)) for i in range(8)]
    bad = pkts[5].copy()
    bad[6], bad[7], bad[8], bad[9] = 0xF0, 0xFF, 0xF0, 0xFF  # 65520 x 65520
    pkts[5] = bad
    r, recs = run_pipe(tmp_path, pkts, w, h, depth=3)
    assert r.returncode == 0
    assert "does not match the stream's coded size" in r.stderr
    want = expected_stream(pkts[:5], w, h, w, h, 0)
    assert len(recs) == 5
    for (got, _), (planes, _) in zip(recs, want):
        assert np.array_equal(got, planes)


@pytest.mark.gpu
@pytest.mark.parametrize("inside", [4, 5, 7])
def test_pipelined_decoder_skip_target_inside_the_read_ahead_window(tmp_path, inside):
    """with six packets in flight, a skip after 3 pictures to a time inside picture 4, 5 or 7 — all of
    them on the device already — must deliver exactly that picture next, as the synchronous reference decoder does.
    The read-ahead decoder marks its stream GAVL_COMPRESSION_HAS_P_FRAMES, so bgav_video_skipto hands it the exact
    target (.skipto, lib/video.c:614-634) instead of skipping packets behind its back."""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=9, amp=10)) for i in range(20)]
    r, recs = run_pipe(tmp_path, pkts, w, h, f"skipto=3:{1000 + 40 * inside + 7}", depth=6)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    order = list(range(3)) + list(range(inside, 20))
    assert [pts for _, pts in recs] == [want[i][1] for i in order]
    for (got, _), i in zip(recs, order):
        assert np.array_equal(got, want[i][0]), i


@pytest.mark.gpu
def test_source_side_skip_without_a_jump_in_time_stamps_drops_nothing(tmp_path):
    """the defensive rule for a library that skips packets at the source all the same: only a jump in the time
    stamps of the packets read proves that something was skipped; a target inside the read-ahead window skips
    nothing at the source, and no picture may be lost (round 2's rule lost up to depth - 1)"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=10, amp=10)) for i in range(16)]
    r, recs = run_pipe(tmp_path, pkts, w, h, f"skippkts=3:{1000 + 40 * 5 + 1}", depth=6)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    assert [pts for _, pts in recs] == [want[i][1] for i in range(16)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [4, 12])
def test_a_forward_jump_in_the_time_stamps_loses_no_picture(tmp_path, depth):
    """time stamps that jump forward without anybody skipping packets (an empty edit, dropped-frame chunks, a
    fragment gap) are a property of the stream: one picture per packet comes out, as from lib/video_rtjpeg.c.  (The
    decoder marks its stream GAVL_COMPRESSION_HAS_P_FRAMES, so the library never skips packets at the source.)"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=12, amp=10)) for i in range(20)]
    r, recs = run_pipe(tmp_path, pkts, w, h, "ptsjump=7:1000", depth=depth)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    assert len(recs) == 20
    assert [pts for _, pts in recs] == [want[i][1] + (1000 if i >= 7 else 0) for i in range(20)]
    for (got, _), (planes, _) in zip(recs, want):
        assert np.array_equal(got, planes)


@pytest.mark.gpu
def test_a_stream_without_compression_info_drops_what_a_source_side_skip_made_stale(tmp_path):
    """without s->ci the decoder cannot ask the library for .skipto, packets may be skipped at the source, and then a
    jump in the time stamps of the packets read is the proof: what was in flight before it is not shown"""
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=13, amp=10)) for i in range(20)]
    r, recs = run_pipe(tmp_path, pkts, w, h, f"skippkts=3:{1000 + 40 * 12 + 1}", "noci=1", depth=6)
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    order = list(range(3)) + list(range(12, 20))
    assert [pts for _, pts in recs] == [want[i][1] for i in order]


@pytest.mark.gpu
def test_option_keys_come_before_the_environment(tmp_path):
    """device and depth are read from the stream's options dictionary (s->opt, as lib/video_v4l2_m2m.c:66 reads
    BGAV_OPT_VIDEOBUFFER), the environment is the fallback: a device that does not exist fails init whatever
    MI_RTJ_DEVICE says; a depth given as an option decodes the stream like any other"""
    build_harness()
    w, h = 160, 128
    enc = R.OracleEncoder(w, h, 200, 3, 2, 2)
    pkts = [enc.encode(R.synth_frame(w, h, i // 2, seed=11, amp=6)) for i in range(9)]
    pk, out = tmp_path / "p.bin", tmp_path / "o.bin"
    write_packets(pk, pkts)
    env = dict(os.environ, MI_RTJ_DEVICE="0", MI_RTJ_DEPTH="64")
    r = subprocess.run([PIPE, str(pk), str(w), str(h), str(out), "opt=mi355x-device:63"], capture_output=True, text=True, env=env)
    assert r.returncode == 4 and "Cannot open MI355X decoder" in r.stderr
    r, recs = run_pipe(tmp_path, pkts, w, h, "opt=mi355x-depth:2", "opt=mi355x-device:0")
    assert r.returncode == 0, r.stderr
    want = expected_stream(pkts, w, h, w, h, 0)
    assert len(recs) == len(want) and all(np.array_equal(g, p) for (g, _), (p, _) in zip(recs, want))


@pytest.mark.gpu
def test_two_streams_on_two_threads(tmp_path):
    """two decoder instances on two threads of one process (doc/mainpage.incl:54-55), each playing the packet list:
    both deliver every picture (bench mode counts them)"""
    import json
    build_harness()
    w, h = 320, 240
    enc = R.OracleEncoder(w, h, 200)
    pkts = [enc.encode(R.synth_frame(w, h, i, seed=12, amp=10)) for i in range(12)]
    pk = tmp_path / "p.bin"
    write_packets(pk, pkts)
    r = subprocess.run([PIPE, str(pk), str(w), str(h), "/dev/null", "bench=1", "streams=2", "repeat=5"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["streams"] == 2 and d["frames"] == 2 * 12 * 5
