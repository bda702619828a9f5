"""BASELINE.json's other configurations as parity cases (bench.py times only configs[1]):
cfg 4 — 3840x2160 streams decoded in order, one stream per rank; cfg 5 — a mixed batch of 64
intra-only streams (320x240 / 1080p / 4K, Q in {64,128,255}) dealt frame by frame to 8 ranks with
every frame checked against the CPU oracle.  Ranks are emulated one after the other on the one GPU
of the test box, through the same partition code (shard.py) the multi-GPU bench uses."""
import importlib

import numpy as np
import pytest

import rtjlib as R
from pkg import P

pytestmark = pytest.mark.gpu
shard = importlib.import_module("gmerlin-avdecoder_amd.shard")


def fb(w, h):
    return w * h * 3 // 2


def test_cfg4_4k_stream_in_order_with_unchanged_blocks():
    w, h, Q = 3840, 2160, 255  # 2160 is already a multiple of 16
    n = 4
    frames = [R.synth_frame(w, h, i // 2, seed=12345, amp=8) for i in range(n)]
    dev = P.MiRtj()
    d_fr = dev.alloc(fb(w, h) * n)
    dev.h2d(d_fr, np.concatenate(frames))
    # the stream is made on the device too (inter mode), then decoded packet by packet in order
    d_st, po, pl = dev.encode(w, h, Q, n, d_fr, key_rate=3, lmask=2, cmask=2)
    dev.sync()
    enc, od = R.OracleEncoder(w, h, Q, 3, 2, 2), R.OracleDecoder()
    got, want = np.zeros(fb(w, h), np.uint8), np.zeros(fb(w, h), np.uint8)
    dec = P.MiRtj()
    for i in range(n):
        pkt = dev.d2h(d_st, int(pl[i]), offset=int(po[i]))
        assert np.array_equal(pkt, enc.encode(frames[i])), i
        dec.decode(pkt, got)
        od.decode(pkt, want)
        assert np.array_equal(got, want), i
    dec.close()
    dev.close()


def test_cfg5_mixed_streams_frame_scatter_over_8_ranks():
    geoms = [(320, 240), (1920, 1088), (320, 240), (3840, 2160)]
    streams = []  # 64 intra-only streams, 2 frames each
    for sidx in range(64):
        w, h = geoms[sidx % 4] if sidx % 16 else (3840, 2160)
        if (w, h) == (3840, 2160) and sidx % 16:
            w, h = 1920, 1088  # keep the 4K share small: 4 of 64 streams
        Q = (64, 128, 255)[sidx % 3]
        enc = R.OracleEncoder(w, h, Q)
        streams.append([enc.encode(R.synth_frame(w, h, k, seed=sidx, amp=(8, 40)[sidx % 2])) for k in range(2)])
    frames = [p for st in streams for p in st]  # global frame list, stream-major
    world = 8
    checked = 0
    for rank in range(world):
        mine = shard.frames_for_rank(len(frames), rank, world, "cyclic")
        dev = P.MiRtj()  # one instance per rank, as one process per GPU would have
        pkts = [frames[k] for k in mine]
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        sizes = [fb(int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)) for p in pkts]
        oo = np.cumsum([0] + [(s + 255) // 256 * 256 for s in sizes[:-1]]).astype(np.uint64)
        d_out = dev.alloc(int(oo[-1]) + sizes[-1])
        plan = dev.plan(hdrs, po, pl, oo)
        plan.decode(d_stream, d_out)
        dev.sync()
        for j, k in enumerate(mine):
            got = dev.d2h(d_out, sizes[j], offset=int(oo[j]))
            want = np.zeros(sizes[j], np.uint8)
            R.OracleDecoder().decode(frames[k], want)
            assert R.digest(got) == R.digest(want), (rank, k)
            checked += 1
        dev.close()
    assert checked == len(frames) == 128


def test_cfg5_from_movie_files(tmp_path):
    """The same scatter with the packets coming out of container files (QuickTime RTJ0 movies written
    by include/mi_qtrtj.h from GPU-encoded streams), as BASELINE's 'mixed batch of files' has it."""
    import ctypes as C
    from test_qt_rtj0_host import Sample, write_movie
    import subprocess, os
    from pkg import ROOT
    subprocess.run(["make", "-C", os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")], check=True, capture_output=True)
    qt = C.CDLL(os.path.join(ROOT, "gmerlin-avdecoder_amd", "lib", "libmi_qtrtj.so"))
    qt.mi_qt_writer_open.restype = C.c_void_p
    qt.mi_qt_writer_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
    qt.mi_qt_writer_add.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int]
    qt.mi_qt_writer_close.argtypes = [C.c_void_p]
    qt.mi_qt_reader_open.restype = C.c_void_p
    qt.mi_qt_reader_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    qt.mi_qt_reader_close.argtypes = [C.c_void_p]
    qt.mi_qt_reader_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    qt.mi_qt_reader_sample.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Sample)]
    qt.mi_qt_reader_read.restype = C.c_long
    qt.mi_qt_reader_read.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
    gen = P.MiRtj()
    files = []
    for k, (w, h, Q, n) in enumerate([(320, 240, 64, 5), (1920, 1088, 255, 3), (640, 368, 128, 4), (320, 240, 255, 6),
                                      (1920, 1088, 128, 2), (176, 144, 200, 7)]):
        d_fr = gen.synth(w, h, 10 * k, n, seed=k, amp=6)
        d_st, po, pl = gen.encode(w, h, Q, n, d_fr)  # intra-only batch: every packet a key frame
        gen.sync()
        pkts = [gen.d2h(d_st, int(pl[i]), offset=int(po[i])) for i in range(n)]
        path = tmp_path / f"clip{k}.mov"
        write_movie(qt, path, pkts, w, h - (8 if h == 1088 else 0))
        files.append(path)
        gen.free(d_fr)
        gen.free(d_st)
    gen.close()
    frames = []  # (packet) in file order
    for path in files:
        r = qt.mi_qt_reader_open(str(path).encode(), None, 0)
        assert r
        fc, n = C.c_uint32(), C.c_uint64()
        qt.mi_qt_reader_info(r, fc, None, None, None, n)
        assert fc.value == 0x52544A30
        for i in range(n.value):
            s = Sample()
            qt.mi_qt_reader_sample(r, i, s)
            buf = C.create_string_buffer(s.size)
            assert qt.mi_qt_reader_read(r, i, buf, s.size) == s.size
            frames.append(np.frombuffer(buf.raw, dtype=np.uint8).copy())
        qt.mi_qt_reader_close(r)
    assert len(frames) == 27
    world = 4
    for rank in range(world):
        mine = shard.frames_for_rank(len(frames), rank, world, "cyclic")
        dev = P.MiRtj()
        pkts = [frames[k] for k in mine]
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        sizes = [fb(int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)) for p in pkts]
        oo = np.cumsum([0] + [(s + 255) // 256 * 256 for s in sizes[:-1]]).astype(np.uint64)
        d_out = dev.alloc(int(oo[-1]) + sizes[-1])
        plan = dev.plan(hdrs, po, pl, oo)
        plan.decode(d_stream, d_out)
        dev.sync()
        for j, k in enumerate(mine):
            want = np.zeros(sizes[j], np.uint8)
            R.OracleDecoder().decode(frames[k], want)
            assert np.array_equal(dev.d2h(d_out, sizes[j], offset=int(oo[j])), want), (rank, k)
        dev.close()
