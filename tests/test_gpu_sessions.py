"""Pipelined sessions (mi_rtj_pipe_*, include/mi_rtjpeg.h) driven directly through the C ABI: packets of a stream in
order, several in flight, indexed and copied out in groups — against the CPU oracle decoding the same packets one after
the other into one picture (lib/video_rtjpeg.c:81-82 + lib/RTjpeg.c:3565-3586).  Bit-exact."""
import os

import numpy as np
import pytest

import rtjlib as R
from pkg import P

pytestmark = pytest.mark.gpu


def oracle_stream(pkts, w, h):
    dec = R.OracleDecoder()
    pic = np.zeros(w * h * 3 // 2, np.uint8)
    out = []
    for p in pkts:
        dec.decode(p, pic)
        out.append(pic.copy())
    return out


def make_stream(rng, w, h, n):
    """packets as a capture makes them, and as it does not: quality changes inside the stream (tables with and without
    lb8 == cb8), key frames and frames of nothing but unchanged blocks, a header that says quality 0, packets cut
    short, a packet that is only a header, noise"""
    pkts = []
    enc = R.OracleEncoder(w, h, int(rng.choice([255, 200, 128, 64, 20])), int(rng.integers(0, 6)), 2, 2)
    for i in range(n):
        kind = int(rng.integers(0, 12))
        if kind == 0:  # the stream's quality changes: a new encoder (its first frame is whole)
            enc = R.OracleEncoder(w, h, int(rng.choice([255, 230, 171, 170, 100, 1])), int(rng.integers(0, 6)), 2, 2)
        amp = int(rng.choice([0, 4, 8, 30]))
        p = enc.encode(R.synth_frame(w, h, i // 2 if kind != 1 else i, seed=int(rng.integers(1, 1000)) if kind == 2 else 7, amp=amp))
        if kind == 3:
            p = p[: max(12, int(rng.integers(12, p.size + 1)))].copy()  # cut short (bytes past the end read as 0)
        elif kind == 4:
            p = p.copy()
            p[10] = 0  # a header that says quality 0: the tables stay what they were (lib/RTjpeg.c:3577)
        elif kind == 5:
            p = np.concatenate([p[:12], rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8)])  # arbitrary payload
        pkts.append(np.ascontiguousarray(p))
    return pkts


@pytest.mark.parametrize("idx_group,out_group,depth", [(1, 1, 2), (1, 2, 4), (2, 2, 4), (2, 2, 6), (4, 2, 8), (4, 4, 12),
                                                        (2, 4, 12), (0, 0, 12), (0, 0, 6)])  # 0: the library's default
@pytest.mark.parametrize("w,h", [(64, 48), (320, 240)])
def test_sessions_in_groups_decode_like_one_packet_after_the_other(monkeypatch, idx_group, out_group, depth, w, h):
    if idx_group:
        monkeypatch.setenv("MI_RTJ_IDX_GROUP", str(idx_group))
    if out_group:
        monkeypatch.setenv("MI_RTJ_OUT_GROUP", str(out_group))
    rng = np.random.default_rng(1000 * idx_group + 100 * out_group + depth + w)
    dev = P.MiRtj()
    pkts = make_stream(rng, w, h, 61)
    want = oracle_stream(pkts, w, h)
    pipe = dev.pipe(depth=depth, coded_w=w, coded_h=h)
    got, nxt = 0, 0
    while got < len(pkts):
        # the caller's habits vary: fill the pipeline, or ask for a picture while its group is still incomplete
        burst = int(rng.integers(1, depth + 1))
        while nxt < len(pkts) and pipe.room() > 0 and burst > 0:
            pipe.submit(pkts[nxt], nxt)
            nxt += 1
            burst -= 1
        take = int(rng.integers(1, 4))
        while take > 0 and pipe.pending() > 0:
            y, u, v, tag = pipe.next()
            assert tag == got
            pic = np.concatenate([y, u, v])
            d = np.nonzero(pic != want[got])[0]
            assert d.size == 0, (got, int(d[0]), d.size)
            got += 1
            take -= 1
    assert pipe.pending() == 0
    pipe.close()
    dev.close()


@pytest.mark.parametrize("depth", [6, 12])
def test_flush_in_the_middle_of_a_group_and_pictures_dropped_unseen(depth):
    """a seek (mi_rtj_pipe_flush) forgets what is in flight, staged packets included, and leaves the ring in the middle
    of a group; pictures taken with planes == NULL (the wrapper's .skipto) still go through the decoder"""
    w, h = 160, 128
    rng = np.random.default_rng(5)
    dev = P.MiRtj()
    enc = R.OracleEncoder(w, h, 220, 0, 2, 2)  # no key frames: every picture depends on its predecessors
    pkts = [enc.encode(R.synth_frame(w, h, i // 3, seed=9, amp=6)) for i in range(40)]
    pipe = dev.pipe(depth=depth, coded_w=w, coded_h=h)
    dec = R.OracleDecoder()
    pic = np.zeros(w * h * 3 // 2, np.uint8)
    fed = 0

    def feed(k):
        nonlocal fed
        for _ in range(k):
            pipe.submit(pkts[fed], fed)
            fed += 1

    def take(drop=False):
        if drop:
            return pipe.next(drop=True)
        y, u, v, tag = pipe.next()
        return np.concatenate([y, u, v]), tag

    # 1. three pictures out of five packets, then a flush with two in flight (one of them possibly only staged)
    feed(5)
    for i in range(3):
        dec.decode(pkts[i], pic)
        got, tag = take()
        assert tag == i and np.array_equal(got, pic)
    pipe.flush()
    assert pipe.pending() == 0
    # what was in flight went through the decoder or not — the reference's decoder would not have seen packets 3, 4
    # at all after a seek; the next packet here is whole (a new encoder's first frame), so history does not matter
    enc2 = R.OracleEncoder(w, h, 220, 0, 2, 2)
    tail = [enc2.encode(R.synth_frame(w, h, 100 + i // 2, seed=11, amp=6)) for i in range(15)]
    dec.decode(tail[0], pic)
    pipe.submit(tail[0], 1000)
    got, tag = take()
    assert tag == 1000 and np.array_equal(got, pic)
    # 2. drop pictures unseen: their packets still count
    for i in range(1, 5):
        pipe.submit(tail[i], 1000 + i)
    for i in range(1, 5):
        dec.decode(tail[i], pic)
        assert take(drop=True) == 1000 + i
    for i in range(5, 8):
        pipe.submit(tail[i], 1000 + i)
    dec.decode(tail[5], pic)
    assert take(drop=True) == 1005
    for i in range(6, 8):
        dec.decode(tail[i], pic)
        got, tag = take()
        assert tag == 1000 + i and np.array_equal(got, pic), i
    pipe.close()
    dev.close()
