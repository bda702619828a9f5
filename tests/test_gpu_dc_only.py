"""Pictures built around DC-only blocks ("DC, zeros, one run": the pixel clamp((int16(DC * q0) + 4) >> 3) 64 times
over, lib/RTjpeg.c:2223-2238), which is what five of six chroma blocks of the bench content are: flat chroma under busy
luma, half-flat pictures (macroblock groups that mix DC-only and busy blocks), all-flat pictures, unchanged (0xFF)
blocks among them, and DC-only blocks spelled in ways the encoder never uses.  Every case runs through both launch
shapes of k_decode (a wave per part of a group / a wave that takes all three parts, MI_RTJ_ROTATE=1).  Bit-exact
against the oracle everywhere.  (Round 2 had a second kernel for the busy blocks of mostly-flat groups; it was slower
and is gone — the cases stay.)"""
import numpy as np
import pytest

import rtjlib as R
import test_gpu_parity as T
from pkg import P

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["part-per-wave", "three-parts-per-wave"])
def dev(request, monkeypatch):
    monkeypatch.setenv("MI_RTJ_ROTATE", "1" if request.param == "three-parts-per-wave" else "0")
    d = P.MiRtj()
    yield d
    d.close()


def decode_batch(dev, pkts, prefill=0):
    """batch decode through one plan, launched twice over the same buffers; returns the planes per packet."""
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    sizes = [T.frame_bytes(int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)) for p in pkts]
    oo = (np.cumsum([0] + [(s + 255) // 256 * 256 for s in sizes])).astype(np.uint64)
    d_out = dev.alloc(int(oo[-1]))
    dev.memset(d_out, prefill, int(oo[-1]))
    plan = dev.plan(hdrs, po, pl, oo[:-1].copy())
    plan.decode(d_stream, d_out)
    plan.decode(d_stream, d_out)
    dev.sync()
    outs = [dev.d2h(d_out, sizes[i], offset=int(oo[i])) for i in range(len(pkts))]
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
    return outs


def check(pkts, outs, prefill=0):
    for i, (p, got) in enumerate(zip(pkts, outs)):
        w, h = int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)
        want = np.full(T.frame_bytes(w, h), prefill, np.uint8)
        R.OracleDecoder().decode(p, want)
        assert T.first_diff(got, want) is None, (i, T.first_diff(got, want))


def test_bench_content_flat_chroma_under_busy_luma(dev):
    """Q=255, noise +-8 on luma and +-4 on chroma: five of six chroma blocks are DC only."""
    w, h = 1920, 1088
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, amp=8)) for i in range(2)]
    check(pkts, decode_batch(dev, pkts))


def half_flat(w, h, n, seed):
    """left half flat (every block DC only), right half noisy: luma groups mix both kinds."""
    f = R.synth_frame(w, h, n, seed=seed, amp=40)
    y, u, v = R.split_planes(f, w, h)
    y = y.reshape(h, w).copy()
    u = u.reshape(h // 2, w // 2).copy()
    v = v.reshape(h // 2, w // 2).copy()
    y[:, : w // 2] = 77
    u[:, : w // 4] = 90
    v[:, : w // 4] = 200
    return np.concatenate([y.ravel(), u.ravel(), v.ravel()])


@pytest.mark.parametrize("Q", [255, 200, 128, 30])
def test_half_flat_pictures(dev, Q):
    w, h = 640, 368
    pkts = [R.OracleEncoder(w, h, Q).encode(half_flat(w, h, i, 3)) for i in range(3)]
    check(pkts, decode_batch(dev, pkts, prefill=0x33), prefill=0x33)


def test_flat_pictures(dev):
    w, h = 320, 240
    flat = np.concatenate([np.full(w * h, 120, np.uint8), np.full(w * h // 4, 60, np.uint8),
                           np.full(w * h // 4, 190, np.uint8)])
    pkts = [R.OracleEncoder(w, h, Q).encode(flat) for Q in (255, 128, 5)]
    check(pkts, decode_batch(dev, pkts, prefill=1), prefill=1)


def test_unchanged_blocks_among_dc_only_ones(dev):
    """inter stream over a half-flat picture: 0xFF blocks keep what the buffer held (prefill)."""
    w, h, Q = 640, 368, 220
    enc = R.OracleEncoder(w, h, Q, key_rate=8, lmask=2, cmask=2)
    pkts = [enc.encode(half_flat(w, h, n // 3, 5)) for n in range(6)]
    assert any((p[12:] == 255).any() for p in pkts)
    check(pkts, decode_batch(dev, pkts, prefill=0x5A), prefill=0x5A)


def test_dc_only_blocks_spelled_unusually(dev):
    """DC followed by zero coefficients written out, by several short runs, by a run that overshoots, or with a
    non-zero raw byte: none is the encoder's "DC, zeros, one run" form; mixed with blocks that are."""
    rng = np.random.default_rng(12)
    w, h = 512, 64
    nblk = (w // 16) * (h // 16) * 6
    pkts = []
    for Q in (255, 192, 128, 60):
        _, _, lb8, cb8, _, _ = R.oracle_tables(Q)
        body = bytearray()
        for b in range(nblk):
            bt8 = lb8 if (b % 6) < 4 else cb8
            left = 63 - bt8
            kind = int(rng.integers(0, 7))
            dc = int(rng.integers(0, 255))
            if kind <= 2:    # the encoder's form
                blk = [dc] + [0] * bt8 + [63 + left]
            elif kind == 3:  # zeros spelled out, then a shorter run
                k = int(rng.integers(1, min(left, 6)))
                blk = [dc] + [0] * bt8 + [0] * k + ([63 + left - k] if left - k else [])
            elif kind == 4:  # two runs
                k = int(rng.integers(1, left)) if left > 1 else 1
                blk = [dc] + [0] * bt8 + [63 + k] + ([63 + left - k] if left - k else [])
            elif kind == 5:  # a run that overshoots the block
                blk = [dc] + [0] * bt8 + [127]
            else:            # one non-zero coefficient
                blk = [dc] + [0] * bt8 + [3, 63 + left - 1] if left > 1 else [dc] + [0] * bt8 + [3]
                if bt8:
                    blk[1 + int(rng.integers(0, bt8))] = int(rng.integers(1, 256)) if rng.random() < 0.5 else 0
            body += bytes(x & 0xFF for x in blk)
        total = 12 + len(body)
        hdr = bytes([total & 255, (total >> 8) & 255, (total >> 16) & 255, (total >> 24) & 255, 12, 0,
                     w & 255, w >> 8, h & 255, h >> 8, Q, 0])
        pkts.append(np.frombuffer(hdr + bytes(body), dtype=np.uint8).copy())
    check(pkts, decode_batch(dev, pkts, prefill=9), prefill=9)
