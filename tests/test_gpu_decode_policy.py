"""The batch launches' decode policy (csrc/rtj_decode_kernels.h, kDecMode*): k_decode_split — luma waves and chroma waves
that pool three groups' busy blocks (csrc/rtj_decode_chroma.h) — while next to nothing is left to k_decode_list,
k_decode<true, false> for 64 launches after a launch whose chroma the pooling waves did not cover.  Pictures are the
oracle's whatever form runs, through the change of form and back."""
import numpy as np
import pytest

import rtjlib as R
import test_gpu_parity as T
from pkg import P

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev(monkeypatch):
    monkeypatch.setenv("MI_RTJ_ROTATE", "1")  # the batch launch shape for small batches too
    d = P.MiRtj()
    yield d
    d.close()


def run(dev, pkts, w, h, launches):
    fsz = T.frame_bytes(w, h)
    want = []
    for p in pkts:
        o = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, o)
        want.append(o)
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    d_out = dev.alloc(fsz * len(pkts))
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    forms = []
    for k in range(launches):
        dev.memset(d_out, 0, fsz * len(pkts))
        plan.decode(d_stream, d_out)
        dev.sync()
        forms.append(plan.decode_form())
        if k < 3 or k >= launches - 3 or k % 16 == 0:
            for i in range(len(pkts)):
                assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), want[i]), (k, i)
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
    return forms


def test_clean_content_stays_with_the_split_form(dev):
    w, h = 640, 368
    pkts = [R.OracleEncoder(w, h, Q).encode(R.synth_frame(w, h, i, seed=5, amp=a)) for i, (Q, a) in enumerate([(255, 8), (255, 0), (128, 8), (64, 6)] * 2)]
    forms = run(dev, pkts, w, h, 4)
    assert all(f[0] == 0 for f in forms), forms
    assert forms[-1][2] <= 2  # next to nothing for k_decode_list


def test_noisy_content_moves_to_the_classic_form_and_comes_back(dev):
    w, h = 640, 368
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=6, amp=48)) for i in range(6)]
    forms = run(dev, pkts, w, h, 68)
    assert forms[0][0] == 1 and forms[0][2] > 0, forms[0]     # the first launch listed most chroma parts: classic next
    assert all(f[0] == 1 and f[2] == 0 for f in forms[1:64]), forms[1:5]  # 64 classic launches, nothing listed
    assert forms[64][0] == 0, forms[62:66]                     # then the split form is tried again ...
    assert forms[65][0] == 1, forms[62:67]                     # ... and given up again on this content


def test_forced_forms(dev, monkeypatch):
    w, h = 320, 240
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=7, amp=a)) for i, a in enumerate([4, 60, 8, 30])]
    for v in ("0", "1"):
        monkeypatch.setenv("MI_RTJ_SPLIT", v)
        forms = run(dev, pkts, w, h, 3)
        assert all(f[0] == 1 - int(v) for f in forms), (v, forms)


def test_long_to_do_lists_go_to_the_serial_walker(monkeypatch):
    """packets the speculative index refuses (noisy content) are indexed by the serial walker, one wave
    per packet, when there are many of them (k_spec_policy; MI_RTJ_SERIAL_MIN lowers "many" for this small batch), by the
    exact kernels otherwise — the same index and the same pictures either way, launch after launch (the policy moves
    through its leads and pauses the speculation on this content)"""
    monkeypatch.setenv("MI_RTJ_SPEC", "2")  # speculate whatever the batch size, with the policy
    w, h = 320, 240
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=8, amp=64 if i % 3 else 6)) for i in range(24)]
    fsz = T.frame_bytes(w, h)
    want = []
    for p in pkts:
        o = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, o)
        want.append(o)
    nblk = (w // 16) * (h // 16) * 6
    for serial_min in ("4", "0"):
        monkeypatch.setenv("MI_RTJ_SERIAL_MIN", serial_min)
        dev = P.MiRtj()
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        d_out = dev.alloc(fsz * len(pkts))
        plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
        for k in range(8):
            dev.memset(d_out, 0, fsz * len(pkts))
            plan.decode(d_stream, d_out)
            dev.sync()
            idx = plan.read_index()
            at = 0
            for i, p in enumerate(pkts):
                assert np.array_equal(idx[at:at + nblk + 1], R.OracleDecoder().block_offsets(p) - 12), (serial_min, k, i)
                at += nblk + 1
                assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), want[i]), (serial_min, k, i)
        plan.close()
        dev.free(d_stream)
        dev.free(d_out)
        dev.close()


def test_the_serial_walker_on_built_and_arbitrary_streams(monkeypatch):
    """walk_packet (the serial walker's loop: k_index_walk_todo, and k_index_walk under MI_RTJ_INDEX=serial) on streams that
    are not an encoder's: maximal blocks, 1-byte (0xFF) blocks, unit runs and mixtures at two qualities (different raw-byte
    counts for luma and chroma), macroblock counts around the walker's store rounds of 58-64 offsets, arbitrary bytes,
    payloads shorter and longer than the picture needs, an empty payload — index and pictures as the oracle's."""
    monkeypatch.setenv("MI_RTJ_INDEX", "serial")
    rng = np.random.default_rng(99)
    pkts, sizes = [], []
    for (w, h) in [(16, 16), (160, 16), (176, 16), (336, 16), (352, 32), (1024, 64)]:
        nblk = (w // 16) * (h // 16) * 6
        for Q, mode in [(255, "long"), (255, "skip"), (255, "mix"), (100, "mix"), (20, "mix"), (200, "bytes"), (255, "short"), (90, "empty")]:
            _, _, lb8, cb8, _, _ = R.oracle_tables(Q)
            body = bytearray()
            if mode == "bytes":
                body = bytearray(rng.integers(0, 256, int(rng.integers(0, nblk * 70)), dtype=np.uint8).tobytes())
            elif mode != "empty":
                for b in range(nblk if mode != "short" else nblk // 2):
                    bt8 = lb8 if (b % 6) < 4 else cb8
                    kind = mode if mode not in ("mix", "short") else ["long", "skip", "one", "runs"][int(rng.integers(0, 4))]
                    if kind == "skip":
                        body.append(255)
                        continue
                    blk = [int(rng.integers(0, 255))] + [int(x) for x in rng.integers(0, 256, bt8)]
                    left = 63 - bt8
                    if kind == "long":
                        blk += [int(x) & 0xFF for x in rng.integers(-64, 64, left)]
                    elif kind == "one":
                        blk.append(63 + left)
                    else:
                        while left > 0:
                            r = int(rng.integers(1, min(left, 3) + 1))
                            blk.append(63 + r)
                            left -= r
                    body += bytes(blk)
            total = 12 + len(body)
            hdr = bytes([total & 255, (total >> 8) & 255, (total >> 16) & 255, (total >> 24) & 255, 12, 0,
                         w & 255, w >> 8, h & 255, h >> 8, Q, 0])
            pkts.append(np.frombuffer(hdr + bytes(body), dtype=np.uint8).copy())
            sizes.append((w, h))
    dev = P.MiRtj()
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    fs = [T.frame_bytes(w, h) for (w, h) in sizes]
    offs = np.concatenate([[0], np.cumsum(fs)]).astype(np.uint64)
    d_out = dev.alloc(int(offs[-1]))
    dev.memset(d_out, 0x33, int(offs[-1]))
    plan = dev.plan(hdrs, po, pl, offs[:-1])
    plan.decode(d_stream, d_out)
    dev.sync()
    idx = plan.read_index()
    dec = R.OracleDecoder()
    at = 0
    for i, p in enumerate(pkts):
        w, h = sizes[i]
        nblk = (w // 16) * (h // 16) * 6
        want = np.full(fs[i], 0x33, np.uint8)
        dec.decode(p, want)
        assert np.array_equal(idx[at:at + nblk + 1], dec.block_offsets(p) - 12), i
        at += nblk + 1
        assert np.array_equal(dev.d2h(d_out, fs[i], offset=int(offs[i])), want), i
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
    dev.close()


def test_launches_the_serial_walker_takes_never_try_the_longest_lead(monkeypatch):
    """k_spec_policy: walkers with the 6,144-byte lead parse four times their chunk; where the to-do list of a launch goes
    to the serial walker anyway (MI_RTJ_SERIAL_MIN <= packets of the launch) a plan whose 1,536-byte lead fails pauses the
    speculation instead of trying it; launches the walker does not take (MI_RTJ_SERIAL_MIN = 0) still do.  Pictures as the
    oracle's all along."""
    monkeypatch.setenv("MI_RTJ_SPEC", "2")  # speculate whatever the batch size, with the policy
    w, h = 320, 240
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=5, amp=64)) for i in range(16)]
    fsz = T.frame_bytes(w, h)
    want = []
    for p in pkts:
        o = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, o)
        want.append(o)
    seen = {}
    for serial_min in ("4", "0"):
        monkeypatch.setenv("MI_RTJ_SERIAL_MIN", serial_min)
        dev = P.MiRtj()
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        d_out = dev.alloc(fsz * len(pkts))
        plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
        leads, paused = [], 0
        for k in range(10):
            leads.append(plan.spec_lead()[0])
            paused += plan.spec_lead()[1] > 0
            dev.memset(d_out, 0, fsz * len(pkts))
            plan.decode(d_stream, d_out)
            dev.sync()
            for i in range(len(pkts)):
                assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), want[i]), (serial_min, k, i)
        seen[serial_min] = (leads, paused)
        plan.close()
        dev.free(d_stream)
        dev.free(d_out)
        dev.close()
    assert 6144 not in seen["4"][0] and seen["4"][1] > 0, seen["4"]   # paused straight from the 1,536-byte lead
    assert 1536 in seen["4"][0]
    assert 6144 in seen["0"][0], seen["0"]                              # the longest lead was tried first
