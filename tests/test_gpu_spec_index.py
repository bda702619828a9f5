"""The speculative block index (csrc/rtj_spec_kernels.h) is switched on by batch size; MI_RTJ_SPEC=1 / 3
forces it for every plan, also the one-packet path.  The parity tests are run again that way: streams
an encoder made must come out of the speculative path (the exact kernels then skip the packet),
adversarial ones must be rejected by its proof step and indexed by the exact kernels — either way the
block index and the planes equal the oracle's."""
import numpy as np
import pytest

import rtjlib as R
import test_gpu_parity as T
from pkg import P

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["1", "3", "4"], ids=["short-lead", "long-lead", "three-chunk-lead"])
def dev(monkeypatch, request):
    # always speculate, no policy: 1 the short lead, 3 the long one, 4 the lead of three chunks (walkers of a packet's
    # first chunks then start at byte 0 with what lead there is)
    monkeypatch.setenv("MI_RTJ_SPEC", request.param)
    d = P.MiRtj()
    yield d
    d.close()


@pytest.fixture(scope="module")
def G():
    return np.load(R.GOLDEN + "/rtjpeg_golden.npz")


def test_golden_vectors(dev, G):
    T.test_golden_intra_streams_batch(dev, G)
    T.test_golden_adversarial_known_answers(dev, G)


@pytest.mark.parametrize("w,h,Q,amp,n", [(320, 240, 255, 8, 6), (320, 240, 128, 64, 3), (336, 256, 1, 64, 2),
                                          (1920, 1088, 255, 8, 2), (1920, 1088, 64, 64, 1), (4096, 16, 90, 20, 2)])
def test_decode_matches_oracle(dev, w, h, Q, amp, n):
    T.test_decode_matches_oracle(dev, w, h, Q, amp, n)


def test_mixed_batches_skip_blocks_truncation_and_fuzz(dev):
    T.test_mixed_batch_sizes_and_qualities(dev)
    T.test_skip_blocks_leave_destination_untouched_in_batches(dev)
    T.test_truncated_and_empty_packets(dev)
    T.test_chunk_boundaries_and_long_blocks(dev)
    T.test_fuzz_arbitrary_bytes(dev)
    T.test_low_4x4_transform_path_and_its_boundary(dev)


def test_one_packet_path_and_inter_stream(dev, G):
    T.test_golden_inter_sequence_single_stream(dev, G)
    T.test_random_token_streams(dev)
    T.test_quality_zero_state_machine(dev)


def test_speculation_is_taken_or_refused_as_expected(dev):
    """Encoder-made 1080p packets are proven; a packet of random bytes is refused."""
    w, h = 1920, 1088
    good = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, amp=8)) for i in range(2)]
    rng = np.random.default_rng(1)
    n = 600000
    total = 12 + n
    hdr = np.array([total & 255, (total >> 8) & 255, (total >> 16) & 255, 0, 12, 0, w & 255, w >> 8, h & 255, h >> 8, 200, 0], np.uint8)
    bad = [np.concatenate([hdr, rng.integers(0, 256, n, dtype=np.uint8)])]
    for pkts, want_fallback in ((good, False), (bad, True)):
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        fsz = T.frame_bytes(w, h)
        d_out = dev.alloc(fsz * len(pkts))
        plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
        plan.decode(d_stream, d_out)
        dev.sync()
        plan.profile(True)
        plan.decode(d_stream, d_out)
        ms, _ = plan.times()
        dec = R.OracleDecoder()
        for i, p in enumerate(pkts):
            wantp = np.zeros(fsz, np.uint8)
            dec.decode(p, wantp)
            assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), wantp)
        assert ms["k_spec_walk"] > 0
        proven, walkers = plan.spec_stats()
        assert walkers > 0 and proven == (0 if want_fallback else len(pkts)), (proven, walkers)
        plan.close()
        dev.free(d_stream)
        dev.free(d_out)


def test_packets_whose_length_is_a_whole_number_of_walker_chunks_are_proven(dev):
    """the index's last entry is the END position, the start of a block that is not there: when the last block ends
    exactly where the packet does and that is a chunk boundary, the end position is the first byte of a chunk past the
    packet's last one, which needs a walker of its own (without it such packets — 1 in 2048 — were refused and went to
    the exact kernels).  512x512 has 6144 blocks: a picture of nothing but unchanged (one-byte) blocks is three chunks."""
    w, h, chunk = 512, 512, 2048
    enc = R.OracleEncoder(w, h, 255, 100, 2, 2)
    first = enc.encode(R.synth_frame(w, h, 0, amp=8))
    same = enc.encode(R.synth_frame(w, h, 0, amp=8))  # the same picture again: every block unchanged
    assert same.size - 12 == 6 * (w // 16) * (h // 16) == 3 * chunk and np.all(same[12:] == 0xFF)
    pkts = [first, same]
    up = (first.size - 12 + chunk - 1) // chunk * chunk  # zero bytes behind the last block are never parsed: the extra
    for n in (up, up + 1, up + chunk - 1, up + chunk):   # walker then meets nothing but them
        p = np.concatenate([first, np.zeros(12 + n - first.size, np.uint8)])
        t = p.size
        p[0:4] = [t & 255, (t >> 8) & 255, (t >> 16) & 255, (t >> 24) & 255]
        pkts.append(p)
    outs = T.batch_decode(dev, pkts, prefill=0x55)  # (checks the block index against the oracle's as well)
    fsz = T.frame_bytes(w, h)
    want = np.full(fsz, 0x55, np.uint8)
    R.OracleDecoder().decode(first, want)
    for i, got in enumerate(outs):
        assert T.first_diff(got, np.full(fsz, 0x55, np.uint8) if i == 1 else want) is None, i
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    d_out = dev.alloc(fsz * len(pkts))
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    plan.decode(d_stream, d_out)
    dev.sync()
    proven, walkers = plan.spec_stats()
    assert proven == len(pkts), (proven, len(pkts))
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)


def test_plans_that_keep_being_refused_pause_the_speculation(monkeypatch):
    """Device-side policy (k_spec_policy): a launch in which every packet was refused moves the plan to the
    walkers with the next longer lead (768 -> 1536 -> 6144 bytes); after two such launches with the longest, all
    walkers return at once for a while; results do not change.  MI_RTJ_SPEC=2 switches the speculation on regardless of the batch size but,
    unlike =1, leaves the policy active."""
    monkeypatch.setenv("MI_RTJ_SPEC", "2")
    d = P.MiRtj()
    w, h = 640, 368
    n = (w // 16) * (h // 16) * 6 * 64
    rng = np.random.default_rng(3)
    total = 12 + n
    hdr = np.array([total & 255, (total >> 8) & 255, (total >> 16) & 255, 0, 12, 0, w & 255, w >> 8, h & 255, h >> 8, 255, 0], np.uint8)
    # every byte a coefficient (1..63), no zero run, no 0xFF: every block is 64 bytes long whatever its type, and a walk
    # that starts in the wrong macroblock phase never finds the right one, however long its lead (at this quality luma
    # and chroma blocks parse differently, so the phase is part of the proof)
    pkts = [np.concatenate([hdr, rng.integers(1, 64, n, dtype=np.uint8)]) for _ in range(3)]
    d_stream, po, pl, hdrs = d.upload_packets(pkts, align=1)
    fsz = T.frame_bytes(w, h)
    d_out = d.alloc(fsz * len(pkts))
    plan = d.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    dec = R.OracleDecoder()
    want = []
    for p in pkts:
        x = np.zeros(fsz, np.uint8)
        dec.decode(p, x)
        want.append(x)
    walk, lead = [], [plan.spec_lead()]
    for it in range(9):
        d.memset(d_out, 0, fsz * len(pkts))
        plan.profile(True)
        plan.decode(d_stream, d_out)
        ms, _ = plan.times()
        walk.append(ms["k_spec_walk"])
        # (while the walkers run, k_spec_repair may well carry such a packet — one wave re-walks a whole run of walkers
        # that are out of step — but a launch that needs a quarter of its walkers repaired counts as lost all the same)
        if it >= 4:
            assert plan.spec_stats()[0] == 0, it  # paused: every packet on the exact kernels' list
        lead.append(plan.spec_lead())
        for i in range(len(pkts)):
            assert np.array_equal(d.d2h(d_out, fsz, offset=i * fsz), want[i]), (it, i)
    # launches 1, 2: short and long lead, lost; 3, 4: the longest, lost twice; from 5 on the walkers return at once
    assert min(walk[:4]) > 5 * max(walk[4:]), walk
    short, long_, very = lead[0][0], lead[1][0], lead[2][0]
    assert 0 < short < long_ < very, lead
    assert [x[0] for x in lead[2:]] == [very] * 8, lead
    assert lead[1][1] == lead[2][1] == lead[3][1] == 0 and lead[4][1] > lead[5][1] > 0, lead
    plan.close()
    d.free(d_stream)
    d.free(d_out)
    d.close()


def test_many_refused_packets_loop_over_the_todo_rows(dev):
    """More refused packets than the exact kernels have grid rows (64), mixed with provable ones, in
    one plan: the rows loop over the to-do list; every index and picture equals the oracle's."""
    rng = np.random.default_rng(11)
    w, h = 160, 64
    good_enc = R.OracleEncoder(w, h, 255)
    pkts = []
    for i in range(200):
        if i % 4 == 3:
            pkts.append(good_enc.encode(R.synth_frame(w, h, i, amp=6)))
        else:
            n = int(rng.integers(200, 9000))
            total = 12 + n
            hdr = np.array([total & 255, (total >> 8) & 255, 0, 0, 12, 0, w & 255, w >> 8, h & 255, h >> 8, 255, 0], np.uint8)
            pkts.append(np.concatenate([hdr, rng.integers(0, 256, n, dtype=np.uint8)]))
    outs = T.batch_decode(dev, pkts, prefill=0x11)
    dec = R.OracleDecoder()
    for i, (p, got) in enumerate(zip(pkts, outs)):
        want = np.full(T.frame_bytes(w, h), 0x11, np.uint8)
        dec.decode(p, want)
        assert T.first_diff(got, want) is None, i


def test_walkers_that_lock_late_are_repaired(dev):
    """Moderately noisy content: a few walkers per packet have not fallen into step by the end of their
    lead; they are walked again from a block start of their predecessor and the packet is still proven."""
    w, h = 1920, 1088
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, amp=18)) for i in range(3)]
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    fsz = T.frame_bytes(w, h)
    d_out = dev.alloc(fsz * len(pkts))
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    plan.decode(d_stream, d_out)
    dev.sync()
    proven, walkers = plan.spec_stats()
    assert walkers > 0 and proven >= 2, (proven, walkers)  # without repairs: 0 (about 1 % of the walkers lock late)
    idx = plan.read_index()
    k = 0
    dec = R.OracleDecoder()
    for i, p in enumerate(pkts):
        want = dec.block_offsets(p).astype(np.int64) - 12
        assert np.array_equal(idx[k:k + want.size].astype(np.int64), want), i
        k += want.size
        wantp = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, wantp)
        assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), wantp), i
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)


def test_content_that_locks_late_moves_the_plan_to_the_long_lead(monkeypatch):
    """Noise of +-22 at Q=255: with the short lead about one walker in six has to be walked again, so the
    policy moves the plan to the long lead for the next decode; pictures equal the oracle's either way, and
    after enough decodes in a row without any repair the plan tries the short lead again."""
    monkeypatch.setenv("MI_RTJ_SPEC", "2")
    d = P.MiRtj()
    w, h = 1920, 1088
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, amp=22)) for i in range(3)]
    d_stream, po, pl, hdrs = d.upload_packets(pkts, align=1)
    fsz = T.frame_bytes(w, h)
    d_out = d.alloc(fsz * len(pkts))
    plan = d.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    want = []
    for p in pkts:
        x = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, x)
        want.append(x)
    short = plan.spec_lead()[0]
    leads, repaired, proven = [], [], []
    for it in range(20):
        d.memset(d_out, 0, fsz * len(pkts))
        plan.decode(d_stream, d_out)
        proven.append(plan.spec_stats()[0])
        repaired.append(plan.repaired)
        leads.append(plan.spec_lead()[0])
        if it in (0, 1, 17, 18, 19):
            for i in range(len(pkts)):
                assert np.array_equal(d.d2h(d_out, fsz, offset=i * fsz), want[i]), (it, i)
    assert leads[0] > short, (short, leads)           # the first decode saw many repairs
    assert repaired[0] > 8 * max(1, repaired[1]), repaired  # the long lead needs few
    assert proven[1] == len(pkts), proven
    if repaired[1] == 0:  # the same packets every time: all decodes with the long lead are alike
        assert short in leads[1:], leads              # quiet long enough: back to the short lead ...
        k = leads.index(short, 1)
        assert leads[k + 1] > short, leads            # ... which at once shows why it was left
    else:
        assert short not in leads, leads              # a repair in 1760 walkers is not quiet: the long lead stays
    plan.close()
    d.free(d_stream)
    d.free(d_out)
    d.close()


def test_spans_of_one_byte_blocks_need_no_cap(dev):
    """A still picture coded with key frames far apart is mostly 0xFF bytes: one block per byte.  Round 2's walkers
    recorded 16-bit positions and refused a span with more than 2048 blocks (the packet then went to the exact
    kernels); a start BIT per byte has no such limit: the packets are proven by the speculative index, and a chunk
    of them runs through several windows of k_spec_verify's rank ordering (512 ranks each).
    (Quality 128: luma and chroma blocks parse alike there, so a walker is in step as soon as it stands on a block
    start.  At a quality where they differ, walkers in a run of one-byte blocks keep whatever macroblock phase they
    assumed — every block is one byte whatever its type — and such packets go to the exact kernels, by design.)"""
    w, h, Q = 1920, 1088, 128
    assert R.oracle_tables(Q)[2] == R.oracle_tables(Q)[3]
    nblk = (w // 16) * (h // 16) * 6
    enc = R.OracleEncoder(w, h, Q, 100, 16, 16)
    still = R.synth_frame(w, h, 3, amp=8)
    pkts = [enc.encode(still) for _ in range(3)]
    assert (pkts[1][12:] == 255).mean() > 0.9 and pkts[1].size - 12 < nblk * 2  # nearly every block unchanged
    # and an all-unchanged packet by hand, longer than a walker's chunk by far
    total = 12 + nblk
    hdr = np.array([total & 255, (total >> 8) & 255, (total >> 16) & 255, 0, 12, 0, w & 255, w >> 8, h & 255, h >> 8, Q, 0], np.uint8)
    pkts.append(np.concatenate([hdr, np.full(nblk, 255, np.uint8)]))
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    fsz = T.frame_bytes(w, h)
    d_out = dev.alloc(fsz * len(pkts))
    dev.memset(d_out, 0x41, fsz * len(pkts))
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    plan.decode(d_stream, d_out)
    dev.sync()
    want_idx = [R.OracleDecoder().block_offsets(p) - 12 for p in pkts]
    got_idx = plan.read_index()
    at = 0
    for i, p in enumerate(pkts):
        assert np.array_equal(got_idx[at:at + nblk + 1], want_idx[i][:nblk + 1]), i
        at += nblk + 1
        wantp = np.full(fsz, 0x41, np.uint8)  # every plan frame has its own picture: unchanged blocks keep the prefill
        R.OracleDecoder().decode(p, wantp)
        assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), wantp), i
    proven, walkers = plan.spec_stats()
    assert walkers > 0 and proven == len(pkts), (proven, walkers)
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
