"""Batch plans build the block-offset index of launch k + 1 while launch k is being transformed: two indices per plan,
the index kernels on a stream of their own (csrc/mi_rtjpeg.hip, plan_launch).  MI_RTJ_OVERLAP=1 switches that on for
plans of any size: launches queued back to back without a sync in between must each leave the oracle's pictures and
the oracle's index, whichever of the two indices they used, with the exact and with the speculative index."""
import numpy as np
import pytest

import rtjlib as R
import test_gpu_parity as T
from pkg import P

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["0", "1"], ids=["exact-index", "speculative-index"])
def dev(monkeypatch, request):
    monkeypatch.setenv("MI_RTJ_OVERLAP", "1")
    monkeypatch.setenv("MI_RTJ_SPEC", request.param)
    d = P.MiRtj()
    yield d
    d.close()


def test_back_to_back_launches(dev):
    w, h = 640, 368
    pkts = [R.OracleEncoder(w, h, Q).encode(R.synth_frame(w, h, i, seed=31, amp=a))
            for i, (Q, a) in enumerate([(255, 8), (128, 30), (200, 0), (255, 64), (60, 12)])]
    fsz = T.frame_bytes(w, h)
    want = []
    for p in pkts:
        o = np.zeros(fsz, np.uint8)
        R.OracleDecoder().decode(p, o)
        want.append(o)
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    outs = [dev.alloc(fsz * len(pkts)) for _ in range(5)]
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    for d_out in outs:  # five launches in a row, no sync: indices 0 1 0 1 0
        plan.decode(d_stream, d_out)
    dev.sync()
    for k, d_out in enumerate(outs):
        for i in range(len(pkts)):
            assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), want[i]), (k, i)
    nblk = (w // 16) * (h // 16) * 6
    for extra in range(2):  # the index the LAST launch wrote is the one read back: after 5 and after 6 launches
        idx = plan.read_index()
        at = 0
        for i, p in enumerate(pkts):
            assert np.array_equal(idx[at:at + nblk + 1], R.OracleDecoder().block_offsets(p) - 12), (extra, i)
            at += nblk + 1
        plan.decode(d_stream, outs[0])
    dev.sync()
    plan.close()
    dev.free(d_stream)
    for d_out in outs:
        dev.free(d_out)


def test_parity_suite_with_overlapped_plans(dev):
    T.test_mixed_batch_sizes_and_qualities(dev)
    T.test_skip_blocks_leave_destination_untouched_in_batches(dev)
    T.test_truncated_and_empty_packets(dev)
    T.test_chunk_boundaries_and_long_blocks(dev)


def test_a_memset_queued_before_the_launch_is_seen_by_the_index_stream(dev):
    """mi_rtj_dev_memset is asynchronous on the instance's stream, the index kernels of an overlapped plan
    run on another one.  The packets' data bytes are overwritten with 0xFF (every block "unchanged") right before the
    launch, no sync in between: an index that ran ahead of the memset would parse the old packets and pictures
    would appear; ordered behind it, nothing is decoded and the output keeps its filling."""
    w, h = 1920, 1088
    pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=77, amp=8)) for i in range(3)] * 8
    fsz = T.frame_bytes(w, h)
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
    d_out = dev.alloc(fsz * len(pkts))
    plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
    dev.memset(d_out, 0x5A, fsz * len(pkts))
    dev.sync()
    for i in range(len(pkts)):  # (whole packets but their 12 header bytes, which live on the host side of the plan)
        dev.memset(d_stream, 0xFF, int(pl[i]) - 12, offset=int(po[i]) + 12)
    plan.decode(d_stream, d_out)
    dev.sync()
    for i in range(len(pkts)):
        assert (dev.d2h(d_out, fsz, offset=i * fsz) == 0x5A).all(), i
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)


def test_a_long_plan_of_noisy_content_overlaps_once_the_host_has_seen_the_classic_mode(monkeypatch):
    """Plans too long to overlap by default build the index of launch k + 1 next to the transform of launch k while the
    host sees the decode policy in its classic mode (noisy content): the switch happens between launches of one plan,
    both ways of running share the plan's first index, and pictures and index are the oracle's on every launch."""
    import rtjlib as R
    monkeypatch.setenv("MI_RTJ_ROTATE", "1")        # the batch launch shape (and with it the decode policy) for a small batch
    monkeypatch.setenv("MI_RTJ_OVERLAP", "2")       # ... and the long plans' rule for this small one
    monkeypatch.setenv("MI_RTJ_SPEC", "2")
    w, h = 320, 240
    for amp in (64, 2):  # noisy: the policy goes to the classic form and the plan starts to overlap; quiet: it never does
        pkts = [R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, i, seed=3, amp=amp)) for i in range(12)]
        fsz = w * h * 3 // 2
        want = []
        for p in pkts:
            o = np.zeros(fsz, np.uint8)
            R.OracleDecoder().decode(p, o)
            want.append(o)
        nblk = (w // 16) * (h // 16) * 6
        dev = P.MiRtj()
        d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
        d_out = dev.alloc(fsz * len(pkts))
        plan = dev.plan(hdrs, po, pl, np.arange(len(pkts), dtype=np.uint64) * np.uint64(fsz))
        for k in range(12):
            dev.memset(d_out, 0, fsz * len(pkts))
            plan.decode(d_stream, d_out)
            if k % 3 == 2:  # (launches in a row without the host waiting in between, too)
                dev.sync()
                idx = plan.read_index()
                for i, p in enumerate(pkts):
                    assert np.array_equal(idx[i * (nblk + 1):(i + 1) * (nblk + 1)], R.OracleDecoder().block_offsets(p) - 12), (amp, k, i)
                    assert np.array_equal(dev.d2h(d_out, fsz, offset=i * fsz), want[i]), (amp, k, i)
        dev.sync()
        assert plan.overlapped() == (amp == 64), (amp, plan.overlapped())
        plan.close()
        dev.free(d_stream)
        dev.free(d_out)
        dev.close()
