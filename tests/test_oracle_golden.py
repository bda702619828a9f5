"""The CPU oracle against the committed golden vectors (tests/golden/rtjpeg_golden.npz, produced
from the reference's own lib/RTjpeg.c by tests/golden/make_golden.py).  Runs anywhere."""
import numpy as np
import pytest

import rtjlib as R


@pytest.fixture(scope="module")
def G():
    return np.load(R.GOLDEN + "/rtjpeg_golden.npz")


def test_tables_match_reference_for_every_quality(G):
    for Q in range(1, 256):
        l, c, lb8, cb8, _, _ = R.oracle_tables(Q)
        assert np.array_equal(l, G["tab_liqt"][Q - 1]), Q
        assert np.array_equal(c, G["tab_ciqt"][Q - 1]), Q
        assert (lb8, cb8) == tuple(G["tab_b8"][Q - 1]), Q
    # anchors quoted in SURVEY.md Appendix A.2
    assert tuple(G["tab_b8"][254]) == (9, 0) and tuple(G["tab_b8"][191]) == (4, 0)
    assert tuple(G["tab_b8"][127]) == (0, 0)


def test_intra_streams(G):
    for (ci, n, w, h, Q, amp) in G["intra_meta"]:
        pkt, want = G[f"intra{ci}_{n}_pkt"], G[f"intra{ci}_{n}_planes"]
        # decode side
        got = np.zeros_like(want)
        assert R.OracleDecoder().decode(pkt, got) == pkt.size
        assert np.array_equal(got, want), (ci, n)
    # encode side (the encoder is stateless for intra streams)
    for (ci, n, w, h, Q, amp) in G["intra_meta"]:
        f = R.synth_frame(int(w), int(h), int(n), seed=2024, amp=int(amp))
        assert np.array_equal(R.OracleEncoder(int(w), int(h), int(Q)).encode(f), G[f"intra{ci}_{n}_pkt"])


def test_inter_sequence_with_skip_blocks(G):
    w, h, Q, key, lm, cm, nfr = [int(x) for x in G["inter_meta"]]
    enc, dec = R.OracleEncoder(w, h, Q, key, lm, cm), R.OracleDecoder()
    planes = np.zeros(w * h * 3 // 2, np.uint8)
    saw_skip = False
    for n in range(nfr):
        pkt = G[f"inter_{n}_pkt"]
        assert np.array_equal(enc.encode(R.synth_frame(w, h, n // 3, seed=5, amp=2)), pkt)
        saw_skip |= bool((pkt[12:] == 255).any())
        dec.decode(pkt, planes)
        assert np.array_equal(planes, G[f"inter_{n}_planes"]), n
    assert saw_skip


def test_adversarial_known_answers(G):
    for (ki, w, h, Q) in G["kat_meta"]:
        pkt, want = G[f"kat_{ki}_pkt"], G[f"kat_{ki}_planes"]
        got = np.full_like(want, 99)
        assert R.OracleDecoder().decode(pkt, got) == pkt.size
        assert np.array_equal(got, want), ki


@pytest.mark.parametrize("row", [0, 2])
def test_benchmark_size_digests(G, row):
    w, h, Q, amp, n, plen = [int(x) for x in G["big_meta"][row]]
    dfr, dpkt, dpl = [str(x) for x in G["big_digests"][row]]
    f = R.synth_frame(w, h, n, seed=12345, amp=amp)
    assert R.digest(f) == dfr
    pkt = R.OracleEncoder(w, h, Q).encode(f)
    assert pkt.size == plen and R.digest(pkt) == dpkt
    out = np.zeros(w * h * 3 // 2, np.uint8)
    R.OracleDecoder().decode(pkt, out)
    assert R.digest(out) == dpl


def test_block_offsets_consistent(G):
    pkt = G["intra2_0_pkt"]
    offs = R.OracleDecoder().block_offsets(pkt)
    assert offs[0] == 12 and offs[-1] == pkt.size
    assert (np.diff(offs.astype(np.int64)) >= 1).all() and (np.diff(offs.astype(np.int64)) <= 64).all()


def test_bad_geometry_is_rejected():
    pkt = np.zeros(64, np.uint8)
    pkt[4] = 12
    pkt[6], pkt[8] = 24, 16  # width 24 is not a multiple of 16: the reference would never terminate
    assert R.OracleDecoder().decode(pkt, np.zeros(24 * 16 * 3 // 2, np.uint8)) == -1


def test_truncated_packet_reads_zeros():
    w, h = 32, 32
    pkt = R.OracleEncoder(w, h, 255).encode(R.synth_frame(w, h, 0))
    cut = pkt[: pkt.size // 2]
    a = np.zeros(w * h * 3 // 2, np.uint8)
    b = a.copy()
    R.OracleDecoder().decode(cut, a)
    padded = np.concatenate([cut, np.zeros(8192, np.uint8)])
    R.OracleDecoder().decode(padded, b)
    assert np.array_equal(a, b)
