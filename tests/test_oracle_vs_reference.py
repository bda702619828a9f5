"""Pins the CPU oracle to the reference's own lib/RTjpeg.c (compiled into oracle/_ref/
by oracle/Makefile).  Runs only where that build exists (this container; the .so also
travels to the GPU box).  SURVEY.md §8c: the reference's tests pin nothing for this path,
so the compiled reference file is the pin."""
import numpy as np
import pytest

import rtjlib as R

pytestmark = pytest.mark.skipif(not R.have_reference(), reason="oracle/_ref not built")


def test_tables_all_q():
    ref = R.RefCodec()
    for Q in range(1, 256):
        rl, rc = ref.tables(Q)
        ol, oc, lb8, cb8, _, _ = R.oracle_tables(Q)
        assert np.array_equal(rl, ol), Q
        assert np.array_equal(rc, oc), Q


@pytest.mark.parametrize("w,h,Q,amp", [(64, 48, 255, 8), (320, 240, 255, 8), (320, 240, 128, 8),
                                        (320, 240, 64, 64), (320, 240, 1, 64), (16, 16, 200, 30),
                                        (640, 368, 255, 64), (1920, 1088, 255, 8)])
def test_encode_decode_intra(w, h, Q, amp):
    ref_e, ref_d = R.RefCodec(), R.RefCodec()
    ref_e.setup_encoder(w, h, Q)
    ora_e, ora_d = R.OracleEncoder(w, h, Q), R.OracleDecoder()
    nfr = 2 if w >= 1920 else 4
    for n in range(nfr):
        f = R.synth_frame(w, h, n, seed=7, amp=amp)
        pr, po = ref_e.encode(f), ora_e.encode(f)
        assert np.array_equal(pr, po), (n, pr.size, po.size)
        outr = np.full(w * h * 3 // 2, 77, np.uint8)
        outo = outr.copy()
        ref_d.decode(pr, outr)
        used = ora_d.decode(po, outo)
        assert used == po.size
        assert np.array_equal(outr, outo)


@pytest.mark.parametrize("key_rate,lm,cm", [(5, 2, 2), (3, 16, 16), (255, 1, 0)])
def test_encode_decode_inter_skip_blocks(key_rate, lm, cm):
    w, h, Q = 320, 240, 200
    ref_e, ref_d = R.RefCodec(), R.RefCodec()
    ref_e.setup_encoder(w, h, Q, key_rate, lm, cm)
    ora_e, ora_d = R.OracleEncoder(w, h, Q, key_rate, lm, cm), R.OracleDecoder()
    outr = np.zeros(w * h * 3 // 2, np.uint8)
    outo = outr.copy()
    nskip = 0
    for n in range(9):
        # slowly changing content so that some blocks are "unchanged"
        f = R.synth_frame(w, h, n // 3, seed=3, amp=2)
        pr, po = ref_e.encode(f), ora_e.encode(f)
        assert np.array_equal(pr, po), n
        nskip += int((pr[12:] == 255).sum())
        ref_d.decode(pr, outr)
        ora_d.decode(po, outo)
        assert np.array_equal(outr, outo), n
    assert nskip > 0


def test_random_coefficient_blocks_s2b_idct():
    """Adversarial blocks: random bytes through the oracle's decoder vs the reference's, at
    qualities where the int16 narrowing of s2b and DESCALE is observable (Q=1..8)."""
    rng = np.random.default_rng(1)
    w, h = 64, 32
    nblk = (w // 16) * (h // 16) * 6
    for Q in (1, 2, 8, 255):
        _, _, lb8, cb8, _, _ = R.oracle_tables(Q)
        for trial in range(20):
            body = bytearray()
            for b in range(nblk):
                bt8 = lb8 if (b % 6) < 4 else cb8
                dc = int(rng.integers(0, 255))
                blk = [dc] + [int(x) for x in rng.integers(0, 256, bt8)]
                co = bt8 + 1
                while co < 64:
                    if rng.random() < 0.3:
                        run = int(rng.integers(1, 64 - co + 1))
                        blk.append(63 + run)
                        co += run
                    else:
                        v = int(rng.integers(-64, 64))
                        blk.append(v & 0xFF)
                        co += 1
                body += bytes(blk)
            total = 12 + len(body)
            hdr = bytes([total & 255, (total >> 8) & 255, (total >> 16) & 255, 0, 12, 0,
                         w & 255, w >> 8, h & 255, h >> 8, Q, 0])
            pkt = np.frombuffer(hdr + bytes(body), dtype=np.uint8)
            outr = np.zeros(w * h * 3 // 2, np.uint8)
            outo = outr.copy()
            R.RefCodec().decode(pkt, outr)
            used = R.OracleDecoder().decode(pkt, outo)
            assert used == pkt.size
            assert np.array_equal(outr, outo), (Q, trial)


def test_quality_zero_on_fresh_decoder_keeps_zero_tables():
    """RTjpeg_init bzero's the state, so a header with quality 0 never triggers
    RTjpeg_set_quality on a fresh decoder (RTjpeg.c:3576): all coefficients dequantise to 0."""
    w, h = 32, 16
    enc = R.OracleEncoder(w, h, 100)
    pkt = enc.encode(R.synth_frame(w, h, 0))
    pkt[10] = 0
    outr = np.zeros(w * h * 3 // 2, np.uint8)
    outo = outr.copy()
    ref, ora = R.RefCodec(), R.OracleDecoder()
    ref.decode(pkt, outr)
    ora.decode(pkt, outo)
    assert np.array_equal(outr, outo)
    # ...but after a real quality was seen, 0 is clamped to 1 (RTjpeg.c:2410)
    pkt2 = enc.encode(R.synth_frame(w, h, 1))
    ref.decode(pkt2, outr)
    ora.decode(pkt2, outo)
    ref.decode(pkt, outr)
    ora.decode(pkt, outo)
    assert np.array_equal(outr, outo)
    assert ora.quality() == 1
