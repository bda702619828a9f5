"""Regenerates tests/golden/rtjpeg_golden.npz from the REFERENCE's own lib/RTjpeg.c
(oracle/_ref/librtjpeg_ref.so, built by oracle/Makefile from /root/reference — this container
only).  The fixture is data: input packets / frame parameters and the reference's outputs.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import rtjlib as R  # noqa: E402


def ref_lb8_cb8(ref):
    # RTjpeg_t layout (include/RTjpeg.h:40-58): block[64] i16, ws[256] i32, 4 x [64] i32, then lb8, cb8
    off = 128 + 1024 + 4 * 256
    ints = (C.c_int * 2).from_address(ref.h + off)
    return int(ints[0]), int(ints[1])


def adversarial_packet(rng, w, h, Q, lb8, cb8, skip_prob=0.0):
    nblk = (w // 16) * (h // 16) * 6
    body = bytearray()
    for b in range(nblk):
        if rng.random() < skip_prob:
            body.append(255)
            continue
        bt8 = lb8 if (b % 6) < 4 else cb8
        blk = [int(rng.integers(0, 255))] + [int(x) for x in rng.integers(0, 256, bt8)]
        co = bt8 + 1
        while co < 64:
            if rng.random() < 0.3:
                run = int(rng.integers(1, 64 - co + 1))
                blk.append(63 + run)
                co += run
            else:
                blk.append(int(rng.integers(-64, 64)) & 0xFF)
                co += 1
        body += bytes(blk)
    total = 12 + len(body)
    hdr = bytes([total & 255, (total >> 8) & 255, (total >> 16) & 255, (total >> 24) & 255, 12, 0,
                 w & 255, w >> 8, h & 255, h >> 8, Q, 0])
    return np.frombuffer(hdr + bytes(body), dtype=np.uint8).copy()


def main():
    assert R.have_reference(), "build oracle/_ref first (make -C oracle)"
    out = {}
    # 1. tables for every quality, straight from RTjpeg_get_tables + the struct's lb8/cb8
    tl, tc, tb = [], [], []
    for Q in range(1, 256):
        ref = R.RefCodec()
        l, c = ref.tables(Q)
        tl.append(l)
        tc.append(c)
        tb.append(ref_lb8_cb8(ref))
    out["tab_liqt"] = np.array(tl, np.int32)
    out["tab_ciqt"] = np.array(tc, np.int32)
    out["tab_b8"] = np.array(tb, np.int32)

    # 2. small intra streams: reference-encoded packets + reference-decoded planes (full data)
    cases = [(64, 48, 255, 8, 3), (64, 48, 96, 40, 2), (320, 240, 255, 8, 1), (16, 16, 30, 64, 2),
             (176, 144, 160, 16, 1)]
    meta = []
    for ci, (w, h, Q, amp, nfr) in enumerate(cases):
        enc, dec = R.RefCodec(), R.RefCodec()
        enc.setup_encoder(w, h, Q)
        for n in range(nfr):
            f = R.synth_frame(w, h, n, seed=2024, amp=amp)
            pkt = enc.encode(f)
            planes = np.zeros(w * h * 3 // 2, np.uint8)
            dec.decode(pkt, planes)
            out[f"intra{ci}_{n}_pkt"] = pkt
            out[f"intra{ci}_{n}_planes"] = planes
            meta.append((ci, n, w, h, Q, amp))
    out["intra_meta"] = np.array(meta, np.int32)

    # 3. an inter (skip-block) sequence, decoded in order into one persistent frame
    w, h, Q = 160, 128, 200
    enc, dec = R.RefCodec(), R.RefCodec()
    enc.setup_encoder(w, h, Q, key_rate=4, lmask=2, cmask=2)
    planes = np.zeros(w * h * 3 // 2, np.uint8)
    for n in range(7):
        f = R.synth_frame(w, h, n // 3, seed=5, amp=2)
        pkt = enc.encode(f)
        dec.decode(pkt, planes)
        out[f"inter_{n}_pkt"] = pkt
        out[f"inter_{n}_planes"] = planes.copy()
    out["inter_meta"] = np.array([w, h, Q, 4, 2, 2, 7], np.int32)

    # 4. adversarial known-answer packets (random coefficient bytes, random runs, some 0xFF),
    #    including the low-Q range where the int16 narrowing is visible
    rng = np.random.default_rng(99)
    kat = []
    for ki, (w, h, Q, sp) in enumerate([(16, 16, 1, 0.0), (16, 16, 2, 0.0), (32, 16, 8, 0.2),
                                        (32, 32, 255, 0.1), (48, 32, 129, 0.0), (16, 32, 192, 0.3)]):
        ref = R.RefCodec()
        ref.tables(Q)
        lb8, cb8 = ref_lb8_cb8(ref)
        pkt = adversarial_packet(rng, w, h, Q, lb8, cb8, sp)
        planes = np.full(w * h * 3 // 2, 99, np.uint8)
        R.RefCodec().decode(pkt, planes)
        out[f"kat_{ki}_pkt"] = pkt
        out[f"kat_{ki}_planes"] = planes
        kat.append((ki, w, h, Q))
    out["kat_meta"] = np.array(kat, np.int32)

    # 5. digests for the benchmark-size content (regenerated on the GPU box by the product's
    #    own generator + encoder): frame, packet and plane digests
    big = []
    for (w, h, Q, amp, n) in [(1920, 1088, 255, 8, 0), (1920, 1088, 255, 8, 1), (1920, 1088, 128, 64, 0),
                              (3840, 2160, 255, 8, 0)]:
        enc, dec = R.RefCodec(), R.RefCodec()
        enc.setup_encoder(w, h, Q)
        f = R.synth_frame(w, h, n, seed=12345, amp=amp)
        pkt = enc.encode(f)
        planes = np.zeros(w * h * 3 // 2, np.uint8)
        dec.decode(pkt, planes)
        big.append((w, h, Q, amp, n, pkt.size, R.digest(f), R.digest(pkt), R.digest(planes)))
    out["big_meta"] = np.array([b[:6] for b in big], np.int64)
    out["big_digests"] = np.array([b[6:] for b in big])

    path = os.path.join(HERE, "rtjpeg_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
