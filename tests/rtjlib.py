"""Test-side helpers: ctypes bindings for the CPU oracle (oracle/librtj_oracle.so),
for the reference's own lib/RTjpeg.c when oracle/_ref/librtjpeg_ref.so exists, and the
numpy twin of the product's synthetic-frame generator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(os.environ.get("MI_SAN_LIBDIR") or ORACLE_DIR, "librtj_oracle.so")  # MI_SAN_LIBDIR: sanitizer build
REF_SO = os.path.join(ORACLE_DIR, "_ref", "librtjpeg_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

u8p = C.POINTER(C.c_uint8)


def _ptr(a):
    return a.ctypes.data_as(u8p)


def build_oracle():
    """(Re)build the oracle; cheap, and also builds oracle/_ref when /root/reference exists."""
    src = os.path.join(ORACLE_DIR, "rtj_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


class RtjoTables(C.Structure):
    _fields_ = [("lqt", C.c_int32 * 64), ("cqt", C.c_int32 * 64),
                ("liqt", C.c_int32 * 64), ("ciqt", C.c_int32 * 64),
                ("lb8", C.c_int), ("cb8", C.c_int)]


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.rtjo_make_tables.argtypes = [C.c_int, C.POINTER(RtjoTables)]
        L.rtjo_dec_new.restype = C.c_void_p
        L.rtjo_dec_free.argtypes = [C.c_void_p]
        L.rtjo_decode.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, u8p, u8p]
        L.rtjo_decode.restype = C.c_long
        L.rtjo_block_offsets.argtypes = [C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.rtjo_block_offsets.restype = C.c_long
        L.rtjo_dec_quality.argtypes = [C.c_void_p]
        L.rtjo_enc_new.argtypes = [C.c_int] * 6
        L.rtjo_enc_new.restype = C.c_void_p
        L.rtjo_enc_free.argtypes = [C.c_void_p]
        L.rtjo_encode.argtypes = [C.c_void_p, u8p, u8p, u8p, u8p]
        L.rtjo_encode.restype = C.c_long
        L.rtjo_s2b.argtypes = [u8p, C.c_size_t, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int16)]
        L.rtjo_idct.argtypes = [C.POINTER(C.c_int16), u8p, C.c_int]
        L.rtjo_yuv420_to_rgb.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, u8p, C.c_size_t]
        _oracle = L
    return _oracle


def oracle_tables(Q):
    t = RtjoTables()
    oracle().rtjo_make_tables(Q, C.byref(t))
    return (np.array(t.liqt, dtype=np.int32), np.array(t.ciqt, dtype=np.int32), t.lb8, t.cb8,
            np.array(t.lqt, dtype=np.int32), np.array(t.cqt, dtype=np.int32))


def plane_sizes(w, h):
    return w * h, (w // 2) * (h // 2)


def split_planes(buf, w, h):
    ys, cs = plane_sizes(w, h)
    return buf[:ys], buf[ys:ys + cs], buf[ys + cs:ys + 2 * cs]


RGB_BPP = (4, 4, 3, 3, 2)
REF_RGB_FUNCS = ("RTjpeg_yuv420rgb32", "RTjpeg_yuv420bgr32", "RTjpeg_yuv420rgb24", "RTjpeg_yuv420bgr24",
                 "RTjpeg_yuv420rgb16")


def oracle_to_rgb(fmt, w, h, planes, dst, pitch):
    """dst: uint8 array of h*pitch bytes, updated in place (byte 3 of 32-bit pixels is left alone)."""
    y, u, v = split_planes(planes, w, h)
    oracle().rtjo_yuv420_to_rgb(fmt, w, h, _ptr(y), _ptr(u), _ptr(v), _ptr(dst), pitch)


def reference_to_rgb(fmt, w, h, planes, dst, pitch):
    """The reference's own RTjpeg_yuv420* helper; it takes an array of row pointers."""
    L = reference()
    rtj = L.RTjpeg_init()
    cw, ch = C.c_int(w), C.c_int(h)
    L.RTjpeg_set_size(rtj, C.byref(cw), C.byref(ch))
    rows = (u8p * h)(*[C.cast(dst.ctypes.data + r * pitch, u8p) for r in range(h)])
    fn = getattr(L, REF_RGB_FUNCS[fmt])
    fn.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(u8p)]
    fn.restype = None
    fn(rtj, _planes_arg(planes, w, h), rows)
    L.RTjpeg_close(rtj)


class OracleDecoder:
    """Stateful, like one RTjpeg_t used by decode_rtjpeg (video_rtjpeg.c:62-90)."""

    def __init__(self):
        self.L = oracle()
        self.h = self.L.rtjo_dec_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rtjo_dec_free(self.h)
            self.h = None

    def decode(self, pkt, out):
        """pkt: uint8 array (whole packet); out: uint8 array of 1.5*w*h, updated in place."""
        pkt = np.ascontiguousarray(pkt, dtype=np.uint8)
        w = int(pkt[6]) | (int(pkt[7]) << 8)
        h = int(pkt[8]) | (int(pkt[9]) << 8)
        y, u, v = split_planes(out, w, h)
        r = self.L.rtjo_decode(self.h, _ptr(pkt), pkt.size, _ptr(y), _ptr(u), _ptr(v))
        return r

    def block_offsets(self, pkt):
        pkt = np.ascontiguousarray(pkt, dtype=np.uint8)
        w = int(pkt[6]) | (int(pkt[7]) << 8)
        h = int(pkt[8]) | (int(pkt[9]) << 8)
        n = (w // 16) * (h // 16) * 6
        offs = np.zeros(n + 1, dtype=np.uint32)
        r = self.L.rtjo_block_offsets(self.h, _ptr(pkt), pkt.size,
                                      offs.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert r == n, (r, n)
        return offs

    def quality(self):
        return self.L.rtjo_dec_quality(self.h)


class OracleEncoder:
    def __init__(self, w, h, Q, key_rate=0, lmask=0, cmask=0):
        self.L = oracle()
        self.w, self.h_ = w, h
        self.h = self.L.rtjo_enc_new(w, h, Q, key_rate, lmask, cmask)
        assert self.h, "bad encoder geometry"
        self.buf = np.zeros(12 + (w // 16) * (h // 16) * 6 * 64, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rtjo_enc_free(self.h)
            self.h = None

    def encode(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        y, u, v = split_planes(frame, self.w, self.h_)
        n = self.L.rtjo_encode(self.h, _ptr(y), _ptr(u), _ptr(v), _ptr(self.buf))
        return self.buf[:n].copy()


# ---------------------------------------------------------------------------
# the reference's own code (only where oracle/_ref was built, i.e. this container)
# ---------------------------------------------------------------------------
def have_reference():
    return os.path.exists(REF_SO)


_ref = None


def reference():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.RTjpeg_init.restype = C.c_void_p
        L.RTjpeg_close.argtypes = [C.c_void_p]
        ip = C.POINTER(C.c_int)
        L.RTjpeg_set_quality.argtypes = [C.c_void_p, ip]
        L.RTjpeg_set_size.argtypes = [C.c_void_p, ip, ip]
        L.RTjpeg_set_intra.argtypes = [C.c_void_p, ip, ip, ip]
        L.RTjpeg_compress.argtypes = [C.c_void_p, u8p, C.POINTER(u8p)]
        L.RTjpeg_compress.restype = C.c_int
        L.RTjpeg_decompress.argtypes = [C.c_void_p, u8p, C.POINTER(u8p)]
        L.RTjpeg_get_tables.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        _ref = L
    return _ref


def _planes_arg(buf, w, h):
    y, u, v = split_planes(buf, w, h)
    arr = (u8p * 3)(_ptr(y), _ptr(u), _ptr(v))
    return arr


class RefCodec:
    """Thin driver of the reference's RTjpeg_t (lib/RTjpeg.c), as video_rtjpeg.c uses it."""

    def __init__(self):
        self.L = reference()
        self.h = self.L.RTjpeg_init()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.RTjpeg_close(self.h)
            self.h = None

    def setup_encoder(self, w, h, Q, key_rate=0, lmask=0, cmask=0):
        self.w, self.h_ = w, h
        cw, ch, cq = C.c_int(w), C.c_int(h), C.c_int(Q)
        self.L.RTjpeg_set_size(self.h, C.byref(cw), C.byref(ch))
        self.L.RTjpeg_set_quality(self.h, C.byref(cq))
        if key_rate > 0:
            k, l, c = C.c_int(key_rate), C.c_int(lmask), C.c_int(cmask)
            self.L.RTjpeg_set_intra(self.h, C.byref(k), C.byref(l), C.byref(c))
        self.buf = np.zeros(64 + (w // 16) * (h // 16) * 6 * 64, dtype=np.uint8)

    def encode(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        n = self.L.RTjpeg_compress(self.h, _ptr(self.buf), _planes_arg(frame, self.w, self.h_))
        return self.buf[:n].copy()

    def decode(self, pkt, out):
        # The reference reads past the packet end on malformed input; give it zero padding.
        pad = np.zeros(pkt.size + 4096, dtype=np.uint8)
        pad[:pkt.size] = pkt
        w = int(pkt[6]) | (int(pkt[7]) << 8)
        h = int(pkt[8]) | (int(pkt[9]) << 8)
        self.L.RTjpeg_decompress(self.h, _ptr(pad), _planes_arg(out, w, h))

    def tables(self, Q):
        cq = C.c_int(Q)
        self.L.RTjpeg_set_quality(self.h, C.byref(cq))
        t = (C.c_uint32 * 128)()
        self.L.RTjpeg_get_tables(self.h, t)
        a = np.array(t, dtype=np.uint32).astype(np.int64).astype(np.int32)
        return a[:64], a[64:]


# ---------------------------------------------------------------------------
# synthetic content — numpy twin of the product's generator (csrc/rtj_synth.hip)
# ---------------------------------------------------------------------------
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x * np.uint32(0xC2B2AE35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def synth_frame(w, h, n, seed=12345, amp=8):
    """Gradient + hashed uniform noise (SURVEY.md §8d cfg 2 content, counter-based):
    Y = 16 + ((x+y+7n) mod (w+h))*219/(w+h) + U[-amp,amp];  C = 128 + U[-amp/2,amp/2]."""
    with np.errstate(over="ignore"):
        def noise(plane, count, a):
            idx = np.arange(count, dtype=np.uint32)
            key = (np.uint32(seed) * np.uint32(0x9E3779B1) + np.uint32(n) * np.uint32(0x7FEB352D)
                   + np.uint32(plane) * np.uint32(0x846CA68B)).astype(np.uint32)
            hsh = _mix32(idx + key)
            return (hsh % np.uint32(2 * a + 1)).astype(np.int32) - a

        xs = np.arange(w, dtype=np.int64)[None, :]
        ys = np.arange(h, dtype=np.int64)[:, None]
        base = 16 + ((xs + ys + 7 * n) % (w + h)) * 219 // (w + h)
        y = np.clip(base.reshape(-1) + noise(0, w * h, amp), 0, 255).astype(np.uint8)
        cs = (w // 2) * (h // 2)
        u = np.clip(128 + noise(1, cs, amp // 2), 0, 255).astype(np.uint8)
        v = np.clip(128 + noise(2, cs, amp // 2), 0, 255).astype(np.uint8)
    return np.concatenate([y, u, v])


def synth_frame_lcg(w, h, n, seed=12345, amp=8):
    """SURVEY.md 8d / BASELINE.md section 2's generator: the gradient of synth_frame with the noise of ONE linear
    congruential sequence s <- s * 1664525 + 1013904223 over all frames (frame n starts at draw n * 1.5 w h)."""
    fsz = w * h * 3 // 2
    A, Cc, M = 1664525, 1013904223, 1 << 32
    # skip ahead to the state before frame n's first draw
    k, a, c, sa, sc = n * fsz, A, Cc, 1, 0
    while k:
        if k & 1:
            sa, sc = (sa * a) % M, (sc * a + c) % M
        c = (c * (a + 1)) % M
        a = (a * a) % M
        k >>= 1
    s = (sa * seed + sc) % M
    # the fsz states that follow, vectorised: s_j = A^j s + C (A^j - 1)/(A - 1), by doubling blocks
    st = np.empty(fsz, np.uint64)
    cur_a, cur_c, filled = np.uint64(A), np.uint64(Cc), 1
    st[0] = (np.uint64(s) * np.uint64(A) + np.uint64(Cc)) & np.uint64(M - 1)
    while filled < fsz:
        m = min(filled, fsz - filled)
        st[filled:filled + m] = (st[:m] * cur_a + cur_c) & np.uint64(M - 1)
        cur_c = (cur_c * (cur_a + np.uint64(1))) & np.uint64(M - 1)
        cur_a = (cur_a * cur_a) & np.uint64(M - 1)
        filled += m
    draw = (st >> np.uint64(8)).astype(np.int64)
    xs = np.arange(w, dtype=np.int64)[None, :]
    ys = np.arange(h, dtype=np.int64)[:, None]
    base = (16 + ((xs + ys + 7 * n) % (w + h)) * 219 // (w + h)).reshape(-1)
    y = np.clip(base + draw[:w * h] % (2 * amp + 1) - amp, 0, 255).astype(np.uint8)
    ca = amp // 2
    c2 = np.clip(128 + draw[w * h:] % (2 * ca + 1) - ca, 0, 255).astype(np.uint8)
    return np.concatenate([y, c2])


def digest(a):
    """Plane digest used by the golden fixtures (first 128 bits of SHA-256)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint8).tobytes()).hexdigest()[:32]
