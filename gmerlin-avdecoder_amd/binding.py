"""ctypes view of include/mi_rtjpeg.h.  No CPU fallback: if the library is missing or there is no
gfx950 device, construction raises MiRtjError with the library's own message."""
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class MiRtjError(RuntimeError):
    pass


def lib_path():
    # MI_RTJ_LIB: an alternative build of the same library (A/B experiments under tools/)
    return os.environ.get("MI_RTJ_LIB") or os.path.join(HERE, "lib", "libmi_rtjpeg.so")


EXPORTS = ["mi_rtj_device_count", "mi_rtj_create", "mi_rtj_destroy", "mi_rtj_last_error", "mi_rtj_decode",
           "mi_rtj_get_state", "mi_rtj_dev_alloc", "mi_rtj_dev_free", "mi_rtj_h2d", "mi_rtj_d2h",
           "mi_rtj_dev_memset", "mi_rtj_sync", "mi_rtj_plan_create", "mi_rtj_plan_destroy",
           "mi_rtj_plan_decode", "mi_rtj_plan_info", "mi_rtj_plan_profile", "mi_rtj_plan_times",
           "mi_rtj_plan_read_index", "mi_rtj_synth_frames", "mi_rtj_synth_frames_lcg", "mi_rtj_encode_bound", "mi_rtj_encode_frames",
           "mi_rtj_get_tables", "mi_rtj_yuv420_to_rgb", "mi_rtj_encode_stream", "mi_rtj_decode_nocopy",
           "mi_rtj_copy_ceiling", "mi_rtj_plan_spec_stats", "mi_rtj_plan_spec_lead", "mi_rtj_plan_decode_form", "mi_rtj_plan_overlapped",
           "mi_rtj_plan_step_times", "mi_rtj_pipe_create", "mi_rtj_pipe_destroy", "mi_rtj_pipe_room",
           "mi_rtj_pipe_pending", "mi_rtj_pipe_submit", "mi_rtj_pipe_next", "mi_rtj_pipe_peek_tag", "mi_rtj_pipe_flush",
           "mi_rtj_pipe_profile", "mi_rtj_pipe_times"]


KERNELS = ("k_index_summarize", "k_index_resolve", "k_index_emit", "k_decode", "k_spec_walk", "k_spec_verify")


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise MiRtjError(f"{p} is missing: run `python gmerlin-avdecoder_amd/build.py` (there is no CPU path)")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.mi_rtj_device_count.restype = C.c_int
    L.mi_rtj_create.argtypes = [C.c_int]
    L.mi_rtj_create.restype = vp
    L.mi_rtj_destroy.argtypes = [vp]
    L.mi_rtj_last_error.argtypes = [vp]
    L.mi_rtj_last_error.restype = C.c_char_p
    L.mi_rtj_decode.argtypes = [vp, u8p, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_int), C.c_int, C.c_int]
    L.mi_rtj_decode_nocopy.argtypes = [vp, u8p, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_int)]
    L.mi_rtj_get_state.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mi_rtj_dev_alloc.argtypes = [vp, C.c_size_t]
    L.mi_rtj_dev_alloc.restype = vp
    L.mi_rtj_dev_free.argtypes = [vp, vp]
    L.mi_rtj_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.mi_rtj_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.mi_rtj_dev_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.mi_rtj_sync.argtypes = [vp]
    L.mi_rtj_plan_create.argtypes = [vp, C.c_int, u8p, u64p, u32p, u64p]
    L.mi_rtj_plan_create.restype = vp
    L.mi_rtj_plan_destroy.argtypes = [vp]
    L.mi_rtj_plan_decode.argtypes = [vp, vp, vp]
    L.mi_rtj_plan_info.argtypes = [vp, C.POINTER(C.c_int), u64p, u64p, u64p]
    L.mi_rtj_plan_profile.argtypes = [vp, C.c_int]
    L.mi_rtj_plan_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mi_rtj_plan_spec_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.mi_rtj_plan_spec_lead.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mi_rtj_plan_decode_form.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    L.mi_rtj_plan_overlapped.argtypes = [vp]
    L.mi_rtj_plan_step_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.mi_rtj_pipe_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.mi_rtj_pipe_create.restype = vp
    L.mi_rtj_pipe_destroy.argtypes = [vp]
    L.mi_rtj_pipe_destroy.restype = None
    L.mi_rtj_pipe_room.argtypes = [vp]
    L.mi_rtj_pipe_pending.argtypes = [vp]
    L.mi_rtj_pipe_submit.argtypes = [vp, u8p, C.c_size_t, C.c_uint64]
    L.mi_rtj_pipe_next.argtypes = [vp, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_uint64)]
    L.mi_rtj_pipe_peek_tag.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mi_rtj_pipe_flush.argtypes = [vp]
    L.mi_rtj_pipe_profile.argtypes = [vp, C.c_int]
    L.mi_rtj_pipe_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mi_rtj_plan_read_index.argtypes = [vp, u32p, C.c_size_t]
    L.mi_rtj_synth_frames.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, vp]
    L.mi_rtj_encode_bound.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.mi_rtj_encode_bound.restype = C.c_size_t
    L.mi_rtj_encode_frames.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, u64p, u32p]
    L.mi_rtj_yuv420_to_rgb.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t,
                                       C.c_size_t]
    L.mi_rtj_encode_stream.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                       C.c_int, u64p, u32p]
    L.mi_rtj_get_tables.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mi_rtj_copy_ceiling.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    _LIB = L
    return L


def device_count():
    return load().mi_rtj_device_count()


def get_tables(Q):
    t = (C.c_int32 * 128)()
    lb8, cb8 = C.c_int(), C.c_int()
    rc = load().mi_rtj_get_tables(Q, t, C.byref(lb8), C.byref(cb8))
    if rc != 0:
        raise MiRtjError(f"mi_rtj_get_tables({Q}) -> {rc}")
    a = np.array(t, dtype=np.int32)
    return a[:64], a[64:], lb8.value, cb8.value


class Plan:
    def __init__(self, owner, handle, n):
        self.owner, self.h, self.n = owner, handle, n
        owner._plans.add(self)

    def close(self):
        if self.h:
            self.owner.L.mi_rtj_plan_destroy(self.h)
            self.h = None
            self.owner._plans.discard(self)

    def __del__(self):
        self.close()

    def decode(self, d_stream, d_out):
        self.owner._chk(self.owner.L.mi_rtj_plan_decode(self.h, d_stream, d_out))

    def info(self):
        n = C.c_int()
        nb, bi, bo = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.owner.L.mi_rtj_plan_info(self.h, C.byref(n), C.byref(nb), C.byref(bi), C.byref(bo))
        return dict(frames=n.value, blocks=nb.value, bytes_in=bi.value, bytes_out=bo.value)

    def profile(self, on=True):
        self.owner.L.mi_rtj_plan_profile(self.h, 1 if on else 0)

    def spec_stats(self):
        """(packets proven by the speculative index in the last decode, stream chunks it covered; 0 = unused)."""
        pr, wk, rp = C.c_int(), C.c_longlong(), C.c_longlong()
        self.owner._chk(self.owner.L.mi_rtj_plan_spec_stats(self.h, C.byref(pr), C.byref(wk), C.byref(rp)))
        self.repaired = rp.value
        return pr.value, wk.value

    def step_times(self, max_steps=4096):
        """device milliseconds of every decode since profile(True), first kernel's start to k_decode's end."""
        ms = (C.c_float * max_steps)()
        n = C.c_int()
        self.owner._chk(self.owner.L.mi_rtj_plan_step_times(self.h, ms, max_steps, C.byref(n)))
        return [float(ms[i]) for i in range(min(n.value, max_steps))]

    def decode_form(self):
        """(form of the transform kernel the policy chose for the next batch decode: 0 split / 1 classic / -1 none yet,
        classic decodes left, group parts the last decode left to k_decode_list)."""
        form, left, listed = C.c_int(), C.c_int(), C.c_longlong()
        self.owner._chk(self.owner.L.mi_rtj_plan_decode_form(self.h, C.byref(form), C.byref(left), C.byref(listed)))
        return form.value, left.value, listed.value

    def overlapped(self):
        """True if the last decode built its index on the plan's own stream, next to the previous decode's transform."""
        return bool(self.owner.L.mi_rtj_plan_overlapped(self.h))

    def spec_lead(self):
        """(bytes a walker parses before its chunk in the next decode, decodes left without speculation)."""
        lead, paused = C.c_int(), C.c_int()
        self.owner._chk(self.owner.L.mi_rtj_plan_spec_lead(self.h, C.byref(lead), C.byref(paused)))
        return lead.value, paused.value

    def times(self):
        ms = (C.c_float * len(KERNELS))()
        n = C.c_int()
        self.owner._chk(self.owner.L.mi_rtj_plan_times(self.h, ms, C.byref(n)))
        return dict(zip(KERNELS, [float(x) for x in ms])), n.value

    def read_index(self):
        cnt = self.info()["blocks"] + self.n
        a = np.zeros(cnt, dtype=np.uint32)
        self.owner._chk(self.owner.L.mi_rtj_plan_read_index(self.h, a.ctypes.data_as(u32p), cnt))
        return a


class Pipe:
    """View of mi_rtj_pipe (include/mi_rtjpeg.h, "pipelined session") for tests and bench.py."""

    def __init__(self, owner, depth, coded_w, coded_h):
        self.owner = owner
        self.h = owner.L.mi_rtj_pipe_create(owner.h, depth, coded_w, coded_h)
        if not self.h:
            raise MiRtjError("mi_rtj_pipe_create failed: " + owner.L.mi_rtj_last_error(owner.h).decode())

    def close(self):
        if self.h:
            self.owner.L.mi_rtj_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def room(self):
        return self.owner.L.mi_rtj_pipe_room(self.h)

    def pending(self):
        return self.owner.L.mi_rtj_pipe_pending(self.h)

    def submit(self, pkt, tag=0):
        pkt = np.ascontiguousarray(pkt, dtype=np.uint8)
        self.owner._chk(self.owner.L.mi_rtj_pipe_submit(self.h, pkt.ctypes.data_as(u8p), pkt.size, tag))

    def next(self, drop=False):
        """(y, u, v, tag): numpy views of the session's pinned picture, valid until the next call; drop: tag only."""
        tag = C.c_uint64()
        if drop:
            self.owner._chk(self.owner.L.mi_rtj_pipe_next(self.h, None, None, None, None, C.byref(tag)))
            return tag.value
        planes = (u8p * 3)()
        st = (C.c_int * 3)()
        w, h = C.c_int(), C.c_int()
        self.owner._chk(self.owner.L.mi_rtj_pipe_next(self.h, planes, st, C.byref(w), C.byref(h), C.byref(tag)))
        mk = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
        n = w.value * h.value
        return mk(planes[0], n), mk(planes[1], n // 4), mk(planes[2], n // 4), tag.value

    def flush(self):
        self.owner._chk(self.owner.L.mi_rtj_pipe_flush(self.h))

    def profile(self, enable):
        """kernel timing of the packets decoded from here on (flushes the session; not for timed regions)"""
        self.owner._chk(self.owner.L.mi_rtj_pipe_profile(self.h, 1 if enable else 0))

    def times(self):
        """({kernel: ms summed over the packets decoded since profile(True)}, packets)"""
        ms = (C.c_float * len(KERNELS))()
        n = C.c_int()
        self.owner._chk(self.owner.L.mi_rtj_pipe_times(self.h, ms, C.byref(n)))
        return dict(zip(KERNELS, [float(x) for x in ms])), n.value


class MiRtj:
    """One decoder instance (== one RTjpeg_t of the reference) bound to one device."""

    def __init__(self, device=-1):
        self.L = load()
        self._plans = weakref.WeakSet()
        self.h = self.L.mi_rtj_create(device)
        if not self.h:
            raise MiRtjError("mi_rtj_create failed: " + self.L.mi_rtj_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            for p in list(self._plans):  # a plan must not outlive its instance
                p.close()
            self.L.mi_rtj_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise MiRtjError(f"rc={rc}: " + self.L.mi_rtj_last_error(self.h).decode())

    # -- one packet in, one frame out (decode_rtjpeg) --
    def decode(self, pkt, out=None, crop=None, strides=None):
        """pkt: uint8 array.  out: uint8 array receiving Y,U,V back to back (with `strides`, each plane
        has its own row pitch).  Returns rc-checked None."""
        pkt = np.ascontiguousarray(pkt, dtype=np.uint8)
        pp = pkt.ctypes.data_as(u8p)
        if out is None:
            self._chk(self.L.mi_rtj_decode(self.h, pp, pkt.size, None, None, 0, 0))
            return
        w = int(pkt[6]) | (int(pkt[7]) << 8)
        h = int(pkt[8]) | (int(pkt[9]) << 8)
        cw, ch = crop if crop else (w, h)
        if strides is None:
            strides = (cw, (cw + 1) // 2, (cw + 1) // 2)
        ysz = strides[0] * ch
        csz = strides[1] * ((ch + 1) // 2)
        assert out.size >= ysz + 2 * csz
        base = out.ctypes.data
        planes = (u8p * 3)(C.cast(base, u8p), C.cast(base + ysz, u8p), C.cast(base + ysz + csz, u8p))
        st = (C.c_int * 3)(*strides)
        self._chk(self.L.mi_rtj_decode(self.h, pp, pkt.size, planes, st, cw, ch))

    def decode_nocopy(self, pkt):
        """Returns (y, u, v) numpy views of the instance's pinned picture, valid until the next decode."""
        pkt = np.ascontiguousarray(pkt, dtype=np.uint8)
        planes = (u8p * 3)()
        st = (C.c_int * 3)()
        self._chk(self.L.mi_rtj_decode_nocopy(self.h, pkt.ctypes.data_as(u8p), pkt.size, planes, st))
        w = int(pkt[6]) | (int(pkt[7]) << 8)
        h = int(pkt[8]) | (int(pkt[9]) << 8)
        mk = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
        return mk(planes[0], w * h), mk(planes[1], w * h // 4), mk(planes[2], w * h // 4)

    def pipe(self, depth=4, coded_w=0, coded_h=0):
        """A pipelined session (mi_rtj_pipe_*): packets in, pictures out in order, several in flight."""
        return Pipe(self, depth, coded_w, coded_h)

    def state(self):
        w, h, q = C.c_int(), C.c_int(), C.c_int()
        self.L.mi_rtj_get_state(self.h, C.byref(w), C.byref(h), C.byref(q))
        return w.value, h.value, q.value

    # -- device memory --
    def alloc(self, nbytes):
        p = self.L.mi_rtj_dev_alloc(self.h, nbytes)
        if not p:
            raise MiRtjError("alloc failed: " + self.L.mi_rtj_last_error(self.h).decode())
        return p

    def free(self, p):
        self.L.mi_rtj_dev_free(self.h, p)

    def h2d(self, dptr, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.mi_rtj_h2d(self.h, dptr + offset, arr.ctypes.data, arr.nbytes))

    def d2h(self, dptr, nbytes, offset=0, dtype=np.uint8):
        a = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(self.L.mi_rtj_d2h(self.h, a.ctypes.data, dptr + offset, nbytes))
        return a

    def memset(self, dptr, value, nbytes, offset=0):
        self._chk(self.L.mi_rtj_dev_memset(self.h, dptr + offset, value, nbytes))

    def sync(self):
        self._chk(self.L.mi_rtj_sync(self.h))

    # -- batches --
    def plan(self, headers, pkt_offset, pkt_len, out_offset):
        headers = np.ascontiguousarray(headers, dtype=np.uint8).reshape(-1)
        n = headers.size // 12
        po = np.ascontiguousarray(pkt_offset, dtype=np.uint64)
        pl = np.ascontiguousarray(pkt_len, dtype=np.uint32)
        oo = np.ascontiguousarray(out_offset, dtype=np.uint64)
        assert po.size == n and pl.size == n and oo.size == n
        h = self.L.mi_rtj_plan_create(self.h, n, headers.ctypes.data_as(u8p), po.ctypes.data_as(u64p),
                                      pl.ctypes.data_as(u32p), oo.ctypes.data_as(u64p))
        if not h:
            raise MiRtjError("plan_create failed: " + self.L.mi_rtj_last_error(self.h).decode())
        return Plan(self, h, n)

    def upload_packets(self, pkts, align=1):
        """Host packets -> one device stream buffer.  Returns (dptr, offsets, lens, headers)."""
        offs, lens, cur = [], [], 0
        for p in pkts:
            cur = (cur + align - 1) // align * align
            offs.append(cur)
            lens.append(p.size)
            cur += p.size
        host = np.zeros(max(cur, 1), dtype=np.uint8)
        hdrs = np.zeros((len(pkts), 12), dtype=np.uint8)
        for i, p in enumerate(pkts):
            host[offs[i]:offs[i] + p.size] = p
            m = min(12, p.size)
            hdrs[i, :m] = p[:m]
        d = self.alloc(host.size)
        self.h2d(d, host)
        return d, np.array(offs, np.uint64), np.array(lens, np.uint32), hdrs

    # -- colour stage (N2) --
    def to_rgb(self, fmt, w, h, n, d_planes, in_stride, d_rgb, row_pitch, out_stride):
        self._chk(self.L.mi_rtj_yuv420_to_rgb(self.h, fmt, w, h, n, d_planes, in_stride, d_rgb, row_pitch, out_stride))

    def copy_ceiling(self, d_src, d_dst, nbytes, reps=10):
        """GB/s (read + write) a plain streaming copy kernel sustains on this device."""
        g = C.c_double()
        self._chk(self.L.mi_rtj_copy_ceiling(self.h, d_src, d_dst, nbytes, reps, C.byref(g)))
        return g.value

    # -- generator side --
    def synth(self, w, h, first, n, seed=12345, amp=8, dptr=None):
        fsz = w * h * 3 // 2
        d = dptr if dptr is not None else self.alloc(fsz * n)
        self._chk(self.L.mi_rtj_synth_frames(self.h, w, h, first, n, seed, amp, d))
        return d

    def synth_lcg(self, w, h, first, n, seed=12345, amp=8, dptr=None):
        """SURVEY 8d's generator: LCG noise, one sequence over all frames (see include/mi_rtjpeg.h)"""
        fsz = w * h * 3 // 2
        d = dptr if dptr is not None else self.alloc(fsz * n)
        self.L.mi_rtj_synth_frames_lcg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
        self._chk(self.L.mi_rtj_synth_frames_lcg(self.h, w, h, first, n, seed, amp, d))
        return d

    def encode_bound(self, w, h, n, align=64):
        return self.L.mi_rtj_encode_bound(w, h, n, align)

    def encode(self, w, h, Q, n, d_frames, align=64, key_rate=0, lmask=0, cmask=0, d_stream=None):
        """Intra batch (key_rate 0) or one in-order stream with skip blocks (key_rate > 0).  d_stream: a buffer of
        encode_bound() bytes the caller allocated (else one is allocated here)."""
        bound = self.L.mi_rtj_encode_bound(w, h, n, align)
        if d_stream is None:
            d_stream = self.alloc(bound)
        po = np.zeros(n, np.uint64)
        pl = np.zeros(n, np.uint32)
        if key_rate > 0:
            self._chk(self.L.mi_rtj_encode_stream(self.h, w, h, Q, key_rate, lmask, cmask, n, d_frames, d_stream,
                                                  align, po.ctypes.data_as(u64p), pl.ctypes.data_as(u32p)))
        else:
            self._chk(self.L.mi_rtj_encode_frames(self.h, w, h, Q, n, d_frames, d_stream, align,
                                                  po.ctypes.data_as(u64p), pl.ctypes.data_as(u32p)))
        return d_stream, po, pl
