"""Interval proof that every multiplicand of the 8-point inverse butterfly stays inside the signed
24-bit range for ANY int16 coefficient block, in both passes.  The decode kernel relies on it to use
the 24-bit multiplier (v_mad_i32_i24) while staying bit-identical to the reference's 32-bit product
(lib/RTjpeg.c:1206)."""


def mul_bound(b, c):
    # |(x*c + 128) >> 8| <= |x|*|c|/256 + 1
    return b * abs(c) // 256 + 1


def pass_bounds(B):
    """B: bound on |x_i| of the eight inputs.  Returns (max multiplicand, max output)."""
    s04 = d04 = s26 = d26 = 2 * B
    r26 = mul_bound(d26, 362) + s26
    e03 = s04 + s26
    e12 = d04 + r26
    s53 = d53 = s17 = d17 = 2 * B
    o7 = s17 + s53
    m_in = s17 + s53
    m = mul_bound(m_in, 362)
    z5_in = d53 + d17
    z5 = mul_bound(z5_in, 473)
    o6 = mul_bound(d53, 669) + z5 + o7
    o5 = m + o6
    o4 = mul_bound(d17, 277) + z5 + o5
    mult_in = max(d26, m_in, z5_in, d53, d17)
    out = max(e03 + o7, e12 + o6, e12 + o5, e03 + o4)
    return mult_in, out


def test_multiplicands_fit_24_bits():
    lim = 1 << 23
    m1, out1 = pass_bounds(32768 + 4)  # column pass: int16 coefficients (+4 rounding term on DC)
    assert m1 < lim
    m2, out2 = pass_bounds(out1)       # row pass: fed by the column pass
    assert m2 < lim, (m2, lim)
    assert out2 < (1 << 31)


# ---------------------------------------------------------------------------------------------------------
# The packed 16-bit passes of k_decode (csrc/rtj_idct_pk.h): which blocks they may take.
# ---------------------------------------------------------------------------------------------------------
import os
import re

import numpy as np

_K = {362: 362 / 256, 473: 473 / 256, -669: -669 / 256, 277: 277 / 256}
_PK_H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gmerlin-avdecoder_amd", "csrc",
                     "rtj_idct_pk.h")


def _flow(x, mul, half):
    """The 8-point flowgraph on any value type.  Returns (outputs, multiplicands)."""
    x0, x1, x2, x3, x4, x5, x6, x7 = x
    s04, d04, s26, d26 = x0 + x4, x0 - x4, x2 + x6, x2 - x6
    s53, d53, s17, d17 = x5 + x3, x5 - x3, x1 + x7, x1 - x7
    u, w = s17 - s53, d53 + d17
    m1, m2, z5, m4, m5 = mul(d26, 362), mul(u, 362), mul(w, 473), mul(d53, -669), mul(d17, 277)
    r26, o7, e0, e3 = m1 - s26, s17 + s53, s04 + s26, s04 - s26
    e1, e2 = d04 + r26, d04 - r26
    o6 = (m4 + z5) - o7
    o5 = m2 - o6
    o4 = (m5 - z5) + o5
    return [e0 + o7, e1 + o6, e2 + o5, e3 - o4, e3 + o4, e2 - o5, e1 - o6, e0 - o7], [d26, u, w, d53, d17]


def _critical_forms():
    """Linear forms (over the 64 coefficients, row-major) of every value the packed passes need exact: the
    multiplicands of both passes and the 64 results; and the largest rounding error any of them carries."""
    eye = np.eye(64)
    lin = lambda v, k: v * _K[k]
    crit, ws = [], [[None] * 8 for _ in range(8)]
    for c in range(8):
        y, m = _flow([eye[8 * r + c] for r in range(8)], lin, None)
        crit += m
        for r in range(8):
            ws[r][c] = y[r]
    for r in range(8):
        y, m = _flow(ws[r], lin, None)
        crit += m + y

    class Err(float):  # error bounds: every operation adds, a product scales and adds 1/2
        __add__ = __sub__ = lambda a, b: Err(float(a) + float(b))
    err_mul = lambda v, k: Err(float(v) * abs(_K[k]) + 0.5)
    worst, we = 0.0, [[None] * 8 for _ in range(8)]
    for c in range(8):
        y, _ = _flow([Err(0.0)] * 8, err_mul, None)
        for r in range(8):
            we[r][c] = y[r]
    for r in range(8):
        y, m = _flow(we[r], err_mul, None)
        worst = max([worst] + [float(v) for v in y + m])
    return np.array(crit), worst


def _header_constants():
    text = open(_PK_H).read()
    budget = re.search(r"kPkBudget = 4 \* \(32767 - (\d+) - (\d+)\)", text)
    classw = [int(v) for v in re.search(r"kPkClassW\[8\] = \{([^}]*)\}", text).group(1).split(",")]
    rowk = [int(v) for v in re.search(r"rowk\[8\] = \{([^}]*)\}", text).group(1).split(",")]
    cls = [[int(v) for v in row.split(",")] for row in re.findall(r"^\s*\{(\d, \d, \d, \d)\},", text, re.M)]
    assert len(cls) == 5 and len(classw) == 8
    return int(budget.group(1)), int(budget.group(2)), classw, rowk, cls


def test_packed_pass_weights_cover_every_linear_form():
    """kPkClassW[class of dword (pair j, row r)] / 4 is at least the largest factor with which either coefficient
    of that dword enters any multiplicand or result; the slack covers the rounding of all products and the +4."""
    crit, worst = _critical_forms()
    W = np.abs(crit).max(axis=0).reshape(8, 8)
    g = W[:, 0]
    assert np.allclose(W, np.outer(g, g))  # separable: g = 1 1 1 1.18 1 1.77 2.41 5.03
    slack, dc4, classw, rowk, cls = _header_constants()
    assert slack >= worst and dc4 == 4
    for j in range(4):
        for r in range(8):
            assert classw[cls[rowk[r]][j]] >= 4 * max(W[r, 2 * j], W[r, 2 * j + 1]) - 1e-9, (j, r)
    assert (W[:3, :3] <= 1 + 1e-12).all()  # pk_range_lo3: every weight is 1


def _i16(v):
    return ((np.asarray(v, dtype=np.int64) + 32768) % 65536 - 32768).astype(np.int64)


def _packed_model(coef):
    """The packed passes as the kernel computes them: every register half is an int16 (wrap-around adds); a product
    takes the half as it stands, forms the exact x * c + 128 and keeps bits 8..23."""
    mul = lambda v, k: _i16((v * k + 128) >> 8)

    class H:  # an int16 half with wrap-around arithmetic
        def __init__(s, v): s.v = _i16(v)
        __add__ = lambda a, b: H(a.v + b.v)
        __sub__ = lambda a, b: H(a.v - b.v)
    hmul = lambda a, k: H(mul(a.v, k))
    x = [[H(coef[..., 8 * r + c]) for c in range(8)] for r in range(8)]
    x[0][0] = x[0][0] + H(4)
    ws = [[None] * 8 for _ in range(8)]
    for c in range(8):
        y, _ = _flow([x[r][c] for r in range(8)], hmul, None)
        for r in range(8):
            ws[r][c] = y[r]
    out = np.zeros(coef.shape[:-1] + (64,), np.int64)
    for r in range(8):
        y, _ = _flow(ws[r], hmul, None)
        for c in range(8):
            out[..., 8 * r + c] = np.clip(y[c].v >> 3, 16, 235)
    return out.astype(np.uint8)


def _range_ok(coef):
    """pk_range_full of rtj_idct_pk.h (the kernel accumulates 32768 - |half|; the same inequality)."""
    slack, dc4, classw, rowk, cls = _header_constants()
    tot = np.zeros(coef.shape[:-1], np.int64)
    for j in range(4):
        for r in range(8):
            wgt = classw[cls[rowk[r]][j]]
            tot += wgt * (np.abs(coef[..., 8 * r + 2 * j].astype(np.int64)) + np.abs(coef[..., 8 * r + 2 * j + 1].astype(np.int64)))
    return tot <= 4 * (32767 - slack - dc4)


def test_packed_passes_equal_the_reference_transform_inside_the_budget():
    """Random, sparse and adversarial blocks scaled right up to the budget of the range test: the int16 model of the
    packed passes gives the pixels of the oracle's RTjpeg_idct restatement (lib/RTjpeg.c:2209-2332).  Blocks a
    little outside the budget are found that do differ, so the test is not vacuous."""
    import ctypes as C
    import rtjlib as R
    L = R.oracle()
    crit, _ = _critical_forms()
    rng = np.random.default_rng(11)
    slack, dc4, classw, rowk, cls = _header_constants()
    wq = np.zeros(64)
    for j in range(4):
        for r in range(8):
            wq[8 * r + 2 * j] = wq[8 * r + 2 * j + 1] = classw[cls[rowk[r]][j]]
    blocks = []
    for _ in range(300):  # dense random, random density, sign patterns that push one critical value to its extreme
        kind = rng.integers(0, 3)
        if kind == 0:
            b = rng.normal(0, 1, 64)
        elif kind == 1:
            b = rng.normal(0, 1, 64) * (rng.random(64) < rng.random())
        else:
            b = np.sign(crit[rng.integers(0, len(crit))]) * rng.random(64) * (rng.random(64) < 0.5)
        if not np.abs(b).sum():
            b[0] = 1
        for fill in (1.0, 0.999, 0.5):  # scaled to the edge of the budget, and well inside
            s = fill * 4 * (32767 - slack - dc4) / (wq * np.abs(b)).sum()
            blocks.append(np.trunc(b * s))
    coef = np.array(blocks).astype(np.int16)
    ok = _range_ok(coef)
    assert ok.sum() > 600
    got = _packed_model(coef.astype(np.int64))
    for i in np.nonzero(ok)[0]:
        want = np.zeros(64, np.uint8)
        blk = np.ascontiguousarray(coef[i])
        L.rtjo_idct(blk.ctypes.data_as(C.POINTER(C.c_int16)), want.ctypes.data_as(R.u8p), 8)
        assert np.array_equal(got[i], want), i
    # outside the budget the 16-bit halves do overflow
    big = (coef[::3].astype(np.int64) * 8).clip(-32768, 32767).astype(np.int16)
    differ = 0
    gb = _packed_model(big.astype(np.int64))
    for i in range(len(big)):
        want = np.zeros(64, np.uint8)
        blk = np.ascontiguousarray(big[i])
        L.rtjo_idct(blk.ctypes.data_as(C.POINTER(C.c_int16)), want.ctypes.data_as(R.u8p), 8)
        differ += not np.array_equal(gb[i], want)
    assert differ > 0 and not _range_ok(big).any()
