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
