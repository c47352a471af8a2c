"""
CPU test: numerical validation of the division-free r2-mask CERTIFICATE of the fused kernel (DESIGN.md appendix A;
hk_fit_kernel.h stage B) on a numpy model of the same float32 expression.

For millions of random windows -- incl. low-variance / large-mean, wildly scaled, marginal-R2 and integer data -- the
reference's own arithmetic (kernel_model.py:338-351 gains/offsets, :179-195 R2, :363 mask; float64 sums of squares as
cv2.sqrBoxFilter returns them) must say "passes" wherever the certificate says so.  The certificate may only err on the
side of "not sure" (exact evaluation).  A slack constant 2^-24 instead of the proven 2^-17 does produce false
certificates on this data, i.e. the data does probe the bound.
"""
import zlib

import numpy as np
import pytest

F32, F64 = np.float32, np.float64


def _fma32(a, b, c):
    return (np.asarray(a, F64) * np.asarray(b, F64) + np.asarray(c, F64)).astype(F32)


def reference_window_math(s, r):
    """ s, r: (M, N) float32 windows, all pixels valid -> the reference's per-window quantities. """
    n = s.shape[1]
    nf, nd = F32(n), F64(n)
    hs, hr = s.astype(F64).sum(1), r.astype(F64).sum(1)
    hp = (s * r).astype(F32).astype(F64).sum(1)
    hs2, hr2 = (s.astype(F64) ** 2).sum(1), (r.astype(F64) ** 2).sum(1)
    sf, rf, pf = hs.astype(F32), hr.astype(F32), hp.astype(F32)
    with np.errstate(all='ignore'):
        num = ((nf * pf).astype(F32) - (sf * rf).astype(F32)).astype(F32)
        den = nd * hs2 - (sf * sf).astype(F32).astype(F64)
        g = (num.astype(F64) / den).astype(F32)
        o = ((rf - (g * sf).astype(F32)).astype(F32) / nf).astype(F32)
        sstot = nd * hr2 - (rf * rf).astype(F32).astype(F64)
        q = (g * g).astype(F32).astype(F64) * hs2
        q = q + ((F32(2) * (g * o).astype(F32)).astype(F32) * sf).astype(F32).astype(F64)
        q = q - ((F32(2) * g).astype(F32) * pf).astype(F32).astype(F64)
        q = q - ((F32(2) * o).astype(F32) * rf).astype(F32).astype(F64)
        q = q + hr2
        q = q + (nf * (o * o).astype(F32)).astype(F32).astype(F64)
        r2 = (F32(1) - (q * nd / sstot).astype(F32)).astype(F32)
        t = (g * sf).astype(F32)
        tn = (rf - t).astype(F32)
    return dict(n=nf, g=g, o=o, sf=sf, rf=rf, hs2=hs2, hr2=hr2, r2=r2, num=num, t=t, tn=tn)


def certificate(q, kappa, k2=F32(2.0 ** -17), r2_rel_err=0.0):
    """ The kernel's expression, operation for operation (hk_fit_kernel.h, stage A/B of the gain-offset kernels):
    fl32(g * num) > kappa * sst + 2^-17 * N*T', with N*T' = g*num + t^2 + N*R2 + tn^2 (t = g*S, tn = R - t).
    r2_rel_err: the certificate-only build sums ref^2 horizontally in float32 (rounded column sums, <= 5 roundings of
    non-negative terms): its float32 window sum is within 4.03 * 2^-24 of the float64 one instead of 2^-24 -- modelled as
    that relative perturbation of the exact sum (sign = the argument's) before the float32 rounding. """
    g, nf, rf = q['g'], q['n'], q['rf']
    with np.errstate(all='ignore'):
        r2f = (q['hr2'] * (1.0 + r2_rel_err)).astype(F32)
        nfull = np.full_like(r2f, nf)
        lhs = (g * q['num']).astype(F32)
        sst = _fma32(nfull, r2f, -(rf * rf).astype(F32))
        nt = _fma32(q['t'], q['t'], _fma32(q['tn'], q['tn'], _fma32(nfull, r2f, lhs)))
        slack = (k2 * nt).astype(F32)
        rhs = _fma32(np.full_like(sst, kappa), sst, slack)
        g_in = (g > F32(2.0 ** -20)) & (g < F32(2.0 ** 20))
        t_in = (nt > F32(2.0 ** -40)) & (nt < F32(2.0 ** 60))
        return (lhs > rhs) & (sst > slack) & g_in & t_in


def fail_certificate(q, kappa_f, k2=F32(2.0 ** -17)):
    """ The mirror image (complete build only, PROOFS.md appendix A, "the fail side"): a pixel certainly FAILS
    `(r2 > thresh) & (gain > 0)` if its gain is not positive (the gain is bit-exact), or if
    0 < fl32(g * num) < kappa_f * sst - 2^-17 * N*T' with sst > 2^-17 * N*T' inside the same magnitude windows. """
    g, nf, rf = q['g'], q['n'], q['rf']
    with np.errstate(all='ignore'):
        r2f = q['hr2'].astype(F32)
        nfull = np.full_like(r2f, nf)
        lhs = (g * q['num']).astype(F32)
        sst = _fma32(nfull, r2f, -(rf * rf).astype(F32))
        nt = _fma32(q['t'], q['t'], _fma32(q['tn'], q['tn'], _fma32(nfull, r2f, lhs)))
        slack = (k2 * nt).astype(F32)
        rhs_f = _fma32(np.full_like(sst, kappa_f), sst, -slack)
        g_in = (g > F32(2.0 ** -20)) & (g < F32(2.0 ** 20))
        t_in = (nt > F32(2.0 ** -40)) & (nt < F32(2.0 ** 60))
        gpos = g > 0
        return ~gpos | ((lhs < rhs_f) & (lhs > 0) & (sst > slack) & g_in & t_in)


def kappa_fail_for(thresh):
    """ Mirror of hk_api.hip r2_failcert_scale(): 1 - c_hi rounded DOWN to float32 (c_hi = r2_fail_above, 2^-40 above the float32
    rounding boundary of the quotient), -inf where failure cannot be certified. """
    t = F32(thresh)
    if not (t < 1):
        return F32(-np.inf)
    q = F32(1) - t
    for _ in range(8):
        q = np.nextafter(q, F32(np.inf))
    while not (F32(1) - q > t):
        q = np.nextafter(q, F32(-np.inf))
    boundary = 0.5 * (float(q) + float(np.nextafter(q, F32(np.inf))))
    if not boundary > 0:
        return F32(-np.inf)
    c_hi = boundary * (1 + 2.0 ** -40)
    k = 1 - c_hi * (1 + 2.0 ** -50)
    if not k > 0:
        return F32(-np.inf)
    kf = F32(k)
    if float(kf) > k - 2.0 ** -60:
        kf = np.nextafter(kf, F32(-np.inf))
    return kf


def fast_quotient(num, den):
    """ Model of hk_fit_kernel.h fast_quot() + quot_guard(): float32(RN64(num/den)) from a reciprocal of relative error
    <= 2^-22 and one Newton step; returns (float32 result, needs_ieee_division). """
    with np.errstate(all='ignore'):
        y = 1.0 / den
        # degrade the reciprocal to the hardware's documented accuracy (worst case): perturb by up to 2^-22 relative
        rng = np.random.default_rng(99)
        y0 = y * (1.0 + rng.uniform(-1, 1, y.shape) * 2.0 ** -22)
        e = 1.0 - den * y0                       # one rounding each, like the two fmas
        y1 = y0 + y0 * e
        q = num * y1
        lo = q.view(np.uint64) & np.uint64(0x1fffffff)
        guard = (lo.astype(np.int64) - (0x10000000 - 1024)) & 0xffffffff
        hi = (q.view(np.uint64) >> np.uint64(32)).astype(np.int64)
        e = hi & 0x7ff00000
        rng_w = np.where(e == 0, 0, (e - 0x38100000) & 0xffffffff)      # a zero quotient is exact
        again = (guard < 2049) | (rng_w > 0x0fd00000)
        return q.astype(F32), again


def kappa_for(thresh):
    """ Mirror of hk_api.hip r2_pass_scale() / r2_fail_scale(). """
    t = F32(thresh)
    q = F32(1) - t
    for _ in range(8):
        q = np.nextafter(q, F32(np.inf))
    while not (F32(1) - q > t):
        q = np.nextafter(q, F32(-np.inf))
    boundary = 0.5 * (float(q) + float(np.nextafter(q, F32(np.inf))))
    c = boundary * (1 - 2.0 ** -40)
    k = max(1 - c * (1 - 2.0 ** -50), 0.0)
    kf = F32(k)
    if float(kf) < k + 2.0 ** -60:
        kf = np.nextafter(kf, F32(np.inf))
    return kf


def _windows(rng, m, n, kind):
    if kind == 'synth':
        s = rng.uniform(0.05, 1, (m, n))
        r = 1.2 * s + 0.05 + rng.normal(0, 0.01, (m, n))
    elif kind == 'lowvar':
        mean = 10 ** rng.uniform(0, 4, (m, 1))
        sd = mean * 10 ** rng.uniform(-5, -1, (m, 1))
        s = mean + sd * rng.normal(size=(m, n))
        r = ((0.5 + rng.random((m, 1))) * s + rng.normal(size=(m, n)) * sd * 10 ** rng.uniform(-3, 0.5, (m, 1))
             + mean * rng.normal(size=(m, 1)))
    elif kind == 'wild':
        s = 10 ** rng.uniform(-8, 8, (m, 1)) * rng.normal(size=(m, n))
        r = (10 ** rng.uniform(-6, 6, (m, 1)) * s * rng.normal(1, 0.1, (m, n))
             + 10 ** rng.uniform(-8, 8, (m, 1)) * rng.normal(size=(m, n)))
    elif kind == 'marginal':
        s = rng.normal(100, 10, (m, n))
        r = s + 10 ** rng.uniform(0.5, 1.6, (m, 1)) * rng.normal(size=(m, n))
    elif kind == 'noisy':   # what bench.py --nodata 3 / 4 look like: weak correlation, many windows fail the r2 mask
        s = rng.uniform(0.05, 1, (m, n))
        r = 1.2 * s + 0.05 + rng.normal(0, 10 ** rng.uniform(-1.5, 0.3, (m, 1)), (m, n))
    elif kind == 'int':
        s = rng.integers(0, 255, (m, n)).astype(float)
        r = rng.integers(0, 4, (m, n)) + np.round(s * rng.uniform(0.3, 2, (m, 1)))
    else:
        raise ValueError(kind)
    return s.astype(F32), r.astype(F32)


@pytest.mark.parametrize('kind', ['synth', 'lowvar', 'wild', 'marginal', 'int'])
@pytest.mark.parametrize('n', [2, 9, 25, 225])
def test_certificate_never_contradicts_the_reference_arithmetic(kind, n):
    rng = np.random.default_rng(zlib.crc32(f'{kind}{n}'.encode()))
    s, r = _windows(rng, 120_000 if n < 100 else 30_000, n, kind)
    q = reference_window_math(s, r)
    n_cert = 0
    for thresh in (0.0, 0.25, 0.5, 0.9, 0.999):
        sure = certificate(q, kappa_for(thresh))
        with np.errstate(all='ignore'):
            passes = (q['r2'] > F32(thresh)) & (q['g'] > 0)
        assert not (sure & ~passes).any()
        n_cert += int(sure.sum())
        # the certificate-only build's float32 horizontal sum of ref^2: worst case either way (3.03 u beyond the rounding)
        for err in (-3.03 * 2.0 ** -24, 3.03 * 2.0 ** -24):
            assert not (certificate(q, kappa_for(thresh), r2_rel_err=err) & ~passes).any()
    if kind in ('synth', 'int') and n >= 9:
        assert n_cert > 0.5 * 4 * len(s)   # and it is not vacuous: well-conditioned data is certified


@pytest.mark.parametrize('kind', ['synth', 'lowvar', 'wild', 'marginal', 'int', 'noisy'])
@pytest.mark.parametrize('n', [2, 9, 25, 225])
def test_fail_certificate_never_contradicts_the_reference_arithmetic(kind, n):
    """ The fail side: wherever it says "certainly fails", the reference's arithmetic says (r2 > thresh) & (gain > 0) is False;
    and it never coincides with the pass certificate. """
    rng = np.random.default_rng(zlib.crc32(f'fail{kind}{n}'.encode()))
    s, r = _windows(rng, 120_000 if n < 100 else 30_000, n, kind)
    q = reference_window_math(s, r)
    n_cert = n_fail = 0
    for thresh in (0.0, 0.25, 0.5, 0.9, 0.999):
        sure_f = fail_certificate(q, kappa_fail_for(thresh))
        sure_p = certificate(q, kappa_for(thresh))
        with np.errstate(all='ignore'):
            passes = (q['r2'] > F32(thresh)) & (q['g'] > 0)
        assert not (sure_f & passes).any()
        assert not (sure_f & sure_p).any()
        n_cert += int(sure_f.sum())
        n_fail += int((~passes).sum())
    if kind in ('noisy', 'marginal') and n >= 9:
        assert n_cert > 0.9 * n_fail   # not vacuous: nearly every failing window is certified


def test_the_data_probes_the_fail_bound():
    rng = np.random.default_rng(15)
    s, r = _windows(rng, 400_000, 25, 'lowvar')
    q = reference_window_math(s, r)
    with np.errstate(all='ignore'):
        passes = (q['r2'] > F32(0.25)) & (q['g'] > 0)
    kf = kappa_fail_for(0.25)
    assert (fail_certificate(q, kf, F32(2.0 ** -24)) & passes).sum() > 0
    assert (fail_certificate(q, kf, F32(2.0 ** -22)) & passes).sum() == 0
    assert (fail_certificate(q, kf) & passes).sum() == 0
    assert not (fail_certificate(q, F32(-np.inf)) & (q['g'] > 0)).any()   # thresh >= 1: only non-positive gains are certain


def test_the_data_probes_the_bound():
    """ With the slack shrunk to 2^-24 (below the rounding errors of the reference expression) false certificates
    appear on the low-variance windows; with the shipped 2^-17 (and even 2^-22) there are none. """
    rng = np.random.default_rng(5)
    s, r = _windows(rng, 400_000, 25, 'lowvar')
    q = reference_window_math(s, r)
    with np.errstate(all='ignore'):
        passes = (q['r2'] > F32(0.25)) & (q['g'] > 0)
    kappa = kappa_for(0.25)
    assert (certificate(q, kappa, F32(2.0 ** -24)) & ~passes).sum() > 0
    assert (certificate(q, kappa, F32(2.0 ** -22)) & ~passes).sum() == 0
    assert (certificate(q, kappa) & ~passes).sum() == 0


def test_nothing_is_certified_outside_the_magnitude_windows_or_for_impossible_thresholds():
    rng = np.random.default_rng(6)
    s, r = _windows(rng, 20_000, 25, 'synth')
    for scale in (1e-18, 1e16):
        q = reference_window_math((s * F32(scale)).astype(F32), (r * F32(scale)).astype(F32))
        assert not certificate(q, kappa_for(0.25)).any()
    q = reference_window_math(s, r)
    assert not certificate(q, F32(np.inf)).any()          # thresh >= 1: r2_fail_scale() = +inf
    assert not certificate(reference_window_math(s, (-r).astype(F32)), kappa_for(0.25)).any()   # negative gains


@pytest.mark.parametrize('kind', ['synth', 'lowvar', 'wild', 'int'])
def test_fast_quotient_guard_catches_every_rounding_difference(kind):
    """ Wherever the guard does NOT ask for the IEEE division, float32(num * y1) equals float32(RN64(num / den)) -- with
    the reciprocal degraded to 2^-22 relative error (the hardware's is 2^-24.4 measured, 2^-23 documented). """
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    s, r = _windows(rng, 400_000, 25, kind)
    n = s.shape[1]
    hs, hr = s.astype(F64).sum(1), r.astype(F64).sum(1)
    hp = (s * r).astype(F32).astype(F64).sum(1)
    hs2 = (s.astype(F64) ** 2).sum(1)
    sf, rf, pf = hs.astype(F32), hr.astype(F32), hp.astype(F32)
    with np.errstate(all='ignore'):
        num = ((F32(n) * pf).astype(F32) - (sf * rf).astype(F32)).astype(F32).astype(F64)
        den = F64(n) * hs2 - (sf * sf).astype(F32).astype(F64)
        exact = (num / den).astype(F32)
    fast, again = fast_quotient(num, den)
    same = (fast == exact) | (np.isnan(fast) & np.isnan(exact))
    assert (same | again).all()
    assert again.mean() < 1e-3 or kind == 'wild'     # and the guard is rare on ordinary data


def test_fast_quotient_special_operands_take_the_ieee_division():
    num = np.array([1.0, 0.0, -1.0, np.nan, 1.0, 1.0, 1e-30, 1e30], F64)
    den = np.array([0.0, 0.0, 0.0, 1.0, np.inf, np.nan, 1e30, 1e-30], F64)
    _, again = fast_quotient(num, den)
    assert again.all()
    fast, again = fast_quotient(np.array([0.0, -0.0], F64), np.array([3.0, 3.0], F64))    # exact zeros stay
    assert not again.any() and (np.signbit(fast) == [False, True]).all()


@pytest.mark.parametrize('thresh', [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.999, 0.9999999, -0.5, 1.0, 1.5])
def test_library_constants_equal_their_numpy_derivation(thresh):
    """ The factors the kernels use (hk_api.hip r2_pass_scale / r2_fail_above / r2_fail_scale / r2_failcert_scale through the host-only
    hk_r2_certificate_constants) are the ones this file's model was validated with. """
    from homonim_amd import _hk
    pb, fa, k, kf = _hk.r2_certificate_constants(thresh)
    if not thresh < 1:
        assert pb == -np.inf and fa == np.inf and k == np.inf and kf == -np.inf
        return
    assert F32(k) == kappa_for(thresh)
    assert F32(kf) == kappa_fail_for(thresh) or (np.isinf(kf) and np.isinf(kappa_fail_for(thresh)))
    assert 0 < pb < fa and abs(fa / pb - 1) < 2.0 ** -38
    assert float(kf) < 1 - fa <= float(k) + 2.0 ** -20 if np.isfinite(kf) else True
