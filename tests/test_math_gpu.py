"""Accuracy of the lean fp64 elementary functions used in the hot loop
(mcf_device.hpp) against numpy, on and beyond the domains the solver produces."""
import numpy as np
import pytest

from microclimf_amd import _abi

pytestmark = pytest.mark.gpu


def run(kind, x, y=None):
    lib = _abi.load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    p = lambda a: None if a is None else a.ctypes.data_as(_abi.c_double_p)
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
    _abi.check(lib.mcf_selftest_math(kind, p(x), p(y), p(out), x.size, 0))
    return out


def relerr(got, want):
    return np.max(np.abs(got - want) / np.abs(want))


def test_exp():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-2, 2, 200000),
                        np.array([0.0, -0.0, 1e-300, -1e-300, 709.0, -745.0])])
    assert relerr(run(0, x), np.exp(x)) < 4e-16
    # very negative arguments (kd*pait with the sun grazing a slope) flush to exactly 0
    big = np.array([-800.0, -1e5, -6e7, -1e15, -1e22])
    assert (run(0, big) == 0.0).all()
    assert np.isnan(run(0, np.array([np.nan]))).all()


def test_log():
    rng = np.random.default_rng(2)
    # (the table route's interval boundaries: mantissa 0.5 + j/512, the switch of reduction target at j = 106, and both
    # sides of 1, where the result must come from log1p(x - 1) alone)
    edges = (0.5 + np.arange(257) / 512.0)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), rng.uniform(0.5, 2.0, 300000), rng.uniform(0.99, 1.01, 100000),
                        1.0 + rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-15, -3, 100000),
                        edges, np.nextafter(edges, 0), np.nextafter(edges, 2), 2 * edges, 8 * np.nextafter(edges, 0),
                        np.array([1.0, 1e-300, 1e300, 0.001, 0.70710678118654746, 0.70710678118654757])])
    got, want = run(1, x), np.log(x)
    err = np.abs(got - want)
    assert np.max(err / np.maximum(np.abs(want), 1e-3)) < 1e-15
    small = np.abs(want) < 1e-3
    assert np.max(err[small]) < 1e-18 and np.max(err[small & (want != 0)] / np.abs(want[small & (want != 0)])) < 4e-16
    assert np.isnan(run(1, np.array([np.nan]))).all()


def test_div_rcp_sqrt():
    rng = np.random.default_rng(3)
    a = rng.normal(size=300000) * np.exp(rng.uniform(-50, 50, 300000))
    b = np.exp(rng.uniform(-200, 200, 300000)) * rng.choice([-1.0, 1.0], 300000)
    assert relerr(run(2, a, b), a / b) < 3e-16
    assert relerr(run(4, b), 1.0 / b) < 3e-16
    x = np.exp(rng.uniform(-600, 600, 300000))
    assert relerr(run(3, x), np.sqrt(x)) < 3e-16


def test_satvap_and_pow():
    t = np.linspace(-60, 60, 100001)
    want = np.where(t > 0, 0.61078 * np.exp(17.27 * t / (t + 237.3)), 0.61078 * np.exp(21.875 * t / (t + 265.5)))
    assert relerr(run(5, t), want) < 2e-15
    rng = np.random.default_rng(4)
    x = rng.uniform(1e-4, 1.0, 200000)
    y = rng.uniform(-12, 1, 200000)
    assert relerr(run(6, x, y), x ** y) < 3e-14


def test_bounded_exp_and_fast_satvap():
    """fexp_b (mcf_device.hpp): n = round(x * 256/ln2) out of the low mantissa bits of x * 256/ln2 + 1.5 * 2^52, for
    |x| < 5e6; the same accuracy as the general form, and exactly 0 / inf once 2^e leaves the double range."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-60, 60, 200000), rng.uniform(-2, 2, 200000),
                        np.array([0.0, -0.0, 1e-300, -1e-300, 709.0, -745.0, 0.5 * np.log(2) / 256, -0.5 * np.log(2) / 256])])
    assert relerr(run(7, x), np.exp(x)) < 4e-16
    assert (run(7, np.array([-800.0, -1e5, -4.9e6])) == 0.0).all() and np.isinf(run(7, np.array([800.0, 4.9e6]))).all()
    assert np.isnan(run(7, np.array([np.nan]))).all()
    # satvap with wave-uniform constants: all-water waves, all-ice waves, waves that straddle 0 degrees C
    for t in (np.linspace(0.01, 150, 64 * 500), np.linspace(-150, 0.0, 64 * 500), rng.uniform(-5, 5, 64 * 500)):
        want = np.where(t > 0, 0.61078 * np.exp(17.27 * t / (t + 237.3)), 0.61078 * np.exp(21.875 * t / (t + 265.5)))
        assert relerr(run(8, t), want) < 1e-14               # |a t / (t + b)| up to 28: the quotient's rounding, amplified
        # round 5: 0.61078 folded into the exponent, one-fma reduction — within the quotient's own rounding of the general form
        q = np.where(t > 0, 17.27 * t / (t + 237.3), 21.875 * t / (t + 265.5))
        assert np.max(np.abs(run(8, t) / run(5, t) - 1.0) / (1.0 + np.abs(q))) < 6e-16


def test_short_exp_and_medium_precision_quotients():
    """Round 5 (mcf_device.hpp): fexp_s reduces with ONE fma against the 53-bit ln2/256 — error |x| 2^-54 on top of the
    polynomial's; frcp_m / fsqrt_m / frcp2_m stop at 46 bits (Penman-Monteith quotients, series conductances)."""
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(-60, 60, 300000), rng.uniform(-4, 4, 300000), np.array([0.0, -0.0, 1e-300, -1e-300])])
    got, want = run(9, x), np.exp(x)
    assert np.max(np.abs(got / want - 1.0) / (4e-16 + 6e-17 * np.abs(x))) < 1.0
    assert np.isnan(run(9, np.array([np.nan]))).all()
    b = np.exp(rng.uniform(-200, 200, 300000)) * rng.choice([-1.0, 1.0], 300000)
    assert relerr(run(10, b), 1.0 / b) < 2.0 ** -45
    v = np.exp(rng.uniform(-600, 600, 300000))
    assert relerr(run(11, v), np.sqrt(v)) < 2.0 ** -45
    a1, b1 = np.exp(rng.uniform(-60, 60, 300000)), np.exp(rng.uniform(-60, 60, 300000))
    assert relerr(run(12, a1, b1), 1.0 / a1 + 1.0 / b1) < 2.0 ** -44
    assert relerr(run(13, a1, b1), 1.0 / a1 + 1.0 / b1) < 5e-16
