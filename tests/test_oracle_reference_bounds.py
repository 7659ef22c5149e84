"""The reference's own tests for this arithmetic, replayed against the oracle.

tests/testthat/test-microclimatemodel_wrapper.R, test-BigLeafCpp.R, test-weatherhgtCpp.R and
test-soilmCpp.R (test-pointmodelsnow.R: tests/test_snow_cpu.py) assert interval
bounds only (the reference ships no golden vectors); every `expect_*` of the two files is
evaluated on the oracle's restatement of the same functions.  This is the only pin the
reference itself provides for the oracle — see DESIGN.md §2."""
import pytest

from oracle import replay_reference_tests as R


def _run(fn):
    checks, info = fn()
    failed = [(label, detail) for label, ok, detail in checks if not ok]
    assert not failed, failed
    return checks, info


def test_microclimatemodel_wrapper_bounds(oracle):
    checks, info = _run(R.replay_wrapper_test)
    assert len(checks) == 22
    # BigLeafCpp(maxiter = 100, yearG = FALSE) converges with err < tol = 0.5 (test-BigLeafCpp.R:113)
    assert info["BL_err"] < 0.5


def test_bigleaf_bounds(oracle):
    checks, info = _run(R.replay_bigleaf_test)
    assert len(checks) == 11 and info["err"] < 0.5


def test_weatherhgt_bounds(oracle):
    """test-weatherhgtCpp.R: BigLeafCpp's diabatic correction moved to 10 m (wind ratio 1.2-1.4, |dT| <= 4, |d ea| <= 0.5)"""
    checks, info = _run(R.replay_weatherhgt_test)
    assert len(checks) == 5
    assert 1.2 <= info["wind_ratio"][0] <= info["wind_ratio"][1] <= 1.4


def test_soilm_bounds(oracle):
    """test-soilmCpp.R: the two-layer bucket model behind pointm$soilm"""
    checks, info = _run(R.replay_soilm_test)
    assert len(checks) == 3 and info["soilm"][0] == 0.419
