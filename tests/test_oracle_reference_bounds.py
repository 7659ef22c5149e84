"""The reference's own tests for this arithmetic, replayed against the oracle.

tests/testthat/test-microclimatemodel_wrapper.R and test-BigLeafCpp.R assert interval
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
