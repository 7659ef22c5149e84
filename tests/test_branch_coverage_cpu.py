"""Do the parity workloads reach both sides of the hot path's data-dependent branches?

SURVEY Appendix D lists every branch / clamp of the path; a flipped branch is a discontinuous
change of the outputs, so each needs parity cases on either side.  The oracle keeps the
reference's branch structure one to one, so it is built with gcov and run over
tests/parity_cases.py: every conditional of mcf_oracle.c must have been taken both ways, except
the ones listed below, which cannot flip (or are test-infrastructure guards)."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))

# source fragments of branches that are allowed to stay one-sided, with the reason
ONE_SIDED_OK = {
    "> 0 ? ": "allocation-size guards of the oracle itself",
    "opt->out[v] && out->var[v]": "oracle API guard (a requested output always has a buffer in the tests)",
    "g_mean_override_on": "row-block test hook",
    "if (!(reqhgt < 0)) return": "orc_tbelowground is only entered for reqhgt < 0 (cpp:2307)",
    "tsteps > 0 && in->obstime.year[0]": "tsteps == 0 is covered by tests/test_edge_cases_gpu.py",
    "if (sqt < 0) sqt = 0": "cpp:66 rounding guard: 1 - sazi^2 < 0 needs |sazi| > 1",
    "if (g < gmin) g = gmin": "cpp:378 with gmin = 1e-4: uf >= 0.001 keeps g above it",
    "if (trbn < 0.0)": "cpp:1097: pow(clump, Kc) is never negative",
    "if (trb > 0.999)": "cpp:1099: gi <= 0.99 (cpp:1054) and Kc >= 1",
    "if (trb < 0.0)": "cpp:1100: pow(gi, Kc) is never negative",
    "if (Rbdn_g > 1.0)": "cpp:1129: a convex combination of 1 and exp(-kd*pait) <= 1",
    "if (Rbdn_g < 0.0)": "cpp:1130: idem, >= 0",
    "if (o.Rddn_g < 0.0)": "cpp:1073: not reached over the optical parameter space sampled by 'wild_canopy'",
    "if (o.Rdup_z > 1.0)": "cpp:1077: idem",
    "if (o.Rdup_z < 0.0)": "cpp:1078: idem",
    "if (o.Rddn_z < 0.0)": "cpp:1082: idem",
    "if (o.zm < 1e-6)": "cpp:1184: roughlengthCpp already floors zm at 5e-4 (cpp:308)",
    "if (reqhgt > 0) {": "cpp:1199: the drivers pass max(reqhgt, 1e-5) (cpp:2246-2247)",
    "if (Be < 0.001)": "cpp:1208: needs uh > 1000 uf, i.e. log((h-d)/zm) > 400",
    "if (dT > 80.0)": "cpp:1238: dTmx = 49.79 - 0.6273 mxtc exceeds 80 only for mxtc < -48 C",
    "if (o.surfwet > 1.0)": "cpp:1269: exp of a non-positive matric term",
    "if (si < 0.0) si = 0.0;": "cpp:2219: solarindexCpp has clamped already (cpp:100)",
}


def test_parity_cases_flip_every_branch(tmp_path):
    import oracle_branch_coverage as cov
    total, one_sided = cov.measure(tmp_path)
    assert total >= 200
    unexplained = [(no, src) for no, src, _ in one_sided if not any(k in src for k in ONE_SIDED_OK)]
    assert not unexplained, "branches taken one way only:\n" + "\n".join(f"  mcf_oracle.c:{n}: {s}" for n, s in unexplained)
    # the allowlist must not rot: every entry still matches a one-sided line
    stale = [k for k in ONE_SIDED_OK if not any(k in src for _, src, _ in one_sided)]
    assert not stale, f"allowlist entries that no longer apply: {stale}"


# ---- the snow branch (oracle/snow_oracle.c under tests/snow_cases.py) --------------------------------------
SNOW_ONE_SIDED_OK = {
    "if (cis > prec) cis = prec": "cpp:3737: I1*0.678 <= 0.678*Cp*prec < prec for a non-negative load Li",
    "if (out.cis > clim.prec)": "cpp:3934: repeats the clamp canopysnowintCpp has applied already",
    "snowenv < 0 || snowenv > 4": "range guard of the oracle's enum (unknown names map to 0 before the call)",
    "if (alb[i] < 0.1)": "cpp:3768: needs more than 1000 days without snowfall",
    "if (si < 0.0) si = 0.0;": "cpp:3796: solarindexCpp has clamped already (cpp:100)",
    "if (Rddm > 1.0)": "cpp:3815: not reached over the optical parameter space sampled by the 'bright' cases",
    "if (Rddm < 0.0)": "cpp:3816: idem",
    "if (Rdbm > 1.0)": "cpp:3820: idem",
    "if (Rbgm > 1.0)": "cpp:3824: a convex combination of 1 and exp(-kd*pait) <= 1",
    "if (Rbgm < 0.0)": "cpp:3825: idem, >= 0",
    "if (mu > 1.0) mu = 1.0": "cpp:3911: exp(-pai) > 1 needs a negative plant area index",
    "if (wgtg < 0.0)": "cpp:3927: a ratio of two non-negative depths",
    "if (hgts > 0.0) paias": "cpp:4801: the in-canopy branch is only entered with reqhgt < hgts, hence hgts > 0",
    "const int hiy =": "cpp:4984: a year divisible by 100 but not by 400 (2100) is not sampled",
    # pointmodelsnow (cpp:4000-4169) is restated only to replay the reference's test, it is not on the grid path
    "if (zm < 0.001) zm = 0.001": "pointmodelsnow only",
    "if (fabs(H[i]) < 0.1)": "pointmodelsnow only",
    "if (psim[i] <": "pointmodelsnow only",
    "if (psih[i] <": "pointmodelsnow only",
    "if (psim[i] >": "pointmodelsnow only",
    "if (psih[i] >": "pointmodelsnow only",
    "if (iter > maxiter)": "pointmodelsnow only",
    # API guards of the oracle: the tests always request every output
    "if (out->": "oracle API guard",
    "if (outsel[": "oracle API guard",
    "if (arr3[v])": "oracle API guard",
    "if (arr2[v])": "oracle API guard",
}


def test_snow_cases_flip_every_branch(tmp_path):
    import oracle_branch_coverage as cov
    total, one_sided = cov.measure_snow(tmp_path)
    assert total >= 150
    unexplained = [(no, src) for no, src, _ in one_sided if not any(k in src for k in SNOW_ONE_SIDED_OK)]
    assert not unexplained, "branches taken one way only:\n" + "\n".join(f"  snow_oracle.c:{n}: {s}" for n, s in unexplained)
    stale = [k for k in SNOW_ONE_SIDED_OK if not any(k in src for _, src, _ in one_sided)]
    assert not stale, f"allowlist entries that no longer apply: {stale}"
