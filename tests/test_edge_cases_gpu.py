"""Edge cases through the C ABI: degenerate sizes, empty selections, all-NA rasters,
error reporting."""
import numpy as np
import pytest

from microclimf_amd import McfError, synthetic
from microclimf_amd.api import Plan, runmicro1Cpp, runmicro2Cpp
from test_parity_gpu import compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 1), (1, 50), (37, 1), (5, 3)])
def test_tiny_rasters(oracle, rows, cols):
    """fewer cells than one workgroup tile, single row / single column rasters"""
    a = synthetic.workload(rows, cols, 48, reqhgt=0.05, variety=True, start_doy=170, )
    a["vegp"]["hgt"] = np.where(np.isnan(a["vegp"]["hgt"]), 0.5, a["vegp"]["hgt"])
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))


def test_less_than_one_day_is_all_na(oracle):
    a = synthetic.workload(6, 4, 17, reqhgt=0.05)          # ndays = 0 (cpp:2116)
    got = runmicro1Cpp(**a)
    want = oracle.run_grid(**a)
    for k in want:
        assert got[k].shape == (6, 4, 17) and np.isnan(got[k]).all() and np.isnan(want[k]).all()


def test_zero_timesteps():
    a = synthetic.workload(4, 4, 0, reqhgt=0.05)
    got = runmicro1Cpp(**a)
    assert all(v.shape == (4, 4, 0) for v in got.values()) and len(got) == 10


def test_no_outputs_requested():
    a = synthetic.workload(4, 4, 24, reqhgt=0.05, out=[0] * 10)
    assert runmicro1Cpp(**a) == {}


def test_all_cells_na(oracle):
    a = synthetic.workload(9, 5, 24, reqhgt=0.05)
    a["vegp"]["hgt"][:] = np.nan
    got = runmicro1Cpp(**a)
    assert all(np.isnan(v).all() for v in got.values())
    compare(got, oracle.run_grid(**a))


def test_bare_ground_only(oracle):
    """every cell pai = hgt = 0: the NaN-comparison paths of the reference (SURVEY §7 j)"""
    a = synthetic.workload(12, 7, 48, reqhgt=0.05, start_doy=170)
    for k in ("hgt", "pai", "paia"):
        a["vegp"][k][:] = 0.0
    with np.errstate(invalid="ignore"):
        a["vegp"]["leafden"] = a["vegp"]["pai"] / a["vegp"]["hgt"]      # 0/0 = NaN as in the marshaller
    want = oracle.run_grid(**a)
    assert np.isfinite(want["Tz"]).all()
    compare(runmicro1Cpp(**a), want)


def test_nan_wind_shelter_is_one(oracle):
    a = synthetic.workload(8, 6, 48, reqhgt=0.05, start_doy=170)
    a["soilc"]["wsa"][2, 3, :] = np.nan                                   # cpp:1193
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))


def test_errors_are_reported_not_crashed():
    a = synthetic.workload(4, 4, 24, reqhgt=0.05)
    with pytest.raises(McfError, match="cells_per_block"):
        runmicro1Cpp(**a, cells_per_block=7)
    with pytest.raises(McfError, match="device ordinal"):
        runmicro1Cpp(**a, device=99)
    b = synthetic.workload(4, 4, 24, reqhgt=0.05, array_forcing=True)
    with pytest.raises(ValueError, match="expected shape"):
        runmicro1Cpp(**b)                       # array forcing handed to the vector-forcing entry
    with Plan(**a, ring_days=1, ring_slots=1) as p:
        with pytest.raises(McfError, match="day range"):
            p.run_days(0, 2, 0)
        with pytest.raises(McfError, match="not requested|bad slot"):
            p.fetch(3, "Tz", 0, 24)


def test_array_forcing_plan_needs_uploaded_days():
    b = synthetic.workload(4, 4, 48, reqhgt=0.05, array_forcing=True)
    with Plan(**b, array_forcing=True, ring_days=1, ring_slots=2) as p:
        with pytest.raises(McfError, match="not been uploaded"):
            p.run_days(0, 1, 0)
        p.upload_forcing_days(1, 1, 1)
        p.run_days(1, 1, 1)
        p.sync()


@pytest.mark.parametrize("reqhgt,out", [
    (0.05, [0, 0, 0, 1, 1, 1, 1, 0, 1, 0]),     # pass-1 outputs only: pass 2 is skipped
    (0.0, [1, 0, 0, 1, 0, 0, 0, 0, 0, 0]),      # ground temperature without TVaboveground
    (0.0, [0, 1, 0, 0, 0, 0, 0, 1, 0, 0]),      # tleaf (always NA at reqhgt 0) + Rlwdown
    (-0.1, [0, 0, 0, 1, 0, 0, 0, 0, 0, 0]),     # soil run without Tz: no smoothing pass
    (0.05, [0, 0, 1, 0, 0, 0, 0, 0, 0, 0]),     # relhum alone
])
def test_output_subsets_skip_unneeded_work(oracle, reqhgt, out):
    a = synthetic.workload(14, 6, 72, reqhgt=reqhgt, variety=True, start_doy=170, out=out)
    a["vegp"]["hgt"][1, 1] = np.nan
    compare(runmicro1Cpp(**a), oracle.run_grid(**a))
