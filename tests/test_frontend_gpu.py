"""BASELINE.json configs[0] — `runmicro()` on the reference's bundled example data (dtmcaerth, vegp with 12 pai layers,
soilc, a year of hourly climdata) — through the host-side front end and the HIP path, against the oracle given the same
prepared solver call.  The terrain inputs come from the device kernels, the point model and the wetness index from
libmcfhip's host code; nothing is synthetic here."""
import numpy as np
import pytest

from bundled import load
from microclimf_amd import frontend as F
from test_parity_gpu import compare

pytestmark = pytest.mark.gpu


def test_bundled_year_through_runmicro(oracle, tmp_path):
    weather, vegp, soilc, dtm = load()
    mp = F.runpointmodel(weather, 0.05, dtm, vegp, soilc)
    a = F.prepare_grid_inputs(mp, 0.05, vegp, soilc, dtm)
    assert a["vegp"]["pai"].shape == (50, 50, 12) and list(a["dfsel"]["lyr"]) == list(range(1, 13))
    assert a["dfsel"]["st"][0] == 0 and a["dfsel"]["ed"][-1] == 8759 and a["complete"]
    got = F.runmicro(mp, 0.05, vegp, soilc, dtm)
    want = oracle.run_grid(**a)
    compare(got, want)
    tz, na = got["Tz"], np.isnan(F.cleanvars(vegp, soilc, dtm["z"])[2])
    assert tz.shape == (50, 50, 8760) and np.array_equal(np.isnan(tz[:, :, 4000]), na)
    # a year on the Lizard peninsula: air 5 cm above the ground between -5 and 65 degC (sunlit, sheltered slopes), warmer than the weather station on summer
    # days, and the netCDF sink takes the result as writetonc would
    assert -5 < np.nanmin(tz) and np.nanmax(tz) < 65
    july_noon = (weather["obstime"]["month"] == 7) & (weather["obstime"]["hour"] == 12)
    assert np.nanmean(tz[:, :, july_noon]) > weather["temp"][july_noon].mean()
    from microclimf_amd import ncsink
    from scipy.io import netcdf_file
    e = dtm["extent"]
    got["tme"] = weather["obstime"]
    ncsink.writetonc(got, tmp_path / "caerth.nc", {"xmin": e[0], "xmax": e[1], "ymin": e[2], "ymax": e[3], "res": dtm["res"]},
                     0.05, vars=("Tz", "relhum"))
    f = netcdf_file(str(tmp_path / "caerth.nc"), "r", mmap=False)
    assert f.variables["Tz"].shape == (8760, 50, 50) and f.variables["east"][0] == e[0] + 0.5
    f.close()


@pytest.mark.parametrize("reqhgt,static", [(1.0, True), (0.0, False), (-0.1, True)])
def test_bundled_month_other_heights(oracle, reqhgt, static):
    weather, vegp, soilc, dtm = load(30 * 24)
    if static:
        vegp = {k: (v[:, :, 6] if v.ndim == 3 else v) for k, v in vegp.items()}          # July's layer, time-invariant
    mp = F.runpointmodel(weather, reqhgt, dtm, vegp, soilc)
    a = F.prepare_grid_inputs(mp, reqhgt, vegp, soilc, dtm)
    got = F.runmicro(mp, reqhgt, vegp, soilc, dtm)
    dfsel = a.get("dfsel")
    assert (dfsel is None) == static
    compare(got, oracle.run_grid(**a))
    assert list(got) == (["Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup"]
                         if reqhgt > 0 else ["Tz", "soilm", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup"]
                         if reqhgt == 0 else ["Tz", "soilm"])
