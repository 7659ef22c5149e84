"""The seeded parity workloads, in one place: tests/test_parity_gpu.py runs them through
libmcfhip against the oracle, tests/test_branch_coverage_cpu.py runs them through a
gcov-instrumented oracle to prove that together they reach both sides of the data-dependent
branches of the hot path."""
import numpy as np

from microclimf_amd import synthetic


def with_na(a, cells=((0, 0), (3, 2))):
    for (i, j) in cells:
        a["vegp"]["hgt"][i, j] = np.nan
    return a


def _bare(a):
    for k in ("hgt", "pai", "paia"):
        a["vegp"][k][:] = 0.0
    with np.errstate(invalid="ignore"):
        a["vegp"]["leafden"] = a["vegp"]["pai"] / a["vegp"]["hgt"]
    return a


def _nan_ws(a):
    a["soilc"]["wsa"][2, 3, :] = np.nan
    return a


def _calm_humid(a):
    """still, humid, clear nights and very wet / very dry soils: wind floors (cpp:1194-1208),
    Ts < tdew (cpp:1241), soil-moisture clamps (cpp:1025-1026), surfwet > 1 (cpp:1269)"""
    T = len(a["climdata"]["temp"])
    a["climdata"]["windspeed"][::3] = 0.01
    a["pointm"]["umu"][::5] = 0.001
    a["climdata"]["tdew"][::4] = a["climdata"]["temp"][::4] + 3.0
    a["pointm"]["soilm"][: T // 2] = 0.60
    a["pointm"]["soilm"][T // 2:] = 0.01
    a["pointm"]["dtrp"][:] = 0.5
    a["pointm"]["G"][:] *= 40.0                       # G clamp at +-0.6 Rmx (cpp:1290-1291)
    return a


def _special_x(a):
    a["vegp"]["x"][1, 1] = 0.0                        # cankCpp x == 0 (cpp:115)
    a["vegp"]["x"][2, 1] = np.inf                     # isinf(x) (cpp:112)
    a["vegp"]["gsmax"][3, 1] = 1000.0                 # gsmax >= 999.99: no stomatal limit (cpp:1351)
    a["vegp"]["leafr"][4, 1] = np.nan                 # isnan(om) in canopycondCpp (cpp:464)
    return a


def _twi_na(a):
    a["soilc"]["twi"][2, 2] = np.nan                  # NA twi in a valid cell: tadd NA -> NaN outputs (cpp:984-989)
    return a


def _extreme_canopy(a):
    """two-stream clamps (cpp:1054-1082, 1095-1130), roughness clamps (cpp:307), ws floor (cpp:1194)"""
    v, s = a["vegp"], a["soilc"]
    v["clump"][1, :] = 0.9995                         # gi, giu > 0.99; trbn, trb > 0.999
    v["clump"][2, :] = 0.97
    v["leafr"][3, :] = 0.92; v["leaft"][3, :] = 0.05  # very bright leaves
    v["leafr"][4, :] = 0.02; v["leaft"][4, :] = 0.01  # nearly black leaves: albd, albb < 0.01
    s["gref"][5, :] = 0.9                             # bright ground under a thin canopy
    v["pai"][5, :] = 0.05; v["paia"][5, :] = 0.03
    v["pai"][6, :] = 200.0; v["paia"][6, :] = 150.0   # zm > 0.9 (h - d)
    s["wsa"][7, :, :] = 0.01
    with np.errstate(invalid="ignore", divide="ignore"):
        v["leafden"] = v["pai"] / v["hgt"]
    return a


def _wild_canopy(a):
    """canopy / ground optical parameters far outside the usual ranges: the clamps of the two-stream
    fluxes (cpp:1068-1082, 1102-1130) are only reached there"""
    v, so = a["vegp"], a["soilc"]
    shp = v["pai"].shape
    idx = np.arange(v["pai"].size, dtype=np.uint64).reshape(shp)
    U = lambda f, lo, hi: lo + (hi - lo) * synthetic.uniform(f, idx, 77)
    v["leafr"] = U(1, 0.01, 0.98)
    v["leaft"] = (1.0 - v["leafr"]) * U(2, 0.0, 0.98)
    v["clump"] = U(3, 0.0, 0.98) ** 2
    v["x"] = np.exp(U(4, np.log(0.02), np.log(20.0)))
    v["pai"] = np.exp(U(5, np.log(0.02), np.log(12.0)))
    v["paia"] = v["pai"] * U(6, 0.0, 1.0)
    so["gref"] = U(7, 0.01, 0.95)
    so["svfa"] = U(8, 0.05, 1.0)
    v["hgt"] = np.where(np.isnan(v["hgt"]), np.nan, U(9, 0.2, 1.8))
    with np.errstate(invalid="ignore", divide="ignore"):
        v["leafden"] = v["pai"] / v["hgt"]
    return a


def _ground_tiny_veg(a):
    a = with_na(a)
    v = a["vegp"]
    v["hgt"][5, 5] = 0.001; v["pai"][5, 5] = 1.0; v["paia"][5, 5] = 0.9    # uh < uf (cpp:1206)
    v["leafden"][5, 5] = 1000.0
    return a


def _neg_wdir(a):
    a["climdata"]["winddir"][::7] = -100.0            # out of bounds in the reference; wrapped here
    return a


def _hot_dry(a):
    a["climdata"]["swdown"] *= 3.0                    # dT > dTmx / 80 (cpp:1237-1238), Rbeam cap (cpp:1123)
    a["climdata"]["difrad"] *= 0.2
    a["pointm"]["soilm"][:] = 0.08
    a["climdata"]["ea"] *= 0.3
    return a


def _chain_pointm(a):
    """pointm from the restated point-model chain soilmCpp -> BigLeafCpp -> pointmprocess (oracle/pointchain.py)
    instead of the synthetic recipe: the scalings the grid solver applies (umu, kp / muGp, dtrp, G) then carry the
    magnitudes and the day-to-day structure a real run hands over"""
    from microclimf_amd.synthetic import uniform
    from oracle.pointchain import pointm_chain
    c = a["climdata"]
    T = len(c["temp"])
    k = np.arange(T, dtype=np.uint64)
    wet = uniform(300, k // np.uint64(8)) < 0.25
    weather = {"temp": c["temp"], "relhum": 100 * c["ea"] / c["es"], "pres": c["pres"], "swdown": c["swdown"],
               "difrad": c["difrad"], "lwdown": c["lwdown"], "windspeed": c["windspeed"],
               "precip": np.where(wet, 1.5 * uniform(301, k), 0.0)}
    pm, err = pointm_chain(a["obstime"], weather, a["lat"], a["lon"], a["zref"])
    assert err < 5.0                                    # BigLeafCpp converged reasonably on the synthetic weather
    a["pointm"] = pm
    a["climdata"]["windspeed"] = np.maximum(c["windspeed"], 0.5)
    return with_na(a)


# name -> (workload kwargs, array_forcing, mutator)
CASES = {
    "below_canopy": (dict(rows=21, cols=13, tsteps=96, reqhgt=0.05, zref=2.0, hgt_range=(0.05, 1.5),
                          variety=True, start_doy=170), False, with_na),
    "mixed_canopy": (dict(rows=21, cols=13, tsteps=96, reqhgt=1.0, zref=2.0, hgt_range=(0.05, 1.9),
                          variety=True, start_doy=170), False, with_na),
    "above_canopy": (dict(rows=21, cols=13, tsteps=96, reqhgt=5.0, zref=10.0, hgt_range=(0.5, 9.0),
                          variety=True, start_doy=170), False, with_na),
    "ground": (dict(rows=21, cols=13, tsteps=96, reqhgt=0.0, zref=2.0, variety=True, start_doy=170), False,
               _ground_tiny_veg),
    "cold_winter": (dict(rows=17, cols=9, tsteps=72, reqhgt=0.05, variety=True, start_doy=1, cold=12.0), False, with_na),
    "cold_dec": (dict(rows=17, cols=9, tsteps=72, reqhgt=0.05, variety=True, start_doy=355, cold=4.0), False, with_na),
    "spring": (dict(rows=17, cols=9, tsteps=72, reqhgt=0.05, variety=True, start_doy=80), False, with_na),
    "tropical": (dict(rows=16, cols=8, tsteps=48, reqhgt=0.05, zref=12.0, hgt_range=(0.2, 11.0), variety=True,
                      start_doy=80, lat=0.5, lon=0.0), False, None),
    "boreal_tall": (dict(rows=12, cols=8, tsteps=48, reqhgt=0.5, zref=25.0, hgt_range=(0.5, 22.0), variety=True,
                         start_doy=172, lat=62.0, lon=10.0), False, None),
    "bare_only": (dict(rows=12, cols=7, tsteps=48, reqhgt=0.05, start_doy=170), False, _bare),
    "nan_shelter": (dict(rows=8, cols=6, tsteps=48, reqhgt=0.05, start_doy=170), False, _nan_ws),
    "calm_humid": (dict(rows=10, cols=6, tsteps=72, reqhgt=0.05, variety=True, start_doy=200), False, _calm_humid),
    "special_x": (dict(rows=10, cols=6, tsteps=48, reqhgt=0.05, variety=True, start_doy=170), False, _special_x),
    "hot_dry": (dict(rows=10, cols=6, tsteps=48, reqhgt=0.05, variety=True, start_doy=180), False, _hot_dry),
    "just_above_short_veg": (dict(rows=10, cols=6, tsteps=48, reqhgt=0.06, hgt_range=(0.05, 0.06), variety=True,
                                  start_doy=170), False, None),
    "twi_na": (dict(rows=8, cols=6, tsteps=48, reqhgt=0.05, start_doy=170), False, _twi_na),
    "extreme_canopy": (dict(rows=9, cols=6, tsteps=48, reqhgt=0.05, start_doy=172, na_frac=0.0), False, _extreme_canopy),
    "wild_canopy": (dict(rows=40, cols=25, tsteps=48, reqhgt=0.05, start_doy=172, na_frac=0.0), False, _wild_canopy),
    "neg_wdir": (dict(rows=6, cols=5, tsteps=48, reqhgt=0.05, start_doy=170), False, _neg_wdir),
    "no_tz": (dict(rows=8, cols=6, tsteps=48, reqhgt=0.05, start_doy=170, out=[0, 1, 0, 1, 1, 1, 0, 0, 1, 1]), False, with_na),
    "soil_no_tz": (dict(rows=8, cols=6, tsteps=48, reqhgt=-0.1, start_doy=170, out=[0, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_1cm": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.01, variety=True, start_doy=100,
                      out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_1cm_incomplete": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.01, variety=True, start_doy=100,
                                 out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "soil_skin_incomplete": (dict(rows=8, cols=5, tsteps=96, reqhgt=-0.0005, variety=True, start_doy=100,
                                  out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "ground_no_tz": (dict(rows=8, cols=6, tsteps=48, reqhgt=0.0, start_doy=170, out=[0, 0, 0, 1, 0, 1, 1, 1, 1, 1]), False, with_na),
    "soil_leap_year_incomplete": (dict(rows=8, cols=5, tsteps=96, reqhgt=-0.4, variety=True, start_doy=60, year=2024,
                                       out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "soil_5cm": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.05, variety=True, start_doy=100,
                      out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_40cm": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.4, variety=True, start_doy=100,
                       out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_3m": (dict(rows=18, cols=7, tsteps=240, reqhgt=-3.0, variety=True, start_doy=100,
                     out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_40m": (dict(rows=18, cols=7, tsteps=240, reqhgt=-40.0, variety=True, start_doy=100,
                      out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), False, with_na),
    "soil_5cm_incomplete": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.05, variety=True, start_doy=100,
                                 out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "soil_40cm_incomplete": (dict(rows=18, cols=7, tsteps=240, reqhgt=-0.4, variety=True, start_doy=100,
                                  out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "soil_3m_incomplete": (dict(rows=18, cols=7, tsteps=240, reqhgt=-3.0, variety=True, start_doy=100,
                                out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "soil_40m_incomplete": (dict(rows=18, cols=7, tsteps=240, reqhgt=-40.0, variety=True, start_doy=100,
                                 out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), False, with_na),
    "partial_day_mask": (dict(rows=16, cols=5, tsteps=60, reqhgt=0.05, variety=True, start_doy=200,
                              out=[1, 0, 1, 0, 1, 0, 0, 1, 0, 0]), False, with_na),
    "chain_pointm": (dict(rows=15, cols=8, tsteps=240, reqhgt=0.05, variety=True, start_doy=140), False, _chain_pointm),
    "chain_pointm_winter": (dict(rows=9, cols=8, tsteps=120, reqhgt=1.0, hgt_range=(0.05, 1.9), variety=True,
                                 start_doy=20, cold=6.0), False, _chain_pointm),
    "array_below": (dict(rows=19, cols=6, tsteps=72, reqhgt=0.05, variety=True, start_doy=170, array_forcing=True),
                    True, with_na),
    "array_ground": (dict(rows=19, cols=6, tsteps=72, reqhgt=0.0, variety=True, start_doy=170, array_forcing=True),
                     True, with_na),
    "array_soil": (dict(rows=19, cols=6, tsteps=72, reqhgt=-0.2, variety=True, start_doy=170, array_forcing=True,
                        out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0]), True, with_na),
    "array_soil_incomplete": (dict(rows=9, cols=6, tsteps=72, reqhgt=-0.2, variety=True, start_doy=170,
                                   array_forcing=True, out=[1, 0, 0, 1, 0, 0, 0, 0, 0, 0], complete=False), True, with_na),
    "array_winter_low_sun": (dict(rows=12, cols=6, tsteps=48, reqhgt=0.05, variety=True, start_doy=355,
                                  array_forcing=True, cold=6.0), True, with_na),
}


def build(name):
    kw, af, mut = CASES[name]
    a = synthetic.workload(**kw)
    if mut is not None:
        a = mut(a)
    return a, af
