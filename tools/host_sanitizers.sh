#!/bin/bash
# CPU-only: the host-side C++ of libmcfhip (point models, flow accumulation, netCDF file layer) built with
# AddressSanitizer + UBSan — without the HIP parts — and driven over the reference-test inputs, random series and
# rasters.  (GPU sanitizers are not available on the pool; this covers the code of the product that runs on the host.)
set -e
cd "$(dirname "$0")/.."
cat > /tmp/mcf_host_stub.cpp <<'CPP'
#include <string>
namespace mcf { int api_fail(int code, const std::string&) { return code; } }
CPP
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared \
    -o /tmp/libmcfhost_asan.so microclimf_amd/csrc/mcf_pointmodel.cpp microclimf_amd/csrc/mcf_hydro.cpp /tmp/mcf_host_stub.cpp
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python - <<'PY'
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from microclimf_amd import _abi, pointmodel as PM, terrain as TR, synthetic
lib = C.CDLL("/tmp/libmcfhost_asan.so")
for name in ("mcf_bigleaf", "mcf_soilm", "mcf_pointmprocess", "mcf_weatherhgt", "mcf_man", "mcf_pointmodelsnow", "mcf_flowacc",
             "mcf_topidx", "mcf_snowenv_from_name"):
    if not hasattr(lib, name) and name != "mcf_snowenv_from_name":
        raise SystemExit(f"{name} missing from the host build")
# route the Python mirrors to the sanitizer build: give the ctypes handle the prototypes _abi sets up
real = _abi.load()
for name in ("mcf_bigleaf", "mcf_soilm", "mcf_pointmprocess", "mcf_weatherhgt", "mcf_man", "mcf_pointmodelsnow", "mcf_flowacc",
             "mcf_topidx", "mcf_canintfrac", "mcf_meltmu", "mcf_meltmu2"):
    f = getattr(lib, name)
    f.restype, f.argtypes = getattr(real, name).restype, getattr(real, name).argtypes
    setattr(real, name, f)                       # later calls through _abi.load() hit the ASan build
from bundled import load
from microclimf_amd import frontend as F
weather, vegp, soilc, dtm = load()
mp = F.runpointmodel(weather, -0.1, dtm, vegp, soilc)                      # BigLeaf, soilm, pointmprocess, man on the year
cold = dict(weather, temp=weather["temp"] - 8.0)
pm = PM.pointmodelsnow(cold["obstime"], cold, [1.0, 0.5, 0.1, 0.2], [5, 180, 50, -5, 2, 0.1, 12], "Taiga", 0.5, 20)
rng = np.random.default_rng(0)
for shape in ((1, 1), (1, 40), (37, 1), (50, 50), (64, 33)):
    z = rng.uniform(0, 100, shape)
    z[rng.random(shape) < 0.05] = np.nan
    TR.flowaccCpp(z)
    if min(shape) >= 1:
        TR.topidx(z, 2.0)
TR.flowaccCpp(np.full((5, 5), np.nan)); TR.topidx(dtm["z"], 1.0)
for days in (1, 2, 6, 95):
    a = synthetic.workload(2, 2, days * 24, reqhgt=0.05)
    c = a["climdata"]
    w = {"temp": c["temp"], "relhum": 100 * c["ea"] / c["es"], "pres": c["pres"], "swdown": c["swdown"], "difrad": c["difrad"],
         "lwdown": c["lwdown"], "windspeed": np.maximum(c["windspeed"], 0.5), "precip": np.zeros(days * 24)}
    PM.BigLeafCpp(a["obstime"], w, F.sortvegp_point(vegp), F.sortsoilc_point(soilc), np.full(days * 24, 0.3), 50.0, -5.0,
                  yearG=days >= 90 or days == 1)
    PM.weatherhgtCpp(a["obstime"], w, 2, 2, 10, 50, -5)
    PM.pointmodelsnow(a["obstime"], w, [2, 0.5, 0.05, 0], [0, 180, 50, -5, 2, 0, 0], "Alpine", 0.5, 10)
    for win in (1, 5, 24, 48, 49, 100):
        try:
            PM.manCpp(c["temp"], win)
        except _abi.McfError:
            assert win > days * 24 or win // 24 > days            # refused: the window is longer than the series
from microclimf_amd import snow as S
for shape in ((1, 1), (3, 50), (50, 50)):                                 # the fast snow method's two host entries
    sv = rng.uniform(0, 1, shape); sv[rng.random(shape) < 0.1] = np.nan
    for n in (0, 1, 24, 700):
        st = rng.normal(0, 3, n)
        S.meltmu(sv, st, st - 1.0)
        st3 = rng.normal(0, 3, shape + (n,))
        S.meltmu2(sv, st3, st3 - 1.0)
    for prec in (0.0, float("nan"), 0.3, 40.0):
        S.canintfrac(np.where(np.isnan(sv), np.nan, 10 * sv), 5 * sv, 2.0, prec, -4.0, 0.0)
print("host-side C++ through ASan + UBSan: clean")
PY
# the netCDF file layer (header-only, with its writer threads): ASan + UBSan, then ThreadSanitizer
unset LD_PRELOAD
cat > /tmp/mcf_nc_harness.cpp <<'CPP'
#include "mcf_ncfile.hpp"
#include <cstdio>
int main() {
    const int64_t rows = 300, cols = 200, T = 48;
    std::vector<double> east(cols), north(rows), hours(T);
    for (int i = 0; i < cols; ++i) east[i] = i + 0.5;
    for (int i = 0; i < rows; ++i) north[i] = i + 0.5;
    for (int i = 0; i < T; ++i) hours[i] = 400000.0 + i;
    std::vector<mcf::NcVarDef> vars = {{"Tz", "Air temperature at height 0.05 m", "deg C x 100"},
                                       {"relhum", "Relative humidity at height 0.05 m", "Percentage"},
                                       {"Rswup", "Upward shortwave radiation", "W/m^2"}};
    mcf::NcFile f;
    std::string e = f.create("/tmp/mcf_nc_harness.nc", rows, cols, T, east.data(), north.data(), hours.data(), "wkt", vars);
    if (!e.empty()) { fprintf(stderr, "%s\n", e.c_str()); return 1; }
    std::vector<uint8_t> recs((size_t)(T * f.rec_bytes), 7);          // 48 records of 720 KB: the threaded write path
    for (int pass = 0; pass < 2; ++pass) {
        e = f.write_records(pass ? 0 : 24, 24, recs.data());
        if (!e.empty()) { fprintf(stderr, "%s\n", e.c_str()); return 1; }
    }
    e = f.write_records(0, T, recs.data());
    if (!e.empty() || !f.write_records(40, 20, recs.data()).size()) { fprintf(stderr, "range check failed\n"); return 1; }
    e = f.close();
    remove("/tmp/mcf_nc_harness.nc");
    return e.empty() ? 0 : 1;
}
CPP
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Imicroclimf_amd/csrc -pthread -o /tmp/mcf_nc_asan /tmp/mcf_nc_harness.cpp
MCF_NC_WRITE_THREADS=4 /tmp/mcf_nc_asan
g++ -O1 -g -std=c++17 -fsanitize=thread -Imicroclimf_amd/csrc -pthread -o /tmp/mcf_nc_tsan /tmp/mcf_nc_harness.cpp
MCF_NC_WRITE_THREADS=4 /tmp/mcf_nc_tsan
echo "netCDF file layer through ASan + UBSan and TSan: clean"
# the netCDF-4 container (mcf_nc4file.hpp: HDF5 bound at run time, chunks deflated by a team of threads): ragged row strips, edge
# chunks, pieces out of order, empty pieces, every deflate level class — ASan + UBSan, then ThreadSanitizer on the deflate team
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Imicroclimf_amd/csrc -pthread -o /tmp/mcf_nc4_asan tools/harness/nc4_harness.cpp -lz -ldl
ASAN_OPTIONS=detect_leaks=0 MCF_NC_DEFLATE_THREADS=4 /tmp/mcf_nc4_asan > /tmp/mcf_nc4_asan.log || { cat /tmp/mcf_nc4_asan.log; exit 1; }
g++ -O1 -g -std=c++17 -fsanitize=thread -Imicroclimf_amd/csrc -pthread -o /tmp/mcf_nc4_tsan tools/harness/nc4_harness.cpp -lz -ldl
MCF_NC_DEFLATE_THREADS=4 /tmp/mcf_nc4_tsan > /tmp/mcf_nc4_tsan.log || { tail -40 /tmp/mcf_nc4_tsan.log; exit 1; }
echo "netCDF-4 container through ASan + UBSan and TSan: clean"
