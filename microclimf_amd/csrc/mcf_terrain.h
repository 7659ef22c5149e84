// mcf_terrain.h — device-level entry of the terrain pre-compute (mcf_terrain.hip).
#pragma once
#include <stdint.h>

namespace mcf {
struct TerrainDev {
    int64_t rows, cols;
    int32_t halo_north, halo_south;
    int64_t row0, rows_total;          // rows_total = 0: the block is the whole raster
    const double* d_dtm;               // device, [(halo_north + rows + halo_south), cols], NaN = NA
    double res, zref;
    int32_t agg;                       // .windsheltera's s (0 -> 10)
    double aspect_na;                  // value of aspect where terra::terrain gives NA: 0 for the solver's
                                       // marshaller (R/internal.R:1136), 180 for the snow driver (R/internal.R:2570)
    double *d_slope, *d_aspect, *d_hor, *d_svfa, *d_wsa;   // device outputs or null
};
// Scratch of terrain_device kept by a caller that runs it repeatedly (the snow plan's 5-day refresh): three buffers grown
// on demand, never shrunk, released by release().  Without one, terrain_device allocates and frees its scratch per call
// (hundreds of MB at 4096 columns: tens of milliseconds of hipMalloc / hipFree).
struct TerrainWork {
    void* p[3] = {nullptr, nullptr, nullptr};
    int64_t cap[3] = {0, 0, 0};
    void release();
};
// All launches on the null stream; returns after the device has finished.
int terrain_device(const TerrainDev& t, TerrainWork* work = nullptr);
}  // namespace mcf
