# mcfhip_overrides.R — route microclimf's grid solver through libmcfhip.
#
# The reference reaches its C++ through two generated stubs (R/RcppExports.R:72-78):
#   runmicro1Cpp <- function(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, long,
#                            Sminp, Smaxp, tfact, complete, mat, out)
#     .Call(`_microclimf_runmicro1Cpp`, ...)
# which .runmodel1Cpp / .runmodel2Cpp call (R/internal.R:1168, 1342).  Replacing those two
# bindings is all it takes: modelin()/runmicro()/runmicro_big()/runbioclim() above stay untouched.
#
# Usage (after building r/mcfhip_glue.so, see INTEGRATION.md):
#   library(microclimf); source("r/mcfhip_overrides.R"); mcfhip_enable()             # one GPU
#   mcfhip_enable(devices = 0:7)                                                     # every GPU of an 8-GPU node

# `devices`: HIP device ordinals (0-based) the grid solver may use from this one R session, e.g. 0:7 on an 8-GPU node — the
# raster is dealt to them in row blocks inside libmcfhip (mcf_runmicro1_multi), results bit for bit those of one device;
# `blocks`: more row blocks than devices (each device solves its blocks one after the other: smaller HBM footprint per block).
mcfhip_enable <- function(glue = "r/mcfhip_glue.so", devices = NULL, blocks = NULL) {
  dyn.load(glue)
  options(mcfhip.devices = if (is.null(devices)) NULL else as.integer(devices),
          mcfhip.blocks = if (is.null(blocks)) NULL else as.integer(blocks))
  rm1 <- function(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, long,
                  Sminp, Smaxp, tfact, complete, mat, out)
    .Call("mcfhip_runmicro1", obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, long,
          Sminp, Smaxp, tfact, complete, mat, out)
  rm2 <- function(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                  Sminp, Smaxp, tfact, complete, mat, out)
    .Call("mcfhip_runmicro2", obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
          Sminp, Smaxp, tfact, complete, mat, out)
  # time-varying vegetation (R/RcppExports.R:80-86, called at R/internal.R:1458 and 1640)
  rm3 <- function(dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                  Sminp, Smaxp, tfact, complete, mat, out)
    .Call("mcfhip_runmicro3", dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
          Sminp, Smaxp, tfact, complete, mat, out)
  rm4 <- function(dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                  Sminp, Smaxp, tfact, complete, mat, out)
    .Call("mcfhip_runmicro4", dfsel, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
          Sminp, Smaxp, tfact, complete, mat, out)
  # fused bioclim sink (R/RcppExports.R runbioclim1Cpp .. runbioclim4Cpp, src/microclimfCpp.cpp:3563-3700)
  for (nm in c("runbioclim1Cpp", "runbioclim2Cpp", "runbioclim3Cpp", "runbioclim4Cpp")) local({
    sym <- paste0("mcfhip_", sub("Cpp$", "", nm))
    utils::assignInNamespace(nm, function(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, long, Sminp,
                                          Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air)
      .Call(sym, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, long, Sminp, Smaxp, tfact, mat, out,
            wetq, dryq, hotq, colq, air), ns = "microclimf")
  })
  utils::assignInNamespace("applycpp3", function(a, fun_name) .Call("mcfhip_applycpp3", a, fun_name),
                           ns = "microclimf")
  # snow branch (R/RcppExports.R:108-114, 124-130; called at R/internal.R:2587 and 3625)
  for (nm in c("gridmodelsnow1", "gridmodelsnow2")) local({
    sym <- paste0("mcfhip_", nm)
    utils::assignInNamespace(nm, function(obstime, climdata, pointm, vegp, other, snowenv)
      .Call(sym, obstime, climdata, pointm, vegp, other, snowenv), ns = "microclimf")
  })
  for (nm in c("gridmicrosnow1", "gridmicrosnow2")) local({
    sym <- paste0("mcfhip_", nm)
    utils::assignInNamespace(nm, function(reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out)
      .Call(sym, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out), ns = "microclimf")
  })
  # output file (R/dataprep.R:1063-1260): same arguments.  terra stays on this side (cell-centre coordinates, projection text);
  # the dataset is written by libmcfhip: format "netcdf4" = the reference's container (deflate 9; through the host's HDF5
  # library, which any host with ncdf4 has), "classic" = uncompressed netCDF classic, needs nothing; options(mcfhip.ncformat).
  wnc <- function(mout, fileout, dtm, reqhgt, vars = NULL) {
    if (class(dtm)[1] == "PackedSpatRaster") dtm <- terra::rast(dtm)
    e <- terra::ext(dtm); r <- terra::res(dtm)
    est <- seq(e$xmin + r[1] / 2, e$xmax - r[1] / 2, r[1])
    nth <- seq(e$ymin + r[2] / 2, e$ymax - r[2] / 2, r[2])
    hours <- as.numeric(as.POSIXct(mout$tme)) / 3600
    if (is.null(vars)) vars <- if (reqhgt > 0) c("Tz", "tleaf", "relhum", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
                               else if (reqhgt == 0) c("Tz", "soilm", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
                               else c("Tz", "soilm")
    fileout <- as.character(fileout); wkt <- as.character(terra::crs(dtm)); vars <- as.character(vars)
    fmt <- getOption("mcfhip.ncformat", "netcdf4")
    invisible(.Call("mcfhip_writetonc", mout, fileout, est, nth, hours, wkt, reqhgt, vars, fmt))
  }
  utils::assignInNamespace("writetonc", wnc, ns = "microclimf")
  utils::assignInNamespace("runmicro1Cpp", rm1, ns = "microclimf")
  utils::assignInNamespace("runmicro2Cpp", rm2, ns = "microclimf")
  utils::assignInNamespace("runmicro3Cpp", rm3, ns = "microclimf")
  utils::assignInNamespace("runmicro4Cpp", rm4, ns = "microclimf")
  invisible(TRUE)
}
