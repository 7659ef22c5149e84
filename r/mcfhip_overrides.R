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
# `snow_resident = TRUE`: runsnowmodel() + runmicro(snow = TRUE) for data.frame weather keep the year's snow series on the
# device (see mcfhip_snow_resident below); FALSE (default): the reference's own R drivers over the replaced bindings.
# `keep_gb` (with snow_resident): pass 1's snow chunks stay in device memory for pass 2, up to this many GB (mcf_snowrun_keep);
# pays when the snow series are wanted anyway or a handle runs more than one period — 0 (default): re-run from checkpoints.
mcfhip_enable <- function(glue = "r/mcfhip_glue.so", devices = NULL, blocks = NULL, snow_resident = FALSE, keep_gb = 0) {
  dyn.load(glue)
  if (snow_resident) mcfhip_snow_resident()
  options(mcfhip.devices = if (is.null(devices)) NULL else as.integer(devices),
          mcfhip.blocks = if (is.null(blocks)) NULL else as.integer(blocks),
          mcfhip.keep_gb = as.numeric(keep_gb))
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
  # the dataset is written by libmcfhip: "classic" (default) = uncompressed netCDF classic, needs nothing and every netCDF
  # reader opens it; options(mcfhip.ncformat = "netcdf4") = the reference's container (deflate 9) written through the HDF5
  # library the session has mapped already (ncdf4's / terra's own copy is used when there is one).  The netCDF-4 files have
  # been read back with HDF5's own tools only — not yet by libnetcdf / ncdf4 — which is why they are not the default.
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
    fmt <- getOption("mcfhip.ncformat", "classic")
    invisible(.Call("mcfhip_writetonc", mout, fileout, est, nth, hours, wkt, reqhgt, vars, fmt))
  }
  utils::assignInNamespace("writetonc", wnc, ns = "microclimf")
  utils::assignInNamespace("runmicro1Cpp", rm1, ns = "microclimf")
  utils::assignInNamespace("runmicro2Cpp", rm2, ns = "microclimf")
  utils::assignInNamespace("runmicro3Cpp", rm3, ns = "microclimf")
  utils::assignInNamespace("runmicro4Cpp", rm4, ns = "microclimf")
  invisible(TRUE)
}


# ---- runmicro(snow = TRUE) with the snow series kept on the device (include/mcf.h mcf_snowrun_*) --------------------------
# The reference hands five [rows, cols, hours] arrays from runsnowmodel() to runmicro(snow = TRUE, snowmod = ) through R.
# With these two bindings the chunk loop of `.snowmodel1` (R/internal.R:2563-2617) and `.runmicrosnow1`'s two models and merge
# (R/internal.R:3581-3659) run inside libmcfhip instead, chunk by chunk on the device, and only the merged microclimate
# comes back.  No line of the reference's R preparation is restated here: its own functions run up to the point where they
# would enter compiled code, and a recording stand-in for that entry point takes what they prepared.
#   .snowmodel1    runs as written up to its first gridmodelsnow1 call (weather height adjustment, pointmodelsnow, `.sortl`,
#                  initial depths and ages), then returns a light "mcfhip_smod": the loop's arguments + umu, no arrays
#   .runmicrosnow1 given a "mcfhip_smod": the solver's fifteen arguments for the WHOLE series through `.runmicronosnow`'s own
#                  marshalling (recorded at runmicro1Cpp), the run's first pass -> snow days / no-snow days, gridmicrosnow1's
#                  inputs through `.prepsnowinputs1` (every day handed over, the true snow-day steps for `.sortl2`), second pass.
#                  Anything else (arrays from the reference's runsnowmodel, reqhgt < 0) goes to the reference's function.
# NOT run in the build image (no R there): tests/test_r_glue_syntax_cpu.py checks the .Call names and arities only.
mcfhip_snow_resident <- function() {
  ns <- asNamespace("microclimf")
  captured <- function() structure(class = c("mcfhip_captured", "error", "condition"), list(message = "mcfhip: captured", call = NULL))
  with_binding <- function(name, fun, expr) {
    old <- get(name, envir = ns)
    utils::assignInNamespace(name, fun, ns = "microclimf")
    on.exit(utils::assignInNamespace(name, old, ns = "microclimf"))
    tryCatch(expr, mcfhip_captured = function(e) NULL)
  }
  sm1_ref <- get(".snowmodel1", envir = ns)
  rms1_ref <- get(".runmicrosnow1", envir = ns)
  sm1 <- function(weather, dtm, vegp, soilc, snowenv = "Taiga", snowinitd = 0, snowinita = 0, zref = 2, windhgt = zref, tfact = 0.02) {
    cap <- new.env()
    pms <- get("pointmodelsnow", envir = ns)
    with_binding("pointmodelsnow", function(obstime, weather, vegp, other, snowenv) {
      r <- pms(obstime, weather, vegp, other, snowenv)
      cap$obstime <- obstime; cap$weather <- weather; cap$pmod <- r
      r
    }, with_binding("gridmodelsnow1", function(obstime, climdata, pointm, vegp, other, snowenv) {
      cap$vegp <- vegp; cap$other <- other; cap$snowenv <- snowenv
      stop(captured())
    }, sm1_ref(weather, dtm, vegp, soilc, snowenv, snowinitd, snowinita, zref, windhgt, tfact)))
    pm <- cap$pmod
    pointm <- data.frame(Gp = pm$G, Tc = pm$Tc, RswabsG = pm$RswabsG, RlwabsG = pm$RlwabsG, umu = pm$umu)
    if (class(dtm)[1] == "PackedSpatRaster") dtm <- terra::rast(dtm)
    structure(list(args = list(cap$obstime, cap$weather, pointm, cap$vegp, cap$other, cap$snowenv,
                               get(".is", envir = ns)(dtm), terra::res(dtm)[1], tfact), umu = pm$umu),
              class = "mcfhip_smod")
  }
  rms1 <- function(micropoint, reqhgt, vegp, soilc, dtm, smod, runchecks = TRUE, pai_a = NA, tfact = 1.5,
                   out = rep(TRUE, 10), slr = NA, apr = NA, hor = NA, twi = NA, wsa = NA, svf = NA) {
    if (!inherits(smod, "mcfhip_smod") || reqhgt < 0)
      return(rms1_ref(micropoint, reqhgt, vegp, soilc, dtm, smod, runchecks, pai_a, tfact, out, slr, apr, hor, twi, wsa, svf))
    if (class(dtm)[1] == "PackedSpatRaster") dtm <- terra::rast(dtm)
    grid <- NULL
    with_binding("runmicro1Cpp", function(...) { grid <<- list(...); stop(captured()) },
                 get(".runmicronosnow", envir = ns)(micropoint, reqhgt, vegp, soilc, dtm, dtmc = NA, altcorrect = 0, runchecks,
                                                    pai_a, tfact, out, slr, apr, hor, twi, wsa, svf))
    h <- .Call("mcfhip_snowrun_create", grid, smod$args)
    days <- .Call("mcfhip_snowrun_pass1", h)
    mi <- list(NULL, NULL, NULL, NULL)
    if (length(days$snowdays) > 0) {
      mps <- microclimf::subsetpointmodel(micropoint, days = days$snowdays)
      mpa <- micropoint
      mpa$subs <- mps$subs
      alld <- seq_len(length(smod$umu) %/% 24)
      blank <- list(Tz = array(NA_real_, dim = c(dim(dtm)[1:2], 0)))
      pin <- get(".prepsnowinputs1", envir = ns)(reqhgt, dtm, vegp, soilc, mpa, alld, integer(0), list(umu = smod$umu), blank,
                                                 micropoint$tmeorig, mps$subs, runchecks, slr, apr, hor, svf, wsa, pai_a)
      mi <- list(pin$obstime, pin$weather, pin$vegp, pin$other)
    }
    .Call("mcfhip_snowrun_pass2", h, mi[[1]], mi[[2]], mi[[3]], mi[[4]], micropoint$matemp)
  }
  utils::assignInNamespace(".snowmodel1", sm1, ns = "microclimf")
  utils::assignInNamespace(".runmicrosnow1", rms1, ns = "microclimf")
  invisible(TRUE)
}
