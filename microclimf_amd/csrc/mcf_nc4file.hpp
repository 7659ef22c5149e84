// mcf_nc4file.hpp — the `writetonc` sink's netCDF-4 container (reference R/dataprep.R:1063-1260: ncvar_def(..., prec =
// "integer", missval = -9999, compression = 9) through ncdf4 -> libnetcdf -> HDF5).
//
// A netCDF-4 file IS an HDF5 file laid out by the conventions of the NetCDF-4 file-format specification: every dimension a
// dimension-scale dataset (CLASS / NAME / _Netcdf4Dimid, here all three with their coordinate values), every variable a
// chunked dataset with the scales attached (DIMENSION_LIST / REFERENCE_LIST), `_FillValue` as an attribute of the variable's
// type, text attributes as fixed-length NULLTERM strings on a scalar space, link and attribute creation order tracked.
// The reference reaches HDF5 through libnetcdf; this image has no libnetcdf but it does have the HDF5 library itself
// (1.10, thread-safe, with the high-level dimension-scale calls), so the file is written through HDF5's own C API, bound at
// run time (dlopen: libmcfhip.so has no build-time dependency on it, and a host without HDF5 gets an error that says so
// from mcf_nc_create — the classic container of mcf_ncfile.hpp needs nothing).
//
// What is NOT left to the library is the compression: deflate 9 runs at ~20 MB/s per core, so handed to H5Dwrite a solved
// day of a 1024^2 raster (1 GB of int32) would take a minute.  A record piece is cut into chunks, the chunks are deflated by
// a team of host threads with zlib, and each finished chunk goes into the file by H5Dwrite_chunk (the raw, already filtered
// bytes; the calls themselves are serialised).  The chunk bytes are the device's: k_pack_nc already produces big-endian
// int32 in [time][north][east] order for the classic container, the variables are therefore declared H5T_STD_I32BE (netCDF's
// NC_ENDIAN_BIG) and the same device kernel and host packing serve both containers.
//
// Chunk shape: [1 time step][strip of rows][all columns], a strip of at most 1 Mi cells (4 MB raw) — a day's records
// arrive whole and in time order, and a reader of a map (one step) touches exactly the chunks it needs.  Unwritten chunks
// read as the fill value (-9999).
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "mcf_ncfile.hpp"

namespace mcf {

// ---- the part of HDF5's C API used here, bound by name (HDF5 >= 1.10: hid_t is 64 bits wide) ------------------------------
struct H5Api {
    typedef int64_t hid_t;
    typedef int herr_t;
    typedef unsigned long long hsize_t;

    herr_t (*open)();
    herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*);
    herr_t (*Eset_auto2)(hid_t, void*, void*);
    herr_t (*Eget_auto2)(hid_t, void**, void**);
    hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t);
    herr_t (*Fclose)(hid_t);
    hid_t (*Pcreate)(hid_t);
    herr_t (*Pclose)(hid_t);
    herr_t (*Pset_link_creation_order)(hid_t, unsigned);
    herr_t (*Pset_attr_creation_order)(hid_t, unsigned);
    herr_t (*Pset_chunk)(hid_t, int, const hsize_t*);
    herr_t (*Pset_deflate)(hid_t, unsigned);
    herr_t (*Pset_fill_value)(hid_t, hid_t, const void*);
    hid_t (*Screate)(int);
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*);
    herr_t (*Sclose)(hid_t);
    hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
    herr_t (*Dwrite_chunk)(hid_t, hid_t, uint32_t, const hsize_t*, size_t, const void*);
    herr_t (*Dclose)(hid_t);
    hid_t (*Acreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t);
    herr_t (*Awrite)(hid_t, hid_t, const void*);
    herr_t (*Aclose)(hid_t);
    hid_t (*Tcopy)(hid_t);
    herr_t (*Tset_size)(hid_t, size_t);
    herr_t (*Tset_strpad)(hid_t, int);
    herr_t (*Tclose)(hid_t);
    herr_t (*DSset_scale)(hid_t, const char*);
    herr_t (*DSattach_scale)(hid_t, hid_t, unsigned);
    hid_t T_STD_I32BE, T_IEEE_F64LE, T_NATIVE_DOUBLE, T_NATIVE_INT, T_C_S1, P_FILE_CREATE, P_DATASET_CREATE;
    unsigned ver[3];

    // nullptr + the reason when HDF5 cannot be bound
    static const H5Api* get(std::string& err) {
        static std::mutex mu;
        static H5Api api;
        static int state = 0;          // 0 untried, 1 bound, -1 failed
        static std::string why;
        std::lock_guard<std::mutex> lk(mu);
        if (state == 0) state = api.bind(why) ? 1 : -1;
        if (state < 0) { err = why; return nullptr; }
        return &api;
    }

private:
    // A host process (an R session with ncdf4 / terra, Python with netCDF4 / h5py) usually has ONE HDF5 mapped already, and
    // libnetcdf resolves its H5* symbols against it.  So: first the library that is already there (RTLD_NOLOAD on the versioned
    // sonames — nothing new enters the process); only then a fresh one, and then with RTLD_LOCAL, so that a second HDF5 of
    // another soname can never capture another library's symbol look-ups.
    static void* open_first(const std::vector<std::string>& names, std::string& tried) {
        for (const std::string& n : names)
            if (!n.empty())
                if (void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL)) return h;
        for (const std::string& n : names) {
            if (n.empty()) continue;
            if (void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL)) return h;
            tried += (tried.empty() ? "" : ", ") + n;
        }
        return nullptr;
    }
    bool bind(std::string& why) {
        const char* e1 = getenv("MCF_HDF5_LIB");
        const char* e2 = getenv("MCF_HDF5_HL_LIB");
        std::string tried;
        // MCF_HDF5_LIB / MCF_HDF5_HL_LIB name THE library to use; without them the usual names are tried in turn
        typedef std::vector<std::string> Names;
        void* h = open_first(e1 ? Names{e1} : Names{"libhdf5.so.310", "libhdf5.so.200", "libhdf5.so.103", "libhdf5.so.101", "libhdf5.so.100",
                                                    "libhdf5_serial.so.310", "libhdf5_serial.so.200", "libhdf5_serial.so.103",
                                                    "libhdf5_serial.so.100", "libhdf5.so", "libhdf5_serial.so",
                                                    "/opt/conda/lib/libhdf5.so"}, tried);
        if (!h) { why = "the netCDF-4 container needs the HDF5 library and none could be loaded (tried " + tried + "; MCF_HDF5_LIB names one)"; return false; }
        // (libhdf5_hl names the core library it was linked against in its own DT_NEEDED; with the core mapped above the loader
        // resolves it to that copy)
        void* hl = open_first(e2 ? Names{e2} : Names{"libhdf5_hl.so.310", "libhdf5_hl.so.200", "libhdf5_hl.so.100", "libhdf5_serial_hl.so.310",
                                                     "libhdf5_serial_hl.so.200", "libhdf5_serial_hl.so.100", "libhdf5_hl.so",
                                                     "libhdf5_serial_hl.so", "/opt/conda/lib/libhdf5_hl.so"}, tried);
        if (!hl) { why = "HDF5's high-level library (dimension scales) could not be loaded (tried " + tried + "; MCF_HDF5_HL_LIB names one)"; return false; }
        bool ok = true;
        std::string missing;
        auto sym = [&](void* lib, const char* name, auto& fp) {
            void* p = dlsym(lib, name);
            if (!p) { ok = false; missing += std::string(" ") + name; }
            fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(p);
        };
        sym(h, "H5open", open); sym(h, "H5get_libversion", get_libversion); sym(h, "H5Eset_auto2", Eset_auto2);
        sym(h, "H5Eget_auto2", Eget_auto2);
        sym(h, "H5Fcreate", Fcreate); sym(h, "H5Fclose", Fclose); sym(h, "H5Pcreate", Pcreate); sym(h, "H5Pclose", Pclose);
        sym(h, "H5Pset_link_creation_order", Pset_link_creation_order); sym(h, "H5Pset_attr_creation_order", Pset_attr_creation_order);
        sym(h, "H5Pset_chunk", Pset_chunk); sym(h, "H5Pset_deflate", Pset_deflate); sym(h, "H5Pset_fill_value", Pset_fill_value);
        sym(h, "H5Screate", Screate); sym(h, "H5Screate_simple", Screate_simple); sym(h, "H5Sclose", Sclose);
        sym(h, "H5Dcreate2", Dcreate2); sym(h, "H5Dwrite", Dwrite); sym(h, "H5Dwrite_chunk", Dwrite_chunk); sym(h, "H5Dclose", Dclose);
        sym(h, "H5Acreate2", Acreate2); sym(h, "H5Awrite", Awrite); sym(h, "H5Aclose", Aclose);
        sym(h, "H5Tcopy", Tcopy); sym(h, "H5Tset_size", Tset_size); sym(h, "H5Tset_strpad", Tset_strpad); sym(h, "H5Tclose", Tclose);
        sym(hl, "H5DSset_scale", DSset_scale); sym(hl, "H5DSattach_scale", DSattach_scale);
        if (!ok) { why = "this HDF5 library lacks" + missing + " (1.10.3 or newer is needed)"; return false; }
        if (open() < 0 || get_libversion(&ver[0], &ver[1], &ver[2]) < 0) { why = "H5open failed"; return false; }
        if (ver[0] == 1 && ver[1] < 10) { why = "HDF5 1.10 or newer is needed"; return false; }
        // the library's predefined identifiers are variables it fills in H5open
        auto id = [&](const char* name, hid_t& dst) {
            void* p = dlsym(h, name);
            if (!p) { ok = false; missing += std::string(" ") + name; return; }
            dst = *reinterpret_cast<hid_t*>(p);
        };
        id("H5T_STD_I32BE_g", T_STD_I32BE); id("H5T_IEEE_F64LE_g", T_IEEE_F64LE); id("H5T_NATIVE_DOUBLE_g", T_NATIVE_DOUBLE);
        id("H5T_NATIVE_INT_g", T_NATIVE_INT); id("H5T_C_S1_g", T_C_S1);
        id("H5P_CLS_FILE_CREATE_ID_g", P_FILE_CREATE); id("H5P_CLS_DATASET_CREATE_ID_g", P_DATASET_CREATE);
        if (!ok) { why = "this HDF5 library lacks" + missing; return false; }
        return true;
    }

public:
    // HDF5 builds without thread safety are common: every call of this translation unit into the library is made under ONE
    // process-wide lock (two files written from two host threads must not meet inside it).
    static std::mutex& lock() { static std::mutex m; return m; }
    // The library's automatic error printing is the HOST's setting (an R session's ncdf4 relies on it): it is switched off
    // only while one of our calls runs — errors come back as return codes — and put back as it was.
    struct Quiet {
        const H5Api* a;
        void* fn = nullptr;
        void* data = nullptr;
        explicit Quiet(const H5Api* api) : a(api) { if (a->Eget_auto2(0, &fn, &data) >= 0) a->Eset_auto2(0, nullptr, nullptr); else fn = nullptr; }
        ~Quiet() { if (fn) a->Eset_auto2(0, fn, data); }
    };
};

class Nc4File : public NcFile {
public:
    typedef H5Api::hid_t hid_t;
    typedef H5Api::hsize_t hsize_t;
    static constexpr int64_t kChunkCells = (int64_t)1 << 20;
    static constexpr int kDeflateThreads = 16;    // MCF_NC_DEFLATE_THREADS overrides

    ~Nc4File() override { close(); }

    // level: 1..9 deflate, 0 none (chunked all the same)
    std::string create4(const char* path, int64_t rows_, int64_t cols_, int64_t nsteps_, const double* east, const double* north,
                        const double* time_hours_, const char* crs_wkt, const std::vector<NcVarDef>& vars, int level) {
        std::string err;
        h5 = H5Api::get(err);
        if (!h5) return err;
        std::lock_guard<std::mutex> lk(H5Api::lock());
        H5Api::Quiet quiet(h5);
        rows = rows_; cols = cols_; nsteps = nsteps_; nvars = (int)vars.size();
        rec_bytes = 8 + (int64_t)nvars * rows * cols * 4;      // a record piece arrives in the classic container's shape
        rec_begin = 0;
        level_ = level;
        strip_rows = std::max<int64_t>(1, std::min(rows, kChunkCells / std::max<int64_t>(cols, 1)));
        if (strip_rows * cols * 4 >= ((int64_t)1 << 32)) return "one raster row must stay below 4 GiB (HDF5's chunk limit)";
        nstrips = (rows + strip_rows - 1) / strip_rows;
        (void)::unlink(path);
        const unsigned order = 0x0001u | 0x0002u;                 // H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED
        const hid_t fcpl = h5->Pcreate(h5->P_FILE_CREATE);
        if (fcpl < 0) return "H5Pcreate failed";
        h5->Pset_link_creation_order(fcpl, order);
        h5->Pset_attr_creation_order(fcpl, order);
        file_ = h5->Fcreate(path, 0x0002u /* H5F_ACC_TRUNC */, fcpl, 0);
        h5->Pclose(fcpl);
        if (file_ < 0) return std::string("cannot create ") + path + " (H5Fcreate)";
        bool ok = true;
        // (no `_NCProperties`: the attribute is libnetcdf's own provenance stamp and optional — files of netCDF < 4.4.1 have
        // none — and a reader must not be told that libnetcdf wrote this file)
        // dimensions = coordinate variables, in ncdf4's order of definition (dimids 0, 1, 2)
        hid_t d_east = -1, d_north = -1, d_time = -1;
        ok = ok && coord(d_east, "east", cols, east, 0) && att_text(d_east, "units", "metres") && att_text(d_east, "long_name", "Eastings");
        ok = ok && coord(d_north, "north", rows, north, 1) && att_text(d_north, "units", "metres") && att_text(d_north, "long_name", "Northings");
        ok = ok && coord(d_time, "time", nsteps, time_hours_, 2) && att_text(d_time, "units", "hours since 1970-01-01 00:00") &&
             att_text(d_time, "standard_name", "time") && att_text(d_time, "calendar", "gregorian");
        // the data variables
        for (int k = 0; ok && k < nvars; ++k) {
            const hid_t dcpl = h5->Pcreate(h5->P_DATASET_CREATE);
            const hsize_t dims[3] = {(hsize_t)nsteps, (hsize_t)rows, (hsize_t)cols};
            const hsize_t chunk[3] = {1, (hsize_t)strip_rows, (hsize_t)cols};
            const int fill = kMissval;
            ok = dcpl >= 0 && h5->Pset_attr_creation_order(dcpl, order) >= 0 && h5->Pset_chunk(dcpl, 3, chunk) >= 0 &&
                 h5->Pset_fill_value(dcpl, h5->T_NATIVE_INT, &fill) >= 0 && (level_ <= 0 || h5->Pset_deflate(dcpl, (unsigned)level_) >= 0);
            const hid_t sp = ok ? h5->Screate_simple(3, dims, nullptr) : -1;
            const hid_t ds = sp >= 0 ? h5->Dcreate2(file_, vars[k].name.c_str(), h5->T_STD_I32BE, sp, 0, dcpl, 0) : -1;
            if (sp >= 0) h5->Sclose(sp);
            if (dcpl >= 0) h5->Pclose(dcpl);
            ok = ok && ds >= 0;
            if (ds >= 0) dsets_.push_back(ds);
            ok = ok && att_text(ds, "units", vars[k].units.c_str()) && att_fill(ds) && att_text(ds, "long_name", vars[k].long_name.c_str()) &&
                 att_text(ds, "grid_mapping", "crs");
            ok = ok && h5->DSattach_scale(ds, d_time, 0) >= 0 && h5->DSattach_scale(ds, d_north, 1) >= 0 && h5->DSattach_scale(ds, d_east, 2) >= 0;
        }
        // `crs`: a scalar int holding 1 with the projection as text (add_crs_info, dataprep.R:1078-1092); last, as in writetonc's list
        if (ok) {
            const hid_t dcpl = h5->Pcreate(h5->P_DATASET_CREATE);
            h5->Pset_attr_creation_order(dcpl, order);
            const hid_t sp = h5->Screate(0 /* H5S_SCALAR */);
            const hid_t ds = h5->Dcreate2(file_, "crs", h5->T_NATIVE_INT, sp, 0, dcpl, 0);
            const int one = 1;
            ok = ds >= 0 && h5->Dwrite(ds, h5->T_NATIVE_INT, 0, 0, 0, &one) >= 0 && att_text(ds, "crs_wkt", crs_wkt ? crs_wkt : "") &&
                 att_text(ds, "grid_mapping_name", "longitude_latitude");
            if (ds >= 0) h5->Dclose(ds);
            h5->Sclose(sp);
            h5->Pclose(dcpl);
        }
        for (hid_t d : {d_east, d_north, d_time}) if (d >= 0) h5->Dclose(d);
        if (!ok) { close_locked(); return "HDF5 refused a call while the dataset was being defined"; }
        return "";
    }

    // records [step0, step0 + n) in the classic container's record shape (8 time bytes, then per variable [rows][cols] big-endian int32)
    std::string write_records(int64_t step0, int64_t n, uint8_t* recs) override {
        if (file_ < 0) return "file is closed";
        if (step0 < 0 || n < 0 || step0 + n > nsteps) return "record range outside the file";
        const int64_t jobs = n * nvars * nstrips;
        static const int max_threads = [] { const char* e = getenv("MCF_NC_DEFLATE_THREADS"); int t = e ? atoi(e) : 0; return t > 0 ? t : kDeflateThreads; }();
        const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)max_threads, (int64_t)hw, jobs}));
        std::atomic<int64_t> next{0};
        std::vector<std::string> errs(nt);
        auto work = [&](int t) {
            const size_t raw_cap = (size_t)(strip_rows * cols * 4);
            std::vector<uint8_t> padded, comp(level_ > 0 ? compressBound((uLong)raw_cap) : 0);
            for (;;) {
                const int64_t j = next.fetch_add(1);
                if (j >= jobs || !errs[t].empty()) return;
                const int64_t s = j / (nvars * nstrips), k = (j / nstrips) % nvars, st = j % nstrips;
                const int64_t r0 = st * strip_rows, nr = std::min(strip_rows, rows - r0);
                const uint8_t* src = recs + s * rec_bytes + 8 + k * rows * cols * 4 + r0 * cols * 4;
                if (nr < strip_rows) {                         // an edge chunk is stored whole: pad it with the fill value
                    padded.resize(raw_cap);
                    memcpy(padded.data(), src, (size_t)(nr * cols * 4));
                    for (size_t o = (size_t)(nr * cols * 4); o < raw_cap; o += 4) store_i32(padded.data() + o, kMissval);
                    src = padded.data();
                }
                const uint8_t* out = src;
                size_t out_n = raw_cap;
                if (level_ > 0) {
                    uLongf cn = (uLongf)comp.size();
                    if (compress2(comp.data(), &cn, src, (uLong)raw_cap, level_) != Z_OK) { errs[t] = "zlib compress2 failed"; return; }
                    out = comp.data(); out_n = (size_t)cn;
                }
                const hsize_t off[3] = {(hsize_t)(step0 + s), (hsize_t)r0, 0};
                std::lock_guard<std::mutex> lk(H5Api::lock());
                H5Api::Quiet quiet(h5);
                if (h5->Dwrite_chunk(dsets_[k], 0, 0, off, out_n, out) < 0) { errs[t] = "H5Dwrite_chunk failed"; return; }
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        for (auto& e : errs) if (!e.empty()) return e;
        return "";
    }

    std::string close() override {
        if (!h5) return "";
        std::lock_guard<std::mutex> lk(H5Api::lock());
        H5Api::Quiet quiet(h5);
        return close_locked();
    }

private:
    std::string close_locked() {
        std::string e;
        if (!h5) return e;
        for (hid_t d : dsets_) if (h5->Dclose(d) < 0) e = "H5Dclose failed";
        dsets_.clear();
        if (file_ >= 0 && h5->Fclose(file_) < 0) e = "H5Fclose failed";
        file_ = -1;
        return e;
    }

    const H5Api* h5 = nullptr;
    hid_t file_ = -1;
    std::vector<hid_t> dsets_;
    int level_ = 9;
    int64_t strip_rows = 1, nstrips = 1;

    bool att_text(hid_t obj, const char* name, const char* text) {
        const size_t n = strlen(text);
        const hid_t t = h5->Tcopy(h5->T_C_S1);
        bool ok = t >= 0 && h5->Tset_size(t, n ? n : 1) >= 0 && h5->Tset_strpad(t, 0 /* H5T_STR_NULLTERM */) >= 0;
        const hid_t sp = h5->Screate(0 /* H5S_SCALAR */);
        const hid_t a = ok && sp >= 0 ? h5->Acreate2(obj, name, t, sp, 0, 0) : -1;
        const char zero = 0;
        ok = a >= 0 && h5->Awrite(a, t, n ? (const void*)text : (const void*)&zero) >= 0;
        if (a >= 0) h5->Aclose(a);
        if (sp >= 0) h5->Sclose(sp);
        if (t >= 0) h5->Tclose(t);
        return ok;
    }
    bool att_int(hid_t obj, const char* name, hid_t filetype, bool scalar, int v) {
        const hsize_t one = 1;
        const hid_t sp = scalar ? h5->Screate(0) : h5->Screate_simple(1, &one, nullptr);
        const hid_t a = sp >= 0 ? h5->Acreate2(obj, name, filetype, sp, 0, 0) : -1;
        const bool ok = a >= 0 && h5->Awrite(a, h5->T_NATIVE_INT, &v) >= 0;
        if (a >= 0) h5->Aclose(a);
        if (sp >= 0) h5->Sclose(sp);
        return ok;
    }
    bool att_fill(hid_t ds) { return att_int(ds, "_FillValue", h5->T_STD_I32BE, false, kMissval); }
    // a dimension with its coordinate variable: a 1-D double dataset made a dimension scale under its own name
    bool coord(hid_t& ds, const char* name, int64_t n, const double* vals, int dimid) {
        const unsigned order = 0x0001u | 0x0002u;
        const hid_t dcpl = h5->Pcreate(h5->P_DATASET_CREATE);
        h5->Pset_attr_creation_order(dcpl, order);
        const hsize_t len = (hsize_t)n;
        const hid_t sp = h5->Screate_simple(1, &len, nullptr);
        ds = sp >= 0 ? h5->Dcreate2(file_, name, h5->T_IEEE_F64LE, sp, 0, dcpl, 0) : -1;
        bool ok = ds >= 0 && (n == 0 || h5->Dwrite(ds, h5->T_NATIVE_DOUBLE, 0, 0, 0, vals) >= 0);
        ok = ok && h5->DSset_scale(ds, name) >= 0 && att_int(ds, "_Netcdf4Dimid", h5->T_NATIVE_INT, true, dimid);
        if (sp >= 0) h5->Sclose(sp);
        if (dcpl >= 0) h5->Pclose(dcpl);
        return ok;
    }
};

}  // namespace mcf
