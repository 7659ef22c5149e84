// mcf_ncfile.hpp — the file side of the `writetonc` sink (reference R/dataprep.R:1063-1260).
//
// writetonc goes through ncdf4 -> libnetcdf -> HDF5 (netCDF-4, deflate 9).  That container is mcf_nc4file.hpp (format =
// MCF_NC_NETCDF4; it needs an HDF5 library on the host at run time).  This is the container that needs nothing and streams
// fastest (format = MCF_NC_CLASSIC, the default): it writes the same dataset — dimensions east, north, time;
// int variables named as the solver's outputs with writetonc's long names, units and missval -9999; the `crs`
// variable and the time attributes of add_crs_info — in netCDF CLASSIC 64-bit-offset format ("CDF\x02"), which
// ncdf4::nc_open, terra and every other netCDF reader open like a netCDF-4 file.  Differences a reader can see:
// no compression, and `time` is the record dimension, so that
//   * one time step of all variables is one contiguous record — a solved day streams to disk as ONE sequential
//     write of 24 records, produced in its final byte order by the device (k_pack_nc), and
//   * a variable is not bounded by the classic format's 4 GiB per fixed-size variable.
//
// Layout: header | east[cols] f64 | north[rows] f64 | crs i32 | records…,  record = time f64 | var0[rows][cols] i32 | …
// Everything is big-endian (the classic format's byte order).
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace mcf {

struct NcVarDef {
    std::string name, long_name, units;
};

class NcFile {
public:
    static constexpr int32_t kMissval = -9999;   // writetonc's missval
    static constexpr int kWriteThreads = 2;   // MCF_NC_WRITE_THREADS overrides

    NcFile() = default;
    NcFile(const NcFile&) = delete;
    NcFile& operator=(const NcFile&) = delete;
    virtual ~NcFile() { NcFile::close(); }

    int64_t rows = 0, cols = 0, nsteps = 0;
    int nvars = 0;
    int64_t rec_bytes = 0;      // 8 + nvars * rows * cols * 4
    int64_t rec_begin = 0;      // file offset of record 0
    std::vector<double> time_hours;

    // "" on success, else the reason
    std::string create(const char* path, int64_t rows_, int64_t cols_, int64_t nsteps_, const double* east,
                       const double* north, const double* time_hours_, const char* crs_wkt,
                       const std::vector<NcVarDef>& vars) {
        rows = rows_; cols = cols_; nsteps = nsteps_; nvars = (int)vars.size();
        const int64_t slab = rows * cols * 4;
        if (slab >= ((int64_t)1 << 32) - 4) return "one time step of a variable must stay below 4 GiB in the classic format";
        if (nsteps > INT32_MAX) return "too many time steps";
        rec_bytes = 8 + (int64_t)nvars * slab;
        time_hours.assign(time_hours_, time_hours_ + nsteps);

        // The header holds absolute offsets, whose position depends on the header's own length: build it twice.
        std::vector<uint8_t> h;
        int64_t fixed_begin = 0;
        for (int pass = 0; pass < 2; ++pass) {
            h.clear();
            put_bytes(h, "CDF\x02", 4);
            put_i32(h, (int32_t)nsteps);                                  // numrecs
            put_i32(h, 0x0A); put_i32(h, 3);                              // dim_list: east, north, time
            put_name(h, "east"); put_i32(h, (int32_t)cols);
            put_name(h, "north"); put_i32(h, (int32_t)rows);
            put_name(h, "time"); put_i32(h, 0);                           // the record dimension
            put_i32(h, 0); put_i32(h, 0);                                 // no global attributes
            put_i32(h, 0x0B); put_i32(h, 4 + nvars);                      // var_list
            int64_t off = fixed_begin;
            // coordinate variables as ncdim_def/nc_create lay them down
            put_var_head(h, "east", {0});
            put_i32(h, 0x0C); put_i32(h, 2);
            put_att_text(h, "units", "metres"); put_att_text(h, "long_name", "Eastings");
            put_var_tail(h, 6, cols * 8, off); off += cols * 8;
            put_var_head(h, "north", {1});
            put_i32(h, 0x0C); put_i32(h, 2);
            put_att_text(h, "units", "metres"); put_att_text(h, "long_name", "Northings");
            put_var_tail(h, 6, rows * 8, off); off += rows * 8;
            put_var_head(h, "crs", {});
            put_i32(h, 0x0C); put_i32(h, 2);
            put_att_text(h, "crs_wkt", crs_wkt ? crs_wkt : "");
            put_att_text(h, "grid_mapping_name", "longitude_latitude");
            put_var_tail(h, 4, 4, off); off += 4;
            rec_begin = off;
            put_var_head(h, "time", {2});
            put_i32(h, 0x0C); put_i32(h, 3);
            put_att_text(h, "units", "hours since 1970-01-01 00:00");
            put_att_text(h, "standard_name", "time"); put_att_text(h, "calendar", "gregorian");
            put_var_tail(h, 6, 8, off); off += 8;
            for (const NcVarDef& v : vars) {
                put_var_head(h, v.name.c_str(), {2, 1, 0});
                put_i32(h, 0x0C); put_i32(h, 4);
                put_att_text(h, "units", v.units.c_str());
                put_name(h, "_FillValue"); put_i32(h, 4); put_i32(h, 1); put_i32(h, kMissval);
                put_att_text(h, "long_name", v.long_name.c_str());
                put_att_text(h, "grid_mapping", "crs");
                put_var_tail(h, 4, slab, off); off += slab;
            }
            fixed_begin = (int64_t)h.size();
        }
        // an existing file is removed rather than truncated: ext4 treats truncate-then-rewrite as a replace and makes
        // close() wait for the data to reach the disk (auto_da_alloc), 1.4 s for a 13 GB file on the test box
        (void)::unlink(path);
        fd_ = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fd_ < 0) return std::string("cannot create ") + path + ": " + strerror(errno);
        std::vector<uint8_t> fixed;
        for (int64_t i = 0; i < cols; ++i) put_f64(fixed, east[i]);
        for (int64_t i = 0; i < rows; ++i) put_f64(fixed, north[i]);
        put_i32(fixed, 1);                                                // ncvar_put(ncnew, "crs", 1)
        std::string e = write_at(h.data(), h.size(), 0);
        if (e.empty()) e = write_at(fixed.data(), fixed.size(), (int64_t)h.size());
        // size the file now, so that records may arrive in any order and unwritten ones read as zeros
        if (e.empty() && ::ftruncate(fd_, rec_begin + nsteps * rec_bytes) != 0) e = std::string("ftruncate: ") + strerror(errno);
        if (!e.empty()) close();
        return e;
    }

    // records [step0, step0 + n) as they lie in the file, except for the 8 leading time bytes of each, which are set here
    virtual std::string write_records(int64_t step0, int64_t n, uint8_t* recs) {
        if (fd_ < 0) return "file is closed";
        if (step0 < 0 || n < 0 || step0 + n > nsteps) return "record range outside the file";
        for (int64_t s = 0; s < n; ++s) store_f64(recs + s * rec_bytes, time_hours[step0 + s]);
        const int64_t bytes = n * rec_bytes, off0 = rec_begin + step0 * rec_bytes;
        // A buffered write is a copy into the page cache under the file's inode lock, so writers to ONE file take turns:
        // measured on the MI355X box 8.9 GB/s with 2 threads, 8.7 with 8, 8.3 with 32.  Two threads keep the lock busy
        // while one of them is between pwrite calls; more only contend.
        static const int max_threads = [] { const char* e = getenv("MCF_NC_WRITE_THREADS"); int n = e ? atoi(e) : 0; return n > 0 ? n : kWriteThreads; }();
        const int nt = (int)std::min<int64_t>(max_threads, bytes / ((int64_t)16 << 20));
        if (nt <= 1) return write_at(recs, (size_t)bytes, off0);
        std::vector<std::string> errs(nt);
        std::vector<std::thread> th;
        const int64_t per = ((bytes / nt) + 4095) & ~(int64_t)4095;
        for (int t = 0; t < nt; ++t) {
            const int64_t a = std::min(bytes, per * t), b = (t == nt - 1) ? bytes : std::min(bytes, a + per);
            th.emplace_back([this, &errs, t, recs, a, b, off0] { errs[t] = write_at(recs + a, (size_t)(b - a), off0 + a); });
        }
        for (auto& x : th) x.join();
        for (auto& e : errs) if (!e.empty()) return e;
        return "";
    }

    virtual std::string close() {
        std::string e;
        if (fd_ >= 0 && ::close(fd_) != 0) e = std::string("close: ") + strerror(errno);
        fd_ = -1;
        return e;
    }
    bool is_open() const { return fd_ >= 0; }

    static void store_i32(uint8_t* p, int32_t v) {
        const uint32_t u = (uint32_t)v;
        p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
    }
    static void store_f64(uint8_t* p, double v) {
        uint64_t u;
        memcpy(&u, &v, 8);
        for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(u >> (56 - 8 * i));
    }

private:
    int fd_ = -1;

    std::string write_at(const uint8_t* p, size_t n, int64_t off) const {
        while (n > 0) {
            const ssize_t w = ::pwrite(fd_, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n, (off_t)off);
            if (w < 0) {
                if (errno == EINTR) continue;
                return std::string("pwrite: ") + strerror(errno);
            }
            p += w; n -= (size_t)w; off += w;
        }
        return "";
    }
    static void put_bytes(std::vector<uint8_t>& h, const char* s, size_t n) { h.insert(h.end(), s, s + n); }
    static void put_i32(std::vector<uint8_t>& h, int32_t v) { uint8_t b[4]; store_i32(b, v); h.insert(h.end(), b, b + 4); }
    static void put_i64(std::vector<uint8_t>& h, int64_t v) { put_i32(h, (int32_t)(v >> 32)); put_i32(h, (int32_t)(v & 0xffffffff)); }
    static void put_f64(std::vector<uint8_t>& h, double v) { uint8_t b[8]; store_f64(b, v); h.insert(h.end(), b, b + 8); }
    static void put_padded(std::vector<uint8_t>& h, const char* s, size_t n) {
        put_bytes(h, s, n);
        while (h.size() % 4) h.push_back(0);
    }
    static void put_name(std::vector<uint8_t>& h, const char* s) { put_i32(h, (int32_t)strlen(s)); put_padded(h, s, strlen(s)); }
    static void put_att_text(std::vector<uint8_t>& h, const char* name, const char* text) {
        put_name(h, name); put_i32(h, 2); put_i32(h, (int32_t)strlen(text)); put_padded(h, text, strlen(text));
    }
    static void put_var_head(std::vector<uint8_t>& h, const char* name, std::initializer_list<int> dims) {
        put_name(h, name); put_i32(h, (int32_t)dims.size());
        for (int d : dims) put_i32(h, d);
    }
    static void put_var_tail(std::vector<uint8_t>& h, int32_t type, int64_t vsize, int64_t begin) {
        put_i32(h, type); put_i32(h, (int32_t)(uint32_t)((vsize + 3) & ~(int64_t)3)); put_i64(h, begin);
    }
};

}  // namespace mcf
