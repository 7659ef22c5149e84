// mcf_hostpipe.hpp — device -> pageable host copies at PCIe rate.
//
// The one-shot entry points hand results back into caller-owned pageable memory (R vectors, numpy
// arrays).  hipMemcpy stages such copies through an internal pinned buffer with a single-threaded
// CPU copy (measured 25 GB/s on the MI355X box), which bounds mcf_runmicro1..4 end to end: the solver
// produces 10 GB of results in 8 ms and they take 0.41 s to come back (0.31 s through this pipe).
// HostPipe keeps the DMA engine busy instead: results stream into a ring of pinned pieces with hipMemcpyAsync on a dedicated copy
// stream while a small pool of host threads copies finished pieces into the destination (each thread
// a contiguous slice, so first-touch page faults of a fresh destination are spread over the threads).
// Host-to-device needs no such help: hipMemcpyAsync from pageable memory already runs at ~40 GB/s here
// (a pinned-ring upload measured slower, 28 GB/s, and was dropped).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace mcf {

class HostPipe {
public:
    static constexpr int kBufs = 4;
    static constexpr size_t kPiece = (size_t)32 << 20;

    HostPipe() = default;
    HostPipe(const HostPipe&) = delete;
    HostPipe& operator=(const HostPipe&) = delete;
    ~HostPipe() { shutdown(); }

    // false: could not set up (caller falls back to a plain hipMemcpy)
    bool init() {
        if (ready_) return true;
        if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return false;
        for (int b = 0; b < kBufs; ++b) {
            if (hipHostMalloc(&pin_[b], kPiece, hipHostMallocDefault) != hipSuccess) { shutdown(); return false; }
            if (hipEventCreateWithFlags(&ev_[b], hipEventDisableTiming) != hipSuccess) { shutdown(); return false; }
        }
        int nt = 0;
        if (const char* e = getenv("MCF_COPY_THREADS")) nt = atoi(e);
        if (nt <= 0) nt = (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
        nthreads_ = nt;
        for (int t = 1; t < nthreads_; ++t) workers_.emplace_back([this, t] { worker(t); });
        ready_ = true;
        return true;
    }

    // Copies `bytes` from device memory to pageable host memory; `after` (may be null) is an event on
    // the producing stream that the first DMA must wait for.  Returns when `dst` is complete.
    // The same into a PITCHED destination: `height` rows of `width` bytes, contiguous on the device, land `dpitch` bytes apart
    // in host memory (a row block of a taller column-major raster: width = the block's rows x 8 B, one row per raster
    // column and step).  hipMemcpy2D into pageable memory copies such rows one by one from a single thread (measured: half
    // the contiguous rate at 8 KB rows); here the DMA stays contiguous and the copy threads do the scatter.
    hipError_t copy_pitched(void* dst, size_t dpitch, const void* dev_src, size_t width, size_t height, hipEvent_t after) {
        if (width == 0 || height == 0) return hipSuccess;
        if (dpitch == width) return copy(dst, dev_src, width * height, after);
        if (after) {
            hipError_t e = hipStreamWaitEvent(stream_, after, 0);
            if (e != hipSuccess) return e;
        }
        const size_t rows_per_piece = std::max<size_t>(1, kPiece / width);
        if (width > kPiece) return hipErrorInvalidValue;              // (a single row larger than a piece: the caller falls back)
        const size_t npieces = (height + rows_per_piece - 1) / rows_per_piece;
        auto issue = [&](size_t i) -> hipError_t {
            const int b = (int)(i % kBufs);
            const size_t r0 = i * rows_per_piece, nr = std::min(rows_per_piece, height - r0);
            hipError_t e = hipMemcpyAsync(pin_[b], (const char*)dev_src + r0 * width, nr * width, hipMemcpyDeviceToHost, stream_);
            if (e != hipSuccess) return e;
            return hipEventRecord(ev_[b], stream_);
        };
        for (size_t i = 0; i < std::min<size_t>(kBufs, npieces); ++i) {
            hipError_t e = issue(i);
            if (e != hipSuccess) return e;
        }
        for (size_t i = 0; i < npieces; ++i) {
            const int b = (int)(i % kBufs);
            hipError_t e = hipEventSynchronize(ev_[b]);
            if (e != hipSuccess) return e;
            const size_t r0 = i * rows_per_piece, nr = std::min(rows_per_piece, height - r0);
            parallel_scatter((char*)dst + r0 * dpitch, dpitch, (const char*)pin_[b], width, nr);
            if (i + kBufs < npieces && (e = issue(i + kBufs)) != hipSuccess) return e;
        }
        return hipSuccess;
    }

    hipError_t copy(void* dst, const void* dev_src, size_t bytes, hipEvent_t after) {
        // A fresh destination is first touched by the copy threads below, and with 4 KB pages those faults, not PCIe, bound the
        // call.  numpy asks for transparent huge pages on its big arrays itself; R's vectors (plain malloc) do not — so the
        // 2 MB-aligned inside of the destination is advised here (a no-op where THP is `never` or already `always`).
        static const bool thp = getenv("MCF_NO_THP") == nullptr;
        if (thp && bytes >= ((size_t)8 << 20)) {
            const uintptr_t two = (uintptr_t)2 << 20;
            const uintptr_t a = ((uintptr_t)dst + two - 1) & ~(two - 1), b = ((uintptr_t)dst + bytes) & ~(two - 1);
            if (b > a) (void)madvise((void*)a, b - a, MADV_HUGEPAGE);
        }
        if (after) {
            hipError_t e = hipStreamWaitEvent(stream_, after, 0);
            if (e != hipSuccess) return e;
        }
        const size_t npieces = (bytes + kPiece - 1) / kPiece;
        auto issue = [&](size_t i) -> hipError_t {
            const int b = (int)(i % kBufs);
            const size_t off = i * kPiece, n = std::min(kPiece, bytes - off);
            hipError_t e = hipMemcpyAsync(pin_[b], (const char*)dev_src + off, n, hipMemcpyDeviceToHost, stream_);
            if (e != hipSuccess) return e;
            return hipEventRecord(ev_[b], stream_);
        };
        for (size_t i = 0; i < std::min<size_t>(kBufs, npieces); ++i) {
            hipError_t e = issue(i);
            if (e != hipSuccess) return e;
        }
        for (size_t i = 0; i < npieces; ++i) {
            const int b = (int)(i % kBufs);
            hipError_t e = hipEventSynchronize(ev_[b]);
            if (e != hipSuccess) return e;
            const size_t off = i * kPiece, n = std::min(kPiece, bytes - off);
            parallel_copy((char*)dst + off, (const char*)pin_[b], n);
            if (i + kBufs < npieces && (e = issue(i + kBufs)) != hipSuccess) return e;
        }
        return hipSuccess;
    }

private:
    // rows of `width` bytes from a contiguous source to rows `dpitch` apart: the rows are dealt to the copy threads in slices
    void parallel_scatter(char* dst, size_t dpitch, const char* src, size_t width, size_t nrows) {
        if (nthreads_ <= 1 || nrows * width < ((size_t)1 << 20)) {
            for (size_t r = 0; r < nrows; ++r) memcpy(dst + r * dpitch, src + r * width, width);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_dst_ = dst; job_src_ = src; job_n_ = nrows; job_width_ = width; job_dpitch_ = dpitch;
            pending_ = nthreads_ - 1;
            ++generation_;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        job_width_ = 0;
    }
    void parallel_copy(char* dst, const char* src, size_t n) {
        if (nthreads_ <= 1 || n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_dst_ = dst; job_src_ = src; job_n_ = n;
            pending_ = nthreads_ - 1;
            ++generation_;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
    }
    void slice(int t) {
        if (job_width_) {               // scatter job: job_n_ rows
            const size_t per = (job_n_ + nthreads_ - 1) / nthreads_;
            const size_t a = std::min(job_n_, per * t), b = std::min(job_n_, a + per);
            for (size_t r = a; r < b; ++r) memcpy(job_dst_ + r * job_dpitch_, job_src_ + r * job_width_, job_width_);
            return;
        }
        const size_t per = ((job_n_ / nthreads_) + 4095) & ~(size_t)4095;   // page-aligned slices
        const size_t a = std::min(job_n_, per * t), b = std::min(job_n_, a + per);
        const size_t end = (t == nthreads_ - 1) ? job_n_ : b;
        if (end > a) memcpy(job_dst_ + a, job_src_ + a, end - a);
    }
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            slice(t);
            {
                std::lock_guard<std::mutex> lk(mu_);
                --pending_;
            }
            done_cv_.notify_one();
        }
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
        workers_.clear();
        for (int b = 0; b < kBufs; ++b) {
            if (pin_[b]) (void)hipHostFree(pin_[b]);
            if (ev_[b]) (void)hipEventDestroy(ev_[b]);
            pin_[b] = nullptr; ev_[b] = nullptr;
        }
        if (stream_) (void)hipStreamDestroy(stream_);
        stream_ = nullptr;
        ready_ = false;
        stop_ = false;
    }

    bool ready_ = false;
    hipStream_t stream_ = nullptr;
    void* pin_[kBufs] = {};
    hipEvent_t ev_[kBufs] = {};
    int nthreads_ = 1;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool stop_ = false;
    char* job_dst_ = nullptr;
    const char* job_src_ = nullptr;
    size_t job_n_ = 0, job_width_ = 0, job_dpitch_ = 0;
};

}  // namespace mcf
