// tpt_hostcopy.h -- helper threads of the host-pointer DrawTest (tptDraw): copies between the caller's pageable image and
// the pinned staging buffer the DMA engines read and write.
//
// Why: hipMemcpyAsync on pageable memory does not return before the copy is done, so the upload of the previous image and
// the download of the new one (Test.cpp:344-367's contract: the backbuffer is read AND written by every DrawTest) ran one
// after the other, 2 x 0.27 ms for a 1280x720 image on a link that is full duplex.  Through pinned memory both directions
// overlap; the copy between the caller's buffer and the staging buffer then has to be faster than the link (55 GB/s), which
// takes more than one core (profiles/r03/r03_h2d_probe.log: 1 thread 50 GB/s, 2: 100, 4: 200).
//
// The helpers spin for work for kSpinUs after their last job (a synchronous caller's next frame comes within that) and
// sleep on a condition variable otherwise; the calling thread always copies the first slice itself.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <thread>
#include <vector>
#include <immintrin.h>

namespace tpt {

class HostCopyPool {
public:
    explicit HostCopyPool(int threads) : k_(threads < 1 ? 1 : threads)
    {
        for (int i = 1; i < k_; ++i) helpers_.emplace_back([this, i] { run(i); });
    }
    ~HostCopyPool()
    {
        quit_.store(true);
        { std::lock_guard<std::mutex> lock(m_); }
        cv_.notify_all();
        for (auto& t : helpers_) t.join();
    }
    HostCopyPool(const HostCopyPool&) = delete;
    HostCopyPool& operator=(const HostCopyPool&) = delete;
    int threads() const { return k_; }

    // memcpy(dst, src, bytes) by all threads; returns when every byte is in place
    void copy(void* dst, const void* src, size_t bytes)
    {
        if (k_ == 1 || bytes < kMinParallelBytes) {
            memcpy(dst, src, bytes);
            return;
        }
        dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes;
        done_.store(0, std::memory_order_relaxed);
        gen_.fetch_add(1); // (seq_cst: ordered against the sleepers_ read below, see run())
        if (sleepers_.load() > 0) {
            { std::lock_guard<std::mutex> lock(m_); }
            cv_.notify_all();
        }
        slice(0);
        while (done_.load(std::memory_order_acquire) != (unsigned)(k_ - 1)) _mm_pause();
    }

private:
    static constexpr size_t kMinParallelBytes = 256 * 1024;
    static constexpr int kSpinUs = 500;

    void slice(int i) const
    {
        const size_t a = (bytes_ * (size_t)i / (size_t)k_) & ~(size_t)63;
        const size_t b = i + 1 == k_ ? bytes_ : (bytes_ * (size_t)(i + 1) / (size_t)k_) & ~(size_t)63;
        if (b > a) memcpy(dst_ + a, src_ + a, b - a);
    }
    void run(int i)
    {
        using clock = std::chrono::steady_clock;
        unsigned seen = 0;
        auto idleSince = clock::now();
        int polls = 0;
        for (;;) {
            if (quit_.load(std::memory_order_relaxed)) return;
            const unsigned g = gen_.load(std::memory_order_acquire);
            if (g != seen) {
                seen = g;
                slice(i);
                done_.fetch_add(1, std::memory_order_release);
                idleSince = clock::now();
                polls = 0;
                continue;
            }
            _mm_pause();
            if (++polls < 256) continue;
            polls = 0;
            if (clock::now() - idleSince < std::chrono::microseconds(kSpinUs)) continue;
            // nothing for a while: sleep.  copy() increments gen_ BEFORE it reads sleepers_, this thread increments sleepers_
            // BEFORE the predicate reads gen_ (all seq_cst): either copy() sees the sleeper and notifies, or the predicate
            // sees the new generation and the wait returns at once.
            std::unique_lock<std::mutex> lock(m_);
            sleepers_.fetch_add(1);
            cv_.wait(lock, [&] { return gen_.load() != seen || quit_.load(); });
            sleepers_.fetch_sub(1);
            idleSince = clock::now();
        }
    }

    const int k_;
    std::vector<std::thread> helpers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> gen_{0}, done_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> quit_{false};
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0;
};

} // namespace tpt
