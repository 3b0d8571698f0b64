// tools/probes/h2d_probe.cpp -- measurement tool, not part of the product.
// How fast can a 14.7 MB pageable host image (C2's backbuffer) reach the GPU?  (a) hipMemcpyAsync from the pageable buffer
// (what tptDraw did up to round 3), (b) k threads copy bands into a pinned staging buffer, each band's DMA issued as soon
// as the band is staged.  Also the way down: (c) hipMemcpyAsync into the pageable buffer, (d) DMA into pinned + k threads out.
// Build: hipcc -O2 -std=c++17 -pthread tools/probes/h2d_probe.cpp -o /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pool {
    int k;
    std::vector<std::thread> th;
    std::atomic<unsigned> gen{0}, done{0};
    std::atomic<bool> quit{false};
    const char* src = nullptr; char* dst = nullptr; size_t bytes = 0;
    explicit Pool(int k_) : k(k_) {
        for (int i = 1; i < k; ++i) th.emplace_back([this, i] { run(i); });
    }
    ~Pool() { quit = true; for (auto& t : th) t.join(); }
    void slice(int i) {
        size_t a = (bytes * i / k) & ~(size_t)63, b = i + 1 == k ? bytes : (bytes * (i + 1) / k) & ~(size_t)63;
        memcpy(dst + a, src + a, b - a);
    }
    void run(int i) {
        unsigned seen = 0;
        while (!quit.load(std::memory_order_relaxed)) {
            unsigned g = gen.load(std::memory_order_acquire);
            if (g == seen) { _mm_pause(); continue; }
            seen = g;
            slice(i);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void copy(char* d, const char* s, size_t n) {
        dst = d; src = s; bytes = n;
        done.store(0, std::memory_order_relaxed);
        gen.fetch_add(1, std::memory_order_release);
        slice(0);
        while (done.load(std::memory_order_acquire) != (unsigned)(k - 1)) _mm_pause();
    }
};

int main(int argc, char** argv)
{
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1280 * 720 * 16;
    const int reps = 30, bands = 4;
    char* host = (char*)aligned_alloc(4096, bytes);
    memset(host, 1, bytes);
    char *pinned, *dev;
    CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, bytes));
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    auto timeit = [&](const char* name, auto&& fn) {
        for (int i = 0; i < 5; ++i) fn();
        double best = 1e9, sum = 0;
        for (int i = 0; i < reps; ++i) { double t0 = now(); fn(); double t = now() - t0; sum += t; if (t < best) best = t; }
        printf("%-58s avg %.3f ms  best %.3f ms  (%.1f GB/s)\n", name, sum / reps * 1e3, best * 1e3, bytes / (sum / reps) / 1e9);
    };
    timeit("H2D pageable, one hipMemcpyAsync", [&] { CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    timeit("H2D pinned, one hipMemcpyAsync", [&] { CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    timeit("D2H pageable, one hipMemcpyAsync", [&] { CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
    timeit("D2H pinned, one hipMemcpyAsync", [&] { CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
    timeit("duplex pageable: H2D first half | D2H second half", [&] {
        CK(hipMemcpyAsync(dev, host, bytes / 2, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(host + bytes / 2, dev + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, st2));
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2)); });
    for (int k : {1, 2, 4, 8, 16}) {
        Pool pool(k);
        char name[128];
        snprintf(name, sizeof name, "host memcpy pageable -> pinned, %d threads", k);
        timeit(name, [&] { pool.copy(pinned, host, bytes); });
        snprintf(name, sizeof name, "H2D staged: %d threads, %d bands, DMA per band", k, bands);
        timeit(name, [&] {
            for (int b = 0; b < bands; ++b) {
                size_t a = bytes * b / bands, e = bytes * (b + 1) / bands;
                pool.copy(pinned + a, host + a, e - a);
                CK(hipMemcpyAsync(dev + a, pinned + a, e - a, hipMemcpyHostToDevice, st));
            }
            CK(hipStreamSynchronize(st)); });
        snprintf(name, sizeof name, "D2H staged: DMA per band, %d threads copy out", k);
        timeit(name, [&] {
            hipEvent_t ev[bands];
            for (int b = 0; b < bands; ++b) {
                size_t a = bytes * b / bands, e = bytes * (b + 1) / bands;
                CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
                CK(hipMemcpyAsync(pinned + a, dev + a, e - a, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev[b], st));
            }
            for (int b = 0; b < bands; ++b) {
                size_t a = bytes * b / bands, e = bytes * (b + 1) / bands;
                CK(hipEventSynchronize(ev[b]));
                pool.copy(host + a, pinned + a, e - a);
                CK(hipEventDestroy(ev[b]));
            } });
        snprintf(name, sizeof name, "round trip staged (up band b+1 | down band b), %d threads", k);
        timeit(name, [&] {
            // up: stage + DMA on st; down: DMA on st2 into pinned2 == pinned region (separate halves not needed for timing)
            for (int b = 0; b < bands; ++b) {
                size_t a = bytes * b / bands, e = bytes * (b + 1) / bands;
                pool.copy(pinned + a, host + a, e - a);
                CK(hipMemcpyAsync(dev + a, pinned + a, e - a, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(host + a, dev + a, e - a, hipMemcpyDeviceToHost, st)); // pageable on the way down, as tptDraw does
            }
            CK(hipStreamSynchronize(st)); });
    }
    return 0;
}
