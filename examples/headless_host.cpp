// examples/headless_host.cpp -- a host that knows ONLY the reference's Test API (Test.h:10-17), linked
// against libtoypathtracer_hip.so instead of the reference's Test.cpp + Maths.cpp + enkiTS.
// Shape of the loop follows the reference's only headless harness, Cs/Program.cs:16-31 (N frames,
// UpdateTest + DrawTest each, rays/seconds at the end) and its hosts' zero-initialised float backbuffer
// (TestWin.cpp:73-74).
//
//   g++ -O2 -I include examples/headless_host.cpp -L toypathtracer_amd/lib -ltoypathtracer_hip \
//       -Wl,-rpath,$PWD/toypathtracer_amd/lib -o examples/headless_host
//   (or -I /root/reference/Cpp/Source -include Test.h -DUSE_REFERENCE_HEADER: the reference's own header works unchanged)
#ifdef USE_REFERENCE_HEADER
#include "Test.h"
#else
#include "tpt_test_api.h"
#endif
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

int main(int argc, char** argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 1280, h = argc > 2 ? atoi(argv[2]) : 720, frames = argc > 3 ? atoi(argv[3]) : 30;
    std::vector<float> backbuffer((size_t)w * h * 4, 0.0f);
    InitializeTest();
    int count, objSize, matSize, camSize;
    GetObjectCount(count, objSize, matSize, camSize);
    printf("scene: %d spheres, sizeof(Sphere)=%d sizeof(Material)=%d sizeof(Camera)=%d\n", count, objSize, matSize, camSize);
    const unsigned flags = kFlagProgressive;
    long long rays = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) {
        UpdateTest(0.0f, f, w, h, flags);
        int r = 0;
        DrawTest(0.0f, f, w, h, backbuffer.data(), r, flags);
        rays += r;
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // FNV-1a over the float buffer, the hash BASELINE.md's golden vectors use
    unsigned hsh = 0x811c9dc5u;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(backbuffer.data());
    for (size_t i = 0; i < backbuffer.size() * 4; ++i) hsh = (hsh ^ p[i]) * 16777619u;
    printf("%dx%d %d frames: %lld rays, %.2f ms/frame, %.1f Mray/s (host buffer round trip included), fnv %08x\n", w, h, frames, rays,
           s / frames * 1e3, rays / s * 1e-6, hsh);
    ShutdownTest();
    return 0;
}
