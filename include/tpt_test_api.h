// tpt_test_api.h -- the reference's "Test API", verbatim in meaning: the drop-in boundary.
//
// These are the six free functions (C++ linkage) and two flags that every host of the reference
// links against (/root/reference/Cpp/Source/Test.h:4-17).  libtoypathtracer_hip.so exports them
// with identical signatures -- and therefore identical mangled names -- so an existing host keeps
// its own `#include "Test.h"` and simply links this library instead of Test.cpp + Maths.cpp + enkiTS.
// This header exists for hosts that do not have the reference tree at hand.
//
// Behavioural contract kept from the reference:
//   * call order per frame: UpdateTest, then DrawTest (TestWin.cpp:315-316);
//   * `backbuffer` is caller-owned host memory, width*height*4 floats, row 0 = bottom of the image,
//     RGB is blended in place with the previous contents (progressive accumulation), alpha is never
//     written (Test.cpp:291-296);
//   * `outRayCount` = number of HitWorld calls of this frame (camera + bounce + shadow rays);
//   * all functions return void; on a HIP failure the library prints to stderr and aborts (the
//     reference has no error channel, and there is no CPU fallback here).
// Differences, by design (see include/tpt_hip.h for the knobs):
//   * default RNG seeding is per pixel (the reference's own GPU formula) instead of per row, so the
//     default image is a different -- equally valid -- noise realisation than the CPU reference;
//     tptSetSeedMode(0) gives the reference's exact CPU image.
#pragma once
#include <stdint.h>
#if defined(__GNUC__)
#define TPT_CXX_API __attribute__((visibility("default")))
#else
#define TPT_CXX_API
#endif

enum TestFlags {
    kFlagAnimate = (1 << 0),
    kFlagProgressive = (1 << 1),
};

TPT_CXX_API void InitializeTest();
TPT_CXX_API void ShutdownTest();

TPT_CXX_API void UpdateTest(float time, int frameCount, int screenWidth, int screenHeight, unsigned testFlags);
TPT_CXX_API void DrawTest(float time, int frameCount, int screenWidth, int screenHeight, float* backbuffer, int& outRayCount, unsigned testFlags);

TPT_CXX_API void GetObjectCount(int& outCount, int& outObjectSize, int& outMaterialSize, int& outCamSize);
TPT_CXX_API void GetSceneDesc(void* outObjects, void* outMaterials, void* outCam, void* outEmissives, int* outEmissiveCount);
