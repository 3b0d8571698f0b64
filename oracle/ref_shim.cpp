// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// extern "C" veneer over the *pristine* reference CPU renderer (Cpp/Source/Test.h:10-17), so the
// Python tests and bench.py's cpu_baseline leg can drive it through ctypes.  This file contains no
// reference code; it only calls the reference's public Test.h API.  It is compiled together with
// the reference sources where they lie under /root/reference by oracle/build_ref.sh into
// oracle/_ref/libtpt_ref.so (git-ignored).  Precedent for such a veneer in the reference itself:
// Cpp/Emscripten/main.cpp:46-61.
#include "Test.h"
#include <stdint.h>

// Samples per pixel of the reference build.  The reference hard-codes DO_SAMPLES_PER_PIXEL
// (Config.h:22); build_ref.sh re-defines that macro to this variable while streaming Test.cpp to
// the compiler, so one reference build serves spp = 1/4/8/16.  Default 4 == Config.h:22.
int g_tpt_ref_spp = 4;

extern "C" {

void tptref_init(void) { InitializeTest(); }
void tptref_shutdown(void) { ShutdownTest(); }
void tptref_set_spp(int spp) { g_tpt_ref_spp = spp; }
int tptref_get_spp(void) { return g_tpt_ref_spp; }

void tptref_update(float time, int frame, int w, int h, unsigned flags)
{
    UpdateTest(time, frame, w, h, flags);
}

int tptref_draw(float time, int frame, int w, int h, float* backbuffer, unsigned flags)
{
    int rays = 0;
    DrawTest(time, frame, w, h, backbuffer, rays, flags);
    return rays;
}

void tptref_object_count(int* count, int* objSize, int* matSize, int* camSize)
{
    GetObjectCount(*count, *objSize, *matSize, *camSize);
}

void tptref_scene_desc(void* objs, void* mats, void* cam, void* emissives, int* emissiveCount)
{
    GetSceneDesc(objs, mats, cam, emissives, emissiveCount);
}

} // extern "C"
