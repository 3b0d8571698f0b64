// tests/lane_emu.cpp -- TEST HARNESS ONLY (built into tests/_build/, never part of the product).
//
// Compiles the per-lane device logic of toypathtracer_amd/csrc (tpt_math.h, tpt_trace.h,
// tpt_scene.h) for the HOST and runs one lane at a time over an image, so that the `-m "not gpu"`
// suite can check the flattened Trace/Scatter state machine, the two-phase HitSpheres and the scene
// packing against the oracle without a GPU.  What it cannot cover (wave-level refill, LDS staging,
// atomics, the device compiler) is covered by the `-m gpu` parity tests.
#include "tpt_scene.h"
#include <stdint.h>
#include <vector>

using namespace tpt;

extern "C" {

// spheres/mats: reference layouts (20 B / 36 B).  cam: 88 B.  Returns ray count.
int64_t emu_render(const void* spheres, const void* mats, int count, const void* cam, int w, int h, int y0, int y1,
                   int spp, int frame, unsigned flags, int seedMode, int hs, int fold, float* backbuffer)
{
    std::vector<SpherePOD> S((const SpherePOD*)spheres, (const SpherePOD*)spheres + count);
    std::vector<MaterialPOD> M((const MaterialPOD*)mats, (const MaterialPOD*)mats + count);
    PackedScene P;
    packScene(S, M, P);
    SceneView sv = viewOf(P);
    CameraPOD c;
    memcpy(&c, cam, sizeof(c));
    FrameConsts fc = makeFrameConsts(c, w, h, spp, frame, flags, seedMode);
    f4 stackMem[TPT_MAX_DEPTH];
    BounceStack stack;
    stack.base = stackMem;
    stack.stride = 1;
    int64_t rays = 0;
    for (int y = y0; y < y1; ++y) {
        Lane L;
        L.rays = 0;
        L.active = false;
        for (int x = 0; x < w; ++x) {
            laneBeginPixel(L, fc, x, y, y * w + x, seedMode == SEED_PER_PIXEL || x == 0);
            L.prev = ld3(backbuffer + (size_t)L.pix * 4);
            for (;;) {
                bool done;
                if (hs == HS_SIMPLE)
                    done = fold == FOLD_FORWARD ? laneStep<HS_SIMPLE, FOLD_FORWARD>(L, sv, fc, stack)
                                                : laneStep<HS_SIMPLE, FOLD_RECURSIVE>(L, sv, fc, stack);
                else
                    done = fold == FOLD_FORWARD ? laneStep<HS_TWO_PHASE, FOLD_FORWARD>(L, sv, fc, stack)
                                                : laneStep<HS_TWO_PHASE, FOLD_RECURSIVE>(L, sv, fc, stack);
                if (done) break;
            }
            laneStorePixel(L, fc, backbuffer);
        }
        rays += L.rays;
    }
    return rays;
}

// default scene / camera as the product builds them (compared with the reference's GetSceneDesc)
int emu_default_scene(void* spheres, void* mats, int capacity)
{
    std::vector<SpherePOD> S;
    std::vector<MaterialPOD> M;
    defaultScene(S, M);
    if ((int)S.size() > capacity) return -1;
    PackedScene P;
    packScene(S, M, P); // fills invRadius
    memcpy(spheres, S.data(), S.size() * sizeof(SpherePOD));
    memcpy(mats, M.data(), M.size() * sizeof(MaterialPOD));
    return (int)S.size();
}
void emu_default_camera(void* cam, int w, int h)
{
    CameraPOD c = makeCamera(defaultCameraSetup(), float(w) / float(h));
    memcpy(cam, &c, sizeof(c));
}
float emu_sinf(float x) { return tsinf(x); }
float emu_cosf(float x) { return tcosf(x); }
float emu_pow5f(float x) { return tpow5f(x); }

} // extern "C"
