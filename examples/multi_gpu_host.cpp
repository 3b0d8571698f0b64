// examples/multi_gpu_host.cpp -- a C++ host that renders ONE image on N GPUs of a node with nothing but the C ABI of
// libtoypathtracer_hip.so: one process per GPU (fork), the 128-byte RCCL id handed from rank 0 to the others through pipes,
// every rank calls tptDrawSharded per frame, rank 0 ends up with the assembled image.  What the reference does with an
// enkiTS task set over rows inside one process (Cpp/Source/Test.cpp:357-361), across GPUs.
//
//   g++ -O2 -I include examples/multi_gpu_host.cpp -L toypathtracer_amd/lib -ltoypathtracer_hip \
//       -Wl,-rpath,$PWD/toypathtracer_amd/lib -o examples/multi_gpu_host
//   examples/multi_gpu_host [ranks=1] [width=1280] [height=720] [frames=20] [stripeRows=8] [framesPerLaunch=1]
//
// Prints rays, Mray/s and the FNV-1a hash of the final image (rank 0); with the same arguments the hash is the same for
// every number of ranks (seeds depend on the global pixel position only).
#include "tpt_hip.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

// the HIP runtime is used by the library only; this host needs two calls of it to own the image buffer
extern "C" int hipMalloc(void**, size_t);
extern "C" int hipMemcpy(void*, const void*, size_t, int);
extern "C" int hipMemset(void*, int, size_t);

static void die(const char* what)
{
    fprintf(stderr, "multi_gpu_host: %s: %s\n", what, tptGetLastError());
    exit(1);
}

static int runRank(int rank, int ranks, const char* id, int w, int h, int frames, int stripeRows, int batch)
{
    char dev[16];
    snprintf(dev, sizeof(dev), "%d", rank);
    setenv("TPT_DEVICE", dev, 1);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    if (tptInitialize()) die("tptInitialize");
    if (tptCommInit(id, ranks, rank, stripeRows)) die("tptCommInit");
    float* image = nullptr;
    if (rank == 0) {
        if (hipMalloc(reinterpret_cast<void**>(&image), (size_t)w * h * 16) || hipMemset(image, 0, (size_t)w * h * 16)) die("hipMalloc");
    }
    const unsigned flags = TPT_FLAG_PROGRESSIVE;
    for (int f = 0; f < 4; ++f) { // warm-up (not timed; the accumulation restarts at frame 0 below)
        if (tptUpdate(0.0f, f, w, h, flags) || tptDrawSharded(0.0f, f, w, h, image, flags)) die("warm-up");
    }
    int64_t rays0 = 0, rays1 = 0;
    if (tptShardedFinish(&rays0)) die("tptShardedFinish");
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f += batch) { // batch > 1: that many frames per launch and per exchange (same image)
        const int n = frames - f < batch ? frames - f : batch;
        if (tptUpdate(0.0f, f, w, h, flags) || tptDrawShardedBatch(0.0f, f, n, w, h, image, flags)) die("tptDrawSharded");
    }
    if (tptShardedFinish(&rays1)) die("tptShardedFinish");
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) {
        std::vector<float> host((size_t)w * h * 4);
        if (hipMemcpy(host.data(), image, host.size() * 4, 2 /* hipMemcpyDeviceToHost */)) die("hipMemcpy");
        unsigned hsh = 0x811c9dc5u;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(host.data());
        for (size_t i = 0; i < host.size() * 4; ++i) hsh = (hsh ^ p[i]) * 16777619u;
        printf("%d rank(s), %dx%d, %d frames: %lld rays, %.3f ms/frame, %.1f Mray/s, fnv %08x (device %s)\n", ranks, w, h, frames,
               (long long)(rays1 - rays0), s / frames * 1e3, (rays1 - rays0) / s * 1e-6, hsh, tptGetDeviceName());
    }
    if (tptCommDestroy() || tptShutdown()) die("shutdown");
    return 0;
}

int main(int argc, char** argv)
{
    const int ranks = argc > 1 ? atoi(argv[1]) : 1, w = argc > 2 ? atoi(argv[2]) : 1280, h = argc > 3 ? atoi(argv[3]) : 720,
              frames = argc > 4 ? atoi(argv[4]) : 20, stripeRows = argc > 5 ? atoi(argv[5]) : 8, batch = argc > 6 ? atoi(argv[6]) : 1;
    if (ranks < 1 || ranks > 64 || batch < 1 || batch > 32) return 2;
    // children first (they must not inherit an initialised HIP runtime), each with a pipe it reads the id from
    std::vector<int> wr(ranks, -1);
    std::vector<pid_t> pids(ranks, 0);
    for (int r = 1; r < ranks; ++r) {
        int fd[2];
        if (pipe(fd)) return 3;
        pid_t pid = fork();
        if (pid == 0) {
            close(fd[1]);
            char id[TPT_COMM_ID_BYTES];
            size_t got = 0;
            while (got < sizeof(id)) {
                ssize_t n = read(fd[0], id + got, sizeof(id) - got);
                if (n <= 0) return 4;
                got += (size_t)n;
            }
            return runRank(r, ranks, id, w, h, frames, stripeRows, batch);
        }
        close(fd[0]);
        wr[r] = fd[1];
        pids[r] = pid;
    }
    char id[TPT_COMM_ID_BYTES];
    if (tptCommGetUniqueId(id)) die("tptCommGetUniqueId");
    for (int r = 1; r < ranks; ++r) {
        if (write(wr[r], id, sizeof(id)) != (ssize_t)sizeof(id)) return 5;
        close(wr[r]);
    }
    int rc = runRank(0, ranks, id, w, h, frames, stripeRows, batch);
    for (int r = 1; r < ranks; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 6;
    }
    return rc;
}
