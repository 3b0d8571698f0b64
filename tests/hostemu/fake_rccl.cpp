// TEST INFRASTRUCTURE (tests/hostemu): a stand-in for librccl.so.1 that csrc/tpt_host.cpp finds with dlopen when the host-emulation
// build of the library is asked for a communicator of more than one rank.  One process per rank, as in the real thing; the ranks meet in
// a directory (env FAKE_RCCL_DIR), the unique id names a sub-directory, ncclGather is a unit of work on the emulated stream it is given
// (hostemuEnqueue of tests/hostemu/hip_shim.cpp, looked up in the already loaded emulation library): every rank writes its send buffer
// to a file, the root waits for all of them and lays them out rank after rank -- the semantics of rccl.h's ncclGather, nothing else.
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

struct ncclComm {
    int nRanks, rank;
    std::string dir;
    unsigned long long seq;
};
namespace {
typedef void (*EnqueueFn)(hipStream_t, void (*)(void*), const void*, size_t);
EnqueueFn g_enqueue = nullptr;
bool findEnqueue()
{
    if (g_enqueue) return true;
    const char* lib = getenv("TPT_LIB");
    void* h = lib ? dlopen(lib, RTLD_NOW | RTLD_NOLOAD) : nullptr;
    if (!h) return false;
    g_enqueue = reinterpret_cast<EnqueueFn>(dlsym(h, "_Z14hostemuEnqueueP12ihipStream_tPFvPvEPKvm"));
    return g_enqueue != nullptr;
}
std::string baseDir()
{
    const char* d = getenv("FAKE_RCCL_DIR");
    return d ? d : "/tmp/fake_rccl";
}
void nap() { struct timespec ts = {0, 2000000}; nanosleep(&ts, nullptr); }
bool waitFor(const std::string& path, size_t bytes, std::vector<char>& out)
{
    for (int tries = 0; tries < 60000; ++tries) { // two minutes
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            out.resize(bytes);
            const size_t got = fread(out.data(), 1, bytes, f);
            fclose(f);
            if (got == bytes) return true;
        }
        nap();
    }
    return false;
}
struct GatherJob {
    const void* send;
    void* recv;
    size_t bytes;
    int root;
    ncclComm* comm;
    unsigned long long seq;
};
void runGather(void* p)
{
    const GatherJob& J = *static_cast<const GatherJob*>(p);
    char name[64];
    snprintf(name, sizeof(name), "/g%llu_r%d", J.seq, J.comm->rank);
    const std::string mine = J.comm->dir + name, tmp = mine + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(J.send, 1, J.bytes, f) != J.bytes) { fprintf(stderr, "fake rccl: cannot write %s\n", tmp.c_str()); abort(); }
    fclose(f);
    rename(tmp.c_str(), mine.c_str());
    if (J.comm->rank != J.root) return;
    for (int r = 0; r < J.comm->nRanks; ++r) {
        snprintf(name, sizeof(name), "/g%llu_r%d", J.seq, r);
        std::vector<char> buf;
        if (!waitFor(J.comm->dir + name, J.bytes, buf)) { fprintf(stderr, "fake rccl: rank %d never sent gather %llu\n", r, J.seq); abort(); }
        memcpy(static_cast<char*>(J.recv) + (size_t)r * J.bytes, buf.data(), J.bytes);
        unlink((J.comm->dir + name).c_str());
    }
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "comm_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!findEnqueue()) return ncclSystemError;
    id.internal[sizeof(id.internal) - 1] = 0;
    ncclComm* c = new ncclComm;
    c->nRanks = nranks; c->rank = rank; c->seq = 0;
    c->dir = baseDir() + "/" + id.internal;
    mkdir(baseDir().c_str(), 0777);
    mkdir(c->dir.c_str(), 0777);
    // rendezvous: every rank announces itself and waits for the others (ncclCommInitRank is collective)
    char name[64];
    snprintf(name, sizeof(name), "/hello_%d", rank);
    FILE* f = fopen((c->dir + name).c_str(), "wb");
    if (!f) return ncclSystemError;
    fputc('x', f);
    fclose(f);
    for (int r = 0; r < nranks; ++r) {
        snprintf(name, sizeof(name), "/hello_%d", r);
        std::vector<char> b;
        if (!waitFor(c->dir + name, 1, b)) return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = comm->nRanks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { *rank = comm->rank; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }
ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclFloat32) return ncclInternalError;
    GatherJob J = {sendbuff, recvbuff, sendcount * 4, root, comm, comm->seq++};
    g_enqueue(stream, runGather, &J, sizeof(J));
    return ncclSuccess;
}
}
