// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): streams as FIFO queues of closures, events, "device" memory.
//
// Execution policy (env HOSTEMU_POLICY, read when the first stream is created):
//   eager          every operation runs when it is enqueued (its dependencies first)
//   lazy           nothing runs until the host waits for it (hipStreamSynchronize, hipEventSynchronize, a synchronous copy ...), and
//                  then only what that wait needs -- the LATEST legal schedule.  An event query that finds its event pending lets
//                  the "device" make one operation of progress somewhere, so polling loops of the host terminate.
//   random:<seed>  after every enqueue a seeded coin decides how many queued operations of which streams run
// Memory handed back with hipFree is filled with 0xFF bytes (NaNs, huge counters) and kept in quarantine: work that was still
// queued on it computes garbage instead of crashing, and the parity check of the test sees it.
// Blocking streams (hipStreamDefault) and the legacy null stream order against each other as HIP specifies: an operation on the
// null stream -- the synchronous hipMemcpy / hipMemset -- first drains every blocking stream; work enqueued on a blocking stream
// does not start before what the null stream was given earlier (there is never anything pending there: it runs synchronously).
#include <hip/hip_runtime.h>
#include <deque>
#include <functional>
#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

struct ihipEvent_t {
    uint64_t recorded = 0, completed = 0; // sequence number of the latest record enqueued / executed
    ihipStream_t* where = nullptr;        // the stream that holds the latest record
    double t = 0;                         // virtual time of the latest executed record
};
struct Op {
    enum Kind { WORK, RECORD, WAIT } kind;
    std::function<void()> fn;
    ihipEvent_t* ev = nullptr;
    uint64_t seq = 0;
};
struct ihipStream_t {
    std::deque<Op> q;
    unsigned flags = 0;
    bool running = false;
    int id = 0;
};

namespace {
enum Policy { EAGER, LAZY, RANDOM };
Policy g_policy = EAGER;
bool g_policyRead = false;
uint64_t g_rng = 1, g_seq = 0;
double g_clock = 0;
std::vector<ihipStream_t*> g_streams;
std::map<void*, size_t> g_live;
long long g_opsRun = 0, g_frees = 0;

void readPolicy()
{
    if (g_policyRead) return;
    g_policyRead = true;
    const char* e = getenv("HOSTEMU_POLICY");
    if (!e || !strcmp(e, "eager")) g_policy = EAGER;
    else if (!strcmp(e, "lazy")) g_policy = LAZY;
    else if (!strncmp(e, "random", 6)) {
        g_policy = RANDOM;
        g_rng = e[6] == ':' ? strtoull(e + 7, nullptr, 10) * 2654435761ull + 1 : 12345;
    }
}
uint64_t rnd()
{
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return g_rng;
}
[[noreturn]] void die(const char* what)
{
    fprintf(stderr, "hostemu: %s\n", what);
    abort();
}

bool runOne(ihipStream_t* s); // executes the operation at the head of s (driving what it waits for); false: nothing queued
void drain(ihipStream_t* s)
{
    while (runOne(s)) {
    }
}
// runs stream `s` until the record `seq` of event `e` has executed
void driveTo(ihipEvent_t* e, uint64_t seq)
{
    while (e->completed < seq) {
        ihipStream_t* s = e->where;
        if (!s) die("wait on an event whose record was never enqueued");
        if (s->running) die("deadlock: a stream waits (through events) for work queued behind that very wait");
        if (!runOne(s)) die("an event's record is not in the stream it was recorded on");
    }
}
bool runOne(ihipStream_t* s)
{
    if (s->q.empty()) return false;
    if (s->running) die("re-entrant execution of one stream");
    s->running = true;
    Op op = std::move(s->q.front());
    if (op.kind == Op::WAIT) {
        // (the wait stays at the head of the queue while what it waits for is driven)
        s->q.front().kind = Op::WAIT;
        driveTo(op.ev, op.seq);
        s->q.pop_front();
    } else {
        s->q.pop_front();
        if (op.kind == Op::RECORD) {
            if (op.seq > op.ev->completed) op.ev->completed = op.seq;
            op.ev->t = g_clock;
        } else {
            op.fn();
            g_clock += 1.0;
        }
    }
    g_opsRun++;
    s->running = false;
    return true;
}
void afterEnqueue(ihipStream_t* s)
{
    if (g_policy == EAGER) drain(s);
    else if (g_policy == RANDOM) {
        int n = (int)(rnd() % 4); // 0..3 operations of random streams
        for (int i = 0; i < n && !g_streams.empty(); ++i) {
            ihipStream_t* t = g_streams[rnd() % g_streams.size()];
            if (t && !t->running) runOne(t);
        }
    }
}
void progressSomewhere()
{
    static size_t next = 0;
    for (size_t i = 0; i < g_streams.size(); ++i) {
        ihipStream_t* t = g_streams[(next + i) % g_streams.size()];
        if (t && !t->running && !t->q.empty()) {
            next = (next + i + 1) % g_streams.size();
            runOne(t);
            return;
        }
    }
}
ihipStream_t g_null; // the legacy default stream: nothing ever stays queued on it
ihipStream_t* real(hipStream_t s) { return s ? s : &g_null; }
void drainBlockingStreams()
{
    for (ihipStream_t* t : g_streams)
        if (t && !(t->flags & hipStreamNonBlocking)) drain(t);
}
void push(ihipStream_t* s, Op&& op)
{
    if (s == &g_null) { // synchronous with respect to the host and to every blocking stream
        drainBlockingStreams();
        s->q.push_back(std::move(op));
        drain(s);
        return;
    }
    s->q.push_back(std::move(op));
    afterEnqueue(s);
}
} // namespace

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorNotReady ? "hipErrorNotReady" : "hip error (hostemu)"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = getenv("HOSTEMU_NO_DEVICE") ? 0 : 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "host emulation (tests/hostemu)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "none");
    p->multiProcessorCount = getenv("HOSTEMU_CUS") ? atoi(getenv("HOSTEMU_CUS")) : 256;
    p->totalGlobalMem = 64ull << 30;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* freeB, size_t* totalB) { *freeB = 60ull << 30; *totalB = 64ull << 30; return hipSuccess; }
hipError_t hipDeviceSynchronize()
{
    for (size_t i = 0; i < g_streams.size(); ++i)
        if (g_streams[i]) drain(g_streams[i]);
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n)
{
    const size_t bytes = (n + 255) & ~(size_t)255;
    void* m = aligned_alloc(256, bytes ? bytes : 256);
    if (!m) return hipErrorOutOfMemory;
    memset(m, 0xA5, bytes ? bytes : 256); // fresh device memory holds garbage
    g_live[m] = bytes ? bytes : 256;
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    auto it = g_live.find(p);
    if (it == g_live.end()) die("hipFree of a pointer hipMalloc did not return (or a double free)");
    memset(p, 0xFF, it->second); // poisoned, and kept: queued work that still uses it computes garbage the tests see
    g_live.erase(it);
    g_frees++;
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s)
{
    Op op; op.kind = Op::WORK;
    op.fn = [dst, src, n] { memmove(dst, src, n); };
    push(real(s), std::move(op));
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(dst, src, n, k, nullptr); }
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s)
{
    Op op; op.kind = Op::WORK;
    op.fn = [dst, v, n] { memset(dst, v, n); };
    push(real(s), std::move(op));
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t n) { return hipMemsetAsync(dst, v, n, nullptr); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags)
{
    readPolicy();
    ihipStream_t* t = new ihipStream_t;
    t->flags = flags;
    t->id = (int)g_streams.size();
    g_streams.push_back(t);
    *s = t;
    return hipSuccess;
}
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, hipStreamDefault); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return hipErrorInvalidValue;
    drain(s); // (HIP lets the queued work finish)
    for (auto& t : g_streams)
        if (t == s) t = nullptr;
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (!s) drainBlockingStreams();
    else drain(s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    if (!e) return hipErrorInvalidValue;
    if (e->completed >= e->recorded) return hipSuccess; // never recorded, or its latest record has executed
    Op op; op.kind = Op::WAIT; op.ev = e; op.seq = e->recorded;
    push(real(s), std::move(op));
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t; return hipSuccess; } // (never freed: queued waits may name it)
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (!e) return hipErrorInvalidValue;
    Op op; op.kind = Op::RECORD; op.ev = e; op.seq = ++g_seq;
    e->recorded = op.seq;
    e->where = real(s);
    push(real(s), std::move(op));
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e)
{
    if (e->completed >= e->recorded) return hipSuccess;
    if (g_policy != EAGER) progressSomewhere(); // time passes on the device while the host polls
    return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    driveTo(e, e->recorded);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    const double d = (b->t - a->t) * 1e-3;
    *ms = (float)(d > 1e-3 ? d : 1e-3);
    return hipSuccess;
}

void hostemuEnqueue(hipStream_t s, void (*fn)(void*), const void* arg, size_t bytes)
{
    std::vector<unsigned char> copy((const unsigned char*)arg, (const unsigned char*)arg + bytes);
    Op op; op.kind = Op::WORK;
    op.fn = [fn, copy]() mutable { fn(copy.data()); };
    push(real(s), std::move(op));
}

// statistics for the tests: operations executed, hipFree calls
extern "C" void hostemu_stats(long long* out2)
{
    out2[0] = g_opsRun;
    out2[1] = g_frees;
}
