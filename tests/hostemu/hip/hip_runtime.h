// TEST INFRASTRUCTURE (tests/hostemu): a host-only stand-in for the handful of HIP runtime calls csrc/tpt_host.cpp makes, so that the
// library's HOST logic -- slot ring, tickets, look-ahead, stream / row-serial batches, scene-set ring, sharded exchange, buffer growth
// and shrink -- runs in the CPU test suite against the oracle.  "Device memory" is host memory, streams are FIFO queues of closures,
// events order them; tests/hostemu/hip_shim.cpp executes the queues eagerly (at enqueue), lazily (only when something waits: the
// latest legal moment, so a buffer freed or overwritten while work that uses it is still queued shows up as wrong pixels) or in a
// seeded random order.  The kernels' launch functions (csrc/tpt_device.h) are restated on the lane headers in hostemu_kernels.cpp.
// Nothing of this is part of the product: toypathtracer_amd/ never loads it, and the product library has no CPU path.
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
struct ihipStream_t;
struct ihipEvent_t;
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2, hipEventDisableSystemFence = 0x20000000 };
enum { hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMemGetInfo(size_t* freeB, size_t* totalB);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void* dst, int v, size_t n);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t words, const uint32_t* mask);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

// for hostemu_kernels.cpp: run `fn(arg)` as one unit of work of stream s (arg is copied: `bytes` of it)
void hostemuEnqueue(hipStream_t s, void (*fn)(void*), const void* arg, size_t bytes);
