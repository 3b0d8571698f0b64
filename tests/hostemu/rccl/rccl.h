// TEST INFRASTRUCTURE (tests/hostemu): the types and prototypes of rccl.h that csrc/tpt_host.cpp names (it dlopen()s the real library;
// the host-logic tests use the loopback communicator, which needs none of it).
#pragma once
#include <hip/hip_runtime.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank);
const char* ncclGetErrorString(ncclResult_t result);
}
