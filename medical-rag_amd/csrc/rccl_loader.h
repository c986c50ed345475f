// RCCL resolved at run time (dlopen), shared by the two exchange forms of the
// row-sharded search: one process per GPU (comm_rccl.hip) and one process
// driving several GPUs (multi_index.hip).  A single-GPU host needs no RCCL
// installation; when PyTorch already has its RCCL mapped, that copy is used;
// MTR_RCCL_LIBRARY names a particular build by path.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace mtr {

struct Rccl {
    void* handle = nullptr;   // NULL: no usable library, `why` says what went wrong
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // Several communicators in one process; may be NULL in a library that lacks them.
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    char why[256] = "";
};

const Rccl* rccl();   // defined in comm_rccl.hip

}  // namespace mtr
