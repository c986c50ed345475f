// TEST DOUBLE for RCCL, for one purpose: to run the library's exchange code
// (csrc/comm_rccl.hip: block layout, buffer rotation, merge over the gathered
// blocks, ranks without rows) with MORE THAN ONE RANK on a box that has a single
// GPU, where the real RCCL refuses to put two ranks on one device.  It says
// nothing about RCCL itself.  Never loaded unless MTR_RCCL_LIBRARY names it.
//
// It implements the five entry points the library resolves, for ranks that are
// processes of one host: an all-gather is a stream synchronisation, a copy of
// the send block to a shared-memory segment, a barrier, and copies of every
// rank's block back to the device -- a legal (blocking, slow) implementation
// of the same contract.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

namespace {

constexpr size_t kSlotBytes = 8u << 20;   // per rank and collective
constexpr int kTimeoutSeconds = 60;

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};

struct Double {
    int rank = 0, world = 1;
    char name[64] = "";
    Header* header = nullptr;
    char* slots = nullptr;
    size_t bytes = 0;
};

bool barrier(Double* d) {
    const int gen = d->header->generation.load();
    if (d->header->arrived.fetch_add(1) + 1 == d->world) {
        d->header->arrived.store(0);
        d->header->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (d->header->generation.load() == gen) {
        usleep(20);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kTimeoutSeconds)) return false;
    }
    return true;
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    std::random_device rd;
    snprintf(id->internal, sizeof(id->internal), "/mtr_double_%08x%08x_%d", rd(), rd(), (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    Double* d = new Double;
    d->rank = rank;
    d->world = nranks;
    snprintf(d->name, sizeof(d->name), "%s", id.internal);
    d->bytes = 4096 + kSlotBytes * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(d->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)d->bytes) != 0) { delete d; return ncclSystemError; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(d->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == d->bytes) break;
            if (fd >= 0) close(fd);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kTimeoutSeconds)) { delete d; return ncclSystemError; }
            usleep(1000);
        }
    }
    void* p = mmap(nullptr, d->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete d; return ncclSystemError; }
    d->header = static_cast<Header*>(p);      // a fresh segment is all zeros: valid initial atomics
    d->slots = static_cast<char*>(p) + 4096;
    d->header->attached.fetch_add(1);
    if (!barrier(d)) { munmap(p, d->bytes); delete d; return ncclSystemError; }
    if (rank == 0) shm_unlink(d->name);        // everyone has it mapped; the name can go
    *comm = reinterpret_cast<ncclComm_t>(d);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Double* d = reinterpret_cast<Double*>(comm);
    if (!d) return ncclSuccess;
    munmap(d->header, d->bytes);
    delete d;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    Double* d = reinterpret_cast<Double*>(comm);
    const size_t bytes = sendcount * type_size(datatype);
    if (!d || !sendbuff || !recvbuff || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(d->slots + kSlotBytes * (size_t)d->rank, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    if (!barrier(d)) return ncclSystemError;           // every block is in the segment
    for (int r = 0; r < d->world; ++r)
        if (hipMemcpyAsync(static_cast<char*>(recvbuff) + bytes * (size_t)r, d->slots + kSlotBytes * (size_t)r, bytes,
                           hipMemcpyHostToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(d)) return ncclSystemError;           // every rank has read: the slots may be overwritten
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument (test double)";
        case ncclSystemError: return "system error or timeout (test double)";
        case ncclUnhandledCudaError: return "HIP error (test double)";
        default: return "error (test double)";
    }
}

}  // extern "C"
