// TEST DOUBLE for RCCL, for one purpose: to run the library's exchange code
// (csrc/comm_rccl.hip: block layout, buffer rotation, merge over the gathered
// blocks, ranks without rows) with MORE THAN ONE RANK on a box that has a single
// GPU, where the real RCCL refuses to put two ranks on one device.  It says
// nothing about RCCL itself.  Never loaded unless MTR_RCCL_LIBRARY names it.
//
// It implements the entry points the library resolves.  For ranks that are
// processes of one host (ncclCommInitRank): an all-gather is a stream
// synchronisation, a copy of the send block to a shared-memory segment, a
// barrier, and copies of every rank's block back to the device -- a legal
// (blocking, slow) implementation of the same contract.  For the ranks of ONE
// process (ncclCommInitAll, any device list, repeats included): all-gathers
// issued between ncclGroupStart and ncclGroupEnd are carried out at
// ncclGroupEnd, once every rank of the communicator has issued its own, by
// device copies behind a synchronisation of the ranks' streams.
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
#include <vector>

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
    int local_group = 0;   // > 0: a rank of a communicator made by ncclCommInitAll (all ranks in this process)
};

// All-gathers of in-process ranks waiting for ncclGroupEnd.
struct Pending {
    Double* d;
    const void* send;
    void* recv;
    size_t bytes;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;
std::atomic<int> g_next_group{1};

// Every rank of a group has issued its all-gather: synchronise their streams,
// then copy every send block into every receive buffer.
ncclResult_t run_local(std::vector<Pending>& ops) {
    for (auto& op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (auto& dst : ops)
        for (auto& src : ops) {
            if (src.d->local_group != dst.d->local_group) continue;
            if (src.bytes != dst.bytes) return ncclInvalidArgument;
            if (hipMemcpy(static_cast<char*>(dst.recv) + src.bytes * (size_t)src.d->rank, src.send, src.bytes,
                          hipMemcpyDefault) != hipSuccess)
                return ncclUnhandledCudaError;
        }
    return ncclSuccess;
}

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

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    (void)devlist;   // copies between the ranks' buffers go by unified addressing, whatever device holds them
    const int group = g_next_group.fetch_add(1);
    for (int r = 0; r < ndev; ++r) {
        Double* d = new Double;
        d->rank = r;
        d->world = ndev;
        d->local_group = group;
        comms[r] = reinterpret_cast<ncclComm_t>(d);
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth < 1) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(g_pending);
    // a group must hold the all-gather of every rank of each communicator it touches
    for (auto& op : ops) {
        int members = 0;
        for (auto& other : ops) members += other.d->local_group == op.d->local_group;
        if (members != op.d->world) return ncclInvalidUsage;
    }
    return run_local(ops);
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Double* d = reinterpret_cast<Double*>(comm);
    if (!d) return ncclSuccess;
    if (!d->local_group) munmap(d->header, d->bytes);
    delete d;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    Double* d = reinterpret_cast<Double*>(comm);
    const size_t bytes = sendcount * type_size(datatype);
    if (!d || !sendbuff || !recvbuff || bytes == 0) return ncclInvalidArgument;
    if (d->local_group) {
        // In-process ranks: inside a group the call is only noted; outside one, a lone rank can
        // proceed and several would wait for each other for ever (as with the real library).
        if (g_depth > 0) {
            g_pending.push_back(Pending{d, sendbuff, recvbuff, bytes, stream});
            return ncclSuccess;
        }
        if (d->world != 1) return ncclInvalidUsage;
        std::vector<Pending> one{Pending{d, sendbuff, recvbuff, bytes, stream}};
        return run_local(one);
    }
    if (bytes > kSlotBytes) return ncclInvalidArgument;
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
