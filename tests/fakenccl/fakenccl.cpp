// TEST DOUBLE for librccl - NOT product code, never loaded unless PCG_RCCL_LIB names it.
//
// The engine's native communicator (csrc/rccl_comm.hip) calls ncclSend / ncclRecv / ncclAllReduce / ncclGroup*.  Real
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the test box has ONE GPU.  This library
// implements the same entry points between ranks that SHARE a device - threads of one process or separate processes -
// through a file-backed shared mapping, so that the engine's own code path (event fences, comm stream, per-neighbour
// offsets and counts, all-reduce in place on the status block, look-ahead loop) runs on the real GPU with 2..8 parts.
// Semantics kept: point-to-point messages are matched per (source, destination) pair in issue order; a group's sends are
// all posted before its receives are waited for; the all-reduce sums the ranks' values in rank order (deterministic).
// Everything is synchronous (stream synchronise, host copy): correctness, not speed.
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
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kPairBytes = 8u << 20;             // capacity of one (source, destination) mailbox
constexpr int kRedMax = 2048;                       // doubles per all-reduce (the mailbox bootstrap exchanges 72 per rank)

struct Mailbox {
    std::atomic<uint64_t> full;                     // sequence number of the message in `data` (0 = empty)
    std::atomic<uint64_t> consumed;                 // sequence number of the last message taken out
    uint64_t bytes;
};
struct Shared {
    std::atomic<int> init_count;
    std::atomic<int> bar_count;
    std::atomic<int> bar_sense;
    std::atomic<int> alive;
    double red[kMaxRanks][kRedMax];
    Mailbox box[kMaxRanks][kMaxRanks];
    // followed by kMaxRanks * kMaxRanks * kPairBytes of message data
};

struct FakeComm {
    int rank = 0, n = 1;
    Shared *sh = nullptr;
    char *data = nullptr;
    size_t map_bytes = 0;
    std::string path;
    int sense = 0;
    uint64_t sent[kMaxRanks] = {}, recvd[kMaxRanks] = {};
    char *pair(int src, int dst) { return data + ((size_t)src * kMaxRanks + dst) * kPairBytes; }
};

struct Op { bool send; void *buf; size_t bytes; int peer; FakeComm *c; hipStream_t s; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

void spin() { std::this_thread::sleep_for(std::chrono::microseconds(20)); }

bool wait_until(const std::function<bool()> &f, double timeout_s = 120.0)
{
    auto t0 = std::chrono::steady_clock::now();
    while (!f()) {
        spin();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    }
    return true;
}

ncclResult_t barrier(FakeComm *c)
{
    Shared *s = c->sh;
    const int my = c->sense ^ 1;
    c->sense = my;
    if (s->bar_count.fetch_add(1) + 1 == c->n) {
        s->bar_count.store(0);
        s->bar_sense.store(my);
    } else if (!wait_until([&] { return s->bar_sense.load() == my; }))
        return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    // stream order: everything queued before the group is complete before data is touched
    for (auto &o : ops) if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    for (auto &o : ops) {                            // post every send first ...
        if (!o.send) continue;
        FakeComm *c = o.c;
        if (o.bytes > kPairBytes) return ncclInvalidArgument;
        Mailbox &b = c->sh->box[c->rank][o.peer];
        const uint64_t seq = ++c->sent[o.peer];
        if (!wait_until([&] { return b.consumed.load() == seq - 1; })) return ncclSystemError;
        if (hipMemcpy(c->pair(c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        b.bytes = o.bytes;
        b.full.store(seq);
    }
    for (auto &o : ops) {                            // ... then take the receives
        if (o.send) continue;
        FakeComm *c = o.c;
        Mailbox &b = c->sh->box[o.peer][c->rank];
        const uint64_t seq = ++c->recvd[o.peer];
        if (!wait_until([&] { return b.full.load() == seq; })) return ncclSystemError;
        if (b.bytes != o.bytes) return ncclInvalidArgument;
        if (hipMemcpy(o.buf, c->pair(o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        b.consumed.store(seq);
    }
    ops.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof(*id));
    unsigned long long r[2] = {0, 0};
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (f) { (void)!std::fread(r, sizeof(r), 1, f); std::fclose(f); }
    std::snprintf(id->internal, sizeof(id->internal), "fakenccl_%016llx%016llx_%d", r[0], r[1], (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    auto *c = new FakeComm();
    c->rank = rank; c->n = nranks;
    const char *tmp = std::getenv("TMPDIR");
    id.internal[sizeof(id.internal) - 1] = 0;
    c->path = std::string(tmp && *tmp ? tmp : "/tmp") + "/" + id.internal;
    c->map_bytes = sizeof(Shared) + (size_t)kMaxRanks * kMaxRanks * kPairBytes;
    int fd = open(c->path.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }     // sparse: zero pages
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared *)p;
    c->data = (char *)p + sizeof(Shared);
    c->sh->alive.fetch_add(1);
    c->sh->init_count.fetch_add(1);
    if (!wait_until([&] { return c->sh->init_count.load() >= nranks; })) { delete c; return ncclSystemError; }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    auto *c = (FakeComm *)comm;
    if (!c) return ncclSuccess;
    const bool last = c->sh->alive.fetch_sub(1) == 1;
    munmap((void *)c->sh, c->map_bytes);
    if (last) unlink(c->path.c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return run_ops(g_ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    auto *c = (FakeComm *)comm;
    if (dt != ncclDouble || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{true, const_cast<void *>(buf), count * sizeof(double), peer, c, s});
    return g_depth > 0 ? ncclSuccess : run_ops(g_ops);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    auto *c = (FakeComm *)comm;
    if (dt != ncclDouble || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * sizeof(double), peer, c, s});
    return g_depth > 0 ? ncclSuccess : run_ops(g_ops);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s)
{
    auto *c = (FakeComm *)comm;
    if (dt != ncclDouble || op != ncclSum || count > (size_t)kRedMax) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    double v[kRedMax];
    if (hipMemcpy(v, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    std::memcpy(c->sh->red[c->rank], v, count * sizeof(double));
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    for (size_t k = 0; k < count; ++k) {
        double t = c->sh->red[0][k];
        for (int q = 1; q < c->n; ++q) t = t + c->sh->red[q][k];       // rank order, like the oracle's _allreduce
        v[k] = t;
    }
    r = barrier(c);                                                     // nobody overwrites red[] before all have read it
    if (r != ncclSuccess) return r;
    if (hipMemcpy(recv, v, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success";
    case ncclUnhandledCudaError: return "fakenccl: HIP error";
    case ncclSystemError: return "fakenccl: timeout / system error";
    case ncclInvalidArgument: return "fakenccl: invalid argument";
    case ncclInvalidUsage: return "fakenccl: invalid usage";
    default: return "fakenccl: error";
    }
}

}  // extern "C"
