// RCCL communicator of the engine: the reference's mpi4py calls on the hot path, issued natively.
//
//   reference (src/solver/pcg_solver.py)                      here
//   --------------------------------------------------------  ------------------------------------------------
//   Comm.Isend(buf_j, dest=nbr_j) / Comm.Recv / Waitall        ONE ncclGroupStart .. ncclSend/ncclRecv x neighbours ..
//   (:318-328), one message per neighbour per mat-vec          ncclGroupEnd on a dedicated communication stream;
//                                                              every GPU pair of an MI355X node has its own xGMI link,
//                                                              so the (up to 7) neighbour messages travel concurrently
//   MPI_SUM -> Comm.allreduce (:622-628)                       ncclAllReduce(ncclDouble, ncclSum) in place on the device
//                                                              status block, on the compute stream
//
// Stream choreography of one operator apply (pcg_driver.cpp apply()):
//   compute:  interface rows -> pack -> [ev_packed] ............ interior rows -> wait [ev_done] -> fix-up
//   comm   :                      wait [ev_packed] -> group(send/recv) -> [ev_done]
// Two ncclComm_t are used - one only ever sees the comm stream, the other only the compute stream - so RCCL
// never has to serialise one communicator across two user streams; the two are never in flight together
// (the all-reduce follows the fix-up, the next exchange follows the all-reduce through ev_packed), on
// every rank in the same order.
//
// RCCL is loaded with dlopen("librccl.so.1"): inside a Python process that imported torch this is the RCCL
// torch already mapped (same HIP runtime instance), in a plain C process /opt/rocm's.  PCG_RCCL_LIB names
// another library with the same entry points (the in-process / shared-GPU test double under tests/fakenccl).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcg_internal.hpp"

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw std::runtime_error(std::string(#expr) + " -> " + hipGetErrorString(_e));               \
    } while (0)

namespace pcg {
namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &api()
{
    static RcclApi a;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [&]() {
        std::vector<std::string> cand;
        if (const char *e = std::getenv("PCG_RCCL_LIB")) cand.push_back(e);
        else cand = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const auto &c : cand) {
            a.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (a.handle) { a.path = c; break; }
            err += std::string(c) + ": " + dlerror() + "; ";
        }
        if (!a.handle) return;
        auto sym = [&](const char *n) {
            void *p = dlsym(a.handle, n);
            if (!p) { err += std::string("missing symbol ") + n + "; "; }
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        if (!err.empty() && a.handle && !(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd &&
                                         a.Send && a.Recv && a.AllReduce && a.GetErrorString))
            a.handle = nullptr;
    });
    if (!a.handle) throw std::runtime_error("RCCL not available (" + err + ")");
    return a;
}

void nccl_check(ncclResult_t r, const char *what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + " -> " + api().GetErrorString(r));
}
#define NCCL_CHECK(expr) nccl_check((expr), #expr)

// Pairs of timing events recycled through a ring; a pair is read back (hipEventElapsedTime) when its slot comes round
// again - hundreds of iterations later, long complete - or in drain().
struct EventRing {
    static constexpr int kSlots = 512;
    std::vector<hipEvent_t> a, b;
    std::vector<char> used;
    int next = 0;
    double ms = 0;
    int64_t n = 0;
    void init()
    {
        if (!a.empty()) return;
        a.resize(kSlots); b.resize(kSlots); used.assign(kSlots, 0);
        for (int k = 0; k < kSlots; ++k) { HIP_CHECK(hipEventCreate(&a[k])); HIP_CHECK(hipEventCreate(&b[k])); }
    }
    void harvest(int k)
    {
        if (!used[k]) return;
        HIP_CHECK(hipEventSynchronize(b[k]));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, a[k], b[k]));
        ms += t; n += 1; used[k] = 0;
    }
    int begin(hipStream_t s)
    {
        const int k = next;
        next = (next + 1) % kSlots;
        harvest(k);
        HIP_CHECK(hipEventRecord(a[k], s));
        return k;
    }
    void end(int k, hipStream_t s) { HIP_CHECK(hipEventRecord(b[k], s)); used[k] = 1; }
    void drain() { for (int k = 0; k < (int)a.size(); ++k) harvest(k); }
    void destroy() { for (auto e : a) (void)hipEventDestroy(e); for (auto e : b) (void)hipEventDestroy(e); a.clear(); b.clear(); }
};

class RcclComm : public Comm {
    int dev_, rank_, size_;
    ncclComm_t halo_comm_ = nullptr, red_comm_ = nullptr;
    hipStream_t comm_stream_ = nullptr;
    static constexpr int kFence = 8;                 // fence events: a ring, so a re-record never races a pending wait
    hipEvent_t ev_packed_[kFence] = {}, ev_done_[kFence] = {};
    int fence_ = 0, open_fence_ = -1;
    bool timing_ = false;
    // PCG_RCCL_ALLOW_SELF=1 (tests): a part may list ITSELF as a neighbour, so that a one-GPU box drives the real
    // ncclSend / ncclRecv group, the comm stream and the fences on librccl (RCCL matches a send to self with the recv of the group)
    bool allow_self_ = std::getenv("PCG_RCCL_ALLOW_SELF") != nullptr;
    EventRing t_halo_, t_red_;
    CommStats st_;

public:
    RcclComm(int device, int rank, int nranks, const void *ids) : dev_(device), rank_(rank), size_(nranks)
    {
        if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("rccl comm: bad rank / size");
        RcclApi &A = api();
        HIP_CHECK(hipSetDevice(dev_));
        ncclUniqueId id[2];
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(id, ids, sizeof(id));
        // every rank creates the two communicators in the same order
        NCCL_CHECK(A.CommInitRank(&halo_comm_, nranks, id[0], rank));
        NCCL_CHECK(A.CommInitRank(&red_comm_, nranks, id[1], rank));
        // highest priority: the send / recv kernel of an exchange is queued while the interior rows' workgroups occupy the CUs - it
        // must be dispatched ahead of their remaining workgroups, not after them (that is the overlap)
        int prio_lo = 0, prio_hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_CHECK(hipStreamCreateWithPriority(&comm_stream_, hipStreamNonBlocking, prio_hi));
        for (int k = 0; k < kFence; ++k) {
            HIP_CHECK(hipEventCreateWithFlags(&ev_packed_[k], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ev_done_[k], hipEventDisableTiming));
        }
    }
    ~RcclComm() override
    {
        (void)hipSetDevice(dev_);
        if (comm_stream_) (void)hipStreamSynchronize(comm_stream_);
        t_halo_.destroy(); t_red_.destroy();
        try {
            RcclApi &A = api();
            if (halo_comm_) (void)A.CommDestroy(halo_comm_);
            if (red_comm_) (void)A.CommDestroy(red_comm_);
        } catch (...) {
        }
        for (int k = 0; k < kFence; ++k) {
            if (ev_packed_[k]) (void)hipEventDestroy(ev_packed_[k]);
            if (ev_done_[k]) (void)hipEventDestroy(ev_done_[k]);
        }
        if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
    }
    int rank() const override { return rank_; }
    int size() const override { return size_; }

    void halo_begin(double *send, double *recv, const HaloHost &h, void *compute_stream) override
    {
        RcclApi &A = api();
        hipStream_t cs = (hipStream_t)compute_stream;
        const int f = fence_;
        fence_ = (fence_ + 1) % kFence;
        HIP_CHECK(hipEventRecord(ev_packed_[f], cs));                       // send buffer is packed (and the previous fix-up
        HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_packed_[f], 0));      //  has finished reading recv)
        NCCL_CHECK(A.GroupStart());                                         // :318-326, every neighbour at once
        for (int j = 0; j < h.n_peers; ++j) {
            const int64_t off = h.send_ptr[j], cnt = h.send_ptr[j + 1] - off;
            if (cnt <= 0) continue;
            const int peer = h.peer_ids[j];
            if (peer < 0 || peer >= size_ || (peer == rank_ && !allow_self_)) {
                (void)A.GroupEnd();
                throw std::runtime_error("rccl comm: neighbour part id is not a peer rank (one part per rank, pcg_solver.py:91)");
            }
            NCCL_CHECK(A.Send(send + off, (size_t)cnt, ncclDouble, peer, halo_comm_, comm_stream_));
            NCCL_CHECK(A.Recv(recv + off, (size_t)cnt, ncclDouble, peer, halo_comm_, comm_stream_));
        }
        NCCL_CHECK(A.GroupEnd());
        HIP_CHECK(hipEventRecord(ev_done_[f], comm_stream_));               // :328 Waitall
        open_fence_ = f;
        st_.n_halo++;
    }
    void halo_end(void *compute_stream) override
    {
        if (open_fence_ < 0) throw std::runtime_error("rccl comm: halo_end without halo_begin");
        hipStream_t cs = (hipStream_t)compute_stream;
        int k = -1;
        if (timing_) k = t_halo_.begin(cs);                                 // compute stream idle from here ...
        HIP_CHECK(hipStreamWaitEvent(cs, ev_done_[open_fence_], 0));
        if (timing_) t_halo_.end(k, cs);                                    // ... to here = time blocked in the exchange
        open_fence_ = -1;
    }
    void allreduce(double *buf, int count, void *compute_stream) override
    {
        RcclApi &A = api();
        hipStream_t cs = (hipStream_t)compute_stream;
        int k = -1;
        if (timing_) k = t_red_.begin(cs);
        NCCL_CHECK(A.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, red_comm_, cs));   // :625
        if (timing_) t_red_.end(k, cs);
        st_.n_allreduce++;
    }
    void set_timing(bool on) override
    {
        HIP_CHECK(hipSetDevice(dev_));               // the timing events belong to this communicator's device
        if (on) { t_halo_.init(); t_red_.init(); }
        timing_ = on;
    }
    CommStats stats() override
    {
        HIP_CHECK(hipSetDevice(dev_));
        if (!t_halo_.a.empty()) { t_halo_.drain(); t_red_.drain(); }
        st_.halo_wait_ms = t_halo_.ms; st_.n_halo_timed = t_halo_.n;
        st_.allreduce_ms = t_red_.ms; st_.n_allreduce_timed = t_red_.n;
        return st_;
    }
};

}  // namespace

std::unique_ptr<Comm> make_rccl_comm(int device, int rank, int nranks, const void *unique_ids)
{
    return std::unique_ptr<Comm>(new RcclComm(device, rank, nranks, unique_ids));
}

int rccl_unique_ids(void *out)
{
    RcclApi &A = api();
    ncclUniqueId id[2];
    NCCL_CHECK(A.GetUniqueId(&id[0]));
    NCCL_CHECK(A.GetUniqueId(&id[1]));
    std::memcpy(out, id, sizeof(id));
    return 0;
}

}  // namespace pcg
