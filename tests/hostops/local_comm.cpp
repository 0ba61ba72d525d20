// TEST DOUBLE - in-process stand-in for the engine's native communicator (csrc/rccl_comm.hip) on the CPU test double.
//
// Compiled ONLY into tests/hostops/_build/libpcg_hostops.so.  It gives the `-m "not gpu"` suite what tests/fakenccl gives
// the GPU suite: several parts in ONE process (one host thread per part) talking through pcg::Comm - the branch of
// pcg_driver.cpp that the product takes with RCCL (comm->halo_begin / halo_end / allreduce, no callbacks) - and with it the
// device-group API (csrc/group.cpp, pcg_group_*).  Semantics follow the reference's mpi4py calls:
//   Isend / Recv / Waitall per neighbour (pcg_solver.py:318-328): one mailbox per (source, destination) pair, FIFO;
//   MPI_SUM allreduce (:622-628): every rank deposits, all ranks add the deposits in rank order (same bits everywhere).
// "Device" memory of the test double is host memory and its kernels run at enqueue time, so everything here is synchronous.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_mail.hpp"
#include "pcg_internal.hpp"

namespace pcg {
namespace {

constexpr int kIdBytes = 256;
constexpr auto kTimeout = std::chrono::seconds(120);      // a lost peer fails the test instead of hanging it

struct World {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::vector<double>>> box;     // (src, dst) -> messages in flight
    std::vector<std::vector<double>> dep;                                   // all-reduce deposits by rank
    int arrived = 0, left = 0;
    int64_t gen = 0;
    int attached = 0;
    std::vector<double *> mail;                                             // every rank's mailbox (enable_mailbox)
    int mail_registered = 0;
    // direct exchange (direct_link): the k-th call of every rank forms generation k; a rank's record = its buffer + segment layout
    struct DirectRec { unsigned long long *buf = nullptr; size_t flags_word = 0; std::vector<int32_t> peers; std::vector<int64_t> off, cnt; };
    std::map<int64_t, std::vector<DirectRec>> direct;                       // generation -> records by rank
    std::map<int64_t, int> direct_registered, direct_done, direct_failed;   // per generation: records in, look-ups finished, look-ups failed
};

// One engine's side of the direct exchange on the CPU double (csrc/rccl_comm.hip RcclDirectLink): host memory, the same descriptor.
class LocalDirectLink : public DirectLink {
public:
    std::vector<unsigned long long> store;            // [receive buffer: total doubles][arrival words: kDirectMaxPeers]
    DirectDesc d{};
    unsigned long long seq = 0;
    unsigned err = 0;
    CommStats *stats = nullptr;
    double *recv() override { return reinterpret_cast<double *>(store.data()); }
    DirectDesc next() override { d.seq = ++seq; if (stats) stats->n_halo++; return d; }
    void check() override
    {
        if (err) { err = 0; throw std::runtime_error("local comm: a direct-exchange wait timed out"); }
    }
};

std::mutex g_reg_m;
std::map<std::string, std::shared_ptr<World>> g_reg;
std::atomic<uint64_t> g_next_id{1};

class LocalComm : public Comm {
    std::shared_ptr<World> w_;
    std::string key_;
    int rank_, size_;
    double *recv_ = nullptr;
    const HaloHost *halo_ = nullptr;
    CommStats st_;
    // mailbox all-reduce (pcg_internal.hpp MailDesc; the protocol of csrc/kernels_mail.hpp on host memory)
    std::vector<unsigned long long> box_;
    bool mail_on_ = false;
    unsigned long long mail_seq_ = 0;
    unsigned mail_err_ = 0;
    int64_t direct_calls_ = 0;

public:
    LocalComm(int rank, int nranks, const void *ids) : rank_(rank), size_(nranks)
    {
        if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("local comm: bad rank / size");
        key_.assign((const char *)ids, kIdBytes);
        std::lock_guard<std::mutex> lk(g_reg_m);
        auto &slot = g_reg[key_];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->n = nranks;
            slot->dep.resize(nranks);
        }
        if (slot->n != nranks) throw std::runtime_error("local comm: ranks disagree on the world size");
        slot->attached++;
        w_ = slot;
    }
    ~LocalComm() override
    {
        std::lock_guard<std::mutex> lk(g_reg_m);
        if (--w_->attached == 0) g_reg.erase(key_);
    }
    int rank() const override { return rank_; }
    int size() const override { return size_; }

    void halo_begin(double *send, double *recv, const HaloHost &h, void *) override        // :318-326
    {
        std::lock_guard<std::mutex> lk(w_->m);
        for (int j = 0; j < h.n_peers; ++j) {
            const int64_t off = h.send_ptr[j], cnt = h.send_ptr[j + 1] - off;
            if (cnt <= 0) continue;
            const int peer = h.peer_ids[j];
            if (peer < 0 || peer >= size_ || peer == rank_)
                throw std::runtime_error("local comm: neighbour part id is not a peer rank (one part per rank, pcg_solver.py:91)");
            w_->box[{rank_, peer}].emplace_back(send + off, send + off + cnt);
        }
        w_->cv.notify_all();
        recv_ = recv;
        halo_ = &h;
        st_.n_halo++;
    }
    void halo_end(void *) override                                                          // :328 Waitall
    {
        if (!halo_) throw std::runtime_error("local comm: halo_end without halo_begin");
        const HaloHost &h = *halo_;
        std::unique_lock<std::mutex> lk(w_->m);
        for (int j = 0; j < h.n_peers; ++j) {
            const int64_t off = h.send_ptr[j], cnt = h.send_ptr[j + 1] - off;
            if (cnt <= 0) continue;
            auto &q = w_->box[{h.peer_ids[j], rank_}];
            if (!w_->cv.wait_for(lk, kTimeout, [&] { return !q.empty(); }))
                throw std::runtime_error("local comm: no message from part " + std::to_string(h.peer_ids[j]));
            if ((int64_t)q.front().size() != cnt) throw std::runtime_error("local comm: message length mismatch");
            std::memcpy(recv_ + off, q.front().data(), sizeof(double) * (size_t)cnt);
            q.pop_front();
        }
        halo_ = nullptr;
    }
    bool enable_mailbox(bool on) override
    {
        if (!on) { mail_on_ = false; return false; }
        if (mail_on_) return true;
        if (size_ > kMailMaxRanks) return false;
        box_.assign((size_t)2 * kMailMaxRanks * kMailSlotWords, 0ull);
        std::unique_lock<std::mutex> lk(w_->m);
        w_->mail.resize(size_, nullptr);
        w_->mail[rank_] = reinterpret_cast<double *>(box_.data());
        w_->mail_registered++;
        w_->cv.notify_all();
        if (!w_->cv.wait_for(lk, kTimeout, [&] { return w_->mail_registered >= size_; }))
            throw std::runtime_error("local comm: a peer never registered its mailbox");
        mail_on_ = true;
        return true;
    }
    bool mailbox_enabled() const override { return mail_on_; }
    MailDesc mailbox_next() override
    {
        MailDesc m{};
        for (int r = 0; r < size_; ++r) m.peer[r] = w_->mail[r];
        m.err = &mail_err_; m.seq = ++mail_seq_; m.rank = rank_; m.n = size_; m.timeout_ticks = 0;
        st_.n_allreduce++;
        return m;
    }
    void mailbox_check() override
    {
        if (mail_err_) { mail_err_ = 0; mail_fault_ = true; throw std::runtime_error("local comm: a mailbox poll timed out"); }
    }
    // The product's Comm::engine_side_sync (csrc/rccl_comm.hip) on the double: the plain all-reduce - never the mailbox - in front of
    // a solve's first engine-side wait, carrying "a poll of mine timed out"; PCG_TEST_INJECT_ENGINE_FAULT=<rank>:<k> makes that rank
    // report one at its k-th call (the driver's reaction is what the test is after: every rank retires the forms together).
    bool mail_fault_ = false;
    int sync_calls_ = 0;
    bool engine_side_sync(void *, bool engine_side_on, bool link_fault) override
    {
        if (!engine_side_on) return false;
        ++sync_calls_;
        bool inject = false;
        if (const char *e = std::getenv("PCG_TEST_INJECT_ENGINE_FAULT")) {
            int r = -1, k = -1;
            if (std::sscanf(e, "%d:%d", &r, &k) == 2) inject = r == rank_ && k == sync_calls_;
        }
        double v = (mail_fault_ || link_fault || mail_err_ != 0 || inject) ? 1.0 : 0.0;
        const bool was = mail_on_;
        mail_on_ = false;
        allreduce(&v, 1, nullptr);
        st_.n_allreduce--;                          // (not one of the solve's MPI_SUMs)
        mail_on_ = was;
        mail_fault_ = false;
        if (v == 0.0) return false;
        mail_on_ = false;
        mail_err_ = 0;
        std::fprintf(stderr, "[pcg] rank %d: a poll of an engine-side wait timed out on some rank - all ranks return to the collective library\n", rank_);
        return true;
    }
    // COLLECTIVE like the product's (every rank's k-th call belongs to generation k): the ranks publish buffer + layout, then every
    // rank looks up its segment and arrival word at each neighbour
    std::unique_ptr<DirectLink> direct_link(const HaloHost &h, std::string &why, bool = false) override
    {
        why.clear();
        auto link = std::make_unique<LocalDirectLink>();
        link->stats = &st_;
        const int64_t total = h.n_peers > 0 ? h.send_ptr[(size_t)h.n_peers] : 0;
        link->store.assign((size_t)total + kDirectMaxPeers, 0ull);
        const int64_t gen = ++direct_calls_;
        bool bad = h.n_peers > kDirectMaxPeers;
        std::unique_lock<std::mutex> lk(w_->m);
        auto &recs = w_->direct[gen];
        recs.resize((size_t)size_);
        World::DirectRec &me = recs[(size_t)rank_];
        me.buf = link->store.data();
        me.flags_word = (size_t)total;
        for (int j = 0; j < h.n_peers; ++j) {
            me.peers.push_back(h.peer_ids[(size_t)j]);
            me.off.push_back(h.send_ptr[(size_t)j]);
            me.cnt.push_back(h.send_ptr[(size_t)j + 1] - h.send_ptr[(size_t)j]);
        }
        w_->direct_registered[gen]++;
        w_->cv.notify_all();
        if (!w_->cv.wait_for(lk, kTimeout, [&] { return w_->direct_registered[gen] >= size_; }))
            throw std::runtime_error("local comm: a peer never entered direct_link");
        DirectDesc &d = link->d;
        d = DirectDesc{};
        d.n_peers = h.n_peers; d.err = &link->err; d.timeout_ticks = 0;
        d.my_flags = link->store.data() + total;
        for (int j = 0; j <= h.n_peers && j <= kDirectMaxPeers; ++j) d.seg[j] = h.send_ptr.empty() ? 0 : h.send_ptr[(size_t)j];
        for (int j = 0; j < h.n_peers && !bad; ++j) {
            const int p = h.peer_ids[(size_t)j];
            if (p < 0 || p >= size_ || p == rank_) { bad = true; why = "neighbour part id is not a peer rank"; break; }
            const World::DirectRec &q = recs[(size_t)p];
            int k_me = -1;
            for (size_t k = 0; k < q.peers.size(); ++k)
                if (q.peers[k] == rank_) { k_me = (int)k; break; }
            if (k_me < 0 || q.cnt[(size_t)k_me] != me.cnt[(size_t)j]) { bad = true; why = "a neighbour does not list this rank with the same interface size"; break; }
            d.peer_recv[j] = reinterpret_cast<double *>(q.buf) + q.off[(size_t)k_me];
            d.peer_flag[j] = q.buf + q.flags_word + k_me;
        }
        // agree: every rank takes the same decision (a second rendezvous on the generation)
        if (bad) w_->direct_failed[gen]++;
        w_->direct_done[gen]++;
        w_->cv.notify_all();
        if (!w_->cv.wait_for(lk, kTimeout, [&] { return w_->direct_done[gen] >= size_; }))
            throw std::runtime_error("local comm: a peer never finished direct_link");
        if (w_->direct_failed[gen] != 0) {
            if (why.empty()) why = "another rank could not map a neighbour's buffer";
            return nullptr;
        }
        return link;
    }
    void allreduce(double *buf, int count, void *) override                                 // :622-628
    {
        if (mail_on_) {
            if (count > kMailMaxCount) throw std::runtime_error("local comm: mailbox all-reduce of more than 7 values");
            host_mail_allreduce(mailbox_next(), buf, count);
            return;
        }
        st_.n_allreduce++;
        std::unique_lock<std::mutex> lk(w_->m);
        // a generation is open for deposits only after every rank has left the previous one
        if (!w_->cv.wait_for(lk, kTimeout, [&] { return w_->left == 0; }))
            throw std::runtime_error("local comm: the previous all-reduce never drained");
        const int64_t my_gen = w_->gen;
        w_->dep[rank_].assign(buf, buf + count);
        if (++w_->arrived == size_) {
            w_->arrived = 0;
            w_->left = size_;
            w_->gen++;
            w_->cv.notify_all();
        } else if (!w_->cv.wait_for(lk, kTimeout, [&] { return w_->gen != my_gen; })) {
            throw std::runtime_error("local comm: a peer never reached the all-reduce");
        }
        for (int c = 0; c < count; ++c) {
            double s = 0;
            for (int r = 0; r < size_; ++r) {
                if ((int)w_->dep[r].size() != count) throw std::runtime_error("local comm: all-reduce count mismatch");
                s += w_->dep[r][c];
            }
            buf[c] = s;
        }
        if (--w_->left == 0) w_->cv.notify_all();
    }
    void set_timing(bool) override {}
    CommStats stats() override { return st_; }
};

}  // namespace

std::unique_ptr<Comm> make_rccl_comm(int, int rank, int nranks, const void *unique_ids)
{
    return std::unique_ptr<Comm>(new LocalComm(rank, nranks, unique_ids));
}

int rccl_unique_ids(void *out)
{
    std::memset(out, 0, kIdBytes);
    const uint64_t id = g_next_id.fetch_add(1);
    std::memcpy(out, "pcg-hostops-local-comm", 22);
    std::memcpy((char *)out + 32, &id, sizeof(id));
    return 0;
}

}  // namespace pcg
