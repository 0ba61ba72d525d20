// Device group: ONE host process drives several GPUs (include/pcg_mi355x.h, pcg_group_*).
//
// The reference is one MPI rank per part (mpiexec -np N; pcg_solver.py:91) and the product's default launch mirrors it:
// one process per GPU.  This file is the alternative for a host program that exists once: member k = part k on device
// dev_ids[k].  It is written purely on top of the public per-engine C ABI - a group call is the per-engine call of the
// same name made for every member at the same time - so a group solve runs exactly the code path of N processes
// (same kernels, same native communicator, same decisions on every member) and gives bit-identical results.
//
// Why threads: the per-engine calls are collective.  A member's interface exchange (grouped ncclSend/ncclRecv) and its
// all-reduces complete only when its neighbours have issued theirs, and the host side of the PCG loop reads the
// all-reduced sums of every iteration (pcg_solver.py:507-562), so the members' loops have to advance side by side.
// One persistent host thread per member (created here, bound to the member's device by the engine's entry points)
// does that; the caller sees one thread and plain blocking calls.
#include <algorithm>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pcg_internal.hpp"

using namespace pcg;

namespace {

struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool pending = false, quit = false;
    int rc = 0;
    std::string err;

    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return pending || quit; });
            if (!pending) return;                          // quit, nothing left to run
            std::function<int()> f = std::move(job);
            lk.unlock();
            int r = 0;
            std::string e;
            try {
                r = f();
                if (r != 0) e = last_error_string();       // thread-local: the message of THIS member's failing call
            } catch (const std::exception &ex) {
                r = -1;
                e = ex.what();
            }
            lk.lock();
            rc = r;
            err = e;
            pending = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(f);
        pending = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !pending; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct pcg_group {
    int32_t n = 0;
    std::vector<int32_t> dev;
    std::vector<std::unique_ptr<Worker>> w;
    std::vector<pcg_comm *> comm;
    std::vector<pcg_engine *> eng;

    // run fn(member) on every member's thread at once; 0 when all returned 0, else -1 with the members' messages
    int run_all(const char *where, const std::function<int(int)> &fn)
    {
        for (int k = 0; k < n; ++k) w[k]->post([&fn, k]() { return fn(k); });
        for (int k = 0; k < n; ++k) w[k]->wait();
        std::string msg;
        for (int k = 0; k < n; ++k)
            if (w[k]->rc != 0)
                msg += (msg.empty() ? "" : "; ") + std::string("member ") + std::to_string(k) + " (device " + std::to_string(dev[k]) +
                       "): " + (w[k]->err.empty() ? "failed" : w[k]->err);
        if (msg.empty()) return 0;
        return set_error(std::string(where) + ": " + msg);
    }
    int need_engines(const char *where) const
    {
        for (int k = 0; k < n; ++k)
            if (!eng[k]) return set_error(std::string(where) + ": member " + std::to_string(k) + " has no engine (pcg_group_attach)");
        return 0;
    }
    ~pcg_group()
    {
        if ((int)w.size() == n && (int)comm.size() == n)   // the communicators go on the threads that created them
            (void)run_all("pcg_group_destroy", [this](int k) {
                if (comm[k]) pcg_comm_destroy(comm[k]);
                comm[k] = nullptr;
                return 0;
            });
        for (auto &x : w)
            if (x) x->stop();
    }
};

namespace {

template <class T>
T *member_ptr(T *const *arr, int k) { return arr ? arr[k] : nullptr; }

}  // namespace

extern "C" {

int pcg_group_create(int32_t n_dev, const int32_t *dev_ids, pcg_group **out)
{
    if (!out || !dev_ids || n_dev < 1) return set_error("pcg_group_create: bad argument");
    *out = nullptr;
    try {
        std::unique_ptr<pcg_group> g(new pcg_group());
        g->n = n_dev;
        g->dev.assign(dev_ids, dev_ids + n_dev);
        const int have = pcg_device_count();
        for (int k = 0; k < n_dev; ++k)
            if (dev_ids[k] < 0 || (have > 0 && dev_ids[k] >= have))
                return set_error("pcg_group_create: device id " + std::to_string(dev_ids[k]) + " out of range (" + std::to_string(have) + " visible)");
        // RCCL refuses two ranks of one communicator on the same device ("Duplicate GPU detected"), and a member that fails
        // inside the collective ncclCommInitRank can leave the others waiting: reject duplicates before any thread starts.
        // Only the tests' shared-GPU stand-in for librccl (PCG_RCCL_LIB) and the CPU test double (no devices) take them.
        if (have > 0 && !std::getenv("PCG_RCCL_LIB")) {
            std::vector<int32_t> seen(dev_ids, dev_ids + n_dev);
            std::sort(seen.begin(), seen.end());
            for (int k = 1; k < n_dev; ++k)
                if (seen[k] == seen[k - 1])
                    return set_error("pcg_group_create: device " + std::to_string(seen[k]) + " is listed twice - one member per GPU (" +
                                     std::to_string(n_dev) + " members, " + std::to_string(have) + " devices visible)");
        }
        g->comm.assign(n_dev, nullptr);
        g->eng.assign(n_dev, nullptr);
        unsigned char ids[PCG_RCCL_ID_BYTES];
        if (pcg_rccl_unique_id(ids) != 0) return -1;       // message already set
        for (int k = 0; k < n_dev; ++k) {
            g->w.emplace_back(new Worker());
            Worker *wk = g->w.back().get();
            wk->th = std::thread([wk]() { wk->loop(); });
        }
        // collective: ncclCommInitRank of every member at once, each on the thread that will issue its collectives
        pcg_group *gp = g.get();
        const int rc = g->run_all("pcg_group_create", [gp, &ids](int k) {
            return pcg_comm_create_rccl(gp->dev[k], k, gp->n, ids, &gp->comm[k]);
        });
        if (rc != 0) return rc;                            // ~pcg_group destroys what was created
        *out = g.release();
        return 0;
    } catch (const std::exception &ex) {
        return set_error(std::string("pcg_group_create: ") + ex.what());
    }
}

void pcg_group_destroy(pcg_group *g) { delete g; }

int pcg_group_size(const pcg_group *g) { return g ? g->n : -1; }

int pcg_group_device(const pcg_group *g, int32_t member) { return g && member >= 0 && member < g->n ? g->dev[member] : -1; }

pcg_comm *pcg_group_comm(pcg_group *g, int32_t member) { return g && member >= 0 && member < g->n ? g->comm[member] : nullptr; }

int pcg_group_attach(pcg_group *g, int32_t member, pcg_engine *e)
{
    if (!g || member < 0 || member >= g->n) return set_error("pcg_group_attach: bad member");
    if (!e) {
        if (g->eng[member]) (void)pcg_set_comm_native(g->eng[member], nullptr);
        g->eng[member] = nullptr;
        return 0;
    }
    const int d = pcg_engine_device(e);
    if (d != g->dev[member])
        return set_error("pcg_group_attach: member " + std::to_string(member) + " lives on device " + std::to_string(g->dev[member]) +
                         ", the engine was created on device " + std::to_string(d));
    if (pcg_set_comm_native(e, g->comm[member]) != 0) return -1;
    g->eng[member] = e;
    return 0;
}

int pcg_group_apply(pcg_group *g, const double *const *x, double *const *y)
{
    if (!g || !x || !y) return set_error("pcg_group_apply: null");
    if (g->need_engines("pcg_group_apply")) return -1;
    return g->run_all("pcg_group_apply", [&](int k) { return pcg_apply(g->eng[k], x[k], y[k]); });
}

int pcg_group_diag(pcg_group *g, double *const *d)
{
    if (!g || !d) return set_error("pcg_group_diag: null");
    if (g->need_engines("pcg_group_diag")) return -1;
    return g->run_all("pcg_group_diag", [&](int k) { return pcg_diag(g->eng[k], d[k]); });
}

int pcg_group_build_jacobi(pcg_group *g, double *const *inv_diag_out)
{
    if (!g) return set_error("pcg_group_build_jacobi: null");
    if (g->need_engines("pcg_group_build_jacobi")) return -1;
    return g->run_all("pcg_group_build_jacobi", [&](int k) { return pcg_build_jacobi(g->eng[k], member_ptr(inv_diag_out, k)); });
}

int pcg_group_update_bc(pcg_group *g, const double *const *ref_load, const double *const *ud, double delta, double *const *fext_out,
                        double *const *udi_out)
{
    if (!g || !ref_load || !ud) return set_error("pcg_group_update_bc: null");
    if (g->need_engines("pcg_group_update_bc")) return -1;
    return g->run_all("pcg_group_update_bc", [&](int k) {
        return pcg_update_bc(g->eng[k], ref_load[k], ud[k], delta, member_ptr(fext_out, k), member_ptr(udi_out, k));
    });
}

int pcg_group_dot_w(pcg_group *g, const double *const *a, const double *const *b, double *out)
{
    if (!g || !a || !b || !out) return set_error("pcg_group_dot_w: null");
    if (g->need_engines("pcg_group_dot_w")) return -1;
    std::vector<double> r((size_t)g->n, 0.0);
    const int rc = g->run_all("pcg_group_dot_w", [&](int k) { return pcg_dot_w(g->eng[k], a[k], b[k], &r[k]); });
    if (rc == 0) *out = r[0];                              // all-reduced: the same value on every member
    return rc;
}

int pcg_group_solve(pcg_group *g, const double *const *b, const double *const *x0, const double *const *inv_diag, double tol,
                    int64_t max_iter, int64_t glob_n_eff, double *const *x_out, double *const *hist, int64_t hist_cap, pcg_result *res)
{
    if (!g || !b) return set_error("pcg_group_solve: null");
    if (g->need_engines("pcg_group_solve")) return -1;
    return g->run_all("pcg_group_solve", [&](int k) {
        return pcg_solve(g->eng[k], b[k], member_ptr(x0, k), member_ptr(inv_diag, k), tol, max_iter, glob_n_eff, member_ptr(x_out, k),
                         member_ptr(hist, k), hist_cap, res ? res + k : nullptr);
    });
}

int pcg_group_set_timing(pcg_group *g, int32_t on)
{
    if (!g) return set_error("pcg_group_set_timing: null");
    return g->run_all("pcg_group_set_timing", [&](int k) { return pcg_comm_set_timing(g->comm[k], on); });
}

int pcg_group_enable_mailbox(pcg_group *g, int32_t on, int32_t *enabled_out)
{
    if (!g) return set_error("pcg_group_enable_mailbox: null");
    std::vector<int32_t> got((size_t)g->n, 0);
    const int rc = g->run_all("pcg_group_enable_mailbox", [&](int k) { return pcg_comm_enable_mailbox(g->comm[k], on, &got[(size_t)k]); });
    if (enabled_out) {
        *enabled_out = 1;
        for (int32_t v : got) if (!v) *enabled_out = 0;
    }
    return rc;
}

int pcg_group_enable_direct_exchange(pcg_group *g, int32_t on, int32_t *enabled_out)
{
    if (!g) return set_error("pcg_group_enable_direct_exchange: null");
    if (g->need_engines("pcg_group_enable_direct_exchange")) return -1;
    std::vector<int32_t> got((size_t)g->n, 0);
    const int rc = g->run_all("pcg_group_enable_direct_exchange", [&](int k) { return pcg_enable_direct_exchange(g->eng[k], on, &got[(size_t)k]); });
    if (enabled_out) {
        *enabled_out = 1;
        for (int32_t v : got) if (!v) *enabled_out = 0;
    }
    return rc;
}

}  // extern "C"
