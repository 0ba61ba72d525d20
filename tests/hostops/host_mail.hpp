// TEST DOUBLE of csrc/kernels_mail.hpp: the mailbox all-reduce protocol (MailDesc, pcg_internal.hpp) in plain host code - "device"
// memory of the CPU double is host memory and every part runs on its own host thread, so the same posts and polls work with
// __atomic builtins.  Compiled only into tests/hostops/_build/libpcg_hostops.so.
#pragma once
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "pcg_internal.hpp"

namespace pcg {

inline void host_mail_allreduce(const MailDesc &m, double *vals, int count)
{
    const int par = (int)(m.seq & 1ull);
    for (int t = 0; t < m.n; ++t) {                      // post: values first, then the sequence number (release)
        unsigned long long *post = reinterpret_cast<unsigned long long *>(m.peer[t]) + ((size_t)par * kMailMaxRanks + m.rank) * kMailSlotWords;
        for (int k = 0; k < count; ++k) {
            unsigned long long w;
            std::memcpy(&w, vals + k, 8);
            __atomic_store_n(post + 1 + k, w, __ATOMIC_RELAXED);
        }
        __atomic_store_n(post, m.seq, __ATOMIC_RELEASE);
    }
    double got[kMailMaxRanks][kMailSlotWords];
    for (int t = 0; t < m.n; ++t) {                      // poll this rank's own mailbox
        const unsigned long long *box = reinterpret_cast<const unsigned long long *>(m.peer[m.rank]) + ((size_t)par * kMailMaxRanks + t) * kMailSlotWords;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        while (__atomic_load_n(box, __ATOMIC_ACQUIRE) != m.seq) {
            std::this_thread::yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { ok = false; break; }
        }
        for (int k = 0; k < count; ++k) {
            const unsigned long long w = __atomic_load_n(box + 1 + k, __ATOMIC_RELAXED);
            std::memcpy(&got[t][k], &w, 8);
            if (!ok) got[t][k] = std::nan("");
        }
        if (!ok) __atomic_store_n(m.err, 1u, __ATOMIC_RELAXED);
    }
    for (int k = 0; k < count; ++k) {                    // rank order
        double s = got[0][k];
        for (int r = 1; r < m.n; ++r) s += got[r][k];
        vals[k] = s;
    }
}

}  // namespace pcg
