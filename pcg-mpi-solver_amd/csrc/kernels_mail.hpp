// All-reduce through peer-mapped mailboxes (pcg_internal.hpp MailDesc): MPI_SUM of the reference (pcg_solver.py:622-628; three per
// iteration at :462-463, :487-488, :504-507, two here) formed INSIDE the launch that produced the local values - or by a one-wave
// kernel of its own - instead of by a collective library's kernel.  Sum in rank order: the same bits on every rank.
#pragma once
#include <hip/hip_runtime.h>

#include "pcg_internal.hpp"

namespace pcg {

// Called by EVERY thread of a workgroup (>= kMailMaxRanks threads).  `in[0 .. count)`: this rank's values in shared memory.  Returns with out[0 .. count) (shared memory) holding the sums over the
// ranks, valid for every thread.  tmp: kMailMaxRanks * kMailSlotWords doubles of shared memory.
// Thread t < n serves peer t in both directions: it posts this rank's values into rank t's mailbox (values first, then the sequence
// number with release semantics at system scope: whoever sees the number sees the values) and waits for rank t's values in this
// rank's own mailbox.  A poll that lasts longer than m.timeout_ticks of the 100 MHz wall clock (default 30 s, PCG_MAIL_TIMEOUT_S)
// reports through m.err and delivers NaN: the solve ends on it, nothing hangs.
__device__ __forceinline__ void mail_allreduce(const MailDesc &m, const double *in, int count, double *out, double *tmp)
{
    const int tid = threadIdx.x;
    const int par = (int)(m.seq & 1ull);
    if (tid < m.n) {
        unsigned long long *post = reinterpret_cast<unsigned long long *>(m.peer[tid]) + ((size_t)par * kMailMaxRanks + m.rank) * kMailSlotWords;
        for (int k = 0; k < count; ++k)
            __hip_atomic_store(post + 1 + k, (unsigned long long)__double_as_longlong(in[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(post, m.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long *box = reinterpret_cast<const unsigned long long *>(m.peer[m.rank]) + ((size_t)par * kMailMaxRanks + tid) * kMailSlotWords;
        const unsigned long long t0 = (unsigned long long)wall_clock64();
        bool ok = true;
        while (__hip_atomic_load(box, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != m.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (m.timeout_ticks && (unsigned long long)wall_clock64() - t0 > m.timeout_ticks) { ok = false; break; }   // wall clock, not a poll count
        }
        for (int k = 0; k < count; ++k) {
            const unsigned long long w = __hip_atomic_load(box + 1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            tmp[tid * kMailSlotWords + k] = ok ? __longlong_as_double((long long)w) : __longlong_as_double(0x7ff8000000000000ll);
        }
        if (!ok) __hip_atomic_store(m.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (tid < count) {
        double t = tmp[tid];
        for (int r = 1; r < m.n; ++r) t += tmp[r * kMailSlotWords + tid];      // rank order (the oracle's _allreduce, MPI_SUM :622-628)
        out[tid] = t;
    }
    __syncthreads();
}

#ifdef PCG_MAIL_STANDALONE_KERNEL      // (defined by the one translation unit that launches it: rccl_comm.hip)
// buf[0 .. count) := sum over the ranks, in place (the all-reduces outside the fused launches: true residual, norms, a part without
// neighbours).  One workgroup of 64 threads.
__global__ __launch_bounds__(64) void k_mail_allreduce(double *__restrict__ buf, int count, const MailDesc m)
{
    __shared__ double in[kMailSlotWords], out[kMailSlotWords], tmp[kMailMaxRanks * kMailSlotWords];
    if ((int)threadIdx.x < count) in[threadIdx.x] = buf[threadIdx.x];
    __syncthreads();
    mail_allreduce(m, in, count, out, tmp);
    if ((int)threadIdx.x < count) buf[threadIdx.x] = out[threadIdx.x];
}
#endif

}  // namespace pcg
