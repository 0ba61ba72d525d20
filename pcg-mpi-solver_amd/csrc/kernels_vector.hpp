// Vector-phase, interface and reduction kernels of the PCG iteration (reference: pcg_solver.py:447-516, :307-334).
#pragma once
#include "hip_common.hpp"
#include "kernels_mail.hpp"

namespace pcg {

// Sum of count_a values at pa (+ count_b values at pb when pb != null) by the FIRST 256 threads of the workgroup, in one
// fixed order whatever the workgroup size: thread t < 256 keeps 4 independent partial sums over pa[t], pa[t + 256], ...
// (loads in flight), the values of pb go into the second one, (s0 + s1) + (s2 + s3), a wave shuffle tree, the four waves in
// turn.  Every thread of the workgroup gets the total.  No float atomics anywhere: bit-reproducible run to run, and a sum
// formed by k_reduce equals the sum the fused vector kernel forms for itself from the same partials.
// SC1: the partials were published inside this launch by other workgroups (agent-scope stores): agent-scope loads.
template <bool SC1>
__device__ __forceinline__ double reduce_fixed_256(const double *pa, int count_a, const double *pb, int count_b, double *lds /* 5 */)
{
    auto ld = [](const double *q) -> double {
        if constexpr (SC1) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *q;
    };
    const int tid = threadIdx.x;
    if (tid < 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = tid;
        for (; b + 3 * 256 < count_a; b += 4 * 256) {
            s0 += ld(pa + b); s1 += ld(pa + b + 256); s2 += ld(pa + b + 2 * 256); s3 += ld(pa + b + 3 * 256);
        }
        for (; b < count_a; b += 256) s0 += ld(pa + b);
        if (pb)
            for (int c = tid; c < count_b; c += 256) s1 += ld(pb + c);
        const double w = wave_sum((s0 + s1) + (s2 + s3));
        if ((tid & 63) == 0) lds[tid >> 6] = w;
    }
    __syncthreads();
    if (tid == 0) {
        double t = lds[0];
        t += lds[1]; t += lds[2]; t += lds[3];
        lds[4] = t;
    }
    __syncthreads();
    const double r = lds[4];
    __syncthreads();                                    // lds may be reused by the caller
    return r;
}

// The same sum, in the same order, of ONE value per thread already in registers (thread t < count holds pa[t]; count <= 256):
// the in-band grid barrier of k_vec<FUSED> polls the published sums themselves and feeds them in from here.
__device__ __forceinline__ double reduce_fixed_256_regs(double mine, int count, double *lds /* 5 */)
{
    const int tid = threadIdx.x;
    if (tid < 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (tid < count) s0 += mine;
        const double w = wave_sum((s0 + s1) + (s2 + s3));
        if ((tid & 63) == 0) lds[tid >> 6] = w;
    }
    __syncthreads();
    if (tid == 0) {
        double t = lds[0];
        t += lds[1]; t += lds[2]; t += lds[3];
        lds[4] = t;
    }
    __syncthreads();
    const double r = lds[4];
    __syncthreads();
    return r;
}

// out[v] = sum_{b<count_a} pa[v*stride + b] (+ sum_{b<count_b} pb[b] when pb != null), v = blockIdx.x: one block per value.
// mirror (may be null): host-visible copy of the words written, so the host needs no device->host copy.
__global__ __launch_bounds__(kBlock) void k_reduce(const double *__restrict__ pa, int count_a, int stride,
                                                   const double *__restrict__ pb, int count_b, double *out, double *mirror)
{
    __shared__ double lds[5];
    const int k = blockIdx.x;
    const double v = reduce_fixed_256<false>(pa + (size_t)k * stride, count_a, pb, count_b, lds);
    if (threadIdx.x == 0) {
        out[k] = v;
        if (mirror) mirror[k] = v;
    }
}

// "Last workgroup to finish reduces" (round 4, multi-part loop): a launch that leaves one partial per workgroup can also form the
// total - in reduce_fixed_256's order, whichever workgroup happens to be last - instead of leaving it to a k_reduce launch.  Thread 0
// has published the workgroup's partial(s) with agent-scope stores; it counts the workgroup in (acq_rel: the partials are out
// before, and the last arriver sees everybody's).  Self-contained (round 5): the last arriver - the one that counts gridDim.x - puts
// the counter back to 0 for the next launch on the stream, so the test never depends on a launch number the host keeps in step
// (a lost launch or a changed grid cannot leave every workgroup believing it is not the last).  -> true in every thread of the
// last workgroup.
__device__ __forceinline__ bool last_workgroup(unsigned long long *counter, int *lds_flag)
{
    if (threadIdx.x == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1 == (unsigned long long)gridDim.x;
        if (last) __hip_atomic_store(counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = last ? 1 : 0;
    }
    __syncthreads();
    const bool last = *lds_flag != 0;
    __syncthreads();
    return last;
}

// ------------------------------------------------------------------------------------------------
// interface kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_halo_pack(const double *__restrict__ y, const int *__restrict__ idx,
                                                      double *__restrict__ send, int64_t count)
{
    for (int64_t m = blockIdx.x * (int64_t)kBlock + threadIdx.x; m < count; m += (int64_t)gridDim.x * kBlock)
        send[m] = y[idx[m]];
}

// Direct exchange (pcg_internal.hpp DirectDesc; round 5, opt-in): :307-309 AND :318-326 in one launch - entry m of the packed list goes
// straight into the receive buffer of the neighbour whose segment holds m (a store over xGMI into uncached memory mapped here); every
// workgroup makes its stores visible system-wide before it counts itself in, and the last one posts this rank's arrival word at every
// neighbour (release, system scope: whoever sees the number sees the values).
__global__ __launch_bounds__(kBlock) void k_halo_put(const double *__restrict__ y, const int *__restrict__ idx, int64_t count,
                                                     const DirectDesc d, unsigned long long *counter)
{
    for (int64_t m = blockIdx.x * (int64_t)kBlock + threadIdx.x; m < count; m += (int64_t)gridDim.x * kBlock) {
        int j = 0;
        while (j + 1 < d.n_peers && m >= d.seg[j + 1]) ++j;           // (<= 26 neighbours; the segments are contiguous runs of m)
        d.peer_recv[j][m - d.seg[j]] = y[idx[m]];
    }
    __threadfence_system();
    __shared__ int flag;
    if (last_workgroup(counter, &flag)) {
        __threadfence_system();
        if ((int)threadIdx.x < d.n_peers)
            __hip_atomic_store(d.peer_flag[threadIdx.x], d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The other side: until every neighbour has posted exchange w.seq (acquire).  Called by every thread of a workgroup at the start of
// k_fixup.  A poll that runs out of patience (seconds) reports through w.err; the launch goes on with whatever is in the buffer and
// the host ends the solve on the report (DirectLink::check).
__device__ __forceinline__ void wait_for_neighbours(const FixWait &w)
{
    if (w.n <= 0) return;
    if ((int)threadIdx.x < w.n) {
        const unsigned long long t0 = (unsigned long long)wall_clock64();
        while (__hip_atomic_load(w.flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < w.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (w.timeout_ticks && (unsigned long long)wall_clock64() - t0 > w.timeout_ticks) {
                __hip_atomic_store(w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

// y[d] += recv[...] in neighbour order for the interface dofs; optional dot over all boundary-slice dofs
// REDUCE (with DOT): the last workgroup to finish sums the apply's dot partials - pa[0 .. count_a) of the operator launches, then
// this launch's - in k_reduce's fixed order into red[0]: the p.Ap of the multi-part loop without a reduce launch.
// MAIL (with REDUCE, round 5): that workgroup then all-reduces the sum across the ranks through the mailboxes (kernels_mail.hpp):
// red[0] is the GLOBAL p.Ap (:487-488) when the launch is done.
struct FixReduce { const double *pa; int count_a; double *red; unsigned long long *counter; MailDesc mail; };

template <bool DOT, bool REDUCE = false, bool MAIL = false>
__global__ __launch_bounds__(kBlock) void k_fixup(double *__restrict__ y, const double *__restrict__ recv,
                                                  const int *__restrict__ fptr, const int *__restrict__ fpos,
                                                  const double *__restrict__ xdot, const uint8_t *__restrict__ flags,
                                                  int64_t nb, double *__restrict__ partials, FixReduce fr, const FixWait fw)
{
    wait_for_neighbours(fw);                                       // direct exchange: `recv` is being filled by the neighbours' launches
    double dot = 0.0;
    for (int64_t d = blockIdx.x * (int64_t)kBlock + threadIdx.x; d < nb; d += (int64_t)gridDim.x * kBlock) {
        double v = y[d];
        const int q0 = fptr[d], q1 = fptr[d + 1];
        for (int q = q0; q < q1; ++q) v += recv[fpos[q]];
        if (q1 > q0) y[d] = v;
        if constexpr (DOT)
            if ((flags[d] & 3) == 3) dot += xdot[d] * v;
    }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock + 5];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if constexpr (!REDUCE) {
            if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
        } else {
            __shared__ int flag;
            if (threadIdx.x == 0) __hip_atomic_store(partials + blockIdx.x, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last_workgroup(fr.counter, &flag)) {
                const double tot = reduce_fixed_256<true>(fr.pa, fr.count_a, partials, (int)gridDim.x, lds);
                if constexpr (!MAIL) {
                    if (threadIdx.x == 0) fr.red[0] = tot;
                } else {
                    __shared__ double m_in[kMailSlotWords], m_out[kMailSlotWords], m_tmp[kMailMaxRanks * kMailSlotWords];
                    if (threadIdx.x == 0) m_in[0] = tot;
                    __syncthreads();
                    mail_allreduce(fr.mail, m_in, 1, m_out, m_tmp);
                    if (threadIdx.x == 0) fr.red[0] = m_out[0];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// vector kernels (grid-stride, 16 B per lane, scalar tail)
// ------------------------------------------------------------------------------------------------
// beta = rho / rho_prev (:475) with rho = st[RHO_NEXT] read on the device: the host need not know rho yet when it
// enqueues this kernel (look-ahead), and divides the same two doubles later for its own Flag-4 test (:476-478).
// mirror != null (multi-part loop, round 4): workgroup 0 first copies the status block into the host-visible ring slot of the
// iteration BEFORE (k_publish folded in: the all-reduce rewrote the block in place after the kernels had mirrored it).
__global__ __launch_bounds__(kBlock) void k_update_p(double *__restrict__ po, const double *__restrict__ pi,
                                                     const double *__restrict__ r, const double *__restrict__ minv,
                                                     const double *__restrict__ st, double rho_prev, int first, int nt, int64_t n,
                                                     double *__restrict__ mirror)
{
    if (mirror && blockIdx.x == 0 && threadIdx.x < ST_COUNT) mirror[threadIdx.x] = st[threadIdx.x];
    const double beta = first ? 0.0 : st[ST_RHO_NEXT] / rho_prev;
    const int64_t n2 = n >> 1;
    const int64_t t0 = blockIdx.x * (int64_t)kBlock + threadIdx.x, ts = (int64_t)gridDim.x * kBlock;
    double2 *po2 = reinterpret_cast<double2 *>(po);
    const double2 *pi2 = reinterpret_cast<const double2 *>(pi);
    const double2 *r2 = reinterpret_cast<const double2 *>(r), *m2 = reinterpret_cast<const double2 *>(minv);
    for (int64_t t = t0; t < n2; t += ts) {
        const double2 rr = (nt & 4) ? ntload(r2 + t) : r2[t], mm = (nt & 4) ? ntload(m2 + t) : m2[t];
        double2 z = make_double2(mm.x * rr.x, mm.y * rr.y);        // :447
        if (!first) { const double2 pp = (nt & 4) ? ntload(pi2 + t) : pi2[t]; z.x = z.x + beta * pp.x; z.y = z.y + beta * pp.y; }   // :479
        if (nt & 1) ntstore(po2 + t, z); else po2[t] = z;
    }
    if ((n & 1) && t0 == 0) {
        const int64_t i = n - 1;
        double z = minv[i] * r[i];
        if (!first) z = z + beta * pi[i];
        po[i] = z;
    }
}

// whole status block -> host-visible ring slot (multi-GPU: the all-reduce rewrote the block in place)
__global__ void k_publish(const double *__restrict__ st, double *__restrict__ mirror)
{
    if (threadIdx.x < ST_COUNT) mirror[threadIdx.x] = st[threadIdx.x];
}

struct Up { double sqp, sqx, sqr, rho, ninf; };

// -> z = M^-1 r' (:447 of the next iteration)
__device__ __forceinline__ double update_one(double alpha, double p, double q, double &r, double xo, double &xn, double m,
                                             uint8_t f, Up &u)
{
    const bool w = (f & 3) == 3;
    if (w) { u.sqp += p * p; u.sqx += xo * xo; }                  // :504-505 (x BEFORE the update)
    const double rn = r - alpha * q;                               // :501
    r = rn;
    xn = xo + alpha * p;                                           // :516
    const double z = m * rn;                                       // :447 of the next iteration
    if ((f & 2) && isinf(z)) u.ninf += 1.0;                        // :448
    if (w) { u.sqr += rn * rn; u.rho += z * rn; }                  // :506, :462
    return z;
}

// ------------------------------------------------------------------------------------------------
// k_vec: the vector phase of one iteration in ONE launch (pcg_solver.py:487-516 and :447-479 of the next iteration).
//
//   [alpha]   pq = p.Ap: pq_src 2 = every workgroup sums the operator's dot partials itself (fixed order, so all get the
//             same bits; no reduce launch), 1 = st[PQ] (all-reduced, multi-GPU), 0 = alpha given in st[ALPHA] (tests);
//             alpha = rho / pq and the Flag-4 tests of :492-498 (sticky stop flag: a frozen launch updates nothing)
//   [update]  sums of p^2 w and x^2 w, r' = r - alpha q, x' = x + alpha p, z = M^-1 r', sums of r'^2 w, z r' w, #inf(z)
//   FUSED (single part):
//   [reduce]  the workgroups publish their five partial sums with agent-scope stores and meet at a grid barrier (8 sharded
//             arrival counters, eight lanes poll); every workgroup then sums the partials of rho' in the same fixed order ->
//             beta = rho' / rho (:475), workgroups 0..4 write one of the five sums each to the status block (and its mirror)
//   [p]       p' = z + beta p (:479) with z still in REGISTERS (kVecKreg x 16 B per thread): r' and M^-1 are not read
//             again; beyond that (more than 2 * kVecKreg * threads dofs) z is recomputed from r', M^-1 - the same product.
//   !FUSED    the partial sums go to `partials` for k_reduce (then the all-reduce, then k_update_p), multi-GPU loop.
//
// One workgroup of 1024 threads per CU: the grid barrier needs every workgroup resident (checked at start-up with the
// occupancy query; 16 waves per CU at <= 128 VGPRs) and costs ~4 us at 256 workgroups (MI355X_MICROARCH.md barrier-xcd);
// it replaces two reduce launches, one vector kernel launch and 16 B per dof of re-reads.  Thread t of the grid owns the
// 16-byte chunks t, t + T, t + 2T, ... in both forms and the reductions share reduce_fixed_256, so the fused and the
// split form produce identical bits (tests/test_gpu_parity.py::test_fused_vector_phase_is_bit_identical).
// ------------------------------------------------------------------------------------------------
constexpr int kVecBlock = 1024;
constexpr int kVecWaves = kVecBlock / 64;
constexpr int kVecKreg = 20;              // 10.1 M dof on 256 CUs: 19.3 chunks per thread
constexpr int kVecSyncWords = 16 * 8;     // 8 shard counters, 128 B apart
// In-band grid barrier (round 5): a slot of `pub` holds this NaN pattern (both halves equal: one 32-bit fill writes it) until its
// workgroup publishes a sum there; a reader polls the SLOT - no arrival counter, no second round trip for the value.  A sum that
// happened to be this very NaN would read as "not there yet" until the poll times out (reported like any barrier time-out).
constexpr unsigned kVecSentinelHalf = 0x7ff85ea1u;
constexpr unsigned long long kVecSentinel = ((unsigned long long)kVecSentinelHalf << 32) | kVecSentinelHalf;

struct VecArgs {
    double *st, *mirror;
    const double *p, *q, *r;
    double *rn;
    const double *xo;
    double *xn;
    const double *minv;
    const uint8_t *flags;
    double *p_next;                       // FUSED
    double *partials;                     // 5 x kMaxPartials
    const double *pa, *pb;                // pq_src 2: the operator's dot partials (interior launches; boundary fix-up)
    int count_a, count_b;
    unsigned long long *sync;             // FUSED: monotonic arrival counters
    double *pub;                          // FUSED, in-band barrier (round 5): 2 x 5 x kMaxPartials doubles, kVecSentinel between uses
    int inband;                           // ... 1 = the published sums are their own arrival flags (5 <= grid <= 256)
    unsigned long long seq;               // FUSED: number of this launch (1, 2, ...): targets = arrivals per launch x seq
    int pq_src, nt;
    int kreg;                             // FUSED: chunks of z kept in registers, <= kVecKreg (tests lower it: PCG_VEC_KREG)
    unsigned spin_limit;                  // FUSED: polls of the grid barrier before a workgroup gives up (2^22 = seconds; tests: PCG_TEST_VEC_SPINS)
    int reduce_last;                      // !FUSED: the last workgroup to finish reduces the five sums into st[SQP..NINF] (multi-part loop)
    unsigned long long *last_counter;     // ... its arrival counter (0 between launches: the last arriver resets it)
    int mail_on;                          // ... and all-reduces them across the ranks through the mailboxes (round 5)
    MailDesc mail;
    int64_t n;
};

template <int NV, int NW>
__device__ __forceinline__ void block_sum_w(double (&v)[NV], double *lds /* NV * NW */)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[k * NW + wid] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = lds[k * NW];
            for (int w = 1; w < NW; ++w) s += lds[k * NW + w];
            v[k] = s;
        }
    }
    __syncthreads();
}

constexpr int kVecPre = 4;                // FUSED, small systems: chunks per thread whose operands are requested before the alpha prologue

template <bool FUSED, bool PRE = false>
__global__ __launch_bounds__(kVecBlock) void k_vec(const VecArgs a)
{
    __shared__ double lds[5 * kVecWaves + 8];
    const int tid = threadIdx.x;
    // Small systems (at most kVecPre chunks per thread: up to 2.1 M dof on 256 CUs - BASELINE configs[1], and a GPU's share of the
    // 10 M-dof system on 8): the launch is a chain of dependent round trips, not a stream (25 us for 92 MB at 1.27 M dof, round 3).
    // Round 4 takes two of them out: the operands of ALL of a thread's chunks are requested BEFORE the alpha prologue (its partial
    // sums, two block barriers) instead of after it, and p stays in registers for p' = z + beta p instead of being read again
    // behind the grid barrier.  Same chunks per thread, same arithmetic in the same order: the bits of the general path.
    const int64_t n2_ = a.n >> 1, T_ = (int64_t)gridDim.x * kVecBlock, t0_ = (int64_t)blockIdx.x * kVecBlock + tid;
    constexpr bool pre = PRE;                                  // (chosen by the launcher: Backend::vec_update)
    double2 pre_p[PRE ? kVecPre : 1], pre_q[PRE ? kVecPre : 1], pre_x[PRE ? kVecPre : 1], pre_m[PRE ? kVecPre : 1], pre_r[PRE ? kVecPre : 1];
    uchar2 pre_f[PRE ? kVecPre : 1];
    if constexpr (PRE) {
        {
            const bool ntl0 = (a.nt & 4) != 0;
#pragma unroll
            for (int k = 0; k < kVecPre; ++k) {
                const int64_t t = t0_ + k * T_;
                if (t < n2_) {
                    const double2 *P2 = reinterpret_cast<const double2 *>(a.p) + t, *Q2 = reinterpret_cast<const double2 *>(a.q) + t,
                                  *X2 = reinterpret_cast<const double2 *>(a.xo) + t, *M2 = reinterpret_cast<const double2 *>(a.minv) + t,
                                  *R2 = reinterpret_cast<const double2 *>(a.r) + t;
                    pre_p[k] = ntl0 ? ntload(P2) : *P2; pre_q[k] = ntl0 ? ntload(Q2) : *Q2; pre_x[k] = ntl0 ? ntload(X2) : *X2;
                    pre_m[k] = ntl0 ? ntload(M2) : *M2; pre_r[k] = ntl0 ? ntload(R2) : *R2;
                    pre_f[k] = reinterpret_cast<const uchar2 *>(a.flags)[t];
                }
            }
        }
    }
    if constexpr (FUSED) {
        // in-band barrier: this launch publishes into the slots of parity seq & 1; its own slots of the OTHER parity - read for the
        // last time by the launch before (complete: same stream) - go back to the sentinel for the launch after.  Frozen or not.
        if (a.inband && tid < 5)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.pub) + ((size_t)((a.seq + 1) & 1) * 5 + tid) * kMaxPartials + blockIdx.x,
                               kVecSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double stop = a.st[ST_STOP], alpha = a.st[ST_ALPHA];
    const double rho = a.st[ST_RHO_NEXT];
    if (a.pq_src) {                                                // :487-498
        const double pq = a.pq_src == 2 ? reduce_fixed_256<false>(a.pa, a.count_a, a.pb, a.count_b, lds) : a.st[ST_PQ];
        if (pq <= 0.0 || isinf(pq)) stop = 1.0;                    // the sticky stop flag only ever goes 0 -> 1
        else { alpha = rho / pq; if (isinf(alpha)) stop = 1.0; }
        if (blockIdx.x == 0 && tid == 0) {
            a.st[ST_RHO] = rho; a.st[ST_PQ] = pq; a.st[ST_ALPHA] = alpha; a.st[ST_STOP] = stop;
            if (a.mirror) { a.mirror[ST_RHO] = rho; a.mirror[ST_PQ] = pq; a.mirror[ST_ALPHA] = alpha; a.mirror[ST_STOP] = stop; }
        }
    }
    const bool ntl = (a.nt & 4) != 0, nts = (a.nt & 1) != 0;
    const int64_t n2 = a.n >> 1, T = (int64_t)gridDim.x * kVecBlock, t0 = (int64_t)blockIdx.x * kVecBlock + tid;
    const double2 *p2 = reinterpret_cast<const double2 *>(a.p), *q2 = reinterpret_cast<const double2 *>(a.q);
    const double2 *x2 = reinterpret_cast<const double2 *>(a.xo), *m2 = reinterpret_cast<const double2 *>(a.minv);
    const double2 *r2 = reinterpret_cast<const double2 *>(a.r);
    double2 *rn2 = reinterpret_cast<double2 *>(a.rn), *xn2 = reinterpret_cast<double2 *>(a.xn);
    const uchar2 *f2 = reinterpret_cast<const uchar2 *>(a.flags);
    Up u = {0, 0, 0, 0, 0};
    double2 z[PRE ? kVecPre : (FUSED ? kVecKreg : 1)];
    double z_tail = 0.0;
    auto chunk = [&](int64_t t) -> double2 {
        const double2 pp = ntl ? ntload(p2 + t) : p2[t], qq = ntl ? ntload(q2 + t) : q2[t], xx = ntl ? ntload(x2 + t) : x2[t],
                      mm = ntl ? ntload(m2 + t) : m2[t];
        double2 rr = ntl ? ntload(r2 + t) : r2[t], xo2, zz;
        const uchar2 ff = f2[t];
        zz.x = update_one(alpha, pp.x, qq.x, rr.x, xx.x, xo2.x, mm.x, ff.x, u);
        zz.y = update_one(alpha, pp.y, qq.y, rr.y, xx.y, xo2.y, mm.y, ff.y, u);
        if (nts) { ntstore(rn2 + t, rr); ntstore(xn2 + t, xo2); }
        else { rn2[t] = rr; xn2[t] = xo2; }
        return zz;
    };
    if (stop == 0.0) {                                             // frozen when pq / alpha broke down (:492-498)
        if constexpr (pre) {                                       // (round 5: the split form of the multi-part loop preloads as well)
#pragma unroll
            for (int k = 0; k < kVecPre; ++k) {
                const int64_t t = t0 + k * T;
                if (t < n2) {                                      // chunk() on the operands already in registers
                    double2 rr = pre_r[k], xo2, zz;
                    zz.x = update_one(alpha, pre_p[k].x, pre_q[k].x, rr.x, pre_x[k].x, xo2.x, pre_m[k].x, pre_f[k].x, u);
                    zz.y = update_one(alpha, pre_p[k].y, pre_q[k].y, rr.y, pre_x[k].y, xo2.y, pre_m[k].y, pre_f[k].y, u);
                    if (nts) { ntstore(rn2 + t, rr); ntstore(xn2 + t, xo2); }
                    else { rn2[t] = rr; xn2[t] = xo2; }
                    if constexpr (FUSED) z[k] = zz;
                }
            }
        } else if constexpr (FUSED) {
#pragma unroll
            for (int k = 0; k < kVecKreg; ++k) {
                const int64_t t = t0 + k * T;
                if (k < a.kreg && t < n2) z[k < (int)(sizeof(z) / sizeof(z[0])) ? k : 0] = chunk(t);
            }
            for (int64_t t = t0 + a.kreg * T; t < n2; t += T) (void)chunk(t);
        } else {
            for (int64_t t = t0; t < n2; t += T) (void)chunk(t);
        }
        if ((a.n & 1) && t0 == 0) {
            const int64_t i = a.n - 1;
            double rr = a.r[i], xnew;
            z_tail = update_one(alpha, a.p[i], a.q[i], rr, a.xo[i], xnew, a.minv[i], a.flags[i], u);
            a.rn[i] = rr;
            a.xn[i] = xnew;
        }
    }
    double v[5] = {u.sqp, u.sqx, u.sqr, u.rho, u.ninf};
    block_sum_w<5, kVecWaves>(v, lds);                             // valid in thread 0
    if constexpr (!FUSED) {
        if (!a.reduce_last) {
            if (tid == 0)
#pragma unroll
                for (int k = 0; k < 5; ++k) a.partials[(size_t)k * kMaxPartials + blockIdx.x] = v[k];
        } else {
            // multi-part loop (round 4): no k_reduce launch before the all-reduce - the last workgroup to finish forms the five sums
            // from everybody's partials in k_reduce's order (a frozen launch publishes zeros: its sums are not used)
            __shared__ int flag;
            if (tid == 0)
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    __hip_atomic_store(a.partials + (size_t)k * kMaxPartials + blockIdx.x, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last_workgroup(a.last_counter, &flag)) {
                double s5[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) s5[k] = reduce_fixed_256<true>(a.partials + (size_t)k * kMaxPartials, (int)gridDim.x, nullptr, 0, lds);
                if (a.mail_on) {                                   // :504-507 across the ranks, inside this launch 
                    __shared__ double m_in[kMailSlotWords], m_out[kMailSlotWords], m_tmp[kMailMaxRanks * kMailSlotWords];
                    if (tid == 0)
#pragma unroll
                        for (int k = 0; k < 5; ++k) m_in[k] = s5[k];
                    __syncthreads();
                    mail_allreduce(a.mail, m_in, 5, m_out, m_tmp);
#pragma unroll
                    for (int k = 0; k < 5; ++k) s5[k] = m_out[k];
                }
                if (tid == 0)
#pragma unroll
                    for (int k = 0; k < 5; ++k) { a.st[ST_SQP + k] = s5[k]; if (a.mirror) a.mirror[ST_SQP + k] = s5[k]; }
            }
        }
    } else {
        // ---- grid barrier.  Every workgroup adds ONE non-returning agent-scope arrival to the counter of its shard (8 shards,
        // blockIdx % 8: the arrivals of a launch spread over 8 words instead of queueing on one) after its partial sums are out;
        // lanes 0..7 of its first wave then poll the 8 counters until each has all the arrivals of THIS launch.  The counters are
        // monotonic over the launches of an engine (target = arrivals per launch x launch number; nothing is reset in between),
        // so a frozen launch still counts its arrivals.
        const int G = gridDim.x, ns = G < 8 ? G : 8, shard = blockIdx.x % ns;
        const bool inband = a.inband != 0;                         // (uniform over the grid: chosen by the launcher, 5 <= G <= 256)
        double got_rho = 0.0, got_k = 0.0;                         // in-band: thread t < G ends up with workgroup t's rho' / row-k sum
        const int krow = blockIdx.x < 5 ? (int)blockIdx.x : 3;     // workgroups 0..4 also total one of the five sums each
        if (inband) {
            // ---- in-band grid barrier (round 5): the published sums are their own arrival flags.  Thread 0 stores the five sums
            // (plain copies to `partials` as well: the host's recovery path reads those after a time-out) and nobody waits for the
            // stores; threads t < G of EVERY workgroup poll slot t of the rho' row (workgroups 0..4: of their own row too) until the
            // sentinel is gone.  One round trip between "my sums are out" and "I have everybody's", where the counter form has
            // three (stores acknowledged -> arrival counted -> arrival seen -> sums read).
            const unsigned long long *pub = reinterpret_cast<const unsigned long long *>(a.pub) + (size_t)(a.seq & 1) * 5 * kMaxPartials;
            if (tid == 0 && stop == 0.0) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    __hip_atomic_store(a.pub + ((size_t)(a.seq & 1) * 5 + k) * kMaxPartials + blockIdx.x, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.partials[(size_t)k * kMaxPartials + blockIdx.x] = v[k];
                }
            }
            if (tid < 256) {
                bool ok = true;
                if (stop == 0.0) {
                    const bool act = tid < G;
                    unsigned long long w3 = act ? kVecSentinel : 0ull, wk = act && krow != 3 ? kVecSentinel : 0ull;
                    unsigned spins = 0;
                    for (;;) {
                        if (w3 == kVecSentinel) w3 = __hip_atomic_load(pub + (size_t)3 * kMaxPartials + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (wk == kVecSentinel) wk = __hip_atomic_load(pub + (size_t)krow * kMaxPartials + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all(w3 != kVecSentinel && wk != kVecSentinel)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > a.spin_limit) { ok = false; break; }   // seconds: a workgroup of the grid is not resident
                    }
                    got_rho = __longlong_as_double((long long)w3);
                    got_k = krow != 3 ? __longlong_as_double((long long)wk) : got_rho;
                }
                if ((tid & 63) == 0) lds[5 * kVecWaves + 1 + (tid >> 6)] = ok ? 1.0 : 0.0;
            }
            __syncthreads();
            if (tid == 0) lds[5 * kVecWaves] = (lds[5 * kVecWaves + 1] != 0.0 && lds[5 * kVecWaves + 2] != 0.0 && lds[5 * kVecWaves + 3] != 0.0 && lds[5 * kVecWaves + 4] != 0.0) ? 1.0 : 0.0;
        } else
        if (tid < 64) {
            if (tid == 0) {
                if (stop == 0.0)
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        __hip_atomic_store(a.partials + (size_t)k * kMaxPartials + blockIdx.x, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partials are out before the arrival is counted
                (void)__hip_atomic_fetch_add(a.sync + 16 * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool ok = true;
            if (stop == 0.0) {
                const unsigned long long want = tid < ns ? (unsigned long long)((G - tid + ns - 1) / ns) * a.seq : 0ull;
                unsigned spins = 0;
                for (;;) {
                    const unsigned long long got = tid < ns ? __hip_atomic_load(a.sync + 16 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    if (__all(got >= want)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.spin_limit) { ok = false; break; }   // seconds: a workgroup of the grid is not resident
                }
            }
            if (tid == 0) lds[5 * kVecWaves] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        const bool ok = lds[5 * kVecWaves] != 0.0;
        __syncthreads();
        if (stop != 0.0 || !ok) {                                  // uniform over the grid (frozen) or reported (time-out)
            if (blockIdx.x == 0 && tid == 0 && stop != 0.0) {
#pragma unroll
                for (int k = 0; k < 5; ++k) { a.st[ST_SQP + k] = 0.0; if (a.mirror) a.mirror[ST_SQP + k] = 0.0; }
            }
            // EVERY workgroup that gives up says so (the store is idempotent): one that starts late may find all arrivals counted and
            // pass while others left without their part of p'.  r', x' and the partial sums of every workgroup are complete either
            // way (written before the barrier); the host recovers from them in the split form (pcg_driver.cpp iterate_once).
            if (!ok && tid == 0) { a.st[ST_ERR] = 1.0; if (a.mirror) a.mirror[ST_ERR] = 1.0; }
            return;
        }
        // ---- every workgroup: rho' from the G partials, same order everywhere -> same beta everywhere (:462, :475);
        // the five sums of the status block are written by workgroups 0..4, one each (by workgroup 0 alone on a tiny grid)
        const double rho_next = inband ? reduce_fixed_256_regs(got_rho, G, lds) : reduce_fixed_256<true>(a.partials + (size_t)3 * kMaxPartials, G, nullptr, 0, lds);
        if (G >= 5) {
            if (blockIdx.x < 5) {
                const int k = blockIdx.x;
                const double sk = k == 3 ? rho_next : inband ? reduce_fixed_256_regs(got_k, G, lds) : reduce_fixed_256<true>(a.partials + (size_t)k * kMaxPartials, G, nullptr, 0, lds);
                if (tid == 0) { a.st[ST_SQP + k] = sk; if (a.mirror) a.mirror[ST_SQP + k] = sk; }
            }
        } else if (blockIdx.x == 0) {
            double s[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                s[k] = k == 3 ? rho_next : reduce_fixed_256<true>(a.partials + (size_t)k * kMaxPartials, G, nullptr, 0, lds);
            if (tid == 0)
#pragma unroll
                for (int k = 0; k < 5; ++k) { a.st[ST_SQP + k] = s[k]; if (a.mirror) a.mirror[ST_SQP + k] = s[k]; }
        }
        const double beta = rho_next / rho;
        const double2 *pc2 = reinterpret_cast<const double2 *>(a.p);
        double2 *pn2 = reinterpret_cast<double2 *>(a.p_next);
        if constexpr (pre) {
#pragma unroll
            for (int k = 0; k < kVecPre; ++k) {
                const int64_t t = t0 + k * T;
                if (t < n2) {
                    const double2 o = make_double2(z[k].x + beta * pre_p[k].x, z[k].y + beta * pre_p[k].y);      // :479, p still in registers
                    if (nts) ntstore(pn2 + t, o); else pn2[t] = o;
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < kVecKreg; ++k) {
            const int64_t t = t0 + k * T;
            if (k < a.kreg && t < n2) {
                const double2 pp = pc2[t];
                const double2 zk = z[k < (int)(sizeof(z) / sizeof(z[0])) ? k : 0];
                const double2 o = make_double2(zk.x + beta * pp.x, zk.y + beta * pp.y);      // :479
                if (nts) ntstore(pn2 + t, o); else pn2[t] = o;
            }
        }
        }
        for (int64_t t = t0 + a.kreg * T; t < n2; t += T) {        // beyond the register-resident part: z again from r', M^-1
            const double2 rr = rn2[t], mm = m2[t], pp = pc2[t];
            const double2 o = make_double2(mm.x * rr.x + beta * pp.x, mm.y * rr.y + beta * pp.y);
            if (nts) ntstore(pn2 + t, o); else pn2[t] = o;
        }
        if ((a.n & 1) && t0 == 0) a.p_next[a.n - 1] = z_tail + beta * a.p[a.n - 1];
    }
}

__global__ __launch_bounds__(kBlock) void k_residual(const double *__restrict__ b, const double *__restrict__ ax,
                                                     double *__restrict__ r, const double *__restrict__ minv,
                                                     const uint8_t *__restrict__ flags, double *__restrict__ partials, int64_t n)
{
    __shared__ double lds[3 * kWavesPerBlock];
    double sqr = 0, rho = 0, ninf = 0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double rn = b[i] - ax[i];                            // :414, :531
        r[i] = rn;
        const double z = minv[i] * rn;
        const uint8_t f = flags[i];
        if ((f & 2) && isinf(z)) ninf += 1.0;
        if ((f & 3) == 3) { sqr += rn * rn; rho += z * rn; }       // :415, :462
    }
    double v[3] = {sqr, rho, ninf};
    block_sum<3>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) partials[(size_t)k * kMaxPartials + blockIdx.x] = v[k];
}

__global__ __launch_bounds__(kBlock) void k_dot_w(const double *__restrict__ a, const double *__restrict__ b,
                                                  const uint8_t *__restrict__ flags, double *__restrict__ partials, int64_t n)
{
    __shared__ double lds[kWavesPerBlock];
    double s = 0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if ((flags[i] & 3) == 3) s += a[i] * b[i];
    double v[1] = {s};
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
}

__global__ __launch_bounds__(kBlock) void k_invert_free(double *__restrict__ minv, const double *__restrict__ d,
                                                        const uint8_t *__restrict__ flags, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        minv[i] = (flags[i] & 2) ? 1.0 / d[i] : 0.0;               // :351-352
}

__global__ __launch_bounds__(kBlock) void k_axpby(double *__restrict__ o, double a, const double *__restrict__ x, double b,
                                                  const double *__restrict__ y, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        o[i] = a * x[i] + b * y[i];
}

__global__ __launch_bounds__(kBlock) void k_scale(double *__restrict__ o, double a, const double *__restrict__ x, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) o[i] = a * x[i];
}

__global__ __launch_bounds__(kBlock) void k_mask_free(double *__restrict__ x, const uint8_t *__restrict__ flags, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (!(flags[i] & 2)) x[i] = 0.0;
}

}  // namespace pcg
