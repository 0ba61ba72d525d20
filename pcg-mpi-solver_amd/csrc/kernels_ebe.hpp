// Matrix-free (element-by-element) operator kernels (reference: pcg_solver.py:277-280 + :300).
#pragma once
#include "hip_common.hpp"

namespace pcg {

// ------------------------------------------------------------------------------------------------
// Matrix-free operator (the reference's element-by-element form, pcg_solver.py:277-280 + :300).
// One thread per element of ONE colour (no two elements of a launch share a node -> plain
// read-modify-write of y, no atomics, summation order = colour order).  All lanes of a wave work on
// the same pattern type, so Ke[a][b] is wave-uniform: it is fetched with scalar loads into SGPRs and
// used as the scalar operand of v_fma_f64 - no LDS, no per-lane copy of the 24x24 matrix.  Four
// output rows are accumulated at a time (independent FMA chains).
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(kBlock) void k_ebe(const int *__restrict__ dof, const unsigned *__restrict__ sgn,
                                                const double *__restrict__ ck, const double *__restrict__ ke,
                                                const double *__restrict__ x, double *__restrict__ y, int64_t ne,
                                                int64_t e_lo, int64_t e_hi)
{
    const int64_t e = e_lo + blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (e >= e_hi) return;
    int d[ND];
    double u[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) d[a] = __builtin_nontemporal_load(dof + (size_t)a * ne + e);
    const unsigned sg = __builtin_nontemporal_load(sgn + e);
    const double c = __builtin_nontemporal_load(ck + e);
#pragma unroll
    for (int b = 0; b < ND; ++b) {
        double v = x[d[b]];                                  // :277 gather
        if ((sg >> b) & 1u) v = -v;                          // :278
        u[b] = c * v;                                        // :279 Ck * U
    }
    static_assert(ND % 4 == 0, "ND must be a multiple of 4");
#pragma unroll
    for (int a0 = 0; a0 < ND; a0 += 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < ND; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fma(ke[(a0 + i) * ND + b], u[b], acc[i]);   // :279 Ke @ (.)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double o = acc[i];
            if ((sg >> (a0 + i)) & 1u) o = -o;               // :280
            y[d[a0 + i]] += o;                               // :300 (conflict-free inside a colour)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// hex8 chunks, second form (k_ebe_hex).  Same algorithm, same summation order per node and the same host-side chunk
// structures as the round-1 kernel (k_ebe_chunk, removed in round 3); what changed is how a workgroup gets to its arithmetic and back:
//   * per-LAUNCH tables (HexTab): block b finds its header, node list (padded to MAXN entries, -1 = none) and element
//     slots at fixed strides of b - no chunk-id list, no header -> offsets dependency: the node ids, the element data
//     and the header are three independent loads issued together, the x gather is the only dependent round trip
//     (round 1: chunk id -> header -> node ids -> x);
//   * element slot of thread t, copy j: ((t >> 6) * EPT + j) * 64 + (t & 63): a wave owns EPT consecutive 64-slot runs,
//     so slot order = (wave, j) order, and the LDS accumulation runs wave after wave (each wave its sub-colours in
//     ascending order, LDS operations of one wave are ordered) with ONE block barrier per wave instead of one per
//     sub-colour: 4 instead of 8-10 per chunk, same order of additions;
//   * a sign is an XOR of the sign bit (shift, and, xor) instead of compare + select + two moves.
// EPT / NPT (tile nodes per thread) / LB (blocks per CU asked of the register allocator) are template parameters so that
// the occupancy trade-off can be measured (tools/ebe_lab.py).
// ------------------------------------------------------------------------------------------------
struct HexTab {
    const int4 *hdr;              // per chunk: n_nodes, n_sub, ke index, any sign bit set
    const int *nodes;             // [n][MAXN]  node id, -1 = padding
    const int *dst;               // [n][MAXN]  >= 0: y offset (exclusive node); < 0: -(boundary slot + 1)
    const unsigned short *tslot;  // [n][MAXN]  bits 0..9 slot in the LDS tile; bits 12..14: dof 0..2 of the node is owned and free
                                  //            (the weight of the fused p.Ap, patched in by upload_masks: no flag loads in the kernel)
    const unsigned short *lid;    // [n][8][CE]
    const double *ck;             // [n][CE]
    const unsigned *sgn;          // [n][CE]    24 sign bits, sub-colour in bits 24..31 (255 = padding slot)
    int xcd;                      // chunk = xcd_chunk(workgroup) instead of the workgroup index
    int flags;                    // k_ebe_hexs: bit 0 = ordered adds by block barriers (round 3) instead of tickets (round 4)
};

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i -> XCD i % 8).  Consecutive chunks are spatial neighbours and share
// the nodes on their common faces: with this mapping an XCD works through RUNS of consecutive chunks, so a neighbour's x lines are
// in the XCD's own L2.  mode 1: XCD k takes the k-th contiguous eighth of the list; mode G >= 2: runs of G chunks dealt round-robin
// (same locality inside a run, and the XCDs stay balanced where the cost of a chunk drifts along the list); 0: chunk = workgroup.
__device__ __forceinline__ int xcd_chunk(int bid, int n, int mode)
{
    if (mode == 0) return bid;
    const int xcd = bid & 7, idx = bid >> 3;
    if (mode == 1) {
        const int q = n >> 3, r = n & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int grp = idx / mode, within = idx - grp * mode;
    if ((grp + 1) * 8 * mode > n) return bid;                // the last, partial super-group keeps its order
    return (grp * 8 + xcd) * mode + within;
}

__device__ __forceinline__ double flip_sign(double v, unsigned sg, int b)
{
    const unsigned long long m = (unsigned long long)((sg >> b) & 1u) << 63;
    return __longlong_as_double(__double_as_longlong(v) ^ (long long)m);
}

// ACCM: how a lane adds its 24 outputs into the LDS y tile.  0: read - add - write in two batches of 12 (the signs are
// applied beforehand, outside the serial part, and the wave whose turn it is runs at raised priority: its few VALU adds
// must not queue behind the other workgroups' FMA streams while three waves wait at the barrier).  1: ds_add_f64 - the
// LDS unit adds in place, the serial part of a wave is 24 * EPT LDS instructions and no VALU work at all.  The order of
// additions per node is the same in both modes (wave after wave, sub-colour after sub-colour): bit-reproducible.
template <int EPT, int NPT, int LB, bool DOT, int ACCM>
__global__ __launch_bounds__(kChunkThreads, LB) void k_ebe_hex(HexTab T, const double *__restrict__ ke_col, const double *__restrict__ x,
                                                               double *__restrict__ y, double *__restrict__ buf,
                                                               const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                               long long dot_lo)
{
    constexpr int CE = kChunkThreads * EPT, MAXN = kChunkThreads * NPT, ND = 24;
    __shared__ double xs[3 * MAXN];
    __shared__ double ys[3 * MAXN];
    const int b = xcd_chunk(blockIdx.x, gridDim.x, T.xcd), wave = threadIdx.x >> 6;
    const int4 h = T.hdr[b];
    // ---- three independent groups of loads: element slots, node table, (header above) --------------------------
    unsigned sg[EPT];
    double c[EPT];
    int l3[EPT][8];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const size_t slot = (size_t)b * CE + (wave * EPT + j) * 64 + (threadIdx.x & 63);
        sg[j] = ntload(T.sgn + slot);
        c[j] = ntload(T.ck + slot);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            l3[j][k] = 3 * (int)__builtin_nontemporal_load(T.lid + ((size_t)b * 8 + k) * CE + (wave * EPT + j) * 64 + (threadIdx.x & 63));
    }
    int g[NPT], dst[NPT], sl3[NPT], wmask[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const size_t n = (size_t)b * MAXN + threadIdx.x + j * kChunkThreads;
        g[j] = ntload(T.nodes + n);
        dst[j] = ntload(T.dst + n);
        const int ts = (int)__builtin_nontemporal_load(T.tslot + n);
        sl3[j] = 3 * (ts & 0x3ff);
        wmask[j] = ts >> 12;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (g[j] >= 0) {
            const double *xp = x + 3 * (size_t)g[j];
            const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
            xs[sl3[j]] = x0; xs[sl3[j] + 1] = x1; xs[sl3[j] + 2] = x2;
            ys[sl3[j]] = 0.0; ys[sl3[j] + 1] = 0.0; ys[sl3[j] + 2] = 0.0;
        }
    __syncthreads();
    const double *K = ke_col + (size_t)h.z * ND * ND;
    double acc[EPT][ND];
#pragma unroll
    for (int j = 0; j < EPT; ++j)
#pragma unroll
        for (int a = 0; a < ND; ++a) acc[j][a] = 0.0;
    // h.w: some element of this chunk has a sign bit set (ElemList_SignVector, :278,:280); chunks without any - every chunk
    // of a mesh whose patterns are in reference orientation - skip the two sign passes (block-uniform branch)
    auto contract = [&](auto with_signs) {
        constexpr bool SIG = decltype(with_signs)::value;
#pragma unroll
        for (int bb = 0; bb < ND; ++bb) {
            double u[EPT];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const double xv = xs[l3[j][bb / 3] + bb % 3];                                            // :277 gather
                u[j] = c[j] * (SIG ? flip_sign(xv, sg[j], bb) : xv);                                     // :278-279 sign, Ck
            }
            {
#pragma unroll
                for (int a = 0; a < ND; ++a) {
                    const double k = K[bb * ND + a];                                                     // wave-uniform -> SGPR pair
#pragma unroll
                    for (int j = 0; j < EPT; ++j) acc[j][a] = fma(k, u[j], acc[j][a]);                   // :279 Ke @ (.)
                }
            }
        }
        if constexpr (SIG) {
#pragma unroll
            for (int j = 0; j < EPT; ++j)
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[j][a] = flip_sign(acc[j][a], sg[j], a);                 // :280 (every wave at once)
        }
    };
    if (h.w) contract(std::true_type());
    else contract(std::false_type());
    // ---- LDS-staged partial sums, wave after wave (slot order = sub-colour order) ----------------------------------
    for (int w = 0; w < kWavesPerBlock; ++w) {
        if (wave == w) {
            if constexpr (ACCM == 0) __builtin_amdgcn_s_setprio(3);
            for (int s = 0; s < h.y; ++s) {
#pragma unroll
                for (int j = 0; j < EPT; ++j)
                    if ((int)(sg[j] >> 24) == s) {
                        if constexpr (ACCM == 1) {
#pragma unroll
                            for (int a = 0; a < ND; ++a)                                                 // :300, added by the LDS unit
                                __hip_atomic_fetch_add(&ys[l3[j][a / 3] + a % 3], acc[j][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
#pragma unroll
                            for (int q0 = 0; q0 < ND; q0 += 12) {
                                double old[12];
#pragma unroll
                                for (int q = 0; q < 12; ++q) { const int a = q0 + q; old[q] = ys[l3[j][a / 3] + a % 3]; }
#pragma unroll
                                for (int q = 0; q < 12; ++q) { const int a = q0 + q; ys[l3[j][a / 3] + a % 3] = old[q] + acc[j][a]; }   // :300
                            }
                        }
                    }
            }
            if constexpr (ACCM == 0) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (g[j] >= 0) {
            double *out = dst[j] >= 0 ? y + dst[j] : buf + 3 * (size_t)(-dst[j] - 1);
            const double y0 = ys[sl3[j]], y1 = ys[sl3[j] + 1], y2 = ys[sl3[j] + 2];
            out[0] = y0; out[1] = y1; out[2] = y2;
            if (DOT && dst[j] >= 0 && dst[j] >= dot_lo) {    // fused p.Ap.w (:487) on the dofs this chunk finalises
                if (wmask[j] & 1) dot += xs[sl3[j]] * y0;
                if (wmask[j] & 2) dot += xs[sl3[j] + 1] * y1;
                if (wmask[j] & 4) dot += xs[sl3[j] + 2] * y2;
            }
        }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}



// ------------------------------------------------------------------------------------------------
// hex8 chunks of 512 elements in TWO sequential passes of 256 (k_ebe_hexs).  The geometry of the 512-element chunk (an
// 8x8x8 cell: 729 tile nodes, 386 of them shared with other chunks = 0.75 boundary slots per element, against 1.0 for
// the 8x8x4 cells of the 256-element chunks) with the register footprint of one element per thread (96 VGPRs): the two
// halves of the cell are contracted and accumulated one after the other into the SAME LDS y tile, so the plane between
// them never leaves the workgroup and the tile is staged / written out once.  Why bytes matter here: at 10 M dof an apply
// moves ~0.65 GB with 256-element chunks (0.49 GB with 512), of which a third is the boundary-slot round trip; the
// element kernel + shared-node sums run within 1.4x of what that traffic costs at the stream rate.
// Slot of thread t in pass p: p * 256 + t, i.e. slot order = (pass, wave) order = sub-colour order: same sums as k_ebe_hex.
// ------------------------------------------------------------------------------------------------
template <int LB, bool DOT, int ACCM>
__global__ __launch_bounds__(kChunkThreads, LB) void k_ebe_hexs(HexTab T, const double *__restrict__ ke_col, const double *__restrict__ x,
                                                                double *__restrict__ y, double *__restrict__ buf,
                                                                const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                                long long dot_lo)
{
    constexpr int SEQ = 2, NPT = 3, CE = kChunkThreads * SEQ, MAXN = kChunkThreads * NPT, ND = 24;
    __shared__ double xs[3 * MAXN];
    __shared__ double ys[3 * MAXN];
    __shared__ int turn;
    const int b = xcd_chunk(blockIdx.x, gridDim.x, T.xcd), wave = threadIdx.x >> 6;
    const int4 h = T.hdr[b];
    if (threadIdx.x == 0) turn = 0;
    unsigned sg;
    double c;
    int l3[8];
    auto load_elem = [&](int ps) {
        const size_t slot = (size_t)b * CE + ps * kChunkThreads + threadIdx.x;
        sg = ntload(T.sgn + slot);
        c = ntload(T.ck + slot);
#pragma unroll
        for (int k = 0; k < 8; ++k) l3[k] = 3 * (int)__builtin_nontemporal_load(T.lid + ((size_t)b * 8 + k) * CE + ps * kChunkThreads + threadIdx.x);
    };
    load_elem(0);
    int g[NPT], dst[NPT], sl3[NPT], wmask[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const size_t n = (size_t)b * MAXN + threadIdx.x + j * kChunkThreads;
        g[j] = ntload(T.nodes + n);
        dst[j] = ntload(T.dst + n);
        const int ts = (int)__builtin_nontemporal_load(T.tslot + n);
        sl3[j] = 3 * (ts & 0x3ff);
        wmask[j] = ts >> 12;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (g[j] >= 0) {
            const double *xp = x + 3 * (size_t)g[j];
            const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
            xs[sl3[j]] = x0; xs[sl3[j] + 1] = x1; xs[sl3[j] + 2] = x2;
            ys[sl3[j]] = 0.0; ys[sl3[j] + 1] = 0.0; ys[sl3[j] + 2] = 0.0;
        }
    __syncthreads();
    const double *K = ke_col + (size_t)h.z * ND * ND;
#pragma unroll
    for (int ps = 0; ps < SEQ; ++ps) {
        double acc[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) acc[a] = 0.0;
        auto contract = [&](auto with_signs) {
            constexpr bool SIG = decltype(with_signs)::value;
#pragma unroll
            for (int bb = 0; bb < ND; ++bb) {
                const double xv = xs[l3[bb / 3] + bb % 3];                                               // :277 gather
                const double u = c * (SIG ? flip_sign(xv, sg, bb) : xv);                                 // :278-279 sign, Ck
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[a] = fma(K[bb * ND + a], u, acc[a]);                    // :279 Ke @ (.)
            }
            if constexpr (SIG) {
#pragma unroll
                for (int a = 0; a < ND; ++a) acc[a] = flip_sign(acc[a], sg, a);                          // :280
            }
        };
        if (h.w) contract(std::true_type());
        else contract(std::false_type());
        const unsigned my_colour = sg >> 24;
        const int a0[8] = {l3[0], l3[1], l3[2], l3[3], l3[4], l3[5], l3[6], l3[7]};
        if (ps + 1 < SEQ) load_elem(ps + 1);                 // the other half's slots arrive under this accumulation
        auto add_mine = [&]() {
                if constexpr (ACCM == 0) __builtin_amdgcn_s_setprio(3);
                for (int s = 0; s < h.y; ++s)
                    if ((int)my_colour == s) {
                        if constexpr (ACCM == 1) {
#pragma unroll
                            for (int a = 0; a < ND; ++a)                                                 // :300, added by the LDS unit
                                __hip_atomic_fetch_add(&ys[a0[a / 3] + a % 3], acc[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
#pragma unroll
                            for (int q0 = 0; q0 < ND; q0 += 12) {
                                double old[12];
#pragma unroll
                                for (int q = 0; q < 12; ++q) { const int a = q0 + q; old[q] = ys[a0[a / 3] + a % 3]; }
#pragma unroll
                                for (int q = 0; q < 12; ++q) { const int a = q0 + q; ys[a0[a / 3] + a % 3] = old[q] + acc[a]; }
                            }
                        }
                    }
                if constexpr (ACCM == 0) __builtin_amdgcn_s_setprio(0);
        };
        if (T.flags & 1) {
            for (int w = 0; w < kWavesPerBlock; ++w) {
                if (wave == w) add_mine();
                __syncthreads();
            }
        } else {
            // ticket ps * 4 + wave (round 4, as k_ebe_mixed): the holder adds and hands over - the LDS unit serves a wave's operations
            // in issue order - then goes straight on with the next pass's contraction instead of idling at a block barrier through the
            // other three waves' adds.  Same order of additions per node.
            const int ticket = ps * kWavesPerBlock + wave;
            while (__hip_atomic_load(&turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket) __builtin_amdgcn_s_sleep(1);
            add_mine();
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&turn, ticket + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (!(T.flags & 1)) __syncthreads();                     // every add is in before the tile is written out
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (g[j] >= 0) {
            double *out = dst[j] >= 0 ? y + dst[j] : buf + 3 * (size_t)(-dst[j] - 1);
            const double y0 = ys[sl3[j]], y1 = ys[sl3[j] + 1], y2 = ys[sl3[j] + 2];
            out[0] = y0; out[1] = y1; out[2] = y2;
            if (DOT && dst[j] >= 0 && dst[j] >= dot_lo) {    // fused p.Ap.w (:487) on the dofs this chunk finalises
                if (wmask[j] & 1) dot += xs[sl3[j]] * y0;
                if (wmask[j] & 2) dot += xs[sl3[j] + 1] * y1;
                if (wmask[j] & 4) dot += xs[sl3[j] + 2] * y2;
            }
        }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}


// ------------------------------------------------------------------------------------------------
// Pattern types other than the full hex8 class (hanging-node octree patterns with up to 32 nodes, and patterns with fewer
// than 8): k_ebe_rows.  A chunk is 64 elements - ONE per lane - and the four waves of the workgroup each contract a
// quarter of the output rows (whole local nodes: NDP/4 = 6 / 12 / 18 / 24 rows) of the same 64 elements:
//   * a lane carries NDP/4 accumulators instead of NDP (12 instead of 48 for the 13-node transition cells): 5-8 waves
//     per SIMD instead of 2, and a workgroup's critical path is a quarter of the element's contraction;
//   * four times as many workgroups for the same elements.  On the two-level octree mesh of bench.py (4 608 transition
//     cells among 1.1 M hex8 cells) the round-1 kernel ran 18 workgroups for 47.6 us - more than the 33 us the 1.1 M
//     hex8 cells took - because each of them streamed a 48 x 48 Ke through one wave per SIMD with nothing to hide the
//     scalar-load waits behind;
//   * Ke is laid out per wave (ke_rows: wave, column, row-in-wave), so a wave's slice of a column is one contiguous
//     scalar load; the wave index is read with readfirstlane so the loads stay scalar.
// Gather, signs, Ck, LDS accumulation (wave after wave, sub-colour after sub-colour, ds_add_f64) and the exclusive /
// shared write-out are those of k_ebe_hex; every wave gathers all NDP inputs (the tile is in LDS, the redundancy is 4 LDS
// reads instead of 1 per input).
// ------------------------------------------------------------------------------------------------
// KLDS (round 3): the workgroup copies the pattern's Ke (NDP x NDP doubles in the per-wave layout, 18 / 41 / 74 KB for 16 / 24 / 32
// nodes) into LDS once, with coalesced vector loads, and the contraction reads its coefficients from there (the lanes of a wave
// read ONE address: broadcast ds_read_b128).  Through scalar loads a 41 KB matrix does not stay in the 16 KB scalar cache: every
// one of the 72 columns was an L2 round trip (~0.5 us) with nothing to hide it behind - 35 us for 12.7 k elements on the 1 M-dof
// octree mesh, against 30 us for 229 k hex8 elements (whose 4.6 KB Ke does stay cached; they keep the SGPR path).
template <int NNP, bool DOT, bool KLDS = false>
__global__ __launch_bounds__(kChunkThreads) void k_ebe_rows(
    const int *__restrict__ chunk_list, const int4 *__restrict__ hdr, const int *__restrict__ nodes, const int *__restrict__ dstl,
    const unsigned short *__restrict__ tslot, const unsigned short *__restrict__ lid, const double *__restrict__ ck, const unsigned *__restrict__ sgn,
    const double *__restrict__ ke_rows, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ buf,
    const uint8_t *__restrict__ flags, double *__restrict__ partials, long long dot_lo)
{
    constexpr int NPT = kChunkMaxNodes / kChunkThreads;      // tile nodes per thread (3)
    constexpr int CE = 64, NDP = 3 * NNP, RPW = NDP / 4, NPW = NNP / 4, W = NDP / 32 + 1;
    static_assert(RPW % 3 == 0, "a wave owns whole local nodes");
    __shared__ double xs[3 * kChunkMaxNodes];
    __shared__ double ys[3 * kChunkMaxNodes];
    const int chunk = chunk_list[blockIdx.x];
    const int4 h = hdr[2 * chunk];                           // node_off, n_nodes, n_sub, ke index in class
    const int4 h2 = hdr[2 * chunk + 1];                      // chunk index in class, nd, class, -
    const int kci = h2.x, nd = h2.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // element `lane` of the chunk (every wave loads it: same cache lines)
    unsigned sg[W];
    int l3[NNP], lown[NPW];
#pragma unroll
    for (int w = 0; w < W; ++w) sg[w] = ntload(sgn + ((size_t)kci * W + w) * CE + lane);
    const double c = ntload(ck + (size_t)kci * CE + lane);
#pragma unroll
    for (int k = 0; k < NNP; ++k) l3[k] = 3 * (int)__builtin_nontemporal_load(lid + ((size_t)kci * NNP + k) * CE + lane);
#pragma unroll
    for (int k = 0; k < NPW; ++k) lown[k] = 3 * (int)__builtin_nontemporal_load(lid + ((size_t)kci * NNP + wave * NPW + k) * CE + lane);
    int dst[NPT], sl3[NPT];
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = threadIdx.x + j * kChunkThreads;
        int g = -1;
        dst[j] = 0;
        sl3[j] = 0;
        if (n < h.y) {
            g = ntload(nodes + h.x + n); dst[j] = ntload(dstl + h.x + n);
            sl3[j] = 3 * (int)__builtin_nontemporal_load(tslot + h.x + n);
        }
        if (g >= 0) {
            const double *xp = x + 3 * (size_t)g;
            const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
            xs[sl3[j]] = x0; xs[sl3[j] + 1] = x1; xs[sl3[j] + 2] = x2;
            ys[sl3[j]] = 0.0; ys[sl3[j] + 1] = 0.0; ys[sl3[j] + 2] = 0.0;
        }
    }
    __syncthreads();
    double acc[RPW];
#pragma unroll
    for (int a = 0; a < RPW; ++a) acc[a] = 0.0;
    if constexpr (KLDS) {
        extern __shared__ __align__(16) double ks[];         // [wave][column b][row a], the layout of ke_rows
        const double2 *src = reinterpret_cast<const double2 *>(ke_rows + (size_t)h.w * NDP * NDP);
        double2 *dst2 = reinterpret_cast<double2 *>(ks);
        for (int i = threadIdx.x; i < NDP * NDP / 2; i += kChunkThreads) dst2[i] = src[i];
        __syncthreads();
        const double *K = ks + (size_t)wave * NDP * RPW;
#pragma unroll 4
        for (int b = 0; b < NDP; ++b) {
            if (b < nd) {
                const double u = c * flip_sign(xs[l3[b / 3] + b % 3], sg[b >> 5], b & 31);               // :277-279
#pragma unroll
                for (int a = 0; a < RPW; ++a) acc[a] = fma(K[b * RPW + a], u, acc[a]);                   // :279 Ke @ (.)
            }
        }
    } else {
        const double *K = ke_rows + ((size_t)h.w * 4 + wave) * NDP * RPW;   // this wave's rows: [column b][row a]
#pragma unroll
        for (int b = 0; b < NDP; ++b) {
            if (b < nd) {                                    // nd is block-uniform: scalar compare, loops stay unrolled
                const double u = c * flip_sign(xs[l3[b / 3] + b % 3], sg[b >> 5], b & 31);               // :277-279
#pragma unroll
                for (int a = 0; a < RPW; ++a) acc[a] = fma(K[b * RPW + a], u, acc[a]);                   // :279 Ke @ (.)
            }
        }
    }
    const int row0 = wave * RPW;                             // global row of acc[0]
#pragma unroll
    for (int a = 0; a < RPW; ++a) {                          // :280 (dynamic bit position: the wave index is not a constant)
        const int r = row0 + a;
        acc[a] = flip_sign(acc[a], sg[W == 1 ? 0 : (r >> 5)], r & 31);
    }
    const int my_colour = (int)(sg[W - 1] >> 24);
    for (int w = 0; w < kWavesPerBlock; ++w) {
        if (wave == w)
            for (int s = 0; s < h.z; ++s)
                if (my_colour == s) {
#pragma unroll
                    for (int a = 0; a < RPW; ++a)
                        if (row0 + a < nd)                   // padded rows alias local node 0: never add them
                            __hip_atomic_fetch_add(&ys[lown[a / 3] + a % 3], acc[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :300
                }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = threadIdx.x + j * kChunkThreads;
        if (n < h.y) {
            double *out = dst[j] >= 0 ? y + dst[j] : buf + 3 * (size_t)(-dst[j] - 1);
            const double y0 = ys[sl3[j]], y1 = ys[sl3[j] + 1], y2 = ys[sl3[j] + 2];
            out[0] = y0; out[1] = y1; out[2] = y2;
            if (DOT && dst[j] >= 0 && dst[j] >= dot_lo) {    // fused p.Ap.w (:487) on the dofs this chunk finalises
                const uint8_t *fp = flags + dst[j];
                if ((fp[0] & 3) == 3) dot += xs[sl3[j]] * y0;
                if ((fp[1] & 3) == 3) dot += xs[sl3[j] + 1] * y1;
                if ((fp[2] & 3) == 3) dot += xs[sl3[j] + 2] * y2;
            }
        }
    }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Hanging-node pattern types WITHOUT a node tile, on the matrix cores: k_ebe_direct (round 3).
// A tile pays when the elements of a chunk share nodes (8 x 8 x 8 hex8 cells: 729 nodes for 512 elements).  The elements of
// ONE hanging-node pattern type do not: they lie scattered along the transition shells of an octree mesh, their neighbours are
// of other types.  Measured on the graded octree mesh (planner statistics, PCG_EBE_STATS=1): 8.9 / 15.3 tile nodes per element
// for 11.3 / 18.4 incidences in the 16- / 24-node classes (a fifth re-used inside the chunk), every tile node is shared with OTHER chunks and leaves through the
// boundary slots anyway, and the tile's node limit keeps the chunks at 55 / 42 of 64 elements.  k_ebe_rows spent its time in the
// chain chunk -> header -> node lists -> x -> LDS -> barrier -> ... -> LDS -> barrier -> y at 2 - 4 workgroups per CU (128 us per
// class at 10 M dof for work that takes the arithmetic units 15 us; neither Ke from LDS, nor the matrix cores on the same tile
// skeleton, nor dropping the ordered LDS accumulation changed that: profiles/r03_octree_rows_kernels_sessionO.md).
// Here a chunk is 64 elements of one type and nothing else:
//   * per element-node incidence the planner stores the node id and the destination (ebe.cpp: an incidence of a node that no
//     other element touches goes straight to y, every other one to its own boundary slot; k_ebe_shared sums the slots of a node
//     in a fixed order) - the chunk's part of `nodes` / `dst` is (local node, element slot)-major, 64 consecutive elements per
//     local node: coalesced index loads;
//   * Y(NDP x 64) = Ke(NDP x NDP) . U(NDP x 64) on v_mfma_f64_16x16x4_f64: wave w owns the elements 16 w .. 16 w + 15,
//       A (16 x 4): lane l holds Ke[16 mt + col][4 ks + g]   (col = l & 15, g = l >> 4)  - LDS copy of Ke (one ds_read_b64 per
//                   lane feeds a whole 16 x 4 tile; a broadcast read or a scalar load feeds ONE coefficient)
//       B (4 x 16): lane l holds u[dof 4 ks + g] of element col                          - gathered from x, sign and Ck applied
//       D (16x16):  lane l, register r holds y[dof 16 mt + g + 4 r] of element col
//     so a lane gathers and scatters the SAME dofs {g, g + 4, g + 8, ...} of its element;
//   * no LDS besides Ke (18 / 41 / 74 KB for 16 / 24 / 32 nodes), one barrier, no atomics: every output has its own address.
// The f64 matrix cores of gfx950 run at the rate of the vector FMAs (78.6 TFLOP/s): the point is the operand supply and the
// shape of the kernel, not the arithmetic peak.
// ------------------------------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NNP, bool DOT>
__global__ __launch_bounds__(kChunkThreads, (NNP <= 16 ? 4 : NNP <= 24 ? 3 : 2)) void k_ebe_direct(
    const int *__restrict__ chunk_list, const int4 *__restrict__ hdr, const int *__restrict__ nodes, const int *__restrict__ dstl,
    const double *__restrict__ ck, const unsigned *__restrict__ sgn, const double *__restrict__ ke_col, const double *__restrict__ x,
    double *__restrict__ y, double *__restrict__ buf, const uint8_t *__restrict__ flags, double *__restrict__ partials, long long dot_lo,
    int xcd)
{
    constexpr int CE = 64, NDP = 3 * NNP, MT = (NDP + 15) / 16, KS = NDP / 4, W = NDP / 32 + 1;
    static_assert(NDP % 4 == 0, "whole k-steps");
    extern __shared__ __align__(16) double ks_lds[];         // Ke, column-major: ks_lds[b * NDP + a] = Ke[a][b]
    const int chunk = chunk_list[xcd_chunk(blockIdx.x, gridDim.x, xcd)];
    const int4 h = hdr[2 * chunk];                           // entry offset, -, -, ke index in class
    const int4 h2 = hdr[2 * chunk + 1];                      // chunk index in class, nd, class, 16-element tiles in use
    const int kci = h2.x, nd = h2.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int e = 16 * wave + col;                           // element slot of this lane (four lanes per element)
    {                                                        // Ke -> LDS (L2 hits: every chunk of the type reads the same matrix)
        const double2 *src = reinterpret_cast<const double2 *>(ke_col + (size_t)h.w * NDP * NDP);
        double2 *dst2 = reinterpret_cast<double2 *>(ks_lds);
#pragma unroll 4
        for (int i = threadIdx.x; i < NDP * NDP / 2; i += kChunkThreads) dst2[i] = src[i];
    }
    unsigned sg[W];
#pragma unroll
    for (int w = 0; w < W; ++w) sg[w] = ntload(sgn + ((size_t)kci * W + w) * CE + e);
    const double c = ntload(ck + (size_t)kci * CE + e);      // 0 for padding slots
    auto sign_bit = [&](int d) -> unsigned {                 // d is not a compile-time constant (g): select the word
        unsigned word = sg[0];
#pragma unroll
        for (int w = 1; w < W; ++w) word = (d >> 5) == w ? sg[w] : word;
        return (word >> (d & 31)) & 1u;
    };
    const int *my_nodes = nodes + h.x + e, *my_dst = dstl + h.x + e;
    double xv[KS];                                           // x at the dofs g + 4 m of this lane's element (signed, times Ck)
#pragma unroll
    for (int m = 0; m < KS; ++m) {
        const unsigned d = g + 4 * m, k = d / 3;
        int node = -1;
        if ((int)d < nd) node = ntload(my_nodes + k * CE);   // -1: padding slot
        xv[m] = node >= 0 ? x[3 * (size_t)node + (d - 3 * k)] : 0.0;                                      // :277 gather
    }
    __syncthreads();
    d4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = d4_t{0.0, 0.0, 0.0, 0.0};
    if (wave < h2.w) {                                       // 16-element tiles in use (the last chunk of a type may be short)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = 4 * ks + g;
            const double u = c * (sign_bit(d) ? -xv[ks] : xv[ks]);                                        // :278-279 sign, Ck
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int a = 16 * mt + col;                 // rows beyond NDP (last tile of 24 / 72): zero
                const double kv = (16 * mt + 15 < NDP || a < NDP) ? ks_lds[d * NDP + a] : 0.0;
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv, u, acc[mt], 0, 0, 0);                  // :279 Ke @ (.)
            }
        }
    }
    double dot = 0.0;
#pragma unroll
    for (int m = 0; m < KS; ++m) {
        const unsigned d = g + 4 * m, k = d / 3;
        if ((int)d < nd) {
            const int dc = ntload(my_dst + k * CE);          // >= 0: y offset of the node; < 0: -(boundary slot + 1); INT_MIN: padding
            if (dc != INT_MIN) {
                const double a = acc[m / 4][m % 4];
                const double o = sign_bit(d) ? -a : a;                                                    // :280
                const int comp = d - 3 * k;
                double *out = dc >= 0 ? y + dc + comp : buf + 3 * (size_t)(-dc - 1) + comp;
                *out = o;                                                                                  // :300 (summed by k_ebe_shared)
                if (DOT && dc >= 0 && dc + comp >= dot_lo && (flags[dc + comp] & 3) == 3) dot += x[dc + comp] * o;   // fused p.Ap.w (:487)
            }
        }
    }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Mixed-type chunks (round 4; EbeMixedHost in pcg_internal.hpp): k_ebe_mixed.
// A chunk is a run of the GLOBAL Morton order of the elements - every node-blocked pattern type together - so the elements
// around a node sit in one workgroup whatever their type, the node is summed in the chunk's LDS tile and only the nodes on the
// surface of the run go through boundary slots (1 M-dof graded octree mesh: 816 k -> 314 k slots, 68 % -> 41 % of the nodes
// shared; the per-type chunks of round 3 sent EVERY node incidence of a hanging-node element through a slot).
//   [stage]  the chunk's nodes: x tile in, y tile zeroed (as k_ebe_hexs: per-launch tables at fixed strides of the workgroup)
//   [hex]    the elements of the mesh's main 8-node type: k_ebe_hexs' two passes of one element per thread, Ke as SGPR operands
//   [tiles]  every other element sits in a 16-element tile of ONE pattern type; a wave takes a tile:
//            Y(nd x 16) = Ke(nd x nd) . U(nd x 16) on v_mfma_f64_16x16x4_f64 with the A operand streamed from the type's
//            pre-permuted fragments (one coalesced 512-B load per instruction, L2-resident: 46 KB per 24-node type).  The
//            permutation (ebe.cpp build_mixed_types) makes lane (g, e) gather and scatter WHOLE nodes g, g + 4, ... of element e.
//            Four tiles are contracted side by side, then added into the y tile wave after wave (tile order), inside a tile
//            colour by colour (ds_add_f64): one order of additions per node, bit-reproducible.
//   [out]    exclusive nodes -> y, shared nodes -> their boundary slot (k_ebe_shared), fused p.Ap
// MTM: M-tiles the largest pattern type needs (3 J / 4 rounded up, J = node quartets): sizes the accumulators.
// ------------------------------------------------------------------------------------------------
struct MixTab {
    const int4 *hdr;              // per chunk 2 x int4: {n_nodes, hex sub-colours, hex slots in use, any hex sign bit}, {tiles, first tile, -, -}
    const int *nodes;             // [n][768]   node id, -1 = padding
    const int *dst;               // [n][768]   >= 0: y offset (exclusive node); < 0: -(boundary slot + 1)
    const unsigned short *tslot;  // [n][768]   bits 0..9 slot in the LDS tile; bits 12..14: dot weights (upload_masks)
    const unsigned short *lid;    // [n][8][512]
    const double *ck;             // [n][512]
    const unsigned *sgn;          // [n][512]   24 sign bits, sub-colour in bits 24..31 (255 = padding slot)
    const int2 *tinfo;            // per tile: nodes | node quartets << 8 | colours << 16 ; fragment offset / 64
    const unsigned short *tlid;   // [tiles][np][16]
    const double *tck;            // [tiles][16]
    const unsigned *tsgw;         // [tiles][4][16]   lane group g of element e: bit 3 j + c = sign of dof c of the element's node 4 j + g
    const unsigned char *tcol;    // [tiles][16]   tile-local colour, 255 = padding slot
    const unsigned char *tperm;   // [tiles][16]   dof order of the element: slot 3 l + c is component (tperm >> 2 c) & 3 of node l
    const double *frag;
    int np, xcd;
    int flags;                    // bit 0: ordered adds by barriers instead of tickets, bit 2: empty waves of a hex pass do not skip (A/B);
                                  // bits 4..6: development ablations (wrong results)
    const uint2 *hrec;            // k_ebe_mtile, hex tiles: [tiles][64] per lane (g, e): x = slot of node g | slot of node g + 4 << 16 in the LDS tile,
                                  //            y = sign bits of the lane's six dofs (bit 3 j + c), bits 8..13 the element's component order (tperm),
                                  //                bit 31 = the element exists
    const int *twait;             // k_ebe_mtile: [tiles] completed tiles of the chunk this tile's adds wait for
    unsigned long long *stamps;   // STAMP instantiation only (PCG_EBE_STAMPS=1, development): [workgroup][wave][16] shader-clock readings
};

typedef double d4m_t __attribute__((ext_vector_type(4)));

// one tile, J node quartets (compile time): acc[mt] = sum over the k-steps of A(ks, mt) . U(ks)
// The A fragments come from L2 (one coalesced 512-B load per matrix instruction).  They are requested D k-steps AHEAD of their
// instruction through an explicit ring of registers: left to itself the compiler issued ONE load, waited for it (vmcnt(0)) and ran
// ONE instruction - 27-60 dependent L2 round trips per tile, 21.7 k cycles per tile for 2-4 k cycles of matrix-core time (clock
// stamps, profiles/r04_stamps_sessionJ.log).
template <int J, int MTM>
__device__ __forceinline__ void mixed_tile_contract(const double *__restrict__ F, const double *xs, const int (&l3)[(4 * MTM) / 3], double c,
                                                    unsigned sw, int pw, int g, int nn, d4m_t (&acc)[MTM])
{
    constexpr int MT = (3 * J + 3) / 4, KS = 3 * J;
    constexpr int D = KS < kMixedFragAhead ? KS : kMixedFragAhead;   // k-steps of fragments in flight
    static_assert(MT <= MTM, "tile type larger than the kernel instantiation");
    double fr[D][MT];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fr[d][mt] = F[(size_t)(d * MT + mt) * 64];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool live = g < nn - 4 * j;                        // (nn is wave-uniform: a scalar subtraction, no per-lane node number)
        const double x0 = xs[l3[j] + (pw & 3)], x1 = xs[l3[j] + ((pw >> 2) & 3)], x2 = xs[l3[j] + ((pw >> 4) & 3)];   // the element's own dof order (:277)
        const double xv[3] = {x0, x1, x2};
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const int ks = 3 * j + cc;
            const double u = live ? c * flip_sign(xv[cc], sw, ks) : 0.0;                                    // :278-279 sign, Ck * U
            double a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = fr[ks % D][mt];
            if (ks + D < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fr[ks % D][mt] = F[(size_t)((ks + D) * MT + mt) * 64];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], u, acc[mt], 0, 0, 0);               // :279 Ke @ (.)
        }
    }
}

template <int MTM, bool DOT, bool STAMP = false>
__global__ __launch_bounds__(kChunkThreads, (MTM <= 4 ? 4 : 3)) void k_ebe_mixed(MixTab T, const double *__restrict__ ke_col,
                                                                                   const double *__restrict__ x, double *__restrict__ y,
                                                                                   double *__restrict__ buf, double *__restrict__ partials,
                                                                                   long long dot_lo)
{
    constexpr int NPT = 3, MAXN = kChunkThreads * NPT, CE = kMixedHexSlots, ND = 24, JM = (4 * MTM) / 3;
    __shared__ double xs[3 * MAXN];
    __shared__ double ys[3 * MAXN];
    // The sums a chunk forms for a node are ORDERED (bit-reproducible): hex section (pass, wave, sub-colour), then the tiles in
    // ascending order.  The order is kept by a ticket in LDS: the holder of ticket t adds, then hands over to t + 1 - the LDS unit
    // serves one wave's operations in issue order, so the hand-over follows the adds - and goes on with its NEXT contraction while
    // the others add (a block barrier per turn kept three of four waves idle through every add phase; T.flags bit 0 = that form).
    __shared__ int turn;
    const bool use_barriers = (T.flags & 1) != 0;
    auto take_turn = [&](int ticket) {
        while (__hip_atomic_load(&turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket) __builtin_amdgcn_s_sleep(1);
    };
    auto pass_turn = [&](int next) {
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&turn, next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const int b = xcd_chunk(blockIdx.x, gridDim.x, T.xcd), wave = threadIdx.x >> 6;
    // development: where a wave's lifetime goes - slot k of (workgroup, wave) <- shader clock (STAMP instantiation only)
    unsigned long long tile_acc[4] = {0, 0, 0, 0};
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0) T.stamps[((size_t)blockIdx.x * kWavesPerBlock + wave) * 16 + k] = t;
        }
    };
    stamp(0);
    const int4 h = T.hdr[2 * b], h2 = T.hdr[2 * b + 1];
    const int n_hex = h.z;
    if (threadIdx.x == 0) turn = 0;
    const int hex_tickets = n_hex > kChunkThreads ? 2 * kWavesPerBlock : (n_hex > 0 ? kWavesPerBlock : 0);   // the tiles' tickets follow
    unsigned sg = 0xff000000u;
    double c = 0.0;
    int l3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_elem = [&](int ps) {
        const size_t slot = (size_t)b * CE + ps * kChunkThreads + threadIdx.x;
        sg = ntload(T.sgn + slot);
        c = ntload(T.ck + slot);
#pragma unroll
        for (int k = 0; k < 8; ++k) l3[k] = 3 * (int)__builtin_nontemporal_load(T.lid + ((size_t)b * 8 + k) * CE + ps * kChunkThreads + threadIdx.x);
    };
    if (n_hex > 0) load_elem(0);
    // (two registers per tile node stay live to the write-out: the destination - INT_MIN = no node - and the slot word)
    int dst[NPT], ts[NPT];
    {
        int g[NPT];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const size_t n = (size_t)b * MAXN + threadIdx.x + j * kChunkThreads;
            g[j] = ntload(T.nodes + n);
            dst[j] = ntload(T.dst + n);
            ts[j] = (int)__builtin_nontemporal_load(T.tslot + n);
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (g[j] >= 0) {
                const int sl3 = 3 * (ts[j] & 0x3ff);
                const double *xp = x + 3 * (size_t)g[j];
                const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
                xs[sl3] = x0; xs[sl3 + 1] = x1; xs[sl3 + 2] = x2;
                ys[sl3] = 0.0; ys[sl3 + 1] = 0.0; ys[sl3 + 2] = 0.0;
            }
    }
    __syncthreads();
    stamp(1);
    // ---- hex section: two passes of one element per thread (k_ebe_hexs), skipped when empty ----------------------------------
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (n_hex > ps * kChunkThreads) {                        // block-uniform
          // (a wave whose 64 slots of this pass are all padding skips the contraction - a pass is rarely full on graded meshes:
          //  348 hex elements per chunk on the 10 M-dof octree mesh - but still takes its turn)
          const bool wave_has = (T.flags & 4) || n_hex > ps * kChunkThreads + wave * 64;      // (bit 2: no skipping, A/B)
          {
            double acc[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) acc[a] = 0.0;
            auto contract = [&](auto with_signs) {
                constexpr bool SIG = decltype(with_signs)::value;
#pragma unroll
                for (int bb = 0; bb < ND; ++bb) {
                    const double xv = xs[l3[bb / 3] + bb % 3];                                               // :277 gather
                    const double u = c * (SIG ? flip_sign(xv, sg, bb) : xv);                                 // :278-279 sign, Ck
#pragma unroll
                    for (int a = 0; a < ND; ++a) acc[a] = fma(ke_col[bb * ND + a], u, acc[a]);               // :279 Ke @ (.)
                }
                if constexpr (SIG) {
#pragma unroll
                    for (int a = 0; a < ND; ++a) acc[a] = flip_sign(acc[a], sg, a);                          // :280
                }
            };
            if (wave_has) {                                      // (padding slots carry colour 255: they never add)
                if (h.w) contract(std::true_type());
                else contract(std::false_type());
            }
            stamp(2 + 3 * ps);
            const unsigned my_colour = sg >> 24;
            const int a0[8] = {l3[0], l3[1], l3[2], l3[3], l3[4], l3[5], l3[6], l3[7]};
            if (ps == 0 && n_hex > kChunkThreads) load_elem(1);  // the other half's slots arrive under this accumulation
            auto add_hex = [&]() {
                for (int s = 0; s < h.y; ++s)
                    if ((int)my_colour == s) {
#pragma unroll
                        for (int a = 0; a < ND; ++a)                                                         // :300, added by the LDS unit
                            __hip_atomic_fetch_add(&ys[a0[a / 3] + a % 3], acc[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            };
            if (use_barriers) {
                for (int w = 0; w < kWavesPerBlock; ++w) {
                    if (wave == w) add_hex();
                    __syncthreads();
                }
            } else {                                             // ticket ps * 4 + wave: wave after wave, nobody else waits
                take_turn(ps * kWavesPerBlock + wave);
                stamp(3 + 3 * ps);
                add_hex();
                pass_turn(ps * kWavesPerBlock + wave + 1);
                stamp(4 + 3 * ps);
            }
          }
        }
    }
    // ---- tiles: four at a time (one per wave) on the matrix cores, added wave after wave ----------------------------------------
    const int n_tiles = (T.flags & 64) ? 0 : h2.x, lane = threadIdx.x & 63, lg = lane >> 4, le = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);    // (provably wave-uniform: the tile's header comes by scalar loads)
    int2 info_next = wave_u < n_tiles ? T.tinfo[(size_t)h2.y + wave_u] : make_int2(0, 0);
    stamp(8);
    for (int t0 = 0; t0 < n_tiles; t0 += kWavesPerBlock) {       // block-uniform
        unsigned long long tt0 = 0, tt1 = 0, tt2 = 0;
        if constexpr (STAMP) tt0 = __builtin_amdgcn_s_memtime();
        const int ti = t0 + wave_u;
        const bool have = ti < n_tiles;                          // wave-uniform
        const int2 info = info_next;                             // this tile's header was requested a round ago:
        if (ti + kWavesPerBlock < n_tiles) info_next = T.tinfo[(size_t)h2.y + ti + kWavesPerBlock];   // header -> fragments is the dependent chain
        d4m_t acc[MTM];
        int tl3[JM];
        unsigned tsw = 0u;
        int pw = 0 | 1 << 2 | 2 << 4;
        int nn = 0, ncol = 0, mycol = 255;
#pragma unroll
        for (int mt = 0; mt < MTM; ++mt) acc[mt] = d4m_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < JM; ++j) tl3[j] = 0;
        if (have) {
            const size_t tg = (size_t)h2.y + ti;
            nn = __builtin_amdgcn_readfirstlane(info.x & 255);
            const int J = __builtin_amdgcn_readfirstlane((info.x >> 8) & 255);
            ncol = __builtin_amdgcn_readfirstlane(info.x >> 16);
            const double *F = T.frag + (size_t)__builtin_amdgcn_readfirstlane(info.y) * 64 + lane;
            const double tc = T.tck[tg * 16 + le];
            mycol = (int)T.tcol[tg * 16 + le];
            pw = (int)T.tperm[tg * 16 + le];
            tsw = T.tsgw[(tg * 4 + lg) * 16 + le];
#pragma unroll
            for (int j = 0; j < JM; ++j)
                if (j < J) tl3[j] = 3 * (int)T.tlid[(tg * T.np + 4 * j + lg) * 16 + le];
            switch ((T.flags & 32) ? 0 : J) {                     // wave-uniform: straight-line code per size
            case 1: mixed_tile_contract<1, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 2: if constexpr (JM >= 2) mixed_tile_contract<2, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 3: if constexpr (JM >= 3) mixed_tile_contract<3, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 4: if constexpr (JM >= 4) mixed_tile_contract<4, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 5: if constexpr (JM >= 5) mixed_tile_contract<5, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 6: if constexpr (JM >= 6) mixed_tile_contract<6, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 7: if constexpr (JM >= 7) mixed_tile_contract<7, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 8: if constexpr (JM >= 8) mixed_tile_contract<8, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            default: break;
            }
        }
        auto add_tile = [&]() {
                for (int s = 0; s < ncol; ++s)
                    if (mycol == s) {
#pragma unroll
                        for (int j = 0; j < JM; ++j)
                            if (lg < nn - 4 * j) {
#pragma unroll
                                for (int cc = 0; cc < 3; ++cc) {
                                    const int q = 3 * j + cc;
                                    const double o = flip_sign(acc[q / 4][q % 4], tsw, q);                    // :280
                                    __hip_atomic_fetch_add(&ys[tl3[j] + ((pw >> 2 * cc) & 3)], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :300
                                }
                            }
                    }
        };
        if (use_barriers) {
            for (int w = 0; w < kWavesPerBlock; ++w) {
                if (wave == w && have && !(T.flags & 16)) add_tile();
                __syncthreads();
            }
        } else if (have) {
            if constexpr (STAMP) tt1 = __builtin_amdgcn_s_memtime();
            take_turn(hex_tickets + ti);
            if constexpr (STAMP) tt2 = __builtin_amdgcn_s_memtime();
            if (!(T.flags & 16)) add_tile();
            pass_turn(hex_tickets + ti + 1);
            if constexpr (STAMP) {                               // per wave: tiles taken, cycles to the end of the contraction, waiting, adding
                tile_acc[0] += 1; tile_acc[1] += tt1 - tt0; tile_acc[2] += tt2 - tt1; tile_acc[3] += __builtin_amdgcn_s_memtime() - tt2;
            }
        }
    }
    stamp(9);
    if (!use_barriers) __syncthreads();                          // every add is in before the tile is written out
    stamp(10);
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (dst[j] != INT_MIN) {
            const int sl3 = 3 * (ts[j] & 0x3ff), wmask = ts[j] >> 12;
            double *out = dst[j] >= 0 ? y + dst[j] : buf + 3 * (size_t)(-dst[j] - 1);
            const double y0 = ys[sl3], y1 = ys[sl3 + 1], y2 = ys[sl3 + 2];
            out[0] = y0; out[1] = y1; out[2] = y2;
            if (DOT && dst[j] >= 0 && dst[j] >= dot_lo) {        // fused p.Ap.w (:487) on the dofs this chunk finalises
                if (wmask & 1) dot += xs[sl3] * y0;
                if (wmask & 2) dot += xs[sl3 + 1] * y1;
                if (wmask & 4) dot += xs[sl3 + 2] * y2;
            }
        }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
    stamp(11);
    if constexpr (STAMP) {
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 4; ++k) T.stamps[((size_t)blockIdx.x * kWavesPerBlock + wave) * 16 + 12 + k] = tile_acc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Mixed-type chunks, every element on the matrix cores (round 4; EbeMixedHost::hex_tile_type): k_ebe_mtile.
// The hex section of k_ebe_mixed feeds its FMAs with scalar loads of Ke, and a dependent s_load_dwordx16 costs 157 - 214 cycles
// (tools/micro/smem_latency): a wave's pass of 576 FMAs (2.3 k cycles of issue) takes 11 - 12 k, and the ordered LDS sums of the eight
// (pass, wave) turns of a chunk, 2.3 k cycles each, are the chunk's critical path (clock stamps, profiles/r04_stamps_*).  Here the
// standard 8-node type runs in 16-element tiles like every other type:
//   * Ke as TWELVE matrix fragments held in registers for the whole workgroup (J = 2 node quartets: 6 k-steps x 2 M-tiles), no scalar
//     loads in the contraction; a tile is 12 v_mfma_f64_16x16x4_f64 = 768 cycles of the matrix pipe;
//   * the tiles are COLOUR-PURE (planner): the 16 elements of a tile share no node - six ds_add_f64 per tile, all lanes - and the
//     tiles of one colour need no order among themselves: a tile adds once the tiles of all colours before its own have completed
//     (`twait`), and counts itself in.  One fixed order of additions per node (colour after colour, then the other types' tiles in
//     ascending order): bit-reproducible;
//   * lane (g, e) of a tile handles the nodes g and g + 4 of element e: one 8-byte record (two slots of the x / y tile, six sign
//     bits), Ck, six LDS reads, six products, six adds.  The next tile's record is requested a tile ahead.
// The other types' tiles follow as in k_ebe_mixed (fragments streamed through the register ring).
// ------------------------------------------------------------------------------------------------
// NW (round 6): waves per workgroup.  A chunk's tiles are dealt to its waves round-robin, so more waves shorten every wave's chain of tiles
// (the launch lasts as long as ONE chunk where the chunks do not fill the GPU: 661 chunks x 4 waves = 2.6 waves per SIMD at 1 M dof);
// the tables keep their stride of 768 entries per chunk, thread t stages the entries t, t + 64 NW, ... below 768.  The ORDER of the
// additions at a node does not depend on NW (colour after colour, then the other types' tiles in ascending order): same y, bit for bit.
template <int MTM, bool DOT, bool STAMP = false, int NW = kWavesPerBlock>
__global__ __launch_bounds__(64 * NW, (NW >= 6 ? 5 : (MTM <= 4 ? 4 : 3))) void k_ebe_mtile(MixTab T, const double *__restrict__ x, double *__restrict__ y,
                                                                                            double *__restrict__ buf, double *__restrict__ partials,
                                                                                            long long dot_lo)
{
    constexpr int NT = 64 * NW, MAXN = kChunkThreads * 3, NPT = (MAXN + NT - 1) / NT, JM = (4 * MTM) / 3;
    __shared__ double xs[3 * MAXN];
    __shared__ double ys[3 * MAXN];
    __shared__ int turn;                                         // tiles of this chunk that have completed their adds
    const int b = xcd_chunk(blockIdx.x, gridDim.x, T.xcd), wave = threadIdx.x >> 6;
    unsigned long long tile_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0) T.stamps[((size_t)blockIdx.x * NW + wave) * 16 + k] = t;
        }
    };
    stamp(0);
    const int4 h = T.hdr[2 * b], h2 = T.hdr[2 * b + 1];
    if (threadIdx.x == 0) turn = 0;
    const int lane = threadIdx.x & 63, lg = lane >> 4, le = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int n_hex_tiles = h.z, n_tiles = h2.x;                 // (block-uniform) hex tiles first, then the other types' tiles
    const size_t tile0 = (size_t)h2.y;
    // this wave's first hex tile and the matrix of the 8-node type: requested with the chunk's node tables
    uint2 rec_next = make_uint2(0u, 0u);
    double ck_next = 0.0;
    if (wave_u < n_hex_tiles) {
        rec_next = T.hrec[(tile0 + wave_u) * 64 + lane];
        ck_next = T.tck[(tile0 + wave_u) * 16 + le];
    }
    double A[12];                                                // fragments (k-step ks, M-tile mt) of types[0]: frag[(ks * 2 + mt) * 64 + lane]
    int dst[NPT], ts[NPT];
    {
        int g[NPT];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int idx = (int)threadIdx.x + j * NT;
            const bool in = MAXN % NT == 0 || idx < MAXN;          // (NW = 5: the last round of a thread may lie beyond the chunk's 768 entries)
            const size_t n = (size_t)b * MAXN + (in ? idx : 0);
            g[j] = in ? ntload(T.nodes + n) : -1;
            dst[j] = in ? ntload(T.dst + n) : INT_MIN;
            ts[j] = in ? (int)__builtin_nontemporal_load(T.tslot + n) : 0;
        }
        double xg[NPT][3];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {                          // (x of a padding entry: node 0, never stored)
            const double *xp = x + 3 * (size_t)(g[j] >= 0 ? g[j] : 0);
            xg[j][0] = xp[0]; xg[j][1] = xp[1]; xg[j][2] = xp[2];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) A[i] = n_hex_tiles > 0 ? T.frag[(size_t)i * 64 + lane] : 0.0;   // (requested BEHIND the gather: they return in order)
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (g[j] >= 0) {
                const int sl3 = 3 * (ts[j] & 0x3ff);
                xs[sl3] = xg[j][0]; xs[sl3 + 1] = xg[j][1]; xs[sl3 + 2] = xg[j][2];
                ys[sl3] = 0.0; ys[sl3 + 1] = 0.0; ys[sl3 + 2] = 0.0;
            }
    }
    __syncthreads();
    stamp(1);
    auto wait_done = [&](int count) {                            // until `count` tiles of the chunk have completed
        while (__hip_atomic_load(&turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < count) __builtin_amdgcn_s_sleep(1);
    };
    auto count_in = [&]() {                                      // (the LDS unit serves one wave's operations in issue order: after its adds)
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&turn, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // ---- hex tiles: wave w takes the tiles w, w + 4, ... -----------------------------------------------------------------------
    // (Deferring a tile's adds behind the next tile's matrix instructions was measured and lost, session r: a wave issues in order, its
    //  twelve matrix instructions hold it for their 768 cycles of the pipe, and with four waves per SIMD in this phase the pipe is busy.)
    for (int t = wave_u; t < n_hex_tiles; t += NW) { // wave-uniform
        unsigned long long tt0 = 0, tt1 = 0, tt2 = 0;
        if constexpr (STAMP) tt0 = __builtin_amdgcn_s_memtime();
        const uint2 rec = rec_next;
        const double c = ck_next;
        const int wait_for = T.twait[tile0 + t];                 // (scalar load)
        if (t + NW < n_hex_tiles) {
            rec_next = T.hrec[(tile0 + t + NW) * 64 + lane];
            ck_next = T.tck[(tile0 + t + NW) * 16 + le];
        }
        const int l0 = 3 * (int)(rec.x & 0xffffu), l1 = 3 * (int)(rec.x >> 16);
        const unsigned sw = rec.y;
        const int p0 = (sw >> 8) & 3, p1 = (sw >> 10) & 3, p2 = (sw >> 12) & 3;                             // the element's own component order
        const double xv[6] = {xs[l0 + p0], xs[l0 + p1], xs[l0 + p2], xs[l1 + p0], xs[l1 + p1], xs[l1 + p2]};   // :277 gather
        d4m_t acc0 = d4m_t{0.0, 0.0, 0.0, 0.0}, acc1 = d4m_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const double u = c * flip_sign(xv[ks], sw, ks);                                                 // :278-279 sign, Ck * U
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2 * ks], u, acc0, 0, 0, 0);                       // :279 Ke @ (.)
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2 * ks + 1], u, acc1, 0, 0, 0);
        }
        const double o[6] = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1]};                          // accumulator q = 3 j + c
        if constexpr (STAMP) tt1 = __builtin_amdgcn_s_memtime();
        wait_done(wait_for);
        if constexpr (STAMP) tt2 = __builtin_amdgcn_s_memtime();
        if (sw >> 31) {
#pragma unroll
            for (int q = 0; q < 6; ++q)                                                                      // :280, :300 (added by the LDS unit)
                __hip_atomic_fetch_add(&ys[(q < 3 ? l0 : l1) + (q % 3 == 0 ? p0 : q % 3 == 1 ? p1 : p2)], flip_sign(o[q], sw, q), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        count_in();
        if constexpr (STAMP) {
            tile_acc[0] += 1; tile_acc[1] += tt1 - tt0; tile_acc[2] += tt2 - tt1; tile_acc[3] += __builtin_amdgcn_s_memtime() - tt2;
        }
    }
    stamp(8);
    // ---- the other types' tiles: four at a time (one per wave), added in ascending order (k_ebe_mixed) --------------------------------
    int2 info_next = n_hex_tiles + wave_u < n_tiles ? T.tinfo[tile0 + n_hex_tiles + wave_u] : make_int2(0, 0);
    for (int t0 = n_hex_tiles; t0 < n_tiles; t0 += NW) { // block-uniform
        unsigned long long tt0 = 0, tt1 = 0, tt2 = 0;
        if constexpr (STAMP) tt0 = __builtin_amdgcn_s_memtime();
        const int ti = t0 + wave_u;
        const bool have = ti < n_tiles;                          // wave-uniform
        const int2 info = info_next;                             // this tile's header was requested a round ago:
        if (ti + NW < n_tiles) info_next = T.tinfo[tile0 + ti + NW];   // header -> fragments is the dependent chain
        d4m_t acc[MTM];
        int tl3[JM];
        unsigned tsw = 0u;
        int pw = 0 | 1 << 2 | 2 << 4;
        int nn = 0, ncol = 0, mycol = 255;
#pragma unroll
        for (int mt = 0; mt < MTM; ++mt) acc[mt] = d4m_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < JM; ++j) tl3[j] = 0;
        if (have) {
            const size_t tg = tile0 + ti;
            nn = __builtin_amdgcn_readfirstlane(info.x & 255);
            const int J = __builtin_amdgcn_readfirstlane((info.x >> 8) & 255);
            ncol = __builtin_amdgcn_readfirstlane(info.x >> 16);
            const double *F = T.frag + (size_t)__builtin_amdgcn_readfirstlane(info.y) * 64 + lane;
            const double tc = T.tck[tg * 16 + le];
            mycol = (int)T.tcol[tg * 16 + le];
            pw = (int)T.tperm[tg * 16 + le];
            tsw = T.tsgw[(tg * 4 + lg) * 16 + le];
#pragma unroll
            for (int j = 0; j < JM; ++j)
                if (j < J) tl3[j] = 3 * (int)T.tlid[(tg * T.np + 4 * j + lg) * 16 + le];
            switch (J) {                                          // wave-uniform: straight-line code per size
            case 1: mixed_tile_contract<1, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 2: if constexpr (JM >= 2) mixed_tile_contract<2, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 3: if constexpr (JM >= 3) mixed_tile_contract<3, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 4: if constexpr (JM >= 4) mixed_tile_contract<4, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 5: if constexpr (JM >= 5) mixed_tile_contract<5, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 6: if constexpr (JM >= 6) mixed_tile_contract<6, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 7: if constexpr (JM >= 7) mixed_tile_contract<7, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            case 8: if constexpr (JM >= 8) mixed_tile_contract<8, MTM>(F, xs, tl3, tc, tsw, pw, lg, nn, acc); break;
            default: break;
            }
            if constexpr (STAMP) tt1 = __builtin_amdgcn_s_memtime();
            wait_done(ti);                                       // every tile before this one has added
            if constexpr (STAMP) tt2 = __builtin_amdgcn_s_memtime();
            for (int s = 0; s < ncol; ++s)
                if (mycol == s) {
#pragma unroll
                    for (int j = 0; j < JM; ++j)
                        if (lg < nn - 4 * j) {
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) {
                                const int q = 3 * j + cc;
                                const double o = flip_sign(acc[q / 4][q % 4], tsw, q);                        // :280
                                __hip_atomic_fetch_add(&ys[tl3[j] + ((pw >> 2 * cc) & 3)], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // :300
                            }
                        }
                }
            count_in();
            if constexpr (STAMP) {
                tile_acc[4] += 1; tile_acc[5] += tt1 - tt0; tile_acc[6] += tt2 - tt1; tile_acc[7] += __builtin_amdgcn_s_memtime() - tt2;
            }
        }
    }
    stamp(9);
    __syncthreads();                                             // every add is in before the tile is written out
    stamp(10);
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; ++j)
        if (dst[j] != INT_MIN) {
            const int sl3 = 3 * (ts[j] & 0x3ff), wmask = ts[j] >> 12;
            double *out = dst[j] >= 0 ? y + dst[j] : buf + 3 * (size_t)(-dst[j] - 1);
            const double y0 = ys[sl3], y1 = ys[sl3 + 1], y2 = ys[sl3 + 2];
            out[0] = y0; out[1] = y1; out[2] = y2;
            if (DOT && dst[j] >= 0 && dst[j] >= dot_lo) {        // fused p.Ap.w (:487) on the dofs this chunk finalises
                if (wmask & 1) dot += xs[sl3] * y0;
                if (wmask & 2) dot += xs[sl3 + 1] * y1;
                if (wmask & 4) dot += xs[sl3 + 2] * y2;
            }
        }
    if constexpr (DOT) {
        __shared__ double lds[NW];
        const double sw = wave_sum(dot);
        if ((threadIdx.x & 63) == 0) lds[wave] = sw;
        __syncthreads();
        if (threadIdx.x == 0) {                                      // (block_sum's order for NW = 4: wave after wave)
            double tot = lds[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) tot += lds[w];
            partials[blockIdx.x] = tot;
        }
    }
    stamp(11);
    if constexpr (STAMP) {
        if ((threadIdx.x & 63) == 0) {
            for (int k = 0; k < 4; ++k) T.stamps[((size_t)blockIdx.x * NW + wave) * 16 + 12 + k] = tile_acc[4 + k];
            for (int k = 0; k < 4; ++k) T.stamps[((size_t)blockIdx.x * NW + wave) * 16 + 2 + k] = tile_acc[k];
        }
    }
}

template <bool DOT>
__global__ __launch_bounds__(kBlock) void k_ebe_shared(const int *__restrict__ sh_node, const int *__restrict__ sh_ptr,
                                                       int slot0, const double *__restrict__ buf,
                                                       double *__restrict__ y, int count, const double *__restrict__ x,
                                                       const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                       long long dot_lo)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int k = t / 3, d = t - 3 * k;
    double dot = 0.0;
    if (k < count) {
        const int q0 = sh_ptr[k], cnt = sh_ptr[k + 1] - q0;
        const double *b = buf + 3 * (size_t)(slot0 + q0) + d;
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = q < cnt ? ntload(b + 3 * q) : 0.0;
        double s = v[0];
#pragma unroll
        for (int q = 1; q < 8; ++q)
            if (q < cnt) s += v[q];
        for (int q = 8; q < cnt; ++q) s += ntload(b + 3 * q);      // more than 8 chunks at one node: irregular meshes only
        const size_t dof = 3 * (size_t)sh_node[k] + d;
        y[dof] = s;
        if (DOT && (long long)dof >= dot_lo && (flags[dof] & 3) == 3) dot += x[dof] * s;
    }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}

// any nd (hanging-node patterns): same algorithm, x re-gathered per block of 4 output rows
__global__ __launch_bounds__(kBlock) void k_ebe_generic(const int *__restrict__ dof, const uint8_t *__restrict__ sgn,
                                                        const double *__restrict__ ck, const double *__restrict__ ke,
                                                        const double *__restrict__ x, double *__restrict__ y, int nd,
                                                        int64_t ne, int64_t e_lo, int64_t e_hi)
{
    const int64_t e = e_lo + blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (e >= e_hi) return;
    const double c = ck[e];
    for (int a0 = 0; a0 < nd; a0 += 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nd; ++b) {
            double v = x[dof[(size_t)b * ne + e]];
            if (sgn[(size_t)b * ne + e]) v = -v;
            v = c * v;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (a0 + i < nd) acc[i] = fma(ke[(size_t)(a0 + i) * nd + b], v, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (a0 + i < nd) {
                double o = acc[i];
                if (sgn[(size_t)(a0 + i) * ne + e]) o = -o;
                y[dof[(size_t)(a0 + i) * ne + e]] += o;
            }
    }
}

}  // namespace pcg
