// SpMV kernels of the assembled operator (reference: calcMatVecProd, src/solver/pcg_solver.py:265-300, + the fused
// p.Ap.w of :487): k_spmv (SELL over 3x3 blocks), k_spmv_dict (value dictionary), k_spmv_scalar (literal CSR volume).
#pragma once
#include "hip_common.hpp"

namespace pcg {

// ------------------------------------------------------------------------------------------------
// SpMV over the SELL-C 3x3-block matrix.  One wave per slice at a time; RPL rows per lane
// (RPL=1: C=64, 8-B lane loads; RPL=2: C=128, 16-B lane loads).
// ------------------------------------------------------------------------------------------------
// COL16: the block columns of a slice are stored as 16-bit offsets from the slice's smallest column (colbase[s]) -
// 74 instead of 76 bytes per stored block; chosen at upload when every slice spans fewer than 65536 block columns
// (node numberings with a bandwidth below 32 k nodes, e.g. the 10 M-dof brick: 22 651).  Same columns, same order,
// same arithmetic: results are bit-identical to the 32-bit form.
// PACK (round 4; the interface rows' launch of a part with neighbours, no dot): the rows also write their entries of the send
// buffer - k_halo_pack folded into the epilogue: dof d goes to send[fpos[q]] for q in [fptr[d], fptr[d+1]) (the same positions its
// neighbours' contributions arrive at in the receive buffer, pcg_set_halo).  Rows that continue in an overflow part pack there.
struct PackArgs { const int *fptr, *fpos; double *send; };

// HOLD (round 6): a wave keeps the y of the slices it computes in LDS (kSpmvHold slices of 192 doubles per wave) and writes them at the
// END of the launch, as coalesced 16-B stores; the launcher cuts the slice range into launches a wave's share of which fits the slots.
// Why: 81 MB of y stores trickling into a 6.9 GB read stream can cost up to 13 % of the launch - how much depends on the box and on where
// the vectors landed (the spread of rounds 1 - 5: 1.03 ... 1.20 ms for the same launch).  tools/micro/spmv_ablation rebuilds the kernel
// piece by piece: values + columns + x gathers + multiply-adds run at 1.040 ms on EVERY box; the stores add 0.01 ms on a fast box and
// 0.15 - 0.20 ms on a slow one whatever their width or cache policy (coalesced, nt, sc1: still + 0.10 - 0.13); the same stores issued after
// the reads of a quarter of the slices - four launches, launch overheads included - give 1.053 ms (profiles/r06_spmv_ablation_*).  On a
// slow box reads and writes in separate phases win 5 - 11 %; on a fast one the three extra launches cost 2 - 4 %: the engine TIMES both forms
// on its own vectors at the first solve and keeps the faster (Backend::tune_operator, PCG_SPMV_HOLD=0 / 4 forces one).
//
// Both forms produce the SAME BITS - y and the dot partials - because both walk the slice range in the same `parts` sub-ranges with the
// same slice -> wave assignment inside each: part = q (0 <= q < parts): this launch is sub-range q alone (HOLD: its y from LDS at the end;
// its block partial is added to the one launch q - 1 left); part = -1: one launch walks all sub-ranges, block partial = ((v0 + v1) + ...).
constexpr int kSpmvHold = 4;

template <int RPL, bool DOT, bool COL16, bool PACK = false, bool HOLD = false>
__global__ __launch_bounds__(kBlock) void k_spmv(const int64_t *__restrict__ slice_ptr, const void *__restrict__ cols_any,
                                                 const int *__restrict__ colbase,
                                                 const double *__restrict__ vals, const double *__restrict__ x,
                                                 double *__restrict__ y, const uint8_t *__restrict__ flags,
                                                 double *__restrict__ partials, int64_t slice_lo_all, int64_t slice_hi_all,
                                                 int64_t n_nodes, int xcd_aware, const unsigned long long *__restrict__ ov_mask,
                                                 PackArgs pk, int parts, int part)
{
    constexpr int C = 64 * RPL;
    using DV = typename VecT<RPL>::d;
    using IV = typename VecT<RPL>::i;
    using CV = typename std::conditional<COL16, typename std::conditional<RPL == 1, unsigned short, ushort2>::type, IV>::type;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool xa = (xcd_aware & 1) != 0;                    // bit 1 of the argument: non-temporal y stores
    const int xcd = xa ? (blockIdx.x & 7) : 0;
    const int64_t lb = xa ? (blockIdx.x >> 3) : blockIdx.x;
    const int64_t blocks_per_xcd = xa ? ((gridDim.x + 7 - xcd) >> 3) : gridDim.x;   // blocks with b&7 == xcd
    const int64_t wstride = blocks_per_xcd * kWavesPerBlock;
    __shared__ double held[HOLD ? kWavesPerBlock * kSpmvHold * 192 : 1];
    __shared__ double lds[kWavesPerBlock];
    int n_held = 0;
    int64_t held_s[kSpmvHold];
    auto flush_held = [&]() {                                  // the held slices' 192 doubles each as 16-B pairs, lanes 0..63 then 0..31
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS writes (a wave reads only its own slots)
#pragma unroll
        for (int q = 0; q < kSpmvHold; ++q)
            if (q < n_held) {
                const double *st = held + ((size_t)wid * kSpmvHold + q) * 192;
                double *yb = y + 3 * held_s[q] * 64;
                const int64_t rows_left = n_nodes - held_s[q] * 64;              // (the matrix's last slice may be partial)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int e = 128 * t + 2 * lane;
                    if (e < 192) {
                        if ((e + 1) / 3 < rows_left) *reinterpret_cast<v2d_t *>(yb + e) = *reinterpret_cast<const v2d_t *>(st + e);
                        else if (e / 3 < rows_left) yb[e] = st[e];
                    }
                }
            }
        n_held = 0;
    };
    double total = 0.0;                                          // thread 0: this block's dot partial over the sub-ranges of this launch
    const int sub_lo = part >= 0 ? part : 0, sub_hi = part >= 0 ? part + 1 : parts;
    for (int sub = sub_lo; sub < sub_hi; ++sub) {
    // Slice -> wave mapping inside a sub-range.  Default (xcd_aware = 0): wave g of the grid takes slices g, g + G, ... so the
    // whole chip streams one region of the matrix.  xcd_aware = 1: block b runs on XCD b & 7 (observed;
    // speed only, never correctness) and each XCD owns one contiguous eighth of the slice range.
    const int64_t slice_lo = slice_lo_all + (slice_hi_all - slice_lo_all) * sub / parts;
    const int64_t slice_hi = slice_lo_all + (slice_hi_all - slice_lo_all) * (sub + 1) / parts;
    const int64_t S = slice_hi - slice_lo;
    const int64_t c_lo = xa ? slice_lo + (S * xcd) / 8 : slice_lo;
    const int64_t c_hi = xa ? slice_lo + (S * (xcd + 1)) / 8 : slice_hi;
    double dot = 0.0;
    for (int64_t s = c_lo + lb * kWavesPerBlock + wid; s < c_hi; s += wstride) {
        const int64_t base = slice_ptr[s];
        const int w = (int)(slice_ptr[s + 1] - base);
        const DV *vp = reinterpret_cast<const DV *>(vals + (size_t)base * 9 * C) + lane;
        const CV *cp = reinterpret_cast<const CV *>(cols_any) + (size_t)base * 64 + lane;
        int cb = 0;
        if constexpr (COL16) cb = colbase[s];
        double acc[RPL][3];
#pragma unroll
        for (int h = 0; h < RPL; ++h) acc[h][0] = acc[h][1] = acc[h][2] = 0.0;
#pragma unroll 3
        for (int k = 0; k < w; ++k) {
            IV jv = ntload(cp + (size_t)k * 64);
            if constexpr (COL16) {
                if constexpr (RPL == 1) jv += cb; else { jv.x += cb; jv.y += cb; }
            }
            DV v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = ntload(vp + ((size_t)k * 9 + c) * 64);
            if constexpr (RPL == 1) {
                const double *xp = x + 3 * (size_t)jv;
                const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);      // 24 contiguous bytes per lane: one 16-B + one 8-B load
                const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    acc[0][a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[0][a])));
            } else {
                const double *xa = x + 3 * (size_t)jv.x, *xb = x + 3 * (size_t)jv.y;
                const double a0 = xa[0], a1 = xa[1], a2 = xa[2];
                const double b0 = xb[0], b1 = xb[1], b2 = xb[2];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    acc[0][a] = fma(v[3 * a + 2].x, a2, fma(v[3 * a + 1].x, a1, fma(v[3 * a].x, a0, acc[0][a])));
                    acc[1][a] = fma(v[3 * a + 2].y, b2, fma(v[3 * a + 1].y, b1, fma(v[3 * a].y, b0, acc[1][a])));
                }
            }
        }
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
            const int64_t row = s * C + (int64_t)lane * RPL + h;
            if (row < n_nodes) {
                double *yp = y + 3 * row;
                if (HOLD && RPL == 1) {                                      // kept in LDS, written when the slots are full / after the loop
                    double *st = held + ((size_t)wid * kSpmvHold + n_held) * 192 + 3 * lane;
                    st[0] = acc[h][0]; st[1] = acc[h][1]; st[2] = acc[h][2];
                } else if (xcd_aware & 2) {
                    __builtin_nontemporal_store(acc[h][0], yp); __builtin_nontemporal_store(acc[h][1], yp + 1);
                    __builtin_nontemporal_store(acc[h][2], yp + 2);
                } else { yp[0] = acc[h][0]; yp[1] = acc[h][1]; yp[2] = acc[h][2]; }
                if constexpr (PACK) {
                    if (RPL != 1 || ov_mask == nullptr || ((ov_mask[s] >> lane) & 1ull) == 0) {
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            for (int q = pk.fptr[3 * row + a], q1 = pk.fptr[3 * row + a + 1]; q < q1; ++q) pk.send[pk.fpos[q]] = acc[h][a];
                    }
                }
                if constexpr (DOT) {
                    // (a row that continues in the overflow part is not final here: k_spmv_ovf adds its term)
                    const bool final_here = RPL != 1 || ov_mask == nullptr || ((ov_mask[s] >> lane) & 1ull) == 0;
                    const uint8_t *fp = flags + 3 * row;
                    const double *xp = x + 3 * row;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (final_here && (fp[a] & 3) == 3) dot += xp[a] * acc[h][a];
                }
            }
        }
        if constexpr (HOLD && RPL == 1) {
            held_s[n_held++] = s;
            if (n_held == kSpmvHold) flush_held();               // (a launch cut to the slots never gets here before its last slice)
        }
    }
    if constexpr (DOT) {
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) total = sub == sub_lo ? v[0] : total + v[0];
        if (sub + 1 < sub_hi) __syncthreads();                         // (the next sub-range's block_sum reuses lds)
    }
    }
    if constexpr (HOLD && RPL == 1) flush_held();                // every read of this launch has been issued: now the stores
    if constexpr (DOT) {
        if (threadIdx.x == 0) partials[blockIdx.x] = part > 0 ? partials[blockIdx.x] + total : total;
    }
}

// Overflow part of a split SELL matrix (SellHost::ov_*, sell.cpp split_overflow): the rows that are longer than their slice's
// base width, compacted, 64 per slice.  A lane RESUMES the sum of its row from the y that k_spmv stored and continues through
// the remaining block columns in their original order with the same three fused multiply-adds per component: y ends up with
// the bits of the unsplit matrix.  The fused p.Ap term of these rows is formed here (k_spmv leaves them out).
template <bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_ovf(const int64_t *__restrict__ slice_ptr, const int *__restrict__ rows,
                                                     const int *__restrict__ cols, const double *__restrict__ vals,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                     int64_t slice_lo, int64_t slice_hi)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wstride = (int64_t)gridDim.x * kWavesPerBlock;
    double dot = 0.0;
    for (int64_t s = slice_lo + (int64_t)blockIdx.x * kWavesPerBlock + wid; s < slice_hi; s += wstride) {
        const int64_t base = slice_ptr[s];
        const int w = (int)(slice_ptr[s + 1] - base);
        const int row = ntload(rows + s * 64 + lane);
        const double *vp = vals + (size_t)base * 9 * 64 + lane;
        const int *cp = cols + (size_t)base * 64 + lane;
        double acc[3] = {0.0, 0.0, 0.0};
        if (row >= 0) { acc[0] = y[3 * (size_t)row]; acc[1] = y[3 * (size_t)row + 1]; acc[2] = y[3 * (size_t)row + 2]; }
#pragma unroll 3
        for (int k = 0; k < w; ++k) {
            const int j = ntload(cp + (size_t)k * 64);
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = ntload(vp + ((size_t)k * 9 + c) * 64);
            const double *xp = x + 3 * (size_t)j;
            const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);
            const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[a])));
        }
        if (row >= 0) {
            double *yp = y + 3 * (size_t)row;
            yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2];
            if constexpr (DOT) {
                const uint8_t *fp = flags + 3 * (size_t)row;
                const double *xr = x + 3 * (size_t)row;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if ((fp[a] & 3) == 3) dot += xr[a] * acc[a];
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

// Windowed form of the split matrix (round 4, SellHost::win_*): ONE launch.  A workgroup takes a window - a dozen consecutive base
// slices and the overflow slices that hold the long rows of exactly those slices - runs the base slices (a wave per slice, the
// loop of k_spmv), meets at a block barrier (the y of the window's rows is then visible to the whole workgroup: same CU, same L1)
// and runs the window's overflow slices (the loop of k_spmv_ovf).  Why: as a second launch the overflow part gathered x with no
// locality at all - 64 rows from anywhere in a 512-row list, their tail columns, one 128-B line per lane for 24 B used
// (47.6 distinct lines per 53 live lanes measured on the 1 M-dof octree mesh) while the whole overflow part was in flight across all
// of x at once: it moved its bytes at 3.2 TB/s where the base part reaches 6.2.  Here the overflow rows of a window gather around
// the lines their own base part has just pulled into L1 / L2, and y does not leave L2 in between.  Same arithmetic in the same
// order per row: y keeps the bits of the unsplit matrix.  The windows are cut so that every workgroup gets the same number.
template <bool DOT, bool COL16, bool PACK = false>
__global__ __launch_bounds__(kBlock) void k_spmv_win(const int64_t *__restrict__ win_slice, const int64_t *__restrict__ win_ov,
                                                     const int64_t *__restrict__ slice_ptr, const void *__restrict__ cols_any,
                                                     const int *__restrict__ colbase, const double *__restrict__ vals,
                                                     const unsigned long long *__restrict__ ov_mask,
                                                     const int64_t *__restrict__ ov_slice_ptr, const int *__restrict__ ov_rows,
                                                     const int *__restrict__ ov_cols, const double *__restrict__ ov_vals,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                     int64_t win_lo, int64_t win_hi, int64_t n_nodes, PackArgs pk)
{
    using CV = typename std::conditional<COL16, unsigned short, int>::type;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double dot = 0.0;
    for (int64_t w = win_lo + blockIdx.x; w < win_hi; w += gridDim.x) {
        const int64_t s0 = win_slice[w], s1 = win_slice[w + 1];
        for (int64_t s = s0 + wid; s < s1; s += kWavesPerBlock) {          // ---- base slices
            const int64_t base = slice_ptr[s];
            const int wd = (int)(slice_ptr[s + 1] - base);
            const double *vp = vals + (size_t)base * 9 * 64 + lane;
            const CV *cp = reinterpret_cast<const CV *>(cols_any) + (size_t)base * 64 + lane;
            int cb = 0;
            if constexpr (COL16) cb = colbase[s];
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll 3
            for (int k = 0; k < wd; ++k) {
                int j = ntload(cp + (size_t)k * 64);
                if constexpr (COL16) j += cb;
                double v[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = ntload(vp + ((size_t)k * 9 + c) * 64);
                const double *xp = x + 3 * (size_t)j;
                const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);
                const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[a])));
            }
            const int64_t row = s * 64 + lane;
            if (row < n_nodes) {
                double *yp = y + 3 * row;
                yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2];
                if constexpr (PACK) {
                    if (((ov_mask[s] >> lane) & 1ull) == 0) {
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            for (int q = pk.fptr[3 * row + a], q1 = pk.fptr[3 * row + a + 1]; q < q1; ++q) pk.send[pk.fpos[q]] = acc[a];
                    }
                }
                if constexpr (DOT) {
                    if (((ov_mask[s] >> lane) & 1ull) == 0) {                // (a row that continues below forms its term there)
                        const uint8_t *fp = flags + 3 * row;
                        const double *xr = x + 3 * row;
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            if ((fp[a] & 3) == 3) dot += xr[a] * acc[a];
                    }
                }
            }
        }
        __syncthreads();
        const int64_t o0 = win_ov[w], o1 = win_ov[w + 1];
        for (int64_t s = o0 + wid; s < o1; s += kWavesPerBlock) {          // ---- the window's long rows continue
            const int64_t base = ov_slice_ptr[s];
            const int wd = (int)(ov_slice_ptr[s + 1] - base);
            const int row = ntload(ov_rows + s * 64 + lane);
            const double *vp = ov_vals + (size_t)base * 9 * 64 + lane;
            const int *cp = ov_cols + (size_t)base * 64 + lane;
            double acc[3] = {0.0, 0.0, 0.0};
            if (row >= 0) { acc[0] = y[3 * (size_t)row]; acc[1] = y[3 * (size_t)row + 1]; acc[2] = y[3 * (size_t)row + 2]; }
#pragma unroll 3
            for (int k = 0; k < wd; ++k) {
                const int j = ntload(cp + (size_t)k * 64);
                double v[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = ntload(vp + ((size_t)k * 9 + c) * 64);
                const double *xp = x + 3 * (size_t)j;
                const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);
                const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[a])));
            }
            if (row >= 0) {
                double *yp = y + 3 * (size_t)row;
                yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2];
                if constexpr (PACK) {
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        for (int q = pk.fptr[3 * (size_t)row + a], q1 = pk.fptr[3 * (size_t)row + a + 1]; q < q1; ++q) pk.send[pk.fpos[q]] = acc[a];
                }
                if constexpr (DOT) {
                    const uint8_t *fp = flags + 3 * (size_t)row;
                    const double *xr = x + 3 * (size_t)row;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if ((fp[a] & 3) == 3) dot += xr[a] * acc[a];
                }
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

// Dictionary variant (SellHost::bidx / dict, sell.cpp compress_blocks; PCG_FORMAT_DICTIONARY): a stored block is a column and a
// 16-bit index into the table of the matrix's DISTINCT 3x3 blocks, 4-6 bytes instead of 74-76.  The same lanes multiply the
// same values in the same order as k_spmv: results are bit-identical, only where the values come from differs.  LDSD: the
// table (72 B per entry) is copied into LDS once per workgroup - the workgroups are persistent, each wave walks many
// slices - and the lanes of a wave read their blocks from there (same index = one broadcast read; the kernel is bound by the
// LDS read rate and the x gathers, not by HBM: 0.5 GB instead of 6.9 GB per launch at 10 M dof).  !LDSD: tables beyond the LDS
// budget are read through L1/L2.
// MIXED (with LDSD): the table is larger than LDS; its n_lds most frequent entries (the host orders the table by descending
// frequency) are the LDS copy, a lane whose block is one of the others reads it through L1/L2 - a divergent branch that
// costs nothing when no lane of the wave needs it.
// COL16: a stored block is ONE 32-bit word - the 16-bit column offset in the low half, the table index in the high half - and
// a lane's words of four consecutive block columns are one 16-byte load (cols_any = uint4[(ptr4[s] + k / 4) * 64 + lane]; a
// slice's last group is padded, its padding never multiplied): 0.25 wave-level loads per block column instead of two.  Together
// with the 16 + 8-byte x gather that is 2.25 vector-memory instructions per block column instead of 5; the L1 takes 16 cycles
// for each whatever its width, which is what bounded this kernel (profiles/r03_pmc_spmv_dict_and_plain_before.md).
// !COL16 (a slice spans 65536 block columns or more): 32-bit columns and 16-bit indices in two arrays, one block column at a time.
template <bool DOT, bool COL16, bool LDSD, int BLK, bool MIXED = false>
__global__ __launch_bounds__(BLK) void k_spmv_dict(const int64_t *__restrict__ slice_ptr, const void *__restrict__ cols_any,
                                                      const int *__restrict__ colbase, const int64_t *__restrict__ ptr4,
                                                      const unsigned short *__restrict__ bidx,
                                                      const double *__restrict__ dict, int n_lds,
                                                      const double *__restrict__ x, double *__restrict__ y,
                                                      const uint8_t *__restrict__ flags, double *__restrict__ partials,
                                                      int64_t slice_lo, int64_t slice_hi, int64_t n_nodes)
{
    // LDS copy of the table: entries padded to 80 B (16-B aligned) so that a block is four ds_read_b128 + one ds_read_b64 -
    // 256 B/clk per CU; the 72-B layout compiles to ds_read2_b64 pairs, which run at half that rate (MI355X_MICROARCH.md, LDS)
    extern __shared__ __align__(16) double sdict[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int WPB = BLK / 64;                              // waves per workgroup: they share one copy of the table
    if constexpr (LDSD) {
        for (int i = threadIdx.x; i < 9 * n_lds; i += BLK) sdict[10 * (i / 9) + i % 9] = dict[i];
        __syncthreads();
    }
    const int64_t wstride = (int64_t)gridDim.x * WPB;
    double dot = 0.0;
    for (int64_t s = slice_lo + (int64_t)blockIdx.x * WPB + wid; s < slice_hi; s += wstride) {
        const int64_t base = slice_ptr[s];
        const int w = (int)(slice_ptr[s + 1] - base);
        double acc[3] = {0.0, 0.0, 0.0};
        auto block = [&](int j, int id) {                      // acc += table[id] . x[3 j .. 3 j + 2], the lanes of k_spmv in its order
            double v[9];
            if (LDSD && (!MIXED || id < n_lds)) {
                const double2 *e2 = reinterpret_cast<const double2 *>(sdict + 10 * id);
#pragma unroll
                for (int c = 0; c < 4; ++c) { const double2 t = e2[c]; v[2 * c] = t.x; v[2 * c + 1] = t.y; }
                v[8] = sdict[10 * id + 8];
            } else {
                const double *b = dict + 9 * (size_t)id;
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = b[c];
            }
            const double *xp = x + 3 * (size_t)j;
            const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);
            const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[a])));
        };
        if constexpr (COL16) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const int cb = colbase[s];
            const u4 *cp = reinterpret_cast<const u4 *>(cols_any) + (size_t)ptr4[s] * 64 + lane;
            const int nfull = w >> 2, rem = w & 3;
#pragma unroll 1
            for (int k4 = 0; k4 < nfull; ++k4) {
                const u4 c = __builtin_nontemporal_load(cp + (size_t)k4 * 64);
                block((int)(c.x & 0xffffu) + cb, (int)(c.x >> 16));
                block((int)(c.y & 0xffffu) + cb, (int)(c.y >> 16));
                block((int)(c.z & 0xffffu) + cb, (int)(c.z >> 16));
                block((int)(c.w & 0xffffu) + cb, (int)(c.w >> 16));
            }
            if (rem) {                                         // wave-uniform
                const u4 c = __builtin_nontemporal_load(cp + (size_t)nfull * 64);
                block((int)(c.x & 0xffffu) + cb, (int)(c.x >> 16));
                if (rem > 1) block((int)(c.y & 0xffffu) + cb, (int)(c.y >> 16));
                if (rem > 2) block((int)(c.z & 0xffffu) + cb, (int)(c.z >> 16));
            }
        } else {
            const int *cp = reinterpret_cast<const int *>(cols_any) + (size_t)base * 64 + lane;
            const unsigned short *ip = bidx + (size_t)base * 64 + lane;
#pragma unroll 3
            for (int k = 0; k < w; ++k) block(ntload(cp + (size_t)k * 64), ntload(ip + (size_t)k * 64));
        }
        const int64_t row = s * 64 + lane;
        if (row < n_nodes) {
            double *yp = y + 3 * row;
            yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2];
            if constexpr (DOT) {
                const uint8_t *fp = flags + 3 * row;
                const double *xr = x + 3 * row;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if ((fp[a] & 3) == 3) dot += xr[a] * acc[a];
            }
        }
    }
    if constexpr (DOT) {                                       // fixed order: lanes (shuffle tree), then the waves in turn
        __shared__ double lds[WPB];
        const double ws = wave_sum(dot);
        if (lane == 0) lds[wid] = ws;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = lds[0];
#pragma unroll
            for (int q = 1; q < WPB; ++q) t += lds[q];
            partials[blockIdx.x] = t;
        }
    }
}

// Scalar-row variant (SellHost::bs == 1): one lane per matrix ROW, one f64 value + one i32 column per stored
// entry - the literal CSR data volume (12 B per non-zero), in the same slice layout, so a wave's loads of a
// slice column are one 512 B + one 256 B coalesced line.  Used by pcg_create_csr(block = 1): systems whose
// rows are not 3-dof node blocks, and the "CSR-format" point of the measurement table.
template <bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_scalar(const int64_t *__restrict__ slice_ptr, const int *__restrict__ cols,
                                                        const double *__restrict__ vals, const double *__restrict__ x,
                                                        double *__restrict__ y, const uint8_t *__restrict__ flags,
                                                        double *__restrict__ partials, int64_t slice_lo, int64_t slice_hi,
                                                        int64_t n_rows)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wstride = (int64_t)gridDim.x * kWavesPerBlock;
    double dot = 0.0;
    for (int64_t s = slice_lo + (int64_t)blockIdx.x * kWavesPerBlock + wid; s < slice_hi; s += wstride) {
        const int64_t base = slice_ptr[s];
        const int w = (int)(slice_ptr[s + 1] - base);
        const double *vp = vals + (size_t)base * 64 + lane;
        const int *cp = cols + (size_t)base * 64 + lane;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        int k = 0;
        for (; k + 9 <= w; k += 9) {                            // 9 independent gathers in flight per lane
            int j[9];
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) { j[c] = ntload(cp + (size_t)(k + c) * 64); v[c] = ntload(vp + (size_t)(k + c) * 64); }
#pragma unroll
            for (int c = 0; c < 9; c += 3) {
                a0 = fma(v[c], x[j[c]], a0);
                a1 = fma(v[c + 1], x[j[c + 1]], a1);
                a2 = fma(v[c + 2], x[j[c + 2]], a2);
            }
        }
        for (; k < w; ++k) a0 = fma(ntload(vp + (size_t)k * 64), x[ntload(cp + (size_t)k * 64)], a0);
        const int64_t row = s * 64 + lane;
        if (row < n_rows) {
            const double r = (a0 + a1) + a2;
            y[row] = r;
            if constexpr (DOT)
                if ((flags[row] & 3) == 3) dot += x[row] * r;
        }
    }
    if constexpr (DOT) {
        __shared__ double lds[kWavesPerBlock];
        double v[1] = {dot};
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
    }
}

// Scalar-row copy of a plain 3x3-block SELL matrix, built on the device (pcg_create_scalar_copy): thread = scalar row r = 3 node + a,
// block column k of the node's row becomes the scalar entries 3 k .. 3 k + 2 (columns 3 j + b, values v[3 a + b]); the rest of the
// scalar slice's width is padding (value 0, the row's own column).  Writes are lane-contiguous in the scalar layout.
template <bool COL16>
__global__ __launch_bounds__(kBlock) void k_expand_scalar(const int64_t *__restrict__ bptr, const void *__restrict__ cols_any,
                                                          const int *__restrict__ colbase, const double *__restrict__ bvals,
                                                          const int64_t *__restrict__ ptr1, int *__restrict__ cols1,
                                                          double *__restrict__ vals1, int64_t n_rows)
{
    using CV = typename std::conditional<COL16, unsigned short, int>::type;
    const int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t node = r / 3, sb = node >> 6, s1 = r >> 6;
    const int a = (int)(r - 3 * node), lb = (int)(node & 63), l1 = (int)(r & 63);
    const int64_t bbase = bptr[sb], base1 = ptr1[s1];
    const int bw = (int)(bptr[sb + 1] - bbase), w1 = (int)(ptr1[s1 + 1] - base1);
    const CV *cp = reinterpret_cast<const CV *>(cols_any) + (size_t)bbase * 64 + lb;
    int cb = 0;
    if constexpr (COL16) cb = colbase[sb];
    for (int k = 0; k < bw; ++k) {
        const int j = (int)cp[(size_t)k * 64] + cb;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const size_t q = (size_t)(base1 + 3 * k + b) * 64 + l1;
            vals1[q] = bvals[((size_t)(bbase + k) * 9 + 3 * a + b) * 64 + lb];
            cols1[q] = 3 * j + b;
        }
    }
    for (int k = 3 * bw; k < w1; ++k) {
        const size_t q = (size_t)(base1 + k) * 64 + l1;
        vals1[q] = 0.0;
        cols1[q] = (int)r;
    }
}

}  // namespace pcg
