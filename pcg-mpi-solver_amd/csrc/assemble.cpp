// Host-side operator assembly: reference element tables -> 3x3-block CSR over nodes.
//
// The reference never forms a matrix: calcMatVecProd (src/solver/pcg_solver.py:265-300) gathers
// x[dofs], flips signs, multiplies by Ke[type] scaled by Ck_e, flips signs and scatter-adds.
// That operator is  A = sum_e P_e^T S_e (Ck_e Ke_type(e)) S_e P_e ; this file builds it once per
// part so the GPU hot path is a (block) sparse mat-vec.  Row-gather formulation: every block row
// is produced by one thread from the elements incident to that node, in ascending
// (group, element, local row slot, local col slot) order -> bit-reproducible, no atomics.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pcg_internal.hpp"

namespace pcg {

struct Assembler {
    int64_t n_nodes = 0;
    int n_threads = 1;
    struct Group {
        int nd;
        int64_t ne;
        std::vector<int32_t> node;   // (ne, nd) element-major: NEW node index of slot a
        std::vector<uint8_t> dir;    // (ne, nd) direction (dof % 3) of slot a - not assumed constant per slot
        std::vector<uint8_t> sgn;    // (ne, nd) sign mask
        const double *ck;
        std::vector<double> ke;      // (nd, nd)
    };
    std::vector<Group> groups;
    // node -> incident (group, element) pairs, ascending
    std::vector<int64_t> adj_ptr;
    std::vector<int32_t> adj_g;
    std::vector<int64_t> adj_e;
    std::vector<int64_t> rowptr;
    std::vector<int32_t> cols;       // pattern (sorted per row)
};

}  // namespace pcg

struct pcg_asm { pcg::Assembler a; };

namespace pcg {

template <class F>
static void parallel_for(int64_t n, int n_threads, F f, int64_t serial_below = 1024)
{
    if (n_threads <= 1 || n < serial_below) { f(0, n); return; }
    std::vector<std::thread> th;
    int64_t chunk = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=] { f(lo, hi); });
    }
    for (auto &t : th) t.join();
}

static int build(Assembler &A, int32_t n_groups, const pcg_elem_group *gs, const int64_t *perm)
{
    const int64_t nn = A.n_nodes;
    A.groups.resize(n_groups);
    for (int g = 0; g < n_groups; ++g) {
        auto &G = A.groups[g];
        const auto &in = gs[g];
        if (in.nd <= 0 || in.nd > 65535) return set_error("pcg_asm: bad nd");
        G.nd = in.nd; G.ne = in.ne; G.ck = in.ck;
        G.node.resize((size_t)in.ne * in.nd);
        G.dir.resize((size_t)in.ne * in.nd);
        G.sgn.resize((size_t)in.ne * in.nd);
        G.ke.assign(in.ke, in.ke + (size_t)in.nd * in.nd);
        int bad = 0;
        // transpose (nd, ne) element-minor -> (ne, nd) element-major, blocked for cache
        parallel_for(in.ne, A.n_threads, [&](int64_t lo, int64_t hi) {
            const int64_t B = 256;
            for (int64_t e0 = lo; e0 < hi; e0 += B) {
                int64_t e1 = std::min(hi, e0 + B);
                for (int a = 0; a < in.nd; ++a) {
                    const int64_t *src = in.dof + (int64_t)a * in.ne;
                    const uint8_t *ss = in.sign + (int64_t)a * in.ne;
                    for (int64_t e = e0; e < e1; ++e) {
                        int64_t d = src[e];
                        int64_t node = d / 3;
                        if (d < 0 || node >= nn) { bad = 1; node = 0; d = 0; }
                        if (perm) node = perm[node];
                        G.node[(size_t)e * in.nd + a] = (int32_t)node;
                        G.dir[(size_t)e * in.nd + a] = (uint8_t)(d % 3);
                        G.sgn[(size_t)e * in.nd + a] = ss[e] ? 1 : 0;
                    }
                }
            }
        });
        if (bad) return set_error("pcg_asm: dof index out of range");
    }
    // ---- node -> (group, element) adjacency, ascending (g, e): counting sort -------------------
    A.adj_ptr.assign(nn + 1, 0);
    auto for_each_incidence = [&](auto &&fn) {
        for (int g = 0; g < n_groups; ++g) {
            auto &G = A.groups[g];
            std::vector<int32_t> tmp(G.nd);
            for (int64_t e = 0; e < G.ne; ++e) {
                const int32_t *nd = &G.node[(size_t)e * G.nd];
                std::copy(nd, nd + G.nd, tmp.begin());
                std::sort(tmp.begin(), tmp.end());
                int m = (int)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
                for (int k = 0; k < m; ++k) fn(tmp[k], g, e);
            }
        }
    };
    for_each_incidence([&](int32_t node, int, int64_t) { A.adj_ptr[node + 1]++; });
    for (int64_t i = 0; i < nn; ++i) A.adj_ptr[i + 1] += A.adj_ptr[i];
    A.adj_g.resize(A.adj_ptr[nn]);
    A.adj_e.resize(A.adj_ptr[nn]);
    {
        std::vector<int64_t> cur(A.adj_ptr.begin(), A.adj_ptr.end() - 1);
        for_each_incidence([&](int32_t node, int g, int64_t e) {
            int64_t k = cur[node]++;
            A.adj_g[k] = g; A.adj_e[k] = e;
        });
    }
    // ---- pattern: sorted unique neighbour nodes per row ---------------------------------------
    A.rowptr.assign(nn + 1, 0);
    std::vector<std::vector<int32_t>> chunk_cols;
    int nt = std::max(1, A.n_threads);
    int64_t chunk = (nn + nt - 1) / nt;
    chunk_cols.resize(nt);
    auto pattern_of_chunks = [&](int64_t tlo, int64_t thi) {
        for (int64_t t = tlo; t < thi; ++t) {
            int64_t lo = t * chunk, hi = std::min(nn, lo + chunk);
            std::vector<int32_t> cand;
            auto &out = chunk_cols[t];
            for (int64_t i = lo; i < hi; ++i) {
                cand.clear();
                for (int64_t k = A.adj_ptr[i]; k < A.adj_ptr[i + 1]; ++k) {
                    auto &G = A.groups[A.adj_g[k]];
                    const int32_t *nd = &G.node[(size_t)A.adj_e[k] * G.nd];
                    cand.insert(cand.end(), nd, nd + G.nd);
                }
                std::sort(cand.begin(), cand.end());
                cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
                A.rowptr[i + 1] = (int64_t)cand.size();
                out.insert(out.end(), cand.begin(), cand.end());
            }
        }
    };
    parallel_for(nt, nn < 4096 ? 1 : nt, pattern_of_chunks, 0);      // one thread per node chunk (nt of them, however few)
    for (int64_t i = 0; i < nn; ++i) A.rowptr[i + 1] += A.rowptr[i];
    A.cols.resize(A.rowptr[nn]);
    {
        int64_t off = 0;
        for (int t = 0; t < nt; ++t) {
            if (!chunk_cols[t].empty())
                std::memcpy(&A.cols[off], chunk_cols[t].data(), chunk_cols[t].size() * sizeof(int32_t));
            off += (int64_t)chunk_cols[t].size();
            std::vector<int32_t>().swap(chunk_cols[t]);
        }
    }
    return 0;
}

// the blocks of row i (its columns are A.cols[rowptr[i] .. rowptr[i+1])) -> rv (9 doubles per block, row-major 3x3);
// contributions summed in ascending (group, element, local row slot, local col slot) order
static void fill_row(const Assembler &A, int64_t i, double *rv, std::vector<int> &pos)
{
    const int64_t r0 = A.rowptr[i], r1 = A.rowptr[i + 1];
    const int32_t *rc = &A.cols[r0];
    std::memset(rv, 0, sizeof(double) * 9 * (size_t)(r1 - r0));
    for (int64_t k = A.adj_ptr[i]; k < A.adj_ptr[i + 1]; ++k) {
        const auto &G = A.groups[A.adj_g[k]];
        const int64_t e = A.adj_e[k];
        const int nd = G.nd;
        const int32_t *en = &G.node[(size_t)e * nd];
        const uint8_t *ed = &G.dir[(size_t)e * nd];
        const uint8_t *es = &G.sgn[(size_t)e * nd];
        const double ck = G.ck[e];
        pos.resize(nd);
        for (int b = 0; b < nd; ++b)
            pos[b] = (int)(std::lower_bound(rc, rc + (r1 - r0), en[b]) - rc);
        for (int a = 0; a < nd; ++a) {
            if (en[a] != (int32_t)i) continue;
            const double *krow = &G.ke[(size_t)a * nd];
            const int da = ed[a];
            for (int b = 0; b < nd; ++b) {
                double v = ck * krow[b];                     // Ck_e * Ke[a,b]
                if (es[a] != es[b]) v = -v;                  // S_e ... S_e  (exact)
                rv[(size_t)pos[b] * 9 + da * 3 + ed[b]] += v;
            }
        }
    }
}

static void fill_values(const Assembler &A, double *vals)
{
    parallel_for(A.n_nodes, A.n_threads, [&](int64_t lo, int64_t hi) {
        std::vector<int> pos;          // column position of each slot of the current element
        for (int64_t i = lo; i < hi; ++i) fill_row(A, i, vals + A.rowptr[i] * 9, pos);
    });
}

static void fill(const Assembler &A, int32_t *cols, double *vals)
{
    std::memcpy(cols, A.cols.data(), A.cols.size() * sizeof(int32_t));
    fill_values(A, vals);
}

void asm_views(const pcg_asm *a, int64_t *n_nodes, const int64_t **rowptr, const int32_t **cols)
{
    *n_nodes = a->a.n_nodes;
    *rowptr = a->a.rowptr.data();
    *cols = a->a.cols.data();
}

void asm_fill_values(const pcg_asm *a, double *vals) { fill_values(a->a, vals); }

bool asm_to_sell(const pcg_asm *h, int64_t n_boundary_nodes, int32_t rows_per_lane, bool want_dict, int64_t max_unique, SellHost &out)
{
    const Assembler &A = h->a;
    if (want_dict) rows_per_lane = 1;
    if (rows_per_lane != 1 && rows_per_lane != 2) throw std::runtime_error("rows_per_lane must be 1 or 2");
    const int C = 64 * rows_per_lane;
    const int64_t nn = A.n_nodes;
    max_unique = std::min<int64_t>(max_unique, 65535);
    out = SellHost();
    out.n_nodes = nn;
    out.C = C;
    out.n_slices = (nn + C - 1) / C;
    out.nnzb = A.rowptr[nn];
    out.n_bnd_slices = std::min<int64_t>(out.n_slices, (n_boundary_nodes + C - 1) / C);
    out.slice_ptr.assign(out.n_slices + 1, 0);
    for (int64_t s = 0; s < out.n_slices; ++s) {
        int64_t w = 0;
        for (int64_t r = s * C; r < std::min<int64_t>(nn, (s + 1) * C); ++r) w = std::max<int64_t>(w, A.rowptr[r + 1] - A.rowptr[r]);
        out.slice_ptr[s + 1] = out.slice_ptr[s] + w;
    }
    const int64_t tot = out.slice_ptr[out.n_slices];
    out.cols.assign((size_t)tot * C, 0);
    out.diag.assign((size_t)nn * 3, 0.0);
    std::vector<uint16_t> bidx(want_dict ? (size_t)tot * C : 0);
    if (!want_dict) out.vals.assign((size_t)tot * C * 9, 0.0);
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(A.n_threads, out.n_slices / 64 + 1));
    std::vector<BlockTable> local((size_t)nt, BlockTable(max_unique));
    std::vector<std::pair<size_t, size_t>> slots((size_t)nt);
    std::vector<char> failed((size_t)nt, 0);
    const int64_t chunk = (out.n_slices + nt - 1) / nt;
    auto work = [&](int t) {
        BlockTable &tab = local[t];
        std::vector<int> pos;
        std::vector<double> rv;
        BlockKey key, zero;
        std::memset(zero.w, 0, sizeof(zero.w));
        const int64_t s_lo = std::min(out.n_slices, t * chunk), s_hi = std::min(out.n_slices, s_lo + chunk);
        slots[t] = {(size_t)out.slice_ptr[s_lo] * C, (size_t)out.slice_ptr[s_hi] * C};
        for (int64_t s = s_lo; s < s_hi; ++s) {
            const int64_t base = out.slice_ptr[s], w = out.slice_ptr[s + 1] - base;
            for (int l = 0; l < C; ++l) {
                const int64_t r = s * C + l;
                const bool live = r < nn;
                const int64_t r0 = live ? A.rowptr[r] : 0, len = live ? A.rowptr[r + 1] - r0 : 0;
                if (live) {
                    rv.resize((size_t)std::max<int64_t>(1, len) * 9);
                    fill_row(A, r, rv.data(), pos);
                }
                for (int64_t k = 0; k < w; ++k) {
                    const size_t ci = (size_t)(base + k) * C + l;
                    const BlockKey *kp = &zero;                    // padding: value 0, a valid column NEAR the slice (as bsr_to_sell)
                    if (k < len) {
                        out.cols[ci] = A.cols[r0 + k];
                        if (want_dict) { std::memcpy(key.w, &rv[(size_t)k * 9], sizeof(key.w)); kp = &key; }
                        else
                            for (int c = 0; c < 9; ++c) out.vals[((size_t)(base + k) * 9 + c) * C + l] = rv[(size_t)k * 9 + c];
                        if (A.cols[r0 + k] == r)
                            for (int a = 0; a < 3; ++a) out.diag[(size_t)r * 3 + a] = rv[(size_t)k * 9 + a * 3 + a];
                    } else {
                        out.cols[ci] = (int32_t)(live ? r : nn - 1);
                    }
                    if (!want_dict) continue;
                    const int32_t id = tab.add(*kp);
                    if (id < 0) { failed[t] = 1; return; }
                    bidx[ci] = (uint16_t)id;
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (char f : failed)
        if (f) return false;
    return want_dict ? finish_dictionary(out, local, slots, bidx, max_unique) : true;
}

}  // namespace pcg

using namespace pcg;

extern "C" {

int pcg_asm_create(int64_t n_nodes, int32_t n_groups, const pcg_elem_group *groups, const int64_t *node_perm,
                   int32_t n_threads, pcg_asm **out)
{
    if (!out || n_nodes <= 0 || n_groups < 0 || n_nodes > INT32_MAX) return set_error("pcg_asm_create: bad argument");
    try {
        auto *h = new pcg_asm();
        h->a.n_nodes = n_nodes;
        int hw = (int)std::thread::hardware_concurrency();
        h->a.n_threads = n_threads > 0 ? n_threads : std::max(1, std::min(hw, 32));
        int rc = build(h->a, n_groups, groups, node_perm);
        if (rc) { delete h; return rc; }
        *out = h;
        return 0;
    } catch (const std::exception &ex) {
        return set_error(std::string("pcg_asm_create: ") + ex.what());
    }
}

int64_t pcg_asm_nnzb(const pcg_asm *a) { return a ? a->a.rowptr.back() : -1; }

int pcg_asm_rowptr(const pcg_asm *a, int64_t *rowptr)
{
    if (!a || !rowptr) return set_error("pcg_asm_rowptr: null");
    std::memcpy(rowptr, a->a.rowptr.data(), a->a.rowptr.size() * sizeof(int64_t));
    return 0;
}

int pcg_asm_fill(const pcg_asm *a, int32_t *cols, double *vals)
{
    if (!a || !cols || !vals) return set_error("pcg_asm_fill: null");
    try { fill(a->a, cols, vals); } catch (const std::exception &ex) { return set_error(ex.what()); }
    return 0;
}

void pcg_asm_destroy(pcg_asm *a) { delete a; }

}  // extern "C"
