// 3x3-block CSR -> SELL-C layout used by the SpMV kernel (see SellHost in pcg_internal.hpp),
// plus the small shared utilities (error string, interface fix-up lists).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>

#include "pcg_internal.hpp"

namespace pcg {

static thread_local std::string g_err;
int set_error(const std::string &msg) { g_err = msg; return -1; }
const std::string &last_error_string() { return g_err; }

void bsr_to_sell(int64_t n_nodes, const int64_t *rowptr, const int32_t *cols, const double *vals,
                 int64_t n_boundary_nodes, int32_t rows_per_lane, int n_threads, SellHost &out)
{
    if (rows_per_lane != 1 && rows_per_lane != 2) throw std::runtime_error("rows_per_lane must be 1 or 2");
    const int C = 64 * rows_per_lane;
    out.n_nodes = n_nodes;
    out.C = C;
    out.n_slices = (n_nodes + C - 1) / C;
    out.nnzb = rowptr[n_nodes];
    out.n_bnd_slices = std::min<int64_t>(out.n_slices, (n_boundary_nodes + C - 1) / C);
    out.slice_ptr.assign(out.n_slices + 1, 0);
    for (int64_t s = 0; s < out.n_slices; ++s) {
        int64_t w = 0;
        for (int64_t r = s * C; r < std::min<int64_t>(n_nodes, (s + 1) * C); ++r)
            w = std::max<int64_t>(w, rowptr[r + 1] - rowptr[r]);
        out.slice_ptr[s + 1] = out.slice_ptr[s] + w;
    }
    const int64_t tot = out.slice_ptr[out.n_slices];
    out.cols.assign((size_t)tot * C, 0);
    out.vals.assign((size_t)tot * C * 9, 0.0);
    out.diag.assign((size_t)n_nodes * 3, 0.0);
    auto work = [&](int64_t s_lo, int64_t s_hi) {
        for (int64_t s = s_lo; s < s_hi; ++s) {
            const int64_t base = out.slice_ptr[s], w = out.slice_ptr[s + 1] - base;
            for (int l = 0; l < C; ++l) {
                const int64_t r = s * C + l;
                const bool live = r < n_nodes;
                const int64_t r0 = live ? rowptr[r] : 0, len = live ? rowptr[r + 1] - r0 : 0;
                for (int64_t k = 0; k < w; ++k) {
                    const size_t ci = (size_t)(base + k) * C + l;
                    if (k < len) {
                        out.cols[ci] = cols[r0 + k];
                        const double *b = vals + (size_t)(r0 + k) * 9;
                        for (int c = 0; c < 9; ++c) out.vals[((size_t)(base + k) * 9 + c) * C + l] = b[c];
                        if (cols[r0 + k] == r)
                            for (int a = 0; a < 3; ++a) out.diag[(size_t)r * 3 + a] = b[a * 3 + a];
                    } else {
                        out.cols[ci] = (int32_t)(live ? r : n_nodes - 1);      // padding: value 0, a valid column NEAR the slice (16-bit column offsets)
                    }
                }
            }
        }
    };
    int nt = std::max(1, n_threads);
    if (nt == 1 || out.n_slices < 64) {
        work(0, out.n_slices);
    } else {
        std::vector<std::thread> th;
        int64_t chunk = (out.n_slices + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            int64_t lo = t * chunk, hi = std::min(out.n_slices, lo + chunk);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto &t : th) t.join();
    }
}

bool split_overflow(SellHost &m, double min_saving, int n_threads, int target_blocks)
{
    if (m.bs != 3 || m.C != 64 || m.ov_slices != 0 || m.n_slices == 0) return false;
    const int C = 64;
    const int64_t S = m.n_slices, tot = m.slice_ptr[S];
    if (m.vals.size() != (size_t)tot * C * 9) return false;                  // plain format only
    // effective row lengths: trailing all-zero blocks are padding (or contribute nothing)
    std::vector<int32_t> len((size_t)S * C, 0);
    auto scan = [&](int64_t s_lo, int64_t s_hi) {
        for (int64_t s = s_lo; s < s_hi; ++s) {
            const int64_t base = m.slice_ptr[s], w = m.slice_ptr[s + 1] - base;
            int32_t *ln = &len[(size_t)s * C];
            for (int64_t k = 0; k < w; ++k)
                for (int c = 0; c < 9; ++c) {
                    const double *v = &m.vals[((size_t)(base + k) * 9 + c) * C];
                    for (int l = 0; l < C; ++l)
                        if (v[l] != 0.0) ln[l] = (int32_t)(k + 1);
                }
        }
    };
    {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, S / 64 + 1));
        std::vector<std::thread> th;
        const int64_t chunk = (S + nt - 1) / nt;
        for (int t = 1; t < nt; ++t)
            if (t * chunk < S) th.emplace_back(scan, t * chunk, std::min(S, (t + 1) * chunk));
        scan(0, std::min(S, chunk));
        for (auto &t : th) t.join();
    }
    // base width per slice: min over w of 64 w + f sum(max(0, len - w)), w among the row lengths of the slice
    double f = 1.5;                                                          // development knobs (tools/iter_ab.py)
    size_t window = 512;
    if (const char *ev = std::getenv("PCG_SELL_SPLIT_F")) f = std::max(0.5, std::atof(ev));
    if (const char *ev = std::getenv("PCG_SELL_SPLIT_WINDOW")) window = (size_t)std::max(64, std::atoi(ev));
    std::vector<int32_t> wb((size_t)S, 0);
    int64_t base_cols = 0, excess = 0;
    for (int64_t s = 0; s < S; ++s) {
        int32_t srt[64];
        std::copy(&len[(size_t)s * C], &len[(size_t)s * C] + C, srt);
        std::sort(srt, srt + C);
        int64_t suffix = 0;                                                  // sum of the lengths above position i
        double best = -1;
        int32_t best_w = srt[C - 1];
        for (int i = C - 1; i >= 0; --i) {
            if (i == C - 1 || srt[i] != srt[i + 1]) {
                const double cost = 64.0 * srt[i] + f * (double)(suffix - (int64_t)(C - 1 - i) * srt[i]);
                if (best < 0 || cost < best) { best = cost; best_w = srt[i]; }
            }
            suffix += srt[i];
        }
        wb[s] = best_w;
        base_cols += best_w;
        for (int l = 0; l < C; ++l) excess += std::max(0, len[(size_t)s * C + l] - best_w);
    }
    if ((double)base_cols * C + 1.3 * (double)excess > (1.0 - min_saving) * (double)tot * C) return false;
    // overflow rows: ascending row order, the interface rows' slices first; windows of 512 rows sorted by excess (stable)
    struct Row { int32_t row, exc; };
    std::vector<Row> rows;
    m.ov_slice_ptr.assign(1, 0);
    m.ov_mask.assign((size_t)S, 0);
    // Windowed form (PCG_SPMV_OVF=split keeps the round-3 form: windows of 512 overflow ROWS, a second launch): the base
    // slices of a range are cut into k x target_blocks windows of (nearly) equal slice counts - every workgroup of k_spmv_win then
    // works through exactly k of them - of about 12 slices each (at least 8: a window's overflow rows should fill a slice or two).
    // Measured (profiles/r04_ab_win_*_sessionB.log, graded octree mesh, same process): at 10 M dof the windowed form is SLOWER - 1727 us
    // with 12-slice windows (1715 with 6, 1786 with 24) against 1599 us for the two launches: the block barrier between the two
    // phases of a window and the 2-3 overflow slices left for four waves cost more than the gathers gain; at 1 M dof (5 129 slices:
    // every launch is a single wave of work per CU) 4-slice windows win, 176 against 187 us.  So: windows of 4 slices below 8 192
    // slices, the two-launch form above; PCG_SPMV_OVF=win / split and PCG_SPMV_OVF_WINDOW override.
    bool windowed = S < 8192;
    int64_t per_win = windowed ? 4 : 12;
    if (const char *ev = std::getenv("PCG_SPMV_OVF")) windowed = std::string(ev) != "split";
    if (const char *ev = std::getenv("PCG_SPMV_OVF_WINDOW")) per_win = std::max(1, std::atoi(ev));
    auto emit_range = [&](int64_t s_lo, int64_t s_hi) {
        if (s_hi <= s_lo) return;                                            // (no interface slices: no empty window either)
        std::vector<int64_t> cuts;                                           // window boundaries (base slices) of this range
        if (windowed && s_hi > s_lo) {
            const int64_t n = s_hi - s_lo;
            int64_t nw = std::max<int64_t>(1, (n + per_win - 1) / per_win);
            if (nw > target_blocks) nw = (nw + target_blocks / 2) / target_blocks * target_blocks;   // a whole number of windows per workgroup
            nw = std::min(nw, n);
            for (int64_t w = 0; w <= nw; ++w) cuts.push_back(s_lo + (n * w) / nw);
        } else {
            cuts = {s_lo, s_hi};
        }
        for (size_t wdx = 0; wdx + 1 < cuts.size(); ++wdx) {
        rows.clear();
        for (int64_t s = cuts[wdx]; s < cuts[wdx + 1]; ++s)
            for (int l = 0; l < C; ++l) {
                const int32_t e = len[(size_t)s * C + l] - wb[s];
                if (e > 0) { rows.push_back(Row{(int32_t)(s * C + l), e}); m.ov_mask[s] |= 1ull << l; }
            }
        const size_t sort_window = windowed ? std::max<size_t>(1, rows.size()) : window;
        for (size_t a = 0; a < rows.size(); a += sort_window)
            std::stable_sort(rows.begin() + a, rows.begin() + std::min(rows.size(), a + sort_window), [](const Row &x, const Row &y) { return x.exc > y.exc; });
        for (size_t a = 0; a < rows.size(); a += C) {
            const size_t b = std::min(rows.size(), a + C);
            // the 64 rows of a slice in ascending row order again: neighbouring lanes gather neighbouring x (the width of the
            // slice is its longest row whatever the order of its lanes)
            if (!std::getenv("PCG_SELL_SPLIT_KEEP_SORTED"))
                std::sort(rows.begin() + a, rows.begin() + b, [](const Row &x, const Row &y) { return x.row < y.row; });
            int32_t w = 0;
            for (size_t q = a; q < b; ++q) w = std::max(w, rows[q].exc);
            const int64_t ob = m.ov_slice_ptr.back();
            m.ov_slice_ptr.push_back(ob + w);
            m.ov_rows.resize(m.ov_rows.size() + C, -1);
            m.ov_cols.resize((size_t)(ob + w) * C, 0);
            m.ov_vals.resize((size_t)(ob + w) * C * 9, 0.0);
            int32_t *orow = &m.ov_rows[m.ov_rows.size() - C];
            for (int l = 0; l < C; ++l) {
                const bool live = a + l < b;
                const int32_t r = live ? rows[a + l].row : rows[b - 1].row;  // padding lanes: zero blocks at a valid column
                if (live) orow[l] = r;
                const int64_t sb = m.slice_ptr[r / C] + wb[r / C];
                for (int32_t j = 0; j < w; ++j) {
                    const size_t dq = (size_t)(ob + j) * C + l;
                    if (live && j < rows[a + l].exc) {
                        m.ov_cols[dq] = m.cols[(size_t)(sb + j) * C + r % C];
                        for (int c = 0; c < 9; ++c)
                            m.ov_vals[((size_t)(ob + j) * 9 + c) * C + l] = m.vals[((size_t)(sb + j) * 9 + c) * C + r % C];
                    } else {
                        m.ov_cols[dq] = r;
                    }
                }
            }
        }
        if (windowed) {
            if (m.win_slice.empty()) { m.win_slice.push_back(cuts[wdx]); m.win_ov.push_back(0); }
            m.win_slice.push_back(cuts[wdx + 1]);
            m.win_ov.push_back((int64_t)m.ov_slice_ptr.size() - 1);
        }
        }
    };
    m.win_slice.clear(); m.win_ov.clear();
    emit_range(0, m.n_bnd_slices);
    m.ov_bnd_slices = (int64_t)m.ov_slice_ptr.size() - 1;
    m.n_bnd_windows = windowed && !m.win_slice.empty() ? (int64_t)m.win_slice.size() - 1 : 0;
    emit_range(m.n_bnd_slices, S);
    m.ov_slices = (int64_t)m.ov_slice_ptr.size() - 1;
    if (windowed && m.win_slice.empty()) { m.win_slice = {0, S}; m.win_ov = {0, m.ov_slices}; }
    // compact the base arrays in place (a slice keeps its first wb block columns)
    int64_t dst = 0;
    for (int64_t s = 0; s < S; ++s) {
        const int64_t src = m.slice_ptr[s];
        if (dst != src && wb[s] > 0) {
            std::memmove(&m.cols[(size_t)dst * C], &m.cols[(size_t)src * C], sizeof(int32_t) * (size_t)wb[s] * C);
            std::memmove(&m.vals[(size_t)dst * C * 9], &m.vals[(size_t)src * C * 9], sizeof(double) * (size_t)wb[s] * C * 9);
        }
        m.slice_ptr[s] = dst;
        dst += wb[s];
    }
    m.slice_ptr[S] = dst;
    m.cols.resize((size_t)dst * C);
    m.vals.resize((size_t)dst * C * 9);
    m.cols.shrink_to_fit();
    m.vals.shrink_to_fit();
    return true;
}

bool BlockKey::operator==(const BlockKey &o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
bool BlockKey::operator<(const BlockKey &o) const
{
    for (int c = 0; c < 9; ++c)
        if (w[c] != o.w[c]) return w[c] < o.w[c];
    return false;
}
size_t BlockHash::operator()(const BlockKey &k) const
{
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int c = 0; c < 9; ++c) {
        h ^= k.w[c] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
    }
    return (size_t)h;
}

int32_t BlockTable::add(const BlockKey &k)
{
    auto it = tab_.find(k);
    if (it != tab_.end()) {
        count[it->second] += 1;
        return (int32_t)it->second;
    }
    if ((int64_t)keys.size() >= cap_) return -1;
    const uint32_t id = (uint32_t)keys.size();
    keys.push_back(k);
    count.push_back(1);
    tab_.emplace(k, id);
    return (int32_t)id;
}

bool finish_dictionary(SellHost &m, std::vector<BlockTable> &local, const std::vector<std::pair<size_t, size_t>> &slots,
                       std::vector<uint16_t> &bidx, int64_t max_unique)
{
    const int nt = (int)local.size();
    std::vector<BlockKey> all;
    for (const auto &d : local) all.insert(all.end(), d.keys.begin(), d.keys.end());
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    if ((int64_t)all.size() > max_unique) return false;
    // final order: most frequent block first (the device keeps the head of the table in LDS when the whole does not fit),
    // ties by bit pattern - a function of the matrix alone, not of the thread count or the scan order
    std::vector<int64_t> count(all.size(), 0);
    for (int t = 0; t < nt; ++t)
        for (size_t i = 0; i < local[t].keys.size(); ++i)
            count[(size_t)(std::lower_bound(all.begin(), all.end(), local[t].keys[i]) - all.begin())] += local[t].count[i];
    std::vector<uint32_t> order(all.size()), rank(all.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return count[a] > count[b]; });
    for (size_t r = 0; r < order.size(); ++r) rank[order[r]] = (uint32_t)r;
    std::vector<std::vector<uint16_t>> remap((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        remap[t].resize(local[t].keys.size());
        for (size_t i = 0; i < local[t].keys.size(); ++i)
            remap[t][i] = (uint16_t)rank[(size_t)(std::lower_bound(all.begin(), all.end(), local[t].keys[i]) - all.begin())];
    }
    auto apply = [&](int t) {
        for (size_t i = slots[t].first; i < slots[t].second; ++i) bidx[i] = remap[t][bidx[i]];
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(apply, t);
        apply(0);
        for (auto &x : th) x.join();
    }
    m.dict.resize(all.size() * 9);
    m.dict_count.resize(all.size());
    for (size_t r = 0; r < order.size(); ++r) {
        std::memcpy(&m.dict[r * 9], all[order[r]].w, sizeof(all[order[r]].w));
        m.dict_count[r] = count[order[r]];
    }
    m.bidx.swap(bidx);
    std::vector<double>().swap(m.vals);                                // the values now live in the dictionary only
    return true;
}

bool compress_blocks(SellHost &m, int64_t max_unique, int n_threads)
{
    if (m.bs != 3 || m.C != 64) return false;
    max_unique = std::min<int64_t>(max_unique, 65535);
    const int C = m.C;
    const int64_t tot = m.slice_ptr[m.n_slices];
    if (m.vals.size() != (size_t)tot * C * 9) return false;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, m.n_slices / 64 + 1));
    std::vector<uint16_t> bidx((size_t)tot * C);
    std::vector<BlockTable> local((size_t)nt, BlockTable(max_unique));
    std::vector<std::pair<size_t, size_t>> slots((size_t)nt);
    std::vector<char> failed((size_t)nt, 0);
    const int64_t chunk = (m.n_slices + nt - 1) / nt;
    auto scan = [&](int t) {
        BlockTable &tab = local[t];
        BlockKey tile[64];
        const int64_t s_lo = std::min(m.n_slices, t * chunk), s_hi = std::min(m.n_slices, s_lo + chunk);
        slots[t] = {(size_t)m.slice_ptr[s_lo] * C, (size_t)m.slice_ptr[s_hi] * C};
        for (int64_t q = m.slice_ptr[s_lo]; q < m.slice_ptr[s_hi]; ++q) {      // q = (slice, k): 64 blocks, value-component major
            for (int c = 0; c < 9; ++c) {
                const double *v = &m.vals[((size_t)q * 9 + c) * C];
                for (int l = 0; l < 64; ++l) std::memcpy(&tile[l].w[c], v + l, 8);
            }
            for (int l = 0; l < 64; ++l) {
                const int32_t id = tab.add(tile[l]);
                if (id < 0) { failed[t] = 1; return; }
                bidx[(size_t)q * C + l] = (uint16_t)id;
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(scan, t);
        scan(0);
        for (auto &x : th) x.join();
    }
    for (char f : failed)
        if (f) return false;
    return finish_dictionary(m, local, slots, bidx, max_unique);
}

// scalar rows: the same slice layout with 1 value per entry (vals[(slice_ptr[s]+k)*C + lane]); duplicates summed
void csr_to_sell1(int64_t n, const int64_t *rowptr, const int32_t *cols, const double *vals, int64_t n_boundary_rows,
                  int n_threads, SellHost &out)
{
    (void)n_threads;
    const int C = 64;
    out.bs = 1;
    out.n_nodes = n;
    out.C = C;
    out.n_slices = (n + C - 1) / C;
    out.nnzb = rowptr[n];
    out.n_bnd_slices = std::min<int64_t>(out.n_slices, (n_boundary_rows + C - 1) / C);
    out.slice_ptr.assign(out.n_slices + 1, 0);
    for (int64_t s = 0; s < out.n_slices; ++s) {
        int64_t w = 0;
        for (int64_t r = s * C; r < std::min<int64_t>(n, (s + 1) * C); ++r) w = std::max<int64_t>(w, rowptr[r + 1] - rowptr[r]);
        out.slice_ptr[s + 1] = out.slice_ptr[s] + w;
    }
    const int64_t tot = out.slice_ptr[out.n_slices];
    out.cols.assign((size_t)tot * C, 0);
    out.vals.assign((size_t)tot * C, 0.0);
    out.diag.assign((size_t)n, 0.0);
    for (int64_t s = 0; s < out.n_slices; ++s) {
        const int64_t base = out.slice_ptr[s], w = out.slice_ptr[s + 1] - base;
        for (int l = 0; l < C; ++l) {
            const int64_t r = s * C + l;
            const bool live = r < n;
            const int64_t r0 = live ? rowptr[r] : 0, len = live ? rowptr[r + 1] - r0 : 0;
            for (int64_t k = 0; k < w; ++k) {
                const size_t i = (size_t)(base + k) * C + l;
                if (k < len) {
                    if (cols[r0 + k] < 0 || cols[r0 + k] >= n) throw std::runtime_error("csr: column index out of range");
                    out.cols[i] = cols[r0 + k];
                    out.vals[i] = vals[r0 + k];
                    if (cols[r0 + k] == r) out.diag[r] += vals[r0 + k];
                } else {
                    out.cols[i] = live ? (int32_t)r : 0;
                }
            }
        }
    }
}

}  // namespace pcg
