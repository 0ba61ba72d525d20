// Host-side set-up of the matrix-free operator: greedy element colouring, (phase, colour) sorting,
// packing of the per-group tables, local diag(A).  See EbeHost in pcg_internal.hpp.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>

#include "pcg_internal.hpp"

namespace pcg {

void build_ebe(int64_t n_nodes, int32_t n_groups, const pcg_elem_group *gs, const int64_t *perm,
               int64_t n_boundary_nodes, EbeHost &out)
{
    out = EbeHost();
    out.n_nodes = n_nodes;
    out.groups.resize(n_groups);
    out.diag.assign((size_t)n_nodes * 3, 0.0);
    // node -> bit mask of the colours already used by its elements (greedy, elements in (group, e) order)
    std::vector<uint64_t> used((size_t)n_nodes, 0);
    std::vector<std::vector<uint8_t>> color(n_groups), phase(n_groups);
    std::vector<int64_t> nodes_of;
    int maxc[2] = {-1, -1};
    for (int g = 0; g < n_groups; ++g) {
        const auto &in = gs[g];
        if (in.nd <= 0 || in.nd > 255) throw std::runtime_error("ebe: nd out of range (1..255)");
        color[g].resize((size_t)in.ne);
        phase[g].resize((size_t)in.ne);
        nodes_of.resize(in.nd);
        for (int64_t e = 0; e < in.ne; ++e) {
            uint64_t forbidden = 0;
            bool bnd = false;
            for (int a = 0; a < in.nd; ++a) {
                int64_t d = in.dof[(int64_t)a * in.ne + e];
                int64_t node = d / 3;
                if (d < 0 || node >= n_nodes) throw std::runtime_error("ebe: dof index out of range");
                if (perm) node = perm[node];
                nodes_of[a] = node;
                forbidden |= used[node];
                bnd |= node < n_boundary_nodes;
            }
            if (~forbidden == 0) throw std::runtime_error("ebe: more than 64 colours needed");
            int c = __builtin_ctzll(~forbidden);
            for (int a = 0; a < in.nd; ++a) used[nodes_of[a]] |= (1ull << c);
            color[g][e] = (uint8_t)c;
            phase[g][e] = bnd ? 0 : 1;
            maxc[bnd ? 0 : 1] = std::max(maxc[bnd ? 0 : 1], c);
        }
    }
    const int n_col = std::max(maxc[0], maxc[1]) + 1;
    out.n_colors[0] = maxc[0] + 1;
    out.n_colors[1] = maxc[1] + 1;
    // per group: stable sort by (phase, colour), pack
    std::vector<std::vector<int64_t>> bucket_start(n_groups);
    for (int g = 0; g < n_groups; ++g) {
        const auto &in = gs[g];
        auto &G = out.groups[g];
        G.nd = in.nd; G.ne = in.ne;
        G.ke.assign(in.ke, in.ke + (size_t)in.nd * in.nd);
        std::vector<int64_t> cnt((size_t)2 * n_col + 1, 0);
        for (int64_t e = 0; e < in.ne; ++e) cnt[(size_t)phase[g][e] * n_col + color[g][e] + 1]++;
        for (size_t k = 0; k + 1 < cnt.size(); ++k) cnt[k + 1] += cnt[k];
        bucket_start[g] = cnt;
        std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1), newpos((size_t)in.ne);
        for (int64_t e = 0; e < in.ne; ++e) newpos[e] = cur[(size_t)phase[g][e] * n_col + color[g][e]]++;
        G.dof.resize((size_t)in.nd * in.ne);
        G.sign.resize((size_t)in.nd * in.ne);
        G.ck.resize((size_t)in.ne);
        for (int64_t e = 0; e < in.ne; ++e) G.ck[newpos[e]] = in.ck[e];
        for (int a = 0; a < in.nd; ++a) {
            const int64_t *src = in.dof + (int64_t)a * in.ne;
            const uint8_t *ss = in.sign + (int64_t)a * in.ne;
            int32_t *dd = &G.dof[(size_t)a * in.ne];
            uint8_t *ds = &G.sign[(size_t)a * in.ne];
            const double kaa = in.ke[(size_t)a * in.nd + a];
            for (int64_t e = 0; e < in.ne; ++e) {
                int64_t d = src[e], node = d / 3;
                if (perm) node = perm[node];
                const int64_t nd_new = 3 * node + d % 3;
                dd[newpos[e]] = (int32_t)nd_new;
                ds[newpos[e]] = ss[e] ? 1 : 0;
                out.diag[nd_new] += in.ck[e] * kaa;            // pcg_solver.py:282-287 (signs cancel on the diagonal)
            }
        }
        out.n_elem += in.ne;
        out.n_slots += (int64_t)in.nd * in.ne;
    }
    // launch ranges: per phase, colour-major, groups inside a colour
    for (int ph = 0; ph < 2; ++ph)
        for (int c = 0; c < n_col; ++c)
            for (int g = 0; g < n_groups; ++g) {
                const int64_t lo = bucket_start[g][(size_t)ph * n_col + c], hi = bucket_start[g][(size_t)ph * n_col + c + 1];
                if (hi > lo) out.ranges[ph].push_back(EbeRange{g, lo, hi});
            }
}

}  // namespace pcg
