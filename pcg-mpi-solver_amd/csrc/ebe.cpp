// Host-side set-up of the matrix-free operator (EbeHost in pcg_internal.hpp):
//   * hex8-like groups (nd == 24, node-blocked slots) -> spatially clustered 256-element chunks with
//     LDS node tiles, sub-colours inside a chunk, chunk colours across launches (build_chunked);
//   * every other pattern type -> greedy element colouring, one conflict-free launch per colour;
//   * local diag(A) in the reference's own accumulation order (pcg_solver.py:282-300).
// Everything is deterministic: orders depend only on the input tables.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>
#include <stdexcept>

#include "pcg_internal.hpp"

namespace pcg {

namespace {

struct ElemRef { uint64_t key; int32_t g; int64_t e; };

inline uint64_t spread21(uint64_t v)
{
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// 1: slots 3l, 3l+1, 3l+2 are the x, y, z dofs of one node, at most 32 nodes per element.
// 2: they are the three dofs of one node in an order of the ELEMENT's own (the same for all its nodes): the reference's pattern
//    library stores one matrix per pattern up to the cube's 48 symmetries and an element of another orientation lists its dofs in
//    the order of the canonical pattern's (LocDofVector, partition_mesh.py:453, with its sign vector :455) - such groups are handled
//    by the tiles of the mixed chunks only.     0: neither.
int node_blocked(const pcg_elem_group &g)
{
    if (g.nd % 3 != 0 || g.nd > 96 || g.nd < 3) return 0;
    int kind = 1;
    for (int l = 0; l < g.nd / 3; ++l) {
        const int64_t *d0 = g.dof + (int64_t)(3 * l) * g.ne, *d1 = d0 + g.ne, *d2 = d1 + g.ne;
        for (int64_t e = 0; e < g.ne; ++e) {
            if (d0[e] % 3 == 0 && d1[e] == d0[e] + 1 && d2[e] == d0[e] + 2) continue;
            const int64_t n = d0[e] / 3;
            const int c0 = (int)(d0[e] % 3), c1 = (int)(d1[e] % 3), c2 = (int)(d2[e] % 3);
            if (d1[e] / 3 != n || d2[e] / 3 != n || c0 == c1 || c0 == c2 || c1 == c2) return 0;
            if (g.dof[e] % 3 != c0 || g.dof[g.ne + e] % 3 != c1 || g.dof[2 * g.ne + e] % 3 != c2) return 0;   // node 0's order
            kind = 2;
        }
    }
    if (kind == 2)                       // (an element whose node 0 is in order but a later one is not)
        for (int l = 1; l < g.nd / 3; ++l)
            for (int64_t e = 0; e < g.ne; ++e)
                for (int c = 0; c < 3; ++c)
                    if (g.dof[(int64_t)(3 * l + c) * g.ne + e] % 3 != g.dof[(int64_t)c * g.ne + e] % 3) return 0;
    return kind;
}


// Pattern types of the mixed chunks (EbeMixedHost): the most populous 8-node type becomes the hex section, every other
// node-blocked type a tile type with its A-operand fragments.  v_mfma_f64_16x16x4_f64: A lane (g, i) holds A[i][k = g], B lane
// (g, e) holds B[k = g][e], D lane (g, e) register r holds D[i = g + 4 r][e] (g = lane >> 4).  Rows and columns of Ke are
// PERMUTED so that lane group g owns whole nodes: k-step ks = 3 j + c carries dof 3 (4 j + g) + c, and permuted row
// 16 mt + g + 4 r = g + 4 q (q = 4 mt + r = 3 j' + c') is dof 3 (4 j' + g) + c'.  Dofs of nodes >= nn are zero rows / columns.
bool build_mixed_types(int32_t n_groups, const pcg_elem_group *gs, const std::vector<char> &chunkable, EbeChunkedHost &C)
{
    auto &M = C.mixed;
    int64_t best = 0;
    // PCG_EBE_HEX_TILES: 1 = no hex section, the standard 8-node type runs on the matrix cores in colour-pure tiles (EbeMixedHost);
    // 2 (development) = as ordinary tiles in run order; 0 = the hex section on the vector FMAs.
    // Default (measured, sessions q / r): the hex tiles win where the chunks do not fill the GPU twice over - a chunk's life is then the
    // latency of its chain of phases, and the hex section's chain (two passes, eight ordered turns of 2.3 k cycles) is the longer one
    // (1 M-dof octree mesh, 661 chunks: 45 vs 50 us); with many chunks in flight per CU the vector FMAs and the matrix cores working
    // side by side win (10 M dof: 259 vs 272 us).
    const char *hv = std::getenv("PCG_EBE_HEX_TILES");
    int64_t chunkable_elems = 0;
    for (int g = 0; g < n_groups; ++g) if (chunkable[g]) chunkable_elems += gs[g].ne;
    const int hex_tiles = hv ? std::atoi(hv) : (chunkable_elems < kMixedHexTilesBelow ? 1 : 0);
    int64_t best_any = 0;
    int any_group = -1;                                  // the most populous 8-node type, whatever the dof order of its elements
    for (int g = 0; g < n_groups; ++g) {                 // (chunkable == 2: per-element dof order - tiles only, never the hex section)
        if (chunkable[g] == 1 && gs[g].nd == 24 && gs[g].ne > best) { best = gs[g].ne; M.hex_group = g; }
        if (chunkable[g] && gs[g].nd == 24 && gs[g].ne > best_any) { best_any = gs[g].ne; any_group = g; }
    }
    int first_group = -1;
    if (hex_tiles && any_group >= 0) { first_group = any_group; M.hex_group = -1; if (hex_tiles == 1) M.hex_tile_type = 0; }
    else if (!hv && M.hex_group < 0 && any_group >= 0) { first_group = any_group; M.hex_tile_type = 0; }   // oriented 8-node elements: the hex section cannot take them
    auto &K = C.cls[kMixedClass];
    K.ke_col.assign(24 * 24, 0.0);
    if (M.hex_group >= 0)
        for (int b = 0; b < 24; ++b)
            for (int a = 0; a < 24; ++a) K.ke_col[(size_t)b * 24 + a] = gs[M.hex_group].ke[(size_t)a * 24 + b];
    bool any = M.hex_group >= 0;
    int max_nn = 4, max_nd = 1;
    std::vector<int> g_order;                            // (the hex tile type is types[0])
    if (first_group >= 0) g_order.push_back(first_group);
    for (int g = 0; g < n_groups; ++g) if (g != first_group) g_order.push_back(g);
    for (int g : g_order) {
        if (!chunkable[g] || g == M.hex_group || gs[g].ne == 0) continue;
        EbeMixedType T;
        T.group = g; T.nd = gs[g].nd; T.nn = gs[g].nd / 3; T.J = (T.nn + 3) / 4;
        const int MT = (3 * T.J + 3) / 4, KS = 3 * T.J;
        T.frag_off = (int64_t)M.frag.size();
        M.frag.resize(M.frag.size() + (size_t)KS * MT * 64, 0.0);
        double *F = M.frag.data() + T.frag_off;
        for (int ks = 0; ks < KS; ++ks)
            for (int mt = 0; mt < MT; ++mt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int gk = lane >> 4, i = lane & 15;
                    const int col_node = 4 * (ks / 3) + gk, col_dof = 3 * col_node + ks % 3;
                    const int q = 4 * mt + (i >> 2), row_node = 4 * (q / 3) + (i & 3), row_dof = 3 * row_node + q % 3;
                    if (col_node < T.nn && row_node < T.nn && q < 3 * T.J)
                        F[((size_t)ks * MT + mt) * 64 + lane] = gs[g].ke[(size_t)row_dof * T.nd + col_dof];
                }
        M.max_mt = std::max(M.max_mt, MT);
        max_nn = std::max(max_nn, T.nn);
        max_nd = std::max(max_nd, T.nd);
        M.types.push_back(T);
        any = true;
    }
    M.nnpt = (max_nn + 3) / 4 * 4;
    M.words = (max_nd + 31) / 32;
    return any;
}

}  // namespace

void build_ebe(int64_t n_nodes, int32_t n_groups, const pcg_elem_group *gs, const int64_t *perm,
               int64_t n_boundary_nodes, const double *coords, bool allow_chunked, int ept, EbeHost &out)
{
    if (ept != 1 && ept != 2) throw std::runtime_error("ebe: elements per thread must be 1 or 2");
    out = EbeHost();
    out.n_nodes = n_nodes;
    out.groups.resize(n_groups);
    out.diag.assign((size_t)n_nodes * 3, 0.0);

    // ---- validation, totals, diag(A) (a-major then e: exactly np.bincount's order, :294-300) ---------
    std::vector<char> chunkable(n_groups, 0);
    for (int g = 0; g < n_groups; ++g) {
        const auto &in = gs[g];
        if (in.nd <= 0 || in.nd > 255) throw std::runtime_error("ebe: nd out of range (1..255)");
        for (int a = 0; a < in.nd; ++a) {
            const int64_t *src = in.dof + (int64_t)a * in.ne;
            const double kaa = in.ke[(size_t)a * in.nd + a];
            for (int64_t e = 0; e < in.ne; ++e) {
                int64_t d = src[e], node = d / 3;
                if (d < 0 || node >= n_nodes) throw std::runtime_error("ebe: dof index out of range");
                if (perm) node = perm[node];
                out.diag[3 * node + d % 3] += in.ck[e] * kaa;      // signs cancel on the diagonal
            }
        }
        out.n_elem += in.ne;
        out.n_slots += (int64_t)in.nd * in.ne;
        chunkable[g] = allow_chunked ? (char)node_blocked(in) : 0;
    }
    // Mixed-type chunks (EbeMixedHost): the default as soon as two node-blocked pattern types have elements, or one whose elements
    // carry their own dof order (tiles only); PCG_EBE_MIXED=0 keeps one chunk list per type (the round-3 form, A/B; groups with a
    // dof order of their own then take the generic colour launches), =1 forces the mixed form for a single type too (tests).
    bool mixed_mode = false;
    {
        int populated = 0, oriented = 0;
        for (int g = 0; g < n_groups; ++g) { populated += chunkable[g] && gs[g].ne > 0; oriented += chunkable[g] == 2 && gs[g].ne > 0; }
        const char *mv = std::getenv("PCG_EBE_MIXED");
        mixed_mode = mv ? (mv[0] != '0' && populated >= 1) : (populated >= 2 || oriented >= 1);
        if (!mixed_mode)
            for (int g = 0; g < n_groups; ++g) if (chunkable[g] == 2) chunkable[g] = 0;
    }

    // ---- one global, spatially coherent element order (Morton code of the first node, or its id) ----
    std::vector<ElemRef> order;
    order.reserve((size_t)out.n_elem);
    double lo[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    if (coords) {
        // lattice units: the smallest element edge (octree meshes have power-of-two multiples of it), so that
        // Morton cells line up with the element lattice; fall back to stretching the box over 21 bits.
        double hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = coords[d]; hi[d] = coords[d]; }
        for (int64_t i = 0; i < n_nodes; ++i)
            for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], coords[3 * i + d]); hi[d] = std::max(hi[d], coords[3 * i + d]); }
        double hmin = 0.0;
        for (int g = 0; g < n_groups; ++g) {
            const auto &in = gs[g];
            for (int64_t e = 0; e < in.ne; ++e) {
                double a0[3], a1[3];
                for (int d = 0; d < 3; ++d) { a0[d] = 1e300; a1[d] = -1e300; }
                for (int a = 0; a < in.nd; a += 3) {
                    const double *c = coords + 3 * (in.dof[(int64_t)a * in.ne + e] / 3);
                    for (int d = 0; d < 3; ++d) { a0[d] = std::min(a0[d], c[d]); a1[d] = std::max(a1[d], c[d]); }
                }
                for (int d = 0; d < 3; ++d)
                    if (a1[d] > a0[d] && (hmin == 0.0 || a1[d] - a0[d] < hmin)) hmin = a1[d] - a0[d];
            }
        }
        for (int d = 0; d < 3; ++d) {
            const double ext = hi[d] - lo[d];
            if (hmin > 0.0 && ext / hmin < 2097151.0) inv[d] = 1.0 / hmin;
            else inv[d] = ext > 0 ? 2097151.0 / ext : 0.0;
        }
    }
    for (int g = 0; g < n_groups; ++g) {
        const auto &in = gs[g];
        for (int64_t e = 0; e < in.ne; ++e) {
            int64_t mn = in.dof[e] / 3;
            for (int a = 1; a < in.nd; ++a) mn = std::min(mn, in.dof[(int64_t)a * in.ne + e] / 3);
            uint64_t key;
            if (coords) {
                const double *c = coords + 3 * mn;
                key = spread21((uint64_t)((c[0] - lo[0]) * inv[0] + 1e-6)) | spread21((uint64_t)((c[1] - lo[1]) * inv[1] + 1e-6)) << 1 |
                      spread21((uint64_t)((c[2] - lo[2]) * inv[2] + 1e-6)) << 2;
            } else {
                key = (uint64_t)mn;
            }
            order.push_back(ElemRef{key, g, e});
        }
    }
    std::stable_sort(order.begin(), order.end(), [](const ElemRef &a, const ElemRef &b) { return a.key < b.key; });

    auto new_node = [&](int64_t d) { int64_t n = d / 3; return perm ? perm[n] : n; };

    // ================= generic path: element colouring over the non-chunkable groups =================
    {
        // per node: bit mask of the colours used by its elements, W 64-bit words (grown on demand, so a
        // hub node of any valence is fine: it just costs as many launches as it has elements)
        int W = 1;
        std::vector<uint64_t> used((size_t)n_nodes, 0);
        std::vector<std::vector<int32_t>> color(n_groups);
        std::vector<std::vector<uint8_t>> phase(n_groups);
        for (int g = 0; g < n_groups; ++g)
            if (!chunkable[g]) { color[g].resize((size_t)gs[g].ne); phase[g].resize((size_t)gs[g].ne); }
        int maxc = -1;
        std::vector<int64_t> nodes_of;
        std::vector<uint64_t> forb;
        for (const auto &r : order) {
            if (chunkable[r.g]) continue;
            const auto &in = gs[r.g];
            nodes_of.resize(in.nd);
            bool bnd = false;
            for (int a = 0; a < in.nd; ++a) {
                nodes_of[a] = new_node(in.dof[(int64_t)a * in.ne + r.e]);
                bnd |= nodes_of[a] < n_boundary_nodes;
            }
            int c = -1;
            while (c < 0) {
                forb.assign(W, 0);
                for (int a = 0; a < in.nd; ++a)
                    for (int w = 0; w < W; ++w) forb[w] |= used[(size_t)nodes_of[a] * W + w];
                for (int w = 0; w < W && c < 0; ++w)
                    if (~forb[w] != 0) c = 64 * w + __builtin_ctzll(~forb[w]);
                if (c < 0) {                                  // all 64*W colours taken: widen the masks
                    std::vector<uint64_t> wide((size_t)n_nodes * 2 * W, 0);
                    for (int64_t i = 0; i < n_nodes; ++i)
                        for (int w = 0; w < W; ++w) wide[(size_t)i * 2 * W + w] = used[(size_t)i * W + w];
                    used.swap(wide);
                    W *= 2;
                }
            }
            for (int a = 0; a < in.nd; ++a) used[(size_t)nodes_of[a] * W + c / 64] |= (1ull << (c % 64));
            color[r.g][r.e] = c;
            phase[r.g][r.e] = bnd ? 0 : 1;
            maxc = std::max(maxc, c);
        }
        const int n_col = maxc + 1;
        out.n_colors[0] = out.n_colors[1] = n_col;
        std::vector<std::vector<int64_t>> bucket_start(n_groups);
        for (int g = 0; g < n_groups; ++g) {
            const auto &in = gs[g];
            auto &G = out.groups[g];
            G.nd = in.nd;
            if (chunkable[g]) { G.ne = 0; continue; }
            G.ne = in.ne;
            G.ke.assign(in.ke, in.ke + (size_t)in.nd * in.nd);
            std::vector<int64_t> cnt((size_t)2 * n_col + 1, 0);
            for (int64_t e = 0; e < in.ne; ++e) cnt[(size_t)phase[g][e] * n_col + color[g][e] + 1]++;
            for (size_t k = 0; k + 1 < cnt.size(); ++k) cnt[k + 1] += cnt[k];
            bucket_start[g] = cnt;
            // inside a (phase, colour) bucket keep the spatial order
            std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1), newpos((size_t)in.ne);
            for (const auto &r : order)
                if (r.g == g) newpos[r.e] = cur[(size_t)phase[g][r.e] * n_col + color[g][r.e]]++;
            G.dof.resize((size_t)in.nd * in.ne);
            G.sign.resize((size_t)in.nd * in.ne);
            G.ck.resize((size_t)in.ne);
            for (int64_t e = 0; e < in.ne; ++e) G.ck[newpos[e]] = in.ck[e];
            for (int a = 0; a < in.nd; ++a) {
                const int64_t *src = in.dof + (int64_t)a * in.ne;
                const uint8_t *ss = in.sign + (int64_t)a * in.ne;
                int32_t *dd = &G.dof[(size_t)a * in.ne];
                uint8_t *ds = &G.sign[(size_t)a * in.ne];
                for (int64_t e = 0; e < in.ne; ++e) {
                    dd[newpos[e]] = (int32_t)(3 * new_node(src[e]) + src[e] % 3);
                    ds[newpos[e]] = ss[e] ? 1 : 0;
                }
            }
        }
        for (int ph = 0; ph < 2; ++ph)
            for (int c = 0; c < n_col; ++c)
                for (int g = 0; g < n_groups; ++g) {
                    if (chunkable[g]) continue;
                    const int64_t lo_ = bucket_start[g][(size_t)ph * n_col + c], hi_ = bucket_start[g][(size_t)ph * n_col + c + 1];
                    if (hi_ > lo_) out.ranges[ph].push_back(EbeRange{g, lo_, hi_});
                }
    }

    // ================= chunked path ====================================================================
    auto &C = out.chunked;
    std::vector<int32_t> cls_of(n_groups, -1), ke_index(n_groups, -1);
    bool any = false;
    for (int c = 0; c < kChunkClasses; ++c) {
        auto &K = C.cls[c];
        K.nnp = (c == 4 || c == kMixedClass) ? 8 : 8 * (c + 1);
        K.full = c == 0;
        K.ept = c == 0 ? ept : 1;
        K.ce = c == 0 ? kChunkThreads * ept : (c == kMixedClass ? kMixedHexSlots : 64);
        K.words = 3 * K.nnp / 32 + 1;
        K.max_nodes = (c == 0 && ept == 1) ? 512 : kChunkMaxNodes;      // 8x8x4 hex cells -> 405 nodes: a 2-nodes-per-thread tile
        const char *dv = std::getenv("PCG_EBE_DIRECT");                              // =0: node tiles for every class (A/B)
        K.direct = c >= 1 && c <= 3 && !(dv && dv[0] == '0');
    }
    if (mixed_mode) {
        any = build_mixed_types(n_groups, gs, chunkable, C);
        for (int g = 0; g < n_groups; ++g) if (chunkable[g]) cls_of[g] = kMixedClass;
    }
    for (int g = 0; g < n_groups && !mixed_mode; ++g)
        if (chunkable[g]) {
            const int nn = gs[g].nd / 3, nd = gs[g].nd;
            const int c = nn == 8 ? 0 : (nn < 8 ? 4 : (nn + 7) / 8 - 1);
            auto &K = C.cls[c];
            const int ndp = 3 * K.nnp;
            cls_of[g] = c;
            ke_index[g] = (int32_t)(K.ke_col.size() / ((size_t)ndp * ndp));
            for (int b = 0; b < ndp; ++b)
                for (int a = 0; a < ndp; ++a) K.ke_col.push_back(a < nd && b < nd ? gs[g].ke[(size_t)a * nd + b] : 0.0);
            if (c != 0 && !K.direct) {                      // row-split layout: wave w owns rows [w*ndp/4, (w+1)*ndp/4)
                const int rpw = ndp / 4;
                for (int w = 0; w < 4; ++w)
                    for (int b = 0; b < ndp; ++b)
                        for (int a = 0; a < rpw; ++a) {
                            const int ar = w * rpw + a;
                            K.ke_rows.push_back(ar < nd && b < nd ? gs[g].ke[(size_t)ar * nd + b] : 0.0);
                        }
            }
            any = true;
        }
    if (!any) return;
    std::vector<int32_t> stamp((size_t)n_nodes, -1);
    std::vector<uint8_t> chunk_phase;
    struct Open { std::vector<int64_t> elems; std::vector<int32_t> nodes; };
    std::vector<int32_t> local_of((size_t)n_nodes, 0);        // node -> slot in the LDS tile of the chunk being built
    // Order of a chunk's nodes in its LDS tile.  The kernels read / update the tile with lane = element, lanes sorted by
    // sub-colour: on a lattice mesh the elements of one sub-colour share the parity of their cell coordinates, so
    // their k-th nodes all lie in ONE parity class of the node lattice and follow the elements' Morton order there.
    // Tile slots sorted by (parity class, Morton code of the half coordinates) therefore give consecutive lanes
    // consecutive slots (stride 24 B: no LDS bank conflict inside 16 lanes) instead of the stride-48 B, 4-way
    // conflicting pattern of node-id order (counters: 70 % of the LDS-active cycles were conflict stalls).
    auto tile_key = [&](int64_t orig_node, int64_t new_id) -> uint64_t {
        if (!coords) return (uint64_t)new_id;
        const double *c = coords + 3 * orig_node;
        const uint64_t ix = (uint64_t)((c[0] - lo[0]) * inv[0] + 1e-6), iy = (uint64_t)((c[1] - lo[1]) * inv[1] + 1e-6),
                       iz = (uint64_t)((c[2] - lo[2]) * inv[2] + 1e-6);
        const uint64_t par = (ix & 1) | (iy & 1) << 1 | (iz & 1) << 2;
        return par << 60 | ((spread21(ix >> 1) | spread21(iy >> 1) << 1 | spread21(iz >> 1) << 2) & ((1ull << 60) - 1));
    };
    // sub-colour the elements of a candidate chunk (greedy, element order); false if > 63 colours are needed
    auto sub_colour = [&](int g, const Open &o, std::vector<int> &sc, std::vector<uint16_t> &lids, int &nsub, bool &bnd) {
        const auto &in = gs[g];
        const int nno = in.nd / 3;
        const int nn = (int)o.nodes.size(), ne = (int)o.elems.size();
        std::vector<uint64_t> used(nn, 0);
        sc.resize(ne); lids.resize((size_t)ne * nno);
        nsub = 0; bnd = false;
        for (int t = 0; t < ne; ++t) {
            uint64_t forb = 0;
            for (int l = 0; l < nno; ++l) {
                const int64_t node = new_node(in.dof[(int64_t)(3 * l) * in.ne + o.elems[t]]);
                const int li = local_of[node];
                lids[(size_t)t * nno + l] = (uint16_t)li;
                forb |= used[li];
                bnd |= node < n_boundary_nodes;
            }
            forb |= 1ull << 63;                               // keep the code below 255 (= padding marker) and 64 bits
            if (~forb == 0) return false;
            const int c = __builtin_ctzll(~forb);
            for (int l = 0; l < nno; ++l) used[lids[(size_t)t * nno + l]] |= (1ull << c);
            sc[t] = c;
            nsub = std::max(nsub, c + 1);
        }
        return true;
    };
    auto close_chunk = [&](int g, Open &o, std::vector<int> &sc, std::vector<uint16_t> &lids, int nsub, bool bnd) {
        if (o.elems.empty()) return;
        const auto &in = gs[g];
        auto &K = C.cls[cls_of[g]];
        const int nno = in.nd / 3, CE = K.ce, W = K.words;
        const int32_t cid = (int32_t)C.n_chunks++;
        const int32_t kci = (int32_t)K.n_chunks++;
        const int nn = (int)o.nodes.size();
        const int ne = (int)o.elems.size();
        // slot of every element: sorted by sub-colour (stable) so that whole waves share a phase; in the hex8 class
        // every sub-colour group starts on a multiple of 16 slots, i.e. a 16-element MFMA tile is sub-colour pure
        std::vector<int> lane_of(ne);
        std::iota(lane_of.begin(), lane_of.end(), 0);
        std::stable_sort(lane_of.begin(), lane_of.end(), [&](int a, int b) { return sc[a] < sc[b]; });
        std::vector<int> slot_of(ne);
        int next_slot = 0;
        for (int k = 0; k < ne; ++k) {
            if (K.full && k > 0 && sc[lane_of[k]] != sc[lane_of[k - 1]]) next_slot = (next_slot + 15) / 16 * 16;
            slot_of[k] = next_slot++;
        }
        const int n_tiles = (next_slot + 15) / 16;
        C.hdr.insert(C.hdr.end(), {(int32_t)C.nodes.size(), nn, nsub, ke_index[g], kci, in.nd, cls_of[g], n_tiles});
        {                                                    // global side in address order, LDS side in slot order
            std::vector<int32_t> by_id(o.nodes);
            std::sort(by_id.begin(), by_id.end());
            for (int32_t nd_ : by_id) { C.nodes.push_back(nd_); C.tslot.push_back((uint16_t)local_of[nd_]); }
        }
        C.max_subcolors = std::max(C.max_subcolors, nsub);
        chunk_phase.push_back(bnd ? 0 : 1);
        K.list[bnd ? 0 : 1].push_back(cid);
        const size_t base = (size_t)kci * CE;
        K.ck.resize(base + CE, 0.0);
        K.sgn.resize((size_t)(kci + 1) * W * CE, 0u);
        K.lid.resize((size_t)(kci + 1) * K.nnp * CE, 0);
        for (int lane = 0; lane < CE; ++lane)                        // padding slots: sub-colour 255
            K.sgn[((size_t)kci * W + (W - 1)) * CE + lane] = 0xff000000u;
        for (int k = 0; k < ne; ++k) {
            const int t = lane_of[k], lane = slot_of[k];
            const int64_t e = o.elems[t];
            K.ck[base + lane] = in.ck[e];
            for (int w = 0; w < W; ++w) {
                uint32_t bits = 0;
                for (int a = 32 * w; a < std::min(in.nd, 32 * w + 32); ++a)
                    if (in.sign[(int64_t)a * in.ne + e]) bits |= (1u << (a - 32 * w));
                if (w == W - 1) bits |= ((uint32_t)sc[t] << 24);
                K.sgn[((size_t)kci * W + w) * CE + lane] = bits;
            }
            for (int l = 0; l < nno; ++l) K.lid[((size_t)kci * K.nnp + l) * CE + lane] = lids[(size_t)t * nno + l];
        }
        o.elems.clear();
        o.nodes.clear();
    };

    // ================= mixed-type chunks: greedy runs of the global Morton order ===========================
    if (mixed_mode) {
        auto &M = C.mixed;
        auto &K = C.cls[kMixedClass];
        int hex_cap = kChunkThreads * ept, tile_cap = ept == 2 ? kMixedMaxTiles : kMixedMaxTiles / 2;
        if (const char *tv = std::getenv("PCG_EBE_TILE_CAP")) tile_cap = std::max(1, std::atoi(tv));          // development knobs (tools/iter_ab.py)
        if (const char *hv = std::getenv("PCG_EBE_HEX_CAP")) hex_cap = std::max(16, std::min(kMixedHexSlots, std::atoi(hv)));
        std::vector<int32_t> type_of(n_groups, -1);
        for (size_t t = 0; t < M.types.size(); ++t) type_of[M.types[t].group] = (int32_t)t;
        std::vector<ElemRef> L;
        for (const auto &r : order)
            if (chunkable[r.g]) L.push_back(r);
        int32_t next_stamp = 0;
        std::vector<std::pair<uint64_t, int32_t>> keyed;
        std::vector<uint64_t> used;
        std::vector<int> sc;
        // emit the run [lo_, hi_) as one chunk; false (nothing emitted) when its hex section needs more than 63 sub-colours
        auto emit = [&](size_t lo_, size_t hi_) -> bool {
            const int32_t id = next_stamp++;
            keyed.clear();
            std::vector<size_t> hex_el;
            std::vector<std::vector<size_t>> by_type(M.types.size());
            bool bnd = false;
            for (size_t k = lo_; k < hi_; ++k) {
                const auto &in = gs[L[k].g];
                for (int l = 0; l < in.nd / 3; ++l) {
                    const int64_t orig = in.dof[(int64_t)(3 * l) * in.ne + L[k].e] / 3;
                    const int64_t node = perm ? perm[orig] : orig;
                    bnd |= node < n_boundary_nodes;
                    if (stamp[node] != id) { stamp[node] = id; keyed.emplace_back(tile_key(orig, node), (int32_t)node); }
                }
                if (L[k].g == M.hex_group) hex_el.push_back(k); else by_type[type_of[L[k].g]].push_back(k);
            }
            std::sort(keyed.begin(), keyed.end());
            const int nn = (int)keyed.size();
            for (int k = 0; k < nn; ++k) local_of[keyed[k].second] = k;
            auto lid_of = [&](const ElemRef &r, int l) {
                const auto &in = gs[r.g];
                return local_of[new_node(in.dof[(int64_t)(3 * l) * in.ne + r.e])];
            };
            // hex section: greedy sub-colours over its elements, slots sorted by colour (slot order = (pass, wave) order)
            const int nh = (int)hex_el.size();
            used.assign(nn, 0);
            sc.assign(nh, 0);
            int nsub = 0;
            for (int t = 0; t < nh; ++t) {
                uint64_t forb = 1ull << 63;
                for (int l = 0; l < 8; ++l) forb |= used[lid_of(L[hex_el[t]], l)];
                if (~forb == 0) return false;
                const int c = __builtin_ctzll(~forb);
                for (int l = 0; l < 8; ++l) used[lid_of(L[hex_el[t]], l)] |= 1ull << c;
                sc[t] = c;
                nsub = std::max(nsub, c + 1);
            }
            std::vector<int> lane_of(nh);
            std::iota(lane_of.begin(), lane_of.end(), 0);
            std::stable_sort(lane_of.begin(), lane_of.end(), [&](int a, int b) { return sc[a] < sc[b]; });
            const int32_t cid = (int32_t)C.n_chunks++;
            const int32_t kci = (int32_t)K.n_chunks++;
            const int32_t tile0 = (int32_t)M.n_tiles;
            const int CE = kMixedHexSlots;
            K.ck.resize((size_t)(kci + 1) * CE, 0.0);
            K.sgn.resize((size_t)(kci + 1) * CE, 0xff000000u);
            K.lid.resize((size_t)(kci + 1) * 8 * CE, 0);
            for (int k = 0; k < nh; ++k) {
                const ElemRef &r = L[hex_el[lane_of[k]]];
                const auto &in = gs[r.g];
                K.ck[(size_t)kci * CE + k] = in.ck[r.e];
                uint32_t bits = 0;
                for (int a = 0; a < 24; ++a)
                    if (in.sign[(int64_t)a * in.ne + r.e]) bits |= 1u << a;
                K.sgn[(size_t)kci * CE + k] = bits | ((uint32_t)sc[lane_of[k]] << 24);
                for (int l = 0; l < 8; ++l) K.lid[((size_t)kci * 8 + l) * CE + k] = (uint16_t)lid_of(r, l);
            }
            M.hex_elems += nh;
            // tiles: 16 elements of one type, in run order; tile-local colours (the 16 elements add in ONE wave instruction per colour)
            const int W = M.words, NP = M.nnpt;
            // hex tiles: the chunk's elements of the standard 8-node type coloured over the chunk, sorted by colour, one colour per tile
            // (padding at the end of a colour: by_type entry SIZE_MAX)
            std::vector<int32_t> wait_of_tile;               // per tile of this chunk: completed tiles its adds wait for
            int n_hex_tiles = 0;
            if (M.hex_tile_type >= 0 && !by_type[M.hex_tile_type].empty()) {
                auto &E = by_type[M.hex_tile_type];
                const int ne_h = (int)E.size();
                used.assign(nn, 0);
                std::vector<int> col(ne_h, 0);
                int ncol_h = 0;
                for (int t = 0; t < ne_h; ++t) {
                    uint64_t forb = 1ull << 63;
                    for (int l = 0; l < 8; ++l) forb |= used[lid_of(L[E[t]], l)];
                    if (~forb == 0) throw std::runtime_error("ebe: more than 63 elements of the 8-node type at one node");
                    const int c = __builtin_ctzll(~forb);
                    for (int l = 0; l < 8; ++l) used[lid_of(L[E[t]], l)] |= 1ull << c;
                    col[t] = c;
                    ncol_h = std::max(ncol_h, c + 1);
                }
                std::vector<size_t> sorted;
                for (int c = 0; c < ncol_h; ++c) {
                    const int first = (int)sorted.size() / 16;
                    for (int t = 0; t < ne_h; ++t) if (col[t] == c) sorted.push_back(E[t]);
                    while (sorted.size() % 16) sorted.push_back(SIZE_MAX);
                    for (int k = first; k < (int)sorted.size() / 16; ++k) wait_of_tile.push_back(first);
                }
                E.swap(sorted);
                n_hex_tiles = (int)E.size() / 16;
            }
            if (M.hex_tile_type >= 0) M.chunk_hex_tiles.push_back(n_hex_tiles);
            for (size_t t = 0; t < by_type.size(); ++t) {
                const auto &T = M.types[t];
                const auto &in = gs[T.group];
                const bool pure = (int)t == M.hex_tile_type;
                for (size_t b0 = 0; b0 < by_type[t].size(); b0 += 16) {
                    const int cnt = (int)std::min<size_t>(16, by_type[t].size() - b0);
                    if (!pure) wait_of_tile.push_back((int32_t)wait_of_tile.size());
                    const size_t ti = (size_t)M.n_tiles++;
                    M.tile_type.push_back((int32_t)t);
                    M.tlid.resize((ti + 1) * NP * 16, 0);
                    M.tck.resize((ti + 1) * 16, 0.0);
                    M.tsgn.resize((ti + 1) * W * 16, 0u);
                    M.tcol.resize((ti + 1) * 16, 255);
                    M.tperm.resize((ti + 1) * 16, 0 | 1 << 2 | 2 << 4);
                    int ncol = 0;
                    std::vector<std::pair<int, uint32_t>> seen;       // (local slot, colour mask) of the nodes this tile has touched
                    for (int e = 0; e < cnt; ++e) {
                        if (by_type[t][b0 + e] == SIZE_MAX) continue;   // padding slot of a colour-pure tile
                        const ElemRef &r = L[by_type[t][b0 + e]];
                        uint32_t forb = 0;
                        for (int l = 0; l < T.nn; ++l) {
                            const int li = lid_of(r, l);
                            M.tlid[(ti * NP + l) * 16 + e] = (uint16_t)li;
                            for (const auto &sn : seen) if (sn.first == li) forb |= sn.second;
                        }
                        const int c = pure ? 0 : __builtin_ctz(~forb); // at most 16 elements: a colour below 16 is always free
                        for (int l = 0; l < T.nn; ++l) {
                            const int li = lid_of(r, l);
                            bool found = false;
                            for (auto &sn : seen) if (sn.first == li) { sn.second |= 1u << c; found = true; break; }
                            if (!found) seen.emplace_back(li, 1u << c);
                        }
                        M.tcol[ti * 16 + e] = (uint8_t)c;
                        ncol = std::max(ncol, c + 1);
                        M.tck[ti * 16 + e] = in.ck[r.e];
                        M.tperm[ti * 16 + e] = (uint8_t)((in.dof[r.e] % 3) | (in.dof[in.ne + r.e] % 3) << 2 | (in.dof[2 * in.ne + r.e] % 3) << 4);
                        for (int a = 0; a < T.nd; ++a)
                            if (in.sign[(int64_t)a * in.ne + r.e]) M.tsgn[(ti * W + a / 32) * 16 + e] |= 1u << (a % 32);
                    }
                    M.tile_ncol.push_back(ncol);
                    for (int e = 0; e < cnt; ++e) M.tile_elems += by_type[t][b0 + e] != SIZE_MAX;
                }
            }
            M.tile_wait.insert(M.tile_wait.end(), wait_of_tile.begin(), wait_of_tile.end());
            C.hdr.insert(C.hdr.end(), {(int32_t)C.nodes.size(), nn, nsub, nh, kci, (int32_t)(M.n_tiles - tile0), kMixedClass, tile0});
            {
                std::vector<int32_t> by_id(nn);
                for (int k = 0; k < nn; ++k) by_id[k] = keyed[k].second;
                std::sort(by_id.begin(), by_id.end());
                for (int32_t nd_ : by_id) { C.nodes.push_back(nd_); C.tslot.push_back((uint16_t)local_of[nd_]); }
            }
            C.max_subcolors = std::max(C.max_subcolors, nsub);
            chunk_phase.push_back(bnd ? 0 : 1);
            K.list[bnd ? 0 : 1].push_back(cid);
            return true;
        };
        std::vector<std::pair<size_t, size_t>> todo;
        auto emit_or_split = [&](size_t lo_, size_t hi_) {
            todo.assign(1, {lo_, hi_});
            while (!todo.empty()) {                                  // (a hub node of valence > 63 in the hex section: halve the run)
                auto [a, b] = todo.back();
                todo.pop_back();
                if (emit(a, b)) continue;
                if (b - a < 2) throw std::runtime_error("ebe: an element does not fit a mixed chunk");
                todo.emplace_back(a + (b - a) / 2, b);
                todo.emplace_back(a, a + (b - a) / 2);
            }
        };
        // Development knobs (tools/iter_ab.py): PCG_EBE_NODE_CAP = tile nodes a chunk may hold (<= 768); PCG_EBE_MIX_SHELLS=1 = the
        // elements of the main 8-node type and all others in SEPARATE runs of the Morton order (fuller tiles - the hanging-node elements of
        // a transition shell sit together - for more boundary slots: every node between a shell and its hex neighbours is shared).
        int node_cap = kChunkMaxNodes;
        if (const char *nv = std::getenv("PCG_EBE_NODE_CAP")) node_cap = std::max(64, std::min(kChunkMaxNodes, std::atoi(nv)));
        const char *shells_env = std::getenv("PCG_EBE_MIX_SHELLS");
        const bool mix_shells = shells_env && shells_env[0] == '1';
        if (mix_shells && M.hex_group >= 0)
            std::stable_partition(L.begin(), L.end(), [&](const ElemRef &r) { return r.g == M.hex_group; });
        std::vector<int32_t> cnt_type(M.types.size(), 0);
        // cut the Morton order into runs that respect the caps; `out(lo, hi)` takes each run (emit it, or just count it)
        auto cut_runs = [&](int cap, auto &&out) {
            std::fill(cnt_type.begin(), cnt_type.end(), 0);
            size_t lo_ = 0;
            int n_run_nodes = 0, n_hex = 0, n_tl = 0;
            int32_t run_id = next_stamp++;
            bool prev_hex = !L.empty() && L[0].g == M.hex_group;
            for (size_t k = 0; k < L.size(); ++k) {
                const auto &in = gs[L[k].g];
                const bool is_hex = L[k].g == M.hex_group || (M.hex_tile_type >= 0 && type_of[L[k].g] == M.hex_tile_type);   // (counted against the hex slots)
                const int t = is_hex ? -1 : type_of[L[k].g];
                auto stamp_new = [&](int32_t id, bool mark) {            // nodes of element k the run `id` does not hold yet
                    int fresh = 0;
                    for (int l = 0; l < in.nd / 3; ++l) {
                        const int64_t node = new_node(in.dof[(int64_t)(3 * l) * in.ne + L[k].e]);
                        if (stamp[node] != id) { if (mark) stamp[node] = id; ++fresh; }
                    }
                    return fresh;
                };
                const int fresh = stamp_new(run_id, false);
                const bool new_tile = !is_hex && cnt_type[t] % 16 == 0;
                const bool kind_change = mix_shells && is_hex != prev_hex;
                prev_hex = is_hex;
                if (k > lo_ && (kind_change || n_run_nodes + fresh > cap || (is_hex && n_hex + 1 > hex_cap) || (new_tile && n_tl + 1 > tile_cap))) {
                    out(lo_, k);
                    lo_ = k;
                    run_id = next_stamp++;
                    n_run_nodes = n_hex = n_tl = 0;
                    std::fill(cnt_type.begin(), cnt_type.end(), 0);
                }
                n_run_nodes += stamp_new(run_id, true);
                if (is_hex) ++n_hex;
                else { if (cnt_type[t] % 16 == 0) ++n_tl; ++cnt_type[t]; }
            }
            if (lo_ < L.size()) out(lo_, L.size());
        };
        // Meshes whose chunks do not fill the GPU once (round 5): a launch then lasts as long as ONE chunk's chain of phases - staging,
        // its tiles wave after wave, the write-out - however many CUs idle beside it (1 M-dof octree mesh: 661 chunks of ~25 tiles for
        // 1 024 resident workgroups).  Smaller chunks shorten the chain: the node cap is lowered until the mesh makes about
        // PCG_EBE_TARGET_CHUNKS chunks (default kMixedTargetChunks; 0 = off), never below kMixedMinNodeCap.  Large meshes keep 768.
        int64_t target_chunks = kMixedTargetChunks;
        if (const char *tv = std::getenv("PCG_EBE_TARGET_CHUNKS")) target_chunks = std::max(0, std::atoi(tv));
        if (!std::getenv("PCG_EBE_NODE_CAP") && target_chunks > 0 && M.hex_tile_type >= 0) {
            auto count = [&](int cap) { int64_t n = 0; cut_runs(cap, [&](size_t, size_t) { ++n; }); return n; };
            if (count(node_cap) < target_chunks) {
                int lo_cap = kMixedMinNodeCap / 16, hi_cap = node_cap / 16;          // smallest cap (x 16) whose chunk count stays <= target
                while (lo_cap < hi_cap) {
                    const int mid = (lo_cap + hi_cap) / 2;
                    if (count(16 * mid) <= target_chunks) hi_cap = mid; else lo_cap = mid + 1;
                }
                node_cap = 16 * hi_cap;
            }
        }
        C.node_cap_used = node_cap;
        cut_runs(node_cap, [&](size_t a, size_t b) { emit_or_split(a, b); });
        if (std::getenv("PCG_EBE_PLAN_STATS")) {             // development: one line per chunk - nodes, hex tiles, other tiles, their k-steps
            const size_t nc = C.hdr.size() / 8;
            for (size_t c = 0; c < nc; ++c) {
                const int32_t *h = &C.hdr[8 * c];
                if (h[6] != kMixedClass) continue;
                const int nht = M.hex_tile_type >= 0 ? M.chunk_hex_tiles[(size_t)h[4]] : 0;
                int other = 0, ksteps = 0;
                for (int t = nht; t < h[5]; ++t) { ++other; ksteps += 3 * M.types[(size_t)M.tile_type[(size_t)h[7] + t]].J; }
                std::fprintf(stderr, "[plan] chunk %zu nodes %d hex_elems %d hex_tiles %d other_tiles %d other_ksteps %d\n", c, h[1], h[3], nht, other, ksteps);
            }
        }
    }

    // Chunks = octree-like cells: the spatially sorted element list of a group is split recursively at
    // Morton-bit boundaries until a cell holds <= 256*ept elements, <= kChunkMaxNodes nodes and <= 63
    // sub-colours.  On a uniform region the cells form a regular lattice of full boxes.  Without coordinates
    // (keys = node ids) cells are plain runs of elements.
    if (!mixed_mode) {
        int32_t next_stamp = 0;
        std::vector<std::vector<ElemRef>> per_group(n_groups);
        for (const auto &r : order)
            if (chunkable[r.g]) per_group[r.g].push_back(r);
        for (int g = 0; g < n_groups; ++g) {
            if (!chunkable[g]) continue;
            const auto &in = gs[g];
            const auto &L = per_group[g];
            const int nno = in.nd / 3;
            const size_t chunk_elems = (size_t)C.cls[cls_of[g]].ce;
            Open o;
            std::vector<std::pair<uint64_t, int32_t>> keyed;
            std::vector<int> sc;
            std::vector<uint16_t> lids;
            // build the candidate chunk [lo_, hi_); emit it if it respects every limit (elements, nodes, sub-colours)
            auto try_emit = [&](size_t lo_, size_t hi_) {
                if (hi_ - lo_ > chunk_elems) return false;
                const int32_t id = next_stamp++;
                o.elems.clear(); o.nodes.clear();
                keyed.clear();
                for (size_t k = lo_; k < hi_; ++k) {
                    for (int l = 0; l < nno; ++l) {
                        const int64_t orig = in.dof[(int64_t)(3 * l) * in.ne + L[k].e] / 3;
                        const int64_t node = perm ? perm[orig] : orig;
                        if (stamp[node] != id) { stamp[node] = id; keyed.emplace_back(tile_key(orig, node), (int32_t)node); }
                    }
                    o.elems.push_back(L[k].e);
                }
                if ((int)keyed.size() > C.cls[cls_of[g]].max_nodes) return false;
                std::sort(keyed.begin(), keyed.end());
                for (size_t k = 0; k < keyed.size(); ++k) {
                    o.nodes.push_back(keyed[k].second);
                    local_of[keyed[k].second] = (int32_t)k;
                }
                int nsub = 0;
                bool bnd = false;
                if (!sub_colour(g, o, sc, lids, nsub, bnd)) return false;
                if (C.cls[cls_of[g]].full) {                 // hex8 class: sub-colour groups start on 16-slot tiles
                    std::vector<int> cnt(nsub, 0);
                    for (int c : sc) cnt[c]++;
                    size_t padded = 0;
                    for (int c : cnt) padded += (size_t)(c + 15) / 16 * 16;
                    if (padded > chunk_elems) return false;
                }
                close_chunk(g, o, sc, lids, nsub, bnd);
                return true;
            };
            // Pattern types with hanging nodes (64-element chunks, one element per lane of k_ebe_rows) are sparse: on a graded
            // octree mesh a type's elements lie scattered along the transition shells, and Morton cells of <= 64 elements are
            // mostly far from full (round 3: 2 222 chunks for 265 k elements at 1 M dof).  They are packed GREEDILY instead:
            // runs of up to 64 consecutive elements of the type's Morton order, shortened only where the tile would exceed its
            // node or sub-colour limits.  The hex8 class keeps the cell recursion (full boxes = regular LDS tiles).
            // Hanging-node classes without a node tile (k_ebe_direct): a chunk is a run of 64 consecutive elements of the type's
            // Morton order, full except for the last one; an entry of `nodes` / `dst` per element-node incidence.
            if (C.cls[cls_of[g]].direct) {
                auto &K = C.cls[cls_of[g]];
                const int CE = K.ce, W = K.words;
                for (size_t lo_ = 0; lo_ < L.size(); lo_ += (size_t)CE) {
                    const int ne = (int)std::min<size_t>((size_t)CE, L.size() - lo_);
                    const int32_t cid = (int32_t)C.n_chunks++;
                    const int32_t kci = (int32_t)K.n_chunks++;
                    const size_t off = C.nodes.size(), nn = (size_t)nno * CE;
                    C.nodes.resize(off + nn, -1);
                    C.tslot.resize(off + nn, 0);
                    C.direct_entries += (int64_t)nn;
                    K.ck.resize((size_t)(kci + 1) * CE, 0.0);
                    K.sgn.resize((size_t)(kci + 1) * W * CE, 0u);
                    for (int lane = ne; lane < CE; ++lane) K.sgn[((size_t)kci * W + (W - 1)) * CE + lane] = 0xff000000u;   // padding marker
                    bool bnd = false;
                    for (int t = 0; t < ne; ++t) {
                        const int64_t e = L[lo_ + t].e;
                        for (int l = 0; l < nno; ++l) {
                            const int64_t node = new_node(in.dof[(int64_t)(3 * l) * in.ne + e]);
                            C.nodes[off + (size_t)l * CE + t] = (int32_t)node;
                            bnd |= node < n_boundary_nodes;
                        }
                        K.ck[(size_t)kci * CE + t] = in.ck[e];
                        for (int w = 0; w < W; ++w) {
                            uint32_t bits = 0;
                            for (int a = 32 * w; a < std::min(in.nd, 32 * w + 32); ++a)
                                if (in.sign[(int64_t)a * in.ne + e]) bits |= (1u << (a - 32 * w));
                            K.sgn[((size_t)kci * W + w) * CE + t] = bits;
                        }
                    }
                    C.hdr.insert(C.hdr.end(), {(int32_t)off, (int32_t)nn, 0, ke_index[g], kci, in.nd, cls_of[g], (ne + 15) / 16});
                    chunk_phase.push_back(bnd ? 0 : 1);
                    K.list[bnd ? 0 : 1].push_back(cid);
                }
                continue;
            }
            const char *gev = std::getenv("PCG_EBE_GREEDY_CHUNKS");                      // =0: Morton cells for every class (A/B)
            if (!C.cls[cls_of[g]].full && coords && !(gev && gev[0] == '0')) {
                size_t lo_ = 0;
                while (lo_ < L.size()) {
                    size_t len = std::min(chunk_elems, L.size() - lo_);
                    while (len > 1 && !try_emit(lo_, lo_ + len)) len = len > 8 ? len * 3 / 4 : len - 1;
                    if (len == 1 && !try_emit(lo_, lo_ + 1)) throw std::runtime_error("ebe: an element does not fit a chunk of its class");
                    lo_ += len;
                }
                continue;
            }
            // explicit stack: (lo, hi, bit)
            struct Cell { size_t lo, hi; int bit; };
            std::vector<Cell> stack;
            if (!L.empty()) stack.push_back(Cell{0, L.size(), coords ? 62 : -1});
            while (!stack.empty()) {
                Cell cdesc = stack.back();
                stack.pop_back();
                const size_t n_el = cdesc.hi - cdesc.lo;
                if (try_emit(cdesc.lo, cdesc.hi)) continue;
                int bit = cdesc.bit;
                size_t mid = cdesc.lo;
                while (bit >= 0) {                          // first Morton bit that actually splits the cell
                    const uint64_t m = 1ull << bit;
                    mid = (size_t)(std::partition_point(L.begin() + cdesc.lo, L.begin() + cdesc.hi,
                                                        [&](const ElemRef &r) { return (r.key & m) == 0; }) - L.begin());
                    if (mid > cdesc.lo && mid < cdesc.hi) break;
                    --bit;
                }
                if (bit < 0) mid = cdesc.lo + std::min<size_t>(n_el / 2, chunk_elems);             // no spatial key left: cut the run
                if (mid == cdesc.lo) mid = cdesc.lo + 1;
                stack.push_back(Cell{mid, cdesc.hi, bit - 1});                                   // right half later
                stack.push_back(Cell{cdesc.lo, mid, bit - 1});                                   // left half first (keeps order)
            }
        }
    }
    // ---- exclusive / shared tile nodes, boundary slots, per-phase fix-up lists ----------------------------
    {
        std::vector<int32_t> cnt((size_t)n_nodes, 0);
        for (int32_t nd : C.nodes)
            if (nd >= 0) cnt[nd]++;                                    // (-1: padding slot of a direct chunk)
        for (int64_t i = 0; i < n_nodes; ++i)
            if (cnt[i] == 0) { C.needs_zero = true; break; }
        // phase in which a shared node becomes final = max phase of the chunks that contain it
        std::vector<uint8_t> final_phase((size_t)n_nodes, 0);
        for (int64_t c = 0; c < C.n_chunks; ++c) {
            const int32_t off = C.hdr[(size_t)c * 8], nn = C.hdr[(size_t)c * 8 + 1];
            for (int k = 0; k < nn; ++k)
                if (C.nodes[off + k] >= 0) final_phase[C.nodes[off + k]] = std::max(final_phase[C.nodes[off + k]], chunk_phase[c]);
        }
        std::vector<int32_t> sh_index((size_t)n_nodes, -1);           // index into sh_node of its final phase
        for (int64_t i = 0; i < n_nodes; ++i)
            if (cnt[i] > 1) {
                const int ph = final_phase[i];
                sh_index[i] = (int32_t)C.sh_node[ph].size();
                C.sh_node[ph].push_back((int32_t)i);
            }
        for (int ph = 0; ph < 2; ++ph) {
            C.sh_ptr[ph].assign(C.sh_node[ph].size() + 1, 0);
            for (size_t k = 0; k < C.sh_node[ph].size(); ++k) C.sh_ptr[ph][k + 1] = C.sh_ptr[ph][k] + cnt[C.sh_node[ph][k]];
            C.sh_slot[ph].assign((size_t)C.sh_ptr[ph].back(), 0);
        }
        // Boundary slots are numbered NODE-major: the slots of one shared node are consecutive (ascending chunk id =
        // the summation order), the nodes of a phase follow each other in ascending node id.  The shared-node kernel
        // then streams the buffer front to back (consecutive threads read consecutive runs) instead of chasing one
        // slot index per addend; a chunk's stores scatter a little more, but stores are fire-and-forget.
        const int32_t phase_base[2] = {0, C.sh_ptr[0].back()};
        C.n_slots = (int64_t)C.sh_ptr[0].back() + C.sh_ptr[1].back();
        std::vector<int32_t> fill[2];
        for (int ph = 0; ph < 2; ++ph) fill[ph].assign(C.sh_ptr[ph].begin(), C.sh_ptr[ph].end() - 1);
        for (int ph = 0; ph < 2; ++ph)
            for (size_t q = 0; q < C.sh_slot[ph].size(); ++q) C.sh_slot[ph][q] = phase_base[ph] + (int32_t)q;
        C.dst.resize(C.nodes.size());
        for (int64_t c = 0; c < C.n_chunks; ++c) {                     // ascending chunk id = summation order
            const int32_t off = C.hdr[(size_t)c * 8], nn = C.hdr[(size_t)c * 8 + 1];
            for (int k = 0; k < nn; ++k) {
                const int32_t nd = C.nodes[off + k];
                if (nd < 0) { C.dst[off + k] = INT32_MIN; continue; }
                if (cnt[nd] == 1) { C.dst[off + k] = 3 * nd; continue; }
                const int ph = final_phase[nd];
                const int32_t slot = phase_base[ph] + fill[ph][sh_index[nd]]++;
                C.dst[off + k] = -(slot + 1);
            }
        }
        for (int ph = 0; ph < 2; ++ph) {
            int launches = 0;
            for (const auto &K : C.cls) launches += K.list[ph].empty() ? 0 : 1;
            out.n_colors[ph] = std::max<int32_t>(out.n_colors[ph], launches);
        }
        if (!out.ranges[0].empty() || !out.ranges[1].empty()) C.needs_zero = true;   // other groups accumulate with +=
        if (const char *sv = std::getenv("PCG_EBE_STATS"); sv && sv[0] == '1') {      // planner statistics (development aid)
            std::vector<int64_t> tile_nodes(kChunkClasses, 0), shared_tile_nodes(kChunkClasses, 0);
            for (int64_t c = 0; c < C.n_chunks; ++c) {
                const int32_t off = C.hdr[(size_t)c * 8], nn = C.hdr[(size_t)c * 8 + 1], cl = C.hdr[(size_t)c * 8 + 6];
                for (int k = 0; k < nn; ++k) {                      // (direct chunks: element-node incidences, padding slots skipped)
                    tile_nodes[cl] += C.nodes[off + k] >= 0;
                    shared_tile_nodes[cl] += C.dst[off + k] < 0 && C.dst[off + k] != INT32_MIN;
                }
            }
            std::vector<int64_t> elems(kChunkClasses, 0);
            for (int g = 0; g < n_groups; ++g)
                if (chunkable[g]) elems[cls_of[g]] += gs[g].ne;
            for (int c = 0; c < kChunkClasses; ++c)
                if (C.cls[c].n_chunks)
                    fprintf(stderr, "ebe plan: class %d (<= %d nodes, %d slots/chunk%s): %lld elements in %lld chunks, %lld %s (%lld to boundary slots)\n",
                            c, C.cls[c].nnp, C.cls[c].ce, C.cls[c].direct ? ", no node tile" : "", (long long)elems[c], (long long)C.cls[c].n_chunks, (long long)tile_nodes[c],
                            C.cls[c].direct ? "element-node incidences" : "tile nodes", (long long)shared_tile_nodes[c]);
            if (C.mixed.n_tiles || C.mixed.hex_elems)
                fprintf(stderr, "ebe plan: mixed chunks: %lld elements in the hex section, %lld in %lld tiles of 16 (%.0f %% full), %zu tile types, "
                                "at most %d M-tiles\n", (long long)C.mixed.hex_elems, (long long)C.mixed.tile_elems, (long long)C.mixed.n_tiles,
                        C.mixed.n_tiles ? 100.0 * C.mixed.tile_elems / (16.0 * C.mixed.n_tiles) : 0.0, C.mixed.types.size(), C.mixed.max_mt);
            if (C.mixed.hex_tile_type >= 0) {
                int64_t ht = 0;
                for (int32_t v : C.mixed.chunk_hex_tiles) ht += v;
                fprintf(stderr, "ebe plan:   the 8-node type runs in %lld colour-pure hex tiles (k_ebe_mtile), no hex section\n", (long long)ht);
            }
            if (C.mixed.n_tiles) {                                  // per tile type: node quartets, tiles, elements; fill histogram
                std::vector<int64_t> tl(C.mixed.types.size(), 0), el(C.mixed.types.size(), 0), hist(17, 0);
                for (int64_t t = 0; t < C.mixed.n_tiles; ++t) {
                    int cnt = 0;
                    for (int e = 0; e < 16; ++e) cnt += C.mixed.tcol[(size_t)t * 16 + e] != 255;
                    tl[C.mixed.tile_type[t]]++; el[C.mixed.tile_type[t]] += cnt; hist[cnt]++;
                }
                for (size_t t = 0; t < tl.size(); ++t)
                    fprintf(stderr, "ebe plan:   tile type %zu: %d nodes (J = %d), %lld tiles, %lld elements\n", t, C.mixed.types[t].nn, C.mixed.types[t].J,
                            (long long)tl[t], (long long)el[t]);
                fprintf(stderr, "ebe plan:   tiles by elements held (1..16):");
                for (int c = 1; c <= 16; ++c) fprintf(stderr, " %lld", (long long)hist[c]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "ebe plan: %lld nodes, %lld shared nodes with %lld boundary slots; %lld chunks (node cap of the mixed chunks %d)\n", (long long)n_nodes,
                    (long long)(C.sh_node[0].size() + C.sh_node[1].size()), (long long)C.n_slots, (long long)C.n_chunks, C.node_cap_used);
        }
    }
}

}  // namespace pcg
