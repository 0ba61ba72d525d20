// TEST DOUBLE - plain-loop implementation of pcg::Backend.
//
// Compiled ONLY into tests/hostops/_build/libpcg_hostops.so together with the product's
// back-end-agnostic sources (pcg_driver.cpp, assemble.cpp, sell.cpp).  It lets the CPU test-suite
// (`pytest -m "not gpu"`, including the world_size-2 gloo tests) exercise the PCG control flow, the
// SELL conversion, the interface fix-up lists and the Python comm hooks without a GPU.  It is never
// part of libpcg_mi355x.so and the pcg_mi355x package never loads it on its own; the product
// library has exactly one back end (HIP, gfx950) and no CPU path.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "host_mail.hpp"
#include "pcg_internal.hpp"

namespace pcg {

class HostBackend : public Backend {
    SellHost m_;
    EbeHost ebe_;
    std::vector<double> ebuf_;
    std::vector<uint8_t> flags_;
    HaloHost h_;
    double dot_spmv_ = 0, dot_fix_ = 0, dotw_ = 0;
    double up_[5] = {0, 0, 0, 0, 0}, res_[3] = {0, 0, 0};
    int64_t n_ = 0;
    bool own_free(int64_t d) const { return (flags_[d] & 3) == 3; }
    bool is_free(int64_t d) const { return (flags_[d] & 2) != 0; }

public:
    const char *name() const override { return "hostops-test"; }
    void *stream() override { return nullptr; }
    void *alloc(size_t b) override { void *p = std::malloc(b ? b : 8); if (!p) throw std::bad_alloc(); return p; }
    void release(void *p) override { std::free(p); }
    void h2d(void *d, const void *s, size_t b) override { std::memcpy(d, s, b); }
    void d2h(void *d, const void *s, size_t b) override { std::memcpy(d, s, b); }
    void d2d(void *d, const void *s, size_t b) override { std::memcpy(d, s, b); }
    void zero(void *d, size_t b) override { std::memset(d, 0, b); }
    void sync() override {}
    void upload_matrix(const SellHost &m) override { m_ = m; n_ = m.bs * m.n_nodes; }
    void upload_ebe(const EbeHost &m) override
    {
        ebe_ = m; n_ = 3 * m.n_nodes;
        m_ = SellHost(); m_.n_nodes = m.n_nodes; m_.diag = m.diag;
    }
    void upload_scalar_copy(Backend &src_be, const std::vector<int64_t> &ptr1, const std::vector<int64_t> &, int64_t n_rows) override
    {
        const SellHost &B = static_cast<HostBackend &>(src_be).m_;
        if (B.bs != 3 || B.C != 64 || !B.bidx.empty() || B.ov_slices) throw std::runtime_error("scalar copy: plain, unsplit 3x3-block format only");
        m_ = SellHost();
        m_.bs = 1; m_.C = 64; m_.n_nodes = n_rows; m_.n_slices = (int64_t)ptr1.size() - 1; m_.slice_ptr = ptr1; m_.diag = B.diag;
        m_.cols.assign((size_t)ptr1.back() * 64, 0); m_.vals.assign((size_t)ptr1.back() * 64, 0.0);
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t node = r / 3, sb = node / 64, s1 = r / 64;
            const int a = (int)(r % 3), lb = (int)(node % 64), l1 = (int)(r % 64);
            const int64_t bbase = B.slice_ptr[sb], bw = B.slice_ptr[sb + 1] - bbase, base1 = ptr1[s1], w1 = ptr1[s1 + 1] - base1;
            for (int64_t k = 0; k < w1; ++k) {
                const size_t q = (size_t)(base1 + k) * 64 + l1;
                if (k < 3 * bw) {
                    m_.vals[q] = B.vals[((size_t)(bbase + k / 3) * 9 + 3 * a + k % 3) * 64 + lb];
                    m_.cols[q] = 3 * B.cols[(size_t)(bbase + k / 3) * 64 + lb] + (int32_t)(k % 3);
                } else {
                    m_.cols[q] = (int32_t)r;
                }
            }
        }
        n_ = n_rows;
    }
    bool ebe_apply(const double *x, double *y, int plo, int phi, bool zero_first, bool with_dot, int64_t dot_lo) override
    {
        const bool fuse = with_dot && ebe_.ranges[0].empty() && ebe_.ranges[1].empty() && ebe_.chunked.n_chunks > 0;
        const auto &C = ebe_.chunked;
        if (zero_first && (C.n_chunks == 0 || C.needs_zero)) std::memset(y, 0, sizeof(double) * n_);
        std::vector<double> u;
        std::vector<double> xs(3 * kChunkMaxNodes), ys(3 * kChunkMaxNodes), acc;
        if (ebuf_.size() < (size_t)C.n_slots * 3) ebuf_.assign((size_t)C.n_slots * 3, 0.0);
        for (int ph = plo; ph < phi; ++ph) {
            for (const auto &K : C.cls)                               // one launch per node-count class
            for (int32_t cid : K.list[ph]) {
                const int32_t *h = &C.hdr[(size_t)cid * 8];
                const int32_t off = h[0], nn = h[1], nsub = h[2], kci = h[4], nd = h[5];
                const int CE = K.ce, W = K.words, ndp = 3 * K.nnp;
                const double *Kc = &K.ke_col[(size_t)h[3] * ndp * ndp];
                if (&K == &C.cls[kMixedClass]) {                        // mixed-type chunk: hex section, then 16-element tiles
                    const auto &M = C.mixed;
                    const int nh = h[3], ntl = h[5], tile0 = h[7];
                    for (int n = 0; n < nn; ++n)
                        for (int d = 0; d < 3; ++d) { xs[3 * C.tslot[off + n] + d] = x[3 * (int64_t)C.nodes[off + n] + d]; ys[3 * n + d] = 0.0; }
                    double a24[24];
                    for (int slot = 0; slot < nh; ++slot) {              // slot order = (pass, wave, sub-colour) order of the kernel
                        const unsigned sg = K.sgn[(size_t)kci * CE + slot];
                        const double c = K.ck[(size_t)kci * CE + slot];
                        for (int k = 0; k < 24; ++k) a24[k] = 0.0;
                        for (int b = 0; b < 24; ++b) {
                            double v = xs[3 * K.lid[((size_t)kci * 8 + b / 3) * CE + slot] + b % 3];
                            if ((sg >> b) & 1u) v = -v;
                            v = c * v;
                            for (int k = 0; k < 24; ++k) a24[k] += K.ke_col[(size_t)b * 24 + k] * v;
                        }
                        for (int k = 0; k < 24; ++k)
                            ys[3 * K.lid[((size_t)kci * 8 + k / 3) * CE + slot] + k % 3] += ((sg >> k) & 1u) ? -a24[k] : a24[k];
                    }
                    const int W = M.words, NP = M.nnpt;
                    std::vector<double> uu, out;
                    for (int ti = tile0; ti < tile0 + ntl; ++ti) {
                        const auto &T = M.types[M.tile_type[ti]];
                        const int MT = (3 * T.J + 3) / 4, KS = 3 * T.J;
                        const double *F = M.frag.data() + T.frag_off;
                        out.assign((size_t)16 * T.nd, 0.0);
                        auto sb = [&](int e, int a) { return (M.tsgn[((size_t)ti * W + a / 32) * 16 + e] >> (a % 32)) & 1u; };
                        auto pc = [&](int e, int c) { return (M.tperm[(size_t)ti * 16 + e] >> (2 * c)) & 3; };   // the element's own dof order
                        for (int e = 0; e < 16; ++e) {
                            if (M.tcol[(size_t)ti * 16 + e] == 255) continue;
                            uu.assign(T.nd, 0.0);
                            for (int b = 0; b < T.nd; ++b) {
                                double v = xs[3 * M.tlid[((size_t)ti * NP + b / 3) * 16 + e] + pc(e, b % 3)];
                                uu[b] = M.tck[(size_t)ti * 16 + e] * (sb(e, b) ? -v : v);
                            }
                            for (int g = 0; g < 4; ++g)                  // the data flow of the matrix-core tile (fragments as uploaded)
                                for (int q = 0; q < 3 * T.J; ++q) {
                                    const int row_node = 4 * (q / 3) + g;
                                    if (row_node >= T.nn) continue;
                                    const int mt = q / 4, i = g + 4 * (q % 4);
                                    double a = 0.0;
                                    for (int ks = 0; ks < KS; ++ks)
                                        for (int gk = 0; gk < 4; ++gk) {
                                            const int col_node = 4 * (ks / 3) + gk;
                                            if (col_node < T.nn) a += F[((size_t)ks * MT + mt) * 64 + gk * 16 + i] * uu[3 * col_node + ks % 3];
                                        }
                                    const int dof = 3 * row_node + q % 3;
                                    out[(size_t)e * T.nd + dof] = sb(e, dof) ? -a : a;
                                }
                        }
                        for (int col = 0; col < M.tile_ncol[ti]; ++col)
                            for (int e = 0; e < 16; ++e)
                                if (M.tcol[(size_t)ti * 16 + e] == col)
                                    for (int k = 0; k < T.nd; ++k) ys[3 * M.tlid[((size_t)ti * NP + k / 3) * 16 + e] + pc(e, k % 3)] += out[(size_t)e * T.nd + k];
                    }
                    for (int n = 0; n < nn; ++n) {
                        const int32_t dst = C.dst[off + n];
                        double *o = dst >= 0 ? y + dst : &ebuf_[(size_t)(-dst - 1) * 3];
                        const int sl = C.tslot[off + n];
                        for (int d = 0; d < 3; ++d) {
                            o[d] = ys[3 * sl + d];
                            if (fuse && dst >= 0 && dst + d >= dot_lo && own_free(dst + d)) dot_spmv_ += xs[3 * sl + d] * ys[3 * sl + d];
                        }
                    }
                    continue;
                }
                acc.assign((size_t)nd * CE, 0.0);
                if (K.direct) {                                        // no node tile: one entry of nodes / dst per element-node incidence
                    for (int lane = 0; lane < CE; ++lane) {
                        if (C.nodes[off + lane] < 0) continue;          // padding slot
                        const double c = K.ck[(size_t)kci * CE + lane];
                        auto sb = [&](int a) { return (K.sgn[((size_t)kci * W + a / 32) * CE + lane] >> (a % 32)) & 1u; };
                        double *a = &acc[(size_t)lane * nd];
                        for (int b = 0; b < nd; ++b) {
                            double v = x[3 * (int64_t)C.nodes[off + (size_t)(b / 3) * CE + lane] + b % 3];
                            if (sb(b)) v = -v;
                            v = c * v;
                            for (int k = 0; k < nd; ++k) a[k] += Kc[(size_t)b * ndp + k] * v;
                        }
                        for (int k = 0; k < nd; ++k) {
                            const int32_t dst = C.dst[off + (size_t)(k / 3) * CE + lane];
                            const double o = sb(k) ? -a[k] : a[k];
                            if (dst >= 0) {
                                y[dst + k % 3] = o;
                                if (fuse && dst + k % 3 >= dot_lo && own_free(dst + k % 3)) dot_spmv_ += x[dst + k % 3] * o;
                            } else {
                                ebuf_[(size_t)(-(int64_t)dst - 1) * 3 + k % 3] = o;
                            }
                        }
                    }
                    continue;
                }
                for (int n = 0; n < nn; ++n)
                    for (int d = 0; d < 3; ++d) { xs[3 * C.tslot[off + n] + d] = x[3 * (int64_t)C.nodes[off + n] + d]; ys[3 * n + d] = 0.0; }
                auto sbit = [&](int lane, int a) { return (K.sgn[((size_t)kci * W + a / 32) * CE + lane] >> (a % 32)) & 1u; };
                auto subc = [&](int lane) { return (int)(K.sgn[((size_t)kci * W + W - 1) * CE + lane] >> 24); };
                for (int lane = 0; lane < CE; ++lane) {
                    if (subc(lane) == 255) continue;
                    const double c = K.ck[(size_t)kci * CE + lane];
                    double *a = &acc[(size_t)lane * nd];
                    for (int b = 0; b < nd; ++b) {
                        double v = xs[3 * K.lid[((size_t)kci * K.nnp + b / 3) * CE + lane] + b % 3];
                        if (sbit(lane, b)) v = -v;
                        v = c * v;
                        for (int k = 0; k < nd; ++k) a[k] += Kc[(size_t)b * ndp + k] * v;
                    }
                }
                for (int s = 0; s < nsub; ++s)
                    for (int lane = 0; lane < CE; ++lane) {
                        if (subc(lane) != s) continue;
                        for (int k = 0; k < nd; ++k) {
                            double o = acc[(size_t)lane * nd + k];
                            if (sbit(lane, k)) o = -o;
                            ys[3 * K.lid[((size_t)kci * K.nnp + k / 3) * CE + lane] + k % 3] += o;
                        }
                    }
                for (int n = 0; n < nn; ++n) {
                    const int32_t dst = C.dst[off + n];
                    double *out = dst >= 0 ? y + dst : &ebuf_[(size_t)(-dst - 1) * 3];
                    const int sl = C.tslot[off + n];
                    for (int d = 0; d < 3; ++d) {
                        out[d] = ys[3 * sl + d];
                        if (fuse && dst >= 0 && dst + d >= dot_lo && own_free(dst + d)) dot_spmv_ += xs[3 * sl + d] * ys[3 * sl + d];
                    }
                }
            }
            for (size_t k = 0; k < C.sh_node[ph].size(); ++k)          // shared nodes: slots in chunk order
                for (int d = 0; d < 3; ++d) {
                    double sum = 0.0;
                    for (int32_t q = C.sh_ptr[ph][k]; q < C.sh_ptr[ph][k + 1]; ++q) sum += ebuf_[(size_t)C.sh_slot[ph][q] * 3 + d];
                    const int64_t dd = 3 * (int64_t)C.sh_node[ph][k] + d;
                    y[dd] = sum;
                    if (fuse && dd >= dot_lo && own_free(dd)) dot_spmv_ += x[dd] * sum;
                }
        }
        for (int ph = plo; ph < phi; ++ph)
            for (const auto &r : ebe_.ranges[ph]) {
                const auto &G = ebe_.groups[r.group];
                u.resize(G.nd);
                for (int64_t e = r.lo; e < r.hi; ++e) {
                    for (int b = 0; b < G.nd; ++b) {
                        double v = x[G.dof[(size_t)b * G.ne + e]];
                        if (G.sign[(size_t)b * G.ne + e]) v = -v;
                        u[b] = G.ck[e] * v;
                    }
                    for (int a = 0; a < G.nd; ++a) {
                        double acc = 0;
                        for (int b = 0; b < G.nd; ++b) acc += G.ke[(size_t)a * G.nd + b] * u[b];
                        if (G.sign[(size_t)a * G.ne + e]) acc = -acc;
                        y[G.dof[(size_t)a * G.ne + e]] += acc;
                    }
                }
            }
        return fuse;
    }
    bool ebe_can_split() const override { return std::getenv("PCG_TEST_FORCE_SPLIT") || (ebe_.ranges[0].empty() && ebe_.ranges[1].empty()); }
    void upload_masks(const uint8_t *f, int64_t n) override { flags_.assign(f, f + n); }
    void upload_halo(const HaloHost &h) override { h_ = h; }

    void spmv(const double *x, double *y, int64_t lo, int64_t hi, bool with_dot, double *pack_send) override
    {
        struct PackAtExit { HostBackend *b; const double *y; double *s; ~PackAtExit() { if (s) b->halo_pack(y, s); } } pack_at_exit{this, y, pack_send};
        const int C = m_.C;
        double acc = 0;
        if (m_.bs == 1) {                                               // scalar rows (pcg_create_csr block = 1)
            for (int64_t s = lo; s < hi; ++s) {
                const int64_t base = m_.slice_ptr[s], w = m_.slice_ptr[s + 1] - base;
                for (int l = 0; l < C; ++l) {
                    const int64_t r = s * C + l;
                    if (r >= m_.n_nodes) break;
                    double t = 0;
                    for (int64_t k = 0; k < w; ++k) t += m_.vals[(size_t)(base + k) * C + l] * x[m_.cols[(size_t)(base + k) * C + l]];
                    y[r] = t;
                    if (with_dot && own_free(r)) acc += x[r] * t;
                }
            }
            if (with_dot) dot_spmv_ = acc;
            return;
        }
        for (int64_t s = lo; s < hi; ++s) {
            const int64_t base = m_.slice_ptr[s], w = m_.slice_ptr[s + 1] - base;
            for (int l = 0; l < C; ++l) {
                const int64_t r = s * C + l;
                if (r >= m_.n_nodes) break;
                double t[3] = {0, 0, 0};
                for (int64_t k = 0; k < w; ++k) {
                    const int64_t j = m_.cols[(size_t)(base + k) * C + l];
                    // dictionary format (compress_blocks): the block's 9 values are dict[9 * bidx[q]..], q = the stored slot
                    const double *blk = m_.bidx.empty() ? nullptr : &m_.dict[(size_t)9 * m_.bidx[(size_t)(base + k) * C + l]];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b)
                            t[a] += (blk ? blk[a * 3 + b] : m_.vals[((size_t)(base + k) * 9 + a * 3 + b) * C + l]) * x[3 * j + b];
                }
                const bool final_here = m_.ov_slices == 0 || ((m_.ov_mask[s] >> l) & 1ull) == 0;
                for (int a = 0; a < 3; ++a) {
                    y[3 * r + a] = t[a];
                    if (with_dot && final_here && own_free(3 * r + a)) acc += x[3 * r + a] * t[a];
                }
            }
        }
        if (m_.ov_slices > 0) {                                         // split matrix: the longer rows continue (split_overflow)
            int64_t olo, ohi;
            if (lo == 0) olo = 0; else if (lo == m_.n_bnd_slices) olo = m_.ov_bnd_slices; else throw std::runtime_error("spmv: slice range does not match the overflow part");
            if (hi == m_.n_slices) ohi = m_.ov_slices; else if (hi == m_.n_bnd_slices) ohi = m_.ov_bnd_slices; else throw std::runtime_error("spmv: slice range does not match the overflow part");
            for (int64_t s = olo; s < ohi; ++s) {
                const int64_t base = m_.ov_slice_ptr[s], w = m_.ov_slice_ptr[s + 1] - base;
                for (int l = 0; l < C; ++l) {
                    const int64_t r = m_.ov_rows[(size_t)s * C + l];
                    if (r < 0) continue;
                    double *t = y + 3 * r;
                    for (int64_t k = 0; k < w; ++k) {
                        const int64_t j = m_.ov_cols[(size_t)(base + k) * C + l];
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b) t[a] += m_.ov_vals[((size_t)(base + k) * 9 + a * 3 + b) * C + l] * x[3 * j + b];
                    }
                    for (int a = 0; a < 3; ++a)
                        if (with_dot && own_free(3 * r + a)) acc += x[3 * r + a] * t[a];
                }
            }
        }
        if (with_dot) dot_spmv_ = acc;
    }
    void halo_pack(const double *y, double *send) override
    {
        for (size_t m = 0; m < h_.send_idx.size(); ++m) send[m] = y[h_.send_idx[m]];
    }
    bool mailbox_kernels_available() const override { return true; }
    // direct exchange (csrc/kernels_vector.hpp k_halo_put / wait_for_neighbours) on host memory: every part runs on its own host thread
    bool direct_kernels_available() const override { return true; }
    void halo_put(const double *y, const DirectDesc &d) override
    {
        for (int j = 0; j < d.n_peers; ++j)
            for (long long m = d.seg[j]; m < d.seg[j + 1]; ++m) d.peer_recv[j][m - d.seg[j]] = y[h_.send_idx[(size_t)m]];
        for (int j = 0; j < d.n_peers; ++j) __atomic_store_n(d.peer_flag[j], d.seq, __ATOMIC_RELEASE);
    }
    void boundary_fixup(double *y, const double *recv, const double *xdot, bool with_dot, double *reduce_pq, const MailDesc *mail,
                        const DirectDesc *direct) override
    {
        if (direct)
            for (int j = 0; j < direct->n_peers; ++j) {
                const auto t0 = std::chrono::steady_clock::now();
                while (__atomic_load_n(direct->my_flags + j, __ATOMIC_ACQUIRE) < direct->seq) {
                    std::this_thread::yield();
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { __atomic_store_n(direct->err, 1u, __ATOMIC_RELAXED); break; }
                }
            }
        if (mail && (h_.fix_dof.empty() || !with_dot || !reduce_pq)) throw std::runtime_error("boundary_fixup: mailbox all-reduce without the fused reduction");
        struct ReduceAtExit { HostBackend *b; double *r; const MailDesc *m; ~ReduceAtExit() { if (r) { b->reduce_dot(r); if (m) host_mail_allreduce(*m, r, 1); } } }
            reduce_at_exit{this, with_dot ? reduce_pq : nullptr, mail};
        for (size_t k = 0; k < h_.fix_dof.size(); ++k) {
            double v = y[h_.fix_dof[k]];
            for (int64_t q = h_.fix_ptr[k]; q < h_.fix_ptr[k + 1]; ++q) v += recv[h_.fix_pos[q]];
            y[h_.fix_dof[k]] = v;
        }
        if (with_dot) {
            double acc = 0;
            const int64_t nb = !ebe_.groups.empty() ? (h_.fix_dof.empty() ? 0 : (int64_t)h_.fix_dof.back() + 1)
                                                     : std::min<int64_t>(n_, (int64_t)m_.n_bnd_slices * m_.C * 3);
            for (int64_t d = 0; d < nb; ++d)
                if (own_free(d)) acc += xdot[d] * y[d];
            dot_fix_ = acc;
        }
    }
    void begin_dot() override { dot_spmv_ = dot_fix_ = 0; }
    // Fault injection for the look-ahead tests: the PCG_TEST_NEG_PQ_AT-th p.Ap reduction of this engine (1-based count of
    // enqueued iterations) reports -1, as a speculative iteration formed from a stale residual may on real data.
    int64_t n_pq_ = 0;
    const int64_t neg_pq_at_ = std::getenv("PCG_TEST_NEG_PQ_AT") ? std::atoll(std::getenv("PCG_TEST_NEG_PQ_AT")) : -1;
    void reduce_dot(double *red) override { red[0] = (++n_pq_ == neg_pq_at_) ? -1.0 : dot_spmv_ + dot_fix_; }
    void scalar_alpha(double *st)
    {
        const double pq = st[ST_PQ], rho = st[ST_RHO_NEXT];
        st[ST_RHO] = rho;
        if (pq <= 0 || std::isinf(pq)) { st[ST_STOP] = 1; return; }     // STOP is sticky (never cleared here)
        st[ST_ALPHA] = rho / pq;
        if (std::isinf(st[ST_ALPHA])) st[ST_STOP] = 1;
    }
    // status ring: every "kernel" here runs at once, so a look-ahead iteration has ALREADY overwritten the block
    // when the host asks for its predecessor's sums - the most adversarial order a GPU could produce
    void set_status_block(double *st) override { st_ = st; }
    bool read_status(double *) override { return false; }
    void set_status_slot(int slot) override { slot_ = slot; }
    void publish_status(bool) override { std::memcpy(ring_[slot_], st_, sizeof(double) * ST_COUNT); }
    void wait_status(int slot, double *out) override { std::memcpy(out, ring_[slot], sizeof(double) * ST_COUNT); }
    double *st_ = nullptr;
    int slot_ = 0;
    double ring_[kStatusSlots][ST_COUNT] = {};
    bool iteration_fusion_available() const override { const char *e = std::getenv("PCG_ITER_FUSED"); return !(e && std::atoi(e) == 0); }
    void update_p(double *po, const double *pi, const double *r, const double *minv, const double *st, double rho_prev,
                  bool first, int publish_slot) override
    {
        if (publish_slot >= 0) std::memcpy(ring_[publish_slot], st_, sizeof(double) * ST_COUNT);
        const double beta = first ? 0.0 : st[ST_RHO_NEXT] / rho_prev;
        for (int64_t i = 0; i < n_; ++i) {
            const double z = minv[i] * r[i];
            po[i] = first ? z : z + beta * pi[i];
        }
    }
    // PCG_VEC_FUSED=0 keeps the driver on the split form, like the product
    bool vec_fused_available() const override { const char *e = std::getenv("PCG_VEC_FUSED"); return !fused_broken_ && !(e && std::atoi(e) == 0); }
    // Fault injection: the PCG_TEST_VEC_ERR_AT-th fused vector "launch" of this engine behaves like a k_vec<true> whose grid barrier
    // timed out: r', x' written, no sums, no p', st[ERR] raised - the driver has to finish the iteration in the split form.
    bool fused_broken_ = false;
    int64_t n_fused_ = 0;
    const int64_t vec_err_at_ = std::getenv("PCG_TEST_VEC_ERR_AT") ? std::atoll(std::getenv("PCG_TEST_VEC_ERR_AT")) : -1;
    void vec_fused_failed() override { fused_broken_ = true; st_[ST_ERR] = 0.0; }
    bool vec_update(double *st, int pq_src, const double *p, const double *q, const double *r, double *rnew, const double *xo,
                    double *xn, const double *minv, double *p_next, bool reduce_sums, const MailDesc *mail) override
    {
        if (mail && (p_next || !reduce_sums)) throw std::runtime_error("vec_update: mailbox all-reduce without the last-workgroup reduction");
        struct ReduceAtExit { HostBackend *b; double *st; bool on; const MailDesc *m;
                              ~ReduceAtExit() { if (on) { b->reduce_update(st + ST_SQP); if (m) host_mail_allreduce(*m, st + ST_SQP, 5); } } }
            reduce_at_exit{this, st, reduce_sums && !p_next, mail};
        const double rho = st[ST_RHO_NEXT];
        if (pq_src == 2) reduce_dot(st + ST_PQ);
        if (pq_src) scalar_alpha(st);
        for (double &v : up_) v = 0;
        const bool fused = p_next != nullptr;
        if (st[ST_STOP] != 0) {
            if (fused) for (int k = 0; k < 5; ++k) st[ST_SQP + k] = 0.0;
            return fused;
        }
        const double alpha = st[ST_ALPHA];
        for (int64_t i = 0; i < n_; ++i) {
            const bool w = own_free(i);
            if (w) { up_[0] += p[i] * p[i]; up_[1] += xo[i] * xo[i]; }
            const double rn = r[i] - alpha * q[i];
            rnew[i] = rn;
            xn[i] = xo[i] + alpha * p[i];
            const double z = minv[i] * rn;
            if (is_free(i) && std::isinf(z)) up_[4] += 1;
            if (w) { up_[2] += rn * rn; up_[3] += z * rn; }
        }
        if (!fused) return false;
        if (++n_fused_ == vec_err_at_) {
            st[ST_ERR] = 1.0;
            for (int64_t i = 0; i < n_; ++i) p_next[i] = std::nan("");
            return true;
        }
        for (int k = 0; k < 5; ++k) st[ST_SQP + k] = up_[k];
        const double beta = up_[3] / rho;                                // :475
        for (int64_t i = 0; i < n_; ++i) p_next[i] = minv[i] * rnew[i] + beta * p[i];   // :447, :479
        return true;
    }
    void reduce_update(double *red5) override { for (int k = 0; k < 5; ++k) red5[k] = up_[k]; }
    void residual(const double *b, const double *ax, double *r, const double *minv) override
    {
        for (double &v : res_) v = 0;
        for (int64_t i = 0; i < n_; ++i) {
            const double rn = b[i] - ax[i];
            r[i] = rn;
            const double z = minv[i] * rn;
            if (is_free(i) && std::isinf(z)) res_[2] += 1;
            if (own_free(i)) { res_[0] += rn * rn; res_[1] += z * rn; }
        }
    }
    void reduce_residual(double *red3) override { for (int k = 0; k < 3; ++k) red3[k] = res_[k]; }
    void dot_w(const double *a, const double *b) override
    {
        dotw_ = 0;
        for (int64_t i = 0; i < n_; ++i) if (own_free(i)) dotw_ += a[i] * b[i];
    }
    void reduce_dotw(double *red1) override { red1[0] = dotw_; }
    void copy_diag(double *d) override { std::memcpy(d, m_.diag.data(), sizeof(double) * n_); }
    void invert_free(double *minv, const double *d) override
    {
        for (int64_t i = 0; i < n_; ++i) minv[i] = is_free(i) ? 1.0 / d[i] : 0.0;
    }
    void axpby(double *o, double a, const double *x, double b, const double *y) override
    {
        for (int64_t i = 0; i < n_; ++i) o[i] = a * x[i] + b * y[i];
    }
    void scale(double *o, double a, const double *x) override { for (int64_t i = 0; i < n_; ++i) o[i] = a * x[i]; }
    void mask_free(double *x) override { for (int64_t i = 0; i < n_; ++i) if (!is_free(i)) x[i] = 0.0; }
    void set_profiling(int) override {}
    void collect_profile(double *ms, int64_t *c) override { *ms = 0; *c = 0; }
    int bench_hbm(size_t, int, int reps, float *ms) override { for (int k = 0; k < reps; ++k) ms[k] = 0.f; return 0; }
    int bench_spmv(const double *x, double *y, int, int reps, float *ms) override
    {
        if (!ebe_.groups.empty()) ebe_apply(x, y, 0, 2, true, false, 0);
        else spmv(x, y, 0, m_.n_slices, false, nullptr);
        for (int k = 0; k < reps; ++k) ms[k] = 0.f;
        return 0;
    }
};

std::unique_ptr<Backend> make_backend(int) { return std::unique_ptr<Backend>(new HostBackend()); }
int backend_device_count() { return 0; }
const char *backend_static_name() { return "hostops-test"; }
// make_rccl_comm / rccl_unique_ids of the test double: tests/hostops/local_comm.cpp (in-process mailboxes; there is no RCCL
// on the CPU).  The gloo tests drive the callback seam (pcg_set_comm) instead.
int64_t part_interface(int, int64_t, int64_t, const int64_t *, const int32_t *, const int32_t *, int64_t, int64_t *)
{
    throw std::runtime_error("the CPU test double has no device-side partition set-up");
}
int64_t part_local_numbering(int, int64_t, int64_t, const int32_t *, int32_t *, int32_t *)
{
    throw std::runtime_error("the CPU test double has no device-side partition set-up");
}

}  // namespace pcg
