// HIP back end for gfx950 (MI355X, CDNA4): the hand-written kernels of the PCG hot path.
//
// Everything here is HBM-bandwidth bound (SpMV arithmetic intensity ~0.2 flop/B), so the design
// rules are: every wave-level load is lane-contiguous (SELL layout: 8 or 16 B per lane, 512 B /
// 1 KiB per wave instruction), matrix data is streamed once with non-temporal loads so it does not
// evict the x vector from L2/MALL, slices are dealt round-robin to all waves of the chip (an
// XCD-partitioned mapping - block b & 7 = XCD owns one contiguous eighth - is kept behind
// PCG_SPMV_XCD=1; it measured 2.5 % slower: one matrix stream beats eight, x comes from MALL),
// reductions are wave64 shuffles -> LDS -> one partial per block -> a fixed tree (no float
// atomics: bit-reproducible run to run), and the vector part of an iteration is ONE streaming
// kernel with a grid barrier and its reductions inside (k_vec, kernels_vector.hpp).  The kernels
// live in kernels_{spmv,ebe,vector,probe}.hpp; this file is the back end that owns the device
// data and launches them.  No MFMA: MI355X's f64 matrix rate equals its vector rate and the assembled
// path is HBM-bound; the matrix-free operator (k_ebe_*) runs its 24x24 contraction on the vector
// FMA pipe with the element matrix as SGPR operands.
//
// Reference semantics implemented (src/solver/pcg_solver.py): k_spmv = calcMatVecProd :265-300 on
// the assembled operator (+ fused p.Ap.w :487); k_fixup = :332-334; k_update_p = :447,:472-479;
// k_vec = :487-516 plus :447-479 of the next iteration (kernels_vector.hpp); k_residual = :413-416/:530-533;
// k_dot_w = np.dot(a, b*w) :381.
#include "hip_common.hpp"
#include "kernels_spmv.hpp"
#include "kernels_ebe.hpp"
#include "kernels_vector.hpp"
#include "kernels_probe.hpp"

namespace pcg {

// ------------------------------------------------------------------------------------------------
// back end
// ------------------------------------------------------------------------------------------------
class HipBackend : public Backend {
    int dev_ = 0;
    int n_cu_ = 256;
    hipStream_t st_ = nullptr;
    // PCG_EBE_STREAMS=1: the element launches of the non-hex8 classes run on a second stream beside the hex8 launch of the same
    // phase (disjoint exclusive nodes, disjoint boundary slots; fork / join with events).  Measured on the two-level octree
    // mesh (1.2 M dof): a stand-alone apply 0.065 -> 0.059 ms, but the PCG iteration 0.097 -> 0.100 ms - the two cross-stream
    // waits per apply cost the look-ahead loop more than the overlap returns - so it is off by default.  ls_ = launch stream.
    // Round 3: ONE side stream per node-count class, measured on the multi-level octree mesh (1 M dof: the hanging-node classes
    // take 35 + 29 us next to the 30 us of the hex8 launch): 105 vs 107 us per apply, 130.8 vs 131.0 us per iteration - the
    // launches are throughput-bound, not tail-bound, so running them side by side buys nothing (profiles/r03_octree_ab_sessionG.log).
    // Still opt-in only: PCG_EBE_STREAMS=1.
    hipStream_t st2_[kChunkClasses] = {}, ls_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_[kChunkClasses] = {};
    int ebe_streams_mode_ = 0;                   // 1: PCG_EBE_STREAMS=1
    // Round 5, multi-part loop (PCG_EBE_PHASE_STREAMS): the INTERIOR phase of a split matrix-free apply on a side stream beside the
    // interface phase.  Kernel timeline of a 1.32 M-dof part (profiles/r05_multi_part_timeline_sessionC.md): the interface chunks (175
    // workgroups on 256 CUs) last 21 us - one chunk's chain of phases, as long as a full launch - and the interior launch (24 us) used
    // to wait behind them, the shared-node sums and the pack.  The two launches touch disjoint nodes and boundary slots and fit the
    // GPU together (904 of 1 024 resident workgroups): side by side the interior phase would hide behind interface phase + exchange.
    // MEASURED AND LOST (session d, same box, alternating processes): 122 - 125 us per iteration with the side stream against 115 - 117
    // without (profiles/r05_ab_phase_streams_sessionD.log) - the fork and the join are two more cross-stream waits per apply, and a wait
    // between streams costs this runtime more than the 17 us the overlap could return (the finding of PCG_EBE_STREAMS again).  Opt-in.
    hipStream_t st_phase_ = nullptr;
    hipEvent_t ev_phase_fork_ = nullptr, ev_phase_join_ = nullptr;
    int ebe_phase_streams_ = 0;
    bool phase_forked_ = false;
    // matrix
    int bs_ = 3;
    int64_t n_nodes_ = 0, n_ = 0, n_slices_ = 0, n_bnd_slices_ = 0;
    int C_ = 64;
    int64_t *d_slice_ptr_ = nullptr;
    int *d_cols_ = nullptr;
    unsigned short *d_cols16_ = nullptr;      // COL16 form (then d_cols_ stays null for 3x3-block matrices)
    int *d_colbase_ = nullptr;
    double *d_vals_ = nullptr, *d_diag_ = nullptr;
    uint8_t *d_flags_ = nullptr;
    // matrix-free operator
    struct EbeGroupDev { int nd; int64_t ne; int *dof; unsigned *sgn_bits; uint8_t *sgn_bytes; double *ck, *ke; };
    std::vector<EbeGroupDev> ebe_groups_;
    std::vector<EbeRange> ebe_ranges_[2];
    bool ebe_ = false;
    // chunked matrix-free operator
    struct ChunkClassDev {
        int nnp = 8, ept = 1;
        bool full = false, direct = false;
        int *list[2] = {nullptr, nullptr};
        int count[2] = {0, 0};
        unsigned short *lid = nullptr;
        double *ck = nullptr, *ke = nullptr, *ke_rows = nullptr;
        unsigned *sgn = nullptr;
    } chc_[kChunkClasses];
    // per-launch tables of the mixed-type chunks (k_ebe_mixed)
    MixTab mix_tab_[2] = {};
    int mix_mtm_ = 0;
    std::vector<int> mix_nodes_host_[2];
    std::vector<unsigned short> mix_tslot_host_[2];
    // per-launch tables of the hex8 class for k_ebe_hex (hex_mode_ > 0)
    HexTab hex_tab_[2] = {};
    std::vector<void *> hex_allocs_;
    std::vector<int> hex_nodes_host_[2];                 // to fold the ownership / free masks into the slot table (upload_masks)
    std::vector<unsigned short> hex_tslot_host_[2];
    // Same-box A/B at 10 M dof (round 2, profiles/r02_ebe_lab_*.log), one apply with the fused p.Ap:
    //   round-1 kernel (512-element chunks, chunk-id lists)            0.195 ms   (removed in round 3)
    //   k_ebe_hex,   256-element chunks, 5 blocks per CU, ds_add_f64   0.180 ms   <- meshes below 131 k elements
    //   k_ebe_hexs,  512-element chunks in two passes, ds_add_f64      0.158 ms   <- default for large meshes
    // (512 elements per thread pair at once, 6 blocks per CU, read-add-write accumulation, several chunks per block with the
    // next one prefetched, the matrix cores: all measured, all slower - DESIGN.md section 4b.)
    int hex_ept_ = 2, hex_npt_ = 3;
    int n_chunks_total_[2] = {0, 0};
    int sh_count_[2] = {0, 0};
    int *d_sh_node_[2] = {nullptr, nullptr}, *d_sh_ptr_[2] = {nullptr, nullptr};
    int sh_slot0_[2] = {0, 0};
    int *d_ch_dst_ = nullptr;
    double *d_ch_buf_ = nullptr;
    double *d_part_ebe_ = nullptr;    // fused-dot partials of the chunk / shared launches of one apply
    int cnt_ebe_ = 0;
    bool ch_needs_zero_ = true;
    int4 *d_ch_hdr_ = nullptr;
    int *d_ch_nodes_ = nullptr;
    unsigned short *d_ch_tslot_ = nullptr;
    // halo
    int *d_send_idx_ = nullptr, *d_fptr_ = nullptr, *d_fpos_ = nullptr;
    int64_t halo_count_ = 0, nb_dofs_ = 0;
    // partials
    double *d_part_ = nullptr;        // 5 * kMaxPartials (vector kernels)
    double *d_part_spmv_ = nullptr;   // kMaxPartials
    double *d_part_fix_ = nullptr;    // kMaxPartials
    int cnt_spmv_ = 0, cnt_fix_ = 0, cnt_vec_ = 0;
    // profiling
    bool prof_ = false, prof_vec_ = false;     // HIP events around the operator launches / the vector-phase launches
    static constexpr int kMaxEv = 8192;
    std::vector<hipEvent_t> ev0_, ev1_, evv0_, evv1_;   // operator launches / vector-phase launches
    int ev_used_ = 0, evv_used_ = 0;
    int64_t ev_applies_ = 0;

    int vec_grid(int64_t n) const
    {
        int64_t g = (n / 2 + kBlock - 1) / kBlock;
        int64_t cap = (int64_t)n_cu_ * 8;
        if (cap > kMaxPartials) cap = kMaxPartials;
        if (g > cap) g = cap;
        return (int)(g < 1 ? 1 : g);
    }
    int spmv_grid(int64_t slices) const
    {
        int64_t g = (slices + kWavesPerBlock - 1) / kWavesPerBlock;
        int64_t cap = (int64_t)n_cu_ * spmv_blocks_per_cu_;
        if (cap > kMaxPartials) cap = kMaxPartials;
        if (g > cap) g = cap;
        g = (g + 7) / 8 * 8;           // whole blocks per XCD
        return (int)(g < 8 ? 8 : g);
    }
    int spmv_blocks_per_cu_ = 4;
    // k_spmv in `parts` launches that write their y at their end (round 6, kernels_spmv.hpp HOLD) or in one launch: the same bits; which is
    // faster depends on the box and on where the vectors landed, so tune_operator() times both on the solve's own vectors.
    // PCG_SPMV_HOLD=0 / 4 forces one form (and switches the timing off).
    bool spmv_split_ = getenv("PCG_SPMV_HOLD") ? atoi(getenv("PCG_SPMV_HOLD")) != 0 : false;
    bool spmv_split_forced_ = getenv("PCG_SPMV_HOLD") != nullptr;
    int spmv_launches_ = 1;           // launches of the last k_spmv apply
    int xcd_aware_ = 0;        // A/B on MI355X (profiles/r01_tune_spmv.json): plain round-robin 1.156 ms vs XCD-partitioned 1.185 ms
    bool bench_dot_ = false;
    // Non-temporal accesses in the vector kernels (PCG_VEC_NT, bit mask; A/B: tools/vec_nt_ab.py re-reads it per solve).
    // bit 0: the vectors the iteration rewrites (p in k_update_p; r', x', p' in k_vec) are stored non-temporally.  Plain
    //        stores leave the rewritten lines dirty in the memory-side cache; they are then written out underneath the next
    //        operator's read stream: a stand-alone SpMV whose x was just rewritten by a plain-store kernel runs 2.5-6 % slower,
    //        with `nt` stores 0.5-1 % (sc1 / sc0 sc1 do not help; profiles/r02_spmv_launch_context.txt).  In the loop at
    //        10 M dof: assembled 780 -> 842 it/s (SpMV 1.120 -> 1.033 ms), matrix-free 3070 -> 3110 it/s, same process.
    // bit 2: the vector kernels' streaming loads are non-temporal too: 845 / 3184 it/s.
    // bit 1: non-temporal y stores in k_spmv: no effect, off.  Results are bit-identical in every combination.
    int vec_nt_ = 5;
    // PCG_ALLOC_CONTIG=1: arrays of 32 MB and more are requested as physically contiguous VRAM (hipDeviceMallocContiguous; plain
    // hipMalloc when that fails), =2 also reports every such allocation on stderr.  OFF by default: in alternating same-box
    // processes it made the SpMV 2.7-4 % faster on one box (1.199 -> 1.151 ms, profiles/r02_alloc_contiguous_ab.txt), changed
    // nothing on another (1.027 vs 1.039 ms) and made the matrix-free operator 17 % SLOWER there (3220 -> 2685 it/s at 10 M
    // dof, twice each; profiles/r02_alloc_contiguous_ab.txt, session AN).
    int alloc_contig_ = 0;

public:
    explicit HipBackend(int device)
    {
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
            throw std::runtime_error("no HIP device visible (this engine has no CPU fallback)");
        if (device < 0 || device >= cnt) throw std::runtime_error("device index out of range");
        dev_ = device;
        HIP_CHECK(hipSetDevice(dev_));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev_));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw std::runtime_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        n_cu_ = prop.multiProcessorCount;
        HIP_CHECK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        for (int c = 1; c < kChunkClasses; ++c) {
            HIP_CHECK(hipStreamCreateWithFlags(&st2_[c], hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&ev_join_[c], hipEventDisableTiming));
        }
        ls_ = st_;
        if (const char *e = getenv("PCG_EBE_STREAMS")) ebe_streams_mode_ = atoi(e) != 0 ? 1 : 0;
        HIP_CHECK(hipStreamCreateWithFlags(&st_phase_, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_phase_fork_, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_phase_join_, hipEventDisableTiming));
        if (const char *e = getenv("PCG_EBE_ROWS_LDS")) rows_lds_mode_ = atoi(e) != 0 ? 1 : 0;
        if (const char *e = getenv("PCG_EBE_XCD")) ebe_xcd_ = std::max(0, atoi(e));
        d_part_ = (double *)alloc(sizeof(double) * 5 * kMaxPartials);
        d_part_spmv_ = (double *)alloc(sizeof(double) * kMaxPartials);
        d_part_fix_ = (double *)alloc(sizeof(double) * kMaxPartials);
        // fused vector phase (k_vec<true>): its grid barrier needs every workgroup resident - one 1024-thread workgroup per
        // CU, asked of the occupancy query here; PCG_VEC_FUSED=0 keeps the split form (k_vec<false> + k_reduce + k_update_p)
        d_vec_sync_ = (unsigned long long *)alloc(sizeof(unsigned long long) * kVecSyncWords);
        HIP_CHECK(hipMemsetAsync(d_vec_sync_, 0, sizeof(unsigned long long) * kVecSyncWords, st_));
        d_vec_pub_ = (double *)alloc(sizeof(double) * 2 * 5 * kMaxPartials);         // in-band form: every slot starts as the sentinel
        HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)d_vec_pub_, (int)kVecSentinelHalf, (size_t)2 * 2 * 5 * kMaxPartials, st_));
        {
            int per_cu = 0;
            const hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_vec<true>, kVecBlock, 0);
            vec_fused_hw_ = rc == hipSuccess && per_cu >= 1;
            (void)hipGetLastError();
        }
        // development knobs (tools/tune_spmv.py); the defaults are the tuned values
        if (const char *e = getenv("PCG_SPMV_BLOCKS_PER_CU")) spmv_blocks_per_cu_ = std::max(1, atoi(e));
        if (const char *e = getenv("PCG_SPMV_XCD")) xcd_aware_ = atoi(e) != 0;
        if (const char *e = getenv("PCG_BENCH_SPMV_DOT")) bench_dot_ = atoi(e) != 0;
        if (const char *e = getenv("PCG_ALLOC_CONTIG")) alloc_contig_ = atoi(e);
        if (const char *e = getenv("PCG_SPMV_DICT_BLOCK")) { const int b = atoi(e); if (b == 256 || b == 512 || b == 640 || b == 1024) dict_block_ = b; }
        reload_tuning();
    }
    ~HipBackend() override
    {
        (void)hipSetDevice(dev_);
        for (void *p : {(void *)d_bidx_, (void *)d_dict_, (void *)d_slice_ptr_, (void *)d_cols_, (void *)d_cols16_, (void *)d_colbase_, (void *)d_vals_, (void *)d_diag_, (void *)d_flags_,
                        (void *)d_send_idx_, (void *)d_fptr_, (void *)d_fpos_, (void *)d_part_, (void *)d_part_spmv_,
                        (void *)d_part_fix_, (void *)d_vec_sync_, (void *)d_vec_pub_, (void *)d_last_cnt_, (void *)d_ptr4_, (void *)d_ov_slice_ptr_, (void *)d_ov_rows_, (void *)d_ov_cols_,
                        (void *)d_ov_vals_, (void *)d_ov_mask_, (void *)d_win_slice_, (void *)d_win_ov_})
            if (p) (void)hipFree(p);
        for (auto &D : chc_)
            for (void *p : {(void *)D.list[0], (void *)D.list[1], (void *)D.lid, (void *)D.ck, (void *)D.sgn, (void *)D.ke, (void *)D.ke_rows})
                if (p) (void)hipFree(p);
        for (void *p : {(void *)d_ch_hdr_, (void *)d_ch_nodes_, (void *)d_ch_tslot_, (void *)d_ch_dst_, (void *)d_ch_buf_, (void *)d_part_ebe_, (void *)d_sh_node_[0],
                        (void *)d_sh_node_[1], (void *)d_sh_ptr_[0], (void *)d_sh_ptr_[1]})
            if (p) (void)hipFree(p);
        for (auto &D : ebe_groups_)
            for (void *p : {(void *)D.dof, (void *)D.sgn_bits, (void *)D.sgn_bytes, (void *)D.ck, (void *)D.ke})
                if (p) (void)hipFree(p);
        for (void *p : hex_allocs_) (void)hipFree(p);
        for (auto e : ev0_) (void)hipEventDestroy(e);
        for (auto e : ev1_) (void)hipEventDestroy(e);
        for (auto e : evv0_) (void)hipEventDestroy(e);
        for (auto e : evv1_) (void)hipEventDestroy(e);
        if (h_mirror_) {
            (void)hipHostFree(h_mirror_);
            for (auto e : ev_slot_) (void)hipEventDestroy(e);
        }
        if (ev_fork_) (void)hipEventDestroy(ev_fork_);
        if (ev_phase_fork_) (void)hipEventDestroy(ev_phase_fork_);
        if (ev_phase_join_) (void)hipEventDestroy(ev_phase_join_);
        if (st_phase_) (void)hipStreamDestroy(st_phase_);
        for (int c = 1; c < kChunkClasses; ++c) {
            if (ev_join_[c]) (void)hipEventDestroy(ev_join_[c]);
            if (st2_[c]) (void)hipStreamDestroy(st2_[c]);
        }
        if (st_) (void)hipStreamDestroy(st_);
    }
    const char *name() const override { return "hip-gfx950"; }
    int device() const override { return dev_; }
    void bind_thread() override { HIP_CHECK(hipSetDevice(dev_)); }
    void *stream() override { return (void *)st_; }
    void *alloc(size_t bytes) override
    {
        HIP_CHECK(hipSetDevice(dev_));
        void *p = nullptr;
        if (alloc_contig_ && bytes >= ((size_t)32 << 20)) {         // physically contiguous VRAM for the big arrays
            const hipError_t rc = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
            if (alloc_contig_ > 1) fprintf(stderr, "[pcg] contiguous alloc of %.1f MB: %s\n", bytes / 1e6, rc == hipSuccess ? "ok" : hipGetErrorString(rc));
            if (rc == hipSuccess && p) return p;
            (void)hipGetLastError();
            p = nullptr;
        }
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
        return p;
    }
    void release(void *p) override { (void)hipFree(p); }
    void h2d(void *d, const void *s, size_t b) override
    {
        HIP_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    void d2h(void *d, const void *s, size_t b) override
    {
        HIP_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    void d2d(void *d, const void *s, size_t b) override { HIP_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, st_)); }
    void zero(void *d, size_t b) override { HIP_CHECK(hipMemsetAsync(d, 0, b, st_)); }
    void sync() override { HIP_CHECK(hipStreamSynchronize(st_)); }

    void upload_matrix(const SellHost &m) override
    {
        bs_ = m.bs;
        n_nodes_ = m.n_nodes; n_ = m.bs * m.n_nodes; n_slices_ = m.n_slices; n_bnd_slices_ = m.n_bnd_slices; C_ = m.C;
        d_slice_ptr_ = (int64_t *)alloc(sizeof(int64_t) * m.slice_ptr.size());
        d_vals_ = m.bidx.empty() ? (double *)alloc(sizeof(double) * m.vals.size()) : nullptr;
        d_diag_ = (double *)alloc(sizeof(double) * m.diag.size());
        d_flags_ = (uint8_t *)alloc((size_t)n_ + 16);
        h2d(d_slice_ptr_, m.slice_ptr.data(), sizeof(int64_t) * m.slice_ptr.size());
        // 16-bit column offsets when every slice's columns span < 65536 (k_spmv COL16); PCG_SPMV_COL16=0 keeps 32 bits
        bool col16 = m.bs == 3 && m.n_slices > 0;
        if (const char *e = getenv("PCG_SPMV_COL16")) col16 = col16 && atoi(e) != 0;
        std::vector<int> cbase;
        if (col16) {
            cbase.assign((size_t)m.n_slices, 0);
            for (int64_t sl = 0; sl < m.n_slices && col16; ++sl) {
                const int64_t a = m.slice_ptr[sl] * m.C, b = m.slice_ptr[sl + 1] * m.C;
                int lo = INT32_MAX, hi = 0;
                for (int64_t k = a; k < b; ++k) { lo = std::min(lo, m.cols[k]); hi = std::max(hi, m.cols[k]); }
                if (b > a) { cbase[sl] = lo; if (hi - lo > 65535) col16 = false; }
            }
        }
        if (col16 && !m.bidx.empty()) {
            // dictionary format: a stored block is ONE 32-bit word (column offset | table index << 16) and a lane's words of four
            // consecutive block columns sit together (k_spmv_dict COL16): ci[((ptr4[s] + k / 4) * 64 + lane) * 4 + k % 4]
            std::vector<int64_t> ptr4((size_t)m.n_slices + 1, 0);
            for (int64_t sl = 0; sl < m.n_slices; ++sl) ptr4[sl + 1] = ptr4[sl] + (m.slice_ptr[sl + 1] - m.slice_ptr[sl] + 3) / 4;
            std::vector<unsigned> ci((size_t)ptr4.back() * 256, 0u);
            for (int64_t sl = 0; sl < m.n_slices; ++sl) {
                const int64_t base = m.slice_ptr[sl], w = m.slice_ptr[sl + 1] - base;
                for (int64_t k = 0; k < w; ++k)
                    for (int l = 0; l < 64; ++l) {
                        const size_t q = (size_t)(base + k) * 64 + l;
                        ci[((size_t)(ptr4[sl] + k / 4) * 64 + l) * 4 + k % 4] = (unsigned)(m.cols[q] - cbase[sl]) | ((unsigned)m.bidx[q] << 16);
                    }
            }
            d_cols16_ = (unsigned short *)alloc(sizeof(unsigned) * std::max<size_t>(4, ci.size()));
            d_colbase_ = (int *)alloc(sizeof(int) * cbase.size());
            d_ptr4_ = (int64_t *)alloc(sizeof(int64_t) * ptr4.size());
            h2d(d_cols16_, ci.data(), sizeof(unsigned) * ci.size());
            h2d(d_colbase_, cbase.data(), sizeof(int) * cbase.size());
            h2d(d_ptr4_, ptr4.data(), sizeof(int64_t) * ptr4.size());
            dict_packed_ = true;
        } else if (col16) {
            std::vector<unsigned short> c16(m.cols.size());
            for (int64_t sl = 0; sl < m.n_slices; ++sl)
                for (int64_t k = m.slice_ptr[sl] * m.C, b = m.slice_ptr[sl + 1] * m.C; k < b; ++k)
                    c16[k] = (unsigned short)(m.cols[k] - cbase[sl]);
            d_cols16_ = (unsigned short *)alloc(sizeof(unsigned short) * std::max<size_t>(1, c16.size()));
            d_colbase_ = (int *)alloc(sizeof(int) * cbase.size());
            h2d(d_cols16_, c16.data(), sizeof(unsigned short) * c16.size());
            h2d(d_colbase_, cbase.data(), sizeof(int) * cbase.size());
        } else {
            d_cols_ = (int *)alloc(sizeof(int) * m.cols.size());
            h2d(d_cols_, m.cols.data(), sizeof(int) * m.cols.size());
        }
        if (!m.bidx.empty()) {                              // dictionary format: indices + table instead of the values
            n_unique_ = (int)m.n_unique();
            d_bidx_ = (unsigned short *)alloc(sizeof(unsigned short) * (dict_packed_ ? 8 : m.bidx.size()));   // unused when packed
            d_dict_ = (double *)alloc(sizeof(double) * std::max<size_t>(9, m.dict.size()));
            if (!dict_packed_) h2d(d_bidx_, m.bidx.data(), sizeof(unsigned short) * m.bidx.size());
            h2d(d_dict_, m.dict.data(), sizeof(double) * m.dict.size());
            dict_lds_ = true;
            if (const char *e = getenv("PCG_SPMV_DICT_LDS")) dict_lds_ = atoi(e) != 0;
            n_lds_ = dict_lds_ ? n_unique_ : 0;
            dict_mixed_ = false;
            if (dict_lds_ && (size_t)n_unique_ * 80 > kDictLdsBytes) {
                // larger than a workgroup's default LDS window: ONE 1024-thread workgroup per CU (16 waves share the copy)
                // with up to kDictLdsBytesMax of the table's head; the tail is read through the caches (k_spmv_dict MIXED)
                n_lds_ = (int)std::min<size_t>((size_t)n_unique_, kDictLdsBytesMax / 80);
                dict_mixed_ = n_lds_ < n_unique_;
                if (const char *e = getenv("PCG_SPMV_DICT_LDS_ENTRIES")) {        // tests: force a small head
                    n_lds_ = std::max(1, std::min(n_unique_, atoi(e)));
                    dict_mixed_ = n_lds_ < n_unique_;
                }
                const int bytes = (int)kDictLdsBytesMax;
                auto raise = [&](const void *fn) { HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); };
                raise((const void *)k_spmv_dict<true, true, true, 1024, true>);   raise((const void *)k_spmv_dict<false, true, true, 1024, true>);
                raise((const void *)k_spmv_dict<true, false, true, 1024, true>);  raise((const void *)k_spmv_dict<false, false, true, 1024, true>);
                raise((const void *)k_spmv_dict<true, true, true, 1024, false>);  raise((const void *)k_spmv_dict<false, true, true, 1024, false>);
                raise((const void *)k_spmv_dict<true, false, true, 1024, false>); raise((const void *)k_spmv_dict<false, false, true, 1024, false>);
                dict_big_ = true;
            } else if (dict_lds_) {
                if (const char *e = getenv("PCG_SPMV_DICT_LDS_ENTRIES")) {
                    n_lds_ = std::max(1, std::min(n_unique_, atoi(e)));
                    dict_mixed_ = n_lds_ < n_unique_;
                    if (dict_mixed_) dict_big_ = true;                            // the MIXED kernel exists for 1024 threads only
                }
            }
        } else {
            h2d(d_vals_, m.vals.data(), sizeof(double) * m.vals.size());
            probe_value_placement(m);
        }
        h2d(d_diag_, m.diag.data(), sizeof(double) * m.diag.size());
        nb_dofs_ = std::min<int64_t>(n_, n_bnd_slices_ * C_ * m.bs);
        if (m.ov_slices > 0) {                              // split matrix: the overflow part (k_spmv_ovf)
            ov_slices_ = m.ov_slices; ov_bnd_slices_ = m.ov_bnd_slices;
            auto up = [&](auto *&dst, const auto &v) {
                using T = std::remove_reference_t<decltype(*dst)>;
                dst = (T *)alloc(sizeof(v[0]) * std::max<size_t>(1, v.size()));
                h2d(dst, v.data(), sizeof(v[0]) * v.size());
            };
            up(d_ov_slice_ptr_, m.ov_slice_ptr); up(d_ov_rows_, m.ov_rows); up(d_ov_cols_, m.ov_cols); up(d_ov_vals_, m.ov_vals);
            static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "mask words");
            d_ov_mask_ = (unsigned long long *)alloc(sizeof(uint64_t) * m.ov_mask.size());
            h2d(d_ov_mask_, m.ov_mask.data(), sizeof(uint64_t) * m.ov_mask.size());
            if (!m.win_slice.empty()) {                          // windowed form: one launch (k_spmv_win)
                n_windows_ = (int64_t)m.win_slice.size() - 1; n_bnd_windows_ = m.n_bnd_windows;
                up(d_win_slice_, m.win_slice); up(d_win_ov_, m.win_ov);
            }
        }
    }
    void upload_scalar_copy(Backend &src_be, const std::vector<int64_t> &ptr1, const std::vector<int64_t> &src_ptr, int64_t n_rows) override
    {
        auto *S = dynamic_cast<HipBackend *>(&src_be);
        if (!S || S->dev_ != dev_) throw std::runtime_error("scalar copy: the source engine lives on another back end / device");
        if (S->bs_ != 3 || S->C_ != 64 || S->d_bidx_ || S->ov_slices_ || !S->d_vals_) throw std::runtime_error("scalar copy: plain, unsplit 3x3-block format only");
        (void)src_ptr;
        bs_ = 1; C_ = 64;
        n_nodes_ = n_rows; n_ = n_rows; n_slices_ = (int64_t)ptr1.size() - 1; n_bnd_slices_ = 0;
        const size_t tot = (size_t)ptr1.back() * 64;
        d_slice_ptr_ = (int64_t *)alloc(sizeof(int64_t) * ptr1.size());
        h2d(d_slice_ptr_, ptr1.data(), sizeof(int64_t) * ptr1.size());
        d_cols_ = (int *)alloc(sizeof(int) * std::max<size_t>(1, tot));
        d_vals_ = (double *)alloc(sizeof(double) * std::max<size_t>(1, tot));
        d_diag_ = (double *)alloc(sizeof(double) * (size_t)n_);
        d_flags_ = (uint8_t *)alloc((size_t)n_ + 16);
        HIP_CHECK(hipStreamSynchronize(S->st_));
        HIP_CHECK(hipMemcpyAsync(d_diag_, S->d_diag_, sizeof(double) * (size_t)n_, hipMemcpyDeviceToDevice, st_));
        // lanes of the last slice past n_rows are written by no thread of k_expand_scalar, yet k_spmv_scalar walks every lane of a
        // slice over its full width before the row guard: value 0 / column 0 there, not whatever the allocation held before
        HIP_CHECK(hipMemsetAsync(d_cols_, 0, sizeof(int) * std::max<size_t>(1, tot), st_));
        HIP_CHECK(hipMemsetAsync(d_vals_, 0, sizeof(double) * std::max<size_t>(1, tot), st_));
        const int grid = (int)((n_rows + kBlock - 1) / kBlock);
        if (S->d_cols16_)
            hipLaunchKernelGGL((k_expand_scalar<true>), dim3(grid), dim3(kBlock), 0, st_, S->d_slice_ptr_, (const void *)S->d_cols16_, S->d_colbase_, S->d_vals_,
                               d_slice_ptr_, d_cols_, d_vals_, n_rows);
        else
            hipLaunchKernelGGL((k_expand_scalar<false>), dim3(grid), dim3(kBlock), 0, st_, S->d_slice_ptr_, (const void *)S->d_cols_, S->d_colbase_, S->d_vals_,
                               d_slice_ptr_, d_cols_, d_vals_, n_rows);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    // Where the value array lands physically is not the engine's choice, and it matters: the identical launch ran at 1.05 and at
    // 1.22 ms in consecutive processes on ONE box, and FOUR allocations of the same 6.5 GB array inside one process gave 1.124 /
    // 1.061 / 1.039 / 1.081 ms per stand-alone launch (profiles/r03_value_array_placements_standalone.txt).  This probe
    // (PCG_SPMV_PLACEMENTS=k, OFF by default) tries k placements at upload - each allocated while the earlier ones are still held,
    // device-to-device copy - and keeps the fastest.  It is off because the stand-alone winner is NOT the in-loop winner: with
    // the probe the SpMV of the PCG loop ran at 1.143 / 1.194 ms against 1.054 / 1.030 ms without it (same box, alternating
    // processes, profiles/r03_value_array_placements_in_loop.txt) - what counts is the placement of the value array RELATIVE
    // to the vectors allocated after it, which the probe's own allocations disturb.  Kept as the reproducer of that finding.
    std::vector<float> placement_ms_;
    void probe_value_placement(const SellHost &m)
    {
        int k = 1;
        if (const char *e = getenv("PCG_SPMV_PLACEMENTS")) k = std::max(1, std::min(8, atoi(e)));
        const size_t bytes = sizeof(double) * m.vals.size();
        if (k < 2 || bs_ != 3 || C_ != 64 || bytes < ((size_t)256 << 20) || bytes > ((size_t)24 << 30) || (!d_cols16_ && !d_cols_)) return;
        double *x = (double *)alloc(sizeof(double) * (size_t)n_), *y = (double *)alloc(sizeof(double) * (size_t)n_);
        HIP_CHECK(hipMemsetAsync(x, 0x3c, sizeof(double) * (size_t)n_, st_));          // finite non-zero doubles
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        const int grid = spmv_grid(n_slices_);
        auto time_it = [&]() {
            float best = 1e30f;
            for (int rep = 0; rep < 8; ++rep) {
                HIP_CHECK(hipEventRecord(e0, st_));
                launch_spmv<1>(x, y, 0, n_slices_, false, grid);
                HIP_CHECK(hipEventRecord(e1, st_));
                HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0;
                HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) best = std::min(best, ms);
            }
            return best;
        };
        std::vector<double *> cand{d_vals_};
        placement_ms_.assign(1, time_it());
        int keep = 0;
        for (int c = 1; c < k; ++c) {
            void *p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            HIP_CHECK(hipMemcpyAsync(p, cand[0], bytes, hipMemcpyDeviceToDevice, st_));
            cand.push_back((double *)p);
            d_vals_ = (double *)p;
            placement_ms_.push_back(time_it());
            if (placement_ms_[c] < placement_ms_[keep]) keep = c;
        }
        d_vals_ = cand[keep];
        for (int c = 0; c < (int)cand.size(); ++c)
            if (c != keep) (void)hipFree(cand[c]);
        HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
        release(x); release(y);
        if (getenv("PCG_SPMV_PLACEMENTS_LOG")) {
            fprintf(stderr, "[pcg] value-array placements (ms per SpMV launch):");
            for (size_t c = 0; c < placement_ms_.size(); ++c) fprintf(stderr, " %.4f%s", placement_ms_[c], (int)c == keep ? "*" : "");
            fprintf(stderr, "\n");
        }
    }
    void upload_ebe(const EbeHost &m) override
    {
        ebe_ = true;
        n_nodes_ = m.n_nodes; n_ = 3 * m.n_nodes;
        d_diag_ = (double *)alloc(sizeof(double) * m.diag.size());
        h2d(d_diag_, m.diag.data(), sizeof(double) * m.diag.size());
        d_flags_ = (uint8_t *)alloc((size_t)n_ + 16);
        for (const auto &G : m.groups) {
            EbeGroupDev D{G.nd, G.ne, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (G.ne == 0) { ebe_groups_.push_back(D); continue; }      // handled by the chunked form
            D.dof = (int *)alloc(sizeof(int) * G.dof.size());
            h2d(D.dof, G.dof.data(), sizeof(int) * G.dof.size());
            D.ck = (double *)alloc(sizeof(double) * G.ck.size());
            h2d(D.ck, G.ck.data(), sizeof(double) * G.ck.size());
            D.ke = (double *)alloc(sizeof(double) * G.ke.size());
            h2d(D.ke, G.ke.data(), sizeof(double) * G.ke.size());
            if (G.nd == 24) {                                  // fast path: 24 sign bits per element in one word
                std::vector<unsigned> bits((size_t)G.ne, 0u);
                for (int a = 0; a < G.nd; ++a)
                    for (int64_t e = 0; e < G.ne; ++e)
                        if (G.sign[(size_t)a * G.ne + e]) bits[e] |= (1u << a);
                D.sgn_bits = (unsigned *)alloc(sizeof(unsigned) * bits.size());
                h2d(D.sgn_bits, bits.data(), sizeof(unsigned) * bits.size());
            } else {
                D.sgn_bytes = (uint8_t *)alloc(G.sign.size());
                h2d(D.sgn_bytes, G.sign.data(), G.sign.size());
            }
            ebe_groups_.push_back(D);
        }
        for (int ph = 0; ph < 2; ++ph) ebe_ranges_[ph] = m.ranges[ph];
        const auto &C = m.chunked;
        if (C.n_chunks > 0) {
            auto up = [&](auto *&dst, const auto &v) {
                using T = std::remove_reference_t<decltype(*dst)>;
                dst = (T *)alloc(sizeof(v[0]) * std::max<size_t>(1, v.size()));
                h2d(dst, v.data(), sizeof(v[0]) * v.size());
            };
            d_ch_hdr_ = (int4 *)alloc(sizeof(int) * C.hdr.size());
            h2d(d_ch_hdr_, C.hdr.data(), sizeof(int) * C.hdr.size());
            up(d_ch_nodes_, C.nodes); up(d_ch_dst_, C.dst); up(d_ch_tslot_, C.tslot);
            d_ch_buf_ = (double *)alloc(sizeof(double) * 3 * (size_t)std::max<int64_t>(1, C.n_slots));
            ch_needs_zero_ = C.needs_zero;
            size_t np = 8;
            for (int c = 0; c < kChunkClasses; ++c) {
                const auto &K = C.cls[c];
                auto &D = chc_[c];
                D.nnp = K.nnp; D.ept = K.ept; D.full = K.full; D.direct = K.direct;
                if (K.n_chunks == 0) continue;
                if (c == kMixedClass) {                           // mixed-type chunks: per-launch tables (build_mixed_tables)
                    up(D.ke, K.ke_col);
                    for (int ph = 0; ph < 2; ++ph) {
                        D.count[ph] = (int)K.list[ph].size();
                        n_chunks_total_[ph] += D.count[ph];
                        np += K.list[ph].size();
                    }
                    build_mixed_tables(C);
                    continue;
                }
                up(D.lid, K.lid); up(D.ck, K.ck); up(D.sgn, K.sgn); up(D.ke, K.ke_col);
                if (!K.ke_rows.empty()) up(D.ke_rows, K.ke_rows);
                for (int ph = 0; ph < 2; ++ph) {
                    D.count[ph] = (int)K.list[ph].size();
                    n_chunks_total_[ph] += D.count[ph];
                    np += K.list[ph].size();
                    if (D.count[ph]) up(D.list[ph], K.list[ph]);
                }
            }
            if (C.cls[0].n_chunks > 0) build_hex_tables(C);
            for (int ph = 0; ph < 2; ++ph) {
                sh_count_[ph] = (int)C.sh_node[ph].size();
                np += (3 * C.sh_node[ph].size() + kBlock - 1) / kBlock;
                sh_slot0_[ph] = ph ? (int)C.sh_ptr[0].back() : 0;
                if (sh_count_[ph]) { up(d_sh_node_[ph], C.sh_node[ph]); up(d_sh_ptr_[ph], C.sh_ptr[ph]); }
            }
            d_part_ebe_ = (double *)alloc(sizeof(double) * np);
        }
    }
    // k_ebe_mixed: the mixed-type chunks laid out per launch (phase), block b's data at fixed strides of b; the tile arrays are
    // shared by both phases (a chunk's header names its first tile)
    void build_mixed_tables(const EbeChunkedHost &C)
    {
        const auto &K = C.cls[kMixedClass];
        const auto &M = C.mixed;
        const int CE = kMixedHexSlots, MAXN = kChunkMaxNodes;
        auto up = [&](const auto &v) {
            void *d = alloc(sizeof(v[0]) * std::max<size_t>(1, v.size()));
            h2d(d, v.data(), sizeof(v[0]) * v.size());
            hex_allocs_.push_back(d);
            return d;
        };
        if (M.words > 3) throw std::runtime_error("k_ebe_mixed: more than 96 dofs per element");
        std::vector<int> tinfo(2 * (size_t)std::max<int64_t>(1, M.n_tiles), 0);
        for (int64_t t = 0; t < M.n_tiles; ++t) {
            const auto &T = M.types[M.tile_type[t]];
            tinfo[2 * t] = T.nn | (T.J << 8) | (M.tile_ncol[t] << 16);
            tinfo[2 * t + 1] = (int)(T.frag_off / 64);
        }
        // sign bits regrouped for the tile's lanes: lane group g of element e handles the nodes g, g + 4, ... - bit 3 j + c of its word is
        // the sign of dof c of node 4 j + g (k-step 3 j + c on the way in, accumulator 3 j + c on the way out: constant shifts in the kernel)
        std::vector<unsigned> tsgw((size_t)std::max<int64_t>(1, M.n_tiles) * 64, 0u);
        for (int64_t t = 0; t < M.n_tiles; ++t) {
            const auto &T = M.types[M.tile_type[t]];
            for (int e = 0; e < 16; ++e)
                for (int l = 0; l < T.nn; ++l)
                    for (int c = 0; c < 3; ++c) {
                        const int a = 3 * l + c;
                        if ((M.tsgn[((size_t)t * M.words + a / 32) * 16 + e] >> (a % 32)) & 1u)
                            tsgw[((size_t)t * 4 + l % 4) * 16 + e] |= 1u << (3 * (l / 4) + c);
                    }
        }
        // k_ebe_mtile (EbeMixedHost::hex_tile_type): per lane of a hex tile the two slots and the six sign bits it handles; per tile its wait count
        mix_hex_tiles_ = M.hex_tile_type >= 0;
        std::vector<uint2> hrec;
        if (mix_hex_tiles_) {
            hrec.assign((size_t)std::max<int64_t>(1, M.n_tiles) * 64, make_uint2(0u, 0u));
            for (int64_t t = 0; t < M.n_tiles; ++t) {
                if (M.tile_type[t] != M.hex_tile_type) continue;
                for (int e = 0; e < 16; ++e) {
                    if (M.tcol[(size_t)t * 16 + e] == 255) continue;
                    for (int g = 0; g < 4; ++g) {
                        uint2 r;
                        r.x = (unsigned)M.tlid[((size_t)t * M.nnpt + g) * 16 + e] | (unsigned)M.tlid[((size_t)t * M.nnpt + g + 4) * 16 + e] << 16;
                        r.y = tsgw[((size_t)t * 4 + g) * 16 + e] | (unsigned)M.tperm[(size_t)t * 16 + e] << 8 | 0x80000000u;
                        hrec[(size_t)t * 64 + g * 16 + e] = r;
                    }
                }
            }
        }
        const void *d_hrec = mix_hex_tiles_ ? up(hrec) : nullptr, *d_twait = mix_hex_tiles_ ? up(M.tile_wait) : nullptr;
        const void *d_tinfo = up(tinfo), *d_tlid = up(M.tlid), *d_tck = up(M.tck), *d_tsgw = up(tsgw), *d_tcol = up(M.tcol), *d_tperm = up(M.tperm), *d_frag = up(M.frag);
        mix_mtm_ = std::max(2, M.max_mt);
        if (const char *e = getenv("PCG_EBE_MIX_MTM")) mix_mtm_ = std::max(mix_mtm_, std::min(6, atoi(e)));   // development: a larger instantiation (>= 5: 168 VGPRs, 3 workgroups per CU)
        for (int ph = 0; ph < 2; ++ph) {
            const size_t n = K.list[ph].size();
            if (!n) continue;
            std::vector<int> hdr(8 * n, 0), nodes(n * MAXN, -1), dst(n * MAXN, INT32_MIN);
            std::vector<unsigned short> tslot(n * MAXN, 0), lid(n * 8 * CE, 0);
            std::vector<double> ck(n * CE, 0.0);
            std::vector<unsigned> sgn(n * CE, 0xff000000u);
            for (size_t b = 0; b < n; ++b) {
                const int32_t *h = &C.hdr[(size_t)K.list[ph][b] * 8];
                const int32_t off = h[0], nn = h[1], nh = h[3], kci = h[4];
                int any_sign = 0;
                for (int e = 0; e < nh; ++e) any_sign |= (K.sgn[(size_t)kci * CE + e] & 0x00ffffffu) != 0;
                hdr[8 * b] = nn; hdr[8 * b + 1] = h[2]; hdr[8 * b + 2] = mix_hex_tiles_ ? M.chunk_hex_tiles[kci] : nh; hdr[8 * b + 3] = any_sign;
                hdr[8 * b + 4] = h[5]; hdr[8 * b + 5] = h[7];
                for (int k = 0; k < nn; ++k) {
                    nodes[b * MAXN + k] = C.nodes[off + k]; dst[b * MAXN + k] = C.dst[off + k]; tslot[b * MAXN + k] = C.tslot[off + k];
                }
                std::copy(&K.ck[(size_t)kci * CE], &K.ck[(size_t)kci * CE] + CE, &ck[b * CE]);
                std::copy(&K.sgn[(size_t)kci * CE], &K.sgn[(size_t)kci * CE] + CE, &sgn[b * CE]);
                std::copy(&K.lid[(size_t)kci * 8 * CE], &K.lid[(size_t)kci * 8 * CE] + 8 * CE, &lid[b * 8 * CE]);
            }
            MixTab T{};
            T.hdr = (const int4 *)up(hdr); T.nodes = (const int *)up(nodes); T.dst = (const int *)up(dst);
            T.tslot = (const unsigned short *)up(tslot); T.lid = (const unsigned short *)up(lid); T.ck = (const double *)up(ck);
            T.sgn = (const unsigned *)up(sgn);
            T.tinfo = (const int2 *)d_tinfo; T.tlid = (const unsigned short *)d_tlid; T.tck = (const double *)d_tck;
            T.tsgw = (const unsigned *)d_tsgw; T.tcol = (const unsigned char *)d_tcol; T.tperm = (const unsigned char *)d_tperm; T.frag = (const double *)d_frag;
            T.hrec = (const uint2 *)d_hrec; T.twait = (const int *)d_twait;
            T.np = M.nnpt; T.xcd = ebe_xcd_;
            T.flags = 0;
            if (const char *e = getenv("PCG_EBE_MIX_FLAGS")) T.flags = atoi(e);      // bit 0: barriers instead of tickets; 16 / 32 / 64: ablations (development)
            mix_tab_[ph] = T;
            mix_nodes_host_[ph] = nodes;
            mix_tslot_host_[ph] = tslot;
        }
    }
    template <int MTM>
    void launch_mixed(int ph, int count, const double *ke, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        if (mix_hex_tiles_) {                                     // every element on the matrix cores (k_ebe_mtile)
            auto gom = [&](auto kern, int nw) { hipLaunchKernelGGL(kern, dim3(count), dim3(64 * nw), 0, ls_, mix_tab_[ph], x, y, d_ch_buf_, part, dot_lo); };
            if constexpr (MTM == 4) {
                if (dot && stamp_launch_ >= 0 && count >= 64 && ++stamp_launch_ == 40) { launch_mtile_stamped(ph, count, x, y, part, dot_lo); return; }
                // waves per workgroup (round 6, PCG_EBE_MTILE_WAVES): 5 / 6 shorten a chunk's chain where the chunks do not fill the GPU
                if (mtile_waves_ == 5) { if (dot) gom(k_ebe_mtile<MTM, true, false, 5>, 5); else gom(k_ebe_mtile<MTM, false, false, 5>, 5); return; }
                if (mtile_waves_ == 6) { if (dot) gom(k_ebe_mtile<MTM, true, false, 6>, 6); else gom(k_ebe_mtile<MTM, false, false, 6>, 6); return; }
            }
            if (dot) gom(k_ebe_mtile<MTM, true>, kWavesPerBlock); else gom(k_ebe_mtile<MTM, false>, kWavesPerBlock);
            return;
        }
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(count), dim3(kChunkThreads), 0, ls_, mix_tab_[ph], ke, x, y, d_ch_buf_, part, dot_lo); };
        if constexpr (MTM == 4) {
            if (dot && stamp_launch_ >= 0 && count >= 64 && ++stamp_launch_ == 40) { launch_mixed_stamped(ph, count, ke, x, y, part, dot_lo); return; }
        }
        if (dot) go(k_ebe_mixed<MTM, true>); else go(k_ebe_mixed<MTM, false>);
    }
    // Development (PCG_EBE_STAMPS=1): the 40th fused-dot launch of the mixed kernel runs the STAMP instantiation - every wave writes the
    // shader clock at its phase boundaries - and the means over the waves go to stderr (cycles).  Same results, one slower launch.
    int stamp_launch_ = getenv("PCG_EBE_STAMPS") && atoi(getenv("PCG_EBE_STAMPS")) ? 0 : -1;
    bool mix_hex_tiles_ = false;
    int mtile_waves_ = getenv("PCG_EBE_MTILE_WAVES") ? atoi(getenv("PCG_EBE_MTILE_WAVES")) : kWavesPerBlock;
    void launch_mtile_stamped(int ph, int count, const double *x, double *y, double *part, long long dot_lo)
    {
        const int NWs = mtile_waves_ == 5 || mtile_waves_ == 6 ? mtile_waves_ : kWavesPerBlock;
        const size_t nw = (size_t)count * NWs;
        unsigned long long *d = nullptr;
        HIP_CHECK(hipMalloc((void **)&d, nw * 16 * sizeof(unsigned long long)));
        HIP_CHECK(hipMemsetAsync(d, 0, nw * 16 * sizeof(unsigned long long), ls_));
        MixTab T = mix_tab_[ph];
        T.stamps = d;
        if (NWs == 5) hipLaunchKernelGGL((k_ebe_mtile<4, true, true, 5>), dim3(count), dim3(320), 0, ls_, T, x, y, d_ch_buf_, part, dot_lo);
        else if (NWs == 6) hipLaunchKernelGGL((k_ebe_mtile<4, true, true, 6>), dim3(count), dim3(384), 0, ls_, T, x, y, d_ch_buf_, part, dot_lo);
        else hipLaunchKernelGGL((k_ebe_mtile<4, true, true>), dim3(count), dim3(kChunkThreads), 0, ls_, T, x, y, d_ch_buf_, part, dot_lo);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ls_));
        std::vector<unsigned long long> h(nw * 16);
        HIP_CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(d);
        // stamps 0, 1, 8, 9, 10, 11 = entry, staged, hex tiles done, other tiles done, final barrier passed, exit; 2..5 / 12..15 = per-tile sums
        double ph_sum[5] = {0}, life = 0, hs[4] = {0}, gs[4] = {0};
        for (size_t w = 0; w < nw; ++w) {
            const unsigned long long *s = &h[w * 16];
            const int idx[6] = {0, 1, 8, 9, 10, 11};
            for (int k = 1; k < 6; ++k) ph_sum[k - 1] += (double)(s[idx[k]] - s[idx[k - 1]]);
            life += (double)(s[11] - s[0]);
            for (int k = 0; k < 4; ++k) { hs[k] += (double)s[2 + k]; gs[k] += (double)s[12 + k]; }
        }
        fprintf(stderr, "[pcg] k_ebe_mtile stamps, phase %d: %d workgroups, mean wave lifetime %.0f cycles: stage %.0f, hex tiles %.0f, other tiles %.0f, final barrier %.0f, "
                        "write-out + dot %.0f\n", ph, count, life / nw, ph_sum[0] / nw, ph_sum[1] / nw, ph_sum[2] / nw, ph_sum[3] / nw, ph_sum[4] / nw);
        if (hs[0] > 0) fprintf(stderr, "[pcg]   hex tiles: %.2f per wave; per tile: %.0f to the end of the contraction, %.0f waiting for its colour, %.0f adds\n", hs[0] / nw,
                               hs[1] / hs[0], hs[2] / hs[0], hs[3] / hs[0]);
        if (gs[0] > 0) fprintf(stderr, "[pcg]   other tiles: %.2f per wave; per tile: %.0f to the end of the contraction, %.0f waiting for its turn, %.0f adds\n", gs[0] / nw,
                               gs[1] / gs[0], gs[2] / gs[0], gs[3] / gs[0]);
    }
    void launch_mixed_stamped(int ph, int count, const double *ke, const double *x, double *y, double *part, long long dot_lo)
    {
        const size_t nw = (size_t)count * kWavesPerBlock;
        unsigned long long *d = nullptr;
        HIP_CHECK(hipMalloc((void **)&d, nw * 16 * sizeof(unsigned long long)));
        HIP_CHECK(hipMemsetAsync(d, 0, nw * 16 * sizeof(unsigned long long), ls_));
        MixTab T = mix_tab_[ph];
        T.stamps = d;
        hipLaunchKernelGGL((k_ebe_mixed<4, true, true>), dim3(count), dim3(kChunkThreads), 0, ls_, T, ke, x, y, d_ch_buf_, part, dot_lo);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ls_));
        std::vector<unsigned long long> h(nw * 16);
        HIP_CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(d);
        // phase k = stamp[k] - stamp[k - 1] where both were written (a wave without a second hex pass has no stamps 5..7)
        static const char *name[12] = {"", "stage (tables, x gather, barrier)", "hex pass 0: contraction", "hex pass 0: wait for the ticket", "hex pass 0: adds",
                                       "hex pass 1: contraction (incl. slot loads)", "hex pass 1: wait for the ticket", "hex pass 1: adds", "-", "tile phase",
                                       "final barrier", "write-out + dot"};
        double sum[12] = {0}; long long cnt[12] = {0};
        double life = 0, tiles = 0, tc = 0, tw = 0, ta = 0;
        unsigned long long t_min = ~0ull, t_max = 0;
        for (size_t w = 0; w < nw; ++w) {
            const unsigned long long *s = &h[w * 16];
            unsigned long long prev = s[0];
            for (int k = 1; k < 12; ++k) {
                if (k == 8) { if (s[8]) prev = s[8]; continue; }
                if (!s[k]) continue;
                sum[k] += (double)(s[k] - prev); cnt[k]++;
                prev = s[k];
            }
            life += (double)(s[11] - s[0]);
            t_min = std::min(t_min, s[0]); t_max = std::max(t_max, s[11]);
            tiles += (double)s[12]; tc += (double)s[13]; tw += (double)s[14]; ta += (double)s[15];
        }
        fprintf(stderr, "[pcg] k_ebe_mixed stamps, phase %d: %d workgroups, kernel %.0f clock ticks first entry to last exit, mean wave lifetime %.0f\n", ph, count,
                (double)(t_max - t_min), life / nw);
        for (int k = 1; k < 12; ++k)
            if (cnt[k]) fprintf(stderr, "[pcg]   %-44s %9.0f per wave that ran it (%lld of %zu waves), %9.0f averaged over all waves\n", name[k], sum[k] / cnt[k], cnt[k], nw,
                                sum[k] / nw);
        if (tiles > 0)
            fprintf(stderr, "[pcg]   tiles: %.2f per wave; per tile: %.0f header -> end of contraction, %.0f waiting for the ticket, %.0f adds\n", tiles / nw, tc / tiles,
                    tw / tiles, ta / tiles);
    }
    // k_ebe_hex: the hex8 class laid out per launch (phase): block b's data at fixed strides of b
    void build_hex_tables(const EbeChunkedHost &C)
    {
        const auto &K = C.cls[0];
        hex_ept_ = K.ept;
        hex_npt_ = (K.max_nodes + kChunkThreads - 1) / kChunkThreads;
        const int CE = kChunkThreads * K.ept, MAXN = kChunkThreads * hex_npt_;
        auto up = [&](const auto &v) {
            void *d = alloc(sizeof(v[0]) * std::max<size_t>(1, v.size()));
            h2d(d, v.data(), sizeof(v[0]) * v.size());
            hex_allocs_.push_back(d);
            return d;
        };
        for (int ph = 0; ph < 2; ++ph) {
            const size_t n_real = K.list[ph].size();
            if (!n_real) continue;
            const size_t n = n_real;
            std::vector<int> hdr(4 * n, 0), nodes(n * MAXN, -1), dst(n * MAXN, 0);
            std::vector<unsigned short> tslot(n * MAXN, 0), lid(n * 8 * CE, 0);
            std::vector<double> ck(n * CE, 0.0);
            std::vector<unsigned> sgn(n * CE, 0xff000000u);
            for (size_t b = 0; b < n_real; ++b) {
                const int32_t *h = &C.hdr[(size_t)K.list[ph][b] * 8];
                const int32_t off = h[0], nn = h[1], kci = h[4];
                hdr[4 * b] = nn; hdr[4 * b + 1] = h[2]; hdr[4 * b + 2] = h[3];
                int any_sign = 0;
                for (int e = 0; e < CE; ++e) any_sign |= (K.sgn[(size_t)kci * CE + e] & 0x00ffffffu) != 0;
                hdr[4 * b + 3] = any_sign;
                for (int k = 0; k < nn; ++k) {
                    nodes[b * MAXN + k] = C.nodes[off + k]; dst[b * MAXN + k] = C.dst[off + k]; tslot[b * MAXN + k] = C.tslot[off + k];
                }
                std::copy(&K.ck[(size_t)kci * CE], &K.ck[(size_t)kci * CE] + CE, &ck[b * CE]);
                std::copy(&K.sgn[(size_t)kci * CE], &K.sgn[(size_t)kci * CE] + CE, &sgn[b * CE]);
                std::copy(&K.lid[(size_t)kci * 8 * CE], &K.lid[(size_t)kci * 8 * CE] + 8 * CE, &lid[b * 8 * CE]);
            }
            hex_tab_[ph] = HexTab{(const int4 *)up(hdr), (const int *)up(nodes), (const int *)up(dst), (const unsigned short *)up(tslot),
                                  (const unsigned short *)up(lid), (const double *)up(ck), (const unsigned *)up(sgn), ebe_xcd_,
                                  getenv("PCG_EBE_HEX_FLAGS") ? atoi(getenv("PCG_EBE_HEX_FLAGS")) : 0};
            hex_nodes_host_[ph] = nodes;
            hex_tslot_host_[ph] = tslot;
        }
    }
    template <int EPT, int NPT, int LB, int ACCM>
    void launch_hex_a(int ph, int count, const double *ke, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        if (dot)
            hipLaunchKernelGGL((k_ebe_hex<EPT, NPT, LB, true, ACCM>), dim3(count), dim3(kChunkThreads), 0, st_, hex_tab_[ph], ke, x, y,
                               d_ch_buf_, d_flags_, part, dot_lo);
        else
            hipLaunchKernelGGL((k_ebe_hex<EPT, NPT, LB, false, ACCM>), dim3(count), dim3(kChunkThreads), 0, st_, hex_tab_[ph], ke, x, y,
                               d_ch_buf_, d_flags_, part, dot_lo);
    }
    template <int LB>
    void launch_hexs(int ph, int count, const double *ke, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(count), dim3(kChunkThreads), 0, ls_, hex_tab_[ph], ke, x, y, d_ch_buf_, d_flags_, part, dot_lo);
        };
        if (dot) go(k_ebe_hexs<LB, true, 1>); else go(k_ebe_hexs<LB, false, 1>);
    }
    template <int EPT, int NPT, int LB>
    void launch_hex(int ph, int count, const double *ke, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        launch_hex_a<EPT, NPT, LB, 1>(ph, count, ke, x, y, dot, part, dot_lo);
    }
    void ebe_launch_range(const EbeRange &r, const double *x, double *y)
    {
        const auto &D = ebe_groups_[r.group];
        const int grid = (int)((r.hi - r.lo + kBlock - 1) / kBlock);
        if (D.nd == 24)
            hipLaunchKernelGGL((k_ebe<24>), dim3(grid), dim3(kBlock), 0, st_, D.dof, D.sgn_bits, D.ck, D.ke, x, y, D.ne, r.lo, r.hi);
        else
            hipLaunchKernelGGL(k_ebe_generic, dim3(grid), dim3(kBlock), 0, st_, D.dof, D.sgn_bytes, D.ck, D.ke, x, y, D.nd, D.ne,
                               r.lo, r.hi);
    }
    // hanging-node classes: Ke through LDS (k_ebe_rows KLDS) when it does not fit the scalar cache AND the launch is latency-bound
    // (at most four workgroups per CU).  Measured on the octree mesh (profiles/r03_octree_rows_kernel_lds_ab_sessionI.log): 1 M dof
    // (a few hundred chunks per class) operator 106 -> 91 us, iteration 131 -> 115 us; 10 M dof (thousands of chunks: throughput-
    // bound, and every workgroup copies 18-41 KB of Ke for its 64 elements) 509 -> 568 us.  PCG_EBE_ROWS_LDS=0 / 1 overrides.
    int rows_lds_mode_ = -1, ebe_xcd_ = 64;     // PCG_EBE_XCD: 0 off, 1 contiguous eighths, G runs of G chunks (measured: 64)
    bool rows_lds_raised_[4] = {false, false, false, false}, direct_raised_[4] = {false, false, false, false};
    template <int NNP>
    void launch_rows(const ChunkClassDev &D, int ph, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        constexpr int NDP = 3 * NNP;
        const bool klds = rows_lds_mode_ >= 0 ? rows_lds_mode_ != 0 : (NNP >= 16 && D.count[ph] <= 4 * n_cu_);
        auto go = [&](auto kern, size_t lds) {
            hipLaunchKernelGGL(kern, dim3(D.count[ph]), dim3(kChunkThreads), lds, ls_, D.list[ph], d_ch_hdr_, d_ch_nodes_, d_ch_dst_,
                               d_ch_tslot_, D.lid, D.ck, D.sgn, D.ke_rows, x, y, d_ch_buf_, d_flags_, part, dot_lo);
        };
        if (klds) {
            const size_t lds = sizeof(double) * NDP * NDP;
            const int slot = NNP == 8 ? 0 : NNP == 16 ? 1 : NNP == 24 ? 2 : 3;
            if (!rows_lds_raised_[slot]) {                       // beyond the default dynamic-LDS window (per device: a member)
                HIP_CHECK(hipFuncSetAttribute((const void *)k_ebe_rows<NNP, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                HIP_CHECK(hipFuncSetAttribute((const void *)k_ebe_rows<NNP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                rows_lds_raised_[slot] = true;
            }
            if (dot) go(k_ebe_rows<NNP, true, true>, lds); else go(k_ebe_rows<NNP, false, true>, lds);
        } else {
            if (dot) go(k_ebe_rows<NNP, true, false>, 0); else go(k_ebe_rows<NNP, false, false>, 0);
        }
    }
    // hanging-node classes without a node tile: Ke(NDP x NDP) . U(NDP x 64) on the matrix cores, Ke in LDS (k_ebe_direct)
    template <int NNP>
    void launch_direct(const ChunkClassDev &D, int ph, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        constexpr int NDP = 3 * NNP;
        const size_t lds = sizeof(double) * NDP * NDP;
        const int slot = NNP == 16 ? 1 : NNP == 24 ? 2 : 3;
        if (!direct_raised_[slot]) {                             // beyond the default dynamic-LDS window (per device: a member)
            HIP_CHECK(hipFuncSetAttribute((const void *)k_ebe_direct<NNP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_ebe_direct<NNP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            direct_raised_[slot] = true;
        }
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(D.count[ph]), dim3(kChunkThreads), lds, ls_, D.list[ph], d_ch_hdr_, d_ch_nodes_, d_ch_dst_, D.ck,
                               D.sgn, D.ke, x, y, d_ch_buf_, d_flags_, part, dot_lo, ebe_xcd_);
        };
        if (dot) go(k_ebe_direct<NNP, true>); else go(k_ebe_direct<NNP, false>);
    }
    // -> number of dot partials the launch writes
    int launch_class(const ChunkClassDev &D, int ph, const double *x, double *y, bool dot, double *part, long long dot_lo)
    {
        if (&D == &chc_[kMixedClass]) {                          // mixed-type chunks
            switch (mix_mtm_) {
            case 2: launch_mixed<2>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo); break;
            case 3: launch_mixed<3>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo); break;
            case 4: launch_mixed<4>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo); break;
            case 5: launch_mixed<5>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo); break;
            case 6: launch_mixed<6>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo); break;
            default: throw std::runtime_error("k_ebe_mixed: unexpected tile size");
            }
            return D.count[ph];
        }
        switch (D.nnp) {
        case 8:
            if (D.full && hex_tab_[ph].hdr) {                     // hex8 class through the per-launch tables
                if (hex_ept_ == 2 && hex_npt_ == 3) launch_hexs<4>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo);          // 512-element chunks, two passes
                else if (hex_ept_ == 1 && hex_npt_ == 2) launch_hex<1, 2, 5>(ph, D.count[ph], D.ke, x, y, dot, part, dot_lo);   // 256-element chunks
                else throw std::runtime_error("k_ebe_hex: unexpected chunk shape");
                break;
            }
            if (D.full) throw std::runtime_error("hex8 class without its launch tables");
            launch_rows<8>(D, ph, x, y, dot, part, dot_lo);                          // fewer than 8 nodes, padded
            break;
        case 16: if (D.direct) launch_direct<16>(D, ph, x, y, dot, part, dot_lo); else launch_rows<16>(D, ph, x, y, dot, part, dot_lo); break;
        case 24: if (D.direct) launch_direct<24>(D, ph, x, y, dot, part, dot_lo); else launch_rows<24>(D, ph, x, y, dot, part, dot_lo); break;
        default: if (D.direct) launch_direct<32>(D, ph, x, y, dot, part, dot_lo); else launch_rows<32>(D, ph, x, y, dot, part, dot_lo); break;
        }
        return D.count[ph];
    }
    bool ebe_apply(const double *x, double *y, int plo, int phi, bool zero_first, bool with_dot, int64_t dot_lo) override
    {
        // the fused dot needs every dof to be finalised by the chunk / shared kernels
        const bool fuse = with_dot && ebe_ranges_[0].empty() && ebe_ranges_[1].empty() && (n_chunks_total_[0] + n_chunks_total_[1]) > 0;
        const bool rec = prof_ && ev_used_ < kMaxEv;
        if (rec) HIP_CHECK(hipEventRecord(ev0_[ev_used_], st_));
        if (zero_first && ch_needs_zero_) HIP_CHECK(hipMemsetAsync(y, 0, sizeof(double) * (size_t)n_, st_));
        // split apply of the multi-part loop: ebe_apply(0, 1) .. pack, exchange begun .. ebe_apply(1, 2).  With phase streams the second
        // call runs on st_phase_, ordered behind everything that was on st_ BEFORE the first call's launches (x is complete, y zeroed).
        hipStream_t ks = st_;                                   // stream of this call's chunk and shared-node launches
        if (plo == 0 && phi == 1) {
            phase_forked_ = ebe_phase_streams_ != 0 && !prof_ && ebe_ranges_[0].empty() && ebe_ranges_[1].empty();
            if (phase_forked_) HIP_CHECK(hipEventRecord(ev_phase_fork_, st_));
        } else if (plo == 1 && phi == 2 && phase_forked_) {
            HIP_CHECK(hipStreamWaitEvent(st_phase_, ev_phase_fork_, 0));
            ks = st_phase_;
        } else {
            phase_forked_ = false;
        }
        for (int ph = plo; ph < phi; ++ph) {                    // chunked groups: one launch per phase + shared-node sums
            int others = 0;
            for (int c = 1; c < kChunkClasses; ++c) others += chc_[c].count[ph] > 0;
            const bool fork = others > 0 && (chc_[0].count[ph] > 0 || others > 1) &&
                              ebe_streams_mode_ == 1;
            if (fork) {                                          // every other class on its own stream beside the hex8 launch
                HIP_CHECK(hipEventRecord(ev_fork_, st_));
                for (int c = 1; c < kChunkClasses; ++c)
                    if (chc_[c].count[ph]) HIP_CHECK(hipStreamWaitEvent(st2_[c], ev_fork_, 0));
            }
            for (int c = 0; c < kChunkClasses; ++c) {            // one launch per node-count class
                const auto &D = chc_[c];
                if (!D.count[ph]) continue;
                ls_ = (fork && c > 0) ? st2_[c] : ks;
                const int np = launch_class(D, ph, x, y, fuse, d_part_ebe_ + cnt_ebe_, dot_lo);
                if (fuse) cnt_ebe_ += np;
            }
            ls_ = st_;
            if (fork)
                for (int c = 1; c < kChunkClasses; ++c)
                    if (chc_[c].count[ph]) {
                        HIP_CHECK(hipEventRecord(ev_join_[c], st2_[c]));
                        HIP_CHECK(hipStreamWaitEvent(ks, ev_join_[c], 0));
                    }
            if (sh_count_[ph]) {
                const int grid = (3 * sh_count_[ph] + kBlock - 1) / kBlock;
                double *part = d_part_ebe_ + cnt_ebe_;
                if (fuse)
                    hipLaunchKernelGGL((k_ebe_shared<true>), dim3(grid), dim3(kBlock), 0, ks, d_sh_node_[ph], d_sh_ptr_[ph],
                                       sh_slot0_[ph], d_ch_buf_, y, sh_count_[ph], x, d_flags_, part, (long long)dot_lo);
                else
                    hipLaunchKernelGGL((k_ebe_shared<false>), dim3(grid), dim3(kBlock), 0, ks, d_sh_node_[ph], d_sh_ptr_[ph],
                                       sh_slot0_[ph], d_ch_buf_, y, sh_count_[ph], x, d_flags_, part, (long long)dot_lo);
                if (fuse) cnt_ebe_ += grid;
            }
        }
        if (ks != st_) {                                        // the interior phase joins: whatever follows on st_ sees all of y
            HIP_CHECK(hipEventRecord(ev_phase_join_, ks));
            HIP_CHECK(hipStreamWaitEvent(st_, ev_phase_join_, 0));
            phase_forked_ = false;
        }
        for (int ph = plo; ph < phi; ++ph)                      // other pattern types: one launch per element colour
            for (const auto &r : ebe_ranges_[ph]) ebe_launch_range(r, x, y);
        HIP_CHECK(hipGetLastError());
        if (rec) { HIP_CHECK(hipEventRecord(ev1_[ev_used_], st_)); ++ev_used_; if (phi == 2) ++ev_applies_; }
        return fuse;
    }
    bool ebe_can_split() const override { return ebe_ranges_[0].empty() && ebe_ranges_[1].empty(); }
    void upload_masks(const uint8_t *f, int64_t n) override
    {
        h2d(d_flags_, f, (size_t)n);
        for (int ph = 0; ph < 2; ++ph) {                     // k_ebe_mixed: the same weights in its slot table
            if (!mix_tab_[ph].tslot) continue;
            std::vector<unsigned short> t(mix_tslot_host_[ph]);
            for (size_t k = 0; k < t.size(); ++k) {
                const int g = mix_nodes_host_[ph][k];
                if (g < 0) continue;
                unsigned m = 0;
                for (int d = 0; d < 3; ++d)
                    if ((f[3 * (size_t)g + d] & 3) == 3) m |= 1u << d;
                t[k] = (unsigned short)((t[k] & 0x3ff) | (m << 12));
            }
            h2d((void *)mix_tab_[ph].tslot, t.data(), sizeof(unsigned short) * t.size());
        }
        for (int ph = 0; ph < 2; ++ph) {                     // k_ebe_hex reads the dot weights from its slot table
            if (!hex_tab_[ph].tslot) continue;
            std::vector<unsigned short> t(hex_tslot_host_[ph]);
            for (size_t k = 0; k < t.size(); ++k) {
                const int g = hex_nodes_host_[ph][k];
                if (g < 0) continue;
                unsigned m = 0;
                for (int d = 0; d < 3; ++d)
                    if ((f[3 * (size_t)g + d] & 3) == 3) m |= 1u << d;
                t[k] = (unsigned short)((t[k] & 0x3ff) | (m << 12));
            }
            h2d((void *)hex_tab_[ph].tslot, t.data(), sizeof(unsigned short) * t.size());
        }
    }
    void upload_halo(const HaloHost &h) override
    {
        for (void *p : {(void *)d_send_idx_, (void *)d_fptr_, (void *)d_fpos_}) if (p) (void)hipFree(p);
        halo_count_ = (int64_t)h.send_idx.size();
        if (ebe_) nb_dofs_ = h.fix_dof.empty() ? 0 : (int64_t)h.fix_dof.back() + 1;
        d_send_idx_ = (int *)alloc(sizeof(int) * h.send_idx.size());
        h2d(d_send_idx_, h.send_idx.data(), sizeof(int) * h.send_idx.size());
        // dense CSR over all boundary-slice dofs
        std::vector<int> fptr((size_t)nb_dofs_ + 1, 0);
        for (size_t k = 0; k < h.fix_dof.size(); ++k) fptr[h.fix_dof[k] + 1] = (int)(h.fix_ptr[k + 1] - h.fix_ptr[k]);
        for (int64_t d = 0; d < nb_dofs_; ++d) fptr[d + 1] += fptr[d];
        d_fptr_ = (int *)alloc(sizeof(int) * fptr.size());
        h2d(d_fptr_, fptr.data(), sizeof(int) * fptr.size());
        d_fpos_ = (int *)alloc(sizeof(int) * std::max<size_t>(1, h.fix_pos.size()));
        h2d(d_fpos_, h.fix_pos.data(), sizeof(int) * h.fix_pos.size());
    }

    template <int RPL>
    void launch_spmv(const double *x, double *y, int64_t lo, int64_t hi, bool dot, int grid)
    {
        if (bs_ == 1) {
            if (dot)
                hipLaunchKernelGGL((k_spmv_scalar<true>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, d_cols_, d_vals_, x, y,
                                   d_flags_, d_part_spmv_, lo, hi, n_nodes_);
            else
                hipLaunchKernelGGL((k_spmv_scalar<false>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, d_cols_, d_vals_, x, y,
                                   d_flags_, d_part_spmv_, lo, hi, n_nodes_);
            return;
        }
        if (d_bidx_) {
            if (d_cols16_) launch_spmv_dict<true>(x, y, lo, hi, dot, grid, d_cols16_);
            else launch_spmv_dict<false>(x, y, lo, hi, dot, grid, d_cols_);
            return;
        }
        if (d_cols16_) launch_spmv_c<RPL, true>(x, y, lo, hi, dot, grid, d_cols16_);
        else launch_spmv_c<RPL, false>(x, y, lo, hi, dot, grid, d_cols_);
    }
    // value dictionary (k_spmv_dict): table in LDS when it fits kDictLdsBytes, else read through the caches
    static constexpr size_t kDictLdsBytes = 60 * 1024;
    unsigned short *d_bidx_ = nullptr;
    double *d_dict_ = nullptr;
    int n_unique_ = 0;
    bool dict_lds_ = false;
    // Workgroup size of the dictionary kernel: every workgroup holds one copy of the table (head) in LDS, so larger workgroups
    // put more waves behind one copy.  0 = automatic: 256 threads while four copies fit next to each other on a CU
    // (tables up to ~40 KB), else 512; tables beyond kDictLdsBytes: 1024 threads, one workgroup per CU.
    // PCG_SPMV_DICT_BLOCK overrides (256 / 512 / 640 / 1024) for tables within kDictLdsBytes.
    int dict_block_ = 0;
    static constexpr size_t kDictLdsBytesMax = 152 * 1024;   // of the CU's 160 KB (1945 entries)
    int n_lds_ = 0;                                           // entries of the LDS copy (all of them unless dict_mixed_)
    bool dict_mixed_ = false, dict_big_ = false;
    bool dict_packed_ = false;                                // 16-bit columns: column offset + table index in one word, four words per load
    int64_t *d_ptr4_ = nullptr;                               // ... groups of four block columns before each slice
    int64_t dict_lds_entries() const override { return n_lds_; }
    template <bool COL16>
    void launch_spmv_dict(const double *x, double *y, int64_t lo, int64_t hi, bool dot, int grid, const void *cols)
    {
        const size_t lds = dict_lds_ ? (size_t)n_lds_ * 80 : 0;            // entries padded to 80 B in LDS
        const int64_t fit = lds ? std::max<int64_t>(1, (int64_t)((160 * 1024) / (lds + 128))) : 8;     // copies per CU (160 KB LDS)
        // 256 threads while four or more copies fit on a CU, else 512 (two workgroups = 16 waves per CU).  640 threads (two
        // workgroups = 20 waves = the 5 per SIMD that 91 VGPRs allow) measured SLOWER at 10 M dof: 242 vs 188 us in the loop
        // (profiles/r03_dict_block_640_vs_512.log) - selectable with PCG_SPMV_DICT_BLOCK=640, not the default.
        int blk = dict_big_ ? 1024 : (dict_block_ ? dict_block_ : (fit >= 4 ? 256 : 512));
        // workgroups per CU: what LDS leaves room for, and at most 20 waves per CU in flight (<= 96 VGPRs: 5 per SIMD)
        const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(fit, 1280 / blk));
        const int wpb = blk / 64;
        int64_t g = std::min<int64_t>((hi - lo + wpb - 1) / wpb, std::min<int64_t>((int64_t)n_cu_ * per_cu, kMaxPartials));
        g = std::max<int64_t>(8, (g + 7) / 8 * 8);
        grid = (int)g;
#define PCG_LAUNCH_DICT(D, L, B, M)                                                                                               \
        hipLaunchKernelGGL((k_spmv_dict<D, COL16, L, B, M>), dim3(grid), dim3(B), lds, st_, d_slice_ptr_, cols, d_colbase_, d_ptr4_, \
                           d_bidx_, d_dict_, n_lds_, x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_)
#define PCG_LAUNCH_DICT_B(B, M)                                                                                                   \
        do {                                                                                                                      \
            if (dot) { if (lds) PCG_LAUNCH_DICT(true, true, B, M); else PCG_LAUNCH_DICT(true, false, B, false); }                 \
            else { if (lds) PCG_LAUNCH_DICT(false, true, B, M); else PCG_LAUNCH_DICT(false, false, B, false); }                   \
        } while (0)
        if (blk == 1024) { if (dict_mixed_) PCG_LAUNCH_DICT_B(1024, true); else PCG_LAUNCH_DICT_B(1024, false); }
        else if (blk == 640) PCG_LAUNCH_DICT_B(640, false);
        else if (blk == 512) PCG_LAUNCH_DICT_B(512, false);
        else PCG_LAUNCH_DICT_B(256, false);
#undef PCG_LAUNCH_DICT_B
#undef PCG_LAUNCH_DICT
        last_spmv_grid_ = grid;
    }
    int last_spmv_grid_ = 0;
    double *pack_send_ = nullptr;                 // set by spmv() for the launch it is about to make: halo_pack folded into the epilogue
    template <int RPL, bool COL16>
    void launch_spmv_c(const double *x, double *y, int64_t lo, int64_t hi, bool dot, int grid, const void *cols)
    {
        const PackArgs pk{d_fptr_, d_fpos_, pack_send_};
        const int fl = xcd_aware_ | (vec_nt_ & 2);
        // Sub-ranges (kernels_spmv.hpp HOLD): the slice range is walked in `parts` pieces a wave's share of which fits the kernel's LDS slots -
        // in ONE launch (part = -1: stores as they come) or in `parts` launches that write their y at their end (spmv_split_); the same
        // bits either way.  Interface rows with the pack epilogue and 128-row slices: one piece.
        int parts = 1;
        if (RPL == 1 && !(pack_send_ && !dot)) {
            const int64_t waves = (int64_t)grid * kWavesPerBlock, per_wave = (hi - lo + waves - 1) / waves;
            parts = (int)std::max<int64_t>(1, (per_wave + kSpmvHold - 1) / kSpmvHold);
        }
        spmv_launches_ = 1;
        if (RPL == 1 && parts > 1 && spmv_split_) {
            for (int q = 0; q < parts; ++q) {
                if (dot)
                    hipLaunchKernelGGL((k_spmv<1, true, COL16, false, true>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, cols, d_colbase_, d_vals_,
                                       x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_, fl, d_ov_mask_, pk, parts, q);
                else
                    hipLaunchKernelGGL((k_spmv<1, false, COL16, false, true>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, cols, d_colbase_, d_vals_,
                                       x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_, fl, d_ov_mask_, pk, parts, q);
            }
            spmv_launches_ = parts;
            return;
        }
        if (dot)
            hipLaunchKernelGGL((k_spmv<RPL, true, COL16>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, cols, d_colbase_, d_vals_,
                               x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_, fl, d_ov_mask_, pk, parts, -1);
        else if (pack_send_ && RPL == 1)
            hipLaunchKernelGGL((k_spmv<1, false, COL16, true>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, cols, d_colbase_, d_vals_,
                               x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_, fl, d_ov_mask_, pk, parts, -1);
        else
            hipLaunchKernelGGL((k_spmv<RPL, false, COL16>), dim3(grid), dim3(kBlock), 0, st_, d_slice_ptr_, cols, d_colbase_, d_vals_,
                               x, y, d_flags_, d_part_spmv_, lo, hi, n_nodes_, fl, d_ov_mask_, pk, parts, -1);
    }
    // overflow part of a split matrix (k_spmv_ovf) for the base slices [lo, hi): its partials follow those of the base launch
    int launch_overflow(const double *x, double *y, int64_t lo, int64_t hi, bool dot, int part_off)
    {
        if (ov_slices_ == 0) return 0;
        int64_t olo, ohi;
        if (lo == 0) olo = 0; else if (lo == n_bnd_slices_) olo = ov_bnd_slices_; else throw std::runtime_error("spmv: slice range does not match the overflow part");
        if (hi == n_slices_) ohi = ov_slices_; else if (hi == n_bnd_slices_) ohi = ov_bnd_slices_; else throw std::runtime_error("spmv: slice range does not match the overflow part");
        if (ohi <= olo) return 0;
        const int grid = spmv_grid(ohi - olo);
        if (part_off + grid > kMaxPartials) throw std::runtime_error("spmv: too many dot partials");
        if (dot)
            hipLaunchKernelGGL((k_spmv_ovf<true>), dim3(grid), dim3(kBlock), 0, st_, d_ov_slice_ptr_, d_ov_rows_, d_ov_cols_, d_ov_vals_, x, y,
                               d_flags_, d_part_spmv_ + part_off, olo, ohi);
        else
            hipLaunchKernelGGL((k_spmv_ovf<false>), dim3(grid), dim3(kBlock), 0, st_, d_ov_slice_ptr_, d_ov_rows_, d_ov_cols_, d_ov_vals_, x, y,
                               d_flags_, d_part_spmv_ + part_off, olo, ohi);
        return grid;
    }
    int64_t ov_slices_ = 0, ov_bnd_slices_ = 0;
    int64_t n_windows_ = 0, n_bnd_windows_ = 0;                  // windowed form of the split matrix (SellHost::win_*)
    int64_t *d_win_slice_ = nullptr, *d_win_ov_ = nullptr;
    // the whole split operator in ONE launch; -> workgroups launched (= dot partials written)
    int launch_windowed(const double *x, double *y, int64_t lo, int64_t hi, bool dot)
    {
        int64_t wlo, whi;
        if (lo == 0) wlo = 0; else if (lo == n_bnd_slices_) wlo = n_bnd_windows_; else throw std::runtime_error("spmv: slice range does not match the windows");
        if (hi == n_slices_) whi = n_windows_; else if (hi == n_bnd_slices_) whi = n_bnd_windows_; else throw std::runtime_error("spmv: slice range does not match the windows");
        int64_t g = std::min<int64_t>(whi - wlo, std::min<int64_t>((int64_t)n_cu_ * spmv_blocks_per_cu_, kMaxPartials));
        const int grid = (int)std::max<int64_t>(1, g);
        const PackArgs pk{d_fptr_, d_fpos_, pack_send_};
        auto go = [&](auto kern, const void *cols) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, st_, d_win_slice_, d_win_ov_, d_slice_ptr_, cols, d_colbase_, d_vals_, d_ov_mask_,
                               d_ov_slice_ptr_, d_ov_rows_, d_ov_cols_, d_ov_vals_, x, y, d_flags_, d_part_spmv_, wlo, whi, n_nodes_, pk);
        };
        const bool pack = pack_send_ != nullptr && !dot;
        if (d_cols16_) { if (dot) go(k_spmv_win<true, true>, d_cols16_); else if (pack) go(k_spmv_win<false, true, true>, d_cols16_); else go(k_spmv_win<false, true>, d_cols16_); }
        else { if (dot) go(k_spmv_win<true, false>, d_cols_); else if (pack) go(k_spmv_win<false, false, true>, d_cols_); else go(k_spmv_win<false, false>, d_cols_); }
        return grid;
    }
    int64_t *d_ov_slice_ptr_ = nullptr;
    int *d_ov_rows_ = nullptr, *d_ov_cols_ = nullptr;
    double *d_ov_vals_ = nullptr;
    unsigned long long *d_ov_mask_ = nullptr;
    int col_index_bytes() const override { return d_cols16_ ? 2 : 4; }
    void spmv(const double *x, double *y, int64_t lo, int64_t hi, bool with_dot, double *pack_send) override
    {
        if (hi <= lo) { if (with_dot) cnt_spmv_ = 0; if (pack_send) halo_pack(y, pack_send); return; }
        // the pack rides on the launch when the rows are final in it: 3x3-block rows of 64-row slices, plain values, no dot, and no
        // second (overflow) launch behind it - the windowed form packs in both of its phases
        const bool can_pack = pack_send && !with_dot && halo_count_ > 0 && bs_ == 3 && C_ == 64 && !d_bidx_ && (ov_slices_ == 0 || n_windows_ > 0);
        pack_send_ = can_pack ? pack_send : nullptr;
        const int grid = spmv_grid(hi - lo);
        const bool rec = prof_ && ev_used_ < kMaxEv;
        if (rec) HIP_CHECK(hipEventRecord(ev0_[ev_used_], st_));
        last_spmv_grid_ = grid;
        int ov_grid = 0;
        if (n_windows_ > 0) {
            last_spmv_grid_ = launch_windowed(x, y, lo, hi, with_dot);
        } else {
            if (C_ == 64) launch_spmv<1>(x, y, lo, hi, with_dot, grid);
            else launch_spmv<2>(x, y, lo, hi, with_dot, grid);
            ov_grid = launch_overflow(x, y, lo, hi, with_dot, last_spmv_grid_);
        }
        HIP_CHECK(hipGetLastError());
        if (rec) { HIP_CHECK(hipEventRecord(ev1_[ev_used_], st_)); ++ev_used_; if (hi == n_slices_) ++ev_applies_; }
        if (with_dot) cnt_spmv_ = last_spmv_grid_ + ov_grid;   // (the dictionary kernel may have launched a smaller grid)
        pack_send_ = nullptr;
        if (pack_send && !can_pack) halo_pack(y, pack_send);   // formats the epilogue does not cover: a launch of its own
    }
    void halo_pack(const double *y, double *send) override
    {
        if (!halo_count_) return;
        int grid = (int)std::min<int64_t>((halo_count_ + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_halo_pack, dim3(grid), dim3(kBlock), 0, st_, y, d_send_idx_, send, halo_count_);
        HIP_CHECK(hipGetLastError());
    }
    // last-workgroup reductions of the multi-part loop (k_fixup<true, true>, k_vec<false> with reduce_last): arrival counters, put
    // back to 0 by the last arriver of every launch
    unsigned long long *d_last_cnt_ = nullptr;      // [0]: k_fixup, [16]: k_vec, [32]: k_halo_put (128 B apart)
    void last_counters()
    {
        if (d_last_cnt_) return;
        d_last_cnt_ = (unsigned long long *)alloc(sizeof(unsigned long long) * 48);
        HIP_CHECK(hipMemsetAsync(d_last_cnt_, 0, sizeof(unsigned long long) * 48, st_));
    }
    bool mailbox_kernels_available() const override { return true; }
    bool direct_kernels_available() const override { return true; }
    void halo_put(const double *y, const DirectDesc &d) override
    {
        if (!halo_count_) return;
        last_counters();
        int grid = (int)std::min<int64_t>((halo_count_ + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(k_halo_put, dim3(grid), dim3(kBlock), 0, st_, y, d_send_idx_, halo_count_, d, d_last_cnt_ + 32);
        HIP_CHECK(hipGetLastError());
    }
    void boundary_fixup(double *y, const double *recv, const double *xdot, bool with_dot, double *reduce_pq, const MailDesc *mail,
                        const DirectDesc *direct) override
    {
        if (!nb_dofs_) {
            if (mail) throw std::runtime_error("boundary_fixup: a part without interface dofs cannot carry the mailbox all-reduce");
            if (with_dot) { cnt_fix_ = 0; if (reduce_pq) reduce_dot(reduce_pq); }
            return;
        }
        int grid = (int)std::min<int64_t>((nb_dofs_ + kBlock - 1) / kBlock, 1024);
        const FixWait fw = fix_wait_of(direct);
        FixReduce fr{};
        if (with_dot && reduce_pq) {
            last_counters();
            fr.pa = ebe_ ? d_part_ebe_ : d_part_spmv_; fr.count_a = ebe_ ? cnt_ebe_ : cnt_spmv_;
            fr.red = reduce_pq; fr.counter = d_last_cnt_;
            if (mail) {
                fr.mail = *mail;
                hipLaunchKernelGGL((k_fixup<true, true, true>), dim3(grid), dim3(kBlock), 0, st_, y, recv, d_fptr_, d_fpos_, xdot, d_flags_,
                                   nb_dofs_, d_part_fix_, fr, fw);
            } else
                hipLaunchKernelGGL((k_fixup<true, true>), dim3(grid), dim3(kBlock), 0, st_, y, recv, d_fptr_, d_fpos_, xdot, d_flags_,
                                   nb_dofs_, d_part_fix_, fr, fw);
        } else if (mail)
            throw std::runtime_error("boundary_fixup: the mailbox all-reduce rides on the fused dot reduction only");
        else if (with_dot)
            hipLaunchKernelGGL((k_fixup<true>), dim3(grid), dim3(kBlock), 0, st_, y, recv, d_fptr_, d_fpos_, xdot, d_flags_,
                               nb_dofs_, d_part_fix_, fr, fw);
        else
            hipLaunchKernelGGL((k_fixup<false>), dim3(grid), dim3(kBlock), 0, st_, y, recv, d_fptr_, d_fpos_, xdot, d_flags_,
                               nb_dofs_, d_part_fix_, fr, fw);
        HIP_CHECK(hipGetLastError());
        if (with_dot) cnt_fix_ = grid;
    }
    void begin_dot() override { cnt_spmv_ = cnt_fix_ = cnt_ebe_ = 0; }
    void reduce_dot(double *red) override
    {
        if (ebe_)
            hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st_, d_part_ebe_, cnt_ebe_, 0, d_part_fix_, cnt_fix_, red,
                               mirror_of(red));
        else
            hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st_, d_part_spmv_, cnt_spmv_, kMaxPartials, d_part_fix_,
                               cnt_fix_, red, mirror_of(red));
        HIP_CHECK(hipGetLastError());
    }
    // ---- status ring in host-visible (pinned, mapped) memory: kStatusSlots copies of the status block ------------
    // The reduce / alpha kernels mirror every status word they write into the CURRENT slot, so the host reads an
    // iteration's sums without a device->host copy; one event per slot lets it wait for exactly that iteration
    // while the next one is already queued behind it.
    double *d_st_base_ = nullptr, *h_mirror_ = nullptr, *d_mirror_ = nullptr;
    int cur_slot_ = 0;
    hipEvent_t ev_slot_[kStatusSlots] = {};
    double *mirror_of(double *p) const
    {
        return (d_mirror_ && p >= d_st_base_ && p < d_st_base_ + ST_COUNT) ? d_mirror_ + (size_t)cur_slot_ * ST_COUNT + (p - d_st_base_)
                                                                            : nullptr;
    }
    void set_status_block(double *st) override
    {
        d_st_base_ = st;
        if (!h_mirror_) {
            HIP_CHECK(hipHostMalloc((void **)&h_mirror_, sizeof(double) * ST_COUNT * kStatusSlots, hipHostMallocMapped));
            for (int k = 0; k < ST_COUNT * kStatusSlots; ++k) h_mirror_[k] = 0.0;
            HIP_CHECK(hipHostGetDevicePointer((void **)&d_mirror_, h_mirror_, 0));
            for (auto &ev : ev_slot_) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
    }
    bool read_status(double *host_out) override
    {
        if (!h_mirror_) return false;
        HIP_CHECK(hipStreamSynchronize(st_));
        const volatile double *m = h_mirror_ + (size_t)cur_slot_ * ST_COUNT;
        for (int k = 0; k < ST_COUNT; ++k) host_out[k] = m[k];
        return true;
    }
    void set_status_slot(int slot) override { cur_slot_ = slot; }
    void reload_tuning() override
    {
        vec_nt_ = 5;
        if (const char *e = getenv("PCG_VEC_NT")) vec_nt_ = atoi(e);          // bit 0: p / r' / x' stores, bit 1: SpMV y stores, bit 2: vector loads
        vec_fused_ok_ = vec_fused_hw_ && !vec_fused_broken_;
        if (h_mirror_)                                   // (called at solve_begin, nothing in flight) no stale time-out report in the ring
            for (int s = 0; s < kStatusSlots; ++s) h_mirror_[(size_t)s * ST_COUNT + ST_ERR] = 0.0;
        vec_spin_limit_ = 1u << 22;
        if (const char *e = getenv("PCG_TEST_VEC_SPINS")) vec_spin_limit_ = (unsigned)std::max(0, atoi(e));   // tests: force the time-out path
        vec_inband_ = true;
        if (const char *e = getenv("PCG_VEC_INBAND")) vec_inband_ = atoi(e) != 0;      // grid barrier of the fused vector launch: in-band (round 5) / counters (A/B)
        vec_kreg_ = kVecKreg;
        if (const char *e = getenv("PCG_VEC_KREG")) vec_kreg_ = std::max(0, std::min(kVecKreg, atoi(e)));
        if (const char *e = getenv("PCG_VEC_FUSED")) vec_fused_ok_ = vec_fused_ok_ && atoi(e) != 0;
        ebe_phase_streams_ = 0;
        if (const char *e = getenv("PCG_EBE_PHASE_STREAMS")) ebe_phase_streams_ = atoi(e) != 0 ? 1 : 0;     // multi-part loop: interior phase beside the interface phase (A/B)
        iter_fused_ = true;
        if (const char *e = getenv("PCG_ITER_FUSED")) iter_fused_ = atoi(e) != 0;      // multi-part loop: pack / reductions / status copy folded in (A/B)
    }
    void publish_status(bool copy_block) override
    {
        if (copy_block) {
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st_, d_st_base_, d_mirror_ + (size_t)cur_slot_ * ST_COUNT);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipEventRecord(ev_slot_[cur_slot_], st_));
    }
    void wait_status(int slot, double *host_out) override
    {
        HIP_CHECK(hipEventSynchronize(ev_slot_[slot]));
        const volatile double *m = h_mirror_ + (size_t)slot * ST_COUNT;
        for (int k = 0; k < ST_COUNT; ++k) host_out[k] = m[k];
    }
    void update_p(double *po, const double *pi, const double *r, const double *minv, const double *st, double rho_prev,
                  bool first, int publish_slot) override
    {
        double *mirror = publish_slot >= 0 && d_mirror_ ? d_mirror_ + (size_t)publish_slot * ST_COUNT : nullptr;
        hipLaunchKernelGGL(k_update_p, dim3(vec_grid(n_)), dim3(kBlock), 0, st_, po, pi, r, minv, st, rho_prev, first ? 1 : 0, vec_nt_, n_, mirror);
        HIP_CHECK(hipGetLastError());
        if (publish_slot >= 0) HIP_CHECK(hipEventRecord(ev_slot_[publish_slot], st_));
    }
    bool iteration_fusion_available() const override { return iter_fused_; }
    bool iter_fused_ = true;
    // ---- vector phase (k_vec) --------------------------------------------------------------------------------------
    unsigned long long *d_vec_sync_ = nullptr;     // arrival counters of the fused form's grid barrier (monotonic)
    unsigned long long vec_seq_ = 0;               // fused launches so far (counter form of the grid barrier)
    double *d_vec_pub_ = nullptr;                  // in-band form of the grid barrier: the published sums, two parities (kernels_vector.hpp)
    unsigned long long vec_seq_inband_ = 0;        // ... its launches so far (parity of the slots)
    bool vec_inband_ = true;                       // PCG_VEC_INBAND=0: the counter form
    int vec_kreg_ = kVecKreg;
    bool vec_fused_hw_ = false, vec_fused_ok_ = false;   // the device admits the grid / and PCG_VEC_FUSED does not say 0
    bool vec_fused_broken_ = false;                      // a launch timed out at its grid barrier: split form for the rest of this engine's life
    unsigned vec_spin_limit_ = 1u << 22;
    void vec_fused_failed() override
    {
        vec_fused_broken_ = true;
        vec_fused_ok_ = false;
        HIP_CHECK(hipMemsetAsync(d_st_base_ + ST_ERR, 0, sizeof(double), st_));
        HIP_CHECK(hipStreamSynchronize(st_));            // nothing queued may still mirror its report
        if (h_mirror_)
            for (int s = 0; s < kStatusSlots; ++s) h_mirror_[(size_t)s * ST_COUNT + ST_ERR] = 0.0;
    }
    int vec_blocks(int64_t n) const
    {
        const int64_t g = ((n >> 1) + kVecBlock - 1) / kVecBlock;
        return (int)std::max<int64_t>(1, std::min<int64_t>(g, n_cu_));      // one workgroup per CU at most: co-resident
    }
    bool vec_fused_available() const override { return vec_fused_ok_; }
    bool vec_update(double *st, int pq_src, const double *p, const double *q, const double *r, double *rn, const double *xo,
                    double *xn, const double *minv, double *p_next, bool reduce_sums, const MailDesc *mail) override
    {
        if (mail && (p_next != nullptr || !reduce_sums)) throw std::runtime_error("vec_update: the mailbox all-reduce rides on the last-workgroup reduction only");
        const bool fused = p_next != nullptr;
        if (fused && !vec_fused_ok_) throw std::runtime_error("vec_update: the fused form is not available on this device");
        cnt_vec_ = vec_blocks(n_);
        VecArgs a{};
        a.st = st; a.mirror = mirror_of(st);
        a.p = p; a.q = q; a.r = r; a.rn = rn; a.xo = xo; a.xn = xn; a.minv = minv; a.flags = d_flags_;
        a.p_next = p_next; a.partials = d_part_;
        if (pq_src == 2) {
            if (ebe_) { a.pa = d_part_ebe_; a.count_a = cnt_ebe_; }
            else { a.pa = d_part_spmv_; a.count_a = cnt_spmv_; }
            a.pb = cnt_fix_ ? d_part_fix_ : nullptr; a.count_b = cnt_fix_;
        }
        a.sync = d_vec_sync_; a.pq_src = pq_src; a.nt = vec_nt_; a.n = n_; a.kreg = vec_kreg_; a.spin_limit = vec_spin_limit_;
        if (reduce_sums && !fused) {
            last_counters();
            a.reduce_last = 1; a.last_counter = d_last_cnt_ + 16;
            if (mail) { a.mail_on = 1; a.mail = *mail; }
        }
        const bool rec = prof_vec_ && evv_used_ < kMaxEv;
        if (rec) HIP_CHECK(hipEventRecord(evv0_[evv_used_], st_));
        if (fused) {
            a.pub = d_vec_pub_;
            a.inband = vec_inband_ && cnt_vec_ >= 5 && cnt_vec_ <= 256 ? 1 : 0;     // (each form keeps its own launch count: the counters are monotonic)
            a.seq = a.inband ? ++vec_seq_inband_ : ++vec_seq_;
            // small systems (every chunk of a thread fits the preloading form: <= kVecPre per thread): operands requested before the
            // alpha prologue, p kept in registers (kernels_vector.hpp); PCG_VEC_NT bit 3 / PCG_VEC_KREG < kVecPre: the general form
            const bool pre = vec_kreg_ >= kVecPre && (n_ >> 1) <= (int64_t)kVecPre * cnt_vec_ * kVecBlock && !(vec_nt_ & 8);
            if (pre) hipLaunchKernelGGL((k_vec<true, true>), dim3(cnt_vec_), dim3(kVecBlock), 0, st_, a);
            else hipLaunchKernelGGL((k_vec<true>), dim3(cnt_vec_), dim3(kVecBlock), 0, st_, a);
        } else {
            // the split form of the multi-part loop (round 5): small parts - a GPU's share of 10 M dof on 8 - preload their operands too:
            // the general form runs a thread's chunks one after the other, 2 - 3 dependent round trips at 1.3 M dof (27 us for 75 MB)
            const bool pre = vec_kreg_ >= kVecPre && (n_ >> 1) <= (int64_t)kVecPre * cnt_vec_ * kVecBlock && !(vec_nt_ & 8);
            if (pre) hipLaunchKernelGGL((k_vec<false, true>), dim3(cnt_vec_), dim3(kVecBlock), 0, st_, a);
            else hipLaunchKernelGGL((k_vec<false>), dim3(cnt_vec_), dim3(kVecBlock), 0, st_, a);
        }
        HIP_CHECK(hipGetLastError());
        if (rec) { HIP_CHECK(hipEventRecord(evv1_[evv_used_], st_)); ++evv_used_; }
        return fused;
    }
    void reduce_update(double *red5) override
    {
        hipLaunchKernelGGL(k_reduce, dim3(5), dim3(kBlock), 0, st_, d_part_, cnt_vec_, kMaxPartials, (const double *)nullptr, 0, red5,
                           mirror_of(red5));
        HIP_CHECK(hipGetLastError());
    }
    void residual(const double *b, const double *ax, double *r, const double *minv) override
    {
        cnt_vec_ = vec_grid(n_);
        hipLaunchKernelGGL(k_residual, dim3(cnt_vec_), dim3(kBlock), 0, st_, b, ax, r, minv, d_flags_, d_part_, n_);
        HIP_CHECK(hipGetLastError());
    }
    void reduce_residual(double *red3) override
    {
        hipLaunchKernelGGL(k_reduce, dim3(3), dim3(kBlock), 0, st_, d_part_, cnt_vec_, kMaxPartials, (const double *)nullptr, 0, red3,
                           mirror_of(red3));
        HIP_CHECK(hipGetLastError());
    }
    void dot_w(const double *a, const double *b) override
    {
        cnt_vec_ = vec_grid(n_);
        hipLaunchKernelGGL(k_dot_w, dim3(cnt_vec_), dim3(kBlock), 0, st_, a, b, d_flags_, d_part_, n_);
        HIP_CHECK(hipGetLastError());
    }
    void reduce_dotw(double *red1) override
    {
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, st_, d_part_, cnt_vec_, kMaxPartials, (const double *)nullptr, 0, red1,
                           mirror_of(red1));
        HIP_CHECK(hipGetLastError());
    }
    void copy_diag(double *d) override { d2d(d, d_diag_, sizeof(double) * (size_t)n_); }
    void invert_free(double *minv, const double *d) override
    {
        hipLaunchKernelGGL(k_invert_free, dim3(vec_grid(n_)), dim3(kBlock), 0, st_, minv, d, d_flags_, n_);
        HIP_CHECK(hipGetLastError());
    }
    void axpby(double *o, double a, const double *x, double b, const double *y) override
    {
        hipLaunchKernelGGL(k_axpby, dim3(vec_grid(n_)), dim3(kBlock), 0, st_, o, a, x, b, y, n_);
        HIP_CHECK(hipGetLastError());
    }
    void scale(double *o, double a, const double *x) override
    {
        hipLaunchKernelGGL(k_scale, dim3(vec_grid(n_)), dim3(kBlock), 0, st_, o, a, x, n_);
        HIP_CHECK(hipGetLastError());
    }
    void mask_free(double *x) override
    {
        hipLaunchKernelGGL(k_mask_free, dim3(vec_grid(n_)), dim3(kBlock), 0, st_, x, d_flags_, n_);
        HIP_CHECK(hipGetLastError());
    }
    void set_profiling(int what) override
    {
        const bool on = what != 0;
        prof_ = (what & 1) != 0; prof_vec_ = (what & 2) != 0;
        if (on && ev0_.empty()) {
            ev0_.resize(kMaxEv); ev1_.resize(kMaxEv); evv0_.resize(kMaxEv); evv1_.resize(kMaxEv);
            for (int k = 0; k < kMaxEv; ++k) {
                HIP_CHECK(hipEventCreate(&ev0_[k])); HIP_CHECK(hipEventCreate(&ev1_[k]));
                HIP_CHECK(hipEventCreate(&evv0_[k])); HIP_CHECK(hipEventCreate(&evv1_[k]));
            }
        }
        ev_used_ = 0; ev_applies_ = 0; evv_used_ = 0;
    }
    void collect_profile(double *ms_sum, int64_t *count) override
    {
        HIP_CHECK(hipStreamSynchronize(st_));
        double s = 0;
        for (int k = 0; k < ev_used_; ++k) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ev0_[k], ev1_[k])); s += ms; }
        *ms_sum = s; *count = ev_applies_;
    }
    void collect_profile_vec(double *ms_sum, int64_t *count) override
    {
        HIP_CHECK(hipStreamSynchronize(st_));
        double s = 0;
        for (int k = 0; k < evv_used_; ++k) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, evv0_[k], evv1_[k])); s += ms; }
        *ms_sum = s; *count = evv_used_;
    }
    int bench_hbm(size_t bytes, int mode, int reps, float *ms_each) override
    {
        const int64_t n2 = (int64_t)(bytes / 16);
        double2 *a = (double2 *)alloc((size_t)n2 * 16), *b = mode == 1 ? (double2 *)alloc((size_t)n2 * 16) : nullptr;
        double *out = (double *)alloc(8);
        HIP_CHECK(hipMemsetAsync(a, 0x3c, (size_t)n2 * 16, st_));          // finite non-zero doubles
        int grid = n_cu_ * (mode == 0 ? 32 : 4);       // measured best of {4, 8, 16, 32} blocks per CU for modes 0 and 1
        if (const char *e = getenv("PCG_STREAM_BLOCKS_PER_CU")) grid = n_cu_ * std::max(1, atoi(e));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        for (int k = -3; k < reps; ++k) {
            HIP_CHECK(hipEventRecord(e0, st_));
            if (mode == 1) hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(kBlock), 0, st_, a, b, n2);
            else hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(kBlock), 0, st_, a, out, n2);
            HIP_CHECK(hipEventRecord(e1, st_));
            HIP_CHECK(hipEventSynchronize(e1));
            if (k >= 0) HIP_CHECK(hipEventElapsedTime(&ms_each[k], e0, e1));
        }
        HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
        release(a); if (b) release(b); release(out);
        return 0;
    }
    int operator_launches_per_apply() const override
    {
        if (ebe_ || bs_ != 3 || C_ != 64 || n_windows_ > 0 || d_bidx_ || !spmv_split_) return 1;
        const int grid = spmv_grid(n_slices_);
        const int64_t waves = (int64_t)grid * kWavesPerBlock, per_wave = (n_slices_ + waves - 1) / waves;
        return (int)std::max<int64_t>(1, (per_wave + kSpmvHold - 1) / kSpmvHold);
    }
    int tune_operator(const double *x, double *y) override
    {
        if (ebe_ || bs_ != 3 || C_ != 64 || n_windows_ > 0 || d_bidx_ || spmv_split_forced_) return operator_launches_per_apply();
        spmv_split_ = true;
        const int parts = operator_launches_per_apply();
        spmv_split_ = false;
        if (parts <= 1) return 1;
        float ms[4];
        double t[2];
        for (int form = 0; form < 2; ++form) {               // 0: one launch, 1: `parts` launches with their y at the end
            spmv_split_ = form == 1;
            bench_spmv(x, y, 1, 4, ms);
            t[form] = std::min(std::min(ms[0], ms[1]), std::min(ms[2], ms[3]));
        }
        spmv_split_ = t[1] < 0.99 * t[0];                     // (the launches' own cost: only a clear win)
        if (getenv("PCG_VEC_PLACEMENT_LOG"))
            fprintf(stderr, "[pcg] k_spmv: one launch %.4f ms, %d launches with y at their end %.4f ms -> %s\n", t[0], parts, t[1], spmv_split_ ? "split" : "one launch");
        return operator_launches_per_apply();
    }
    int bench_spmv(const double *x, double *y, int warmup, int reps, float *ms_each) override
    {
        if (ebe_) {
            for (int k = 0; k < warmup; ++k) { cnt_ebe_ = 0; ebe_apply(x, y, 0, 2, true, bench_dot_, 0); }
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
            for (int k = 0; k < reps; ++k) {
                HIP_CHECK(hipEventRecord(a, st_));
                cnt_ebe_ = 0;
                ebe_apply(x, y, 0, 2, true, bench_dot_, 0);
                HIP_CHECK(hipEventRecord(b, st_));
                HIP_CHECK(hipEventSynchronize(b));
                HIP_CHECK(hipEventElapsedTime(&ms_each[k], a, b));
            }
            HIP_CHECK(hipEventDestroy(a)); HIP_CHECK(hipEventDestroy(b));
            return 0;
        }
        const int grid = spmv_grid(n_slices_);
        const bool dot = bench_dot_;
        const double *xs = x;
        auto apply_once = [&]() {                              // base part + overflow part, as spmv() launches them
            if (n_windows_ > 0) { (void)launch_windowed(xs, y, 0, n_slices_, dot); return; }
            if (C_ == 64) launch_spmv<1>(xs, y, 0, n_slices_, dot, grid); else launch_spmv<2>(xs, y, 0, n_slices_, dot, grid);
            (void)launch_overflow(xs, y, 0, n_slices_, dot, grid);      // (split matrices are plain-format: the base launch used `grid`)
        };
        for (int k = 0; k < warmup; ++k) apply_once();
        std::vector<hipEvent_t> ev((size_t)2 * reps);
        for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
        for (int k = 0; k < reps; ++k) {
            HIP_CHECK(hipEventRecord(ev[2 * k], st_));
            apply_once();
            HIP_CHECK(hipEventRecord(ev[2 * k + 1], st_));
            HIP_CHECK(hipEventSynchronize(ev[2 * k + 1]));
        }
        HIP_CHECK(hipStreamSynchronize(st_));
        for (int k = 0; k < reps; ++k) HIP_CHECK(hipEventElapsedTime(&ms_each[k], ev[2 * k], ev[2 * k + 1]));
        for (auto &e : ev) HIP_CHECK(hipEventDestroy(e));
        return 0;
    }
};

std::unique_ptr<Backend> make_backend(int device) { return std::unique_ptr<Backend>(new HipBackend(device)); }
int backend_device_count()
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}
const char *backend_static_name() { return "hip-gfx950"; }

}  // namespace pcg
