/* pcg_mi355x.h - C ABI of the MI355X-native PCG iteration engine (libpcg_mi355x.so).
 *
 * Drop-in boundary for the hot path of ankitskr/PCG-MPI-solver.  The reference has no native
 * layer and no FFI: its hot path is the Python functions in src/solver/pcg_solver.py that the
 * load-step loop (:1002-1008) calls.  Each entry point below replaces one of them; the binding a
 * reference maintainer adds is the ctypes stub shown in INTEGRATION.md (it is what
 * pcg_mi355x/_lib.py contains).
 *
 *   reference function (src/solver/pcg_solver.py)         C ABI
 *   ----------------------------------------------------  ------------------------------------
 *   calcMatVecProd(..,'Strain', x)        :242-336        pcg_apply()          (+ halo hooks)
 *   calcMatVecProd(..,'Preconditioner')   :282-287        pcg_diag()
 *   updatePreconditioner                  :346-352        pcg_build_jacobi()
 *   updateBC                              :226-238        pcg_update_bc()
 *   PCG(RefMeshPart)                      :356-598        pcg_solve_begin/_run/_end, pcg_solve()
 *   MPI_SUM                               :622-628        pcg_comm_create_rccl(): ncclAllReduce issued by the engine  | pcg_comm_hooks.allreduce
 *   Isend/Recv/Waitall interface sums     :318-334        pcg_comm_create_rccl(): grouped ncclSend/ncclRecv, comm stream | pcg_comm_hooks.halo_*
 *   np.dot(a, b*w)                        :381,415,462..  pcg_dot_w()
 *   mpiexec -np N (one rank per part)     :91, :968-970   one process per GPU, or ONE process for all GPUs: pcg_group_*()
 *   element tables -> operator            (partition_mesh.py:443-491,576-581 data contract)
 *                                                          pcg_asm_*() + pcg_create()   assembled, SELL over 3x3 blocks
 *                                                          pcg_create_ebe()             matrix-free, as the reference
 *                                                          pcg_create_csr()             from a scalar CSR matrix
 *
 * Conventions: every function returns int (0 = OK, <0 = engine error; text via
 * pcg_last_error()).  Solver outcome flags 0-4 keep the reference's meaning and are DATA
 * (pcg_result.flag), never error codes.  The caller owns every host buffer before and after a
 * call; the engine owns all device memory.  A handle is not thread-safe; one handle per GPU.
 * All floating data is IEEE f64; vectors passed to the engine have the part's full local length
 * n = 3 * n_nodes in the ENGINE's node numbering (the Python shim applies the boundary-first
 * permutation).  There is no CPU fallback: without a usable gfx950 device pcg_create() fails.
 */
#ifndef PCG_MI355X_H
#define PCG_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcg_engine pcg_engine;
typedef struct pcg_asm pcg_asm;

/* ---- library ------------------------------------------------------------------------------ */
/* Version of the structs and entry points below.  The structs carry no size field: a caller compiled against another
 * header version passes structs of another size, so check ONCE after loading the library -
 *     if (pcg_abi_version() != PCG_ABI_VERSION) refuse to run
 * (pcg_mi355x/_lib.py and the C examples do).  History: 1 = round 1; 2 = pcg_group_*, pcg_comm_* (round 2);
 * 3 = pcg_comm_hooks.collective_exchange, pcg_result.vec_ms_sum / vec_count (round 3; a library of version < 3 called the
 * hooks of a part without neighbours unconditionally - since 3 only with collective_exchange != 0);
 * 4 = pcg_abi_version(), pcg_result.fused_fallbacks (round 4);
 * 5 = pcg_comm_enable_mailbox(), pcg_group_enable_mailbox() (round 5; no struct changed);
 * 6 = pcg_enable_direct_exchange(), pcg_group_enable_direct_exchange(), pcg_create_ebe flags bit 2 (round 5; no struct changed);
 * 7 = pcg_tuning_info() (round 6; read-only, no struct changed). */
#define PCG_ABI_VERSION 7
int pcg_abi_version(void);
const char *pcg_last_error(void);
const char *pcg_backend_name(void);          /* "hip-gfx950" for the product library */
int pcg_device_count(void);

/* ---- host-side operator assembly ------------------------------------------------------------
 * One pattern-type group of the reference's SubDomainData['StrucDataList'][j]
 * (partition_mesh.py:470-489,577-578): ElemList_LocDofVector (nd,Ne) element-minor int64,
 * ElemList_SignVector (nd,Ne) bool, ElemList_Ck (Ne), ElemStiffMat (nd,nd).
 * Local DOF d belongs to node d/3, direction d%3 (partition_mesh.py:826). */
typedef struct {
    int32_t nd;
    int64_t ne;
    const int64_t *dof;
    const uint8_t *sign;
    const double *ck;
    const double *ke;
} pcg_elem_group;

/* A = sum_e P_e^T S_e (Ck_e Ke_type(e)) S_e P_e  as 3x3-block CSR over nodes, summed per entry
 * in ascending (group, element, local row, local col) order (deterministic).
 * node_perm: new index of each old node (NULL = identity). */
int pcg_asm_create(int64_t n_nodes, int32_t n_groups, const pcg_elem_group *groups,
                   const int64_t *node_perm, int32_t n_threads, pcg_asm **out);
int64_t pcg_asm_nnzb(const pcg_asm *a);
int pcg_asm_rowptr(const pcg_asm *a, int64_t *rowptr /* n_nodes+1 */);
int pcg_asm_fill(const pcg_asm *a, int32_t *cols /* nnzb */, double *vals /* nnzb*9, row-major 3x3 */);
void pcg_asm_destroy(pcg_asm *a);

/* ---- engine ---------------------------------------------------------------------------------
 * n_boundary_nodes: nodes [0, n_boundary_nodes) are shared with another part (their rows are
 * computed first so the interface exchange overlaps the interior rows); 0 for a single part.
 * rows_per_lane: SELL slice = 64*rows_per_lane block rows (1 or 2; 0 = library default), optionally OR-ed with
 *   PCG_FORMAT_DICTIONARY: store every stored block as a 2-byte index into a dictionary of the matrix's DISTINCT 3x3 blocks
 *   (compared bit by bit: lossless, the SpMV is bit-identical to the plain format) instead of its 72 bytes of values - 4 to
 *   6 bytes per block; the dictionary stays in LDS (its most frequent 1945 entries when it is larger, the rest is read
 *   through the caches).  Pattern-based meshes (the reference's domain: a few element stiffness
 *   patterns scaled by a few material factors, partition_mesh.py:443-491) assemble to a few hundred distinct blocks whatever
 *   their size; a matrix with more than 65535 distinct blocks silently keeps the plain format (pcg_matrix_dictionary). */
enum { PCG_FORMAT_DICTIONARY = 0x100 };
int pcg_create(int32_t device, int64_t n_nodes, const int64_t *rowptr, const int32_t *cols,
               const double *vals, int64_t n_boundary_nodes, int32_t rows_per_lane, pcg_engine **out);
/* The same engine straight from the host assembler, slice by slice: every block row is produced once and written into the
 * SELL layout, no 3x3-block CSR copy of the values in between; with PCG_FORMAT_DICTIONARY the 72-byte values are never
 * materialised at all - the row's blocks are hashed and stored as indices (6 bytes of host memory per stored block instead
 * of 2 x 76; a 100 M-dof operator is 5 GB on the host and 3.7 GB on the device).
 * The operator is bit-identical to pcg_asm_fill() -> pcg_create(). */
int pcg_create_asm(int32_t device, const pcg_asm *a, int64_t n_boundary_nodes, int32_t rows_per_lane, pcg_engine **out);
/* The same engine from an already assembled scalar CSR matrix (i64 row pointer, i32 columns, f64 values).
 * block = 0/3: n = 3 * nodes rows (dof = 3*node + dir) grouped into 3x3 blocks internally (the fast format);
 * block = 1:   the scalar format is kept (any n; one f64 + one i32 per non-zero = the literal CSR traffic).
 * block = 3 | PCG_FORMAT_DICTIONARY: 3x3 blocks with the value dictionary (see pcg_create). */
int pcg_create_csr(int32_t device, int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
                   int64_t n_boundary_nodes, int32_t block, pcg_engine **out);
/* Matrix-free variant: the operator stays in the reference's element-by-element form
 * (pcg_solver.py:265-300); nothing is assembled.  Same groups / numbering arguments as pcg_asm_create. */

/* The literal CSR data volume of an ASSEMBLED engine (pcg_create / pcg_create_asm, plain 3x3-block format, not split): a second
 * engine on the same device whose operator stores one f64 value + one i32 column per scalar non-zero (12 B, the format of
 * pcg_create_csr(block = 1), k_spmv_scalar) - expanded from the block format ON THE DEVICE, so that the "CSR SpMV" point of
 * SURVEY 8(d) (12 nnz + 20 n bytes) can be measured at the metric's own size without a 10 GB host CSR copy.  Same product
 * (tested); vectors of the copy have the same length and numbering as src's.  src stays valid and independent. */
int pcg_create_scalar_copy(pcg_engine *src, pcg_engine **out);
/* node_coords (n_nodes x 3, ORIGINAL node numbering, may be NULL: RefMeshPart['NodeCoordVec'],
 * partition_mesh.py:357) only steers the spatial clustering of elements into workgroup chunks.
 * flags bit0: disable the chunked (LDS-tiled) form and use one colour per launch for every group;
 *       bit1: one element per thread (256-element chunks) instead of two (512-element chunks);
 *       bit2 (ABI 6): ONE phase - the elements on the interface are not launched first; the exchange then follows the whole operator
 *             (for pcg_enable_direct_exchange: one element launch instead of two, each as long as a chunk's chain of phases). */
int pcg_create_ebe(int32_t device, int64_t n_nodes, int32_t n_groups, const pcg_elem_group *groups,
                   const int64_t *node_perm, int64_t n_boundary_nodes, const double *node_coords, int32_t flags,
                   pcg_engine **out);
void pcg_destroy(pcg_engine *e);

/* flags[d]: bit0 = this part owns dof d (DofWeightVector == 1, partition_mesh.py:870-887),
 *           bit1 = dof d is free (in LocDofEff, partition_mesh.py:350-351). */
int pcg_set_masks(pcg_engine *e, const uint8_t *flags /* n */);

/* Interface lists (OvrlpLocalDofVecList / NbrMPIdVector, partition_mesh.py:817-830), in the
 * reference's neighbour order; recv layout mirrors send layout. */
int pcg_set_halo(pcg_engine *e, int32_t n_peers, const int32_t *peer_ids, const int64_t *send_ptr /* n_peers+1 */,
                 const int32_t *send_idx /* send_ptr[n_peers] local dofs */);

/* Communication hooks: the callback seam (CPU test-suite with gloo, in-process thread communicators, or a host
 * language that wants to own the transport, e.g. torch.distributed).  The product's multi-GPU path is the native
 * communicator below, which needs no callbacks.  All buffers are DEVICE pointers, `stream` is the
 * engine's hipStream_t.  halo_begin is called once the send buffer has been packed on `stream`;
 * halo_end must make `stream` wait until the receive buffer is complete.  allreduce sums
 * `count` doubles in place across ranks.  NULL hooks = single part.
 * collective_exchange: 0 = the exchange is point-to-point between neighbours (like the reference's Isend / Recv,
 * pcg_solver.py:318-328): a part WITHOUT neighbours never sees halo_begin / halo_end.  Non-zero = halo_begin / halo_end are a
 * group-wide collective that EVERY rank must enter (torch all_to_all_single, a barrier-based test communicator): a part
 * without neighbours then also calls halo_begin(ctx, NULL, NULL, 0, stream) + halo_end once per operator apply / interface
 * sum.  (Round 2 made those calls unconditionally; round 3 made them opt-in.) */
typedef struct {
    void *ctx;
    int (*halo_begin)(void *ctx, double *dev_send, double *dev_recv, int64_t count, void *stream);
    int (*halo_end)(void *ctx, void *stream);
    int (*allreduce)(void *ctx, double *dev_buf, int32_t count, void *stream);
    int32_t collective_exchange;
} pcg_comm_hooks;
int pcg_set_comm(pcg_engine *e, const pcg_comm_hooks *hooks);
void *pcg_stream(pcg_engine *e);

/* ---- partition set-up on the device ------------------------------------------------
 * The whole-model index passes of src/solver/partition_mesh.py as HIP kernels (csrc/part_setup.hip); integer work, exact.
 * pcg_part_interface       identify_PotentialNeighbours :657-741 + config_Neighbours :745-830: which global nodes are
 *                          touched by elements of MORE THAN ONE part, and by which parts.  elem_ptr / flat_nodes = the
 *                          model's NodeGlbOffset / NodeGlbFlat as a CSR over elements (i64 offsets, i32 node ids),
 *                          ele_part = MeshPart_<N>.npy (run_metis.py:88-92).  Writes up to `cap` pairs (node, part) in
 *                          arbitrary order with duplicates (sort + unique on the host: the interface is a small subset);
 *                          *n_pairs = pairs found - call again with that capacity if it exceeds `cap`.
 * pcg_part_local_numbering config_ElemVectors :252-268: the part's ascending unique node ids (np.unique) and the local
 *                          index of every element node (getIndices :58-67), by mark + exclusive scan + gather. */
int pcg_part_interface(int32_t device, int64_t n_glob_nodes, int64_t n_elem, const int64_t *elem_ptr, const int32_t *flat_nodes,
                       const int32_t *ele_part, int64_t cap, int64_t *pairs /* 2*cap */, int64_t *n_pairs);
int pcg_part_local_numbering(int32_t device, int64_t n_glob_nodes, int64_t n_flat, const int32_t *flat_nodes,
                             int32_t *unique_nodes /* n_flat */, int32_t *local_of_flat /* n_flat */, int64_t *n_unique);

/* ---- native communicator: RCCL over xGMI, issued by the engine --------------------------------
 * Replaces the reference's mpi4py calls on the hot path without any callback into the host language:
 *   Isend / Recv / Waitall per neighbour (pcg_solver.py:318-328) -> one ncclGroupStart..ncclSend/ncclRecv..ncclGroupEnd
 *       on a dedicated communication stream, fenced with events against the engine's compute stream
 *       (interface rows -> pack -> exchange || interior rows -> wait -> add in neighbour order);
 *   MPI_SUM -> Comm.allreduce (:622-628) -> ncclAllReduce(ncclDouble, ncclSum) in place on the device status block.
 * One process per GPU, one part per process, part id == rank (pcg_solver.py:91,:320); neighbour ids passed to
 * pcg_set_halo() are peer ranks.  A communicator may serve several engines of the process, one solve at a time; it must
 * outlive every engine it is attached to (detach with pcg_set_comm_native(e, NULL) or destroy the engines first).
 * Bootstrap: rank 0 calls pcg_rccl_unique_id(), the bytes reach the other ranks by any means the launcher has
 * (MPI_Bcast in the reference's mpiexec world, torch.distributed / a file under torchrun), every rank then calls
 * pcg_comm_create_rccl() (collective).  When a native communicator is attached it takes precedence over
 * pcg_comm_hooks; the hooks remain as the seam for the CPU (gloo) test-suite. */
typedef struct pcg_comm pcg_comm;
enum { PCG_RCCL_ID_BYTES = 256 };                 /* two ncclUniqueId: exchange communicator, reduction communicator */
int pcg_rccl_unique_id(void *out /* PCG_RCCL_ID_BYTES */);
int pcg_comm_create_rccl(int32_t device, int32_t rank, int32_t nranks, const void *unique_id, pcg_comm **out);
void pcg_comm_destroy(pcg_comm *c);
int pcg_comm_rank(const pcg_comm *c);
int pcg_comm_size(const pcg_comm *c);
int pcg_set_comm_native(pcg_engine *e, pcg_comm *c /* NULL detaches */);
/* The reference's two timer buckets (updateTime :631-641: everything outside a communication call is 'calculation').
 * With timing on, HIP events bracket the compute stream's wait for the exchange and every all-reduce: the time the
 * GPU was BLOCKED in communication, which is what dT_CommWait means; pcg_result.t_comm_s then reports it per solve. */
typedef struct {
    double halo_wait_ms, allreduce_ms;
    int64_t n_halo, n_allreduce, n_halo_timed, n_allreduce_timed;
} pcg_comm_stats;
int pcg_comm_set_timing(pcg_comm *c, int32_t on);
int pcg_comm_get_stats(pcg_comm *c, pcg_comm_stats *out);
/* Engine-side reduction (round 5, OPT-IN; RCCL's ncclAllReduce stays the default): MPI_SUM (pcg_solver.py:622-628; per iteration
 * :487-488 and :504-507) through peer-mapped mailboxes.  Every rank owns a 2 KB mailbox in uncached device memory that every other
 * rank maps (same process: the pointer + peer access; other processes: hipIpcMemHandle, exchanged through the communicator
 * itself).  An all-reduce is then: write your values + a sequence number into every rank's mailbox (system-scope stores), poll
 * your own for the n contributions, add them IN RANK ORDER - identical bits on every rank, the order of the reference's
 * rank-ordered sum.  In the PCG loop this happens inside the launches that produce the operands (the last workgroup of the
 * interface fix-up for p.Ap, of the vector update for the five sums): no collective kernel, no extra launch per iteration.
 * COLLECTIVE: every rank of the communicator calls it, between solves.  *enabled_out = 1 on every rank or 0 on every rank: when any
 * rank cannot map a peer (no peer access, IPC refused, ranks on different hosts, more than 16 ranks) or the self-test all-reduce
 * returns wrong sums, all ranks keep using ncclAllReduce and pcg_last_error() says why (the call itself still returns 0).
 * A poll that never sees a peer's contribution gives up after seconds and delivers NaN; the host checks the report after every
 * iteration and the running pcg_solve_* call returns an error - nothing hangs, nothing iterates on to MaxIter.  on = 0 switches back to ncclAllReduce (collective too). */
int pcg_comm_enable_mailbox(pcg_comm *c, int32_t on, int32_t *enabled_out);
/* Engine-side neighbour exchange (round 5, OPT-IN; RCCL's grouped ncclSend / ncclRecv stays the default): the reference's
 * Isend / Recv / Waitall over the interface dofs (pcg_solver.py:307-328) as stores over xGMI.  The engine's receive buffer and one
 * arrival word per neighbour live in uncached device memory that its neighbours map (hipIpcMemHandle / peer pointers, exchanged
 * through the communicator like the mailboxes); in the applies of the PCG iteration the pack kernel then writes this rank's
 * partial sums STRAIGHT into its neighbours' buffers and posts a sequence number (release, system scope), and the interface
 * fix-up waits for its neighbours' numbers (acquire) before it adds in neighbour order - same values, same order, same bits; one
 * stream, no collective kernel, no event between streams.  Set-up applies and the true-residual branch keep ncclSend / ncclRecv.
 * Call it after pcg_set_halo and pcg_set_comm_native, between solves.  COLLECTIVE: every rank of the engine's communicator calls it
 * for its engine of the same job, neighbours or not.  *enabled_out = 1 on every rank or 0 on every rank (any rank that cannot map a
 * neighbour - no peer access, IPC refused, another host, two ranks of one process on one device - keeps all on RCCL; pcg_last_error()
 * says why, the call returns 0).  A wait that never sees a neighbour gives up after seconds; the host checks the report after every
 * iteration and the running pcg_solve_* call returns an error.  on = 0 drops the mapping (collective too: no rank may keep writing into a buffer its neighbour has freed). */
int pcg_enable_direct_exchange(pcg_engine *e, int32_t on, int32_t *enabled_out);

/* ---- operator-level calls (host vectors, length n) ----------------------------------------- */
int pcg_apply(pcg_engine *e, const double *x, double *y);            /* y = A x, interface-summed */
int pcg_diag(pcg_engine *e, double *d);                              /* diag(A), interface-summed */
int pcg_build_jacobi(pcg_engine *e, double *inv_diag_out /* n, 0 on fixed dofs; may be NULL */);
int pcg_update_bc(pcg_engine *e, const double *ref_load, const double *ud, double delta,
                  double *fext_out, double *udi_out);                /* Fext = F*d - A*(Ud*d) */
int pcg_dot_w(pcg_engine *e, const double *a, const double *b, double *out);   /* global sum a*b*w */

/* ---- PCG ------------------------------------------------------------------------------------ */
enum { PCG_STATUS_NORMAL = 0, PCG_STATUS_ZERO_RHS = 1, PCG_STATUS_GOOD_X0 = 2,
       PCG_STATUS_TOO_SMALL_TOL = 3, PCG_STATUS_RUNNING = 4 };

typedef struct {
    int32_t flag;          /* 0 converged, 1 MaxIter, 2 inf in M^-1 r, 3 stagnation, 4 breakdown  */
    int32_t status;        /* PCG_STATUS_*; TOO_SMALL_TOL = the reference's raise Warning (:549)  */
    int64_t iter;          /* as the reference stores it (loop index + 1, :584)                   */
    int64_t iters_done;    /* loop iterations completed so far (= rows written to hist)            */
    int64_t n_matvec;
    double relres;
    double norm_b;         /* sqrt(sum Fext^2 w)                                                   */
    double normr_act;
    double t_total_s, t_comm_s;            /* host wall split (dT_Calc = total - comm), :631-641  */
    double spmv_ms_sum;    /* HIP-event time of the SpMV launches when profiling is on            */
    int64_t spmv_count;
    int64_t iters_enqueued; /* iterations whose device work was enqueued: iters_done + look-ahead iterations that
                              were dropped because their predecessor ended the loop or replaced r (:527-549)  */
    double vec_ms_sum;     /* HIP-event time of the vector-phase launches (k_vec) when profiling is on      */
    int64_t vec_count;
    int64_t fused_fallbacks; /* single part: fused vector launches whose grid barrier timed out (a workgroup of the grid was not
                                resident: CU mask, compute partition, a second process on the device); such an iteration is
                                finished in the split form with the same bits, and the engine keeps to the split form afterwards */
} pcg_result;

/* inv_diag may be NULL: use the Jacobi vector built by pcg_build_jacobi().
 * hist (may be NULL): rows [NormP, NormX, NormR] per iteration (the :507 allreduce). */
/* The loop keeps ONE iteration in flight ahead of the host's tests (see pcg_driver.cpp); environment
 * PCG_LOOK_AHEAD=0 (read at engine creation) turns that off.  Results are bit-identical either way. */
int pcg_solve_begin(pcg_engine *e, const double *b, const double *x0, const double *inv_diag,
                    double tol, int64_t max_iter, int64_t glob_n_eff);
int pcg_solve_run(pcg_engine *e, int64_t n_iters /* <0 = to completion */, double *hist, int64_t hist_cap,
                  pcg_result *res);
int pcg_solve_end(pcg_engine *e, double *x_out, pcg_result *res);
int pcg_solve(pcg_engine *e, const double *b, const double *x0, const double *inv_diag, double tol,
              int64_t max_iter, int64_t glob_n_eff, double *x_out, double *hist, int64_t hist_cap,
              pcg_result *res);
/* HIP events on the engine's stream around launches of a solve: bit 0 = the operator launches (pcg_result.spmv_ms_sum / spmv_count),
 * bit 1 = the vector-phase launches (vec_ms_sum / vec_count); 0 = off.  Events cost a few microseconds per iteration. */
int pcg_set_profiling(pcg_engine *e, int32_t what);
int pcg_engine_device(const pcg_engine *e);      /* the device id the engine was created on */

/* ---- device group: ONE process drives several GPUs (SURVEY 8b pcg_group_*) ------------------------------------------
 * The reference runs one MPI rank per part (mpiexec -np N, pcg_solver.py:91); the product's default launch is the same,
 * one process per GPU.  A group is the alternative for a host program that cannot be launched N times (a notebook, an
 * embedding application): member k of the group is part k on device dev_ids[k], and every pcg_group_* call below is the
 * per-engine call of the same name made for ALL members at once - the library runs one persistent host thread per
 * member (bound to the member's device), because the calls are collective: the interface exchange and the all-reduces
 * of one member complete only when its neighbours have issued theirs.  Communication is the native communicator above
 * (one pcg_comm per member, RCCL over xGMI, unique id generated in-process), so a group solve is the same engine code
 * path as N processes; results are bit-identical to the N-process run.
 *   usage: pcg_group_create -> per member: pcg_create*(dev_ids[k], ..), pcg_set_masks, pcg_set_halo (peer id == member
 *   index), pcg_group_attach -> pcg_group_update_bc / _build_jacobi / _solve ... -> pcg_destroy(engines) -> pcg_group_destroy.
 * Array arguments hold one pointer per member (host vectors of THAT member's local length); where a per-member pointer
 * may be NULL for the single-engine call it may be NULL here, and a NULL array means "NULL for every member".
 * On failure the return value is < 0 and pcg_last_error() names the failing member(s).  A member that fails inside a
 * collective cannot be recovered from any more than a crashed rank can (its peers wait); create/attach errors are. */
typedef struct pcg_group pcg_group;
int pcg_group_create(int32_t n_dev, const int32_t *dev_ids, pcg_group **out);
void pcg_group_destroy(pcg_group *g);            /* after the engines attached to it have been destroyed or detached */
int pcg_group_size(const pcg_group *g);
int pcg_group_device(const pcg_group *g, int32_t member);
pcg_comm *pcg_group_comm(pcg_group *g, int32_t member);       /* borrowed: owned by the group */
int pcg_group_attach(pcg_group *g, int32_t member, pcg_engine *e /* created on dev_ids[member]; NULL detaches */);
int pcg_group_apply(pcg_group *g, const double *const *x, double *const *y);
int pcg_group_diag(pcg_group *g, double *const *d);
int pcg_group_build_jacobi(pcg_group *g, double *const *inv_diag_out);
int pcg_group_update_bc(pcg_group *g, const double *const *ref_load, const double *const *ud, double delta,
                        double *const *fext_out, double *const *udi_out);
int pcg_group_dot_w(pcg_group *g, const double *const *a, const double *const *b, double *out /* the global sum */);
int pcg_group_solve(pcg_group *g, const double *const *b, const double *const *x0, const double *const *inv_diag,
                    double tol, int64_t max_iter, int64_t glob_n_eff, double *const *x_out, double *const *hist,
                    int64_t hist_cap, pcg_result *res /* n_dev results, may be NULL */);
int pcg_group_set_timing(pcg_group *g, int32_t on);           /* pcg_comm_set_timing on every member */
int pcg_group_enable_mailbox(pcg_group *g, int32_t on, int32_t *enabled_out);   /* pcg_comm_enable_mailbox on every member (one process: direct peer pointers) */
int pcg_group_enable_direct_exchange(pcg_group *g, int32_t on, int32_t *enabled_out);   /* pcg_enable_direct_exchange on every member's attached engine (ABI 6) */

/* ---- measurement / unit-test entry points --------------------------------------------------- */
/* Back-to-back local SpMV launches timed with HIP events on the engine stream. */
int pcg_bench_spmv(pcg_engine *e, int32_t warmup, int32_t reps, float *ms_each /* reps */);
/* HBM stream microbenchmark on the engine's device and stream (16 B per lane, non-temporal, grid-stride - the access
 * shape of the solver's kernels): mode 0 reads `bytes`, mode 1 copies `bytes` (read + write = 2 * bytes of traffic).
 * The practical bandwidth ceiling of the box a number was measured on, reported beside the 8 TB/s spec (bench.py). */
int pcg_bench_hbm(pcg_engine *e, int64_t bytes, int32_t mode, int32_t reps, float *ms_each /* reps */);
int pcg_operator_info(pcg_engine *e, int32_t *kind /* 0 assembled, 1 matrix-free */, int64_t *n_elem, int64_t *n_slots,
                      int32_t *n_colors, int64_t *n_chunks);
/* What ONE local operator apply has to move (bytes of the stored operator + x in + y out, counted from the uploaded
 * structures) and compute (flops of the un-padded operator): the denominators of the roofline report (bench.py). */
int pcg_operator_cost(pcg_engine *e, double *bytes_per_apply, double *flops_per_apply);
/* nnzb = the matrix's 3x3 blocks; stored_blocks = what the SELL layout holds (padding included; for a matrix the engine split
 * into a base part + an overflow part for its long rows - octree meshes, csrc/sell.cpp split_overflow, PCG_SELL_SPLIT - both parts). */
int pcg_matrix_info(pcg_engine *e, int64_t *nnzb, int64_t *stored_blocks, int64_t *n_slices, int32_t *slice_rows);
/* FNV-1a over the host-side operator arrays (slice pointers, columns, values or indices + table, diagonal), recorded at
 * creation when the environment has PCG_MATRIX_FINGERPRINT (0 otherwise): lets tests assert that two construction paths
 * produce the same operator. */
int pcg_matrix_fingerprint(pcg_engine *e, uint64_t *out);
/* What the engine decided by MEASUREMENT at its first solve (round 6; assembled operators of >= 1 M dof, nothing below):
 * spmv_launches_per_apply: launches of the SpMV kernel one apply makes - 1, or the slice range in several launches that write their y
 * at their end (csrc/kernels_spmv.hpp HOLD; same bits; PCG_SPMV_HOLD=0 / 4 forces a form) - measurement code quotes per-LAUNCH bytes and
 * times next to a profiler's per-kernel average; vectors_placed: 1 when the roles of the solve's vectors were handed out by timing the
 * SpMV on every candidate buffer (PCG_VEC_PLACEMENT=0 switches that off).  Either pointer may be NULL.  Before the first solve: 1 and 0. */
int pcg_tuning_info(pcg_engine *e, int32_t *spmv_launches_per_apply, int32_t *vectors_placed);
/* n_unique: distinct blocks of the value dictionary (0 = plain values); n_in_lds (may be NULL): how many of them - the most
 * frequent ones - the SpMV kernel keeps in LDS; lds_share (may be NULL): the share of the stored blocks those cover. */
int pcg_matrix_dictionary(pcg_engine *e, int64_t *n_unique, int64_t *n_in_lds, double *lds_share);
/* single fused kernels on host vectors, for per-kernel parity tests */
int pcg_k_update_p(pcg_engine *e, double *p, const double *r, const double *inv_diag, double beta, int32_t first);
int pcg_k_fused_update(pcg_engine *e, double alpha, const double *p, const double *q, double *r,
                       const double *x_old, double *x_new, const double *inv_diag,
                       double *sums5 /* sqP, sqX, sqR, rho_next, n_inf */);
/* The whole vector phase of one iteration on given vectors (pcg_solver.py:501-516 and :447-479 of the next iteration):
 * r and x updated with the given alpha, the five sums, p_next = M^-1 r' + (rho' / rho) p.  fused != 0: the single launch of
 * the single-part solve loop (grid-wide reduction inside the kernel); 0: the split form of the multi-GPU loop (update,
 * reduce, k_update_p).  Both produce the same bits (tests). */
int pcg_k_vec_iteration(pcg_engine *e, double alpha, double rho, const double *p, const double *q, double *r,
                        const double *x_old, double *x_new, const double *inv_diag, double *p_next,
                        double *sums5 /* sqP, sqX, sqR, rho_next, n_inf */, int32_t fused);
int pcg_k_residual(pcg_engine *e, const double *b, const double *ax, double *r, const double *inv_diag,
                   double *sums3 /* sqR, rho, n_inf */);
int pcg_k_spmv_local(pcg_engine *e, const double *x, double *y, double *pxy /* sum x*y*w or NULL */);

#ifdef __cplusplus
}
#endif
#endif
