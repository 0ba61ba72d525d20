// Internal interfaces of libpcg_mi355x (not installed; the public surface is include/pcg_mi355x.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pcg_mi355x.h"

namespace pcg {

int set_error(const std::string &msg);   // stores the message, returns -1
const std::string &last_error_string();

// ---- operator storage ------------------------------------------------------------------------
// SELL-C over 3x3 node blocks.  Slice s holds block rows [s*C, s*C+C), padded to the slice's
// widest row (`width`).  Within a slice everything is column-major over the C rows so that
// consecutive lanes read consecutive addresses:
//     vals[((slice_ptr[s] + k) * 9 + c) * C + (row - s*C)]     c = 3*a + b of the 3x3 block
//     cols[ (slice_ptr[s] + k)          * C + (row - s*C)]     block column (node) index
// Padding entries carry value 0 and the row's own (valid) column.  C = 64 * rows_per_lane.
struct SellHost {
    int32_t bs = 3;               // block size: 3 = 3x3 node blocks (the solver's format); 1 = scalar rows (literal CSR data:
                                  // one f64 + one i32 column per non-zero), `n_nodes` then counts scalar rows
    int64_t n_nodes = 0;
    int64_t n_slices = 0;
    int32_t C = 64;
    int64_t nnzb = 0;             // true (unpadded) block count
    int64_t n_bnd_slices = 0;     // slices [0, n_bnd_slices) contain the interface rows
    std::vector<int64_t> slice_ptr;
    std::vector<int32_t> cols;
    std::vector<double> vals;
    std::vector<double> diag;     // diag(A) local, length 3*n_nodes (extracted at build time)
    // Value dictionary (compress_blocks): the 9 values of stored block q = (slice_ptr[s] + k) * C + lane are
    // dict[9 * bidx[q] .. +9) (row-major 3x3) and `vals` is empty - 2 bytes per stored block instead of 72.
    std::vector<uint16_t> bidx;
    std::vector<double> dict;             // most frequent block first
    std::vector<int64_t> dict_count;      // stored blocks per entry (descending)
    int64_t n_unique() const { return (int64_t)(dict.size() / 9); }
    // Overflow part (split_overflow; plain format, C = 64): a slice stores only its first wb block columns (the slice pointers say
    // so); rows longer than that CONTINUE in a second SELL matrix over the compacted list of those rows, in the same column order
    // (k_spmv_ovf resumes from the y the base part left: the sum of a row is formed in the order of the unsplit matrix).
    int64_t ov_slices = 0, ov_bnd_slices = 0;   // overflow slices [0, ov_bnd_slices) hold rows of the base slices [0, n_bnd_slices)
    std::vector<int64_t> ov_slice_ptr;          // block columns per overflow slice (prefix sums)
    std::vector<int32_t> ov_rows;               // (ov_slices, 64): the row of each lane, -1 = padding lane
    std::vector<int32_t> ov_cols;               // like cols
    std::vector<double> ov_vals;                // like vals
    std::vector<uint64_t> ov_mask;              // per base slice: lanes whose row continues in the overflow part
    // Windowed overflow (round 4, the default): the base slices are cut into WINDOWS (consecutive slices); the overflow rows of a
    // window are packed into overflow slices of their own, which follow each other in ov_*: window w = base slices
    // [win_slice[w], win_slice[w+1]) + overflow slices [win_ov[w], win_ov[w+1]).  ONE workgroup works through a window - base
    // slices, block barrier, its overflow slices (k_spmv_win): the x lines the overflow rows gather were just touched by the same
    // rows' base part (L1 / L2 hits instead of a second sweep over x by a second launch), y makes its round trip through L2.
    // Windows [0, n_bnd_windows) cover the interface slices [0, n_bnd_slices).  Empty = the round-3 form (k_spmv + k_spmv_ovf).
    std::vector<int64_t> win_slice, win_ov;
    int64_t n_bnd_windows = 0;
};

// Octree meshes put rows of 27 ... 99 blocks side by side in every 64-row slice: SELL pads them all to the longest (55 % of the
// stored blocks at 1 M / 10 M dof).  split_overflow picks per slice the base width that minimises (stored base blocks + 1.5 x
// overflow blocks), moves the rest of the longer rows into the overflow part (rows sorted by excess length inside windows of 512
// overflow rows: the padding of THAT matrix) and compacts the base arrays in place.  Returns false and leaves `m` unchanged when
// less than `min_saving` of the stored blocks would go (bricks).  Row sums keep their order: results are bit-identical.
bool split_overflow(SellHost &m, double min_saving, int n_threads, int target_blocks = 1024);

void csr_to_sell1(int64_t n, const int64_t *rowptr, const int32_t *cols, const double *vals, int64_t n_boundary_rows,
                  int n_threads, SellHost &out);
void bsr_to_sell(int64_t n_nodes, const int64_t *rowptr, const int32_t *cols, const double *vals,
                 int64_t n_boundary_nodes, int32_t rows_per_lane, int n_threads, SellHost &out);

// Replace the values of a 3x3-block SELL matrix (C = 64) by indices into a dictionary of its DISTINCT blocks (compared by
// bit pattern: lossless).  Pattern-based meshes - the reference's domain: a few element stiffness patterns scaled by a few
// material factors (partition_mesh.py:443-491) - assemble to a few hundred distinct blocks however large the mesh is.
// Returns false and leaves `m` unchanged when there are more than max_unique (<= 65535) of them.  The dictionary is sorted
// by descending frequency (ties by bit pattern), i.e. independent of the thread count.
bool compress_blocks(SellHost &m, int64_t max_unique, int n_threads);

// Building blocks of the dictionary (shared by compress_blocks and the streaming assembler path, assemble.cpp).
struct BlockKey {                     // the bit pattern of a 3x3 block
    uint64_t w[9];
    bool operator==(const BlockKey &o) const;
    bool operator<(const BlockKey &o) const;
};
struct BlockHash { size_t operator()(const BlockKey &k) const; };
class BlockTable {                    // one thread's distinct blocks in first-seen order, with their occurrence counts
    std::unordered_map<BlockKey, uint32_t, BlockHash> tab_;
    int64_t cap_;
public:
    explicit BlockTable(int64_t cap = 65535) : cap_(cap) {}
    std::vector<BlockKey> keys;
    std::vector<int64_t> count;
    int32_t add(const BlockKey &k);   // local id, or -1 when the table would exceed its capacity
};
// bidx holds LOCAL ids, thread t having written the stored slots [slots[t].first, slots[t].second): merges the tables into
// m.dict / m.dict_count (descending frequency, ties by bit pattern), rewrites bidx to final ids and moves it into m.bidx;
// m.vals is released.  false (m untouched) when the union has more than max_unique entries.
bool finish_dictionary(SellHost &m, std::vector<BlockTable> &local, const std::vector<std::pair<size_t, size_t>> &slots,
                       std::vector<uint16_t> &bidx, int64_t max_unique);
// SELL straight from the host assembler, slice by slice: every block row is produced once and written into the slice
// layout - no 3x3-block CSR copy of the values in between.  want_dict: the blocks are hashed and stored as table indices,
// the 72-byte values are never materialised (cols + 2-byte indices: 6 B instead of 76 + 76 B of host memory per stored
// block); false (out unspecified) when there are more than max_unique distinct blocks.  Same SellHost as
// pcg_asm_fill -> bsr_to_sell (-> compress_blocks).
bool asm_to_sell(const pcg_asm *a, int64_t n_boundary_nodes, int32_t rows_per_lane, bool want_dict, int64_t max_unique, SellHost &out);
// the assembled 3x3-block CSR arrays of a pcg_asm (views into it) / its values (pcg_asm_fill without the column copy)
void asm_views(const pcg_asm *a, int64_t *n_nodes, const int64_t **rowptr, const int32_t **cols);
void asm_fill_values(const pcg_asm *a, double *vals);

// ---- matrix-free (element-by-element) operator ------------------------------------------------
// The reference's own algorithm (pcg_solver.py:265-300): per element gather x, flip signs, multiply
// by Ck_e * Ke[type], flip signs, scatter-add.  Elements are greedily coloured so that no two elements
// of one colour share a node: a colour is one conflict-free launch (plain read-modify-write, no
// atomics) and the colour order fixes the summation order -> bit-reproducible.  Elements touching
// interface nodes form phase 0 (computed first, so the exchange overlaps phase 1 = the rest).
struct EbeGroupHost {
    int32_t nd = 0;
    int64_t ne = 0;
    std::vector<int32_t> dof;      // (nd, ne) element-minor, engine dof numbering, elements sorted by (phase, colour)
    std::vector<uint8_t> sign;     // (nd, ne)
    std::vector<double> ck;        // (ne)
    std::vector<double> ke;        // (nd, nd)
};
struct EbeRange { int32_t group; int64_t lo, hi; };      // elements [lo,hi) of `group`: one launch

// Chunked form for node-blocked pattern types (slots 3l..3l+2 = the three dofs of local node l, at most 32
// nodes): a chunk = up to 256*ept spatially clustered elements of one group = one workgroup.  The chunk's
// unique nodes are staged in LDS (x tile in, y tile out); elements of a chunk are sub-coloured and add
// into the LDS y tile phase by phase.  All chunks of a phase run in ONE launch per node-count class: a
// tile node that belongs to a single chunk is stored straight to y, a node shared by several chunks goes
// to that chunk's slot of a boundary buffer and a second small kernel sums the slots of every shared node
// in chunk order (deterministic, no atomics, no colour-by-colour launches, no read-modify-write of y).
constexpr int kChunkThreads = 256;                 // workgroup size
constexpr int kChunkMaxNodes = 768;                // 8x8x8 hex cells -> 729 nodes (LDS x + y tiles = 36.9 KB)
constexpr int kChunkClasses = 6;                   // 0: exactly 8 nodes (hex8, unguarded kernel); 1..3: <= 16 / 24 / 32 nodes;
                                                   // 4: fewer than 8 nodes (padded to 8); 5: mixed-type chunks (EbeMixedHost)
constexpr int kMixedClass = 5;
constexpr int kMixedHexSlots = 512;                // hex8 element slots of a mixed chunk (two passes of 256 threads)
constexpr int64_t kMixedHexTilesBelow = 1200000;   // mixed chunks: below this many elements the 8-node type runs in matrix-core tiles too (ebe.cpp)
constexpr int kMixedMaxTiles = 24;                 // 16-element tiles of the other pattern types per mixed chunk
constexpr int kMixedTargetChunks = 0;              // mixed chunks, hex tiles: lower the node cap of small meshes until they make about this many chunks (0 = off)
constexpr int kMixedMinNodeCap = 256;
constexpr int kMixedFragAhead = 4;                 // k-steps whose matrix fragments k_ebe_mixed requests ahead of their instructions
struct EbeClassHost {
    int32_t nnp = 8;                   // padded nodes per element; the kernel is instantiated for NDP = 3*nnp
    bool full = false;                 // every element of the class has exactly nnp nodes (no padding guards needed)
    int32_t ept = 1;                   // hex8 class: elements per thread, a chunk holds 256*ept elements
    int32_t ce = kChunkThreads;        // element slots per chunk: 256*ept for the hex8 class, 64 for the others (one element per
                                       // LANE; the four waves of the workgroup contract a quarter of the output rows each)
    int32_t words = 1;                 // sign words per element = NDP/32 + 1; bits 24..31 of the last word = sub-colour
    int32_t max_nodes = kChunkMaxNodes; // tile nodes a chunk of this class may have (512 for the 256-element hex8 chunks)
    bool direct = false;               // no node tile (k_ebe_direct): the chunk's part of `nodes` / `dst` is per element-node INCIDENCE,
                                       // (local node, element slot)-major with 64 slots per local node, -1 / INT_MIN in padding slots;
                                       // lid is empty and tslot unused.  The 16- / 24- / 32-node classes unless PCG_EBE_DIRECT=0.
    int64_t n_chunks = 0;
    std::vector<uint16_t> lid;         // (n_chunks, nnp, ce) local node index of element-node l (0 for padding)
    std::vector<double> ck;            // (n_chunks, ce)   (0 for padding slots)
    std::vector<uint32_t> sgn;         // (n_chunks, words, ce); sub-colour 255 = padding slot
    std::vector<double> ke_col;        // (groups of the class, NDP b, NDP a) column-major, zero padded
    std::vector<double> ke_rows;       // 64-slot classes: (groups, 4 waves, NDP b, NDP/4 a): the rows of one wave contiguous per column
    std::vector<int32_t> list[2];      // per phase: global chunk ids of this class (one launch each)
};
// Mixed-type chunks (round 4; class kMixedClass): a chunk is a run of the GLOBAL Morton order of the elements - every node-blocked
// pattern type together - so the elements around a node sit in ONE workgroup whatever their type and the node is summed in the
// chunk's LDS tile; only nodes on the surface of the run go through boundary slots.  Inside a chunk:
//   * the elements of the mesh's most populous 8-node type (`hex_group`) fill the hex section - one element per thread in two
//     passes of 256, the element matrix as SGPR operands (k_ebe_hexs' contraction) - EbeClassHost lid / ck / sgn of class 5;
//   * every other element sits in a 16-element TILE of ONE pattern type: Y(nd x 16) = Ke(nd x nd) . U(nd x 16) on
//     v_mfma_f64_16x16x4_f64, the A operand streamed from `frag` (L2-resident, pre-permuted so that a lane gathers and scatters
//     WHOLE nodes: lane (g, e) of the wave handles the local nodes g, g + 4, g + 8, ... of element e).
// The sums a chunk forms for a node are ordered: hex section (pass, wave, sub-colour), then the tiles in ascending order, the
// elements of a tile by their tile-local colour - bit-reproducible.
struct EbeMixedType {                  // a pattern type that occurs in tiles
    int32_t group = 0, nn = 0, nd = 0; // element group, nodes, dofs
    int32_t J = 0;                     // node quartets: ceil(nn / 4); k-steps = 3 J, M-tiles = ceil(3 J / 4)
    int64_t frag_off = 0;              // offset (doubles) of its fragments: frag[frag_off + ((3 j + c) * MT + mt) * 64 + lane]
};
struct EbeMixedHost {
    int32_t hex_group = -1;            // element group of the hex section, -1 = none
    int32_t nnpt = 8;                  // local-node stride of tlid (max nn over the tile types, rounded up to a multiple of 4)
    int32_t words = 1;                 // sign words per tile element
    int32_t max_mt = 0;                // max M-tiles of any type (selects the kernel instantiation)
    std::vector<EbeMixedType> types;
    std::vector<double> frag;
    int64_t n_tiles = 0;
    std::vector<int32_t> tile_type;    // (n_tiles) index into types
    std::vector<int32_t> tile_ncol;    // (n_tiles) tile-local colours in use
    std::vector<uint16_t> tlid;        // (n_tiles, nnpt, 16) slot in the chunk's LDS tile of local node l of element e (0 padding)
    std::vector<double> tck;           // (n_tiles, 16), 0 in padding slots
    std::vector<uint32_t> tsgn;        // (n_tiles, words, 16) sign bits of the element's dofs
    std::vector<uint8_t> tcol;         // (n_tiles, 16) tile-local colour, 255 = padding slot
    std::vector<uint8_t> tperm;        // (n_tiles, 16) dof order of the element: slot 3 l + c of its type is component (tperm >> 2 c) & 3
                                       //               of local node l (0 | 1 << 2 | 2 << 4 = x, y, z order; ebe.cpp node_blocked kind 2)
    int64_t hex_elems = 0, tile_elems = 0;
    // Hex tiles (round 4, PCG_EBE_HEX_TILES): the standard 8-node type runs on the matrix cores as well - types[hex_tile_type] (= 0),
    // no hex section.  A chunk's hex tiles come first and are COLOUR-PURE: the chunk's elements of the type are coloured (no two of
    // a colour share a node), sorted by colour and cut into tiles colour by colour, so the 16 elements of a tile add in one
    // instruction per dof and the tiles of one colour need no order among themselves: tile t may add once `tile_wait[t]` tiles of
    // its chunk have completed (= the tiles of all colours before its own; the tiles after the hex tiles wait for all before them).
    int32_t hex_tile_type = -1;
    std::vector<int32_t> chunk_hex_tiles;   // (mixed chunks) hex tiles of the chunk
    std::vector<int32_t> tile_wait;         // (n_tiles) completed tiles of the chunk this tile's adds wait for
};

struct EbeChunkedHost {
    int64_t n_chunks = 0;
    std::vector<int32_t> hdr;          // (n_chunks, 8): node_off, n_nodes, n_subcolours, ke index in class,
                                       //                chunk index in class, nd, class, 16-element tiles in use (hex8 class)
                                       // mixed chunks:  node_off, n_nodes, sub-colours of the hex section, hex slots in use,
                                       //                chunk index in class, tiles, class (5), first tile
    std::vector<int32_t> nodes;        // concatenated unique node ids (engine numbering, ascending per chunk: the global
                                       // loads / stores of a chunk walk memory in address order)
    std::vector<uint16_t> tslot;       // same shape: slot of the node in the chunk's LDS tile (what `lid` refers to); the
                                       // slot order is chosen against LDS bank conflicts, see ebe.cpp tile_key
    std::vector<int32_t> dst;          // same shape: >= 0: y offset 3*node (exclusive node); < 0: -(boundary slot + 1)
    int64_t n_slots = 0;               // boundary-buffer slots (one per (chunk, shared node); one per incidence in direct chunks)
    int64_t direct_entries = 0;        // entries of `nodes` that belong to direct chunks (incl. padding slots)
    std::vector<int32_t> sh_node[2];   // per phase: shared nodes whose sum becomes final after that phase (ascending)
    std::vector<int32_t> sh_ptr[2];    //            CSR over their slots
    std::vector<int32_t> sh_slot[2];   //            slots in ascending chunk order; node-major numbering: sh_slot[ph][q] ==
                                       //            (ph ? sh_ptr[0].back() : 0) + q, so a device kernel needs no slot list
    bool needs_zero = false;           // some node is touched by no chunk (isolated, or only by non-chunked groups)
    int32_t node_cap_used = 0;         // mixed chunks: the node cap the planner cut the runs with (kChunkMaxNodes unless lowered for a small mesh)
    EbeClassHost cls[kChunkClasses];
    EbeMixedHost mixed;
    int32_t max_subcolors = 0;
};
struct EbeHost {
    int64_t n_nodes = 0;
    std::vector<EbeGroupHost> groups;
    std::vector<EbeRange> ranges[2];                      // per phase, in colour order (groups NOT covered by `chunked`)
    EbeChunkedHost chunked;                               // nd == 24 node-blocked groups
    int32_t n_colors[2] = {0, 0};
    std::vector<double> diag;                             // local diag(A), length 3*n_nodes
    int64_t n_elem = 0, n_slots = 0;                      // totals (sum nd*ne = NCountDof)
};
void build_ebe(int64_t n_nodes, int32_t n_groups, const pcg_elem_group *groups, const int64_t *node_perm,
               int64_t n_boundary_nodes, const double *node_coords, bool allow_chunked, int ept, EbeHost &out);

// ---- device back end ----------------------------------------------------------------------------
// The product library implements this with hand-written HIP kernels (hip_backend.hip).  The ONLY
// other implementation lives under tests/hostops/ (a plain-loop test double compiled into a
// separate test library so that the control flow in pcg_driver.cpp and the comm plumbing can be
// exercised by the CPU/gloo test-suite).  The product never contains or loads a CPU path.
//
// All `double*` below are buffers obtained from alloc() ("device" pointers).
enum { ST_RHO = 0, ST_PQ = 1, ST_ALPHA = 2, ST_STOP = 3, ST_SQP = 4, ST_SQX = 5, ST_SQR = 6,
       ST_RHO_NEXT = 7, ST_NINF = 8, ST_ERR = 9 /* the fused vector kernel's grid barrier timed out */, ST_COUNT = 16 };
constexpr int kStatusSlots = 4;          // status ring of the solve loop's look-ahead (2 would do; 4 keeps slots apart)

// ---- engine-side all-reduce through peer-mapped mailboxes (round 5, opt-in: pcg_comm_enable_mailbox) -------------------------
// MPI_SUM (pcg_solver.py:622-628) without a collective kernel: every rank owns a small mailbox in device memory that every other
// rank has mapped (same process: the pointer itself + peer access; other processes: hipIpcOpenMemHandle).  All-reduce number
// `seq` of a communicator: a rank writes its values and then `seq` (release, system scope) into slot [seq & 1][its rank] of EVERY
// rank's mailbox, polls the slots [seq & 1][0 .. n) of its own until each carries `seq` (acquire) and sums them IN RANK ORDER -
// the same bits on every rank, and the order of the oracle's _allreduce.  Two parities suffice: a rank can start all-reduce
// q + 1 while a peer still reads q, but nobody starts q + 2 before every rank has contributed to q + 1, i.e. finished reading q.
// The exchange runs INSIDE the launch that produced the values (the last workgroup of k_fixup / k_vec<false>) or as a one-wave
// kernel of its own (k_mail_allreduce) - no ncclAllReduce launch on the critical path of an iteration.
constexpr int kMailMaxRanks = 16;
constexpr int kMailSlotWords = 8;                  // 64 B per (parity, source rank): [0] = seq, [1 .. 7] = values
constexpr int kMailMaxCount = kMailSlotWords - 1;
struct MailDesc {                                  // passed BY VALUE to the kernels
    double *peer[kMailMaxRanks];                   // peer[r]: rank r's mailbox as mapped here; peer[rank] = this rank's own
    unsigned *err;                                 // host-visible word: != 0 after a poll gave up (the result is NaN then)
    unsigned long long seq;                        // number of THIS all-reduce (1, 2, ...: every rank issues the same sequence)
    int rank, n;
    unsigned long long timeout_ticks;              // wall_clock64() ticks (100 MHz) a poll may last; 0 = no limit (the CPU double)
};

// ---- engine-side neighbour exchange through peer-mapped receive buffers (round 5, opt-in: pcg_enable_direct_exchange) ---------
// The reference's Isend / Recv / Waitall (pcg_solver.py:318-328) without a collective kernel and without a second stream: every
// engine owns its receive buffer + one arrival word per neighbour in uncached device memory that its neighbours have mapped (same
// mapping machinery as the mailboxes).  Exchange number `seq` of an engine: the pack kernel (k_halo_put) writes this rank's partial
// sums STRAIGHT into every neighbour's receive buffer (stores over xGMI), its last workgroup then posts `seq` into this rank's
// arrival word at every neighbour (release, system scope); the fix-up kernel polls its own arrival words until every neighbour has
// posted `seq` (acquire) and adds in neighbour order as before.  One stream, two launches, no ncclSend / ncclRecv kernel, no events.
// Safe without a second buffer: a neighbour starts exchange q + 1 only after an all-reduce that needs this rank's p.Ap of
// iteration q, which is formed after the fix-up of q has read the buffer (the driver uses it for the applies of the iteration only).
constexpr int kDirectMaxPeers = 32;
struct DirectDesc {                                // passed BY VALUE to k_halo_put
    double *peer_recv[kDirectMaxPeers];            // [j]: where this rank's segment lives inside neighbour j's receive buffer (mapped here)
    unsigned long long *peer_flag[kDirectMaxPeers];   // [j]: this rank's arrival word at neighbour j
    const unsigned long long *my_flags;            // this rank's arrival words, one per neighbour (neighbour order)
    long long seg[kDirectMaxPeers + 1];            // send_ptr: segment j = packed entries [seg[j], seg[j + 1])
    unsigned *err;                                 // host-visible word: != 0 after a poll gave up
    unsigned long long seq;                        // number of THIS exchange (1, 2, ...)
    int n_peers;
    unsigned long long timeout_ticks;              // as MailDesc
};
struct FixWait {                                   // passed BY VALUE to k_fixup: n == 0 = the receive buffer is complete at launch (RCCL path)
    const unsigned long long *flags;
    unsigned *err;
    unsigned long long seq;
    int n;
    unsigned long long timeout_ticks;
};
inline FixWait fix_wait_of(const DirectDesc *d)
{
    FixWait w{};
    if (d) { w.flags = d->my_flags; w.err = d->err; w.seq = d->seq; w.n = d->n_peers; w.timeout_ticks = d->timeout_ticks; }
    return w;
}
struct HaloHost;
class DirectLink {                                 // one per engine; created collectively by Comm::direct_link
public:
    virtual ~DirectLink() {}
    virtual double *recv() = 0;                    // the receive buffer the fix-up reads from now on (uncached, mapped by the neighbours)
    virtual DirectDesc next() = 0;                 // the descriptor of the NEXT exchange (advances the sequence)
    virtual void check() = 0;                      // throws when a poll has timed out since the last call
    virtual bool faulted() const { return false; } // a poll of this link has ever timed out (Comm::engine_side_sync then retires the forms)
};

struct HaloHost {
    int32_t n_peers = 0;
    std::vector<int32_t> peer_ids;
    std::vector<int64_t> send_ptr;       // n_peers + 1
    std::vector<int32_t> send_idx;       // local dofs, concatenated by peer (neighbour order)
    // derived: for every interface dof (ascending), the receive-buffer slots to add, neighbour order
    std::vector<int32_t> fix_dof;        // unique interface dofs
    std::vector<int64_t> fix_ptr;        // len fix_dof.size()+1
    std::vector<int32_t> fix_pos;        // positions in the receive buffer
};

class Backend {
public:
    virtual ~Backend() {}
    virtual const char *name() const = 0;
    virtual int device() const { return 0; }
    // make the back end's device the calling host thread's current device: every C-ABI entry point does this first, so
    // one process may hold engines on several GPUs (pcg_group_*: one host thread per member) - a no-op once bound
    virtual void bind_thread() {}
    virtual void *stream() = 0;
    virtual void *alloc(size_t bytes) = 0;
    virtual void release(void *p) = 0;
    virtual void h2d(void *dst, const void *src, size_t bytes) = 0;   // ordered on the stream, host-synchronous
    virtual void d2h(void *dst, const void *src, size_t bytes) = 0;   // ordered on the stream, host-synchronous
    virtual void d2d(void *dst, const void *src, size_t bytes) = 0;   // async on the stream
    virtual void zero(void *dst, size_t bytes) = 0;                   // async on the stream
    virtual void sync() = 0;

    virtual void upload_matrix(const SellHost &m) = 0;
    virtual void upload_ebe(const EbeHost &m) = 0;
    // this (empty) back end becomes the scalar-row copy (SellHost::bs == 1 layout: one f64 + one i32 column per non-zero) of the
    // plain 3x3-block matrix `src` holds; ptr1 = scalar slice pointers (n_rows / 64 slices), src_ptr = src's block slice pointers
    virtual void upload_scalar_copy(Backend &src, const std::vector<int64_t> &ptr1, const std::vector<int64_t> &src_ptr, int64_t n_rows) = 0;
    // y (+)= sum over the elements of phases [phase_lo, phase_hi) ; zero_first clears y before
    // with_dot: also accumulate partials of sum x[d]*y[d]*own_free(d) over the dofs d >= dot_lo that become
    // final in these phases (interface dofs < dot_lo get theirs from boundary_fixup); returns false when the
    // operator cannot fuse the dot (then the caller runs dot_w).  reduce with reduce_dot().
    virtual bool ebe_apply(const double *x, double *y, int phase_lo, int phase_hi, bool zero_first, bool with_dot,
                           int64_t dot_lo) = 0;
    // false when some pattern type runs outside the chunked form (its colour launches ADD into y, chunk stores assign):
    // the two phases may then not be interleaved with the interface exchange (pcg_driver.cpp apply())
    virtual bool ebe_can_split() const = 0;
    virtual void reload_tuning() {}                        // re-read the environment switches a solve may be A/B-tested with
    virtual int col_index_bytes() const { return 4; }      // bytes per stored block column after upload_matrix (2: 16-bit offsets)
    virtual int64_t dict_lds_entries() const { return 0; } // value dictionary: leading entries the SpMV kernel keeps in LDS
    virtual void upload_masks(const uint8_t *flags, int64_t n) = 0;
    virtual void upload_halo(const HaloHost &h) = 0;

    // y[rows of slices lo..hi) = A x ; if partial_slot >= 0 also writes per-block partials of
    // sum_{rows} xdot[i]*y[i]*own_free(i) into the partials buffer starting at that slot.
    // pack_send != null (the interface rows' launch of a part with neighbours): the launch also writes the send buffer - halo_pack
    // folded into its epilogue (round 4: the multi-part iteration in five launches); a back end that cannot fuse it for the stored
    // format packs with a launch of its own, the result is the same
    virtual void spmv(const double *x, double *y, int64_t slice_lo, int64_t slice_hi, bool with_dot, double *pack_send = nullptr) = 0;
    virtual void halo_pack(const double *y, double *send) = 0;
    // direct exchange (DirectDesc above): pack AND deliver - the packed values go straight into the neighbours' receive buffers,
    // the last workgroup posts the arrival words
    virtual void halo_put(const double *y, const DirectDesc &d) { (void)y; (void)d; throw std::runtime_error("this back end has no direct exchange"); }
    virtual bool direct_kernels_available() const { return false; }
    // interface rows: y[d] += sum recv[...] (neighbour order); optional dot over the boundary-slice rows
    // reduce_pq != null (with_dot): the LAST workgroup of this launch to finish also sums every dot partial of the apply - the
    // operator launches' and this one's, in reduce_dot()'s fixed order - into reduce_pq[0]: no reduce launch
    // mail != null (with reduce_pq): that last workgroup then also all-reduces the sum ACROSS THE RANKS through the mailboxes -
    // reduce_pq[0] is the global p.Ap when the launch is done, no all-reduce call follows
    // direct != null: `recv` is filled by the neighbours' k_halo_put of exchange direct->seq - every workgroup first waits for their
    // arrival words
    virtual void boundary_fixup(double *y, const double *recv, const double *xdot, bool with_dot, double *reduce_pq = nullptr,
                                const MailDesc *mail = nullptr, const DirectDesc *direct = nullptr) = 0;
    // forget the dot partials of earlier launches (call before an apply that wants the fused dot)
    virtual void begin_dot() = 0;
    // red[0] = sum of the SpMV-dot partials (interior launch, then boundary fix-up; fixed order)
    virtual void reduce_dot(double *red) = 0;
    // Status block: `st` is the device block the reduce/scalar kernels write into.  A back end may mirror
    // those writes into host-visible memory so that read_status() is a stream sync instead of a copy;
    // it returns false when it has no mirror (the caller then copies).
    virtual void set_status_block(double *st) = 0;
    virtual bool read_status(double *host_out) = 0;
    // Status ring for the one-iteration look-ahead of the solve loop: the words an iteration writes go to ring slot
    // `slot` (set before the iteration is enqueued); publish_status() closes the iteration (copy_block: the device
    // block was rewritten in place by an all-reduce after the kernels mirrored it, copy it whole) and marks the point
    // the host may wait for; wait_status() blocks until THAT iteration is done - later work may already be queued.
    virtual void set_status_slot(int slot) = 0;
    virtual void publish_status(bool copy_block) = 0;
    virtual void wait_status(int slot, double *host_out) = 0;
    // p_out = first ? M^-1 r : M^-1 r + beta p_in , beta = st[RHO_NEXT] / rho_prev (device-side division of the same
    // two doubles the host divides for its Flag-4 test)                     (:447,:472-479)
    // publish_slot >= 0: the launch FIRST copies the status block into that ring slot (publish_status(true) of the iteration
    // before, folded in) and the slot counts as published once the launch is done
    virtual void update_p(double *p_out, const double *p_in, const double *r, const double *minv, const double *st,
                          double rho_prev, bool first, int publish_slot = -1) = 0;
    // The vector phase of an iteration (:487-516, and :447-479 of the next one).
    //   alpha: pq_src 2 = p.Ap is the fixed-order sum of the dot partials of the operator launches since begin_dot() (single
    //          part: no reduce launch), 1 = st[PQ] (already all-reduced), 0 = alpha is given in st[ALPHA];
    //          st[RHO] = rho = st[RHO_NEXT], st[ALPHA] = rho / pq, st[STOP] per :492-498 (sticky; a frozen call updates nothing)
    //   sums of p^2 w, x_old^2 w ; r_new = r_old - alpha q ; sum r^2 w ; x_new = x_old + alpha p ; z = M^-1 r_new ;
    //   sum z r w ; count of inf in z
    //   p_next == null: the partial sums are left for reduce_update().  Returns false.
    //   p_next != null (single part only, ask vec_fused_available() first): ONE launch also reduces the five sums into
    //          st[SQP..NINF] and forms the next search direction p_next = z + (rho' / rho) p (:475-479); returns true.
    //   reduce_sums (with p_next == null): the last workgroup to finish reduces the five partial sums into st[SQP..NINF] itself
    //          (reduce_update()'s fixed order): no reduce launch before the all-reduce
    //   mail (with reduce_sums): ... and all-reduces them across the ranks through the mailboxes: st[SQP..NINF] and its mirror hold
    //          the GLOBAL sums when the launch is done
    virtual bool vec_update(double *st, int pq_src, const double *p, const double *q, const double *r_old, double *r_new,
                            const double *x_old, double *x_new, const double *minv, double *p_next, bool reduce_sums = false,
                            const MailDesc *mail = nullptr) = 0;
    virtual bool mailbox_kernels_available() const { return false; }       // the two launches above can take `mail`
    // the multi-part loop may fold pack / the two reductions / the status copy into the neighbouring launches (PCG_ITER_FUSED=0: no)
    virtual bool iteration_fusion_available() const { return false; }
    virtual bool vec_fused_available() const { return false; }
    // the fused launch reported a grid-barrier time-out (st[ERR]): clear the report (block and host mirror) and keep to the split
    // form from now on (this engine; reload_tuning() does not bring the fused form back)
    virtual void vec_fused_failed() {}
    virtual void reduce_update(double *red5) = 0;
    // r = b - ax ; sums r^2 w, (M^-1 r) r w, inf count                      (:413-416,:530-533)
    virtual void residual(const double *b, const double *ax, double *r, const double *minv) = 0;
    virtual void reduce_residual(double *red3) = 0;
    virtual void dot_w(const double *a, const double *b) = 0;             // sum a*b*w
    virtual void reduce_dotw(double *red1) = 0;
    virtual void copy_diag(double *d) = 0;                                // local diag(A)
    virtual void invert_free(double *minv, const double *d) = 0;          // free ? 1/d : 0   (:351-352)
    virtual void axpby(double *out, double a, const double *x, double b, const double *y) = 0;  // a*x + b*y
    virtual void scale(double *out, double a, const double *x) = 0;
    virtual void mask_free(double *x) = 0;                                // zero the fixed dofs
    // profiling of the SpMV launches with events on the stream
    virtual void set_profiling(int what) = 0;              // bit 0: events around the operator launches, bit 1: around vec_update
    virtual void collect_profile(double *ms_sum, int64_t *count) = 0;
    virtual void collect_profile_vec(double *ms_sum, int64_t *count) { *ms_sum = 0; *count = 0; }   // the vec_update launches
    virtual int bench_spmv(const double *x, double *y, int warmup, int reps, float *ms_each) = 0;
    // Round 6: choose, by timing a few applies on the vectors a solve will use, between forms of the operator that produce the same bits
    // (HIP: k_spmv in one launch or in several that write their y at their end).  -> launches per apply of the form kept.
    virtual int tune_operator(const double *x, double *y) { (void)x; (void)y; return 1; }
    virtual int operator_launches_per_apply() const { return 1; }
    // stream microbenchmark over `bytes` of device memory: mode 0 read-only, 1 copy (read + write); ms per repetition
    virtual int bench_hbm(size_t bytes, int mode, int reps, float *ms_each) = 0;
};

// ---- native inter-GPU communication ---------------------------------------------------------------
// One process per GPU, one part per process (pcg_solver.py:91).  The product implements this with RCCL
// calls issued by the engine itself (rccl_comm.hip): the interface exchange Isend/Recv/Waitall (:318-328)
// is ONE group of ncclSend/ncclRecv per neighbour on a dedicated communication stream, fenced against
// the compute stream with events; MPI_SUM (:622-628) is ncclAllReduce(ncclDouble, ncclSum) in place on
// the device status block, on the compute stream.  No Python frame, no host copy.
struct CommStats {
    double halo_wait_ms = 0, allreduce_ms = 0;      // GPU-side: compute stream stalled for the exchange / inside all-reduce
    int64_t n_halo = 0, n_allreduce = 0;            // calls issued
    int64_t n_halo_timed = 0, n_allreduce_timed = 0;
};
class Comm {
public:
    virtual ~Comm() {}
    virtual int rank() const = 0;
    virtual int size() const = 0;
    // `send` has been packed on `compute_stream`; starts the exchange with every neighbour of `h` on the comm stream
    virtual void halo_begin(double *send, double *recv, const HaloHost &h, void *compute_stream) = 0;
    // makes `compute_stream` wait until `recv` is complete (and `send` may be overwritten)
    virtual void halo_end(void *compute_stream) = 0;
    virtual void allreduce(double *buf, int count, void *compute_stream) = 0;
    virtual void set_timing(bool on) = 0;           // HIP events around the waits (the reference's dT_CommWait, :631-641)
    virtual CommStats stats() = 0;                  // synchronises the streams it reads events from
    // Mailbox all-reduce (MailDesc above).  enable_mailbox() is COLLECTIVE: the ranks exchange their mailboxes' handles, map them,
    // run one all-reduce of known values through them and agree on the outcome - false (on every rank) when any rank could not
    // map a peer (no peer access, IPC refused, more than kMailMaxRanks ranks): allreduce() keeps using the collective library.
    virtual bool enable_mailbox(bool on) { (void)on; return false; }
    virtual bool mailbox_enabled() const { return false; }
    virtual MailDesc mailbox_next() { return MailDesc{}; }   // the descriptor of the NEXT all-reduce (advances the sequence)
    virtual void mailbox_check() {}                          // throws when a poll has timed out since the last call
    virtual std::string mailbox_why() const { return "this communicator has no mailbox all-reduce"; }   // why enable_mailbox() said no
    // COLLECTIVE, at the start of every solve while an engine-side form is on (round 6): one all-reduce of the collective
    // library on `compute_stream` in front of the solve's first engine-side wait - the ranks' polling kernels then start within
    // microseconds of each other however uneven the set-up was - carrying "a poll of mine has timed out since the last call" (mailbox
    // or `link_fault`).  -> true on EVERY rank when any rank reported one: the mailbox is switched off here and the caller drops its
    // direct link, i.e. the ranks return to ncclAllReduce / ncclSend / ncclRecv TOGETHER instead of staying on sequence numbers that
    // no longer agree.  `engine_side_on` must be the same on every rank (enable_mailbox / direct_link are collective).
    virtual bool engine_side_sync(void *compute_stream, bool engine_side_on, bool link_fault) { (void)compute_stream; (void)engine_side_on; (void)link_fault; return false; }
    // Direct exchange (DirectDesc above).  COLLECTIVE over every rank of the communicator, neighbours or not: the ranks publish
    // their buffers' handles and segment layouts, map their neighbours' and agree on the outcome.  -> null (on every rank, `why`
    // says why) when any rank could not map a neighbour; the exchange then stays on ncclSend / ncclRecv.
    // `cannot`: this rank cannot use a link whatever the mapping says (its back end has no direct kernels; `why` says so): it still
    // takes part in the agreement, which then fails on every rank.
    virtual std::unique_ptr<DirectLink> direct_link(const HaloHost &h, std::string &why, bool cannot = false) { (void)h; (void)cannot; why = "this communicator has no direct exchange"; return nullptr; }
};
// defined by the HIP side of the product library; the CPU test double has no native communicator
std::unique_ptr<Comm> make_rccl_comm(int device, int rank, int nranks, const void *unique_ids /* 2 x 128 B */);
int rccl_unique_ids(void *out /* 2 x 128 B */);

// ---- partition set-up on the device (part_setup.hip; the CPU test double has none) ---------------------------------
// -> number of (node, part) pairs found (pairs[2i], pairs[2i+1]; only the first `cap` are written)
int64_t part_interface(int device, int64_t n_glob_nodes, int64_t n_elem, const int64_t *elem_ptr, const int32_t *flat_nodes,
                       const int32_t *ele_part, int64_t cap, int64_t *pairs);
// -> number of distinct nodes; unique_nodes ascending, local_of_flat[i] = index of flat_nodes[i] in it
int64_t part_local_numbering(int device, int64_t n_glob_nodes, int64_t n_flat, const int32_t *flat_nodes, int32_t *unique_nodes,
                             int32_t *local_of_flat);

std::unique_ptr<Backend> make_backend(int device);   // defined by exactly one back end per library
int backend_device_count();
const char *backend_static_name();

}  // namespace pcg
