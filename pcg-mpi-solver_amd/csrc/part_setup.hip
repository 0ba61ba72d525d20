// Partition set-up on the device: the whole-model index passes of the reference's partitioner as
// HIP kernels, so that a rank of a multi-GPU job does not walk / sort the global element -> node list on the host.
//
//   reference (src/solver/partition_mesh.py)                        here
//   --------------------------------------------------------------  ----------------------------------------------
//   identify_PotentialNeighbours :657-741 + config_Neighbours        pcg_part_interface: scatter one part id per global
//   :745-830: bounding boxes -> candidate parts -> np.intersect1d    node, mark the nodes some element of ANOTHER part
//   of sorted node-id lists, per pair of parts                        touches, emit their (node, part) pairs
//   config_ElemVectors :252-268: np.unique of the part's node ids    pcg_part_local_numbering: mark bitmap -> exclusive
//   + getIndices (a dict / searchsorted lookup per element node)      scan over the global node range -> ascending unique
//                                                                     list + local index of every element node
//
// All of it is integer work: results are exact, and they are made deterministic where atomics order the output (the
// emitted pairs are sorted by the caller; everything else is order-free by construction).  Memory-bound streaming
// kernels: coalesced reads of the flat lists, random 4-byte scatter / gather into the per-node tables (which fit the
// 256 MB MALL up to ~60 M nodes), a three-pass block scan with 16 B per lane.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcg_internal.hpp"

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw std::runtime_error(std::string(#expr) + " -> " + hipGetErrorString(_e));               \
    } while (0)

namespace pcg {
namespace {

constexpr int kB = 256;

// ---- interface discovery -------------------------------------------------------------------------------------
// rec[node] = part id of SOME element that touches the node (any writer may win: only agreement is tested afterwards)
__global__ __launch_bounds__(kB) void k_part_scatter(const long long *__restrict__ elem_ptr, const int *__restrict__ flat,
                                                     const int *__restrict__ ele_part, long long n_elem, int *__restrict__ rec)
{
    const long long e = blockIdx.x * (long long)kB + threadIdx.x;
    if (e >= n_elem) return;
    const int p = ele_part[e];
    for (long long k = elem_ptr[e]; k < elem_ptr[e + 1]; ++k) rec[flat[k]] = p;
}

// a node is on an interface iff some element touching it belongs to a part other than the recorded one
__global__ __launch_bounds__(kB) void k_part_mark(const long long *__restrict__ elem_ptr, const int *__restrict__ flat,
                                                  const int *__restrict__ ele_part, long long n_elem, const int *__restrict__ rec,
                                                  unsigned char *__restrict__ on_if)
{
    const long long e = blockIdx.x * (long long)kB + threadIdx.x;
    if (e >= n_elem) return;
    const int p = ele_part[e];
    for (long long k = elem_ptr[e]; k < elem_ptr[e + 1]; ++k)
        if (rec[flat[k]] != p) on_if[flat[k]] = 1;
}

// (node, part) of every element node on an interface; duplicates and order are the caller's to remove (sort + unique)
__global__ __launch_bounds__(kB) void k_part_emit(const long long *__restrict__ elem_ptr, const int *__restrict__ flat,
                                                  const int *__restrict__ ele_part, long long n_elem,
                                                  const unsigned char *__restrict__ on_if, long long cap,
                                                  long long *__restrict__ pairs, unsigned long long *__restrict__ counter)
{
    const long long e = blockIdx.x * (long long)kB + threadIdx.x;
    if (e >= n_elem) return;
    const int p = ele_part[e];
    for (long long k = elem_ptr[e]; k < elem_ptr[e + 1]; ++k) {
        const int nd = flat[k];
        if (on_if[nd]) {
            const unsigned long long i = atomicAdd(counter, 1ull);
            if ((long long)i < cap) { pairs[2 * i] = nd; pairs[2 * i + 1] = p; }
        }
    }
}

// ---- local numbering ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kB) void k_mark_nodes(const int *__restrict__ flat, long long n, int *__restrict__ mark)
{
    for (long long i = blockIdx.x * (long long)kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB) mark[flat[i]] = 1;
}

// exclusive scan of `mark` (0/1 ints) over n entries, in place, three passes; tile = 1024 entries per block
constexpr int kTile = 4 * kB;
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds /* kB/64 */, int &total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kB / 64; ++w) { if (w < wid) base += lds[w]; tot += lds[w]; }
    __syncthreads();
    total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(kB) void k_scan_tiles(const int *__restrict__ mark, long long n, int *__restrict__ tile_sum)
{
    __shared__ int lds[kB / 64];
    const long long i0 = (long long)blockIdx.x * kTile + 4 * threadIdx.x;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) s += mark[i0 + k];
    int total;
    (void)block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(kB) void k_scan_tile_sums(int *__restrict__ tile_sum, long long n_tiles, long long *__restrict__ grand_total)
{
    __shared__ int lds[kB / 64];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long t0 = 0; t0 < n_tiles; t0 += kB) {
        const long long t = t0 + threadIdx.x;
        const int v = t < n_tiles ? tile_sum[t] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds, total);
        if (t < n_tiles) tile_sum[t] = (int)(carry + ex);
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

// mark -> local index (exclusive prefix) for marked nodes, and the ascending list of marked nodes
__global__ __launch_bounds__(kB) void k_scan_apply(int *__restrict__ mark, long long n, const int *__restrict__ tile_sum,
                                                   int *__restrict__ unique_nodes)
{
    __shared__ int lds[kB / 64];
    const long long i0 = (long long)blockIdx.x * kTile + 4 * threadIdx.x;
    int m[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { m[k] = i0 + k < n ? mark[i0 + k] : 0; s += m[k]; }
    int total;
    int pos = tile_sum[blockIdx.x] + block_exclusive_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) {
            if (m[k]) { unique_nodes[pos] = (int)(i0 + k); mark[i0 + k] = pos; ++pos; }
            else mark[i0 + k] = -1;
        }
}

__global__ __launch_bounds__(kB) void k_gather_local(const int *__restrict__ flat, long long n, const int *__restrict__ local_of_node,
                                                     int *__restrict__ local_of_flat)
{
    for (long long i = blockIdx.x * (long long)kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB) local_of_flat[i] = local_of_node[flat[i]];
}

struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) { HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() { return (T *)p; }
};

void use_device(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) throw std::runtime_error("no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= cnt) throw std::runtime_error("device index out of range");
    HIP_CHECK(hipSetDevice(device));
}

}  // namespace

int64_t part_interface(int device, int64_t n_glob_nodes, int64_t n_elem, const int64_t *elem_ptr, const int32_t *flat_nodes,
                       const int32_t *ele_part, int64_t cap, int64_t *pairs)
{
    use_device(device);
    const int64_t n_flat = elem_ptr[n_elem];
    DevBuf d_ptr(sizeof(int64_t) * (n_elem + 1)), d_flat(sizeof(int32_t) * n_flat), d_part(sizeof(int32_t) * n_elem);
    DevBuf d_rec(sizeof(int32_t) * n_glob_nodes), d_if(n_glob_nodes), d_pairs(sizeof(int64_t) * 2 * (cap > 0 ? cap : 1)), d_cnt(8);
    HIP_CHECK(hipMemcpy(d_ptr.p, elem_ptr, sizeof(int64_t) * (n_elem + 1), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_flat.p, flat_nodes, sizeof(int32_t) * n_flat, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_part.p, ele_part, sizeof(int32_t) * n_elem, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(d_rec.p, 0xff, sizeof(int32_t) * n_glob_nodes));
    HIP_CHECK(hipMemset(d_if.p, 0, n_glob_nodes));
    HIP_CHECK(hipMemset(d_cnt.p, 0, 8));
    const int grid = (int)((n_elem + kB - 1) / kB);
    hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(kB), 0, 0, d_ptr.as<long long>(), d_flat.as<int>(), d_part.as<int>(), (long long)n_elem, d_rec.as<int>());
    hipLaunchKernelGGL(k_part_mark, dim3(grid), dim3(kB), 0, 0, d_ptr.as<long long>(), d_flat.as<int>(), d_part.as<int>(), (long long)n_elem, d_rec.as<int>(),
                       d_if.as<unsigned char>());
    hipLaunchKernelGGL(k_part_emit, dim3(grid), dim3(kB), 0, 0, d_ptr.as<long long>(), d_flat.as<int>(), d_part.as<int>(), (long long)n_elem,
                       d_if.as<unsigned char>(), (long long)cap, d_pairs.as<long long>(), d_cnt.as<unsigned long long>());
    HIP_CHECK(hipGetLastError());
    unsigned long long n = 0;
    HIP_CHECK(hipMemcpy(&n, d_cnt.p, 8, hipMemcpyDeviceToHost));
    const int64_t got = (int64_t)n < cap ? (int64_t)n : cap;
    if (got > 0) HIP_CHECK(hipMemcpy(pairs, d_pairs.p, sizeof(int64_t) * 2 * got, hipMemcpyDeviceToHost));
    return (int64_t)n;                       // > cap: the caller retries with that capacity
}

int64_t part_local_numbering(int device, int64_t n_glob_nodes, int64_t n_flat, const int32_t *flat_nodes, int32_t *unique_nodes,
                             int32_t *local_of_flat)
{
    use_device(device);
    const int64_t n_tiles = (n_glob_nodes + kTile - 1) / kTile;
    DevBuf d_flat(sizeof(int32_t) * n_flat), d_mark(sizeof(int32_t) * n_glob_nodes), d_tiles(sizeof(int32_t) * (n_tiles + 1));
    DevBuf d_unique(sizeof(int32_t) * (n_flat < n_glob_nodes ? n_flat : n_glob_nodes)), d_local(sizeof(int32_t) * n_flat), d_tot(8);
    HIP_CHECK(hipMemcpy(d_flat.p, flat_nodes, sizeof(int32_t) * n_flat, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(d_mark.p, 0, sizeof(int32_t) * n_glob_nodes));
    const int grid_f = (int)std::min<int64_t>((n_flat + kB - 1) / kB, 65535 * 4);
    hipLaunchKernelGGL(k_mark_nodes, dim3(grid_f), dim3(kB), 0, 0, d_flat.as<int>(), (long long)n_flat, d_mark.as<int>());
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)n_tiles), dim3(kB), 0, 0, d_mark.as<int>(), (long long)n_glob_nodes, d_tiles.as<int>());
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kB), 0, 0, d_tiles.as<int>(), (long long)n_tiles, d_tot.as<long long>());
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles), dim3(kB), 0, 0, d_mark.as<int>(), (long long)n_glob_nodes, d_tiles.as<int>(), d_unique.as<int>());
    hipLaunchKernelGGL(k_gather_local, dim3(grid_f), dim3(kB), 0, 0, d_flat.as<int>(), (long long)n_flat, d_mark.as<int>(), d_local.as<int>());
    HIP_CHECK(hipGetLastError());
    long long n_unique = 0;
    HIP_CHECK(hipMemcpy(&n_unique, d_tot.p, 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(unique_nodes, d_unique.p, sizeof(int32_t) * n_unique, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(local_of_flat, d_local.p, sizeof(int32_t) * n_flat, hipMemcpyDeviceToHost));
    return (int64_t)n_unique;
}

}  // namespace pcg
