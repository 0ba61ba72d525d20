// Shared device helpers of the HIP back end (gfx950): constants, error checking, wave / workgroup reductions,
// non-temporal load / store wrappers.  Included by the kernel headers and hip_backend.hip (one translation unit).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "pcg_internal.hpp"

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw std::runtime_error(std::string(#expr) + " -> " + hipGetErrorString(_e));               \
    } while (0)

namespace pcg {

constexpr int kBlock = 256;            // 4 wave64 per workgroup
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxPartials = 4096;     // upper bound on blocks that write a partial

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;                           // valid in lane 0
}

// block-level sum of NV per-thread values; result valid in thread 0.  Fixed order -> deterministic.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *lds /* NV * kWavesPerBlock */)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = wave_sum(v[k]);
        if (lane == 0) lds[k * kWavesPerBlock + wid] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = lds[k * kWavesPerBlock];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) s += lds[k * kWavesPerBlock + w];
            v[k] = s;
        }
    }
}

template <int RPL> struct VecT;
template <> struct VecT<1> { using d = double; using i = int; };
template <> struct VecT<2> { using d = double2; using i = int2; };

// a 16-byte pair of doubles that is only 8-byte aligned: gfx950 global loads need dword alignment only, so the three
// components of a node (24 contiguous bytes at 24 j) are one global_load_dwordx4 + one dwordx2 instead of three dwordx2 -
// the L1 (TCP) takes 16 cycles per wave-level load instruction whatever its width (profiles/r03_pmc_*: 64 'cache accesses' per
// four 64-lane loads), and the dictionary SpMV is bound by exactly that
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef v2d_t v2d_a8 __attribute__((aligned(8)));

__device__ __forceinline__ double ntload(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ntload(const int *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ unsigned ntload(const unsigned *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double2 ntload(const double2 *p)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(p));
    return make_double2(t.x, t.y);
}
__device__ __forceinline__ void ntstore(double2 *p, double2 v)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<v2 *>(p));
}
__device__ __forceinline__ int2 ntload(const int2 *p)
{
    typedef int v2 __attribute__((ext_vector_type(2)));
    v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(p));
    return make_int2(t.x, t.y);
}

__device__ __forceinline__ int ntload(const unsigned short *p) { return (int)__builtin_nontemporal_load(p); }
__device__ __forceinline__ int2 ntload(const ushort2 *p)
{
    const unsigned t = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p));
    return make_int2((int)(t & 0xffffu), (int)(t >> 16));
}

}  // namespace pcg
