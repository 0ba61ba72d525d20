// Micro-benchmark (development, round 6): what THIS box's HBM gives the VALUE STREAM of k_spmv depending on how the chip's concurrent
// waves are laid over the array.  The SpMV of the 10 M-dof brick reads 6.5 GB of values once: 52 735 slices of 27 block columns,
// a block column of a slice = 9 x 512 B contiguous (4 608 B), a slice = 124 416 B contiguous; wave g of the grid takes the slices
// g, g + G, ... - i.e. ~4 000 concurrent sequential streams one slice apart.  Boxes of this pool run that launch at 1.03 - 1.06 ms or at
// 1.18 - 1.22 ms with the SAME plain read stream (6.7 TB/s): which pattern does the slow kind dislike?
//   linear  : the grid sweeps the array (wave g reads chunk t * G + g of 512 B): what pcg_bench_hbm's read mode does, 8 B per lane
//   slices  : k_spmv's pattern (wave g reads slice g, g + G, ...: 27 steps of 9 x 512 B)
//   blocked : wave g reads a contiguous RUN of slices (streams ~1.6 MB apart)
//   stepmaj : the array regrouped k-major inside groups of G slices: at step k the waves of a round read ADJACENT 4 608-B chunks
//             (a linear sweep of G x 4 608 B per step) - what a "grouped" SELL layout would give
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_patterns.hip -o tools/micro/stream_patterns && tools/micro/stream_patterns [blocks_per_cu=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 27;                      // block columns per slice
constexpr long long SLICE = 9ll * 64 * W;  // doubles per slice

__device__ __forceinline__ double ntl(const double *p) { return __builtin_nontemporal_load(p); }

// mode 0 linear, 1 slices, 2 blocked, 3 step-major
__global__ __launch_bounds__(256) void k(const double *__restrict__ a, double *out, long long n_slices, int mode)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long G = (long long)gridDim.x * 4, g = (long long)blockIdx.x * 4 + wid;
    double s0 = 0, s1 = 0, s2 = 0;
    if (mode == 0) {
        const long long chunks = n_slices * W * 9;                       // 512-B chunks
        for (long long t = g; t < chunks; t += 9 * G) {                  // nine loads in flight, G chunks apart
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = t + c * G < chunks ? ntl(a + (t + c * G) * 64 + lane) : 0.0;
            s0 += v[0] + v[3] + v[6]; s1 += v[1] + v[4] + v[7]; s2 += v[2] + v[5] + v[8];
        }
    } else if (mode == 1 || mode == 2) {
        const long long per = (n_slices + G - 1) / G;
        const long long first = mode == 1 ? g : g * per, last = mode == 1 ? n_slices : (first + per < n_slices ? first + per : n_slices);
        for (long long s = first; s < last; s += mode == 1 ? G : 1) {
            const double *vp = a + s * SLICE + lane;
#pragma unroll 3
            for (int kk = 0; kk < W; ++kk) {
                double v[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = ntl(vp + ((long long)kk * 9 + c) * 64);
                s0 += v[0] + v[3] + v[6]; s1 += v[1] + v[4] + v[7]; s2 += v[2] + v[5] + v[8];
            }
        }
    } else {
        for (long long s = g; s < n_slices; s += G) {
            const long long j = s / G, in = s - j * G;                    // group j, position in the group
            const long long gs = (j + 1) * G <= n_slices ? G : n_slices - j * G;     // slices in this group
            const double *base = a + j * G * SLICE;
#pragma unroll 3
            for (int kk = 0; kk < W; ++kk) {
                const double *vp = base + ((long long)kk * gs + in) * 9 * 64 + lane;
                double v[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) v[c] = ntl(vp + c * 64);
                s0 += v[0] + v[3] + v[6]; s1 += v[1] + v[4] + v[7]; s2 += v[2] + v[5] + v[8];
            }
        }
    }
    const double s = s0 + s1 + s2;
    if (s == 1.2345e-300) out[0] = s;
}

int main(int argc, char **argv)
{
    const int bpc = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const long long n_slices = 52735;
    const size_t bytes = (size_t)n_slices * SLICE * sizeof(double);
    double *a, *out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMemset(a, 0, bytes)); CHECK(hipMalloc(&out, 64));
    const int grid = p.multiProcessorCount * bpc;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[4] = {"linear", "slices", "blocked", "stepmaj"};
    printf("%s, %d CUs, grid %d x 256, %.2f GB\n", p.name, p.multiProcessorCount, grid, bytes / 1e9);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<float> ms;
            for (int it = 0; it < 12; ++it) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, out, n_slices, mode);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float t; CHECK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("rep %d %-8s median %.4f ms = %.0f GB/s (min %.4f)\n", rep, names[mode], ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[0]);
        }
    return 0;
}
