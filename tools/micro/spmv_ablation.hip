// Micro-benchmark (development, round 6): WHERE does k_spmv lose against its own value stream?  The 10 M-dof brick's SpMV reads 6.5 GB
// of values; that stream alone runs at 7.15 TB/s (0.92 ms) on boxes where the kernel takes 1.20 ms, and at the same speed on boxes
// where it takes 1.03 ms (tools/micro/stream_patterns).  This program rebuilds the kernel's loop on a synthetic matrix with the brick's
// real structure (N^3 nodes, 27-point block stencil clamped at the faces, 64 consecutive nodes per slice, 16-bit column offsets from the
// slice's smallest column) and switches its parts on one by one:
//   A  values only                      (the stream)
//   B  A + the column loads             (3 x 128 B per group, the loaded values are consumed)
//   C  A + x gathers at COMPUTED columns (no column load: the gather's own cost, independent of any load)
//   D  B + x gathers at the LOADED columns (the dependent chain) + the 81 fused multiply-adds per group
//   E  D + y stores                      (the kernel without its dot product)
//   F  E with the x gathers served from a 1/8-size x (every gather hits L2: what the kernel would do if x never missed)
//   G  E with non-temporal stores     H  y written as three fully coalesced 8-B stores per lane (through lane shuffles)     I  H non-temporal
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/spmv_ablation.hip -o tools/micro/spmv_ablation && tools/micro/spmv_ablation [N=150]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 27;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef v2d v2d_a8 __attribute__((aligned(8)));
__device__ __forceinline__ double ntl(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ntl(const unsigned short *p) { return (int)__builtin_nontemporal_load(p); }

// columns of the brick: node n = (kz * N + jy) * N + ix; neighbour k = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1), clamped into the grid
__device__ __forceinline__ int col_of(long long n, int k, int N)
{
    const int ix = (int)(n % N), jy = (int)((n / N) % N), kz = (int)(n / ((long long)N * N));
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    const int x = min(max(ix + dx, 0), N - 1), y = min(max(jy + dy, 0), N - 1), z = min(max(kz + dz, 0), N - 1);
    return (z * N + y) * N + x;
}
__global__ void k_build(unsigned short *cols, int *colbase, long long n_nodes, long long n_slices, int N)
{
    const long long s = blockIdx.x;
    const int lane = threadIdx.x;
    if (s >= n_slices) return;
    long long first = min(s * 64, n_nodes - 1);
    const int cb = col_of(first, 0, N);                                     // the slice's smallest column: first row, first neighbour
    if (lane == 0) colbase[s] = cb;
    const long long n = min(s * 64 + lane, n_nodes - 1);
    for (int k = 0; k < W; ++k) cols[(s * W + k) * 64 + lane] = (unsigned short)(col_of(n, k, N) - cb);
}

__global__ void k_fill(double *a, size_t n, unsigned long long seed)      // random mantissas (not zeros: data-dependent power, "DVFS-honest")
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        a[i] = (double)(z >> 11) / 9007199254740992.0 - 0.5;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const double *__restrict__ vals, const unsigned short *__restrict__ cols, const int *__restrict__ colbase,
                                         const double *__restrict__ x, double *__restrict__ y, long long n_slices, long long n_nodes, int N, long long xmask, long long s_lo = 0)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long G = (long long)gridDim.x * 4;
    double sink = 0.0;
    constexpr int HOLD = 6;                                      // O: slices a wave keeps in LDS before it writes them (at the END of the launch)
    __shared__ double held[MODE == 14 ? 4 * HOLD * 192 : 1];
    int n_held = 0;
    long long held_s[HOLD];
    for (long long s = s_lo + (long long)blockIdx.x * 4 + wid; s < n_slices; s += G) {
        const double *vp = vals + s * W * 9 * 64 + lane;
        const unsigned short *cp = cols + s * W * 64 + lane;
        const int cb = MODE == 1 || MODE >= 3 ? colbase[s] : 0;
        const long long n = min(s * 64 + lane, n_nodes - 1);
        int ix = 0, jy = 0, kz = 0;
        if (MODE == 2) { ix = (int)(n % N); jy = (int)((n / N) % N); kz = (int)(n / ((long long)N * N)); }      // (once per slice, not per column)
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll 3
        for (int kk = 0; kk < W; ++kk) {
            int j = 0;
            if (MODE == 1 || MODE >= 3) j = ntl(cp + (long long)kk * 64) + cb;
            if (MODE == 2) {
                const int dx = kk % 3 - 1, dy = (kk / 3) % 3 - 1, dz = kk / 9 - 1;      // (compile-time after unrolling)
                j = (min(max(kz + dz, 0), N - 1) * N + min(max(jy + dy, 0), N - 1)) * N + min(max(ix + dx, 0), N - 1);
            }
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = ntl(vp + ((long long)kk * 9 + c) * 64);
            if (MODE == 0) { acc[0] += v[0] + v[3] + v[6]; acc[1] += v[1] + v[4] + v[7]; acc[2] += v[2] + v[5] + v[8]; }
            else if (MODE == 1) { acc[0] += v[0] + v[3] + v[6] + (double)j; acc[1] += v[1] + v[4] + v[7]; acc[2] += v[2] + v[5] + v[8]; }
            else {
                const double *xp = x + 3 * ((long long)j & xmask);
                const v2d_a8 x01 = *reinterpret_cast<const v2d_a8 *>(xp);
                const double x0 = x01.x, x1 = x01.y, x2 = xp[2];
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = fma(v[3 * a + 2], x2, fma(v[3 * a + 1], x1, fma(v[3 * a], x0, acc[a])));
            }
        }
        if (MODE == 4 || MODE == 5) {                             // E / F: the kernel's stores: 16 B + 8 B per lane, 24 contiguous bytes
            const long long row = s * 64 + lane;
            if (row < n_nodes) { double *yp = y + 3 * row; yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2]; }
        } else if (MODE == 6) {                                   // G: the same, non-temporal
            const long long row = s * 64 + lane;
            if (row < n_nodes) {
                double *yp = y + 3 * row;
                __builtin_nontemporal_store(acc[0], yp); __builtin_nontemporal_store(acc[1], yp + 1); __builtin_nontemporal_store(acc[2], yp + 2);
            }
        } else if (MODE == 7 || MODE == 8) {                      // H / I: three fully coalesced 8-B stores per lane (512 contiguous bytes per instruction)
            double *yb = y + 3 * s * 64;                          //        element i = 64 t + lane of the slice's 192 comes from lane i / 3, component i % 3
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int i = 64 * t + lane, src = i / 3, comp = i - 3 * src;
                const double a0 = __shfl(acc[0], src), a1 = __shfl(acc[1], src), a2 = __shfl(acc[2], src);
                const double v = comp == 0 ? a0 : (comp == 1 ? a1 : a2);
                if (s * 64 + src < n_nodes) {
                    if (MODE == 8) __builtin_nontemporal_store(v, yb + i); else yb[i] = v;
                }
            }
        } else if (MODE >= 9) {                                   // J..N: y staged through LDS, written as coalesced 16-B stores of a given cache policy
            __shared__ double stage[4][192];
            double *st = stage[wid];
            st[3 * lane] = acc[0]; st[3 * lane + 1] = acc[1]; st[3 * lane + 2] = acc[2];       // (one wave: no barrier needed, the LDS unit keeps a wave's order)
            __builtin_amdgcn_s_waitcnt(0xc07f);                                                 // lgkmcnt(0)
            double *yb = y + 3 * s * 64;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e = 128 * t + 2 * lane;                                               // element pair (e, e + 1) of the slice's 192
                if (e < 192 && s * 64 + e / 3 < n_nodes) {
                    const v2d v = *reinterpret_cast<const v2d *>(st + e);
                    double *q = yb + e;
                    if (MODE == 9) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q), "v"(v) : "memory");
                    if (MODE == 10) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
                    if (MODE == 11) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
                    if (MODE == 12) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q), "v"(v) : "memory");
                    if (MODE == 13) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(q), "v"(v) : "memory");
                }
            }
        } else if (MODE == 14) {                                  // O: keep the slice in LDS; written after the loop
            if (n_held < HOLD) {
                double *st = held + ((size_t)wid * HOLD + n_held) * 192;
                st[3 * lane] = acc[0]; st[3 * lane + 1] = acc[1]; st[3 * lane + 2] = acc[2];
                held_s[n_held++] = s;
            } else {                                              // (more slices than slots: the plain stores)
                const long long row = s * 64 + lane;
                if (row < n_nodes) { double *yp = y + 3 * row; yp[0] = acc[0]; yp[1] = acc[1]; yp[2] = acc[2]; }
            }
        } else sink += acc[0] + acc[1] + acc[2];
    }
    if (MODE == 14) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int h = 0; h < HOLD; ++h)
            if (h < n_held) {
                const double *st = held + ((size_t)wid * HOLD + h) * 192;
                double *yb = y + 3 * held_s[h] * 64;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int e = 128 * t + 2 * lane;
                    if (e < 192 && held_s[h] * 64 + e / 3 < n_nodes) *reinterpret_cast<v2d *>(yb + e) = *reinterpret_cast<const v2d *>(st + e);
                }
            }
    }
    if (sink == 1.2345e-300) y[0] = sink;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 150, bpc = argc > 2 ? atoi(argv[2]) : 4;
    const bool rnd = argc > 3 && atoi(argv[3]) != 0;            // 1: random values and x instead of zeros
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const long long n_nodes = (long long)N * N * N, n_slices = (n_nodes + 63) / 64;
    const size_t vbytes = (size_t)n_slices * W * 9 * 64 * sizeof(double), cbytes = (size_t)n_slices * W * 64 * sizeof(unsigned short);
    double *vals, *x, *y; unsigned short *cols; int *colbase;
    CHECK(hipMalloc(&vals, vbytes)); CHECK(hipMemset(vals, 0, vbytes));
    CHECK(hipMalloc(&cols, cbytes)); CHECK(hipMalloc(&colbase, n_slices * sizeof(int)));
    CHECK(hipMalloc(&x, 3 * n_nodes * sizeof(double) + 64)); CHECK(hipMemset(x, 0, 3 * n_nodes * sizeof(double) + 64));
    CHECK(hipMalloc(&y, 3 * n_nodes * sizeof(double) + 64));
    if (rnd) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, vals, vbytes / sizeof(double), 1ull);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, (size_t)3 * n_nodes, 77ull);
    }
    printf("operands: %s\n", rnd ? "random" : "zeros");
    hipLaunchKernelGGL(k_build, dim3((unsigned)n_slices), dim3(64), 0, 0, cols, colbase, n_nodes, n_slices, N);
    CHECK(hipDeviceSynchronize());
    const int grid = p.multiProcessorCount * bpc;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%s, %d CUs, grid %d x 256, N = %d: %lld nodes, %lld slices, values %.2f GB, columns %.0f MB, x %.0f MB\n", p.name, p.multiProcessorCount, grid, N,
           n_nodes, n_slices, vbytes / 1e9, cbytes / 1e6, 3 * n_nodes * 8 / 1e6);
    const char *names[16] = {"A values", "B +cols", "C +x(computed)", "D +cols+x+fma", "E +y store", "F E, x in L2", "G E, nt stores", "H coalesced y", "I coalesced nt",
                              "J 16B coalesced", "K 16B sc1", "L 16B sc0 sc1", "M 16B nt", "N 16B sc1 nt", "O y at the end x3", "P y at the end x4"};
    long long small = 1;
    while (small * 2 <= n_nodes / 8) small *= 2;                 // F: columns masked into a power-of-two eighth of x
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 16; ++mode) {
            std::vector<float> ms;
            for (int it = 0; it < 12; ++it) {
                const long long xmask = mode == 5 ? small - 1 : ~0ll;
                CHECK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 11: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 12: hipLaunchKernelGGL(k<12>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 13: hipLaunchKernelGGL(k<13>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                case 14: case 15: {                                          // the slices in 3 (4) launches, every launch writes its y at its end
                    const int parts = mode == 14 ? 3 : 4;
                    for (int q = 0; q < parts; ++q) {
                        const long long lo = n_slices * q / parts, hi = n_slices * (q + 1) / parts;
                        hipLaunchKernelGGL(k<14>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, hi, n_nodes, N, xmask, lo);
                    }
                    break;
                }
                case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, vals, cols, colbase, x, y, n_slices, n_nodes, N, xmask); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float t; CHECK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("rep %d %-16s median %.4f ms (min %.4f) = %.0f GB/s of values\n", rep, names[mode], ms[ms.size() / 2], ms[0], vbytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
        }
    return 0;
}
