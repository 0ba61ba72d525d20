// Micro-benchmark (development): what does one ds_add_f64 cost on gfx950 - per instruction or per active lane - and what does a
// ds_read_b64 / v_add_f64 / ds_write_b64 sequence cost for the same update?  The ordered LDS sums of k_ebe_hexs / k_ebe_mixed
// (24 adds per element, colour by colour) sit on the critical path of a chunk.   hipcc --offload-arch=gfx950 -O3 lds_atomic_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>     // 0: ds_add_f64, 1: read - add - write, 2: ds_add_f64, two accumulators interleaved (independent addresses)
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *ticks, int active, int reps, int stride)
{
    __shared__ double ys[2304];
    for (int i = threadIdx.x; i < 2304; i += 256) ys[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 8 "nodes" per lane, 3 doubles each; lanes of a wave touch distinct nodes (conflict-free as inside one colour)
    int a[8];
    for (int k8 = 0; k8 < 8; ++k8) a[k8] = 3 * ((wave * 64 + lane) * stride % 96 + 96 * k8);
    double v[24];
    for (int q = 0; q < 24; ++q) v[q] = 1.0 + q + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < active)
        for (int r = 0; r < reps; ++r) {
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int q = 0; q < 24; ++q) __hip_atomic_fetch_add(&ys[a[q / 3] + q % 3], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                double o[24];
#pragma unroll
                for (int q = 0; q < 24; ++q) o[q] = ys[a[q / 3] + q % 3];
#pragma unroll
                for (int q = 0; q < 24; ++q) ys[a[q / 3] + q % 3] = o[q] + v[q];
            }
            asm volatile("" ::: "memory");
        }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
    if (threadIdx.x == 0) out[blockIdx.x] = ys[0] + ys[100];
}

int main()
{
    const int reps = 200;
    double *out; unsigned long long *ticks;
    CHECK(hipMalloc(&out, 4096 * sizeof(double))); CHECK(hipMalloc(&ticks, 4096 * 4 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(4096 * 4);
    auto run = [&](int mode, int blocks, int active, const char *what) {
        for (int it = 0; it < 2; ++it) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, ticks, active, reps, 1);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, ticks, active, reps, 1);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), ticks, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
        printf("%-22s blocks %5d (%.1f per CU)  active lanes %2d : %8.1f ticks per batch of 24 updates per wave, %6.1f per instruction-equivalent\n", what, blocks,
               blocks / 256.0, active, s / (blocks * 4) / reps, s / (blocks * 4) / reps / 24);
    };
    for (int blocks : {256, 1024})
        for (int active : {64, 32, 16, 8}) {
            run(0, blocks, active, "ds_add_f64");
            run(1, blocks, active, "read-add-write");
        }
    return 0;
}
