// Micro-benchmark (development): latency of a scalar load that hits the scalar cache (the element matrix of k_ebe_hexs reaches its FMAs
// through s_load_dwordx16 + s_waitcnt lgkmcnt(0)), alone and with every SIMD of the CU doing the same.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v8d __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const double *__restrict__ K, unsigned long long *ticks, double *out, int reps, int span)
{
    double acc = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned off = 0;
    for (int r = 0; r < reps; ++r) {
        v8d a;
        asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(K), "s"(off));
        acc += a[0] + a[7];
        off = (off + 64) % (unsigned)span;                       // 64 B further, wrapping inside `span` bytes (4.6 KB = one hex8 matrix)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main()
{
    const int reps = 2000;
    double *K, *out; unsigned long long *ticks;
    CHECK(hipMalloc(&K, 1 << 20)); CHECK(hipMemset(K, 0, 1 << 20));
    CHECK(hipMalloc(&out, 8192 * sizeof(double))); CHECK(hipMalloc(&ticks, 8192 * 4 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(8192 * 4);
    for (int span : {4608, 16384, 65536, 1 << 20})
        for (int blocks : {1, 256, 1024}) {
            for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, K, ticks, out, reps, span); CHECK(hipDeviceSynchronize()); }
            CHECK(hipMemcpy(h.data(), ticks, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
            printf("span %7d B, %4d workgroups of 4 waves: %7.1f ticks per dependent s_load_dwordx16 (incl. ~8 of loop overhead)\n", span, blocks, s / (blocks * 4) / reps);
        }
    return 0;
}
