// Development lab (not product): A/B variants of the chunked matrix-free kernel on the real host
// set-up (csrc/ebe.cpp) for a brick, timed with HIP events, every variant checked against variant 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipcg-mpi-solver_amd/csrc \
//         tools/ebe_lab.hip pcg-mpi-solver_amd/csrc/ebe.cpp pcg-mpi-solver_amd/csrc/sell.cpp -o /tmp/ebe_lab
//   /tmp/ebe_lab [N=150] [reps=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pcg_internal.hpp"

using namespace pcg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// KMODE: 0 = Ke through scalar loads (SGPR operands), 1 = Ke staged in LDS (broadcast ds_read), 2 = rows 0..11 scalar, 12..23 LDS
// YPRE : prefetch the y tile at the start (1) or read it at the end (0)
template <int EPT, int KMODE, int MAXN, int YPRE, int MINW, int RMW = 0, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void k_lab(const int *__restrict__ chunk_list, const int4 *__restrict__ hdr,
                                                   const int *__restrict__ nodes, const unsigned short *__restrict__ lid,
                                                   const double *__restrict__ ck, const unsigned *__restrict__ sgn,
                                                   const double *__restrict__ ke_col, const double *__restrict__ x,
                                                   double *__restrict__ y)
{
    constexpr int NPT = (MAXN + 255) / 256;
    constexpr int CE = 256 * EPT;
    __shared__ double xs[3 * MAXN];
    __shared__ double ys[3 * MAXN];
    __shared__ double Ks[KMODE ? 576 : 1];
    const int chunk = chunk_list[blockIdx.x];
    const int4 h = hdr[chunk];
    const int *nd = nodes + h.x;
    unsigned sg[EPT];
    double c[EPT];
    int l3[EPT][8];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const size_t t = (size_t)chunk * CE + j * 256 + threadIdx.x;
        sg[j] = __builtin_nontemporal_load(sgn + t);
        c[j] = __builtin_nontemporal_load(ck + t);
#pragma unroll
        for (int k = 0; k < 8; ++k) l3[j][k] = 3 * (int)__builtin_nontemporal_load(lid + ((size_t)chunk * 8 + k) * CE + j * 256 + threadIdx.x);
    }
    const double *K = ke_col + (size_t)h.w * 576;
    if constexpr (KMODE != 0) {
        for (int i = threadIdx.x; i < 576; i += 256) Ks[i] = K[i];
    }
    int gnode[NPT];
    double yold[YPRE ? NPT : 1][3];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = threadIdx.x + j * 256;
        gnode[j] = n < h.y ? nd[n] : -1;
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = threadIdx.x + j * 256;
        if (gnode[j] >= 0) {
            const double *xp = x + 3 * (size_t)((ABL & 4) ? 0 : gnode[j]);
            xs[3 * n] = xp[0]; xs[3 * n + 1] = xp[1]; xs[3 * n + 2] = xp[2];
            ys[3 * n] = 0.0; ys[3 * n + 1] = 0.0; ys[3 * n + 2] = 0.0;
            if constexpr (YPRE) {
                const double *yp = y + 3 * (size_t)gnode[j];
                yold[j][0] = yp[0]; yold[j][1] = yp[1]; yold[j][2] = yp[2];
            }
        }
    }
    __syncthreads();
    double acc[EPT][24];
#pragma unroll
    for (int j = 0; j < EPT; ++j)
#pragma unroll
        for (int a = 0; a < 24; ++a) acc[j][a] = 0.0;
#pragma unroll
    for (int b = 0; b < 24; ++b) {
        double u[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            double v = xs[l3[j][b / 3] + b % 3];
            if ((sg[j] >> b) & 1u) v = -v;
            u[j] = c[j] * v;
        }
        if constexpr (ABL & 1) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) acc[j][b] = u[j];
            continue;
        }
#pragma unroll
        for (int a = 0; a < 24; ++a) {
            double k;
            if constexpr (KMODE == 0) k = K[b * 24 + a];
            else if constexpr (KMODE == 1) k = Ks[b * 24 + a];
            else k = (a < 12) ? K[b * 24 + a] : Ks[b * 24 + a];
#pragma unroll
            for (int j = 0; j < EPT; ++j) acc[j][a] = fma(k, u[j], acc[j][a]);
        }
    }
    if constexpr (ABL & 2) {
        double t = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j)
#pragma unroll
            for (int a = 0; a < 24; ++a) t += acc[j][a];
        ys[threadIdx.x] = t;
        __syncthreads();
    } else
    for (int s = 0; s < h.z; ++s) {
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if ((int)(sg[j] >> 24) == s) {
                if constexpr (RMW == 0) {
#pragma unroll
                    for (int a = 0; a < 24; ++a) {
                        double o = acc[j][a];
                        if ((sg[j] >> a) & 1u) o = -o;
                        ys[l3[j][a / 3] + a % 3] += o;
                    }
                } else {                                     // the 24 targets of one element are distinct: batch the reads
                    double old[24];
#pragma unroll
                    for (int a = 0; a < 24; ++a) old[a] = ys[l3[j][a / 3] + a % 3];
#pragma unroll
                    for (int a = 0; a < 24; ++a) {
                        double o = acc[j][a];
                        if ((sg[j] >> a) & 1u) o = -o;
                        ys[l3[j][a / 3] + a % 3] = old[a] + o;
                    }
                }
            }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int n = threadIdx.x + j * 256;
        if (gnode[j] >= 0) {
            double *yp = y + 3 * (size_t)((ABL & 4) ? (blockIdx.x * 768 + n) : gnode[j]);
            if constexpr (YPRE) {
                yp[0] = yold[j][0] + ys[3 * n]; yp[1] = yold[j][1] + ys[3 * n + 1]; yp[2] = yold[j][2] + ys[3 * n + 2];
            } else {
                yp[0] += ys[3 * n]; yp[1] += ys[3 * n + 1]; yp[2] += ys[3 * n + 2];
            }
        }
    }
}

struct Dev {
    int *list[2] = {nullptr, nullptr};
    std::vector<int> list_ptr[2];
    int4 *hdr; int *nodes; unsigned short *lid; double *ck; unsigned *sgn; double *ke;
    int n_colors = 0; long n_chunks = 0;
};

template <class T> T *up(const std::vector<T> &v)
{
    T *d = nullptr;
    CK(hipMalloc(&d, sizeof(T) * std::max<size_t>(1, v.size())));
    CK(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return d;
}

static Dev upload(const EbeHost &m)
{
    const auto &C = m.chunked;
    Dev d;
    d.hdr = (int4 *)up(C.hdr); d.nodes = up(C.nodes); d.lid = up(C.lid); d.ck = up(C.ck); d.sgn = up(C.sgn); d.ke = up(C.ke_col);
    for (int ph = 0; ph < 2; ++ph) { d.list[ph] = up(C.list[ph]); d.list_ptr[ph].assign(C.list_ptr[ph].begin(), C.list_ptr[ph].end()); }
    d.n_colors = (int)C.list_ptr[1].size() - 1; d.n_chunks = C.n_chunks;
    return d;
}

template <int EPT, int KMODE, int MAXN, int YPRE, int MINW, int RMW = 0, int ABL = 0>
static float run(const Dev &d, const double *x, double *y, size_t n, int reps, hipStream_t st)
{
    auto apply = [&]() {
        CK(hipMemsetAsync(y, 0, sizeof(double) * n, st));
        for (int ph = 0; ph < 2; ++ph)
            for (size_t k = 0; k + 1 < d.list_ptr[ph].size(); ++k) {
                const int lo = d.list_ptr[ph][k], cnt = d.list_ptr[ph][k + 1] - lo;
                hipLaunchKernelGGL((k_lab<EPT, KMODE, MAXN, YPRE, MINW, RMW, ABL>), dim3(cnt), dim3(256), 0, st, d.list[ph] + lo, d.hdr, d.nodes, d.lid,
                                   d.ck, d.sgn, d.ke, x, y);
            }
    };
    for (int k = 0; k < 3; ++k) apply();
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms(reps);
    for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(a, st)); apply(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[k], a, b));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 150, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t n1 = N - 1, ne = n1 * n1 * n1, nn = (int64_t)N * N * N;
    std::vector<int64_t> dof(24 * ne);
    std::vector<uint8_t> sg(24 * ne, 0);
    std::vector<double> ck(ne), ke(576), xyz(3 * nn);
    for (int64_t e = 0; e < ne; ++e) {
        int64_t i = e % n1, j = (e / n1) % n1, k = e / (n1 * n1);
        ck[e] = (e * 2654435761u) % 7 < 3 ? 1.0 : 3.0;
        for (int a = 0; a < 8; ++a) {
            int64_t node = ((k + (a >> 2)) * N + (j + ((a >> 1) & 1))) * N + i + (a & 1);
            for (int d = 0; d < 3; ++d) dof[(3 * a + d) * ne + e] = 3 * node + d;
        }
    }
    for (int a = 0; a < 576; ++a) ke[a] = std::sin(0.37 * a) + (a % 25 == 0 ? 3.0 : 0.0);
    for (int64_t i = 0; i < nn; ++i) { xyz[3 * i] = i % N; xyz[3 * i + 1] = (i / N) % N; xyz[3 * i + 2] = i / ((int64_t)N * N); }
    pcg_elem_group g{24, ne, dof.data(), sg.data(), ck.data(), ke.data()};
    EbeHost m1, m2;
    build_ebe(nn, 1, &g, nullptr, 0, xyz.data(), true, 1, m1);
    build_ebe(nn, 1, &g, nullptr, 0, xyz.data(), true, 2, m2);
    Dev d1 = upload(m1), d2 = upload(m2);
    printf("N=%d elems=%ld chunks(ept1)=%ld colors=%d  chunks(ept2)=%ld colors=%d\n", N, (long)ne, d1.n_chunks, d1.n_colors, d2.n_chunks, d2.n_colors);
    const size_t n = 3 * nn;
    std::vector<double> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = std::cos(0.001 * i) + 0.3 * std::sin(1.7 * i);
    double *x, *y, *yref;
    CK(hipMalloc(&x, 8 * n)); CK(hipMalloc(&y, 8 * n)); CK(hipMalloc(&yref, 8 * n));
    CK(hipMemcpy(x, hx.data(), 8 * n, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<double> ref(n), got(n);
    auto check = [&](const char *name, float ms, double *yy) {
        CK(hipMemcpy(got.data(), yy, 8 * n, hipMemcpyDeviceToHost));
        double err = 0, nrm = 0;
        for (size_t i = 0; i < n; ++i) { err = std::max(err, std::fabs(got[i] - ref[i])); nrm = std::max(nrm, std::fabs(ref[i])); }
        printf("%-44s %8.3f ms   max|d|/max|y| = %.2e\n", name, ms, err / nrm);
        fflush(stdout);
    };
    float t0 = run<1, 0, 768, 1, 4>(d1, x, yref, n, reps, st);
    CK(hipMemcpy(ref.data(), yref, 8 * n, hipMemcpyDeviceToHost));
    check("ept1 scalarK maxn768 ypre lb4 (product)", t0, yref);
    check("ABL noFMA", run<1, 0, 768, 0, 4, 1, 1>(d1, x, y, n, reps, st), y);
    check("ABL noAccum", run<1, 0, 768, 0, 4, 1, 2>(d1, x, y, n, reps, st), y);
    check("ABL noFMA noAccum", run<1, 0, 768, 0, 4, 1, 3>(d1, x, y, n, reps, st), y);
    check("ABL coalesced x/y (node 0 / block-linear)", run<1, 0, 768, 0, 4, 1, 4>(d1, x, y, n, reps, st), y);
    check("ABL noFMA noAccum coalesced", run<1, 0, 768, 0, 4, 1, 7>(d1, x, y, n, reps, st), y);
    check("full batchedRMW", run<1, 0, 768, 0, 4, 1, 0>(d1, x, y, n, reps, st), y);
    return 0;
}
