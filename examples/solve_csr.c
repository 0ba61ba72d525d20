/* The C ABI without Python or torch: assemble a scalar CSR system on the host, solve it with Jacobi-PCG on the GPU.
 *
 *   gcc -O2 -Iinclude examples/solve_csr.c -Lpcg-mpi-solver_amd/lib -lpcg_mi355x -Wl,-rpath,$PWD/pcg-mpi-solver_amd/lib -lm -o solve_csr
 *   ./solve_csr [m]          3-D 7-point Laplacian on an m x m x m grid (n = m^3 rows, any n: block = 1 keeps the scalar format)
 *
 * Every call returns 0 or a negative code with text in pcg_last_error(); the solver outcome (flag 0..4 with the
 * reference's meanings, pcg_solver.py:356-598) is data in pcg_result. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pcg_mi355x.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pcg_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 48;
    const int64_t n = (int64_t)m * m * m;
    int64_t *rowptr = malloc(sizeof(int64_t) * (n + 1));
    int32_t *col = malloc(sizeof(int32_t) * 7 * n);
    double *val = malloc(sizeof(double) * 7 * n), *b = malloc(sizeof(double) * n), *x = malloc(sizeof(double) * n);
    int64_t nnz = 0;
    for (int k = 0; k < m; ++k)
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) {
                const int64_t r = ((int64_t)k * m + j) * m + i;
                rowptr[r] = nnz;
                const int di[7] = {0, 0, -1, 0, 1, 0, 0}, dj[7] = {0, -1, 0, 0, 0, 1, 0}, dk[7] = {-1, 0, 0, 0, 0, 0, 1};
                for (int s = 0; s < 7; ++s) {
                    const int ii = i + di[s], jj = j + dj[s], kk = k + dk[s];
                    if (ii < 0 || jj < 0 || kk < 0 || ii >= m || jj >= m || kk >= m) continue;
                    col[nnz] = (int32_t)(((int64_t)kk * m + jj) * m + ii);
                    val[nnz++] = s == 3 ? 6.0 : -1.0;
                }
                b[r] = sin(0.1 * (double)r) + 1.0;
            }
    rowptr[n] = nnz;
    if (pcg_abi_version() != PCG_ABI_VERSION) { fprintf(stderr, "libpcg_mi355x ABI version %d, this program was compiled for %d\n", pcg_abi_version(), PCG_ABI_VERSION); return 2; }
    if (pcg_device_count() < 1) { fprintf(stderr, "no HIP device visible (the engine has no CPU fallback)\n"); return 2; }
    pcg_engine *e = NULL;
    CHECK(pcg_create_csr(0, n, rowptr, col, val, 0, /*block=*/1, &e));
    CHECK(pcg_build_jacobi(e, NULL));
    pcg_result res;
    CHECK(pcg_solve(e, b, NULL, NULL, 1e-9, 5000, n, x, NULL, 0, &res));
    /* true residual on the host */
    double rr = 0, bb = 0;
    for (int64_t r = 0; r < n; ++r) {
        double ax = 0;
        for (int64_t q = rowptr[r]; q < rowptr[r + 1]; ++q) ax += val[q] * x[col[q]];
        rr += (b[r] - ax) * (b[r] - ax);
        bb += b[r] * b[r];
    }
    printf("%s: n = %lld, nnz = %lld, flag %d, %lld iterations, relres %.3e (host check %.3e), %.3f s\n", pcg_backend_name(),
           (long long)n, (long long)nnz, res.flag, (long long)res.iter, res.relres, sqrt(rr / bb), res.t_total_s);
    pcg_destroy(e);
    free(rowptr); free(col); free(val); free(b); free(x);
    return res.flag == 0 && sqrt(rr / bb) < 1.01e-9 ? 0 : 3;
}
