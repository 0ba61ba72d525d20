/* TEST INFRASTRUCTURE - C restatement of the reference's element-by-element mat-vec.
 *
 * Follows /root/reference/src/solver/pcg_solver.py:277-280 (gather, sign flip, Ke @ (Ck*U), sign
 * flip) for one pattern-type group and :300 (np.bincount == sequential scatter-add in index
 * order).  Used by oracle/pcg_oracle.py (use_c=True) for sizes where the NumPy temporaries are
 * too slow, and as bench.py's `cpu_baseline` (kind "port", 1 thread - the reference pins BLAS to
 * one thread, pcg_solver.py:10-15).  Never linked into the product library.
 *
 * Differences from the NumPy original, by construction: the (nd x nd)@(nd x Ne) product is
 * accumulated over b in ascending order here, whereas NumPy hands it to BLAS dgemm whose
 * summation order is unspecified; the two agree to rounding (checked to <=1e-14 relative in
 * tests/test_oracle_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>

#define EB 64   /* elements per register block */

void ebe_matvec_group(int nd, int64_t ne, const int64_t *tbl, const uint8_t *sign, const double *Ck,
                      const double *Ke, const double *x, double *out)
{
    double *u = (double *)malloc(sizeof(double) * (size_t)nd * EB);
    for (int64_t e0 = 0; e0 < ne; e0 += EB) {
        int m = (int)((ne - e0) < EB ? (ne - e0) : EB);
        for (int b = 0; b < nd; ++b) {
            const int64_t *t = tbl + (int64_t)b * ne + e0;
            const uint8_t *s = sign + (int64_t)b * ne + e0;
            for (int k = 0; k < m; ++k) {
                double v = x[t[k]];                 /* :277 gather          */
                if (s[k]) v *= -1.0;                /* :278 sign flip in    */
                u[b * EB + k] = Ck[e0 + k] * v;     /* :279 Ck * U          */
            }
        }
        for (int a = 0; a < nd; ++a) {
            double acc[EB];
            for (int k = 0; k < m; ++k) acc[k] = 0.0;
            for (int b = 0; b < nd; ++b) {          /* :279 Ke @ (.)        */
                double kab = Ke[a * nd + b];
                const double *ub = u + b * EB;
                for (int k = 0; k < m; ++k) acc[k] += kab * ub[k];
            }
            const uint8_t *s = sign + (int64_t)a * ne + e0;
            double *o = out + (int64_t)a * ne + e0;
            for (int k = 0; k < m; ++k) o[k] = s[k] ? acc[k] * -1.0 : acc[k];   /* :280 */
        }
    }
    free(u);
}

/* np.bincount(idx, weights=val, minlength=n): y must be zeroed by the caller. (:300) */
void scatter_add_seq(int64_t n, const int64_t *idx, const double *val, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[idx[i]] += val[i];
}
