/* The device-group C ABI (pcg_group_*) without Python or torch: ONE process, G mesh parts on G GPUs.
 *
 *   gcc -O2 -Iinclude examples/solve_group.c -Lpcg-mpi-solver_amd/lib -lpcg_mi355x -Wl,-rpath,$PWD/pcg-mpi-solver_amd/lib -lm -o solve_group
 *   ./solve_group [G] [nodes]        G parts (default: one per visible GPU, at most 8), part g on device g % (visible GPUs);
 *                                    nodes - 1 must be divisible by G (default: about 193)
 *
 * Model: a chain of `nodes` 3-dof nodes joined by springs with the 3x3 stiffness K_e = k_e (I + 0.2 J) (J = all ones,
 * k_e = 1 + e % 3), node 0 clamped, the load F = (1, 2, -1) on the last node.  The chain is cut into G parts the way the
 * reference's partitioner cuts a mesh (partition_mesh.py:745-887): parts own ELEMENTS (springs), the node between two parts
 * exists on both (interface node, owned by the part with the smaller id), and every part holds the un-exchanged sub-domain
 * matrix of its own springs with its interface nodes numbered first.  What mpi4py does between the ranks of the reference
 * (interface sums :318-334, MPI_SUM :622-628) is done here by the engine's native communicator between the group's members.
 * Exact solution: u_i = (sum_{e<i} 1/k_e) (I - 0.125 J) F.
 *
 * Every call returns 0 or a negative code with text in pcg_last_error(); pcg_result.flag 0..4 has the reference's meanings. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcg_mi355x.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pcg_last_error()); return 1; } \
    } while (0)

static double spring(int e) { return 1.0 + e % 3; }

int main(int argc, char **argv)
{
    if (pcg_abi_version() != PCG_ABI_VERSION) { fprintf(stderr, "libpcg_mi355x ABI version %d, this program was compiled for %d\n", pcg_abi_version(), PCG_ABI_VERSION); return 2; }
    const int have = pcg_device_count();
    const int is_double = strcmp(pcg_backend_name(), "hip-gfx950") != 0;   /* tests/hostops: the CPU test double also runs this program */
    if (have < 1 && !is_double) { fprintf(stderr, "no HIP device visible (the engine has no CPU fallback)\n"); return 2; }
    int G = argc > 1 ? atoi(argv[1]) : (have < 1 ? 1 : (have > 8 ? 8 : have));
    const int n_nodes = argc > 2 ? atoi(argv[2]) : 1 + G * (192 / (G > 0 ? G : 1));    /* ~193, springs divisible by G */
    const int n_el = n_nodes - 1;
    if (G < 1 || n_el < G || n_el % G) { fprintf(stderr, "need parts >= 1 and (nodes - 1) divisible by parts\n"); return 1; }
    const int S = n_el / G;                                     /* springs per part; part g: springs [gS, gS+S), nodes gS .. gS+S */
    const double F[3] = {1.0, 2.0, -1.0};

    int32_t dev[64];
    pcg_group *grp = NULL;
    pcg_engine *eng[64];
    double *b[64], *x[64];
    if (G > 64) return 1;
    for (int g = 0; g < G; ++g) dev[g] = have > 0 ? g % have : 0;
    CHECK(pcg_group_create(G, dev, &grp));                      /* + the members' communicators (RCCL, id generated in-process) */

    for (int g = 0; g < G; ++g) {
        /* local numbering: interface nodes first (left one, then right one), then the interior nodes left to right */
        const int nl = S + 1, has_l = g > 0, has_r = g < G - 1, n_bnd = has_l + has_r;
        int *loc = malloc(sizeof(int) * nl);                    /* chain position p (0..S) -> local node */
        int next = 0;
        if (has_l) loc[0] = next++;
        if (has_r) loc[S] = next++;
        for (int p = 0; p <= S; ++p)
            if (!((p == 0 && has_l) || (p == S && has_r))) loc[p] = next++;
        /* 3x3-block CSR of the part's own springs: row of local node loc[p] has the diagonal block and one block per spring */
        int64_t *rowptr = calloc(nl + 1, sizeof(int64_t));
        int32_t *cols = malloc(sizeof(int32_t) * 3 * nl);
        double *vals = calloc((size_t)9 * 3 * nl, sizeof(double));
        int *pos_of = malloc(sizeof(int) * nl);                 /* local node -> chain position */
        for (int p = 0; p <= S; ++p) pos_of[loc[p]] = p;
        int64_t nb = 0;
        for (int l = 0; l < nl; ++l) {
            const int p = pos_of[l];
            rowptr[l] = nb;
            const double kl = p > 0 ? spring(g * S + p - 1) : 0.0, kr = p < S ? spring(g * S + p) : 0.0;
            const int nbr[3] = {p > 0 ? loc[p - 1] : -1, l, p < S ? loc[p + 1] : -1};
            const double w[3] = {-kl, kl + kr, -kr};
            for (int q = 0; q < 3; ++q) {
                if (nbr[q] < 0) continue;
                cols[nb] = nbr[q];
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) vals[nb * 9 + a * 3 + c] = w[q] * ((a == c ? 1.0 : 0.0) + 0.2);
                ++nb;
            }
        }
        rowptr[nl] = nb;
        CHECK(pcg_create(dev[g], nl, rowptr, cols, vals, n_bnd, 0, &eng[g]));
        /* ownership / free-dof flags (partition_mesh.py:868-887, :350-351): the left interface node belongs to part g-1 */
        uint8_t *flags = malloc(3 * nl);
        for (int l = 0; l < nl; ++l) {
            const int p = pos_of[l], owned = !(p == 0 && has_l), fixed = g == 0 && p == 0;
            for (int a = 0; a < 3; ++a) flags[3 * l + a] = (uint8_t)((owned ? 1 : 0) | (fixed ? 0 : 2));
        }
        CHECK(pcg_set_masks(eng[g], flags));
        /* interface lists (OvrlpLocalDofVecList / NbrMPIdVector): neighbours in ascending part id, peer id == member index */
        int32_t peers[2], idx[6];
        int64_t ptr[3] = {0, 0, 0};
        int np = 0;
        if (has_l) { peers[np] = g - 1; for (int a = 0; a < 3; ++a) idx[3 * np + a] = 3 * loc[0] + a; ++np; ptr[np] = 3 * np; }
        if (has_r) { peers[np] = g + 1; for (int a = 0; a < 3; ++a) idx[3 * np + a] = 3 * loc[S] + a; ++np; ptr[np] = 3 * np; }
        CHECK(pcg_set_halo(eng[g], np, peers, ptr, idx));
        CHECK(pcg_group_attach(grp, g, eng[g]));
        b[g] = calloc(3 * nl, sizeof(double));
        x[g] = calloc(3 * nl, sizeof(double));
        if (g == G - 1)
            for (int a = 0; a < 3; ++a) b[g][3 * loc[S] + a] = F[a];
        free(loc); free(rowptr); free(cols); free(vals); free(pos_of); free(flags);
    }

    CHECK(pcg_group_build_jacobi(grp, NULL));                   /* updatePreconditioner on every part, interface-summed */
    pcg_result res[64];
    CHECK(pcg_group_solve(grp, (const double *const *)b, NULL, NULL, 1e-10, 20000, 3 * (int64_t)(n_nodes - 1), x, NULL, 0, res));

    /* every part against the closed form */
    double worst = 0;
    for (int g = 0; g < G; ++g) {
        const int has_l = g > 0, has_r = g < G - 1;
        int next = 0, lL = -1, lR = -1;
        if (has_l) lL = next++;
        if (has_r) lR = next++;
        for (int p = 0; p <= S; ++p) {
            const int l = (p == 0 && has_l) ? lL : (p == S && has_r) ? lR : next++;
            double c = 0;
            for (int e = 0; e < g * S + p; ++e) c += 1.0 / spring(e);
            const double sF = F[0] + F[1] + F[2];
            for (int a = 0; a < 3; ++a) {
                const double want = c * (F[a] - 0.125 * sF), got = x[g][3 * l + a];
                const double err = fabs(got - want) / (1.0 + fabs(want));
                if (err > worst) worst = err;
            }
        }
        if (res[g].flag != res[0].flag || res[g].iter != res[0].iter) { fprintf(stderr, "members disagree on the outcome\n"); return 4; }
    }
    printf("%s: %d part(s) on %d device(s), %d nodes, flag %d, %lld iterations, relres %.3e, worst error vs the closed form %.2e\n",
           pcg_backend_name(), G, have, n_nodes, res[0].flag, (long long)res[0].iter, res[0].relres, worst);
    for (int g = 0; g < G; ++g) { pcg_destroy(eng[g]); free(b[g]); free(x[g]); }      /* engines first, then the group */
    pcg_group_destroy(grp);
    return res[0].flag == 0 && worst < 1e-7 ? 0 : 3;
}
