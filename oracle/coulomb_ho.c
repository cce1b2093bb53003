/*
 * CPU oracle for the two-dimensional harmonic-oscillator Coulomb elements.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/qs_oracle.py): built by
 * __graft_entry__.build() into oracle/_build/, loaded by oracle/coulomb_oracle.py,
 * used by tests/ as the checker of the HIP generator.  Never shipped, never timed
 * as the product.
 *
 * Plain-C restatement of the closed form of Anisimovas & Matulis,
 * J. Phys.: Condens. Matter 10, 601 (1998), as the reference evaluates it:
 *   quantum_systems/quantum_dots/two_dim/coulomb_elements.py:6-92   coulomb_ho
 *   ...:95-152                                                       log-factorial helpers
 *   quantum_systems/quantum_dots/two_dim/two_dim_helper.py:132-166   get_indices_nm
 *   ...:250-268                                                      _get_coulomb_elements
 * Same loop nest and summation order (j1..j4 outer, l1..l4 inner, running sum
 * in double); the reference compiles with numba fastmath, so agreement is to
 * rounding, not bitwise.  Pinned against the reference's own table
 * tests/dat/two_dim_quantum_dots_coulomb_elements.dat (committed as
 * tests/golden/tdho_coulomb_table.npz) and against elements computed by the
 * reference code itself (tests/golden/tdho_coulomb_spot.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define MAXF 256
static double g_logfac[MAXF];
static int g_ready = 0;

static void init_tables(void) {
    if (g_ready) return;
    g_logfac[0] = 0.0;
    g_logfac[1] = 0.0;
    /* log_factorial(n) = sum_{a=2..n} log(a)   (coulomb_elements.py:95-102) */
    for (int n = 2; n < MAXF; ++n) g_logfac[n] = g_logfac[n - 1] + log((double)n);
    g_ready = 1;
}

/* orbital index -> (n, m): shells of 1, 2, 3, ... states, m ascending inside a
 * shell (two_dim_helper.py:132-166) */
void tdho_indices_nm(int p, int* n_out, int* m_out) {
    int previous = 0, current = 1, shell = 1;
    while (current <= p) {
        shell += 1;
        previous = current;
        current = previous + shell;
    }
    const int width = current - previous;
    const double middle = width / 2.0 + previous;
    if ((width & 1) && fabs(p - floor(middle)) < 1e-8) {
        *n_out = shell / 2;
        *m_out = 0;
        return;
    }
    if (p < middle) {
        const int n = p - previous;
        *n_out = n;
        *m_out = -((shell - 1) - 2 * n);
    } else {
        const int n = (current - 1) - p;
        *n_out = n;
        *m_out = (shell - 1) - 2 * n;
    }
}

/* <ij|u|lk> in the paper's index order (last two swapped w.r.t. <ij|u|kl>),
 * coulomb_elements.py:6-92 */
double tdho_coulomb_ho(int n_i, int m_i, int n_j, int m_j, int n_l, int m_l, int n_k, int m_k) {
    init_tables();
    if (m_i + m_j != m_k + m_l) return 0.0;
    const int am_i = abs(m_i), am_j = abs(m_j), am_k = abs(m_k), am_l = abs(m_l);
    const int M_i = (am_i + m_i) / 2, dm_i = (am_i - m_i) / 2;
    const int M_j = (am_j + m_j) / 2, dm_j = (am_j - m_j) / 2;
    const int M_k = (am_k + m_k) / 2, dm_k = (am_k - m_k) / 2;
    const int M_l = (am_l + m_l) / 2, dm_l = (am_l - m_l) / 2;
    const int n[4] = {n_i, n_j, n_k, n_l};
    const int am[4] = {am_i, am_j, am_k, am_l};
    double element = 0.0;
    int j[4], g[4], l[4];
    for (j[0] = 0; j[0] <= n_i; ++j[0])
        for (j[1] = 0; j[1] <= n_j; ++j[1])
            for (j[2] = 0; j[2] <= n_k; ++j[2])
                for (j[3] = 0; j[3] <= n_l; ++j[3]) {
                    g[0] = j[0] + j[3] + M_i + dm_l;
                    g[1] = j[1] + j[2] + M_j + dm_k;
                    g[2] = j[2] + j[1] + M_k + dm_j;
                    g[3] = j[3] + j[0] + M_l + dm_i;
                    const int G = g[0] + g[1] + g[2] + g[3];
                    double ratio_1 = 0.0, prod_2 = 0.0;
                    for (int t = 0; t < 4; ++t) {
                        ratio_1 -= g_logfac[j[t]];                                    /* log_ratio_1 */
                        prod_2 += g_logfac[n[t] + am[t]] - g_logfac[n[t] - j[t]] -    /* log_product_2 */
                                  g_logfac[j[t] + am[t]];
                    }
                    const double ratio_2 = -0.5 * (G + 1) * log(2.0);                 /* log_ratio_2 */
                    double temp = 0.0;
                    for (l[0] = 0; l[0] <= g[0]; ++l[0])
                        for (l[1] = 0; l[1] <= g[1]; ++l[1])
                            for (l[2] = 0; l[2] <= g[2]; ++l[2])
                                for (l[3] = 0; l[3] <= g[3]; ++l[3]) {
                                    if (l[0] + l[1] != l[2] + l[3]) continue;
                                    const int L = l[0] + l[1] + l[2] + l[3];
                                    double prod_3 = 0.0;                              /* log_product_3 */
                                    for (int t = 0; t < 4; ++t)
                                        prod_3 += g_logfac[g[t]] - g_logfac[l[t]] - g_logfac[g[t] - l[t]];
                                    const int sign = -2 * ((g[1] + g[2] - l[1] - l[2]) & 1) + 1;
                                    temp += sign * exp(prod_3 + lgamma(1.0 + 0.5 * L) +
                                                       lgamma(0.5 * (G - L + 1.0)));
                                }
                    const int jsum = j[0] + j[1] + j[2] + j[3];
                    element += (-2 * (jsum & 1) + 1) * exp(ratio_1 + prod_2 + ratio_2) * temp;
                }
    double prod_1 = 0.0;                                                              /* log_product_1 */
    for (int t = 0; t < 4; ++t) prod_1 += g_logfac[n[t]] - g_logfac[n[t] + am[t]];
    return element * exp(0.5 * prod_1);
}

/* u[p,q,r,s] = coulomb_ho(nm(p), nm(q), nm(r), nm(s)) for p in [p_lo, p_hi)
 * (two_dim_helper.py:250-268); out holds (p_hi - p_lo) * l^3 doubles. */
void tdho_coulomb_elements(int l, int p_lo, int p_hi, double* out) {
    int* nn = (int*)malloc(sizeof(int) * l);
    int* mm = (int*)malloc(sizeof(int) * l);
    for (int p = 0; p < l; ++p) tdho_indices_nm(p, &nn[p], &mm[p]);
    for (int p = p_lo; p < p_hi; ++p)
        for (int q = 0; q < l; ++q)
            for (int r = 0; r < l; ++r)
                for (int s = 0; s < l; ++s)
                    out[(((int64_t)(p - p_lo) * l + q) * l + r) * l + s] =
                        tdho_coulomb_ho(nn[p], mm[p], nn[q], mm[q], nn[r], mm[r], nn[s], mm[s]);
    free(nn);
    free(mm);
}
