// Two-dimensional harmonic-oscillator Coulomb elements on the GPU: the input
// generator of BASELINE.json configs[1] (quantum dot, 10 shells, l = 55).
//
// Replaces quantum_systems/quantum_dots/two_dim/two_dim_helper.py:250-268
// (_get_coulomb_elements, numba prange over p) and coulomb_elements.py:6-92
// (coulomb_ho, the closed form of Anisimovas & Matulis, J. Phys.: Condens.
// Matter 10, 601 (1998)).  Embarrassingly parallel: one thread per (p,q,r,s),
// angular-momentum conservation zeroes ~90 % of the elements up front; the
// rest is an alternating double sum of exponentials of log-factorial /
// log-gamma table entries (tables built once per workgroup in LDS, all
// arguments are integers or half-integers).  Compute-bound on fp64 VALU and
// `exp`; persistent workgroups with a grid-stride walk so the tables are built
// once per workgroup.  Same summation order per element as the reference.

#include "qs_common.h"

namespace qs {

static constexpr int TD_MAXF = 256;    // log-factorial table
static constexpr int TD_MAXG = 512;    // lgamma(k/2) table
static constexpr int TD_MAXL = 2048;   // orbitals

// orbital index -> (n, m): shells of 1, 2, 3, ... states (two_dim_helper.py:132-166)
__host__ __device__ inline void tdho_indices_nm(int p, int& n, int& m) {
    int previous = 0, current = 1, shell = 1;
    while (current <= p) {
        shell += 1;
        previous = current;
        current = previous + shell;
    }
    const int width = current - previous;
    // middle of an odd shell is the m = 0 state
    if ((width & 1) && p == previous + width / 2) { n = shell / 2; m = 0; return; }
    if (2 * p < 2 * previous + width) { n = p - previous; m = -((shell - 1) - 2 * n); }
    else { n = (current - 1) - p; m = (shell - 1) - 2 * n; }
}

__device__ double tdho_element(const double* __restrict__ lf, const double* __restrict__ lgh,
                               int n_i, int m_i, int n_j, int m_j, int n_l, int m_l, int n_k, int m_k) {
    if (m_i + m_j != m_k + m_l) return 0.0;
    const int am_i = abs(m_i), am_j = abs(m_j), am_k = abs(m_k), am_l = abs(m_l);
    const int M_i = (am_i + m_i) >> 1, dm_i = (am_i - m_i) >> 1;
    const int M_j = (am_j + m_j) >> 1, dm_j = (am_j - m_j) >> 1;
    const int M_k = (am_k + m_k) >> 1, dm_k = (am_k - m_k) >> 1;
    const int M_l = (am_l + m_l) >> 1, dm_l = (am_l - m_l) >> 1;
    const double ln2 = 0.6931471805599453094;
    double element = 0.0;
    for (int j0 = 0; j0 <= n_i; ++j0)
        for (int j1 = 0; j1 <= n_j; ++j1)
            for (int j2 = 0; j2 <= n_k; ++j2)
                for (int j3 = 0; j3 <= n_l; ++j3) {
                    const int g0 = j0 + j3 + M_i + dm_l;
                    const int g1 = j1 + j2 + M_j + dm_k;
                    const int g2 = j2 + j1 + M_k + dm_j;
                    const int g3 = j3 + j0 + M_l + dm_i;
                    const int G = g0 + g1 + g2 + g3;
                    const double ratio_1 = -(lf[j0] + lf[j1] + lf[j2] + lf[j3]);
                    const double prod_2 = (lf[n_i + am_i] - lf[n_i - j0] - lf[j0 + am_i]) +
                                          (lf[n_j + am_j] - lf[n_j - j1] - lf[j1 + am_j]) +
                                          (lf[n_k + am_k] - lf[n_k - j2] - lf[j2 + am_k]) +
                                          (lf[n_l + am_l] - lf[n_l - j3] - lf[j3 + am_l]);
                    const double ratio_2 = -0.5 * (G + 1) * ln2;
                    const double lfg = lf[g0] + lf[g1] + lf[g2] + lf[g3];
                    double temp = 0.0;
                    for (int l0 = 0; l0 <= g0; ++l0)
                        for (int l1 = 0; l1 <= g1; ++l1) {
                            // l0 + l1 == l2 + l3 fixes l3 once l2 is chosen
                            const int s01 = l0 + l1;
                            const int l2_lo = s01 > g3 ? s01 - g3 : 0;
                            const int l2_hi = s01 < g2 ? s01 : g2;
                            const double c01 = lfg - lf[l0] - lf[g0 - l0] - lf[l1] - lf[g1 - l1];
                            const int L = 2 * s01;
                            const double gam = lgh[2 + L] + lgh[G - L + 1];   // lgamma(1 + L/2) + lgamma((G-L+1)/2)
                            for (int l2 = l2_lo; l2 <= l2_hi; ++l2) {
                                const int l3 = s01 - l2;
                                const double prod_3 = c01 - lf[l2] - lf[g2 - l2] - lf[l3] - lf[g3 - l3];
                                const double term = exp(prod_3 + gam);
                                temp += ((g1 + g2 - l1 - l2) & 1) ? -term : term;
                            }
                        }
                    const double w = exp(ratio_1 + prod_2 + ratio_2) * temp;
                    element += ((j0 + j1 + j2 + j3) & 1) ? -w : w;
                }
    const double prod_1 = (lf[n_i] - lf[n_i + am_i]) + (lf[n_j] - lf[n_j + am_j]) +
                          (lf[n_k] - lf[n_k + am_k]) + (lf[n_l] - lf[n_l + am_l]);
    return element * exp(0.5 * prod_1);
}

// nm_table: optional device table [2][l] of (n, m) per orbital (magnetic-field ordering,
// two_dim_helper.py:284-301); nullptr = the shell order of two_dim_helper.py:132-166
__global__ __launch_bounds__(256) void tdho_coulomb_kernel(double* __restrict__ out, const int* __restrict__ nm_table,
                                                           int l, int p_lo, int64_t total) {
    __shared__ double s_lf[TD_MAXF];
    __shared__ double s_lgh[TD_MAXG];
    extern __shared__ int s_nm[];   // [2][l]
    for (int k = threadIdx.x; k < TD_MAXG; k += blockDim.x) s_lgh[k] = k ? lgamma(0.5 * k) : 0.0;
    if (threadIdx.x == 0) {
        s_lf[0] = 0.0; s_lf[1] = 0.0;
        for (int n = 2; n < TD_MAXF; ++n) s_lf[n] = s_lf[n - 1] + log((double)n);   // coulomb_elements.py:95-102
    }
    for (int p = threadIdx.x; p < l; p += blockDim.x) {
        if (nm_table) { s_nm[p] = nm_table[p]; s_nm[l + p] = nm_table[l + p]; }
        else tdho_indices_nm(p, s_nm[p], s_nm[l + p]);
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t l3 = (int64_t)l * l * l, l2 = (int64_t)l * l;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int p = (int)(idx / l3) + p_lo;
        int64_t rem = idx % l3;
        const int q = (int)(rem / l2); rem %= l2;
        const int r = (int)(rem / l), s = (int)(rem % l);
        // the reference calls coulomb_ho(nm(p), nm(q), nm(r), nm(s)) whose parameter order is (i, j, l, k)
        out[idx] = tdho_element(s_lf, s_lgh, s_nm[p], s_nm[l + p], s_nm[q], s_nm[l + q],
                                s_nm[r], s_nm[l + r], s_nm[s], s_nm[l + s]);
    }
}

}  // namespace qs

using namespace qs;

static int launch_tdho(void* out, const int* nm_table, int64_t l, int shell, int64_t p_lo, int64_t p_hi,
                       void* stream) {
    if (!out) return QS_ERR_NULL_POINTER;
    if (l <= 0 || l > TD_MAXL || p_lo < 0 || p_hi > l || p_lo >= p_hi) return QS_ERR_BAD_EXTENT;
    if (!aligned(out, 8)) return QS_ERR_MISALIGNED;
    // the largest table index is G + 1 <= 4 * (2 n_max + |m|_max) + 1
    if (shell < 1 || 8 * shell + 8 >= TD_MAXF) return QS_ERR_BAD_EXTENT;
    const int64_t total = (p_hi - p_lo) * l * l * l;
    const int64_t want = cdiv(total, 256);
    const unsigned grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);
    hipLaunchKernelGGL(tdho_coulomb_kernel, dim3(grid), dim3(256), sizeof(int) * 2 * l, (hipStream_t)stream,
                       (double*)out, nm_table, (int)l, (int)p_lo, total);
    return launch_status("tdho_coulomb launch");
}

extern "C" {

int qs_tdho_coulomb_elements(void* out, int64_t l, int64_t p_lo, int64_t p_hi, void* stream) {
    if (l <= 0 || l > TD_MAXL) return QS_ERR_BAD_EXTENT;
    // shells grow like sqrt(2 l): the last orbital sits in the highest one
    int n_top, m_top;
    tdho_indices_nm((int)l - 1, n_top, m_top);
    return launch_tdho(out, nullptr, l, 2 * n_top + abs(m_top) + 1, p_lo, p_hi, stream);
}

int qs_tdho_coulomb_elements_nm(void* out, const void* nm_table, int64_t l, int64_t max_shell, int64_t p_lo,
                                int64_t p_hi, void* stream) {
    if (!nm_table) return QS_ERR_NULL_POINTER;
    if (!aligned(nm_table, 4)) return QS_ERR_MISALIGNED;
    if (max_shell < 1 || max_shell > 64) return QS_ERR_BAD_EXTENT;
    return launch_tdho(out, (const int*)nm_table, l, (int)max_shell, p_lo, p_hi, stream);
}

}  // extern "C"
