// Host side of the streamed pair kernel (qs_pair4s.h): eligibility and dispatch of the complex128 fused passes (pair4c_try)
// and of the first pass of a real tensor against complex coefficients (pair4m_try).
#include "qs_pair4s.h"

namespace qs {

namespace {

int launch_any(int n4, bool real_in, const Pair4Args& g, hipStream_t stream) {
    int rc = 1;
    if (real_in) {
        rc = launch_pair4m_a(n4, g, stream);
        if (rc == 1) rc = launch_pair4m_b(n4, g, stream);
        return rc;
    }
    rc = launch_pair4s_a(n4, g, stream);
    if (rc == 1) rc = launch_pair4s_b(n4, g, stream);
    if (rc == 1) rc = launch_pair4s_c(n4, g, stream);
    if (rc == 1) rc = launch_pair4s_d(n4, g, stream);
    return rc;
}

bool fill(Pair4Args& g, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm, int64_t l_sp,
          int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row, int64_t in_col, int64_t out_item,
          int64_t out_row, int64_t out_col, int tensor_is_b) {
    if (nitems < 1 || nitems >= (int64_t(1) << 31)) return false;
    g.in = (const double*)in; g.out = (double*)out;
    g.R = (const double*)R; g.Lm = (const double*)Lm;
    g.r_sk = r_sk; g.r_sj = r_sj; g.l_sp = l_sp; g.l_sa = l_sa;
    g.in_item = in_item; g.in_row = in_row; g.in_col = in_col;
    g.out_item = out_item; g.out_row = out_row; g.out_col = out_col;
    g.L = (int)L; g.M = (int)M;
    g.nitems = (unsigned)nitems;
    g.npairs = (unsigned)cdiv(nitems, 2);
    g.tensor_is_b = tensor_is_b;
    return true;
}

}  // namespace

// Out_t = Lm . In_t . R for t < nitems, complex128 (element strides); QS_OK / error after launching, 1 = not eligible.
int pair4c_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, int tensor_is_b, hipStream_t stream) {
    if (dtype != QS_C128) return 1;
    if (L < 5 || M < 5 || L > 64 || M > 64) return 1;       // (up to 4 orbitals: qs_small4.hip)
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;
    if (in_col != 1 && in_item != 1) return 1;
    if (!aligned(in, 16) || !aligned(out, 16) || !aligned(R, 16) || !aligned(Lm, 16)) return 1;
    Pair4Args g;
    if (!fill(g, in, out, R, r_sk, r_sj, Lm, l_sp, l_sa, nitems, L, M, in_item, in_row, in_col, out_item, out_row, out_col, tensor_is_b))
        return 1;
    return launch_any(n4, false, g, stream);
}

// The same for REAL items against complex R and Lm (element strides of `in` count doubles, those of `out` complex
// elements): the first pass of a real tensor against complex coefficients.  Slabs only (in_col == 1).
int pair4m_try(const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm, int64_t l_sp, int64_t l_sa,
               int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row, int64_t in_col, int64_t out_item,
               int64_t out_row, int64_t out_col, hipStream_t stream) {
    if (L < 5 || M < 5 || L > 56 || M > 56) return 1;
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;
    if (in_col != 1) return 1;
    if (!aligned(in, 8) || !aligned(out, 16) || !aligned(R, 16) || !aligned(Lm, 16)) return 1;
    Pair4Args g;
    if (!fill(g, in, out, R, r_sk, r_sj, Lm, l_sp, l_sa, nitems, L, M, in_item, in_row, in_col, out_item, out_row, out_col, 0))
        return 1;
    return launch_any(n4, true, g, stream);
}

}  // namespace qs
