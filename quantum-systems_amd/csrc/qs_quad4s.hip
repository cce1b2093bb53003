// Host side of the streamed fp64 kernel (qs_quad4s.h): eligibility, dispatch, and the instantiations up to 64 orbitals.
#include "qs_quad4s.h"

namespace qs {

// Out_t = Lm . In_t . R for t < nitems, fp64 (element strides); QS_OK / error after launching, 1 = not eligible.
int quad4s_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, hipStream_t stream) {
    if (dtype != QS_F64) return 1;
    if (L < 5 || M < 5 || L > 96 || M > 96) return 1;
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;
    if (nitems < 1 || nitems >= (int64_t(1) << 31)) return 1;
    if (in_col != 1 && in_item != 1) return 1;
    // the fetch takes 16 bytes: aligned pieces need an even row stride (slabs) / an even item count per row step (columns);
    // the 8-bytes-early form of an odd last element makes the pieces of odd extents misaligned by 8 -- buffer loads of
    // 16 bytes need 4-byte alignment only, so only the base has to be 8-byte aligned (it is: doubles)
    Quad4Args g;
    g.in = (const double*)in - 1;             // (8 bytes early: every lane offset carries + 8, see voffs_of)
    g.out = (double*)out;
    g.R = (const double*)R; g.Lm = (const double*)Lm;
    g.r_sk = r_sk; g.r_sj = r_sj; g.l_sp = l_sp; g.l_sa = l_sa;
    g.in_item = in_item; g.in_row = in_row; g.in_col = in_col;
    g.out_item = out_item; g.out_row = out_row; g.out_col = out_col;
    g.L = (int)L; g.M = (int)M;
    g.nitems = (unsigned)nitems;
    g.nquads = (unsigned)cdiv(nitems, 4);
    switch (n4) {
#define QS_QUAD4S_CASE(N) case N: return launch_quad4s<N>(g, stream);
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: one instantiation
        QS_QUAD4S_CASE(5)
#else
        QS_QUAD4S_CASE(2) QS_QUAD4S_CASE(3) QS_QUAD4S_CASE(4) QS_QUAD4S_CASE(5) QS_QUAD4S_CASE(6) QS_QUAD4S_CASE(7) QS_QUAD4S_CASE(8) QS_QUAD4S_CASE(9) QS_QUAD4S_CASE(10)
        QS_QUAD4S_CASE(11) QS_QUAD4S_CASE(12) QS_QUAD4S_CASE(13) QS_QUAD4S_CASE(14) QS_QUAD4S_CASE(15) QS_QUAD4S_CASE(16)
#endif
#undef QS_QUAD4S_CASE
        default: break;
    }
    int rc = launch_quad4s_w1(n4, g, stream);
    if (rc == 1) rc = launch_quad4s_w2(n4, g, stream);
    return rc;
}

}  // namespace qs
