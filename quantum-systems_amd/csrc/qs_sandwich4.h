// Arguments of the small-basis kernels (qs_sandwich4.hip, qs_sandwich4b.hip): Out_t = Lm . In_t . R for item quads.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace qs {

struct S4Args {
    const double* in;
    double* out;
    const double* R;      // R[k][j]  = R[k * r_sk + j * r_sj],   L x M
    const double* Lm;     // Lm[p][a] = Lm[p * l_sp + a * l_sa],  M x L
    int64_t r_sk, r_sj, l_sp, l_sa;
    int64_t in_item, in_row, in_col;       // element strides of In_t[i][k]: in_col == 1 (a slab) or in_item == 1 (a column)
    int64_t out_item, out_row, out_col;    // element strides of Out_t[p][j]
    int L, M;
    unsigned nitems, nquads;
    unsigned tail_first, tail_parts;   // balanced form only (else 0, 0): item quads from tail_first on -- the partly filled last
                                       // round -- are split into tail_parts column-group parts, one workgroup each
    int mode;      // bit 0: the four waves take four ADJACENT item quads and the same chunk (else: one quad, four chunks);
                   // bit 1: with bit 0, a workgroup barrier per step keeps the four waves' fetches together in L1
};

// The balanced form with a cooperative fetch (qs_sandwich4b.hip): slabs in (in_col == 1), instantiations for n4 in
// {10, 12, 14, 16}, each for 4 (n4 - 2) < L, M <= 4 n4.
// QS_OK / error after launching, 1 = no such instantiation; with `dry_run` nothing is launched (QS_OK = would launch).
int sandwich4b_launch(const S4Args& g, int n4, hipStream_t stream, int dry_run);
// The item quads of the last round of `nquads` (M columns) that the balanced form splits over all workgroups; 0 = it does
// not (no partly filled last round, or more tasks than workgroups; g_tune.sandwich_tail).
unsigned sandwich4b_tail_quads(unsigned nquads, int M);

}  // namespace qs
