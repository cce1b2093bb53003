// Both fused passes of the four-index transform for COMPLEX128 bases of up to 64 orbitals, and the first pass of a REAL tensor
// against complex coefficients: the streamed pair kernel (shared by qs_pair4c.hip and the translation units that hold its
// instantiations, qs_pair4s_*.hip / qs_pair4m_*.hip -- one file took a quarter of an hour to compile).
//
//     Out_t = Lm . In_t . R        for every item t of a batch of L x L matrices        (ceil(L/4) == ceil(M/4) = N4)
//
//   (d, c):  item t = slab (a, b),   In_t = u[a, b, :, :]        R = C,    Lm = C^T    -> T2[a, b, :, :]
//   (b, a):  item t = column (r, s), In_t = T2[:, :, r, s]       R = Ct^T, Lm = Ct     -> out[:, :, r, s]
//
// Two launches, two passes over the tensor instead of four (basis_set.py:341-348).  RandomBasisSet (random_basis.py:52-69) and
// every spin-doubled tensor are complex.
//
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 x 4 blocks per instruction (lane = x + 4 y + 16 z: A row x /
// block y / k z, B k z / block y / column x, D row z / block y / column x), and BOTH operands may differ from block to
// block.  The four blocks here are (item, part) for TWO items and part = re / im of the RESULT:
//
//     Y = In . R      MFMA 1:  A = In_re (both parts)            B = R_re (part re),  R_im (part im)
//                     MFMA 2:  A = In_im (both parts)            B = -R_im (part re), R_re (part im)
//     Out = Lm . Y    MFMA 1:  A = Lm_re                         B = Y        (blocks: Y_re, Y_im)
//                     MFMA 2:  A = Lm_im                         B = Y swapped between the parts, the re block negated
//                                                                    (blocks: -Y_im, Y_re)
//
// so a complex multiply-add costs four real MFMAs per four items, Y = In . R leaves the accumulators in the B-operand layout
// of Lm . Y -- one lane exchange (ds_swizzle, lane ^ 4) per row quad of Y makes the second operand, no LDS round trip -- and no
// negated copies of any table exist: the signs ride in the B operands, which are registers (R: loaded once per wave for the
// whole launch) or the exchanged Y.  Every element is the same chain of fused multiply-adds as on the 16-wide kernels
// (qs_gemm.hip mfma_step: per k-quad re += ar.br, im += ar.bi, re += (-ai).bi, im += ai.br with A / B the operands of THAT
// kernel's call; in the b contraction the tensor is its B operand, so there MFMA 1 takes A = In_re / In_im by part against
// R_re and MFMA 2 A = In_im / In_re against -R_im / R_im): results are bit-identical to the 16-wide path
// (tests/test_gpu_kernels.py).  Algorithmic bytes per launch: 16 (L^2 + M^2) per item.
//
// History (profiles/r03_pair4c.txt): the first form of round 3 staged an item pair WHOLE in LDS (108 KB at 56 orbitals) with
// eight waves of two column groups each; one wave per column group balanced the SIMDs (+9-18 %); its loads were in the open
// above 40 orbitals (12-26 % of a launch).  The streamed form below replaced it at every size and the whole-pair kernel is gone.
#pragma once

#include <type_traits>

#include "qs_common.h"

namespace qs {

template <int I, int N, class F>
__device__ __forceinline__ void unroll(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll<I + 1, N>(f);
    }
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct Pair4Args {
    const double* in;
    double* out;
    const double* R;      // R[k][j]  = R[k * r_sk + j * r_sj],   L x M   (complex elements)
    const double* Lm;     // Lm[p][a] = Lm[p * l_sp + a * l_sa],  M x L
    int64_t r_sk, r_sj, l_sp, l_sa;
    int64_t in_item, in_row, in_col;       // element strides of In_t[i][k]: in_col == 1 (a slab) or in_item == 1 (a column)
    int64_t out_item, out_row, out_col;    // element strides of Out_t[p][j]
    int L, M;
    unsigned nitems, npairs;
    int tensor_is_b;      // the 16-wide kernels' call for the FIRST product has the tensor as its B operand (the b contraction)
};

// development builds only: bit mask of parts to leave out (1 the loads of the item pairs, 2 the stores, 4 the MFMAs)
#ifndef QS_PAIR4C_ABLATE
#define QS_PAIR4C_ABLATE 0
#endif

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    if constexpr (QS_PAIR4C_ABLATE & 4) return a + b + c;
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// value of lane ^ 4 (the other part of the same item)
__device__ __forceinline__ double other_part(double v) {
    i32x2 w = __builtin_bit_cast(i32x2, v);
    w.x = __builtin_amdgcn_ds_swizzle(w.x, 0x101F);       // bit mode: and 0x1f, or 0, xor 4
    w.y = __builtin_amdgcn_ds_swizzle(w.y, 0x101F);
    return __builtin_bit_cast(double, w);
}

// The STREAMED form (later in round 3), ceil(l/4) >= 7: an item pair is not staged whole.  Row quad ka of In is used by
// exactly one step -- Y[ka] = In[ka] . R, then Out[pg] += Lm[pg][ka] . Y[ka] for every pg -- so the pairs pass through a ring of
// TWO row-quad slots (2 x 7.7 KB at 56 orbitals instead of 108 KB), fetched three row quads ahead through registers: one
// 16-byte element per thread and row quad, across pair boundaries.  The whole-pair form had the loads of a pair in the
// open above 40 orbitals (no registers for a pair in flight, no LDS for a second one): compiled out, they were 12 % of the
// launch at 44 ... 48 orbitals and 26 % at 55 (profiles/r03_pair4c.txt section 5).  One workgroup barrier per step, between
// its two products: the slot written at the start of step ka (row quad ka + 1) was last read in the first product of step
// ka - 1, which every wave has left when any wave has passed that step's barrier; and it is first read -- by the operand
// read-ahead -- after this step's barrier.  The barrier waits for the wave's LDS traffic only: the fetches stay in flight.
// One wave per column group (N4 waves); products, operand order and exchange as above: bit-identical.
#ifndef QS_PAIR4S_NJ2_FROM
#define QS_PAIR4S_NJ2_FROM 13
#endif
constexpr int stream_groups(int n4) { return n4 >= QS_PAIR4S_NJ2_FROM ? 2 : 1; }            // column groups per wave (13, 14: 128 registers per wave do not hold one group's state)
constexpr int stream_waves(int n4) { return (n4 + stream_groups(n4) - 1) / stream_groups(n4); }

// REAL_IN: the items are REAL matrices (the first pass of a real tensor against complex coefficients, basis_set.py:341-342 with
// NumPy's promotion; `in` and its strides count doubles): In_im = 0, so the first product is its MFMA 1 alone -- the same
// chain as the real product with B = (R_re | R_im) that the tiled path runs for the d contraction -- and a row quad is half
// the bytes (16-byte pieces of two adjacent k; the last k of an odd L as in qs_quad4s.hip).
template <int N4, bool REAL_IN = false>
__global__ __launch_bounds__(64 * stream_waves(N4)) void pair4s_kernel(const Pair4Args g) {
    constexpr int NJ = stream_groups(N4), NW = stream_waves(N4), NTH = 64 * NW;
    constexpr int K4 = 4 * N4, Lp = (K4 % 8 == 4) ? K4 : K4 + 4;
    constexpr int ITEM = 4 * Lp;                 // == 16 mod 32 doubles (Lp == 4 mod 8)
    constexpr int PLANE = 2 * ITEM + 2;          // == 2 mod 32
    constexpr int SLOT = 2 * PLANE;
    constexpr int NR = NJ == 1 ? 3 : 2;          // ring slots: three for the two-phase form (one group per wave), see its barrier
    constexpr int TABLE = N4 * N4 * 16;
    constexpr int P = N4 > 3 ? 3 : N4 - 1;       // row quads in flight between their fetch and the ring
    static_assert(8 * K4 <= NTH, "one element per thread and row quad");
    static_assert(P < N4, "a fetch reaches into the next pair at most");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* ring = lds;                          // [2 slots][re, im][2 items][4 rows][Lp]
    double* ltab = lds + NR * SLOT;              // [re, im][pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int item = y >> 1, part = y & 1;
    const int e_lane = z * 4 + x;
    const bool tb = g.tensor_is_b != 0;
    const bool slab = g.in_col == 1;

    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = blockIdx.x / n_xcd, slots = gridDim.x / n_xcd;
    const unsigned per = (g.npairs + n_xcd - 1) / n_xcd;
    const unsigned u_end = (xcd + 1) * per < g.npairs ? (xcd + 1) * per : g.npairs;
    unsigned unit = xcd * per + slot;
    if (unit >= u_end) return;                              // (the whole workgroup, before any barrier)

    // ---- the fetch: thread t < 8 K4 owns element (item, row, k) of every row quad.  Buffer loads: the pair's base in the
    // descriptor (scalar), the element's place in the lane offset (a lane without an element is parked past num_records:
    // the hardware returns 0.0), the row quad in the scalar offset -- no vector ALU work per fetch.
    constexpr unsigned kParked = 0x80000000u;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    int f_it, f_r, f_k;
    if (REAL_IN) { f_k = 2 * (tid % (K4 / 2)); f_r = (tid / (K4 / 2)) & 3; f_it = tid / (2 * K4); }      // (slabs only)
    else if (slab) { f_k = tid % K4; f_r = (tid / K4) & 3; f_it = tid / (4 * K4); }
    else { f_it = tid & 1; f_k = (tid >> 1) % K4; f_r = (tid >> 1) / K4; }
    const bool loader = tid < (REAL_IN ? 4 : 8) * K4;
    constexpr int ESZ = REAL_IN ? 8 : 16;
    const unsigned f_off = (unsigned)(((int64_t)f_it * g.in_item + (int64_t)f_r * g.in_row + (int64_t)f_k * g.in_col) * ESZ);
    const unsigned quad_step = (unsigned)(4 * g.in_row * ESZ);
    const int f_pos = f_it * ITEM + f_r * Lp + f_k;
    // (REAL_IN: the base pointer is 8 bytes early and every offset carries + 8, so that the piece of the last k of an odd L
    // can start 8 bytes earlier -- its second half is that k, the missing one's place gets 0.0)
    const bool shift_k = REAL_IN && f_k < L && !(f_k + 1 < L);
    auto rsrc_of = [&](unsigned u) __attribute__((always_inline)) {
        const uint64_t pb = reinterpret_cast<uint64_t>(g.in + (int64_t)(u < u_end ? u : 0) * 2 * g.in_item * (REAL_IN ? 1 : 2)) - (REAL_IN ? 8 : 0);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)pb);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, 0x7fffffff, 0x00020000);
    };
    // lane offsets of pair u: [0] its row quads but the last, [1] the last one (rows beyond L do not exist)
    auto voffs_of = [&](unsigned u, unsigned (&v)[2]) __attribute__((always_inline)) {
        const bool ok = loader && u < u_end && f_k < L && u * 2 + f_it < g.nitems;
        const unsigned b = REAL_IN ? (shift_k ? f_off : f_off + 8) : f_off;
        v[0] = ok ? b : kParked;
        v[1] = ok && 4 * (N4 - 1) + f_r < L ? b : kParked;
    };
    auto fetch = [&](auto rs, const unsigned (&v)[2], auto KA) __attribute__((always_inline)) {
        constexpr int ka = decltype(KA)::value;
        if constexpr (QS_PAIR4C_ABLATE & 1) return f64x2{1.0 + ka, 0.5};
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v[ka == N4 - 1], (int)(ka * quad_step), 0);
        return __builtin_bit_cast(f64x2, q);
    };
    // a fetched piece goes to its place in a slot: (re, im) of an element into the two planes; REAL_IN: two adjacent k of the
    // re plane (the im plane is never read)
    auto settle = [&](int w, f64x2 v) __attribute__((always_inline)) {
        if (loader) {
            if constexpr (REAL_IN) { ring[w] = shift_k ? v.y : v.x; ring[w + 1] = shift_k ? 0.0 : v.y; }
            else { ring[w] = v.x; ring[w + PLANE] = v.y; }
        }
    };
    // ring slot of row quad ka of the CURRENT unit: index ka % NR into these bases, which are rotated by N4 % NR at the end of
    // a unit (the next pair's first row quad follows this one's last)
    const int a_base = item * ITEM + x * Lp + z;
    int a1_of[NR], a2_of[NR], w_of[NR];
#pragma unroll
    for (int sl = 0; sl < NR; ++sl) {
        a1_of[sl] = sl * SLOT + a_base + ((tb && part) ? PLANE : 0);     // In_re; (b contraction) In_re / In_im by part
        a2_of[sl] = sl * SLOT + a_base + ((tb && part) ? 0 : PLANE);     // In_im; (b contraction) In_im / In_re by part
        w_of[sl] = sl * SLOT + f_pos;
    }

    // ---- this wave's column groups and its B operands of the first product: registers for the whole launch
    const int jg0 = NJ * wave;
    auto has = [&](int j) __attribute__((always_inline)) { return jg0 + j < N4; };      // (wave-uniform)
    double b1[N4][NJ], b2[N4][NJ];
    f64x2 pf[P];                                 // pf[ka % P] holds row quad ka + 1 at the start of step ka
    auto rs_cur = rsrc_of(unit), rs_nx = rsrc_of(unit + slots);
    unsigned v_cur[2], v_nx[2];
    voffs_of(unit, v_cur);
    voffs_of(unit + slots, v_nx);
    {
        constexpr int NF = (TABLE + NTH - 1) / NTH;
        double l_re[NF], l_im[NF];
        unroll<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            const int e = f & 15, blk = f >> 4, hi = blk / N4, lo = blk % N4, ez = e >> 2, ex = e & 3;
            double re = 0.0, im = 0.0;
            const int p_ = 4 * hi + ex, a = 4 * lo + ez;                    // Lm[4 hi + ex][4 lo + ez]
            if (f < TABLE && p_ < M && a < L) {
                const f64x2 v = *reinterpret_cast<const f64x2*>(g.Lm + (p_ * g.l_sp + a * g.l_sa) * 2);
                re = v.x; im = v.y;
            }
            l_re[i] = re; l_im[i] = im;
        });
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int k = 4 * ks + z, col = 4 * (jg0 + j) + x;
                double re = 0.0, im = 0.0;
                if (has(j) && k < L && col < M) {
                    const f64x2 v = *reinterpret_cast<const f64x2*>(g.R + (k * g.r_sk + col * g.r_sj) * 2);
                    re = v.x; im = v.y;
                }
                b1[ks][j] = tb ? re : (part ? im : re);
                b2[ks][j] = tb ? (part ? im : -im) : (part ? re : -im);
            });
        });
        const f64x2 q0 = fetch(rs_cur, v_cur, std::integral_constant<int, 0>{});
        unroll<0, P>([&](auto I) __attribute__((always_inline)) {
            pf[decltype(I)::value] = fetch(rs_cur, v_cur, std::integral_constant<int, 1 + decltype(I)::value>{});
        });
        unroll<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            if (f < TABLE) { ltab[f] = l_re[i]; ltab[TABLE + f] = l_im[i]; }
        });
        settle(w_of[0], q0);
    }
    __syncthreads();

    for (; unit < u_end; unit += slots) {
      if constexpr (NJ == 1) {
        // TWO PHASES per item pair (one column group per wave): Y[ka] = In[ka] . R[:, group] for every row quad ka -- Y and its
        // exchanged copy stay in registers, already in B-operand layout -- then Out[pg] = sum_ka Lm[pg][ka] . Y[ka] for every
        // row quad pg, a finished row quad of Out leaving during the next one's MFMAs.  (Both products per ka, as in the
        // two-groups form below, hold all of Out to the end of the unit: every workgroup of the chip then stores a whole pair
        // at once -- in the fp64 sibling those stores were 23-38 % of the launch, profiles/r03_quad4s.txt.)
        double Y[N4], Ys[N4];
        double o2[2];
        constexpr int NG = N4 * N4;
        auto operands = [&](auto T, double& p0, double& p1) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            if constexpr (t < NG) {
                p0 = ring[a1_of[(t / N4) % NR] + 4 * (t % N4)];
                if constexpr (!REAL_IN) p1 = ring[a2_of[(t / N4) % NR] + 4 * (t % N4)];
            } else {
                p0 = ltab[(t - NG) * 16 + e_lane];                      // Lm[pg][ka], (pg, ka) = t - NG
                p1 = ltab[TABLE + (t - NG) * 16 + e_lane];
            }
        };
        constexpr int AHEAD = N4 > 2 ? 2 : 1;               // groups between the read of the operands and their MFMAs
        static_assert(AHEAD < N4, "the read-ahead into the next row quad's slot starts behind the step's barrier");
        double opr[AHEAD + 1][2];
        unroll<0, AHEAD>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            operands(T, opr[t % (AHEAD + 1)][0], opr[t % (AHEAD + 1)][1]);
        });
        const unsigned it_g = unit * 2 + item;              // D: row z, block (item, part), column x
        double* orow = g.out + ((int64_t)it_g * g.out_item + z * g.out_row + x * g.out_col + (int64_t)(4 * jg0) * g.out_col) * 2 + part;
        const bool st_ok = it_g < g.nitems && 4 * jg0 + x < M;
        unroll<0, 2 * NG>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value, sl = t % (AHEAD + 1);
            if constexpr (t < NG && t % N4 == 0) {
                // step start: row quad ka + 1 (the next pair's first after the last step) leaves its registers for its slot,
                // the fetch of row quad ka + 1 + P takes its place
                constexpr int ka = t / N4;
                settle(w_of[(ka + 1) % NR], pf[ka % P]);
                constexpr int tq = ka + 1 + P;
                if constexpr (tq < N4) pf[ka % P] = fetch(rs_cur, v_cur, std::integral_constant<int, tq>{});
                else pf[ka % P] = fetch(rs_nx, v_nx, std::integral_constant<int, tq - N4>{});
            }
            if constexpr (t < NG && t % N4 == 1) {
                // The step's barrier, right behind its write.  The slot written at the start of step ka holds row quad ka + 1:
                // it is first read -- by the operand read-ahead, from group N4 - AHEAD >= 1 of this step on -- behind this
                // barrier; and it was last read in step ka - 2, which every wave had finished when it arrived at the barrier
                // of step ka - 1 (three slots: with two, a wave still in step ka - 1 would be reading it).  It waits for the
                // wave's LDS traffic only: the fetches stay in flight.
                __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), vmcnt / expcnt untouched
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (t + AHEAD < 2 * NG)
                operands(std::integral_constant<int, t + AHEAD>{}, opr[(t + AHEAD) % (AHEAD + 1)][0], opr[(t + AHEAD) % (AHEAD + 1)][1]);
            const double cur0 = opr[sl][0], cur1 = opr[sl][1];
            if constexpr (t < NG) {                         // Y[ka] += In[ka][ks] . R[ks]
                constexpr int ka = t / N4, ks = t % N4;
                Y[ka] = mfma4(cur0, b1[ks][0], ks == 0 ? 0.0 : Y[ka]);
                if constexpr (!REAL_IN) Y[ka] = mfma4(cur1, b2[ks][0], Y[ka]);
                if constexpr (ks == N4 - 1) {
                    // second operand of Lm_im: the other part of the same item, the block that feeds the real part negated
                    const double other = other_part(Y[ka]);
                    Ys[ka] = part ? other : -other;
                }
            } else {                                        // Out[pg] += Lm[pg][ka] . Y[ka]
                constexpr int pg = (t - NG) / N4, ka = (t - NG) % N4;
                o2[pg & 1] = mfma4(cur0, Y[ka], ka == 0 ? 0.0 : o2[pg & 1]);
                o2[pg & 1] = mfma4(cur1, Ys[ka], o2[pg & 1]);
                if constexpr (ka == N4 - 1) {
                    if (st_ok && 4 * pg + z < M && (!(QS_PAIR4C_ABLATE & 2) || o2[pg & 1] == 12345.678))
                        orow[(int64_t)(4 * pg) * g.out_row * 2] = o2[pg & 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        double o[N4][NJ];
        double yv[NJ], ys[NJ];
        constexpr int GPK = 2 * N4;                         // groups per row quad of Y: N4 of the first product, N4 of the second
        auto operands = [&](auto T, double& p0, double& p1) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value, ka = t / GPK, gi = t % GPK;
            if constexpr (gi < N4) {
                p0 = ring[a1_of[ka & 1] + 4 * gi];
                if constexpr (!REAL_IN) p1 = ring[a2_of[ka & 1] + 4 * gi];
            } else {
                constexpr int f = ((gi - N4) * N4 + ka) * 16;
                p0 = ltab[f + e_lane];
                p1 = ltab[TABLE + f + e_lane];
            }
        };
        constexpr int AHEAD = (NJ == 1 && N4 > 2) ? 2 : 1;  // groups between the read of the operands and their MFMAs
        static_assert(AHEAD < N4, "the read-ahead of a step's last groups stays behind its barrier");
        double opr[AHEAD + 1][2];
        unroll<0, AHEAD>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            operands(T, opr[t % (AHEAD + 1)][0], opr[t % (AHEAD + 1)][1]);
        });
        unroll<0, N4 * GPK>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value, ka = t / GPK, gi = t % GPK, sl = t % (AHEAD + 1);
            if constexpr (gi == 0) {
                // step start: row quad ka + 1 (the next pair's first after the last step) leaves its registers for its slot,
                // the fetch of row quad ka + 1 + P takes its place
                settle(w_of[(ka + 1) & 1], pf[ka % P]);
                constexpr int tq = ka + 1 + P;
                if constexpr (tq < N4) pf[ka % P] = fetch(rs_cur, v_cur, std::integral_constant<int, tq>{});
                else pf[ka % P] = fetch(rs_nx, v_nx, std::integral_constant<int, tq - N4>{});
            }
            if constexpr (gi == N4) {                       // between the two products: the step's barrier (see above)
                __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0); vmcnt untouched: the fetches stay in flight
                __builtin_amdgcn_s_barrier();
            }
            // (the read-ahead of the last groups of the LAST step reaches into the next pair's first row quad: slot
            // N4 & 1 of this unit's numbering, written at the start of that step, behind its barrier)
            if constexpr (t + AHEAD < N4 * GPK)
                operands(std::integral_constant<int, t + AHEAD>{}, opr[(t + AHEAD) % (AHEAD + 1)][0], opr[(t + AHEAD) % (AHEAD + 1)][1]);
            const double cur0 = opr[sl][0], cur1 = opr[sl][1];
            if constexpr (gi < N4) {                        // Y[ka] += In[ka][ks] . R[ks]
                constexpr int ks = gi;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    yv[j] = mfma4(cur0, b1[ks][j], ks == 0 ? 0.0 : yv[j]);
                });
                if constexpr (!REAL_IN) {
                    unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                        constexpr int j = decltype(J)::value;
                        yv[j] = mfma4(cur1, b2[ks][j], yv[j]);
                    });
                }
                if constexpr (ks == N4 - 1) {
                    unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                        constexpr int j = decltype(J)::value;
                        const double other = other_part(yv[j]);
                        ys[j] = part ? other : -other;
                    });
                }
            } else {                                        // Out[pg] += Lm[pg][ka] . Y[ka]
                constexpr int pg = gi - N4;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    o[pg][j] = mfma4(cur0, yv[j], ka == 0 ? 0.0 : o[pg][j]);
                });
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    o[pg][j] = mfma4(cur1, ys[j], o[pg][j]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- store.  D: row z, block (item, part), column x.
        const unsigned it_g = unit * 2 + item;
        double* orow = g.out + ((int64_t)it_g * g.out_item + z * g.out_row + x * g.out_col) * 2 + part;
        unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value;
            const int row = 4 * pg + z;
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int col = 4 * (jg0 + j) + x;
                if (has(j) && it_g < g.nitems && row < M && col < M && (!(QS_PAIR4C_ABLATE & 2) || o[pg][j] == 12345.678))
                    orow[((int64_t)(4 * pg) * g.out_row + (int64_t)(4 * (jg0 + j)) * g.out_col) * 2] = o[pg][j];
            });
            __builtin_amdgcn_sched_barrier(0);
        });
      }
        // ---- the next pair: its descriptor and offsets, the fetch registers back in step (pf[i] = row quad 1 + i)
        rs_cur = rs_nx;
        v_cur[0] = v_nx[0]; v_cur[1] = v_nx[1];
        rs_nx = rsrc_of(unit + 2 * slots);
        voffs_of(unit + 2 * slots, v_nx);
        if constexpr (N4 % P != 0) {
            f64x2 t_[P];
            unroll<0, P>([&](auto I) __attribute__((always_inline)) { t_[decltype(I)::value] = pf[(N4 + decltype(I)::value) % P]; });
            unroll<0, P>([&](auto I) __attribute__((always_inline)) { pf[decltype(I)::value] = t_[decltype(I)::value]; });
        }
        if constexpr (N4 % NR != 0) {                       // the next pair's first row quad sits in slot N4 % NR of this numbering
            int t1[NR], t2[NR], tw[NR];
            unroll<0, NR>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                t1[i] = a1_of[(i + N4) % NR]; t2[i] = a2_of[(i + N4) % NR]; tw[i] = w_of[(i + N4) % NR];
            });
            unroll<0, NR>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                a1_of[i] = t1[i]; a2_of[i] = t2[i]; w_of[i] = tw[i];
            });
        }
    }
}

template <int N4, bool REAL_IN = false>
int launch_pair4s(const Pair4Args& g, hipStream_t stream) {
    constexpr int K4 = 4 * N4, Lp = (K4 % 8 == 4) ? K4 : K4 + 4;
    constexpr size_t lds = sizeof(double) * ((stream_groups(N4) == 1 ? 3 : 2) * 2 * (2 * 4 * Lp + 2) + 2 * N4 * N4 * 16);
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)pair4s_kernel<N4, REAL_IN>, lds, lds_opt_in, "hipFuncSetAttribute(pair4s)")) return rc;
    const int n_cu = device_cu_count();
    unsigned wgs = (g.npairs + 7u) / 8u * 8u;
    const unsigned cap = (unsigned)(n_cu - n_cu % 8 > 8 ? n_cu - n_cu % 8 : 8);      // one workgroup of N4 waves per CU
    if (wgs > cap) wgs = cap;
    // every byte offset inside an item pair stays below 2^31 (32-bit lane and scalar offsets of the fetch)
    const int64_t span = ((int64_t)g.in_item + (int64_t)K4 * (g.in_row > g.in_col ? g.in_row : g.in_col) * 2) * 16;
    if (span >= (int64_t(1) << 31)) return 1;
    hipLaunchKernelGGL((pair4s_kernel<N4, REAL_IN>), dim3(wgs), dim3(64 * stream_waves(N4)), lds, stream, g);
    note_dispatch(REAL_IN ? "qs::pair4s_kernel<%d, true>" : "qs::pair4s_kernel<%d, false>", N4);     // (the symbol's name in a profile)
    return launch_status("pair4s launch");
}


// The instantiations, by translation unit (each returns 1 for an N4 it does not hold)
int launch_pair4s_a(int n4, const Pair4Args& g, hipStream_t stream);      // complex items, N4 = 2 ... 9
int launch_pair4s_b(int n4, const Pair4Args& g, hipStream_t stream);      // 10 ... 12
int launch_pair4s_c(int n4, const Pair4Args& g, hipStream_t stream);      // 13, 14
int launch_pair4s_d(int n4, const Pair4Args& g, hipStream_t stream);      // 15, 16
int launch_pair4m_a(int n4, const Pair4Args& g, hipStream_t stream);      // real items, N4 = 2 ... 10
int launch_pair4m_b(int n4, const Pair4Args& g, hipStream_t stream);      // 11 ... 14

}  // namespace qs
