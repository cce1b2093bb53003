// The fused passes of the four-index transform for REAL (fp64) bases of 5 ... 96 orbitals, STREAMED (kernel and launcher;
// qs_quad4s.hip holds eligibility, dispatch and the instantiations up to 64 orbitals, qs_quad4s_w*.hip those of 65 ... 96):
//
//     Out_t = Lm . In_t . R        for every item t of a batch of L x L matrices        (ceil(L/4) == ceil(M/4) = N4, 2 ... 24)
//
//   (d, c):  item t = slab (a, b),   In_t = u[a, b, :, :]        R = C,    Lm = C^T    -> T2[a, b, :, :]
//   (b, a):  item t = column (r, s), In_t = T2[:, :, r, s]       R = Ct^T, Lm = Ct     -> out[:, :, r, s]
//
// The fp64 sibling of the streamed complex kernel (qs_pair4c.hip, pair4s_kernel; basis_set.py:341-348).  The four blocks of
// v_mfma_f64_4x4x4_4b_f64 (lane = x + 4 y + 16 z: A row x / block y / k z, B k z / block y / column x, D row z / block y /
// column x) are four ITEMS.  Row quad ka of an item quad is used by one step only -- Y[ka] = In[ka] . R, then
// Out[pg] += Lm[pg][ka] . Y[ka] for every row quad pg of the result (Y[ka] leaves the accumulators as the B operand of the
// second product) -- so the item quads are not staged whole: they pass through a ring of TWO row-quad slots in LDS, fetched
// three row quads ahead through registers (one 16-byte buffer load per thread and row quad, across quad boundaries, no
// vector ALU work), with one LDS-only workgroup barrier per step between its two products.  One wave per column group
// (N4 waves, up to 1024 threads): three to four waves per SIMD cover each other's LDS latency and the dependent MFMAs.
// Every element is the same k-ordered chain of fused multiply-adds as on the 16-wide kernels: bit-identical results.
// Against qs_small4.hip (whole quads staged, one round trip of loads in front of every quad) this form has the loads of quad
// n + 1 under the products of quad n; against qs_sandwich4*.hip (one wave per SIMD with every instruction placed by hand)
// it is the simple form: where each is used is decided by measurement (profiles/r03_quad4s.txt, qs_api.hip).
// Slot layout: items (2 p + m) at p * ITEM + m * PLANE, rows at Lp doubles, Lp == 4 (mod 8), ITEM = 4 Lp == 16 (mod 32),
// PLANE = 2 ITEM + 2 == 2 (mod 32): the 32 lanes of a half wave (4 rows x 4 items x 2 k) read 32 different bank pairs.
// Algorithmic bytes per launch: 8 (L^2 + M^2) per item.

#pragma once

#include <type_traits>

#include "qs_common.h"

namespace qs {

// f(I), f(I + 1), ... f(N - 1) with compile-time arguments, in this order (halving: the 1152 groups of a 96-orbital quad would
// exceed the template recursion depth one at a time)
template <int I, int N, class F>
__device__ __forceinline__ void unroll_q(F&& f) {
    if constexpr (N - I == 1) {
        f(std::integral_constant<int, I>{});
    } else if constexpr (N - I > 1) {
        constexpr int MID = I + (N - I) / 2;
        unroll_q<I, MID>(f);
        unroll_q<MID, N>(f);
    }
}

typedef double f64x2q __attribute__((ext_vector_type(2)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));

struct Quad4Args {
    const double* in;
    double* out;
    const double* R;      // R[k][j]  = R[k * r_sk + j * r_sj],   L x M
    const double* Lm;     // Lm[p][a] = Lm[p * l_sp + a * l_sa],  M x L
    int64_t r_sk, r_sj, l_sp, l_sa;
    int64_t in_item, in_row, in_col;       // element strides of In_t[i][k]: in_col == 1 (a slab) or in_item == 1 (a column)
    int64_t out_item, out_row, out_col;    // element strides of Out_t[p][j]
    int L, M;
    unsigned nitems, nquads;
};

// development builds only: bit mask of parts to leave out (1 the fetches, 2 the stores, 8 the step barrier -- wrong results)
#ifndef QS_QUAD4S_ABLATE
#define QS_QUAD4S_ABLATE 0
#endif

__device__ __forceinline__ double mfma4q(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// N4 = ceil(L / 4) = ceil(M / 4), 2 ... 24.  H = 1: a workgroup takes whole item quads, one wave per column group (N4 <= 16:
// 1024 threads).  H = 2 (65 ... 96 orbitals): TWO workgroups per item quad, neighbours on one XCD, each with ceil(N4 / 2)
// waves for its half of the column groups -- the groups are independent (Y[ka] and Out[pg] of a group need In, R's columns
// of that group and all of Lm), so nothing is exchanged; both fetch the quad, the second from L2.
template <int N4, int H = 1>
__global__ __launch_bounds__(64 * ((N4 + H - 1) / H)) void quad4s_kernel(const Quad4Args g) {
    constexpr int NW = (N4 + H - 1) / H, NTH = 64 * NW;
    constexpr int K4 = 4 * N4, Lp = (K4 % 8 == 4) ? K4 : K4 + 4;
    constexpr int ITEM = 4 * Lp;                 // == 16 mod 32 doubles
    constexpr int PLANE = 2 * ITEM + 2;          // == 2 mod 32
    constexpr int SLOT = 2 * PLANE;
    constexpr int NR = 3;                        // ring slots (see the hazard note at the step barrier)
    constexpr int TABLE = N4 * N4 * 16;
#ifndef QS_QUAD4S_P
#define QS_QUAD4S_P 3
#endif
#ifndef QS_QUAD4S_AHEAD
#define QS_QUAD4S_AHEAD 3
#endif
    constexpr int P = N4 > QS_QUAD4S_P ? QS_QUAD4S_P : N4 - 1;       // row quads in flight between their fetch and the ring
    static_assert(8 * K4 <= NTH, "one 16-byte element pair per thread and row quad");
    static_assert(P < N4, "a fetch reaches into the next quad at most");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* ring = lds;                          // [NR slots][item & 1][item >> 1][4 rows][Lp]
    double* ltab = lds + NR * SLOT;              // [pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int e_lane = z * 4 + x;
    const bool slab = g.in_col == 1;

    // units (item quads): every XCD a contiguous range, its workgroups neighbouring quads
    // (H = 2: workgroups 8 (2 q) + x and 8 (2 q + 1) + x of XCD x are the halves of the same quads)
    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = (blockIdx.x / n_xcd) / H, slots = (gridDim.x / n_xcd) / H;
    const int half = (int)((blockIdx.x / n_xcd) % H);
    const unsigned per = (g.nquads + n_xcd - 1) / n_xcd;
    const unsigned u_end = (xcd + 1) * per < g.nquads ? (xcd + 1) * per : g.nquads;
    unsigned unit = xcd * per + slot;
    if (unit >= u_end) return;                              // (the whole workgroup, before any barrier)

    // ---- the fetch: thread t < 8 K4 owns one 16-byte piece of every row quad -- two adjacent k of (item, row) when the
    // items are slabs, two adjacent items of (row, k) when they are columns.  Buffer loads: the quad's base in the
    // descriptor, the piece's place in the lane offset (a lane without a piece is parked past num_records: the hardware
    // returns 0.0), the row quad in the scalar offset.
    constexpr unsigned kParked = 0x80000000u;
    int f_it, f_r, f_k;                                     // first element of the piece
    if (slab) { f_k = 2 * (tid % (K4 / 2)); f_r = (tid / (K4 / 2)) & 3; f_it = tid / (2 * K4); }
    else { f_it = 2 * (tid & 1); f_k = (tid >> 1) % K4; f_r = (tid >> 1) / K4; }
    const bool loader = tid < 8 * K4;
    const unsigned f_off = (unsigned)(((int64_t)f_it * g.in_item + (int64_t)f_r * g.in_row + (int64_t)f_k * g.in_col) * 8);
    const unsigned quad_step = (unsigned)(4 * g.in_row * 8);
    auto pos_of = [&](int it, int r, int k) __attribute__((always_inline)) { return (it >> 1) * ITEM + (it & 1) * PLANE + r * Lp + k; };
    const int f_pos0 = pos_of(f_it, f_r, f_k);
    const int f_pos1 = slab ? f_pos0 + 1 : pos_of(f_it + 1, f_r, f_k);
    auto rsrc_of = [&](unsigned u) __attribute__((always_inline)) {
        const uint64_t pb = reinterpret_cast<uint64_t>(g.in + (int64_t)(u < u_end ? u : 0) * 4 * g.in_item);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)pb);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, 0x7fffffff, 0x00020000);
    };
    // Lane offsets of quad u: [0] its row quads but the last, [1] the last one (rows beyond L do not exist).  A piece whose
    // FIRST element exists and whose second does not -- the last k of an odd L in a slab, the last item of an odd item
    // count in a column -- is fetched 8 bytes earlier (memory that always exists: the base pointer is 8 bytes early and
    // every offset carries + 8) and its second half takes the first one's place; the missing element's place gets 0.0.
    const bool shift_k = slab && f_k < L && !(f_k + 1 < L);
    auto voffs_of = [&](unsigned u, unsigned (&v)[2], bool& shifted) __attribute__((always_inline)) {
        const bool it0 = u * 4 + f_it < g.nitems, it1 = slab || u * 4 + f_it + 1 < g.nitems;
        const bool ok = loader && u < u_end && f_k < L && it0;
        shifted = ok && (shift_k || !it1);
        const unsigned b = shifted ? f_off : f_off + 8;
        v[0] = ok ? b : kParked;
        v[1] = ok && 4 * (N4 - 1) + f_r < L ? b : kParked;
    };
    auto fetch = [&](auto rs, const unsigned (&v)[2], auto KA) __attribute__((always_inline)) {
        constexpr int ka = decltype(KA)::value;
        if constexpr (QS_QUAD4S_ABLATE & 1) return f64x2q{1.0 + ka, 0.5};
        const u32x4q q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v[ka == N4 - 1], (int)(ka * quad_step), 0);
        return __builtin_bit_cast(f64x2q, q);
    };
    auto settle = [&](int w0, int w1, f64x2q v, bool shifted) __attribute__((always_inline)) {
        if (loader) {
            const f64x2q pr = shifted ? f64x2q{v.y, 0.0} : v;
            // (slabs: the two elements are neighbours in the slot -- w0 is even, Lp, ITEM, PLANE and SLOT are -- one 16-byte write)
            if (slab) *reinterpret_cast<f64x2q*>(ring + w0) = pr;
            else { ring[w0] = pr.x; ring[w1] = pr.y; }
        }
    };
    // ring slot of row quad ka of the CURRENT quad: index ka % NR into these bases, which are rotated by N4 % NR at the end
    // of a unit (the next quad's first row quad follows this one's last)
    int a_of[NR], w0_of[NR], w1_of[NR];
#pragma unroll
    for (int sl = 0; sl < NR; ++sl) {
        a_of[sl] = sl * SLOT + (y >> 1) * ITEM + (y & 1) * PLANE + x * Lp + z;
        w0_of[sl] = sl * SLOT + f_pos0;
        w1_of[sl] = sl * SLOT + f_pos1;
    }

    // ---- this wave's column group and its B operands of the first product: registers for the whole launch
    const int jg = half * NW + wave;             // (a wave past the last group -- odd N4, second half -- multiplies zeros and stores nothing)
    double bq[N4];
    f64x2q pf[P];                                // pf[ka % P] holds row quad ka + 1 at the start of step ka
    auto rs_cur = rsrc_of(unit), rs_nx = rsrc_of(unit + slots);
    unsigned v_cur[2], v_nx[2];
    bool sh_cur, sh_nx;
    voffs_of(unit, v_cur, sh_cur);
    voffs_of(unit + slots, v_nx, sh_nx);
    {
        constexpr int NF = (TABLE + NTH - 1) / NTH;
        double l_v[NF];
        unroll_q<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            const int e = f & 15, blk = f >> 4, hi = blk / N4, lo = blk % N4, ez = e >> 2, ex = e & 3;
            const int p_ = 4 * hi + ex, a = 4 * lo + ez;                    // Lm[4 hi + ex][4 lo + ez]
            l_v[i] = (f < TABLE && p_ < M && a < L) ? g.Lm[p_ * g.l_sp + a * g.l_sa] : 0.0;
        });
        unroll_q<0, N4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            const int k = 4 * ks + z, col = 4 * jg + x;
            bq[ks] = (k < L && col < M) ? g.R[k * g.r_sk + col * g.r_sj] : 0.0;
        });
        const f64x2q q0 = fetch(rs_cur, v_cur, std::integral_constant<int, 0>{});
        unroll_q<0, P>([&](auto I) __attribute__((always_inline)) {
            pf[decltype(I)::value] = fetch(rs_cur, v_cur, std::integral_constant<int, 1 + decltype(I)::value>{});
        });
        unroll_q<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            if (f < TABLE) ltab[f] = l_v[i];
        });
        settle(w0_of[0], w1_of[0], q0, sh_cur);
    }
    __syncthreads();

    for (; unit < u_end; unit += slots) {
        // Two phases per item quad (as in qs_sandwich4.hip): Y[ka] = In[ka] . R[:, group] for every row quad ka -- Y stays in
        // the accumulators, already in B-operand layout -- then Out[pg] = sum_ka Lm[pg][ka] . Y[ka] for every row quad pg, a
        // finished row quad of Out leaving during the next one's MFMAs.  (The first version ran both products per ka and held
        // all of Out to the end: every workgroup of the chip stored a whole quad at once -- compiled out, those stores were
        // 23 % of the launch at 55 orbitals and 38 % at 32, profiles/r03_quad4s.txt.)  One stream of 2 N4^2 groups, one MFMA
        // and one LDS operand each, the operand read AHEAD groups before its MFMA.
        double Y[N4];
        double o2[2];
        constexpr int NG = N4 * N4;
        auto operand = [&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            if constexpr (t < NG) return ring[a_of[(t / N4) % NR] + 4 * (t % N4)];                 // In[ka][ks]
            else return ltab[(t - NG) * 16 + e_lane];                                              // Lm[pg][ka], (pg, ka) = t - NG
        };
        constexpr int AHEAD = N4 > QS_QUAD4S_AHEAD ? QS_QUAD4S_AHEAD : N4 - 1;          // groups between the read of an operand and its MFMA
        static_assert(AHEAD < N4, "the read-ahead into the next row quad's slot starts behind the step's barrier");
        double opr[AHEAD + 1];
        unroll_q<0, AHEAD>([&](auto T) __attribute__((always_inline)) { opr[decltype(T)::value % (AHEAD + 1)] = operand(T); });
        const unsigned it_g = unit * 4 + y;                 // D: row z, block y = item, column x
        double* orow = g.out + (int64_t)it_g * g.out_item + z * g.out_row + x * g.out_col + (int64_t)(4 * jg) * g.out_col;
        const bool st_ok = it_g < g.nitems && 4 * jg + x < M;
        unroll_q<0, 2 * NG>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value, sl = t % (AHEAD + 1);
            if constexpr (t < NG && t % N4 == 0) {
                // step start: row quad ka + 1 (the next quad's first after the last step) leaves its registers for its slot,
                // the fetch of row quad ka + 1 + P takes its place
                constexpr int ka = t / N4;
                settle(w0_of[(ka + 1) % NR], w1_of[(ka + 1) % NR], pf[ka % P], ka + 1 < N4 ? sh_cur : sh_nx);
                constexpr int tq = ka + 1 + P;
                if constexpr (tq < N4) pf[ka % P] = fetch(rs_cur, v_cur, std::integral_constant<int, tq>{});
                else pf[ka % P] = fetch(rs_nx, v_nx, std::integral_constant<int, tq - N4>{});
            }
            if constexpr (t < NG && t % N4 == 1 % N4 && !(QS_QUAD4S_ABLATE & 8)) {
                // The step's barrier, right behind its write.  The slot written at the start of step ka holds row quad ka + 1:
                // it is first read -- by the operand read-ahead, from group N4 - AHEAD >= 1 of this step on -- behind this
                // barrier; and it was last read in step ka - 2, which every wave had finished when it arrived at the barrier
                // of step ka - 1 (three slots: with two, a wave still in step ka - 1 would be reading it).  It waits for the
                // wave's LDS traffic only: the fetches stay in flight.
                __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), vmcnt / expcnt untouched
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (t + AHEAD < 2 * NG) opr[(t + AHEAD) % (AHEAD + 1)] = operand(std::integral_constant<int, t + AHEAD>{});
            const double cur = opr[sl];
            if constexpr (t < NG) {                         // Y[ka] += In[ka][ks] . R[ks]
                constexpr int ka = t / N4, ks = t % N4;
                Y[ka] = mfma4q(cur, bq[ks], ks == 0 ? 0.0 : Y[ka]);
            } else {                                        // Out[pg] += Lm[pg][ka] . Y[ka]
                constexpr int pg = (t - NG) / N4, ka = (t - NG) % N4;
                o2[pg & 1] = mfma4q(cur, Y[ka], ka == 0 ? 0.0 : o2[pg & 1]);
                if constexpr (ka == N4 - 1) {
                    if (st_ok && 4 * pg + z < M && (!(QS_QUAD4S_ABLATE & 2) || o2[pg & 1] == 12345.678))
                        orow[(int64_t)(4 * pg) * g.out_row] = o2[pg & 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- the next quad: its descriptor and offsets, the fetch registers back in step (pf[i] = row quad 1 + i)
        rs_cur = rs_nx;
        v_cur[0] = v_nx[0]; v_cur[1] = v_nx[1];
        sh_cur = sh_nx;
        rs_nx = rsrc_of(unit + 2 * slots);
        voffs_of(unit + 2 * slots, v_nx, sh_nx);
        if constexpr (N4 % P != 0) {
            f64x2q t_[P];
            unroll_q<0, P>([&](auto I) __attribute__((always_inline)) { t_[decltype(I)::value] = pf[(N4 + decltype(I)::value) % P]; });
            unroll_q<0, P>([&](auto I) __attribute__((always_inline)) { pf[decltype(I)::value] = t_[decltype(I)::value]; });
        }
        if constexpr (N4 % NR != 0) {                       // the next quad's first row quad sits in slot N4 % NR of this numbering
            int ta[NR], t0[NR], t1[NR];
            unroll_q<0, NR>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                ta[i] = a_of[(i + N4) % NR]; t0[i] = w0_of[(i + N4) % NR]; t1[i] = w1_of[(i + N4) % NR];
            });
            unroll_q<0, NR>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                a_of[i] = ta[i]; w0_of[i] = t0[i]; w1_of[i] = t1[i];
            });
        }
    }
}

template <int N4, int H = 1>
int launch_quad4s(const Quad4Args& g, hipStream_t stream) {
    constexpr int NW = (N4 + H - 1) / H;
    constexpr int K4 = 4 * N4, Lp = (K4 % 8 == 4) ? K4 : K4 + 4;
    constexpr size_t lds = sizeof(double) * (3 * 2 * (2 * 4 * Lp + 2) + N4 * N4 * 16);
    // every byte offset inside an item quad stays below 2^31 (32-bit lane and scalar offsets of the fetch)
    const int64_t span = (3 * (int64_t)g.in_item + (int64_t)K4 * (g.in_row > g.in_col ? g.in_row : g.in_col) * 2) * 8 + 16;
    if (span >= (int64_t(1) << 31)) return 1;
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)quad4s_kernel<N4, H>, lds, lds_opt_in, "hipFuncSetAttribute(quad4s)")) return rc;
    const int n_cu = device_cu_count();
    unsigned wgs = (g.nquads + 7u) / 8u * 8u * H;       // (a multiple of 8 H: whole quads per XCD slot pair)
    // workgroups resident per CU: by threads (2048 per CU) and LDS
#ifndef QS_QUAD4S_TWO_PER_CU_TO
#define QS_QUAD4S_TWO_PER_CU_TO 8
#endif
    const unsigned per_cu = (N4 <= QS_QUAD4S_TWO_PER_CU_TO && lds <= 80 * 1024) ? 2u : 1u;
    const unsigned cap = per_cu * (unsigned)(n_cu - n_cu % 8 > 8 ? n_cu - n_cu % 8 : 8);
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL((quad4s_kernel<N4, H>), dim3(wgs), dim3(64 * NW), lds, stream, g);
    note_dispatch(H == 1 ? "qs::quad4s_kernel<%d>" : "qs::quad4s_kernel<%d, 2>", N4);
    return launch_status("quad4s launch");
}


// instantiations for 65 ... 96 orbitals (two workgroups per item quad), by translation unit; 1 = not held here
int launch_quad4s_w1(int n4, const Quad4Args& g, hipStream_t stream);      // N4 = 17 ... 20
int launch_quad4s_w2(int n4, const Quad4Args& g, hipStream_t stream);      // N4 = 21 ... 24

}  // namespace qs
