// The four-index transform of a SMALL basis (L, M <= 32, ceil(L/4) == ceil(M/4)), fp64 and complex128, as TWO launches:
//
//     Out_t = Lm . In_t . R        for every item t of a batch of L x L matrices
//
//   (d, c):  item t = slab (a, b),   In_t = u[a, b, :, :]        R = C,    Lm = C^T    -> T2[a, b, :, :]
//   (b, a):  item t = column (r, s), In_t = T2[:, :, r, s]       R = Ct^T, Lm = Ct     -> out[:, :, r, s]
//
// (the form qs_sandwich4.hip runs for 33 ... 64 real orbitals; basis_set.py:341-348).  Below ~33 orbitals the transform is
// not work but launches: the 16-wide kernels need three to five dependent launches of ~5 us whatever the size (l = 20:
// 14.6 us, profiles/r02_small_basis_sweep.txt), and RandomBasisSet -- the reference's own test input,
// random_basis.py:52-69 -- and every spin-doubled tensor are complex, which the fused kernels of round 2 do not take.
// This kernel is the simple end of the same idea, written for latency instead of throughput:
//   * v_mfma_f64_4x4x4_4b_f64, four ITEMS per instruction (lane = x + 4 y + 16 z: A row x / item y / k z, B k z / item y /
//     column x, D row z / item y / column x): extents pad to 4, and Y = In . R leaves the accumulators in the B-operand
//     layout of Lm . Y -- no LDS round trip between the two products;
//   * a workgroup takes an item quad: all four items (<= 64 KB) are staged into LDS with coalesced loads, zero-padded, and
//     every wave reads its A fragments from there (conflict-free pitches); the fragments of R and Lm come from LDS tables
//     built once per workgroup; a wave owns one or two column groups of four columns;
//   * complex128: re / im planes, four real MFMAs per fragment pair in the order of the 16-wide kernels (qs_gemm.hip
//     mfma_step: re += ar.br, im += ar.bi, re += (-ai).bi, im += ai.br, with A / B the operands of THAT kernel's call) --
//     every element is the same chain of fused multiply-adds in the same order, so results are bit-identical to the
//     16-wide path (tests/test_gpu_kernels.py), for both dtypes.
// Why two launches and not one with a grid-wide barrier between the passes: measured (tools/probe_gridsync.hip,
// profiles/r03_grid_barrier_probe.txt) a dependent launch costs 2.6 us, a hand-written agent-scope barrier 1.7 us for 16
// workgroups but 3.5 / 6.2 / 11 us for 64 / 128 / 256, and the cooperative-groups grid sync 24-33 us.
// Algorithmic bytes per launch: e (L^2 + M^2) per item (e = 8 / 16); the tensor (<= 16.8 MB) lives in L2 / Infinity Cache.

#include <type_traits>

#include "qs_common.h"

// Development builds only (never defined in the shipped library): bit mask of parts to leave out, to find what bounds the
// kernel.  1 loads of the item quad, 2 stores, 4 all MFMAs, 8 LDS reads of A fragments
#ifndef QS_SMALL4_ABLATE
#define QS_SMALL4_ABLATE 0
#endif

namespace qs {

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void unroll(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll<I + 1, N>(f);
    }
}

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct Small4Args {
    const double* in;
    double* out;
    const double* R;      // R[k][j]  = R[k * r_sk + j * r_sj],   L x M
    const double* Lm;     // Lm[p][a] = Lm[p * l_sp + a * l_sa],  M x L
    int64_t r_sk, r_sj, l_sp, l_sa;
    int64_t in_item, in_row, in_col;       // element strides of In_t[i][k]: in_col == 1 (a slab) or in_item == 1 (a column)
    int64_t out_item, out_row, out_col;    // element strides of Out_t[p][j]
    int L, M;
    unsigned nitems, nquads;
    int tensor_is_b;      // complex only: in the 16-wide kernels' call for the FIRST product the tensor is the B operand
                          // (the b contraction, gemm(Ct, T2)): the two imaginary-part products then come in the other order
};

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    if constexpr (QS_SMALL4_ABLATE & 4) return a + b + c;
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// LDS geometry for N4 = ceil(L / 4): row pitch Lp (doubles) with Lp % 8 == 4 and item pitch == 16 mod 32, so that the
// 64 lanes of an A-fragment read (item y, row x, k z: y * item pitch + x * Lp + z) fall on 32 different 8-byte bank
// pairs twice -- the minimum for 512 bytes.
template <int N4>
struct Geo {
    static constexpr int K4 = 4 * N4;
    static constexpr int Lp = (K4 % 8 == 4) ? K4 : K4 + 4;
    static constexpr int item_raw = K4 * Lp;
    static constexpr int item_pitch = item_raw + ((16 - item_raw % 32) + 32) % 32;
    static constexpr int plane = 4 * item_pitch;           // one plane (re or im) of an item quad
    static constexpr int table = N4 * N4 * 16;             // one table of fragments
};

}  // namespace

// How the column groups of a quad are shared out (measured, profiles/r03_small4.txt): a wave takes NJ groups at a time (an
// A fragment read from LDS feeds NJ MFMAs: with NJ = 1 four fp64 waves would ask LDS for 128 bytes per clock, its
// peak), NW waves per workgroup.  Up to 16 orbitals one group per wave keeps all four waves busy; complex128 reads an A
// fragment per two MFMAs, so eight waves with one group each halve the longest chain of a workgroup.  (LDS reads 256
// bytes per clock for 8-byte accesses on gfx950, not 128: the same eight-wave split for fp64 above 16 orbitals measured
// 3-9 % faster, l = 17 10.5 -> 9.7 us, 20 11.1 -> 10.2, 28 20.1 -> 18.3, and is the default since; QS_SMALL4_WIDE_F64=0
// builds the four-wave form.)
#ifndef QS_SMALL4_WIDE_F64
#define QS_SMALL4_WIDE_F64 1
#endif
template <bool CX, int N4>
struct Split {
    static constexpr int NJ = (CX || QS_SMALL4_WIDE_F64 || N4 <= 4) ? 1 : 2;
    static constexpr int NW = ((CX || QS_SMALL4_WIDE_F64) && N4 > 4) ? 8 : 4;
};

constexpr int small4_threads(bool cx, int n4) { return ((cx || QS_SMALL4_WIDE_F64) && n4 > 4) ? 512 : 256; }

// N4 = ceil(L / 4) = ceil(M / 4), 1 ... 8
template <bool CX, int N4>
__global__ __launch_bounds__(small4_threads(CX, N4)) void small4_kernel(const Small4Args g) {
    static_assert(small4_threads(CX, N4) == 64 * Split<CX, N4>::NW, "launch bounds follow the split");
    using G = Geo<N4>;
    constexpr int K4 = G::K4, Lp = G::Lp, NPL = CX ? 2 : 1, NT = CX ? 3 : 1;
    constexpr int NJ = Split<CX, N4>::NJ;                   // column groups a wave works on at a time
    constexpr int NW = Split<CX, N4>::NW, NTH = 64 * NW;    // waves, threads of a workgroup
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* in_pl = lds;                                    // [NPL][4 items][K4 rows][Lp]
    double* rtab = lds + NPL * G::plane;                    // [NT][ks][jg][16]: re, (im, -im): R[4 ks + z][4 jg + x] at z * 4 + x
    double* ltab = rtab + NT * G::table;                    // [NT][pg][ka][16]: re, (im, -im): Lm[4 pg + x][4 ka + z] at z * 4 + x
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int e_lane = z * 4 + x;
    constexpr int es = CX ? 2 : 1;                          // doubles per element in memory

    const bool slab = g.in_col == 1;
    const int a_lane = y * G::item_pitch + x * Lp + z;      // A fragment (ka, ks): + 4 ka Lp + 4 ks

    // An item quad on its way into LDS, zero-padded to K4 x K4 per item: every thread issues ALL its loads, then writes them
    // (a load - store - load chain would pay the memory latency once per element: the first version of this kernel spent
    // 15 of its 16 us per launch at l = 32 there).  The fastest lane index is the one contiguous in memory.
    constexpr int ITER = (4 * K4 * K4 + NTH - 1) / NTH;
    double s_re[ITER], s_im[ITER];
    auto stage_load = [&](unsigned unit) __attribute__((always_inline)) {
        const double* base = g.in + (int64_t)unit * 4 * g.in_item * es;
        unroll<0, ITER>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int idx = tid + NTH * i;
            int item, row, k;
            if (slab) { k = idx % K4; row = (idx / K4) % K4; item = idx / (K4 * K4); }
            else { item = idx & 3; k = (idx >> 2) % K4; row = (idx >> 2) / K4; }
            double re = 0.0, im = 0.0;
            if (!(QS_SMALL4_ABLATE & 1) && idx < 4 * K4 * K4 && row < L && k < L && unit * 4 + item < g.nitems) {
                const double* p = base + (item * g.in_item + row * g.in_row + k * g.in_col) * es;
                if constexpr (CX) { const f64x2 v = *reinterpret_cast<const f64x2*>(p); re = v.x; im = v.y; }
                else re = *p;
            }
            s_re[i] = re;
            s_im[i] = im;
        });
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
        unroll<0, ITER>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int idx = tid + NTH * i;
            int item, row, k;
            if (slab) { k = idx % K4; row = (idx / K4) % K4; item = idx / (K4 * K4); }
            else { item = idx & 3; k = (idx >> 2) % K4; row = (idx >> 2) / K4; }
            if (idx < 4 * K4 * K4) {
                const int at = item * G::item_pitch + row * Lp + k;
                in_pl[at] = s_re[i];
                if constexpr (CX) in_pl[G::plane + at] = s_im[i];
            }
        });
    };

    // Work units (item quads): every XCD takes a contiguous range and its workgroups neighbouring quads.  In the (b, a) pass
    // the four items of a quad are 32 bytes of a 128-byte line whose other quarters belong to the next three quads: on one
    // XCD they meet in its L2 (one fetch, one whole-line write-back); spread round-robin over the XCDs -- the first
    // version -- every line was fetched by four L2s and written back as four partial lines (at l = 32: 12 of 29 us in the
    // loads, 16 in the stores; profiles/r03_small4.txt).
    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = blockIdx.x / n_xcd, slots = gridDim.x / n_xcd;
    const unsigned per = (g.nquads + n_xcd - 1) / n_xcd;
    const unsigned u_end = (xcd + 1) * per < g.nquads ? (xcd + 1) * per : g.nquads;
    unsigned unit = xcd * per + slot;
    if (unit >= u_end) return;                              // (the whole workgroup, before any barrier)
    {   // ---- tables: every element of R and Lm once per workgroup (they sit in L2 after the first workgroup); the loads of
        // the tables and of the first item quad all go out before anything is written
        constexpr int NF = (G::table + NTH - 1) / NTH;
        double r_re[NF], r_im[NF], l_re[NF], l_im[NF];
        unroll<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            const int e = f & 15, blk = f >> 4, hi = blk / N4, lo = blk % N4, ez = e >> 2, ex = e & 3;
            double re = 0.0, im = 0.0;
            const int k = 4 * hi + ez, j = 4 * lo + ex;                     // R[4 hi + ez][4 lo + ex]
            if (f < G::table && k < L && j < M) {
                const double* p = g.R + (k * g.r_sk + j * g.r_sj) * es;
                if constexpr (CX) { const f64x2 v = *reinterpret_cast<const f64x2*>(p); re = v.x; im = v.y; }
                else re = *p;
            }
            r_re[i] = re; r_im[i] = im;
            re = 0.0; im = 0.0;
            const int p_ = 4 * hi + ex, a = 4 * lo + ez;                    // Lm[4 hi + ex][4 lo + ez]
            if (f < G::table && p_ < M && a < L) {
                const double* p = g.Lm + (p_ * g.l_sp + a * g.l_sa) * es;
                if constexpr (CX) { const f64x2 v = *reinterpret_cast<const f64x2*>(p); re = v.x; im = v.y; }
                else re = *p;
            }
            l_re[i] = re; l_im[i] = im;
        });
        stage_load(unit);
        unroll<0, NF>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int f = tid + NTH * i;
            if (f < G::table) {
                rtab[f] = r_re[i];
                ltab[f] = l_re[i];
                if constexpr (CX) {
                    rtab[G::table + f] = r_im[i]; rtab[2 * G::table + f] = -r_im[i];
                    ltab[G::table + f] = l_im[i]; ltab[2 * G::table + f] = -l_im[i];
                }
            }
        });
    }

    for (; unit < u_end; unit += slots) {
        __syncthreads();                                    // the previous quad's readers are done
        stage_store();
        __syncthreads();
        if (unit + slots < u_end) stage_load(unit + slots);                 // the next quad travels during this one's products

        // ---- a wave takes column groups jg0, jg0 + 1 (the second may not exist: wave-uniform)
        for (int jg0 = NJ * wave; jg0 < N4; jg0 += NJ * NW) {
            const bool two = jg0 + 1 < N4;
            // fragments of R for these column groups (B operand: k z, item y -- the same for every item --, column x)
            double br[N4][NJ], bi[N4][NJ], bn[N4][NJ];
            unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    const int f = (ks * N4 + (j == 0 || two ? jg0 + j : jg0)) * 16 + e_lane;
                    br[ks][j] = rtab[f];
                    if constexpr (CX) { bi[ks][j] = rtab[G::table + f]; bn[ks][j] = rtab[2 * G::table + f]; }
                });
            });
            // ---- Y[ka] = In[ka] . R[:, groups]: stays in the accumulators, which are the B operand of the second product
            double yr[N4][NJ], yi[N4][NJ];
            unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
                constexpr int ka = decltype(KA)::value;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) { yr[ka][decltype(J)::value] = 0.0; yi[ka][decltype(J)::value] = 0.0; });
                unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                    constexpr int ks = decltype(KS)::value;
                    const double ar = (QS_SMALL4_ABLATE & 8) ? 1.0 + x : in_pl[a_lane + 4 * ka * Lp + 4 * ks];
                    if constexpr (!CX) {
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            yr[ka][j] = mfma4(ar, br[ks][j], yr[ka][j]);
                        });
                    } else {
                        const double ai = in_pl[G::plane + a_lane + 4 * ka * Lp + 4 * ks];
                        // re: ar.br then ai.(-bi) in both roles; im: (A = tensor) ar.bi then ai.br, (B = tensor) the other way
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            yr[ka][j] = mfma4(ar, br[ks][j], yr[ka][j]);
                        });
                        if (!g.tensor_is_b) {
                            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                                constexpr int j = decltype(J)::value;
                                yi[ka][j] = mfma4(ar, bi[ks][j], yi[ka][j]);
                            });
                        } else {
                            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                                constexpr int j = decltype(J)::value;
                                yi[ka][j] = mfma4(ai, br[ks][j], yi[ka][j]);
                            });
                        }
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            yr[ka][j] = mfma4(ai, bn[ks][j], yr[ka][j]);
                        });
                        if (!g.tensor_is_b) {
                            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                                constexpr int j = decltype(J)::value;
                                yi[ka][j] = mfma4(ai, br[ks][j], yi[ka][j]);
                            });
                        } else {
                            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                                constexpr int j = decltype(J)::value;
                                yi[ka][j] = mfma4(ar, bi[ks][j], yi[ka][j]);
                            });
                        }
                    }
                });
            });
            // ---- Out[pg] = sum_ka Lm[pg][ka] . Y[ka] (A = Lm: the coefficient matrix, as in the 16-wide kernels' calls), stored
            // row quad by row quad.  D: row z, item y, column x.
            const unsigned q4 = unit * 4 + y;
            double* orow = g.out + ((int64_t)q4 * g.out_item + z * g.out_row + x * g.out_col) * es;
            unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
                constexpr int pg = decltype(PG)::value;
                double o_r[NJ], o_i[NJ];
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) { o_r[decltype(J)::value] = 0.0; o_i[decltype(J)::value] = 0.0; });
                unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
                    constexpr int ka = decltype(KA)::value;
                    const int f = (pg * N4 + ka) * 16 + e_lane;
                    const double lr = ltab[f];
                    if constexpr (!CX) {
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            o_r[j] = mfma4(lr, yr[ka][j], o_r[j]);
                        });
                    } else {
                        const double li = ltab[G::table + f], ln = ltab[2 * G::table + f];
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            o_r[j] = mfma4(lr, yr[ka][j], o_r[j]);
                            o_i[j] = mfma4(lr, yi[ka][j], o_i[j]);
                        });
                        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                            constexpr int j = decltype(J)::value;
                            o_r[j] = mfma4(ln, yi[ka][j], o_r[j]);
                            o_i[j] = mfma4(li, yr[ka][j], o_i[j]);
                        });
                    }
                });
                const int row = 4 * pg + z;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    const int col = 4 * (jg0 + j) + x;
                    if (!(QS_SMALL4_ABLATE & 2) && (j == 0 || two) && q4 < g.nitems && row < M && col < M) {
                        double* p = orow + ((int64_t)(4 * pg) * g.out_row + (int64_t)(4 * (jg0 + j)) * g.out_col) * es;
                        if constexpr (CX) *reinterpret_cast<f64x2*>(p) = f64x2{o_r[j], o_i[j]};
                        else *p = o_r[j];
                    }
                });
            });
        }
    }
}

template <bool CX, int N4>
static int launch_small4(const Small4Args& g, hipStream_t stream) {
    using G = Geo<N4>;
    const size_t lds = sizeof(double) * ((CX ? 2 : 1) * G::plane + 2 * (CX ? 3 : 1) * G::table);
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)small4_kernel<CX, N4>, lds, lds_opt_in, "hipFuncSetAttribute(small4)"))
        return rc;
    const int n_cu = device_cu_count();
    // an item quad per workgroup; above two quads per CU the workgroups walk the list (the tables are built once each).
    // Whole multiples of the eight XCDs: an XCD's workgroups share its range of quads.
    unsigned wgs = (g.nquads + 7u) / 8u * 8u;
    const unsigned cap = 2u * (unsigned)(n_cu - n_cu % 8 > 8 ? n_cu - n_cu % 8 : 8);
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL((small4_kernel<CX, N4>), dim3(wgs), dim3(64 * Split<CX, N4>::NW), lds, stream, g);
    note_dispatch("qs::small4_kernel<%s, %d>", CX ? "true" : "false", N4);
    return launch_status("small4 launch");
}

// Out_t = Lm . In_t . R for t < nitems (element strides); QS_OK / error after launching, 1 = not eligible.
int small4_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, int tensor_is_b, hipStream_t stream) {
    if (L < 1 || M < 1 || L > 32 || M > 32) return 1;
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;
    if (nitems < 1 || nitems >= (int64_t(1) << 31)) return 1;
    if (in_col != 1 && in_item != 1) return 1;
    if (dtype == QS_C128 && (!aligned(in, 16) || !aligned(out, 16) || !aligned(R, 16) || !aligned(Lm, 16))) return 1;
    Small4Args g;
    g.in = (const double*)in; g.out = (double*)out;
    g.R = (const double*)R; g.Lm = (const double*)Lm;
    g.r_sk = r_sk; g.r_sj = r_sj; g.l_sp = l_sp; g.l_sa = l_sa;
    g.in_item = in_item; g.in_row = in_row; g.in_col = in_col;
    g.out_item = out_item; g.out_row = out_row; g.out_col = out_col;
    g.L = (int)L; g.M = (int)M;
    g.nitems = (unsigned)nitems;
    g.nquads = (unsigned)cdiv(nitems, 4);
    g.tensor_is_b = tensor_is_b;
    const bool cx = dtype == QS_C128;
    switch (n4) {
#define QS_SMALL4_CASE(N) case N: return cx ? launch_small4<true, N>(g, stream) : launch_small4<false, N>(g, stream);
        QS_SMALL4_CASE(1) QS_SMALL4_CASE(2) QS_SMALL4_CASE(3) QS_SMALL4_CASE(4)
        QS_SMALL4_CASE(5) QS_SMALL4_CASE(6) QS_SMALL4_CASE(7) QS_SMALL4_CASE(8)
#undef QS_SMALL4_CASE
        default: return 1;
    }
}

}  // namespace qs
