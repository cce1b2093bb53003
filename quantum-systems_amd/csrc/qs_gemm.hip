// Batched row-major matrix product on the gfx950 fp64 matrix cores, real
// fp64 and complex128 (interleaved re/im) in one kernel template.
//
//   C[b] (m x n) = A[b] (m x k) . B[b] (k x n)
//
// This is the one compute kernel of the four-index transform: each of the four
// single-index contractions of quantum_systems/basis_set.py:341-348 is this
// product with the rank-4 tensor viewed as a (batched) row-major matrix, so
// every global access of the big tensor is a contiguous run along its last
// axis (DESIGN.md "contraction -> GEMM map").
//
// Machine mapping (MI355X / CDNA4):
//   * v_mfma_f64_16x16x4_f64: one wave owns a (16*TM) x (16*TN) block of C in
//     TM*TN accumulators of 4 fp64 each (8 VGPRs); operand fragments are one
//     fp64 per lane: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15];
//     result reg r of a lane holds C[(lane>>4) + 4r][lane&15].
//   * complex128: re and im are split into two LDS planes while staging, the
//     product is the plain 4-multiply form on four MFMAs per fragment pair
//     (re += ar*br; re += (-ai)*bi; im += ar*bi; im += ai*br) -- no 3M trick,
//     so rounding behaves like the reference's zgemm.
//   * a workgroup of WM x WN waves computes a (16*TM*WM) x (16*TN*WN) tile; K
//     is walked KT at a time through two LDS stages; the next stage is fetched
//     global->registers before the current one feeds the MFMAs and is written
//     to LDS after them (one barrier per K step).
//   * LDS rows are padded so both fragment reads are conflict-free
//     ds_read_b64 (A row stride KT+2 doubles, B row stride BN+16).
//   * 4-wave workgroups run two per CU, so one workgroup's epilogue stores and
//     prologue loads hide under the other's MFMAs.
//   * blockIdx -> tile mapping hands every XCD a contiguous run of tiles and
//     walks the tiles that share a panel of the streamed operand first, so the
//     second reader of a panel hits that XCD's L2.
//
// fp64 MFMA issues one 16x16x4 (2048 flop) per 64 cycles per SIMD:
// 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz = 78.6 TFLOP/s peak.

#include <type_traits>

#include <cmath>

#include "qs_common.h"
#include "qs_fast_items.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

enum { MODE_F64_SCALAR = 0, MODE_F64_VEC2 = 1, MODE_C128 = 2 };

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int64_t sa, sb, sc;      // elements
    int m, n, k;
    int tiles_m, tiles_n;
    int group_along_m;   // 1: tiles that share a B panel (same n-tile) are adjacent
    int accumulate;      // 1: C += A.B (C is read in the epilogue), 0: C = A.B
};

// Work index of a workgroup: XCD x gets the x-th contiguous chunk of the work
// list (bijective for every grid size); consecutive slots of one XCD are
// consecutive work items.  Placement only affects speed, never results.
__device__ __forceinline__ unsigned xcd_chunked_index(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int WM, int WN, int TM, int TN, int KT, int MODE, bool PIPE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4) ? 2 : 1)
void gemm_kernel(const GemmArgs g) {
    constexpr bool CX = (MODE == MODE_C128);
    constexpr bool SCALAR = (MODE == MODE_F64_SCALAR);
    constexpr int NP = CX ? 2 : 1;            // LDS planes (re, im)
    constexpr int ES = CX ? 2 : 1;            // doubles per element
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 16 * TM * WM;
    constexpr int BN = 16 * TN * WN;
    constexpr int SA = KT + 2;                // (SA/2) odd -> 16 rows hit 16 distinct bank pairs
    constexpr int SB = BN + 16;               // consecutive k rows land 16 bank pairs apart
    static_assert(KT == 8 || KT == 16, "KT");   // KT/4 k-steps per stage, must be even
    // staging: one item = 16 bytes (f64x2 or one complex) except scalar mode (8 bytes)
    constexpr int IPR_A = SCALAR ? KT : (CX ? KT : KT / 2);   // items per A row
    constexpr int IPR_B = SCALAR ? BN : (CX ? BN : BN / 2);   // items per B row
    constexpr int NA = BM * IPR_A / NT;
    constexpr int NB = KT * IPR_B / NT;
    static_assert(NA * NT == BM * IPR_A && NA > 0, "A stage not divisible");
    static_assert(NB * NT == KT * IPR_B && NB > 0, "B stage not divisible");
    constexpr int A_STAGE = NP * BM * SA;     // doubles
    constexpr int B_STAGE = NP * KT * SB;

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                        // [2][NP][BM][SA]
    double* Bs = smem + 2 * A_STAGE;          // [2][NP][KT][SB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    unsigned w = xcd_chunked_index(blockIdx.x, gridDim.x);
    int mt, nt;
    if (g.group_along_m) {
        mt = w % g.tiles_m; w /= g.tiles_m;
        nt = w % g.tiles_n; w /= g.tiles_n;
    } else {
        nt = w % g.tiles_n; w /= g.tiles_n;
        mt = w % g.tiles_m; w /= g.tiles_m;
    }
    const int64_t b = w;
    const int m0 = mt * BM, n0 = nt * BN;
    const double* __restrict__ A = g.A + b * g.sa * ES;
    const double* __restrict__ B = g.B + b * g.sb * ES;
    double* __restrict__ C = g.C + b * g.sc * ES;
    const int M = g.m, N = g.n, K = g.k;

    typedef typename std::conditional<SCALAR, double, f64x2>::type item_t;
    item_t ra[NA], rb[NB];

    // Staging.  Rows of A beyond m and columns of B beyond n only feed output rows / columns
    // that are never stored (an MFMA output row depends on its own A row only, a column on its
    // own B column), so those loads are simply CLAMPED to the last valid row / column: no
    // zero fill, no per-stage address arithmetic -- each item is one base pointer (set up
    // once) plus a uniform step.  Only the K edge needs zeros, and only in the final stage of
    // a K that is not a multiple of KT: that stage takes the masked variants.  stash() sits
    // behind a scheduling fence so that the wait for the loads cannot drift up in front of
    // the MFMAs that cover their latency.
    constexpr int VW = (SCALAR || CX) ? 1 : 2;   // elements per staged item
    const double* pa[NA];
    const double* pb[NB];
    int kca[NA], krb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = tid + i * NT;
        const int row = min(m0 + c / IPR_A, M - 1);
        kca[i] = (c % IPR_A) * VW;
        pa[i] = A + ((int64_t)row * g.lda + kca[i]) * ES;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int c = tid + i * NT;
        krb[i] = c / IPR_B;
        const int col = min(n0 + (c % IPR_B) * VW, N - VW);
        pb[i] = B + ((int64_t)krb[i] * g.ldb + col) * ES;
    }
    const int64_t b_kstep = g.ldb * ES;          // doubles per unit of k in B

    auto fetch = [&](int k0) {
        if (k0 + KT <= K) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const double* p = pa[i] + (int64_t)k0 * ES;
                if constexpr (SCALAR) ra[i] = *p; else ra[i] = *reinterpret_cast<const f64x2*>(p);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double* p = pb[i] + k0 * b_kstep;
                if constexpr (SCALAR) rb[i] = *p; else rb[i] = *reinterpret_cast<const f64x2*>(p);
            }
        } else {   // K edge: keep every address inside the operand (the values are zeroed in stash)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const double* p = pa[i] + (int64_t)((k0 + kca[i] < K) ? k0 : 0) * ES - (k0 + kca[i] < K ? 0 : kca[i] * ES);
                if constexpr (SCALAR) ra[i] = *p; else ra[i] = *reinterpret_cast<const f64x2*>(p);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double* p = pb[i] + ((k0 + krb[i] < K) ? k0 : -krb[i]) * b_kstep;
                if constexpr (SCALAR) rb[i] = *p; else rb[i] = *reinterpret_cast<const f64x2*>(p);
            }
        }
    };

    auto stash = [&](int buf, int k0) {
        __builtin_amdgcn_sched_barrier(0);
        double* as = As + buf * A_STAGE;
        double* bs = Bs + buf * B_STAGE;
        const bool edge = k0 + KT > K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + i * NT;
            double* d = as + (c / IPR_A) * SA + kca[i];
            item_t v = ra[i];
            if (edge && k0 + kca[i] >= K) {
                if constexpr (SCALAR) v = 0.0; else v = f64x2{0.0, 0.0};
            }
            if constexpr (SCALAR) *d = v;
            else if constexpr (CX) { d[0] = v[0]; d[BM * SA] = v[1]; }
            else *reinterpret_cast<f64x2*>(d) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = tid + i * NT;
            double* d = bs + krb[i] * SB + (c % IPR_B) * VW;
            item_t v = rb[i];
            if (edge && k0 + krb[i] >= K) {
                if constexpr (SCALAR) v = 0.0; else v = f64x2{0.0, 0.0};
            }
            if constexpr (SCALAR) *d = v;
            else if constexpr (CX) { d[0] = v[0]; d[KT * SB] = v[1]; }
            else *reinterpret_cast<f64x2*>(d) = v;
        }
    };

    f64x4 acc[NP][TM][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[p][i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + KT - 1) / KT;
    const int a_off = (wm * 16 * TM + (lane & 15)) * SA + (lane >> 4);
    const int b_off = (lane >> 4) * SB + wn * 16 * TN + (lane & 15);
    constexpr int KS = KT / 4;   // MFMA k-steps per stage

    auto read_frags = [&](int buf, int kk, double (&af)[NP][TM], double (&bf)[NP][TN]) {
        const double* as = As + buf * A_STAGE + a_off;
        const double* bs = Bs + buf * B_STAGE + b_off;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[p][i] = as[p * BM * SA + i * 16 * SA + kk * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[p][j] = bs[p * KT * SB + kk * 4 * SB + j * 16];
        }
    };
    auto mfma_step = [&](const double (&af)[NP][TM], const double (&bf)[NP][TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!CX) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[0][j], acc[0][i][j], 0, 0, 0);
                } else {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[0][j], acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[1][j], acc[1][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-af[1][i], bf[1][j], acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][i], bf[0][j], acc[1][i][j], 0, 0, 0);
                }
            }
        }
    };

    fetch(0);
    stash(0, 0);
    __syncthreads();

    if constexpr (!PIPE) {
        // plain schedule: [fetch next] [all k-steps of this stage] [stash next] barrier
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) fetch((t + 1) * KT);
            const int ks_live = t + 1 < nk ? KS : (K - t * KT + 3) / 4;     // (the last stage's k-steps beyond K multiply zeros)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (kk < ks_live) {
                    double af[NP][TM], bf[NP][TN];
                    read_frags(cur, kk, af, bf);
                    mfma_step(af, bf);
                }
            }
            if (t + 1 < nk) stash(cur ^ 1, (t + 1) * KT);
            __syncthreads();
        }
    } else {
        // rotated schedule: the last k-step of every stage runs AFTER the stage
        // barrier, from fragments read before it, while the first fragments of
        // the next stage and the global loads of the stage after are in flight.
        // The MFMA stream therefore continues across the barrier; a wave only
        // loses issue time to barrier skew.  Fragments are double-buffered in
        // registers (f0/f1), k-step kk+1 is read while kk multiplies.
        double a0[NP][TM], b0[NP][TN], a1[NP][TM], b1[NP][TN];
        if (nk > 1) fetch(KT);
        read_frags(0, 0, a0, b0);
        auto stage = [&](int t, auto do_stash, auto do_fetch, auto do_next) {
            const int cur = t & 1;
            // the last stage's k-steps beyond K multiply zeros: skipped (wave-uniform branches around the MFMAs only; adding
            // 0 . 0 never changed a value, so the results are the same bits)
            constexpr bool last = !decltype(do_stash)::value;
            const int ks_live = last ? (K - t * KT + 3) / 4 : KS;
#pragma unroll
            for (int kk = 0; kk + 1 < KS; ++kk) {
                // the fence keeps the fragment reads of step kk+1 AHEAD of the
                // MFMAs of step kk (the compiler otherwise sinks them behind the
                // MFMA cluster and every k-step pays the LDS latency)
                if ((kk & 1) == 0) { read_frags(cur, kk + 1, a1, b1); __builtin_amdgcn_sched_barrier(0); if (!last || kk < ks_live) mfma_step(a0, b0); }
                else               { read_frags(cur, kk + 1, a0, b0); __builtin_amdgcn_sched_barrier(0); if (!last || kk < ks_live) mfma_step(a1, b1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (decltype(do_stash)::value) stash(cur ^ 1, (t + 1) * KT);
            __syncthreads();
            if constexpr (decltype(do_fetch)::value) fetch((t + 2) * KT);
            // KS is even: the last k-step lives in (a1, b1); (a0, b0) is free again
            if constexpr (decltype(do_next)::value) read_frags(cur ^ 1, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (!last || KS - 1 < ks_live) mfma_step(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        int t = 0;
        for (; t + 2 < nk; ++t) stage(t, T_{}, T_{}, T_{});
        if (t + 1 < nk) { stage(t, T_{}, F_{}, T_{}); ++t; }
        stage(t, F_{}, F_{}, F_{});
    }

    // epilogue: reg r of a lane -> row (lane>>4) + 4r, col lane&15 of each 16x16 block
    const int crow = m0 + wm * 16 * TM + (lane >> 4);
    const int ccol = n0 + wn * 16 * TN + (lane & 15);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = crow + i * 16 + 4 * r;
            if (row < M) {
                double* crow_ptr = C + (int64_t)row * g.ldc * ES;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = ccol + j * 16;
                    if (col < N) {
                        if constexpr (CX) {
                            f64x2* dst = reinterpret_cast<f64x2*>(crow_ptr + 2 * (int64_t)col);
                            f64x2 v = f64x2{acc[0][i][j][r], acc[1][i][j][r]};
                            if (g.accumulate) v += *dst;
                            *dst = v;
                        } else {
                            double v = acc[0][i][j][r];
                            if (g.accumulate) v += crow_ptr[col];
                            crow_ptr[col] = v;
                        }
                    }
                }
            }
        }
    }
}

// Schedule / tile-shape overrides for tuning runs (qs_tuning_set); 0 = automatic.

template <int WM, int WN, int TM, int TN, int KT, int MODE>
static int launch_one(GemmArgs g, int64_t batch, hipStream_t stream) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    constexpr int NP = MODE == MODE_C128 ? 2 : 1;
    g.tiles_m = (int)cdiv(g.m, BM);
    g.tiles_n = (int)cdiv(g.n, BN);
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * batch;
    if (nwg <= 0 || nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    const size_t lds = sizeof(double) * 2 * NP * (BM * (KT + 2) + KT * (BN + 16));
    static PerDeviceLds lds_opt_in[2];   // per instantiation, schedule and device
    // the rotated schedule needs a second fragment set; shapes where that would
    // spill (8-byte staging with 16 accumulators, the 96x96 complex tile, the
    // 1-WG/CU tuning shapes) keep the plain schedule
    constexpr bool pipe_fits = !((MODE == MODE_F64_SCALAR && TM * TN >= 16) ||
                                 (MODE == MODE_C128 && TM * TN >= 9) || WM * WN != 4 || WM != WN);
    const int pipe = (g_tune.gemm_pipe && pipe_fits) ? 1 : 0;
    auto kern = gemm_kernel<WM, WN, TM, TN, KT, MODE, false>;
    if constexpr (pipe_fits) {        // (the rotated form is only instantiated where it can run: the fitted shapes below are many)
        if (pipe) kern = gemm_kernel<WM, WN, TM, TN, KT, MODE, true>;
    }
    if (int rc = opt_in_dynamic_lds((const void*)kern, lds, lds_opt_in[pipe], "hipFuncSetAttribute(gemm)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * WM * WN), lds, stream, g);
    note_dispatch("qs::gemm_kernel<%d, %d, %d, %d, %d, %d, %s>", WM, WN, TM, TN, KT, MODE, pipe ? "true" : "false");
    return launch_status("gemm launch");
}

static bool fill_args(GemmArgs& g, const double* A, const double* B, double* C, int64_t m,
                      int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch,
                      int64_t sa, int64_t sb, int64_t sc, int accumulate) {
    if (m <= 0 || n <= 0 || k <= 0 || batch <= 0) return false;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return false;
    if (lda < k || ldb < n || ldc < n) return false;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sa = sa; g.sb = sb; g.sc = sc;
    g.m = (int)m; g.n = (int)n; g.k = (int)k;
    g.tiles_m = g.tiles_n = 0;
    g.accumulate = accumulate ? 1 : 0;
    // Which operand is the stream that neighbouring tiles should share in L2:
    // a shared (stride-0) A, or a short-and-wide product, streams B.
    g.group_along_m = ((sa == 0 && batch > 1) || m < n) ? 1 : 0;
    return true;
}

// ---------------------------------------------------------------------------
// Tile-shape choice.  All shapes are 4-wave workgroups (two resident per CU)
// except the two 8-wave ones kept for tuning.  The automatic choice minimises
// padded work  tiles_m*BM * tiles_n*BN / weight  over the candidate list; the
// weights are the measured relative rates of the shapes on full tiles
// (MI355X, l=256 / l=128: profiles/r01_tile_sweep.txt).
// ---------------------------------------------------------------------------
struct TileShape { int id, bm, bn; double weight; };

// Tile shape of the smallest estimated time: rounds of the tile list over the resident workgroups (two per CU) times the
// work of a tile over its relative rate.  For a long list that is the padded area, as before; for a short one -- the
// single-particle functions of a 55-orbital dot on a 101 x 101 grid are a 55 x 10201 product: 80 tiles of 64 x 128 --
// it prefers tiles that fill the chip (g_tune.gemm_pick == 0: padded area only, the round-1 rule).
static int pick_shape(const TileShape* cand, int ncand, int64_t m, int64_t n, int64_t batch, double* cost_out = nullptr) {
    int best = cand[0].id;
    double best_cost = 1e300;
    const double slots = 2.0 * device_cu_count();
    for (int i = 0; i < ncand; ++i) {
        const double tiles = (double)cdiv(m, cand[i].bm) * (double)cdiv(n, cand[i].bn) * (double)batch;
        const double area = (double)cand[i].bm * (double)cand[i].bn;
        const double rounds = g_tune.gemm_pick == 0 ? tiles / slots : std::ceil(tiles / slots);
        const double cost = rounds * area / cand[i].weight;
        if (cost < best_cost) { best_cost = cost; best = cand[i].id; }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

// full-tile rate of this kernel relative to the VALU-free one (l = 256: 59.6 against 67 TFLOP/s)
constexpr double kGeneralRelativeRate = 0.89;
// relative rate of the fitted shapes (plain schedule, a quarter-wide wave tile): profiles/r03_mid_size_shapes.txt
constexpr double kFitWeight = 0.80;      // (measured on 8-byte staging, in units of this kernel's 128 x 128 tile on 16-byte staging)
// 8-byte staging (odd extents or strides) against 16-byte staging, both tiled kernels
constexpr double kScalarStagingRate = 0.88;

template <int MODE>
static int dispatch_f64(int cfg, const GemmArgs& g, int64_t batch, hipStream_t s) {
    switch (cfg) {
        case 1: return launch_one<2, 2, 4, 4, 16, MODE>(g, batch, s);    // 128 x 128
        case 2: return launch_one<4, 1, 4, 4, 16, MODE>(g, batch, s);    // 256 x  64 (1 WG/CU)
        case 3: return launch_one<1, 4, 4, 4, 16, MODE>(g, batch, s);    //  64 x 256 (1 WG/CU)
        case 4: return launch_one<1, 4, 2, 4, 16, MODE>(g, batch, s);    //  32 x 256
        case 5: return launch_one<2, 2, 2, 2, 16, MODE>(g, batch, s);    //  64 x  64
        case 6: return launch_one<4, 2, 4, 4, 16, MODE>(g, batch, s);    // 256 x 128, 8 waves
        case 7: return launch_one<2, 4, 4, 4, 16, MODE>(g, batch, s);    // 128 x 256, 8 waves
        case 8: return launch_one<2, 2, 3, 3, 16, MODE>(g, batch, s);    //  96 x  96
        case 9: return launch_one<2, 2, 4, 2, 16, MODE>(g, batch, s);    // 128 x  64
        case 10: return launch_one<2, 2, 2, 4, 16, MODE>(g, batch, s);   //  64 x 128
        case 11: return launch_one<2, 2, 1, 1, 16, MODE>(g, batch, s);   //  32 x  32
        case 12: return launch_one<2, 2, 3, 4, 16, MODE>(g, batch, s);   //  96 x 128
        case 13: return launch_one<2, 2, 4, 3, 16, MODE>(g, batch, s);   // 128 x  96
        default: break;
    }
    // FITTED shapes (round 3): the small extent of a contraction -- the basis size, as m in the c, b, a contractions and as n
    // in d and c -- covered EXACTLY by one tile, to the next multiple of 16: 100 + t = (16 t) x 64 with the four waves side
    // by side along n, 200 + t = 64 x (16 t) with the waves stacked along m; t = 5 ... 16.  16-byte staging needs 16 t to be
    // a multiple of 32: odd t takes the 8-byte form.
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side: two dozen instantiations less to compile
    return QS_ERR_BAD_EXTENT;
#else
    if constexpr (MODE == MODE_F64_VEC2) {
        switch (cfg) {
#define QS_FIT(T) case 100 + T: return launch_one<1, 4, T, 1, 16, MODE>(g, batch, s); case 200 + T: return launch_one<4, 1, 1, T, 16, MODE>(g, batch, s);
            QS_FIT(6) QS_FIT(8) QS_FIT(10) QS_FIT(12) QS_FIT(14) QS_FIT(16)
#undef QS_FIT
            default: return QS_ERR_BAD_EXTENT;
        }
    } else {
        switch (cfg) {
#define QS_FIT(T) case 100 + T: return launch_one<1, 4, T, 1, 16, MODE>(g, batch, s); case 200 + T: return launch_one<4, 1, 1, T, 16, MODE>(g, batch, s);
            QS_FIT(5) QS_FIT(6) QS_FIT(7) QS_FIT(8) QS_FIT(9) QS_FIT(10) QS_FIT(11) QS_FIT(12) QS_FIT(13) QS_FIT(14) QS_FIT(15) QS_FIT(16)
#undef QS_FIT
            default: return QS_ERR_BAD_EXTENT;
        }
    }
#endif
}

int gemm_f64(const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k,
             int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb,
             int64_t sc, int accumulate, hipStream_t stream) {
    GemmArgs g;
    if (!fill_args(g, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate))
        return QS_ERR_BAD_EXTENT;
    // 16-byte loads need even extents/strides and 16-byte aligned bases.
    const bool vec = aligned(A, 16) && aligned(B, 16) && !(lda & 1) && !(ldb & 1) && !(k & 1) &&
                     !(n & 1) && !(sa & 1) && !(sb & 1);
    int cfg = g_tune.gemm_f64_cfg;
    if (cfg == 0) {
        int rc = gemm_skinny_try(QS_F64, A, B, C, m, n, k, lda, ldb, ldc, batch, accumulate, stream);
        if (rc != 1) return rc;
        rc = gemm_stream_try(QS_F64, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate, stream);
        if (rc != 1) return rc;
        static const TileShape cand[] = {
            {1, 128, 128, 1.00}, {12, 96, 128, 0.98}, {13, 128, 96, 0.98}, {8, 96, 96, 0.96},
            {5, 64, 64, 0.95},   {9, 128, 64, 0.90},  {10, 64, 128, 0.90}, {11, 32, 32, 0.73},
        };
        double cost = 0.0;
        cfg = pick_shape(cand, sizeof(cand) / sizeof(cand[0]), m, n, batch, &cost);
        // Fitted shapes (see dispatch_f64): a basis size between 65 and 256 as m or as n.  Measured (same-box sweep with the
        // shapes forced, profiles/r03_mid_size_shapes.txt): on their plain schedule they reach ~0.7 of the VALU-free kernel's
        // rate -- they pay where everything is on 8-byte staging anyway (odd basis sizes) and the tile they save is large:
        // l = 65 +14 %, 97 +12 %, 105 +11 %, 129 +4 %; with 16-byte staging available the other kernels win.
        int fit_cfg = 0;
        if (!vec) cost /= kScalarStagingRate;
        if (g_tune.gemm_fit == 2 || (g_tune.gemm_fit == 1 && !vec)) {
            const int tm = (int)cdiv(m, 16), tn = (int)cdiv(n, 16);
            TileShape fit[2];
            int nfit = 0;
            // (from 13 tiles up the stages of a tile take more than half of a CU's LDS: one workgroup per CU)
            auto weight = [](int t) { return t <= 9 ? 1.10 * kFitWeight : t <= 12 ? kFitWeight : 0.75 * kFitWeight; };
            if (m <= 256 && tm >= 5) fit[nfit++] = TileShape{100 + tm, 16 * tm, 64, weight(tm)};
            if (n <= 256 && tn >= 5) fit[nfit++] = TileShape{200 + tn, 64, 16 * tn, weight(tn)};
            if (nfit) {
                double fcost = 0.0;
                const int fc = pick_shape(fit, nfit, m, n, batch, &fcost);
                if (g_tune.gemm_fit == 2 || fcost < cost) { fit_cfg = fc; cost = fcost; }
            }
        }
        if (g_tune.gemm_strip == 2 || g_tune.gemm_fast == 1) {
            // strip kernels (qs_gemm_strip.hip): the small extent of the product covered by one tile to the next multiple of 16
            // (not when a tuning key forces one of the other tiled kernels)
            const bool even = vec && aligned(C, 16) && !(ldc & 1) && !(sc & 1);
            const double fast = gemm_fast_estimate(QS_F64, m, n, k, batch, even);
            const double gen = cost / kGeneralRelativeRate;
            rc = gemm_strip_try(QS_F64, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate, fast < gen ? fast : gen, stream);
            if (rc != 1) return rc;
        }
        rc = gemm_fast_try(QS_F64, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc,
                           accumulate, g.group_along_m, g_tune.gemm_fit == 2 && fit_cfg ? 0.0 : cost / kGeneralRelativeRate, stream);
        if (rc != 1) return rc;
        if (fit_cfg) {
            // 16-byte staging only for an even tile count; an odd one takes the 8-byte form whatever the alignment
            const bool odd = (fit_cfg % 100) & 1;
            return (vec && !odd) ? dispatch_f64<MODE_F64_VEC2>(fit_cfg, g, batch, stream)
                                 : dispatch_f64<MODE_F64_SCALAR>(fit_cfg, g, batch, stream);
        }
    }
    return vec ? dispatch_f64<MODE_F64_VEC2>(cfg, g, batch, stream)
               : dispatch_f64<MODE_F64_SCALAR>(cfg, g, batch, stream);
}

int gemm_c128(const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k,
              int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb,
              int64_t sc, int accumulate, hipStream_t stream) {
    GemmArgs g;
    if (!fill_args(g, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate))
        return QS_ERR_BAD_EXTENT;
    int cfg = g_tune.gemm_c128_cfg;
    if (cfg == 0) {
        int rc = gemm_skinny_try(QS_C128, A, B, C, m, n, k, lda, ldb, ldc, batch, accumulate, stream);
        if (rc != 1) return rc;
        rc = gemm_stream_try(QS_C128, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate, stream);
        if (rc != 1) return rc;
        static const TileShape cand[] = {
            {1, 64, 128, 1.00}, {2, 128, 64, 1.00}, {6, 64, 64, 1.00}, {9, 96, 96, 0.97},
            {7, 96, 64, 0.95},  {8, 64, 96, 0.95},  {4, 32, 32, 0.94},
        };
        double cost = 0.0;
        cfg = pick_shape(cand, sizeof(cand) / sizeof(cand[0]), m, n, batch, &cost);
        if (g_tune.gemm_strip == 2 || g_tune.gemm_fast == 1) {
            // strip kernels (qs_gemm_strip.hip), complex form: a basis of up to 128 orbitals covered by one tile to the next multiple of 16
            const double fast = gemm_fast_estimate(QS_C128, m, n, k, batch, true);
            rc = gemm_strip_try(QS_C128, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, accumulate, fast < cost ? fast : cost, stream);
            if (rc != 1) return rc;
        }
        rc = gemm_fast_try(QS_C128, A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc,
                           accumulate, g.group_along_m, 0.0, stream);
        if (rc != 1) return rc;
    }
    switch (cfg) {
        case 1: return launch_one<2, 2, 2, 4, 8, MODE_C128>(g, batch, stream);   //  64 x 128
        case 2: return launch_one<2, 2, 4, 2, 8, MODE_C128>(g, batch, stream);   // 128 x  64
        case 3: return launch_one<4, 1, 2, 4, 8, MODE_C128>(g, batch, stream);   // 128 x  64 (tall waves)
        case 4: return launch_one<2, 2, 1, 1, 8, MODE_C128>(g, batch, stream);   //  32 x  32
        case 5: return launch_one<1, 4, 2, 4, 8, MODE_C128>(g, batch, stream);   //  32 x 256
        case 6: return launch_one<2, 2, 2, 2, 8, MODE_C128>(g, batch, stream);   //  64 x  64
        case 7: return launch_one<2, 2, 3, 2, 8, MODE_C128>(g, batch, stream);   //  96 x  64
        case 8: return launch_one<2, 2, 2, 3, 8, MODE_C128>(g, batch, stream);   //  64 x  96
        case 9: return launch_one<2, 2, 3, 3, 8, MODE_C128>(g, batch, stream);   //  96 x  96
        default: return QS_ERR_BAD_EXTENT;
    }
}

}  // namespace qs
