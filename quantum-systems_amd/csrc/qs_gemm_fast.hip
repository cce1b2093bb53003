// Fast path of the batched row-major product (see qs_gemm.hip for the general
// kernel and the contraction -> GEMM map).
//
// Why a second kernel: on gfx950 the fp64 MFMA shares the SIMD's vector issue
// with ordinary VALU work -- every VALU instruction in the K loop takes its
// issue cycles away from the matrix pipe (measured with tools/probe_mix.hip:
// 32 integer adds per 16 MFMAs cost 12 % of the MFMA rate, and two co-resident
// waves do not hide it).  The general kernel spends ~130 VALU instructions per
// K step on 64-bit address arithmetic and edge handling.  The K loop of this
// kernel contains no VALU work at all:
//   * global loads use the scalar-base form (SGPR buffer descriptor + one
//     loop-invariant 32-bit lane offset); the bases advance on the scalar ALU;
//   * LDS addresses are one VGPR + immediates (the loop is unrolled by two so
//     the stage buffer is a compile-time constant);
//   * no per-lane bounds checks.
// Two forms (template parameter EDGE):
//   * exact: every extent a whole multiple of the tile (l = 128, 256, 512, ...);
//   * edge: any extent.  Memory safety comes from the buffer descriptors
//     (num_records = bytes left in the operand, kept on the scalar ALU; the
//     hardware returns 0 for dwords past it), exactness from zeroing the
//     k >= K part of the LAST k-stage in registers before it is written to
//     LDS (selects, so stray non-finite values cannot leak), and the epilogue
//     of a border tile stores under a per-lane predicate.  Rows >= m and
//     columns >= n of a border tile multiply whatever the loads returned and
//     are never stored.
// VEC = 16-byte global accesses (even extents and strides, 16-byte aligned
// bases), otherwise 8-byte ones (odd l, e.g. the 55 orbitals of config 2).
//
// Persistent workgroups: 2 per CU, each walks its share of the tile list
// (virtual block ids bid, bid + P, bid + 2P, ... of the XCD-chunked order the
// general kernel uses, so concurrently running workgroups still share operand
// panels in their XCD's L2).  The global stage stream is flat across tiles:
// the loads for the first stages of tile n+1 are issued during the last stages
// of tile n and its first fragments are read before tile n's epilogue, so the
// MFMA stream only pauses for the issue of the epilogue stores.  (With one
// workgroup per tile the prologue + epilogue + launch gap cost 8 % at K = 256.)
//
// Schedule inside a stage (rotated): fragments are double-buffered in
// registers, k-step kk+1 is read while kk multiplies; the last k-step of a
// stage runs after the stage barrier while the next stage's first fragments and
// the global loads of the stage after are in flight.

#include <type_traits>

#include <cmath>

#include "qs_common.h"
#include "qs_fast_items.h"

// Epilogue stores are non-temporal: the product's output is read again only by the NEXT contraction, after the
// whole tensor has passed through the caches (same-box A/B, three alternating runs each: l = 256 66.9-67.2 ->
// 67.1-67.2 TFLOP/s, l = 128 57.7-58.1 -> 58.2-58.7; -DQS_FAST_NT=0 builds the plain-store form).
#ifndef QS_FAST_NT
#define QS_FAST_NT 1
#endif
#if QS_FAST_NT
#define QS_FAST_STORE(dst, v) __builtin_nontemporal_store((v), (dst))
#else
#define QS_FAST_STORE(dst, v) (*(dst) = (v))
#endif

namespace qs {

struct FastArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int64_t sa, sb, sc;      // elements
    uint64_t a_end, b_end;   // address one past the last byte of each operand (edge form)
    int64_t m;               // extents (edge form: epilogue predicate, k-tail)
    int n, k;
    int nk;                  // ceil(K / KT)
    int tiles_m, tiles_n;
    unsigned total;          // tiles_m * tiles_n * batch = size of the virtual grid
    int group_along_m;
    int accumulate;
};

template <bool CX, int TM, int TN, bool VEC, bool EDGE>
__global__ __launch_bounds__(256, 2)
void gemm_fast_kernel(const FastArgs g) {
    static_assert(!CX || VEC, "a complex element is one 16-byte item");
    constexpr int NP = CX ? 2 : 1;
    constexpr int ES = CX ? 2 : 1;
    constexpr int KT = CX ? 8 : 16;
    constexpr int KS = KT / 4;
    constexpr int NT = 256;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    // B rows: complex fragments are ds_read_b64 (row stride = 16 mod 32 doubles is conflict-free);
    // fp64 fragments are ds_read_b128 column PAIRS (row stride = 0 mod 32 is conflict-free)
    // A rows: fp64 pads the row to KT + 2 (stride 18: conflict-free fragment reads, tolerable writes).  complex128 (round 4): no
    // padding, the k index of a row XOR-ed with 2 ((row >> 2) & 3) instead -- a half wave of the stage WRITE (four rows x eight
    // k) then covers 32 different bank pairs, and so does a half wave of the fragment READ (sixteen rows x two k): with the
    // padded rows of rounds 1-3 (stride 10) writes of the A stage collided.  Measured (profiles/r04_pmc_c128.txt, same-box A/B
    // r04_c128_swizzle_ab.txt): SQ_LDS_BANK_CONFLICT 40 -> 31 % and 25 -> 18 % of the LDS-active cycles of the two tile forms,
    // l = 256 66.9 -> 67.5, l = 128 64.2 -> 64.6 TFLOP/s.  What is left scales with the number of A-stage writes (8 conflict
    // cycles per write instruction in both forms): the lanes a 64-bit LDS write serves together are not the half waves assumed
    // here -- open.
#ifdef QS_FAST_CX_PAD      // (A/B builds: the padded complex rows of rounds 1-3)
    constexpr bool kSwizzleA = false;
#else
    constexpr bool kSwizzleA = CX;
#endif
    constexpr int SA = kSwizzleA ? KT : KT + 2, SB = CX ? BN + 16 : BN;      // (B rows 8 mod 16 instead of 16 mod 32: measured, level)
    constexpr int EPI = (!CX && VEC) ? 2 : 1;           // tensor elements per global item
    constexpr int DPI = CX ? 1 : EPI;                   // doubles per item inside one LDS plane
    constexpr int IPR_A = KT / EPI;                     // items per A row of a stage
    constexpr int RA = NT / IPR_A;                      // A rows covered by one item step
    constexpr int NA = BM / RA;                         // items per thread, A stage
    constexpr int IPR_B = BN / EPI;                     // items per B row
    static_assert(NT % IPR_B == 0 && BM % RA == 0, "item steps cover whole rows");
    constexpr int RPS = NT / IPR_B;                     // B rows covered by one item step
    static_assert(KT % RPS == 0, "B item steps tile the stage");
    constexpr int NB = KT / RPS;
    constexpr int A_STAGE = NP * BM * SA, B_STAGE = NP * KT * SB;
    constexpr size_t ESZ = 8 * ES;
    constexpr unsigned IB = (unsigned)(EPI * ESZ);      // bytes per item
    static_assert(KS % 2 == 0, "fragment double buffering needs an even k-step count");
    static_assert(CX || TN % 2 == 0, "fp64 n-tiles come in column pairs");
    using Item = FastItem<CX || VEC>;
    using item_t = typename Item::type;

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = g.nk;
    const unsigned P = gridDim.x;                       // persistent grid (multiple of 8 unless P == total)
    // edge form: valid k-steps of a tile's last stage (KT when K is a whole number of stages)
    const int k_tail = EDGE ? g.k - (nk - 1) * KT : KT;

    // tile coordinates of virtual block v
    auto decode = [&](unsigned v, int& m0, int& n0, int64_t& b) {
        unsigned w = xcd_chunked_index_fast(v, g.total);
        int mt, nt;
        if (g.group_along_m) {
            mt = w % g.tiles_m; w /= g.tiles_m;
            nt = w % g.tiles_n; w /= g.tiles_n;
        } else {
            nt = w % g.tiles_n; w /= g.tiles_n;
            mt = w % g.tiles_m; w /= g.tiles_m;
        }
        // integer division runs on the vector ALU; bring the (wave-uniform) results back to
        // scalar registers so that every pointer derived from them stays an SGPR base
        b = __builtin_amdgcn_readfirstlane((int)w);
        m0 = __builtin_amdgcn_readfirstlane(mt) * BM;
        n0 = __builtin_amdgcn_readfirstlane(nt) * BN;
    };

    // ---- fetch cursor: scalar bases of the tile being loaded + loop-invariant lane offsets
    // The bases are kept as scalar 64-bit integers and the loads go through buffer
    // descriptors built from them: a descriptor lives in SGPRs by construction, which pins
    // the scalar-base addressing even though the bases are re-aimed at every tile change.
    uint64_t a_ptr[NA];
    uint64_t b_ptr[NB];
    unsigned f_v = blockIdx.x;   // virtual block the cursor is in
    int f_k = 0;                 // next k-stage to load in that tile
    bool f_valid = true;
    auto aim = [&](unsigned v) {
        int m0, n0; int64_t b;
        decode(v, m0, n0, b);
        const char* Ab = reinterpret_cast<const char*>(g.A + (b * g.sa + (int64_t)m0 * g.lda) * ES);
        const char* Bb = reinterpret_cast<const char*>(g.B + (b * g.sb + n0) * ES);
#pragma unroll
        for (int i = 0; i < NA; ++i) a_ptr[i] = uniform64(reinterpret_cast<uint64_t>(Ab + (size_t)i * RA * g.lda * ESZ));
        // B rows: the wave's first row of an item step is part of the scalar base (no limit on
        // ldb); only when a wave-instruction spans several rows (IPR_B < 64) does the lane
        // offset carry a row term
        const int brow0 = (wave * 64) / IPR_B;
#pragma unroll
        for (int i = 0; i < NB; ++i) b_ptr[i] = uniform64(reinterpret_cast<uint64_t>(Bb + (size_t)(brow0 + i * RPS) * g.ldb * ESZ));
    };
    aim(f_v);
    const unsigned voff_a = (unsigned)(tid / IPR_A) * (unsigned)g.lda * (unsigned)ESZ + (unsigned)(tid % IPR_A) * IB;
    const unsigned voff_b = (IPR_B < 64 ? (unsigned)(lane / IPR_B) * (unsigned)g.ldb * (unsigned)ESZ : 0u) +
                            (unsigned)(tid % IPR_B) * IB;
    const size_t a_step = KT * ESZ;
    const size_t b_step = (size_t)KT * g.ldb * ESZ;

    // ---- LDS addressing: one base per operand and direction, the rest immediates
    double* st_a = kSwizzleA ? As + (tid / IPR_A) * SA + ((tid % IPR_A) ^ (2 * (((tid / IPR_A) >> 2) & 3)))
                             : As + (tid / IPR_A) * SA + (tid % IPR_A) * DPI;
    double* st_b = Bs + (tid / IPR_B) * SB + (tid % IPR_B) * DPI;
    const double* rd_a = As + (wm * 16 * TM + (lane & 15)) * SA + (lane >> 4);
    // complex: the swizzled position of k = 4 kk + (lane >> 4) depends on kk -- one base per k-step of a stage (KS = 2)
    const int swz_r = kSwizzleA ? 2 * (((lane & 15) >> 2) & 3) : 0;
    const double* rd_a_cx[2] = {As + (wm * 16 * TM + (lane & 15)) * SA + ((lane >> 4) ^ swz_r),
                                As + (wm * 16 * TM + (lane & 15)) * SA + ((4 + (lane >> 4)) ^ swz_r)};
    // fp64: lane c of n-tile pair (2jp, 2jp+1) owns the ADJACENT columns 32jp + 2c, 32jp + 2c + 1,
    // so one 16-byte LDS read feeds two MFMA tiles and the epilogue stores 16 bytes per lane
    const double* rd_b = Bs + (lane >> 4) * SB + wn * 16 * TN + (CX ? 1 : 2) * (lane & 15);

    // two staging register sets: the data of global stage s waits in set s & 1, so a load has
    // two stages (not one) to arrive from HBM before it is written to LDS
    item_t ra[2][NA], rb[2][NB];

    // load the cursor's stage into registers and advance the cursor (to the
    // next tile of this workgroup after the last k-stage)
    auto fetch = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[set][i] = Item::load(a_ptr[i], EDGE ? bytes_left(g.a_end, a_ptr[i]) : 0xFFFFFFFFu, voff_a);
            a_ptr[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[set][i] = Item::load(b_ptr[i], EDGE ? bytes_left(g.b_end, b_ptr[i]) : 0xFFFFFFFFu, voff_b);
            b_ptr[i] += b_step;
        }
        if (++f_k == nk) {
            f_k = 0;
            f_v += P;
            f_valid = f_v < g.total;
            if (f_valid) aim(f_v);
        }
    };

    int s_k = 0;   // k-stage (inside its tile) of the next stage to be written to LDS
    auto stash = [&](auto buf_c) {   // stage data of parity `buf` -> LDS stage `buf`
        constexpr int buf = decltype(buf_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EDGE) {
            // last k-stage of a tile with a K tail: zero the k >= K part of both operands
            const bool tail = (s_k == nk - 1) && (k_tail < KT);
            if (++s_k == nk) s_k = 0;
            if (tail) {
                const item_t zero = item_t(0.0);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    if constexpr (EPI == 2) {
                        if ((tid % IPR_A) * 2 >= k_tail) ra[buf][i][0] = 0.0;
                        if ((tid % IPR_A) * 2 + 1 >= k_tail) ra[buf][i][1] = 0.0;
                    } else {
                        if ((tid % IPR_A) >= k_tail) ra[buf][i] = zero;
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (tid / IPR_B + i * RPS >= k_tail) rb[buf][i] = zero;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            double* d = st_a + buf * A_STAGE + i * RA * SA;
            if constexpr (CX) { d[0] = ra[buf][i][0]; d[BM * SA] = ra[buf][i][1]; }
            else if constexpr (VEC) *reinterpret_cast<f64x2*>(d) = ra[buf][i];
            else d[0] = ra[buf][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double* d = st_b + buf * B_STAGE + i * RPS * SB;
            if constexpr (CX) { d[0] = rb[buf][i][0]; d[KT * SB] = rb[buf][i][1]; }
            else if constexpr (VEC) *reinterpret_cast<f64x2*>(d) = rb[buf][i];
            else d[0] = rb[buf][i];
        }
    };

    f64x4 acc[NP][TM][TN];

    auto read_frags = [&](auto buf_c, int kk, double (&af)[NP][TM], double (&bf)[NP][TN]) {
        constexpr int buf = decltype(buf_c)::value;
        const double* as = (CX ? rd_a_cx[kk & 1] - kk * 4 : rd_a) + buf * A_STAGE;
        const double* bs = rd_b + buf * B_STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[p][i] = as[p * BM * SA + i * 16 * SA + kk * 4];
            if constexpr (CX) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[p][j] = bs[p * KT * SB + kk * 4 * SB + j * 16];
            } else {
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    const f64x2 v = *reinterpret_cast<const f64x2*>(bs + kk * 4 * SB + jp * 32);
                    bf[p][2 * jp] = v[0];
                    bf[p][2 * jp + 1] = v[1];
                }
            }
        }
    };
    // `fresh`: first k-step of a tile -- the accumulators start from C = 0
    auto mfma_step = [&](const double (&af)[NP][TM], const double (&bf)[NP][TN], auto fresh_c) {
        constexpr bool fresh = decltype(fresh_c)::value;
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!CX) {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[0][j], fresh ? zero : acc[0][i][j], 0, 0, 0);
                } else {
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[0][j], fresh ? zero : acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[1][j], fresh ? zero : acc[1][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-af[1][i], bf[1][j], acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][i], bf[0][j], acc[1][i][j], 0, 0, 0);
                }
            }
        }
    };

    // epilogue of the tile at virtual block v: reg r of a lane -> row (lane>>4) + 4r of
    // each 16-row block; 16 bytes per lane (8-byte pieces in the !VEC form).  `guard`:
    // border tile of the edge form, rows >= m and columns >= n are not stored.
    auto epilogue = [&](unsigned v) __attribute__((always_inline)) {
        int m0, n0; int64_t b;
        decode(v, m0, n0, b);
        const int64_t row_l = (int64_t)m0 + wm * 16 * TM + (lane >> 4);
        const int col_l = n0 + wn * 16 * TN + (CX ? 1 : 2) * (lane & 15);
        double* __restrict__ C = g.C + (b * g.sc + row_l * g.ldc + col_l) * ES;
        auto store_all = [&](auto add_c, auto guard_c) __attribute__((always_inline)) {
            constexpr bool add = decltype(add_c)::value;
            constexpr bool guard = decltype(guard_c)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* crow = C + (int64_t)(i * 16 + 4 * r) * g.ldc * ES;
                    const bool row_ok = !guard || row_l + i * 16 + 4 * r < g.m;
                    if constexpr (CX) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (row_ok && (!guard || col_l + j * 16 < g.n)) {
                                f64x2* dst = reinterpret_cast<f64x2*>(crow + 2 * j * 16);
                                f64x2 v2 = f64x2{acc[0][i][j][r], acc[1][i][j][r]};
                                if constexpr (add) v2 += *dst;
                                QS_FAST_STORE(dst, v2);
                            }
                        }
                    } else if constexpr (VEC) {
#pragma unroll
                        for (int jp = 0; jp < TN / 2; ++jp) {
                            if (row_ok && (!guard || col_l + jp * 32 + 1 < g.n)) {
                                f64x2* dst = reinterpret_cast<f64x2*>(crow + jp * 32);
                                f64x2 v2 = f64x2{acc[0][i][2 * jp][r], acc[0][i][2 * jp + 1][r]};
                                if constexpr (add) v2 += *dst;
                                QS_FAST_STORE(dst, v2);
                            } else if (guard && row_ok && col_l + jp * 32 < g.n) {   // odd n: the last column alone
                                double* dst = crow + jp * 32;
                                double v1 = acc[0][i][2 * jp][r];
                                if constexpr (add) v1 += *dst;
                                *dst = v1;
                            }
                        }
                    } else {
#pragma unroll
                        for (int jp = 0; jp < TN / 2; ++jp) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (row_ok && (!guard || col_l + jp * 32 + h < g.n)) {
                                    double* dst = crow + jp * 32 + h;
                                    double v1 = acc[0][i][2 * jp + h][r];
                                    if constexpr (add) v1 += *dst;
                                    *dst = v1;
                                }
                            }
                        }
                    }
                }
            }
        };
        const bool border = EDGE && ((int64_t)m0 + BM > g.m || n0 + BN > g.n);   // wave-uniform
        if (border) {
            if constexpr (EDGE) {
                if (g.accumulate) store_all(std::true_type{}, std::true_type{});
                else store_all(std::false_type{}, std::true_type{});
            }
        } else {
            if (g.accumulate) store_all(std::true_type{}, std::false_type{});
            else store_all(std::false_type{}, std::false_type{});
        }
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;

    // number of tiles of this workgroup and of global stages
    const unsigned my_tiles = (g.total - blockIdx.x + P - 1) / P;
    const int64_t stages = (int64_t)my_tiles * nk;

    fetch(B0{});                       // global stage 0
    stash(B0{});
    __syncthreads();
    double a0[NP][TM], b0[NP][TN], a1[NP][TM], b1[NP][TN];
    if (f_valid) fetch(B1{});          // stage 1
    if (f_valid) fetch(B0{});          // stage 2
    read_frags(B0{}, 0, a0, b0);

    unsigned c_v = blockIdx.x;   // virtual block being computed
    int c_k = 0;                 // its current k-stage

    // one global stage: k-steps 0 .. KS-2, [stash the next stage], barrier, [fetch the
    // stage after next], [first fragments of the next stage], k-step KS-1, and at a tile's
    // last k-stage its epilogue.  All conditions are wave-uniform (scalar branches).
    auto stage = [&](auto cur_c, int64_t gs) {
        constexpr int cur = decltype(cur_c)::value;
        using NXT = std::integral_constant<int, cur ^ 1>;
        const bool has_next = gs + 1 < stages;
        // Edge form, last k-stage of a tile: its k-steps beyond K multiply zeros -- they are skipped (wave-uniform branches
        // around the MFMAs only: K = 130 walks 33 k-steps instead of 36, K = 66 17 instead of 20).  Adding 0 . 0 never
        // changed a value, so results stay bit-identical to the kernels that do not skip.
        const int ks_live = (EDGE && c_k == nk - 1) ? (k_tail + 3) / 4 : KS;
        // k-step 0 (fresh accumulators at the start of a tile)
        read_frags(cur_c, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (c_k == 0) mfma_step(a0, b0, T_{}); else mfma_step(a0, b0, F_{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 1; kk + 1 < KS; ++kk) {
            if ((kk & 1) == 0) { read_frags(cur_c, kk + 1, a1, b1); __builtin_amdgcn_sched_barrier(0); if (!EDGE || kk < ks_live) mfma_step(a0, b0, F_{}); }
            else               { read_frags(cur_c, kk + 1, a0, b0); __builtin_amdgcn_sched_barrier(0); if (!EDGE || kk < ks_live) mfma_step(a1, b1, F_{}); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) stash(NXT{});        // stage gs+1, loaded two stages ago
        __syncthreads();
        if (f_valid) fetch(NXT{});         // stage gs+3 into the set just written out
        if (has_next) read_frags(NXT{}, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (!EDGE || KS - 1 < ks_live) mfma_step(a1, b1, F_{});
        __builtin_amdgcn_sched_barrier(0);
        if (++c_k == nk) {
            epilogue(c_v);
            c_k = 0;
            c_v += P;
        }
    };

    for (int64_t gs = 0; gs < stages; gs += 2) {
        stage(B0{}, gs);
        if (gs + 1 < stages) stage(B1{}, gs + 1);
    }
}


template <bool CX, int TM, int TN, bool VEC, bool EDGE>
static int launch_fast(const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb,
                       int64_t sc, int accumulate, int group_along_m, hipStream_t stream) {
    constexpr int KT = CX ? 8 : 16;
    constexpr int NP = CX ? 2 : 1;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int64_t ESZ = CX ? 16 : 8;
    constexpr int IPR_B = BN / ((!CX && VEC) ? 2 : 1);
    // a wave-instruction of the B loads spans 64 / IPR_B rows through its 32-bit lane offset
    if (IPR_B < 64 && (64 / IPR_B) * ldb * ESZ + 4096 >= (int64_t(1) << 32)) return 1;
    FastArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sa = sa; g.sb = sb; g.sc = sc;
    g.a_end = reinterpret_cast<uint64_t>(A) + (uint64_t)(((batch - 1) * sa + (m - 1) * lda + k) * ESZ);
    g.b_end = reinterpret_cast<uint64_t>(B) + (uint64_t)(((batch - 1) * sb + (k - 1) * ldb + n) * ESZ);
    g.m = m; g.n = (int)n; g.k = (int)k;
    g.nk = (int)cdiv(k, KT);
    g.tiles_m = (int)cdiv(m, BM);
    g.tiles_n = (int)cdiv(n, BN);
    g.group_along_m = group_along_m;
    g.accumulate = accumulate ? 1 : 0;
    const int64_t total = (int64_t)g.tiles_m * g.tiles_n * batch;
    if (total <= 0 || total >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    g.total = (unsigned)total;
    // two persistent workgroups per CU (the LDS and register budget admits exactly two)
    const int n_cu = device_cu_count();
    int64_t P = 2 * (int64_t)n_cu;
    P -= P % 8;
    // How many tiles a workgroup walks (cross-tile prefetch hides the next tile's first loads
    // behind the current tile's last stages and epilogue):
    //   short tile lists (<= 8 per resident workgroup): fully persistent grid, +7 % at l = 64;
    //   long lists: 4 tiles per workgroup -- the dispatcher keeps balancing the grid, which a
    //   static split of a long list loses more on than the prefetch wins (-4 % at l = 256 when
    //   fully persistent, +1.5 % with 4 tiles; profiles/r01_gemm_notes.txt).
    // g_tune.gemm_fast_persist: 0 one tile per workgroup, 1 this policy, 2 always fully
    // persistent, >= 3 that many tiles per workgroup.
    int64_t tiles_per_wg = 1;
    bool full = false;
    if (g_tune.gemm_fast_persist == 1) { full = total <= 8 * P; tiles_per_wg = 4; }
    else if (g_tune.gemm_fast_persist == 2) full = true;
    else if (g_tune.gemm_fast_persist >= 3) tiles_per_wg = g_tune.gemm_fast_persist;
    if (!full) {
        P = (total + tiles_per_wg - 1) / tiles_per_wg;
        P = (P + 7) & ~int64_t(7);
    }
    if (P > total) P = total;
#if defined(QS_FAST_CX_PAD)
    const size_t lds = sizeof(double) * 2 * NP * (BM * (KT + 2) + KT * (CX ? BN + 16 : BN));
#else
    const size_t lds = sizeof(double) * 2 * NP * (BM * (CX ? KT : KT + 2) + KT * (CX ? BN + 16 : BN));
#endif
    auto kern = gemm_fast_kernel<CX, TM, TN, VEC, EDGE>;
    static PerDeviceLds lds_opt_in;   // per instantiation and per device
    if (int rc = opt_in_dynamic_lds((const void*)kern, lds, lds_opt_in, "hipFuncSetAttribute(gemm_fast)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(256), lds, stream, g);
    note_dispatch("qs::gemm_fast_kernel<%s, %d, %d, %s, %s>", CX ? "true" : "false", TM, TN, VEC ? "true" : "false",
                  EDGE ? "true" : "false");
    return launch_status("gemm_fast launch");
}

namespace {
struct FastShape { int id, tm, tn; double weight; };
// Edge-form shapes: BM = 32 tm rows (any tm: two waves of 16 tm rows each), BN = 32 tn columns with tn a power of two (the
// B stage is handed out row-wise to the 256 threads).  Round 3 added the tall-and-narrow shapes 5-9: in the c, b and a
// contractions the NEW basis size M is the m extent of the product, and 128-row tiles waste up to half of the matrix pipe
// on bases just above a multiple of 128 (l = 130: 256 rows for 130; with tm = 5: 160); the 2-D oscillator shells (66, 78,
// 91, 105, 120, 136, 153, 171, 190, 210 ...) are such sizes.  Weights: relative full-tile rates (shapes 1-4 measured in
// round 1, profiles/r01_gemm_notes.txt; 5-9 from the round-3 sweep, profiles/r03_mid_size_shapes.txt).
const FastShape kF64Shapes[] = {{1, 4, 4, 1.00}, {2, 2, 4, 0.93}, {3, 4, 2, 0.93}, {4, 2, 2, 0.85}, {5, 3, 4, 0.91},
                                {6, 5, 2, 0.92}, {7, 6, 2, 0.89}, {8, 7, 2, 0.92}, {9, 3, 2, 0.87}};
const FastShape kC128Shapes[] = {{1, 4, 2, 1.00}, {2, 2, 4, 1.00}, {3, 2, 2, 0.95}};

// Shape of the smallest estimated time -- rounds of the tile list over the resident workgroups (two per CU) times the
// work of a tile over its relative rate (the rule of the general kernel, qs_gemm.hip pick_shape) -- and that estimate.
template <size_t N>
int pick_fast_shape(const FastShape (&cand)[N], int64_t m, int64_t n, int64_t batch, double* cost_out) {
    const double slots = 2.0 * device_cu_count();
    int best = cand[0].id;
    double best_cost = 1e300;
    for (const FastShape& c : cand) {
        if (g_tune.gemm_fast_shape >= 1 && g_tune.gemm_fast_shape <= (int)N && c.id != g_tune.gemm_fast_shape) continue;
        const double tiles = (double)cdiv(m, 32 * c.tm) * (double)cdiv(n, 32 * c.tn) * (double)batch;
        const double rounds = tiles > 8 * slots ? tiles / slots : ceil(tiles / slots);
        const double cost = rounds * (32.0 * c.tm) * (32.0 * c.tn) / c.weight;
        if (cost < best_cost) { best_cost = cost; best = c.id; }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}
}  // namespace

double gemm_fast_estimate(int dtype, int64_t m, int64_t n, int64_t k, int64_t batch, bool even) {
    if (!g_tune.gemm_fast) return 1e300;
    const bool cx = dtype == QS_C128;
    const double slots = 2.0 * device_cu_count();
    auto whole = [&](int bm, int bn, double w) {
        const double tiles = (double)(m / bm) * (double)(n / bn) * (double)batch;
        const double rounds = tiles > 8 * slots ? tiles / slots : ceil(tiles / slots);
        return rounds * bm * bn / w;
    };
    // (the exact form against the edge form of the same shape: no guards, no range arithmetic: l = 128 58.4 TFLOP/s against the
    // ~51 the edge form's l = 120 scales to)
    if (!cx && even && k % 16 == 0) {
        if (m % 128 == 0 && n % 128 == 0) return whole(128, 128, 1.12);
        if (m % 64 == 0 && n % 128 == 0) return whole(64, 128, 1.04);
    }
    if (cx) {      // (units of the complex tile shapes: the 64 x 128 tile = 1)
        if (k % 8 == 0 && ((m % 128 == 0 && n % 64 == 0) || (m % 64 == 0 && n % 128 == 0))) return whole(m % 128 == 0 ? 128 : 64, m % 128 == 0 ? 64 : 128, 1.10);
        if (k % 8 == 0 && m % 64 == 0 && n % 64 == 0) return whole(64, 64, 1.04);
        return 1e300;
    }
    double cost = 1e300;
    pick_fast_shape(kF64Shapes, m, n, batch, &cost);
    if (!even && !g_tune.gemm_fast_unaligned) cost /= 0.85;
    return cost;
}

// Returns QS_OK after launching, or 1 when the product does not qualify
// (caller falls back to the general kernel).
int gemm_fast_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                  int64_t sb, int64_t sc, int accumulate, int group_along_m, double general_cost, hipStream_t stream) {
    if (!g_tune.gemm_fast) return 1;
    const bool cx = dtype == QS_C128;
    const int64_t esz = cx ? 16 : 8;
    if (cx && (!aligned(A, 16) || !aligned(B, 16) || !aligned(C, 16))) return 1;
    // the lane offsets of the loads are 32-bit: one item step of rows must fit
    if (32 * lda * esz + 256 >= (int64_t(1) << 32)) return 1;
    if (m >= (int64_t(1) << 31) - 256 || n >= (int64_t(1) << 31) - 256 || k >= (int64_t(1) << 31) - 256) return 1;
    // 16-byte accesses: aligned bases, even strides (complex elements are 16 bytes by themselves)
    // gfx950 carries out 16-byte buffer loads and global stores at ANY 8-byte-aligned address (tools/probe_unaligned.hip), so odd
    // strides and extents -- every odd basis size -- stage with 16-byte items too (g_tune.gemm_fast_unaligned = 0: 8-byte items
    // there, the rule of rounds 1-3); LDS accesses stay 16-byte aligned (the stage layout does not depend on the strides).
    const bool even = aligned(A, 16) && aligned(B, 16) && aligned(C, 16) && !(lda & 1) && !(ldb & 1) &&
                      !(ldc & 1) && !(sa & 1) && !(sb & 1) && !(sc & 1) && !(n & 1);
    const bool vec = cx || even || g_tune.gemm_fast_unaligned != 0;
#define QS_FAST(CXF, TMF, TNF, VECF, EDGEF)                                                          \
    return launch_fast<CXF, TMF, TNF, VECF, EDGEF>(A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, \
                                                    accumulate, group_along_m, stream)
    // ---- exact form: every extent a whole number of tiles
    if (!cx && vec && k % 16 == 0) {
        if (m % 128 == 0 && n % 128 == 0) QS_FAST(false, 4, 4, true, false);
        if (m % 64 == 0 && n % 128 == 0) QS_FAST(false, 2, 4, true, false);
    }
    if (cx && k % 8 == 0) {
        if (m >= n && m % 128 == 0 && n % 64 == 0) QS_FAST(true, 4, 2, true, false);
        if (m % 64 == 0 && n % 128 == 0) QS_FAST(true, 2, 4, true, false);
        if (m % 128 == 0 && n % 64 == 0) QS_FAST(true, 4, 2, true, false);
        if (m % 64 == 0 && n % 64 == 0) QS_FAST(true, 2, 2, true, false);
    }
    // ---- edge form.  Round 1 measured it against the general kernel with the four shapes of that time: +3...9 % for fp64
    // products whose m and n are both >= 100, slower below (short tile lists: the general kernel's small and 96-wide shapes
    // won) and for complex128 (whose general kernel is already light on VALU work per MFMA).  With the tall shapes the
    // rule is an estimate: the edge form runs when its best shape's estimated time beats the general kernel's best
    // (`general_cost`, same formula, scaled by the general kernel's full-tile rate relative to this one).
    // g_tune.gemm_fast: 1 = that policy, 3 = edge form wherever it applies (tests, tuning).
    if (g_tune.gemm_fast != 1 && g_tune.gemm_fast != 3) return 1;
    if (cx) {
        if (g_tune.gemm_fast == 1) return 1;
        switch (pick_fast_shape(kC128Shapes, m, n, batch, nullptr)) {
            case 1: QS_FAST(true, 4, 2, true, true);
            case 2: QS_FAST(true, 2, 4, true, true);
            default: QS_FAST(true, 2, 2, true, true);
        }
    }
    double cost = 0.0;
    const int shape = pick_fast_shape(kF64Shapes, m, n, batch, &cost);
    if (!vec) cost /= 0.85;                 // 8-byte staging (l = 255 against 256: 52.9 / 67.3 TFLOP/s with the K tail and the padding taken out)
    if (g_tune.gemm_fast == 1 && !(cost <= general_cost)) return 1;
    if (vec) {
        switch (shape) {
            case 1: QS_FAST(false, 4, 4, true, true);
            case 2: QS_FAST(false, 2, 4, true, true);
            case 3: QS_FAST(false, 4, 2, true, true);
            case 5: QS_FAST(false, 3, 4, true, true);
            case 6: QS_FAST(false, 5, 2, true, true);
            case 7: QS_FAST(false, 6, 2, true, true);
            case 8: QS_FAST(false, 7, 2, true, true);
            case 9: QS_FAST(false, 3, 2, true, true);
            default: QS_FAST(false, 2, 2, true, true);
        }
    }
    switch (shape) {
        case 1: QS_FAST(false, 4, 4, false, true);
        case 2: QS_FAST(false, 2, 4, false, true);
        case 3: QS_FAST(false, 4, 2, false, true);
        case 5: QS_FAST(false, 3, 4, false, true);
        case 6: QS_FAST(false, 5, 2, false, true);
        case 7: QS_FAST(false, 6, 2, false, true);
        case 8: QS_FAST(false, 7, 2, false, true);
        case 9: QS_FAST(false, 3, 2, false, true);
        default: QS_FAST(false, 2, 2, false, true);
    }
#undef QS_FAST
}

}  // namespace qs
