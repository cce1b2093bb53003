// Strip kernels: the VALU-free tiled product for extents OFF the tile grid (round 4; VERDICT r03 "next" 2).
//
// Every contraction of the four-index transform has ONE small extent -- the basis size -- and one huge one:
//   d      T1[(abc), s]   = u[(abc), d] C[d, s]            m = L^3 rows,  n = M columns (small),  B = C shared
//   c,b,a  out[q, (cols)] = Ct[q, b] X[b, (cols)]          m = M rows (small), n = M ... M^3 columns, A = Ct shared
// The 128 x 128 (or 160 x 64, ...) tiles of qs_gemm_fast.hip quantise BOTH extents: a basis of 130 orbitals pays for 192
// columns in d and c and 160 rows in c, b, a (x 1.44 over the transform, profiles/r03_mid_size_shapes.txt), and its short tile
// lists run small tiles at ~0.75 of the full-tile rate.  Here the small extent is covered by ONE tile to the next multiple of 16
// (T = ceil(extent / 16) <= 16 row or column blocks), and the huge extent is cut into 128-wide pieces:
//   FORM 0 ("tall"):  tile = (16 T rows = all of m) x 128 columns; the eight waves sit side by side, each 16 columns wide,
//                     every wave multiplies ALL T row blocks of A (shared through LDS) with its own B fragment;
//                     the columns are VIRTUAL: column j of the product is column j % W of batch entry j / W, so a batch of
//                     narrow products (contraction c: L^2 products of M columns each) tiles as ONE long row of columns --
//                     no padding of the column extent at all;
//   FORM 1 ("wide"):  tile = 128 rows x (16 T columns = all of n); the eight waves are stacked, each 16 rows high, every
//                     wave multiplies its own A fragment with ALL T column blocks of B (the coefficient matrix, shared).
// One workgroup of eight waves per CU (two waves per SIMD), persistent over an XCD-chunked tile list.  The K loop is the one
// of qs_gemm_fast.hip: no VALU instruction (buffer loads with SGPR descriptors, scalar bases advanced on the scalar ALU, one
// loop-invariant lane offset; LDS addresses one VGPR + immediates), 16-deep stages double-buffered in LDS, global data two
// stages ahead in registers, rotated fragment schedule, k-steps of a tile's last stage beyond K skipped.  Rows / columns
// beyond the small extent multiply whatever the loads returned (zeros past the end of an operand: buffer range check) and
// are never stored; k >= K is zeroed in registers before the last stage is written to LDS.  Every element is the same chain
// of fused multiply-adds in increasing k as in the other tiled kernels: results are bit-identical (tests/test_gpu_kernels.py).
// Global items are 16 bytes (two adjacent elements) at ANY 8-byte-aligned address -- gfx950 carries such buffer loads out
// (tools/probe_unaligned.hip), so odd basis sizes stage like even ones; an odd column segment gets one dummy virtual column
// so that an item never straddles two segments.  (VEC = false, 8-byte items, is kept for A/B runs: -DQS_STRIP_ITEMS8.)

#include <type_traits>

#include <cmath>
#include <cstdlib>

#include "qs_common.h"
#include "qs_fast_items.h"

// Cache policy of the stores of the result: 0 = default (write-back through L2), 2 = non-temporal.  Non-temporal stores looked
// right (the output is read again only by the next contraction, after the whole tensor has passed through the caches) and are
// level at sizes whose rows are whole 128-byte lines -- but at every other size a row piece ends in PARTIAL lines, which L2
// merges under the default policy and the non-temporal path writes through one by one: default against non-temporal, same box:
// l = 97 37.6 -> 40.0, 100 41.9 -> 43.5, 130 45.8 -> 47.8 TFLOP/s (112, 144, 160: level; profiles/r04_strip_ablation.txt).
// (In the fused small-basis kernels, whose second pass reads the first one's result straight back, non-temporal stores cost
// a factor 1.75: l = 55 105 -> 184 us.)
#ifndef QS_STRIP_STORE_AUX
#define QS_STRIP_STORE_AUX 0
#endif

namespace qs {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct StripArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int64_t sb, sc;          // FORM 0: distance between the column segments (batch entries) of B and of C (elements)
    uint64_t a_end, b_end, c_end;   // one past the last byte of each operand
    int64_t big;             // FORM 0: virtual columns (segments x Wp);  FORM 1: rows m
    int64_t W;               // FORM 0: columns per segment
    int64_t Wp;              // FORM 0, fp64: W rounded up to even: the width of a segment in VIRTUAL columns -- an odd segment gets
                             // one dummy column at its end, so that a 16-byte item (two adjacent columns) never straddles two segments
    int small;               // FORM 0: rows m (of A and of the result);  FORM 1: columns n
    int k, nk;               // K, ceil(K / KT)
    unsigned total;          // tiles
    int nsmall;              // tiles along the SMALL extent (1: the whole of it in one tile of 16 T; up to 4 for extents beyond
                             // 16 T: virtual block w = tile w % nsmall of the small extent, tile w / nsmall of the big one -- the
                             // tiles that share a piece of the streamed operand are neighbours on one XCD and meet in its L2)
};

// CX = complex128 (re / im planes in LDS, four real matrix instructions per fragment pair in the order of the other tiled kernels:
// re += ar br, im += ar bi, re += (-ai) bi, im += ai br with a / b the LEFT / RIGHT operand of the product), T = blocks of 16
// along the small extent (shared fragments), WN = blocks of 16 per wave along the big extent (tile = 128 WN of it), SETS =
// register sets of global data in flight (2: two stages of lookahead).
template <bool CX, int FORM, int T, int WN, int SETS>
__global__ __launch_bounds__(512, 1)
void gemm_strip_kernel(const StripArgs g) {
    constexpr int NP = CX ? 2 : 1;                            // LDS planes
    // A rows: fp64 pads a row to KT + 2 (stride 18: conflict-free fragment reads).  complex128: no padding, the k index of a row
    // XOR-ed with 2 ((row >> 2) & 3) instead, as in qs_gemm_fast.hip (the padded rows of stride 10 collided on every write of the
    // A stage: SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles of the tall form at l = 100, profiles/r04_final_pmc_l100_c128.txt)
    constexpr int KT = CX ? 8 : 16, KS = KT / 4, NT = 512, SA = CX ? KT : KT + 2;
    constexpr int EPI = CX ? 1 : 2;                           // elements per 16-byte global item
    constexpr unsigned ESZ = CX ? 16 : 8;                     // bytes per element
    constexpr unsigned IB = 16;                               // bytes per global item
    constexpr int IPR_A = KT / EPI;                           // items per A row of a stage (8)
    constexpr int RA = NT / IPR_A;                            // A rows covered by one item step (64)
    constexpr int TILE = 128 * WN;                            // extent of a tile along the big extent
    constexpr int A_ROWS = FORM == 0 ? 16 * T : TILE;
    constexpr int NA = (A_ROWS + RA - 1) / RA;                // item steps of the A stage
    constexpr int B_COLS = FORM == 0 ? TILE : 16 * T;
    constexpr int IPR_B = B_COLS / EPI;                       // items per B row
    constexpr int B_ITEMS = KT * IPR_B;
    constexpr int NB = (B_ITEMS + NT - 1) / NT;               // item steps of the B stage
    constexpr int RPS = FORM == 0 ? NT / IPR_B : 0;           // FORM 0: B rows per item step, whole (half) rows per wave
    // row pitch of the B stage: 16 mod 32 doubles, so that the four k rows of a fragment read fall on different banks
    // fp64, FORM 0, two column blocks per wave: lane c owns the ADJACENT columns 32 w + 2 c and 32 w + 2 c + 1 (the trick of
    // qs_gemm_fast.hip): ONE 16-byte LDS read feeds both blocks and the result leaves in 16-byte stores -- half the store
    // instructions (the stores are what these kernels wait on, profiles/r04_strip_ablation.txt).  Row pitch 0 mod 32 for those reads.
#ifdef QS_STRIP_NO_PAIR      // (A/B builds)
    constexpr bool kPair = false;
#else
    constexpr bool kPair = !CX && FORM == 0 && WN == 2;
#endif
    constexpr int SB = FORM == 0 ? (kPair ? TILE : TILE + 16) : 16 * T + ((T & 1) ? 32 : 16);
    constexpr int A_PLANE = NA * RA * SA, B_PLANE = KT * SB;
    constexpr int A_STAGE = NP * A_PLANE, B_STAGE = NP * B_PLANE;
    static_assert(FORM == 1 || (NT % IPR_B == 0 && KT % RPS == 0), "FORM 0: item steps cover whole B rows");
    static_assert(KS == 2 || KS == 4, "two or four k-steps per stage");
    using Item = FastItem<true>;
    using item_t = typename Item::type;

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.nk;
    const unsigned P = gridDim.x;
    const int k_tail = g.k - (nk - 1) * KT;                   // valid k of a tile's last stage (1 ... KT)

    // ---- fetch cursor
    uint64_t a_ptr[NA];
    uint64_t b_ptr[FORM == 0 ? NB : 1];
    unsigned voff_b[FORM == 0 ? 1 : NB];
    unsigned f_v = blockIdx.x;
    int f_k = 0;
    bool f_valid = true;
    const unsigned voff_a = (unsigned)(tid / IPR_A) * (unsigned)g.lda * ESZ + (unsigned)(tid % IPR_A) * IB;
    // start of tile w of the big extent
    // start of virtual block v along the big extent, and (s0) along the small one
    auto tile_start = [&](unsigned v, int& s0) -> int64_t {
        unsigned w = __builtin_amdgcn_readfirstlane(xcd_chunked_index_fast(v, g.total));
        s0 = 0;
        if (g.nsmall > 1) {
            const unsigned b = __builtin_amdgcn_readfirstlane(w / (unsigned)g.nsmall);
            s0 = (int)(w - b * (unsigned)g.nsmall) * 16 * T;
            w = b;
        }
        return (int64_t)w * TILE;
    };
    auto aim = [&](unsigned v) __attribute__((always_inline)) {
        int s0;
        const int64_t t0 = tile_start(v, s0);
        if constexpr (FORM == 0) {
            // virtual column j -> element (j / Wp) * sb + j % Wp of its B row; the tile's first segment goes into the scalar base
            // (32-bit divisions: the host admits big + 512 < 2^32 only)
            const unsigned W = (unsigned)g.Wp;
            const unsigned seg0 = __builtin_amdgcn_readfirstlane((unsigned)t0 / W);
            const unsigned j = (unsigned)t0 + (unsigned)(tid % IPR_B) * EPI;
            const unsigned sj = j / W;
            voff_b[0] = (unsigned)(((int64_t)(sj - seg0) * g.sb + (j - sj * W)) * ESZ);
            const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)seg0 * g.sb * ESZ;
#pragma unroll
            for (int i = 0; i < NB; ++i)
                b_ptr[i] = uniform64(reinterpret_cast<uint64_t>(Bb + (size_t)(wave * 64 / IPR_B + i * RPS) * g.ldb * ESZ));
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_ptr[i] = uniform64(reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(g.A) + (size_t)(s0 + i * RA) * g.lda * ESZ));
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_ptr[i] = uniform64(reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(g.A) + (size_t)(t0 + i * RA) * g.lda * ESZ));
            b_ptr[0] = uniform64(reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(g.B) + (size_t)s0 * ESZ));
        }
    };
    // FORM 1: the B stage (a KT x 16 T piece of the coefficient matrix) as a flat item list; threads beyond it are parked on
    // the pad columns of LDS row 0 and on an offset past the end of the (small) matrix
    unsigned st_b_off[FORM == 0 ? 1 : NB];
    if constexpr (FORM == 0) {
        st_b_off[0] = (unsigned)((tid / IPR_B) * SB + (tid % IPR_B) * EPI);
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + NT * i;
            const bool ok = idx < B_ITEMS;
            voff_b[i] = ok ? (unsigned)(idx / IPR_B) * (unsigned)g.ldb * ESZ + (unsigned)(idx % IPR_B) * IB : 0x7FFFFFF0u;
            st_b_off[i] = ok ? (unsigned)((idx / IPR_B) * SB + (idx % IPR_B) * EPI) : (unsigned)(16 * T + 2 * (tid & 7));
        }
    }
    aim(f_v);
    const size_t a_step = KT * ESZ;
    const size_t b_step = (size_t)KT * g.ldb * ESZ;

    double* st_a = CX ? As + (tid / IPR_A) * SA + ((tid % IPR_A) ^ (2 * (((tid / IPR_A) >> 2) & 3)))
                      : As + (tid / IPR_A) * SA + (tid % IPR_A) * EPI;
    const double* rd_a = FORM == 0 ? As + (lane & 15) * SA + (lane >> 4)
                                   : As + (wave * 16 * WN + (lane & 15)) * SA + (lane >> 4);
    // complex: the swizzled position of k = 4 kk + (lane >> 4) depends on kk -- one base per k-step of a stage (KS = 2)
    const int swz_r = CX ? 2 * (((lane & 15) >> 2) & 3) : 0;
    const double* rd_a_cx[2] = {rd_a - (lane >> 4) + ((lane >> 4) ^ swz_r), rd_a - (lane >> 4) + ((4 + (lane >> 4)) ^ swz_r) - 4};
    const double* rd_b = FORM == 0 ? Bs + (lane >> 4) * SB + wave * 16 * WN + (kPair ? 2 : 1) * (lane & 15)
                                   : Bs + (lane >> 4) * SB + (lane & 15);

    item_t ra[SETS][NA], rb[SETS][NB];

    // Loads are issued UNCONDITIONALLY (a cursor that has run out of tiles loads with a zero range: the descriptor returns
    // zeros without touching memory), and so is the write of the next stage to LDS: with a load count that does not depend on
    // the path the compiler can wait for exactly the loads a stage needs instead of draining to the newest one.
    auto fetch = [&](auto set_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value;
#ifdef QS_STRIP_ABLATE_LOADS      // development: no memory traffic on the load side (zero range: the descriptor answers with zeros)
        const unsigned live = 0u;
#else
        const unsigned live = f_valid ? 0xFFFFFFFFu : 0u;
#endif
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[set][i] = Item::load(a_ptr[i], bytes_left(g.a_end, a_ptr[i]) & live, voff_a);
            a_ptr[i] += a_step;
        }
        if constexpr (FORM == 0) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                rb[set][i] = Item::load(b_ptr[i], bytes_left(g.b_end, b_ptr[i]) & live, voff_b[0]);
                b_ptr[i] += b_step;
            }
        } else {
            const unsigned room = bytes_left(g.b_end, b_ptr[0]) & live;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[set][i] = Item::load(b_ptr[0], room, voff_b[i]);
            b_ptr[0] += b_step;
        }
        if (f_valid && ++f_k == nk) {
            f_k = 0;
            f_v += P;
            f_valid = f_v < g.total;
            if (f_valid) aim(f_v);
        }
    };

    int s_k = 0;
    auto stash = [&](auto buf_c, auto set_c) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr int set = decltype(set_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        const bool tail = (s_k == nk - 1) && (k_tail < KT);
        if (++s_k == nk) s_k = 0;
        if (tail) {      // last k-stage of a tile with a K tail: zero the k >= K part of both operands
            const item_t zero = item_t(0.0);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if constexpr (CX) {
                    if ((tid % IPR_A) >= k_tail) ra[set][i] = zero;
                } else {
                    if ((tid % IPR_A) * 2 >= k_tail) ra[set][i][0] = 0.0;
                    if ((tid % IPR_A) * 2 + 1 >= k_tail) ra[set][i][1] = 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int row = FORM == 0 ? tid / IPR_B + i * RPS : (tid + NT * i) / IPR_B;
                if (row >= k_tail) rb[set][i] = zero;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            double* d = st_a + buf * A_STAGE + i * RA * SA;
            if constexpr (CX) { d[0] = ra[set][i][0]; d[A_PLANE] = ra[set][i][1]; }
            else *reinterpret_cast<f64x2*>(d) = ra[set][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double* d = Bs + buf * B_STAGE + (FORM == 0 ? st_b_off[0] + i * RPS * SB : st_b_off[i]);
            if constexpr (CX) { d[0] = rb[set][i][0]; d[B_PLANE] = rb[set][i][1]; }
            else *reinterpret_cast<f64x2*>(d) = rb[set][i];
        }
    };

    f64x4 acc[NP][T][WN];      // [re | im][block of the small extent][the wave's block of the big extent]

    // fragments of k-step kk: the T shared ones (`sf`) and the wave's own (`of`)
    auto read_frags = [&](auto buf_c, int kk, double (&sf)[NP][T], double (&of)[NP][WN]) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        const double* as = (CX ? rd_a_cx[kk & 1] : rd_a) + buf * A_STAGE;      // (rd_a_cx[1] is pre-biased by -4: the index below adds 4 kk)
        const double* bs = rd_b + buf * B_STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (FORM == 0) {
#pragma unroll
                for (int i = 0; i < T; ++i) sf[p][i] = as[p * A_PLANE + i * 16 * SA + kk * 4];
                if constexpr (kPair) {
                    const f64x2 v = *reinterpret_cast<const f64x2*>(bs + kk * 4 * SB);
                    of[0][0] = v[0];
                    of[0][1] = v[1];
                } else {
#pragma unroll
                    for (int o = 0; o < WN; ++o) of[p][o] = bs[p * B_PLANE + kk * 4 * SB + o * 16];
                }
            } else {
#pragma unroll
                for (int o = 0; o < WN; ++o) of[p][o] = as[p * A_PLANE + o * 16 * SA + kk * 4];
#pragma unroll
                for (int j = 0; j < T; ++j) sf[p][j] = bs[p * B_PLANE + kk * 4 * SB + j * 16];
            }
        }
    };
    auto mfma_step = [&](const double (&sf)[NP][T], const double (&of)[NP][WN], auto fresh_c) __attribute__((always_inline)) {
        constexpr bool fresh = decltype(fresh_c)::value;
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < T; ++s) {
#pragma unroll
            for (int o = 0; o < WN; ++o) {
                // a = fragment of the LEFT operand, b = of the RIGHT one
                if constexpr (!CX) {
                    const double a = FORM == 0 ? sf[0][s] : of[0][o], b = FORM == 0 ? of[0][o] : sf[0][s];
                    acc[0][s][o] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, fresh ? zero : acc[0][s][o], 0, 0, 0);
                } else {
                    const double ar = FORM == 0 ? sf[0][s] : of[0][o], ai = FORM == 0 ? sf[1][s] : of[1][o];
                    const double br = FORM == 0 ? of[0][o] : sf[0][s], bi = FORM == 0 ? of[1][o] : sf[1][s];
                    acc[0][s][o] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, fresh ? zero : acc[0][s][o], 0, 0, 0);
                    acc[1][s][o] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, fresh ? zero : acc[1][s][o], 0, 0, 0);
                    acc[0][s][o] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, acc[0][s][o], 0, 0, 0);
                    acc[1][s][o] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, acc[1][s][o], 0, 0, 0);
                }
            }
        }
    };

    // ---- the result leaves through buffer stores: SGPR descriptor (base advanced per row on the scalar ALU) + one 32-bit lane
    // offset per block; lanes that must not store carry an offset past the descriptor's range and are dropped by the hardware:
    // no VALU instruction and no branch per store.  Register r of a lane holds row (lane >> 4) + 4 r, column lane & 15 of a
    // 16 x 16 block.
    constexpr unsigned kDropped = 0xFFFFFFFFu, kRange = 0x80000000u;
    unsigned voff_c[WN];           // FORM 0: per tile; FORM 1: fixed (column block 0; later blocks through the immediate offset)
    uint64_t c_base = 0;           // FORM 0: first segment and first row of the tile;  FORM 1: first row of the wave and first column of the tile
    int c_s0 = 0;                  // the tile's start along the small extent
    if constexpr (FORM == 1) voff_c[0] = (unsigned)(((int64_t)(lane >> 4) * g.ldc + (lane & 15)) * ESZ);
    auto aim_stores = [&](unsigned v) __attribute__((always_inline)) {
        int s0;
        const int64_t t0 = tile_start(v, s0);
        c_s0 = s0;
        if constexpr (FORM == 0) {
            const unsigned W = (unsigned)g.Wp;
            const unsigned seg0 = __builtin_amdgcn_readfirstlane((unsigned)t0 / W);
            c_base = uniform64(reinterpret_cast<uint64_t>(reinterpret_cast<char*>(g.C) + ((int64_t)seg0 * g.sc + (int64_t)s0 * g.ldc) * ESZ));
            if constexpr (kPair) {
                // the lane's column pair (j even, never across a segment boundary: Wp is even): voff_c[0] for the 16-byte store of
                // both, voff_c[1] for the 8-byte store of the first one alone (the last column of an odd segment)
                const unsigned j = (unsigned)t0 + wave * 32 + 2 * (lane & 15);
                const unsigned sj = j / W;
                const unsigned rc = j - sj * W;
                const unsigned off = (unsigned)(((int64_t)(sj - seg0) * g.sc + rc + (int64_t)(lane >> 4) * g.ldc) * ESZ);
                const bool in = j < (unsigned)g.big;
                voff_c[0] = (in && rc + 1 < (unsigned)g.W) ? off : kDropped;
                voff_c[1] = (in && rc + 1 == (unsigned)g.W) ? off : kDropped;
            } else {
#pragma unroll
                for (int o = 0; o < WN; ++o) {
                    const unsigned j = (unsigned)t0 + wave * 16 * WN + o * 16 + (lane & 15);
                    const unsigned sj = j / W;
                    const bool ok = j < (unsigned)g.big && j - sj * W < (unsigned)g.W;
                    voff_c[o] = ok ? (unsigned)(((int64_t)(sj - seg0) * g.sc + (j - sj * W) + (int64_t)(lane >> 4) * g.ldc) * ESZ) : kDropped;
                }
            }
        } else {
            c_base = uniform64(reinterpret_cast<uint64_t>(reinterpret_cast<char*>(g.C) + ((t0 + wave * 16 * WN) * g.ldc + s0) * ESZ));
        }
    };
    auto store_one = [&](int s, int o, int r, const auto& rsrc, unsigned off, int soff) __attribute__((always_inline)) {
        const double re = acc[0][s][o][r];      // (a bit cast of a vector ELEMENT reads element 0: through scalars)
        if constexpr (CX) {
            const double im = acc[1][s][o][r];
            const f64x2 v = {re, im};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)off, soff, QS_STRIP_STORE_AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, re), rsrc, (int)off, soff, QS_STRIP_STORE_AUX);
        }
    };
    auto store_block = [&](int s, uint64_t ldc_b) __attribute__((always_inline)) {
#ifndef QS_STRIP_ABLATE_STORES
        if constexpr (FORM == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t base = uniform64(c_base) + (uint64_t)(s * 16 + 4 * r) * ldc_b;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), (short)0, (int)kRange, 0x00020000);
                // (a block that reaches past the last row -- wave-uniform; with one tile along the small extent only the last
                // block can)
                const bool partial = c_s0 + s * 16 + 16 > g.small;
                const bool row_ok = c_s0 + s * 16 + 4 * r + (lane >> 4) < g.small;
                if constexpr (kPair) {
                    const double v0 = acc[0][s][0][r], v1 = acc[0][s][1][r];
                    const f64x2 v = {v0, v1};
                    unsigned off = voff_c[0];
                    if (partial) off = row_ok ? off : kDropped;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)off, 0, QS_STRIP_STORE_AUX);
                    if (g.W & 1) {      // (wave-uniform: odd segments end in a single column)
                        unsigned off1 = voff_c[1];
                        if (partial) off1 = row_ok ? off1 : kDropped;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v0), rsrc, (int)off1, 0, QS_STRIP_STORE_AUX);
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < WN; ++o) {
                        unsigned off = voff_c[o];
                        if (partial) off = row_ok ? off : kDropped;
                        store_one(s, o, r, rsrc, off, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int o = 0; o < WN; ++o) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (the column block goes into the BASE, not into the instruction's scalar offset: a 16-byte store with a
                    // REGISTER scalar offset gets no wait states from the compiler before its data registers are rewritten --
                    // the complex form, whose (re, im) pairs are staged through one register quad, stored torn values)
                    const uint64_t base = uniform64(c_base) + (uint64_t)(o * 16 + 4 * r) * ldc_b + (uint64_t)(s * 16) * ESZ;
                    // rows past the last one lie past the end of C: dropped by the range check
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), (short)0,
                                                                        (int)bytes_left(g.c_end, base), 0x00020000);
                    unsigned off = voff_c[0];
                    if (c_s0 + s * 16 + 16 > g.small) off = (c_s0 + s * 16 + (lane & 15) < g.small) ? off : kDropped;      // (columns >= n)
                    store_one(s, o, r, rsrc, off, 0);
                }
            }
        }
#endif
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;
    using S1 = std::integral_constant<int, SETS == 2 ? 1 : 0>;      // register set of the odd stages

    const unsigned my_tiles = (g.total - blockIdx.x + P - 1) / P;
    const int stages = (int)my_tiles * nk;      // (the host keeps tiles x stages below 2^31)

    fetch(B0{});                       // global stage 0
    stash(B0{}, B0{});
    __syncthreads();
    double s0[NP][T], s1[NP][T], o0[NP][WN], o1[NP][WN];
    fetch(S1{});                       // stage 1
    if constexpr (SETS == 2) fetch(B0{});      // stage 2
    read_frags(B0{}, 0, s0, o0);

    unsigned c_v = blockIdx.x;
    int c_k = 0;

    // One global stage.  The LDS buffer the next stage goes into was last read before the PREVIOUS barrier, so it is free
    // from the start of this stage: the next stage is written (and the registers it leaves refilled from memory) right
    // after the first block of MFMAs, ahead of the barrier.
    auto stage = [&](auto cur_c, int gs) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_c)::value;
        using NXT = std::integral_constant<int, cur ^ 1>;
        using NSET = std::integral_constant<int, SETS == 2 ? (cur ^ 1) : 0>;
        const bool has_next = gs + 1 < stages;
        const int ks_live = (c_k == nk - 1) ? (k_tail + 3) / 4 : KS;
        read_frags(cur_c, 1, s1, o1);
        __builtin_amdgcn_sched_barrier(0);
        if (c_k == 0) mfma_step(s0, o0, T_{}); else mfma_step(s0, o0, F_{});
        __builtin_amdgcn_sched_barrier(0);
        stash(NXT{}, NSET{});                    // stage gs + 1 (behind the last stage: zeros, never read)
        __builtin_amdgcn_sched_barrier(0);
        fetch(NSET{});                           // stage gs + 3 (two register sets) / gs + 2 (one)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KS == 4) {
            read_frags(cur_c, 2, s0, o0);
            __builtin_amdgcn_sched_barrier(0);
            if (1 < ks_live) mfma_step(s1, o1, F_{});
            __builtin_amdgcn_sched_barrier(0);
            read_frags(cur_c, 3, s1, o1);
            __builtin_amdgcn_sched_barrier(0);
            if (2 < ks_live) mfma_step(s0, o0, F_{});
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef QS_STRIP_ABLATE_BARRIER   // development: no stage barrier (results wrong; what the barrier costs)
        __syncthreads();
#endif
        if (has_next) read_frags(NXT{}, 0, s0, o0);
        __builtin_amdgcn_sched_barrier(0);
        if (KS - 1 < ks_live) mfma_step(s1, o1, F_{});
        __builtin_amdgcn_sched_barrier(0);
        if (++c_k == nk) {      // the tile's result leaves behind its last block
            aim_stores(c_v);
            // (opaque per tile: the row offsets (16 s + 4 r) ldc are loop-invariant, and hoisted out of the kernel's main loop
            // by the dozen they would not fit the scalar registers)
            uint64_t ldc_b = (uint64_t)g.ldc * ESZ;
            asm volatile("" : "+s"(ldc_b));
#pragma unroll
            for (int s = 0; s < T; ++s) store_block(s, ldc_b);
            c_k = 0;
            c_v += P;
        }
    };

    for (int gs = 0; gs < stages; gs += 2) {
        stage(B0{}, gs);
        if (gs + 1 < stages) stage(B1{}, gs + 1);
    }
}

namespace {

// relative rate of a strip tile against the 128 x 128 tile of qs_gemm_fast.hip (same units as its shape weights)
inline double strip_weight(bool cx, int t, bool wide) {
    if (g_tune.gemm_strip_w > 0) return 0.01 * g_tune.gemm_strip_w;
    // complex (against the 64 x 128 tile of the general kernel; same-box sweep profiles/r04_strip_c128_sweep.txt: at equal tile
    // extents the general kernel's whole tiles stay ahead by 3-5 %: l = 96 60.5 against 57.7, l = 120 57.5 against 55.9 TFLOP/s)
    if (cx) return t >= 8 ? 0.95 : t == 6 ? 0.96 : t >= 4 ? 1.0 : 0.85;
    // (same-box sweep, profiles/r04_strip_sweep.txt: at equal tile extents -- l = 120, T = 8 -- the strip tile runs at 0.99 of
    // the 128 x 128 edge-form tile)
    return (t >= 8 ? 1.0 : t >= 5 ? 0.95 : 0.80) * (wide ? 1.06 : 1.0);
}

template <bool CX, int FORM, int T, int WN>
int launch_strip(StripArgs g, hipStream_t stream) {
    // two register sets of global data in flight where the registers allow it
    constexpr int SETS = ((CX ? 2 : 1) * T * WN <= 12) ? 2 : 1;
    constexpr int KT = CX ? 8 : 16, NP = CX ? 2 : 1;
    constexpr int TILE = 128 * WN;
    constexpr int A_ROWS = FORM == 0 ? 16 * T : TILE;
    constexpr int NA = (A_ROWS + 63) / 64;
#ifdef QS_STRIP_NO_PAIR
    constexpr bool kPair = false;
#else
    constexpr bool kPair = !CX && FORM == 0 && WN == 2;
#endif
    constexpr int SB = FORM == 0 ? (kPair ? TILE : TILE + 16) : 16 * T + ((T & 1) ? 32 : 16);
    const size_t lds = sizeof(double) * 2 * NP * (size_t)(NA * 64 * (CX ? KT : KT + 2) + KT * SB);
    const int64_t tiles = cdiv(g.big, TILE) * g.nsmall;
    if (tiles * g.nk >= (int64_t(1) << 31)) return 1;
    g.total = (unsigned)tiles;
    int64_t P = device_cu_count();
    P -= P % 8;
    if (P < 8) P = 8;
    if (P > tiles) P = tiles;
    auto kern = gemm_strip_kernel<CX, FORM, T, WN, SETS>;
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)kern, lds, lds_opt_in, "hipFuncSetAttribute(gemm_strip)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(512), lds, stream, g);
    note_dispatch("qs::gemm_strip_kernel<%s, %d, %d, %d, %d>", CX ? "true" : "false", FORM, T, WN, SETS);
    return launch_status("gemm_strip launch");
}

// wide: two blocks of 16 per wave along the big extent (tiles of 256): half the staging of the shared operand per product, where
// the accumulators fit (fp64: T <= kWideMaxT) and the tile list is long enough to fill the chip
constexpr int kWideMaxT = 10;
constexpr int kMaxT = 16, kMaxTComplex = 8;
constexpr int kMaxSmallTiles = 4;      // tiles along the small extent (fp64 up to 1024, complex128 up to 512 orbitals)

template <int FORM>
int launch_strip_t(bool cx, int t, bool wide, const StripArgs& g, hipStream_t stream) {
    if (cx) {
        switch (t) {
#ifdef QS_DEV_FEW_SHAPES      // development / sanitizer builds of the HOST side
            case 5: return launch_strip<true, FORM, 5, 1>(g, stream);
#else
            // (256-wide complex tiles measured and dropped: T = 5 spills, 31.9 against 39.7 TFLOP/s at l = 66; T = 4 level)
#define QS_STRIP(TT) case TT: return launch_strip<true, FORM, TT, 1>(g, stream);
            QS_STRIP(1) QS_STRIP(2) QS_STRIP(3) QS_STRIP(4) QS_STRIP(5) QS_STRIP(6) QS_STRIP(7) QS_STRIP(8)
#undef QS_STRIP
#endif
            default: return 1;
        }
    }
    switch (t) {
#ifdef QS_DEV_FEW_SHAPES
        case 9: return wide ? launch_strip<false, FORM, 9, 2>(g, stream) : launch_strip<false, FORM, 9, 1>(g, stream);
#else
#define QS_STRIP(TT) case TT: return launch_strip<false, FORM, TT, 1>(g, stream);
#define QS_STRIP_W(TT) case TT: return wide ? launch_strip<false, FORM, TT, 2>(g, stream) : launch_strip<false, FORM, TT, 1>(g, stream);
        QS_STRIP_W(1) QS_STRIP_W(2) QS_STRIP_W(3) QS_STRIP_W(4) QS_STRIP_W(5) QS_STRIP_W(6) QS_STRIP_W(7) QS_STRIP_W(8)
        QS_STRIP_W(9) QS_STRIP_W(10) QS_STRIP(11) QS_STRIP(12) QS_STRIP(13) QS_STRIP(14) QS_STRIP(15) QS_STRIP(16)
#undef QS_STRIP
#undef QS_STRIP_W
#endif
        default: return 1;
    }
}

}  // namespace

// QS_OK after launching, 1 = not eligible / not the cheapest (the caller goes on to the other kernels).
// other_cost: the best estimate of the other tiled kernels for this product, in the units of qs_gemm_fast.hip (tiles x tile
// area / relative rate over two workgroups per CU).
int gemm_strip_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k, int64_t lda,
                   int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb, int64_t sc, int accumulate,
                   double other_cost, hipStream_t stream) {
    if (!g_tune.gemm_strip || accumulate) return 1;
    if (m <= 0 || n <= 0 || k <= 0 || batch <= 0 || k >= (int64_t(1) << 30)) return 1;
    const bool cx = dtype == QS_C128;
    const int64_t esz = cx ? 16 : 8;
    static const int max_t_env = [] { const char* e = getenv("QS_STRIP_MAXT"); return e ? atoi(e) : 0; }();      // (tuning runs)
    const int max_t = (max_t_env > 0 && !cx) ? max_t_env : (cx ? kMaxTComplex : kMaxT);
    const int max_small = 16 * max_t * kMaxSmallTiles;
    if (cx && (!aligned(A, 16) || !aligned(B, 16) || !aligned(C, 16))) return 1;
    // which extent is the small one: A shared by the batch and m small -> tall tiles over virtual columns; otherwise one
    // product with n small -> wide tiles over the rows
    int form;
    if ((batch == 1 || sa == 0) && m <= max_small && (batch > 1 || n >= m)) form = 0;
    else if (batch == 1 && n <= max_small) form = 1;
    else return 1;
    if (batch == 1) { sb = 0; sc = 0; }
    StripArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sb = sb; g.sc = sc;
    g.k = (int)k; g.nk = (int)cdiv(k, cx ? 8 : 16);
    int t;
    if (form == 0) {
        // Lane offsets are 32-bit and unsigned: a tile's columns reach at most 256 / n + 1 segments past its first one.  (Columns
        // past the last one of the last tile compute offsets of their own -- whatever they address is either inside the
        // operand or cut off by the range check of the buffer descriptor, and they are never stored.)
        if (sb < 0 || sc < 0 || lda < k || ldb < n || ldc < n) return 1;
        if (((256 / n + 2) * sb + n) * esz >= (int64_t(1) << 32) - 65536 || 64 * lda * esz >= (int64_t(1) << 31)) return 1;
        if (((256 / n + 2) * sc + n + 4 * ldc) * esz >= (int64_t(1) << 31)) return 1;      // store offsets stay below the descriptor's range
        g.W = n; g.Wp = cx ? n : n + (n & 1);
        if (g.Wp * batch + 512 >= (int64_t(1) << 32)) return 1;      // the kernel's column arithmetic is 32-bit
        g.big = g.Wp * batch; g.small = (int)m;
        t = (int)cdiv(m, 16);
    } else {
        if (lda < k || ldb < n || ldc < n) return 1;
        if (64 * lda * esz >= (int64_t(1) << 31) || 16 * ldb * esz + 4096 >= (int64_t(1) << 30) || 4 * ldc * esz + 8192 >= (int64_t(1) << 31)) return 1;
        g.big = m; g.W = g.Wp = 0; g.small = (int)n;
        t = (int)cdiv(n, 16);
    }
    // an extent beyond one tile (16 blocks of fp64, 8 of complex128): the fewest tiles along it, equally high
    g.nsmall = (int)cdiv(t, max_t);
    // fp64, an even number of 12 ... 16 blocks: two tiles of half the height -- they fit the 256-wide form (half the staging of
    // the shared operand per product), the streamed operand is read twice, the repeat from L2: l = 190 +1.5 %, 224 +3 %, 253 +3 %
    // (176 = 11 and 208 = 13 blocks would pad to 12 / 14: -5 % / -3.5 %; profiles/r04_strip_ablation.txt)
    if (!cx && max_t_env <= 0 && g.nsmall == 1 && t >= 12 && t % 2 == 0) g.nsmall = 2;
    t = (int)cdiv(t, g.nsmall);
    g.a_end = reinterpret_cast<uint64_t>(A) + (uint64_t)(((m - 1) * lda + k) * esz);
    g.b_end = reinterpret_cast<uint64_t>(B) + (uint64_t)(((batch - 1) * sb + (k - 1) * ldb + n) * esz);
    g.c_end = reinterpret_cast<uint64_t>(C) + (uint64_t)(((batch - 1) * sc + (m - 1) * ldc + n) * esz);
    // wide tiles (256 of the big extent) where the accumulators fit and the list still fills the chip a few times over
    const double slots = device_cu_count();
    static const int wide_env = [] { const char* e = getenv("QS_STRIP_WIDE"); return e ? atoi(e) : -1; }();      // (tuning runs)
    bool wide = !cx && t <= kWideMaxT && cdiv(g.big, 256) >= 4 * (int64_t)slots;
    if (wide_env >= 0) wide = !cx && wide_env != 0 && t <= kWideMaxT;
    const int64_t tiles = cdiv(g.big, wide ? 256 : 128) * g.nsmall;
    if (g_tune.gemm_strip == 1 && t * g.nsmall < 3) return 1;      // (up to 32 rows / columns: the other kernels' ground, not measured here)
    if (g_tune.gemm_strip == 1) {
        // estimated time: rounds of the tile list over the CUs (one eight-wave workgroup each = both slots of the other
        // kernels' two four-wave workgroups) x tile area / relative rate
        const double rounds = tiles > 8 * slots ? tiles / slots : ceil(tiles / slots);
        // (several tiles along the small extent read the streamed operand that often, the repeats from L2)
        const double cost = rounds * (16.0 * t) * (wide ? 256.0 : 128.0) / 2.0 / (strip_weight(cx, t, wide) * (g.nsmall > 1 ? 0.97 : 1.0));
        if (!(cost < other_cost)) return 1;
    }
    return form == 0 ? launch_strip_t<0>(cx, t, wide, g, stream) : launch_strip_t<1>(cx, t, wide, g, stream);
}

}  // namespace qs
