// Two contractions in one pass for small bases, on the 4-wide fp64 matrix instruction:
//
//     Out_t = Lm . In_t . R        for a batch of L x L matrices In_t  (L, M <= 64, fp64)
//
// Both halves of the four-index transform are this product (basis_set.py:341-348):
//   (d, c):  item t = slab (a, b),   In_t = u[a, b, :, :]        R = C,    Lm = C^T    -> T2[a, b, :, :]
//   (b, a):  item t = column (r, s), In_t = T2[:, :, r, s]       R = Ct^T, Lm = Ct     -> out[:, :, r, s]
// so the transform of a small basis is TWO passes over the tensor (read once, write once each) instead of
// four.  The two passes differ only in the strides of an item's elements.
//
// Why v_mfma_f64_4x4x4_4b_f64: it issues four independent 4x4x4 products ("blocks") at the same flop rate
// as v_mfma_f64_16x16x4_f64 (measured: tools/probe_mfma4.hip, 75.9-77.3 vs 75.2 TFLOP/s), and
//   * extents pad to a multiple of 4, not 16: 55 orbitals -> 56 instead of 64, 0.77x the MFMA work of the
//     16-wide form per product (1.38x -> 1.06x of the unpadded work);
//   * the four blocks are four ITEMS (four adjacent slabs / four adjacent columns), so nothing is replicated in
//     the tensor operand and no remainder logic exists anywhere: every loop runs ceil(l / 4) times;
//   * its accumulator layout is its own B-operand layout (lane = x + 4 y + 16 z: A holds row x, block y, k z;
//     B holds k z, block y, column x; D holds row z, block y, column x -- tools/probe_mfma4_layout.hip), so
//     Y = In . R goes from the accumulators straight into Lm . Y: no LDS round trip, no shuffle -- and the
//     per-element sums are the same k-ordered FMA chains as in the 16-wide kernels (bit-identical results).
//
// Work split: a workgroup takes an item quad; its four waves (one per SIMD) take the column groups of R in
// chunks of <= 4 groups and rotate the chunks from round to round (14 groups = 4 + 4 + 3 + 3 would otherwise
// leave two SIMDs idle a quarter of the time).  Per chunk a wave keeps in registers: its fragments of R
// (ceil(l/4) x 4), the output accumulators (ceil(l/4) x 4), a two-deep ring of In fragments loaded straight from
// global memory two row quads ahead -- across chunk and item boundaries -- and reads the fragments of Lm from an
// LDS table with compile-time offsets.  No VALU work, no barrier and no branch inside a chunk: addresses are
// one lane offset (four variants: interior / last row quad / last column quad / both, with out-of-range lanes
// parked past num_records so that the hardware returns 0.0 / drops the store) plus scalar offsets.
// Algorithmic bytes per launch: 8 (L^2 + M^2) per item; roofline: HBM below l ~ 64 (the matrix pipe needs
// 2 x 2 x ceil(l/4)^3 x 16 cycles per item quad per CU).

#include <type_traits>

#include "qs_common.h"

// Development builds only (never defined in the shipped library): bit mask of parts to leave out, to find
// what bounds the kernel.  1 fetch, 2 stores, 4 transit through LDS, 8 Lm fragment reads, 16 all MFMAs, 32 return
// after the tables are built, 64 return at once
#ifndef QS_S4_ABLATE
#define QS_S4_ABLATE 0
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

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifdef QS_S4_TRACE      // development: shader-clock stamps of one workgroup's wave 0 at every step (qs_s4_trace symbol)
__device__ unsigned long long qs_s4_trace[4096];
__device__ unsigned qs_s4_trace_n;
// the stamp index lives in a register (a counter in memory would put a global load in front of every stamp)
#define QS_S4_STAMP(tag)                                                                                  \
    if (blockIdx.x == QS_S4_TRACE && wave == 0) {                                                         \
        if (lane == 0 && tr_n < 4096) qs_s4_trace[tr_n] = (__builtin_amdgcn_s_memtime() << 8) | (tag);    \
        ++tr_n;                                                                                           \
    }
#define QS_S4_TRACE_DONE                                                                     \
    if (blockIdx.x == QS_S4_TRACE && wave == 0 && lane == 0) {                                \
        qs_s4_trace_n = tr_n;                                                                \
        qs_s4_trace[4095] = __builtin_amdgcn_s_memrealtime() - tr_real0;     /* 100 MHz ticks, entry to exit */ \
    }
#else
#define QS_S4_STAMP(tag)
#define QS_S4_TRACE_DONE
#endif

constexpr unsigned kParked = 0x80000000u;     // lane offset of an out-of-range lane (>= num_records)

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    if constexpr (QS_S4_ABLATE & 16) return a + b + c;
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

}  // namespace

struct S4Args {
    const double* in;
    double* out;
    const double* R;      // R[k][j]  = R[k * r_sk + j * r_sj],   L x M
    const double* Lm;     // Lm[p][a] = Lm[p * l_sp + a * l_sa],  M x L
    int64_t r_sk, r_sj, l_sp, l_sa;
    int64_t in_item, in_row, in_col;       // element strides of In_t[i][k]
    int64_t out_item, out_row, out_col;    // element strides of Out_t[p][j]
    int L, M;
    unsigned nitems, nquads;
    int mode;      // bit 0: the four waves take four ADJACENT item quads and the same chunk (else: one quad, four chunks);
                   // bit 1: with bit 0, a workgroup barrier per step keeps the four waves' fetches together in L1
};

// N4 = ceil(L / 4) = ceil(M / 4)
template <int N4>
__global__ __launch_bounds__(256, 1) void sandwich4_kernel(const S4Args g) {
    constexpr int NCH = (N4 + 3) / 4;                       // chunks of column groups, one per wave
    constexpr int NJ_BIG = (N4 + NCH - 1) / NCH;            // the first N4 % NCH chunks (all if it divides) have this many
    constexpr int NJ_SMALL = N4 / NCH;
    constexpr int N_BIG = (N4 % NCH) ? (N4 % NCH) : NCH;
    static_assert(NCH <= 4 && NJ_BIG <= 4, "a workgroup has four waves");
    static_assert(N4 >= 3, "the fetch runs three row quads ahead");

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* rtab = lds;                    // [ks][jg][16]: R[4 ks + z][4 jg + x]  at z * 4 + x
    double* ltab = lds + N4 * N4 * 16;     // [pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    double* transit = lds + 2 * N4 * N4 * 16;   // [wave][ks][64]: a row quad of In fragments on its way into MFMA lane order
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned tr_n = 0;
    (void)tr_n;
#ifdef QS_S4_TRACE
    const unsigned long long tr_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (QS_S4_ABLATE & 64) return;
    QS_S4_STAMP(253)
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int e_lane = z * 4 + x;
    const int rl = L & 3, rm = M & 3;      // valid rows / columns of the last quad (0 = all four)

    // ---- work units.  Every XCD takes a contiguous range of them and neighbouring workgroups of an XCD take
    // neighbouring units (they share 128-byte lines of the tensor in the XCD's L2).
    //   mode bit 0 clear: unit = item quad; the four waves take its chunks and rotate them from round to round
    //   mode bit 0 set:   unit = (group of four adjacent item quads, chunk); wave w takes quad 4 g + w.  The four
    //                     fetch streams then cover whole 128-byte lines when the item is the fastest index (the
    //                     (b, a) pass), and no fragment is fetched by more than one wave.  The chunk of a unit
    //                     rotates with the round, so every workgroup sees all chunk sizes.
    const bool grouped = g.mode & 1, step_barrier = (g.mode & 3) == 3;
    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = blockIdx.x / n_xcd, slots = gridDim.x / n_xcd;
    const unsigned ngroups = (g.nquads + 3) / 4;
    const unsigned nunits = grouped ? ngroups * NCH : g.nquads;
    // (grouped: the chunk units of a group must fall into the same round of the same XCD, or the rotation would
    // hand one chunk out twice: an XCD's range and the workgroup stride are whole groups)
    unsigned per = (nunits + n_xcd - 1) / n_xcd;
    if (grouped) per = (per + NCH - 1) / NCH * NCH;
    const unsigned u_end = (xcd + 1) * per < nunits ? (xcd + 1) * per : nunits;
    unsigned unit = xcd * per + slot;
    if (unit >= u_end) return;                         // (whole workgroup: before any barrier)
    const bool idle = !grouped && wave >= NCH;         // (such a wave still helps to build the tables, then leaves)
    auto quad_of = [&](unsigned u) __attribute__((always_inline)) { return grouped ? (u / NCH) * 4 + wave : u; };
    auto chunk_of = [&](unsigned u, unsigned round) __attribute__((always_inline)) {
        return (int)(grouped ? (u % NCH + round) % NCH : (wave + round) % NCH);
    };
    unsigned iq = quad_of(unit);

    const unsigned ka_step = (unsigned)(4 * g.in_row * 8), ks_step = (unsigned)(4 * g.in_col * 8);
    const unsigned pg_step = (unsigned)(4 * g.out_row * 8), jg_step = (unsigned)(4 * g.out_col * 8);

    auto rsrc = [&](const double* base, unsigned quad, int64_t item_stride) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base + (int64_t)quad * 4 * item_stride);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 0x7fffffff, 0x00020000);
    };
    // An In fragment (A operand: row x, block y, k z) is FETCHED in memory order and put into MFMA lane order on
    // its way through LDS: the texture-address unit only merges ADJACENT lanes, and in the MFMA order adjacent
    // lanes are rows (d, c: 440 bytes apart) -- 64 separate accesses per instruction, which bound the first
    // version of this kernel (TA 73 % busy, matrix pipe 38 %).  For the fetch, the lane digit that runs fastest
    // takes the role whose stride is smallest (k for a slab: 32-byte runs; the item for a column: 32-byte runs).
    int fr, fi, fk;          // role values (row, item, k in 0..3) of this lane when it fetches
    {
        const int d0 = lane & 3, d1 = (lane >> 2) & 3, d2 = lane >> 4;
        const int64_t sr = g.in_row, si = g.in_item, sk = g.in_col;
        // rank of each role's stride (0 = smallest); ties broken row < item < k
        const int rank_r = (si < sr) + (sk < sr), rank_i = (sr <= si) + (sk < si), rank_k = (sr <= sk) + (si <= sk);
        fr = rank_r == 0 ? d0 : rank_r == 1 ? d1 : d2;
        fi = rank_i == 0 ? d0 : rank_i == 1 ? d1 : d2;
        fk = rank_k == 0 ? d0 : rank_k == 1 ? d1 : d2;
    }
    // Position of element (row, item, k) of a fragment in its 512-byte transit slot: 128-byte line k, 8-byte bank
    // pair (row ^ k) + 4 (item ^ k).  Any 16 lanes that go to LDS together -- (row, item) at fixed k when the MFMA
    // lanes read, (k, row) at fixed item or (item, k) at fixed row when the fetch lanes write -- hit 16 different
    // bank pairs; the plain lane order would put four lanes of every write on each bank.
    auto slot_pos = [](int row, int item, int k) __attribute__((always_inline)) {
        return (unsigned)(16 * k + ((row ^ k) + 4 * (item ^ k)));
    };
    const unsigned transit_wr = slot_pos(fr, fi, fk);
    const unsigned transit_rd = slot_pos(lane & 3, (lane >> 2) & 3, lane >> 4);
    double* const my_transit = transit + wave * (N4 * 64);
    // lane offsets of a fetch.  Variant bit 0: last row quad, bit 1: last k quad.  A lane whose row / k / item
    // does not exist is parked.
    auto in_offsets = [&](unsigned quad, bool live, unsigned (&v)[4]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)((fr * g.in_row + fi * g.in_item + fk * g.in_col) * 8);
        const bool item_ok = live && quad * 4 + fi < g.nitems;
        const bool row_ok = rl == 0 || fr < rl, k_ok = rl == 0 || fk < rl;
        v[0] = item_ok ? base : kParked;
        v[1] = item_ok && row_ok ? base : kParked;
        v[2] = item_ok && k_ok ? base : kParked;
        v[3] = item_ok && row_ok && k_ok ? base : kParked;
    };
    // lane offsets of an Out fragment (D: row z, block y, column x).  Bit 0: last row quad, bit 1: last column quad
    auto out_offsets = [&](unsigned quad, unsigned (&v)[4]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)((z * g.out_row + y * g.out_item + x * g.out_col) * 8);
        const bool item_ok = quad * 4 + y < g.nitems;
        const bool row_ok = rm == 0 || z < rm, col_ok = rm == 0 || x < rm;
        v[0] = item_ok ? base : kParked;
        v[1] = item_ok && row_ok ? base : kParked;
        v[2] = item_ok && col_ok ? base : kParked;
        v[3] = item_ok && row_ok && col_ok ? base : kParked;
    };

    double ring[2][N4];       // In fragments of two row quads in MFMA lane order: stage, register ks
    double stg[2][N4];        // the same as fetched (memory lane order), on their way to the transit buffer

    // One fragment of row quad ka (k quad ks) is fetched with lane offset v (which carries the row quad: one
    // VALU add per row quad) and scalar offset ks * ks_step (ceil(l/4) loop-invariant SGPRs).
    auto fetch_frag = [&](auto rs, unsigned v, auto KS) __attribute__((always_inline)) {
        if constexpr (QS_S4_ABLATE & 1) return 1.0 + decltype(KS)::value;
        else
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)v, (int)(decltype(KS)::value * ks_step), 0));
    };
    auto fetch_quad_row = [&](auto rs, const unsigned (&v)[4], auto KA, auto STAGE) __attribute__((always_inline)) {
        constexpr int ka = decltype(KA)::value, st = decltype(STAGE)::value;
        const unsigned v_mid = v[ka == N4 - 1 ? 1 : 0] + ka * ka_step;
        const unsigned v_end = v[(ka == N4 - 1 ? 1 : 0) | 2] + ka * ka_step;      // last k quad
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            stg[st][decltype(KS)::value] = fetch_frag(rs, decltype(KS)::value == N4 - 1 ? v_end : v_mid, KS);
        });
    };
    // fetched fragment -> transit buffer (each lane writes where the MFMA lane that wants its value will read)
    // -> ring stage, lane-linear.  One wave, one buffer: LDS executes a wave's accesses in order, no barrier.
    auto settle_quad_row = [&](auto FROM, auto TO) __attribute__((always_inline)) {
        constexpr int from = decltype(FROM)::value, to = decltype(TO)::value;
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            my_transit[decltype(KS)::value * 64 + transit_wr] = stg[from][decltype(KS)::value];
        });
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            ring[to][decltype(KS)::value] = my_transit[decltype(KS)::value * 64 + transit_rd];
        });
    };

    // One chunk: NJ column groups starting at jg0 for the item quad behind (rs_in, v_in); the ring already
    // holds row quads 0 and 1 of it.  While row quad ka multiplies, row quad ka + 2 is fetched -- of this item
    // quad, or (the last two) of the next one, behind (rs_pf, v_pf).  P = parity of the ring stage of ka = 0.
    // ONE body serves chunks of NJ and of NJ - 1 column groups: the last group rides in two blocks of their own
    // per step that a narrow chunk (full == false, wave-uniform) branches around.  Two bodies would leave the ring
    // and the in-flight fetch registers in different physical registers, and the copies at the merge cost a
    // drain to vmcnt(0) plus ~90 moves per unit.
    auto chunk = [&](auto NJC, auto PC, int jg0, bool full, auto rs_in, auto rs_out, const unsigned (&v_out)[4],
                     auto rs_pf, const unsigned (&v_pf)[4], const unsigned (&v_in)[4]) __attribute__((always_inline)) {
        constexpr int NJ = decltype(NJC)::value, P = decltype(PC)::value;
        constexpr int NJM = (NJ_BIG == NJ_SMALL) ? NJ : NJ - 1;      // groups every chunk has
        double bf[N4][NJ];
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                bf[decltype(KS)::value][decltype(J)::value] =
                    rtab[(decltype(KS)::value * N4 + jg0 + decltype(J)::value) * 16 + e_lane];
            });
        });
        double acc2[N4][NJ];
        // fragments of Lm for row quad ka; each register is refilled for row quad ka + 1 right after its last use
        double lf[N4];
        unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
            lf[decltype(PG)::value] = ltab[(decltype(PG)::value * N4) * 16 + e_lane];
        });
        auto store_row = [&](auto PG) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value;
            const unsigned v_mid = v_out[pg == N4 - 1 ? 1 : 0] + pg * pg_step;
            const unsigned v_end = v_out[(pg == N4 - 1 ? 1 : 0) | 2] + pg * pg_step;     // last column quad
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const double val = acc2[pg][j];
                const bool last_col = jg0 + j == N4 - 1;                  // wave-uniform
                unsigned vo = last_col ? v_end : v_mid;
                if (j >= NJM && !full) vo = kParked;                      // this chunk has no such group
                if constexpr (QS_S4_ABLATE & 2) { if (val == 12345.678) my_transit[0] = val; }
                else
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), rs_out, (int)vo,
                                                      (int)((jg0 + j) * jg_step), 0);
            });
        };
        // Every memory instruction of a step rides between MFMAs (one k quad / one row quad of MFMAs carries at most
        // three of them); the sched_barriers keep the compiler from gathering them into one block, during which
        // the matrix pipe of this one-wave SIMD would stand still.
        unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
            constexpr int ka = decltype(KA)::value, st = (ka + P) & 1;
            QS_S4_STAMP(ka)
            if (step_barrier) __builtin_amdgcn_s_barrier();
            constexpr bool pf_next = ka + 3 >= N4;             // the fetch already belongs to the next item quad
            constexpr int ka_f = pf_next ? ka + 3 - N4 : ka + 3;
            const unsigned vf_mid = (pf_next ? v_pf : v_in)[ka_f == N4 - 1 ? 1 : 0] + ka_f * ka_step;
            const unsigned vf_end = (pf_next ? v_pf : v_in)[(ka_f == N4 - 1 ? 1 : 0) | 2] + ka_f * ka_step;
            // ---- Y[ka rows][own columns] = In[ka rows][:] . R[:][own columns];   meanwhile row quad ka + 1
            // (fetched two steps ago) goes into the transit buffer and row quad ka + 3 is fetched into its registers
            double acc1[NJ];
            unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                unroll<0, NJM>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    acc1[j] = mfma4(ring[st][ks], bf[ks][j], ks == 0 ? 0.0 : acc1[j]);
                });
                if constexpr (!(QS_S4_ABLATE & 4)) my_transit[ks * 64 + transit_wr] = stg[st ^ 1][ks];
                else ring[st ^ 1][ks] = stg[st ^ 1][ks];
                if constexpr (pf_next) stg[st ^ 1][ks] = fetch_frag(rs_pf, ks == N4 - 1 ? vf_end : vf_mid, KS);
                else stg[st ^ 1][ks] = fetch_frag(rs_in, ks == N4 - 1 ? vf_end : vf_mid, KS);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (NJM < NJ) {
                if (full) {     // the last column group of a wide chunk: both products, no memory operation in between
                    unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                        constexpr int ks = decltype(KS)::value;
                        acc1[NJ - 1] = mfma4(ring[st][ks], bf[ks][NJ - 1], ks == 0 ? 0.0 : acc1[NJ - 1]);
                    });
                    unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
                        constexpr int pg = decltype(PG)::value;
                        acc2[pg][NJ - 1] = mfma4(lf[pg], acc1[NJ - 1], ka == 0 ? 0.0 : acc2[pg][NJ - 1]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            QS_S4_STAMP(64 + ka)
            // ---- Out[:][own columns] += Lm[:][ka rows] . Y[ka rows][own columns]   (Y straight from the accumulators);
            // meanwhile row quad ka + 1 comes back from the transit buffer in MFMA lane order, the Lm fragments of
            // row quad ka + 1 are read, and in the last step the finished rows of Out go out
            unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
                constexpr int pg = decltype(PG)::value;
                unroll<0, NJM>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    acc2[pg][j] = mfma4(lf[pg], acc1[j], ka == 0 ? 0.0 : acc2[pg][j]);
                });
                if constexpr (!(QS_S4_ABLATE & 4)) ring[st ^ 1][pg] = my_transit[pg * 64 + transit_rd];
                if constexpr (ka + 1 < N4) {
                    if constexpr (QS_S4_ABLATE & 8) lf[pg] = 0.5;
                    else lf[pg] = ltab[(pg * N4 + ka + 1) * 16 + e_lane];
                }
                if constexpr (ka == N4 - 1 && pg >= 1) store_row(std::integral_constant<int, pg - 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        store_row(std::integral_constant<int, N4 - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        QS_S4_STAMP(255)
    };

    // chunk c: first column group and size
    auto chunk_first = [&](int c) __attribute__((always_inline)) {
        return c < N_BIG ? c * NJ_BIG : N_BIG * NJ_BIG + (c - N_BIG) * NJ_SMALL;
    };

    unsigned v_in[4], v_nx[4], v_out[4];
    auto rs_in = rsrc(g.in, iq < g.nquads ? iq : 0, g.in_item);
    in_offsets(iq, !idle && iq < g.nquads, v_in);
    // the first fetches go out before the tables are built: the two latencies overlap
    fetch_quad_row(rs_in, v_in, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    fetch_quad_row(rs_in, v_in, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    {   // every thread issues all its loads before the first LDS write (a load-store-load chain would pay the
        // memory latency ceil(l/4)^2/16 times before the first MFMA)
        constexpr int NF = (N4 * N4 * 16 + 255) / 256;
        double rv[NF], lv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + 256 * i;
            const int blk = f >> 4, e = f & 15, hi = blk / N4, lo = blk % N4, ez = e >> 2, ex = e & 3;
            const int k = 4 * hi + ez, j = 4 * lo + ex;
            const int p = 4 * hi + ex, a = 4 * lo + ez;
            const bool in = f < N4 * N4 * 16;
            rv[i] = (in && k < L && j < M) ? g.R[k * g.r_sk + j * g.r_sj] : 0.0;
            lv[i] = (in && p < M && a < L) ? g.Lm[p * g.l_sp + a * g.l_sa] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + 256 * i;
            if (f < N4 * N4 * 16) { rtab[f] = rv[i]; ltab[f] = lv[i]; }
        }
    }
    __syncthreads();
    if (idle) return;
    if constexpr (QS_S4_ABLATE & 32) return;
    QS_S4_STAMP(254)

    QS_S4_STAMP(254)
    settle_quad_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    fetch_quad_row(rs_in, v_in, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});

    unsigned parity = 0;      // ring stage of row quad 0 of the current unit (alternates when N4 is odd)
    for (unsigned round = 0; unit < u_end; ++round) {
        const unsigned nu = unit + slots;
        const unsigned nq = nu < u_end ? quad_of(nu) : g.nquads;
        const bool more = nq < g.nquads;
        auto rs_nx = rsrc(g.in, more ? nq : 0, g.in_item);
        in_offsets(nq, more, v_nx);
        auto rs_out = rsrc(g.out, iq < g.nquads ? iq : 0, g.out_item);
        out_offsets(iq, v_out);
        const int c = chunk_of(unit, round);
        const int jg0 = chunk_first(c);
        const bool big = c < N_BIG;
        if constexpr (N4 % 2 == 0) {
            chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 0>{}, jg0, big, rs_in, rs_out, v_out,
                  rs_nx, v_nx, v_in);
        } else {
            if (parity) chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 1>{}, jg0, big, rs_in,
                              rs_out, v_out, rs_nx, v_nx, v_in);
            else chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 0>{}, jg0, big, rs_in, rs_out,
                       v_out, rs_nx, v_nx, v_in);
            parity ^= 1;
        }
        unit = nu;
        iq = nq;
        rs_in = rs_nx;
#pragma unroll
        for (int i = 0; i < 4; ++i) v_in[i] = v_nx[i];
    }
    QS_S4_TRACE_DONE
}

template <int N4>
static int launch_sandwich4(const S4Args& g, hipStream_t stream) {
    const int n_cu = device_cu_count();
    int64_t wgs = n_cu - n_cu % 8;                       // one workgroup (four waves, one per SIMD) per CU
    if (wgs < 8) wgs = 8;
    const int64_t units = (g.mode & 1) ? ((int64_t)g.nquads + 3) / 4 * ((N4 + 3) / 4) : (int64_t)g.nquads;
    const int64_t gran = (g.mode & 1) ? 8 * ((N4 + 3) / 4) : 8;       // grouped: workgroups per XCD in whole groups
    wgs -= wgs % gran;
    if (wgs < gran) wgs = gran;
    const int64_t need = (units + gran - 1) / gran * gran;
    if (wgs > need) wgs = need;                          // short item lists: no idle workgroups
    const size_t lds = sizeof(double) * (2 * N4 * N4 * 16 + 4 * N4 * 64);
    static PerDeviceOnce lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)sandwich4_kernel<N4>, lds, lds_opt_in, "hipFuncSetAttribute(sandwich4)"))
        return rc;
    hipLaunchKernelGGL(sandwich4_kernel<N4>, dim3((unsigned)wgs), dim3(256), lds, stream, g);
    note_dispatch("qs::sandwich4_kernel<%d>", N4);
    return launch_status("sandwich4 launch");
}

// Out_t = Lm . In_t . R for t < nitems (strides in elements); QS_OK / error after launching, 1 = not eligible.
int sandwich4_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
                  int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
                  int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, hipStream_t stream) {
    if (dtype != QS_F64) return 1;
    if (L < 1 || M < 1 || L > 64 || M > 64) return 1;
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;                 // (near-)square products only
    if (nitems < 1024 || nitems >= (int64_t(1) << 31)) return 1;    // enough item quads to occupy the chip
    // Where this kernel measures faster than the 16-wide path (same-box sweep over l = 21 ... 64,
    // profiles/r02_small_basis.txt; g_tune.sandwich >= 4 overrides for tuning runs):
    //   * ceil(l/4) = 15 spills registers (two chunk bodies per parity x 15 x 4 accumulators) and loses;
    //   * contiguous items (the (d, c) pass) lose for ceil(l/4) in {8, 11, 12};
    //   * interleaved items (the (b, a) pass): when l is a multiple of 4 from 52 up, the 16 runs of a fetch are
    //     25 KB and 1.4 MB apart in whole 128-byte lines and pile up on a quarter of the L2 channels.
    if (g_tune.sandwich < 4) {
        if (n4 == 15 || n4 < 9) return 1;
        if (in_item != 1 && (n4 == 11 || n4 == 12)) return 1;
        if (in_item == 1 && (M % 4 == 0) && M >= 52) return 1;
    }
    // every byte offset inside an item quad stays below 2^31
    const int64_t in_span = (3 * in_item + (4 * n4) * (in_row > in_col ? in_row : in_col) * 2) * 8;
    const int64_t out_span = (3 * out_item + (4 * n4) * (out_row > out_col ? out_row : out_col) * 2) * 8;
    if (in_span >= (int64_t(1) << 31) || out_span >= (int64_t(1) << 31)) return 1;
    S4Args g;
    g.in = (const double*)in; g.out = (double*)out;
    g.R = (const double*)R; g.Lm = (const double*)Lm;
    g.r_sk = r_sk; g.r_sj = r_sj; g.l_sp = l_sp; g.l_sa = l_sa;
    g.in_item = in_item; g.in_row = in_row; g.in_col = in_col;
    g.out_item = out_item; g.out_row = out_row; g.out_col = out_col;
    g.L = (int)L; g.M = (int)M;
    g.nitems = (unsigned)nitems;
    g.nquads = (unsigned)cdiv(nitems, 4);
    // items 8 bytes apart (the (b, a) pass): four adjacent quads per workgroup make whole lines; contiguous items
    // (the (d, c) pass): a workgroup per quad keeps its four fetch streams on the same lines
    g.mode = g_tune.sandwich_mode >= 0 ? g_tune.sandwich_mode : (in_item == 1 ? 3 : 0);
    switch (n4) {
#ifdef QS_S4_ONLY          // development builds: one instantiation compiles in seconds
        case QS_S4_ONLY: return launch_sandwich4<QS_S4_ONLY>(g, stream);
#else
        case 6: return launch_sandwich4<6>(g, stream);
        case 7: return launch_sandwich4<7>(g, stream);
        case 8: return launch_sandwich4<8>(g, stream);
        case 9: return launch_sandwich4<9>(g, stream);
        case 10: return launch_sandwich4<10>(g, stream);
        case 11: return launch_sandwich4<11>(g, stream);
        case 12: return launch_sandwich4<12>(g, stream);
        case 13: return launch_sandwich4<13>(g, stream);
        case 14: return launch_sandwich4<14>(g, stream);
        case 15: return launch_sandwich4<15>(g, stream);
        case 16: return launch_sandwich4<16>(g, stream);
#endif
        default: return 1;
    }
}

}  // namespace qs

#ifdef QS_S4_TRACE
extern "C" int qs_s4_trace_reset(void) {
    unsigned zero = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qs::qs_s4_trace_n), &zero, sizeof(zero));
}
extern "C" int qs_s4_trace_read(void* dst) {      // dst: device buffer of 4097 x 8 bytes: count, stamps
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(qs::qs_s4_trace_n), sizeof(n));
    unsigned long long nn = n;
    (void)hipMemcpy(dst, &nn, 8, hipMemcpyHostToDevice);
    void* src = nullptr;
    (void)hipGetSymbolAddress(&src, HIP_SYMBOL(qs::qs_s4_trace));
    return (int)hipMemcpy((char*)dst + 8, src, 4096 * 8, hipMemcpyDeviceToDevice);
}
#endif
