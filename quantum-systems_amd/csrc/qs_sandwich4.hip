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
// (ceil(l/4) x 4), the accumulators of Y (ceil(l/4) x 4), a two-deep ring of In fragments fetched three row quads
// ahead -- across chunk and item boundaries -- and reads the fragments of Lm from an LDS table with compile-time
// offsets.
//
// What the instruction stream of a one-wave SIMD may contain between two fp64 MFMAs without costing matrix-pipe
// time (tools/probe_mfma4c.hip, cycles per MFMA with one extra instruction per three MFMAs: 16.5 alone):
//   * one LDS read, or two (16.6); one LDS write (16.8) -- but a write and a read behind the same MFMA 22.1;
//   * a vector memory instruction every OTHER group of three MFMAs (16.5; x2 or x4 dwords alike), every group 21.3:
//     the CU's address unit takes one 64-lane instruction per ~16 cycles from its four waves;
//   * s_waitcnt, SALU: free; s_nop n: n + 1 cycles; ANY VALU instruction (v_add_u32): 12 cycles.
// Hence: fragments are fetched two at a time (16 bytes per lane), every memory instruction has an MFMA of its
// own to hide behind (a sched_barrier after each), and the steady state has no VALU instruction: addresses are one
// lane offset (variants: interior / last row quad / last k pair / both, with out-of-range lanes parked past
// num_records so that the hardware returns 0.0 / drops the store) plus scalar offsets.
// Algorithmic bytes per launch: 8 (L^2 + M^2) per item; roofline: HBM below l ~ 64 (the matrix pipe needs
// 2 x 2 x ceil(l/4)^3 x 16 cycles per item quad per CU).

#include <type_traits>

#include "qs_common.h"
#include "qs_sandwich4.h"

// Development builds only (never defined in the shipped library): bit mask of parts to leave out, to find
// what bounds the kernel.  1 fetch, 2 stores, 16 all MFMAs, 32 return after the tables are built, 64 return at once,
// 128 every fetch of a wave from the same addresses (L1 hits)
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

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

// N4 = ceil(L / 4) = ceil(M / 4)
template <int N4>
__global__ __launch_bounds__(256, 1) void sandwich4_kernel(const S4Args g) {
    constexpr int NCH = (N4 + 3) / 4;                       // chunks of column groups, one per wave
    constexpr int NJ_BIG = (N4 + NCH - 1) / NCH;            // the first N4 % NCH chunks (all if it divides) have this many
    constexpr int NJ_SMALL = N4 / NCH;
    constexpr int N_BIG = (N4 % NCH) ? (N4 % NCH) : NCH;
    constexpr int NP = (N4 + 1) / 2;                        // fragment pairs of a row quad (one 16-byte fetch each)
    constexpr int NS = 2 * NP;                              // transit slots (the last one is a dummy when N4 is odd)
    constexpr int NST = 2;                                  // row quads in flight between their fetch and the transit buffer
    static_assert(NCH <= 4 && NJ_BIG <= 4, "a workgroup has four waves");
    static_assert(N4 > NST + 1 && NJ_SMALL >= 3, "the fetch runs NST + 1 row quads ahead; three MFMAs per k quad carry its instructions");

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* rtab = lds;                    // [ks][jg][16]: R[4 ks + z][4 jg + x]  at z * 4 + x
    double* ltab = lds + N4 * N4 * 16;     // [pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    double* transit = lds + 2 * N4 * N4 * 16;   // [wave][2][NS][64]: a row quad of In fragments on its way into MFMA
                                                // lane order, and as many words nobody reads
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

    const unsigned ka_step = (unsigned)(4 * g.in_row * 8), pair_step = (unsigned)(8 * g.in_col * 8);
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
    // version of this kernel (TA 73 % busy, matrix pipe 38 %).  A fetch takes the two fragments of a k-quad pair,
    // 16 bytes per lane along the index that is contiguous in memory:
    //   slab   (in_col == 1):  lane = h + 4 row + 16 item;  k = 8 m + 2 h, 2 h + 1          (64-byte runs)
    //   column (in_item == 1): lane = e + 2 k + 8 row + 32 f;  items 2 e, 2 e + 1 of fragment 2 m + f   (32-byte runs)
    const bool slab = g.in_col == 1;
    int f_row, f_par, f_item[2], f_k[2];     // this lane's two elements when it fetches: row, fragment of the pair, item, k
    if (slab) {
        const int h = lane & 3;
        f_row = (lane >> 2) & 3; f_par = h >> 1;
        f_item[0] = f_item[1] = lane >> 4;
        f_k[0] = 2 * (h & 1); f_k[1] = f_k[0] + 1;
    } else {
        const int e = lane & 1;
        f_row = (lane >> 3) & 3; f_par = lane >> 5;
        f_item[0] = 2 * e; f_item[1] = 2 * e + 1;
        f_k[0] = f_k[1] = (lane >> 1) & 3;
    }
    // Position of element (row, item, k) of a fragment in its 512-byte transit slot (slot parity par): 128-byte line
    // k, 8-byte bank pair (row ^ 2 (k & 1)) + 4 (item ^ ((k >> 1) | 2 par)).  Any 16 adjacent lanes that go to LDS
    // together -- (row, item) at fixed k when the MFMA lanes read, either of the two fetch layouts when the fetch
    // lanes write their first or their second element -- hit 16 different bank pairs (searched over the XOR masks
    // that are linear in (k, par)); the plain lane order would put four lanes of every write on each bank.
    auto slot_pos = [](int row, int item, int k, int par) __attribute__((always_inline)) {
        return (unsigned)(16 * k + ((row ^ ((k & 1) << 1)) + 4 * (item ^ ((k >> 1) | (par << 1)))));
    };
    double* const my_transit = transit + wave * (2 * NS * 64);
    const unsigned rd_even = slot_pos(x, y, z, 0), rd_odd = slot_pos(x, y, z, 1);
    // Lane offsets of a fetch (v: bit 0 last row quad, bit 1 last k pair) and where its two elements go in the transit
    // buffer (w[0], w[1]; w[2], w[3] for the last k pair).  A lane whose row / k / item does not exist is parked (the
    // hardware returns zeros).  A lane whose FIRST element exists and whose second does not -- the last k of an odd L
    // in a slab, the last item of an odd item count in a column -- fetches 8 bytes earlier (memory that always
    // exists), drops the first half into words nobody reads and puts the second half where the first belongs; the
    // place of the missing element keeps the zero the buffer starts with (k) or belongs to a block whose results are
    // never stored (item).  All offsets carry + 8 against a base pointer 8 bytes early, so that none is negative.
    auto in_offsets = [&](unsigned quad, bool live, unsigned (&v)[4], unsigned (&w)[4]) __attribute__((always_inline)) {
        const int64_t el = f_row * g.in_row + f_item[0] * g.in_item + (4 * f_par + f_k[0]) * g.in_col;
        const unsigned base = (unsigned)(el * 8) + 8;
        const bool row_ok = rl == 0 || f_row < rl;
        const int k_last = 8 * (NP - 1) + 4 * f_par;             // first k of this lane's fragment in the last pair
        const unsigned p0 = f_par * 64 + slot_pos(f_row, f_item[0], f_k[0], f_par);
        const unsigned p1 = f_par * 64 + slot_pos(f_row, f_item[1], f_k[1], f_par);
        const unsigned w_dump = NS * 64 + p0;                    // (the pair's offset is added to all of these)
        // (one set of unconditional assignments: stores into the arrays from both sides of a branch leave them in
        // scratch memory)
        const bool i0_ok = live && quad * 4 + f_item[0] < g.nitems, i1_ok = quad * 4 + f_item[1] < g.nitems;
        const bool e0_ok = k_last + f_k[0] < L, e1_ok = k_last + f_k[1] < L;
        const bool shift_k = slab && e0_ok && !e1_ok, shift_i = !slab && i0_ok && !i1_ok;
        const unsigned b_mid = i0_ok ? (shift_i ? base - 8 : base) : kParked;
        const unsigned b_last = i0_ok && e0_ok ? (shift_k || shift_i ? base - 8 : base) : kParked;
        v[0] = b_mid;
        v[1] = row_ok ? b_mid : kParked;
        v[2] = b_last;
        v[3] = row_ok ? b_last : kParked;
        w[0] = shift_i ? w_dump : p0;
        w[1] = shift_i ? p0 : p1;
        w[2] = shift_k || shift_i ? w_dump : p0;
        w[3] = shift_k || shift_i ? p0 : p1;
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
    // the same as fetched (memory lane order), on their way to the transit buffer.  (Two row quads in flight: a fetch
    // has two steps, ~1.1 us, to arrive, and the transit writes still wait ~110 cycles per step for the tensor.  Three
    // or four stages spill: the stage of a row quad is (row quad + phase) % NST, the phase advances by N4 per item
    // quad and every phase is one more copy of the chunk body.)
    double stg[NST][NS];

    // (s_row: the row quad's offset, made opaque once per step -- or the compiler keeps all ceil(l/4)^2 sums of row
    // and pair offsets in SGPRs and spills them)
    auto opaque = [](unsigned v) __attribute__((always_inline)) {
        asm volatile("" : "+s"(v));
        return v;
    };
    // the fragment pair m of a row quad: lane offset v, scalar offset row quad + pair
    auto fetch_pair = [&](auto rs, unsigned v, unsigned s_row, auto MP, double& d0, double& d1) __attribute__((always_inline)) {
        if constexpr (QS_S4_ABLATE & 1) { d0 = 1.0 + decltype(MP)::value; d1 = 0.5; }
        else {
            const unsigned s_off = (QS_S4_ABLATE & 128) ? 0u : s_row + decltype(MP)::value * pair_step;
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v, (int)s_off, 0);
            const f64x2 d = __builtin_bit_cast(f64x2, q);
            d0 = d.x; d1 = d.y;
        }
    };
    auto fetch_quad_row = [&](auto rs, const unsigned (&v)[4], auto KA, auto STAGE) __attribute__((always_inline)) {
        constexpr int ka = decltype(KA)::value, st = decltype(STAGE)::value;
        const unsigned s_row = opaque(ka * ka_step);
        unroll<0, NP>([&](auto MP) __attribute__((always_inline)) {
            constexpr int m = decltype(MP)::value;
            fetch_pair(rs, v[(ka == N4 - 1 ? 1 : 0) | (m == NP - 1 ? 2 : 0)], s_row, MP, stg[st][2 * m], stg[st][2 * m + 1]);
        });
    };
    // fetched pair -> transit buffer (each lane writes where the MFMA lanes that want its values will read) -> ring
    // stage, lane-linear.  One wave, one buffer: LDS executes a wave's accesses in order, no barrier.
    auto settle_quad_row = [&](auto FROM, auto TO, const unsigned (&w)[4]) __attribute__((always_inline)) {
        constexpr int from = decltype(FROM)::value, to = decltype(TO)::value;
        unroll<0, NP>([&](auto MP) __attribute__((always_inline)) {
            constexpr int m = decltype(MP)::value;
            my_transit[m * 128 + w[m == NP - 1 ? 2 : 0]] = stg[from][2 * m];
            my_transit[m * 128 + w[m == NP - 1 ? 3 : 1]] = stg[from][2 * m + 1];
        });
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            ring[to][ks] = my_transit[ks * 64 + ((ks & 1) ? rd_odd : rd_even)];
        });
    };

    // One chunk: NJ column groups starting at jg0 for the item quad behind (rs_in, v_in, w_in), in two phases:
    //   1. Y[ka] = In[ka] . R[:, own columns] for every row quad ka -- Y stays in the accumulators.  The ring already
    //      holds row quad 0; while row quad ka multiplies, row quad ka + 1 passes through the transit buffer
    //      and row quad ka + 1 + NST is fetched -- of this item quad, or (the last ones) of the next one, behind
    //      (rs_pf, v_pf, w_pf).  Q = stage of row quad 0 in the fetch registers (for an odd N4 also its ring stage).
    //   2. Out[pg] = sum_ka Lm[pg][ka] . Y[ka] for every row quad pg (Y is already in B-operand layout); a finished
    //      row quad leaves during the next one's MFMAs.
    // (The first version ran both products per ka and accumulated all of Out: every row of Out was finished in the
    // last step, all workgroups of the chip reached it together, and the 29 MB burst drained at the HBM write rate
    // -- 5-10k cycles per chunk with the matrix pipe idle.  Now the chip reads in phase 1 and writes in phase 2, both
    // at ~2.5 TB/s.)
    // ONE body serves chunks of NJ and of NJ - 1 column groups: the last group rides in blocks of its own that a
    // narrow chunk (full == false, wave-uniform) branches around -- two row quads per block, so that its two MFMA
    // chains alternate (a single chain waits 4 cycles per link).  Two bodies would leave the ring and the in-flight
    // fetch registers in different physical registers, and the copies at the merge cost a drain to vmcnt(0) plus
    // ~90 moves per unit.
    auto chunk = [&](auto NJC, auto PC, int jg0, bool full, auto rs_in, auto rs_out, const unsigned (&v_out)[4],
                     auto rs_pf, const unsigned (&v_pf)[4], const unsigned (&w_pf)[4], const unsigned (&v_in)[4],
                     const unsigned (&w_in)[4]) __attribute__((always_inline)) {
        constexpr int NJ = decltype(NJC)::value, Q = decltype(PC)::value, P = (N4 & 1) ? Q & 1 : 0;
        constexpr int NJM = (NJ_BIG == NJ_SMALL) ? NJ : NJ - 1;      // groups every chunk has
        constexpr int JX = NJ - 1;                                   // the group only a wide chunk has (if NJM < NJ)
        double bf[N4][NJ];
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                bf[decltype(KS)::value][decltype(J)::value] =
                    rtab[(decltype(KS)::value * N4 + jg0 + decltype(J)::value) * 16 + e_lane];
            });
        });
        double Y[N4][NJ];
        double lf[2][N4];       // fragments of Lm for row quads pg (stage pg & 1) and pg + 1
        double ov[2][NJ];       // Out fragments of row quads pg and pg - 1
        if constexpr (NJM < NJ) ov[0][JX] = ov[1][JX] = 0.0;
        // lane offsets of the chunk's column groups, for the interior row quads and for the last one: chosen once per
        // chunk (last column quad of the matrix / a group this chunk does not have)
        unsigned vo[2][NJ];
        unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const bool last_col = jg0 + j == N4 - 1;                  // wave-uniform
            const bool absent = j >= NJM && !full;
            vo[0][j] = absent ? kParked : v_out[last_col ? 2 : 0];
            vo[1][j] = absent ? kParked : v_out[last_col ? 3 : 1];
        });
        auto store_frag = [&](auto PG, auto J, unsigned s_row, const double (&src)[NJ]) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value, j = decltype(J)::value;
            const double val = src[j];
            if constexpr (QS_S4_ABLATE & 2) { if (val == 12345.678) my_transit[0] = val; }
            else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), rs_out, (int)vo[pg == N4 - 1][j],
                                                  (int)(s_row + (jg0 + j) * jg_step), 0);
        };
        // ---- phase 1.  Behind the MFMAs of an even k quad 2 m: the two transit writes of pair m (row quad ka + 1) and,
        // into the registers they free, the fetch of pair m of row quad ka + 1 + NST; behind those of the odd k quad: the two
        // ring reads of that pair (and, in the last step, the first fragments of Lm).
        unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
            constexpr int ka = decltype(KA)::value, st = (ka + P) & 1;
            QS_S4_STAMP(ka)
            if (step_barrier) __builtin_amdgcn_s_barrier();
            constexpr int sa = (ka + 1 + Q) % NST;             // fetch stage of row quad ka + 1, refilled with ka + 1 + NST
            constexpr bool pf_next = ka + 1 + NST >= N4;       // the fetch already belongs to the next item quad
            constexpr int ka_f = pf_next ? ka + 1 + NST - N4 : ka + 1 + NST;
            const unsigned (&vf)[4] = pf_next ? v_pf : v_in;
            const unsigned (&ws)[4] = ka + 1 >= N4 ? w_pf : w_in;      // row quad ka + 1 is the next item quad's first
            const unsigned s_row = opaque(ka_f * ka_step);
            if constexpr (NJM < NJ && ((ka & 1) || ka == N4 - 1)) {
                if (full) {     // ring[st ^ 1] still holds row quad ka - 1
                    unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                        constexpr int ks = decltype(KS)::value;
                        if constexpr (ka & 1)
                            Y[ka - 1][JX] = mfma4(ring[st ^ 1][ks], bf[ks][JX], ks == 0 ? 0.0 : Y[ka - 1][JX]);
                        Y[ka][JX] = mfma4(ring[st][ks], bf[ks][JX], ks == 0 ? 0.0 : Y[ka][JX]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value, m = ks / 2;
                unroll<0, NJM>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    Y[ka][j] = mfma4(ring[st][ks], bf[ks][j], ks == 0 ? 0.0 : Y[ka][j]);
                    if constexpr (!(ks & 1)) {
                        if constexpr (j == 0) my_transit[m * 128 + ws[m == NP - 1 ? 2 : 0]] = stg[sa][ks];
                        if constexpr (j == 1) my_transit[m * 128 + ws[m == NP - 1 ? 3 : 1]] = stg[sa][ks + 1];
                        if constexpr (j == 2) {
                            const unsigned v = vf[(ka_f == N4 - 1 ? 1 : 0) | (m == NP - 1 ? 2 : 0)];
                            if constexpr (pf_next) fetch_pair(rs_pf, v, s_row, std::integral_constant<int, m>{},
                                                              stg[sa][ks], stg[sa][ks + 1]);
                            else fetch_pair(rs_in, v, s_row, std::integral_constant<int, m>{}, stg[sa][ks],
                                            stg[sa][ks + 1]);
                        }
                    } else {
                        if constexpr (j == 0) ring[st ^ 1][ks - 1] = my_transit[(ks - 1) * 64 + rd_even];
                        if constexpr (j == 1) ring[st ^ 1][ks] = my_transit[ks * 64 + rd_odd];
                        if constexpr (j == 2 && ka == N4 - 1) {
                            lf[0][ks - 1] = ltab[(ks - 1) * 16 + e_lane];
                            lf[0][ks] = ltab[ks * 16 + e_lane];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if constexpr (N4 & 1) {      // the last pair has one fragment: no odd k quad behind it
                ring[st ^ 1][N4 - 1] = my_transit[(N4 - 1) * 64 + rd_even];
                if constexpr (ka == N4 - 1) lf[0][N4 - 1] = ltab[(N4 - 1) * 16 + e_lane];
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // ---- phase 2.  Behind the MFMAs of row quad ka of Y: the fragment of Lm for (pg + 1, ka); the stores of row quad
        // pg - 1 go behind every other one (no more than one vector memory instruction per two groups of MFMAs).
        unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value, b = pg & 1;
            QS_S4_STAMP(64 + pg)
            const unsigned s_prev = opaque((pg ? pg - 1 : 0) * pg_step);
            if constexpr (NJM < NJ && ((pg & 1) || pg == N4 - 1)) {
                if (full) {     // lf[b ^ 1] still holds the fragments of row quad pg - 1
                    unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
                        constexpr int ka = decltype(KA)::value;
                        if constexpr (pg & 1)
                            ov[b ^ 1][JX] = mfma4(lf[b ^ 1][ka], Y[ka][JX], ka == 0 ? 0.0 : ov[b ^ 1][JX]);
                        ov[b][JX] = mfma4(lf[b][ka], Y[ka][JX], ka == 0 ? 0.0 : ov[b][JX]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
                constexpr int ka = decltype(KA)::value;
                unroll<0, NJM>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    ov[b][j] = mfma4(lf[b][ka], Y[ka][j], ka == 0 ? 0.0 : ov[b][j]);
                    if constexpr (j == 0 && pg + 1 < N4) lf[b ^ 1][ka] = ltab[((pg + 1) * N4 + ka) * 16 + e_lane];
                    if constexpr (j == 1 && pg >= 1 && (ka & 1) && ka / 2 < NJ)
                        store_frag(std::integral_constant<int, pg - 1>{}, std::integral_constant<int, (ka / 2) % NJ>{},
                                   s_prev, ov[b ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        {
            const unsigned s_row = opaque((N4 - 1) * pg_step);
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                store_frag(std::integral_constant<int, N4 - 1>{}, J, s_row, ov[(N4 - 1) & 1]);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        QS_S4_STAMP(255)
    };

    // chunk c: first column group and size
    auto chunk_first = [&](int c) __attribute__((always_inline)) {
        return c < N_BIG ? c * NJ_BIG : N_BIG * NJ_BIG + (c - N_BIG) * NJ_SMALL;
    };

    unsigned v_in[4], v_nx[4], w_in[4], w_nx[4], v_out[4];
    auto rs_in = rsrc(g.in - 1, iq < g.nquads ? iq : 0, g.in_item);
    in_offsets(iq, !idle && iq < g.nquads, v_in, w_in);
    {   // every thread issues all its loads before the first LDS write (a load-store-load chain would pay the
        // memory latency ceil(l/4)^2/16 times before the first MFMA).  Thread t takes element t & 15 of the 4 x 4
        // blocks t / 16, t / 16 + 16, ...: the block coordinates advance by additions (the first version spent
        // ~3.5k cycles here on 64-bit multiply-adds and divisions by N4), and a block or element that does not exist
        // is a parked buffer offset that reads 0.0.
        constexpr int NF = (N4 * N4 + 15) / 16;
        const auto rs_R = rsrc(g.R, 0, 0), rs_L = rsrc(g.Lm, 0, 0);
        const int ez = (tid >> 2) & 3, ex = tid & 3, b0 = tid >> 4;        // b0 < 16 <= 3 N4
        int hi = (b0 >= N4) + (b0 >= 2 * N4), lo = b0 - hi * N4;
        const int r_sk = (int)g.r_sk, r_sj = (int)g.r_sj, l_sp = (int)g.l_sp, l_sa = (int)g.l_sa;
        int off_r = (4 * hi + ez) * r_sk + (4 * lo + ex) * r_sj;           // R[4 hi + ez][4 lo + ex]
        int off_l = (4 * hi + ex) * l_sp + (4 * lo + ez) * l_sa;           // Lm[4 hi + ex][4 lo + ez]
        double rv[NF], lv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const bool in = hi < N4;
            const bool r_ok = in && 4 * hi + ez < L && 4 * lo + ex < M, l_ok = in && 4 * hi + ex < M && 4 * lo + ez < L;
            rv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_R, r_ok ? off_r * 8 : (int)kParked, 0, 0));
            lv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_L, l_ok ? off_l * 8 : (int)kParked, 0, 0));
            hi += 16 / N4; lo += 16 % N4;
            off_r += (16 / N4) * 4 * r_sk + (16 % N4) * 4 * r_sj;
            off_l += (16 / N4) * 4 * l_sp + (16 % N4) * 4 * l_sa;
            const bool wrap = lo >= N4;
            lo -= wrap ? N4 : 0; hi += wrap;
            off_r += wrap ? 4 * r_sk - N4 * 4 * r_sj : 0;
            off_l += wrap ? 4 * l_sp - N4 * 4 * l_sa : 0;
        }
        // the first fetches go out before the tables are built: the two latencies overlap.  They follow the table
        // loads (which every workgroup finds in L2): loads return in order, and the first touch of the tensor -- all
        // workgroups at once -- would otherwise stand in front of the tables
        __builtin_amdgcn_sched_barrier(0);
        fetch_quad_row(rs_in, v_in, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        fetch_quad_row(rs_in, v_in, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // the slots of the last k pair start as zeros (the place of a k that does not exist is never written)
        for (int i = tid; i < 4 * 128; i += 256) transit[(i >> 7) * (2 * NS * 64) + (NS - 2) * 64 + (i & 127)] = 0.0;
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
    settle_quad_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, w_in);
    unroll<2, NST + 1>([&](auto R) __attribute__((always_inline)) {
        fetch_quad_row(rs_in, v_in, R, std::integral_constant<int, decltype(R)::value % NST>{});
    });

    unsigned phase = 0;       // fetch stage of row quad 0 of the current unit
    for (unsigned round = 0; unit < u_end; ++round) {
        const unsigned nu = unit + slots;
        const unsigned nq = nu < u_end ? quad_of(nu) : g.nquads;
        const bool more = nq < g.nquads;
        auto rs_nx = rsrc(g.in - 1, more ? nq : 0, g.in_item);
        in_offsets(nq, more, v_nx, w_nx);
        auto rs_out = rsrc(g.out, iq < g.nquads ? iq : 0, g.out_item);
        out_offsets(iq, v_out);
        const int c = chunk_of(unit, round);
        const int jg0 = chunk_first(c);
        const bool big = c < N_BIG;
        auto run = [&](auto QC) __attribute__((always_inline)) {
            chunk(std::integral_constant<int, NJ_BIG>{}, QC, jg0, big, rs_in, rs_out, v_out, rs_nx, v_nx, w_nx, v_in, w_in);
        };
        if constexpr (N4 % NST == 0) run(std::integral_constant<int, 0>{});
        else {
            if (phase == 0) run(std::integral_constant<int, 0>{});
            else if (phase == 1 || NST == 2) run(std::integral_constant<int, 1>{});
            else run(std::integral_constant<int, 2 % NST>{});
            phase = (phase + N4) % NST;
        }
        unit = nu;
        iq = nq;
        rs_in = rs_nx;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v_in[i] = v_nx[i]; w_in[i] = w_nx[i]; }
    }
    QS_S4_TRACE_DONE
}

template <int N4>
static int launch_sandwich4(const S4Args& g, hipStream_t stream, int dry_run) {
    if (dry_run) return QS_OK;
    const int n_cu = device_cu_count();
    int64_t wgs = n_cu - n_cu % 8;                       // one workgroup (four waves, one per SIMD) per CU
    if (wgs < 8) wgs = 8;
    const int64_t units = (g.mode & 1) ? ((int64_t)g.nquads + 3) / 4 * ((N4 + 3) / 4) : (int64_t)g.nquads;
    const int64_t gran = (g.mode & 1) ? 8 * ((N4 + 3) / 4) : 8;       // grouped: workgroups per XCD in whole groups
    wgs -= wgs % gran;
    if (wgs < gran) wgs = gran;
    const int64_t need = (units + gran - 1) / gran * gran;
    if (wgs > need) wgs = need;                          // short item lists: no idle workgroups
    const size_t lds = sizeof(double) * (2 * N4 * N4 * 16 + 4 * 2 * (2 * ((N4 + 1) / 2)) * 64);
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)sandwich4_kernel<N4>, lds, lds_opt_in, "hipFuncSetAttribute(sandwich4)"))
        return rc;
    hipLaunchKernelGGL(sandwich4_kernel<N4>, dim3((unsigned)wgs), dim3(256), lds, stream, g);
    note_dispatch("qs::sandwich4_kernel<%d>", N4);
    return launch_status("sandwich4 launch");
}

// Out_t = Lm . In_t . R for t < nitems (strides in elements); QS_OK / error after launching, 1 = not eligible.
// With `dry_run` nothing is launched: QS_OK = this call would launch (the ONE eligibility rule, asked by
// qs_transform_two_body before it commits the intermediate to the layout of the second fused pass).
int sandwich4_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
                  int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
                  int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, hipStream_t stream, int dry_run) {
    if (dtype != QS_F64) return 1;
    if (L < 1 || M < 1 || L > 64 || M > 64) return 1;
    const int n4 = (int)cdiv(L, 4);
    if (n4 != (int)cdiv(M, 4)) return 1;                 // (near-)square products only
    if (nitems < 1024 || nitems >= (int64_t(1) << 31)) return 1;    // enough item quads to occupy the chip
    if (in_col != 1 && in_item != 1) return 1;           // the fetch takes 16 contiguous bytes per lane along k or the item
    // Slabs in: the balanced form with a cooperative fetch (qs_sandwich4b.hip) where it exists and measured faster
    // (same-box sweep, profiles/r02_small_basis_sweep.txt: 2-7 % for ceil(l/4) in {10, 14, 16} except l = 40 and 56,
    // and ceil(l/4) = 12 -- where the four-chunk split of this file leaves a SIMD idle -- wins its first pass back).
    // Its instantiation for N4 also takes ceil(l/4) = N4 - 1 (the last two quads may be incomplete or empty): 15, which
    // spills here, runs on the one for 16 (l = 57 ... 60: 150-185 -> 130-140 us), 11 on the one for 12 (44 -> 38 us);
    // 9 and 13 measured slower that way (g_tune.sandwich_v2 == 2: every odd one on the next even one).
    int n4k = n4;
    const unsigned tail_quads = sandwich4b_tail_quads((unsigned)cdiv(nitems, 4), (int)M);
    bool v2 = in_col == 1 && (g_tune.sandwich_v2 > 0 ||
                              (g_tune.sandwich_v2 < 0 && (n4 == 10 || n4 == 12 || n4 == 14 || n4 == 16) &&
                               (!(L % 4 == 0 && (n4 == 10 || n4 == 14)) ||
                                tail_quads)));                                           // (l = 56: its last round is split there)
    // (9: only for the sake of the tail -- 35 and 36 orbitals are 1.2 and 1.27 rounds of item quads: l = 35 31.1 -> 28.6 us with the
    // first pass there, l = 36 37.0 -> 32.6 with both; 33 and 34, 17 and 33 quads over, are not worth the wider instantiation)
    if (in_col == 1 && (n4 & 1) && n4 >= 9 && (g_tune.sandwich_v2 == 2 || (g_tune.sandwich_v2 < 0 && (n4 == 11 || n4 == 15 ||
                                                                                                       (n4 == 9 && tail_quads >= 48))))) {
        v2 = true;
        n4k = n4 + 1;
    }
    // Where these kernels measure faster than the 16-wide path (same sweep; g_tune.sandwich >= 4 overrides for tuning
    // runs):
    //   * ceil(l/4) = 15 spills registers here (two chunk bodies for the parity of the ring x 15 x 4 accumulators):
    //     only the balanced form takes it;
    //   * below ceil(l/4) = 9 the transform is launch-bound either way;
    //   * slabs in, slabs out (the (d, c) pass with T2 in its natural layout) lose for ceil(l/4) = 11, and for 12 unless
    //     the balanced form runs;
    //   * slabs in, interleaved items out (either pass when T2 is stored transposed) lose for l = 48 (round 3, with the
    //     balanced kernel's tail: l = 45 50.6 -> 44.2 us, 46 55.5 -> 54.2, 47 58.6 -> 57.5, 48 60.5 -> 63.3).
    if (g_tune.sandwich < 4) {
        if (n4 < 9 || (n4 == 15 && !v2)) return 1;
        if (in_item != 1 && out_item != 1 && (n4 == 11 || (n4 == 12 && !v2))) return 1;
        if (in_item != 1 && out_item == 1 && n4 == 12 && L % 4 == 0) return 1;     // (45 ... 47: see qs_api.hip)
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
    g.tail_first = g.nquads; g.tail_parts = 0;
    // items 8 bytes apart (the (b, a) pass): four adjacent quads per workgroup make whole lines; contiguous items
    // (the (d, c) pass): a workgroup per quad keeps its four fetch streams on the same lines
    g.mode = g_tune.sandwich_mode >= 0 ? g_tune.sandwich_mode : (in_item == 1 ? 3 : 0);
    if (v2) {
        const int rc = sandwich4b_launch(g, n4k, stream, dry_run);
        if (rc != 1) return rc;
    }
    switch (n4) {
#ifdef QS_S4_ONLY          // development builds: one instantiation compiles in seconds
        case QS_S4_ONLY: return launch_sandwich4<QS_S4_ONLY>(g, stream, dry_run);
#else
        case 6: return launch_sandwich4<6>(g, stream, dry_run);
        case 7: return launch_sandwich4<7>(g, stream, dry_run);
        case 8: return launch_sandwich4<8>(g, stream, dry_run);
        case 9: return launch_sandwich4<9>(g, stream, dry_run);
        case 10: return launch_sandwich4<10>(g, stream, dry_run);
        case 11: return launch_sandwich4<11>(g, stream, dry_run);
        case 12: return launch_sandwich4<12>(g, stream, dry_run);
        case 13: return launch_sandwich4<13>(g, stream, dry_run);
        case 14: return launch_sandwich4<14>(g, stream, dry_run);
        case 15: return launch_sandwich4<15>(g, stream, dry_run);
        case 16: return launch_sandwich4<16>(g, stream, dry_run);
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
