// Out_t = Lm . In_t . R for item quads of slabs (see qs_sandwich4.hip for the product, the 4-wide fp64 matrix
// instruction, the lane layouts and what may ride between two MFMAs): the BALANCED form with a COOPERATIVE fetch.
//
// In qs_sandwich4.hip the four waves of a workgroup take column chunks of 4, 4, 3, 3 groups of one item quad
// (ceil(l/4) = 14) and every wave fetches every In fragment for itself: 28 vector memory instructions per step and
// CU keep the address unit busy for ~1040 cycles against 693-924 cycles of MFMAs, and the narrow waves wait for
// the wide ones at the end of the launch.  Here
//   * every wave takes floor(N4 / 4) column groups of its own; when N4 = 4 q + 2 the two groups left over belong to
//     the wave pairs (0, 1) and (2, 3): of such a group a wave computes the row quads of ITS parity, in both
//     products, and the pair exchanges its halves of Y through LDS in between -- 3.5 groups per wave, the same
//     instruction stream in all four;
//   * waves that run in step can share: a row quad of In is fetched ONCE per workgroup (each wave a quarter of its
//     fragment pairs), goes into one transit buffer, and all four waves read their A operands from it -- a quarter
//     of the vector memory instructions and of the LDS writes, one workgroup barrier per TWO steps (in every step two
//     of the four waves carry the extra chain of their shared group: only pairs of steps weigh the same);
//   * with two fragment pairs per wave in flight instead of seven, the fetch registers are a sixth of the old ones.
// Per-element sums are the same k-ordered FMA chains: results are bit-identical to the other kernels'.

#include <type_traits>

#include "qs_common.h"

// cache policy of the stores of the result (development: -DQS_S4B_STORE_AUX=n; 0 = default, 2 = non-temporal, 1 / 16 / 17 = sc0 / sc1 / both)
#ifndef QS_S4B_STORE_AUX
#define QS_S4B_STORE_AUX 0
#endif
#include "qs_sandwich4.h"

// development: -DQS_S4B_EXPERIMENT=1 nobody runs the shared group's chains (every step branches around them),
// =2 the chains are compiled out (timing experiments; results are wrong)
#ifndef QS_S4B_EXPERIMENT
#define QS_S4B_OWNER(par) ((par) == mpar)
#elif QS_S4B_EXPERIMENT == 1
#define QS_S4B_OWNER(par) (mpar == 2 + (par))
#else
#define QS_S4B_OWNER(par) false
#endif

namespace qs {

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void unroll_b(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll_b<I + 1, N>(f);
    }
}

typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef double f64x2b __attribute__((ext_vector_type(2)));

constexpr unsigned kParkedB = 0x80000000u;     // lane offset of an out-of-range lane (>= num_records)

#ifdef QS_S4_TRACE      // development: shader-clock stamps of one workgroup's wave 0 (tools/trace_s4.py)
__device__ unsigned long long qs_s4b_trace[4096];
__device__ unsigned qs_s4b_trace_n;
#define QS_S4B_STAMP(tag)                                                                                 \
    if (blockIdx.x == QS_S4_TRACE && wave == 0) {                                                         \
        if (lane == 0 && tr_n < 4096) qs_s4b_trace[tr_n] = (__builtin_amdgcn_s_memtime() << 8) | (tag);   \
        ++tr_n;                                                                                           \
    }
#define QS_S4B_TRACE_DONE                                                                    \
    if (blockIdx.x == QS_S4_TRACE && wave == 0 && lane == 0) {                                \
        qs_s4b_trace_n = tr_n;                                                               \
        qs_s4b_trace[4095] = __builtin_amdgcn_s_memrealtime() - tr_real0;                    \
    }
#else
#define QS_S4B_STAMP(tag)
#define QS_S4B_TRACE_DONE
#endif

__device__ __forceinline__ double mfma4b(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads would also drain the fetches in flight)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}

}  // namespace

// 4 (N4 - 2) < L, M <= 4 N4 (the last TWO row / column quads may be incomplete or empty: ceil(l/4) = N4 - 1 runs here
// too); N4 even, N4 % 4 in {0, 2}
template <int N4>
__global__ __launch_bounds__(256, 1) void sandwich4b_kernel(const S4Args g) {
    constexpr int Q = N4 / 4;                 // column groups of a wave's own
    constexpr int E = N4 % 4;                 // groups shared by the wave pairs: 0, or 2 (one per pair)
    constexpr int NP = N4 / 2;                // fragment pairs of a row quad (one 16-byte fetch each)
    constexpr int PW = (NP + 3) / 4;          // pairs a wave fetches: pair numbers wave, wave + 4, ...
    static_assert(N4 % 2 == 0 && (E == 0 || E == 2) && Q >= 2 && Q <= 4 && N4 >= 8, "see sandwich4b_launch");
    constexpr int SET = N4 * 64;              // doubles of a row quad in the transit buffer

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* rtab = lds;                    // [ks][jg][16]: R[4 ks + z][4 jg + x]  at z * 4 + x
    double* ltab = lds + N4 * N4 * 16;     // [pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    double* transit = lds + 2 * N4 * N4 * 16;   // [8][N4][64]: four row quads of In on their way into MFMA lane order
                                                // (set = row quad counted over the item quads, mod 4) and, behind them,
                                                // as many words nobody reads
    double* xch = transit + 8 * SET;            // [2][N4][64]: the halves of Y of the shared groups, per wave pair
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, mpar = wave & 1;
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int e_lane = z * 4 + x;
    unsigned tr_n = 0;
    (void)tr_n;
#ifdef QS_S4_TRACE
    const unsigned long long tr_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    QS_S4B_STAMP(253)

    // ---- work units = item quads.  Every XCD takes a contiguous range and neighbouring workgroups of an XCD take
    // neighbouring quads.
    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = blockIdx.x / n_xcd, slots = gridDim.x / n_xcd;
    const unsigned nmain = g.tail_parts ? g.tail_first : g.nquads;      // (with a tail: a whole number of rounds of the grid)
    const unsigned per = (nmain + n_xcd - 1) / n_xcd;
    const unsigned u_end = (xcd + 1) * per < nmain ? (xcd + 1) * per : nmain;
    unsigned iq = xcd * per + slot;
    if (iq >= u_end) return;                           // (whole workgroup: before any barrier)

    const unsigned ka_step = (unsigned)(4 * g.in_row * 8), pair_step = 64u;      // (in_col == 1: a pair is 8 k)
    const unsigned pg_step = (unsigned)(4 * g.out_row * 8), jg_step = (unsigned)(4 * g.out_col * 8);

    auto rsrc = [&](const double* base, unsigned quad, int64_t item_stride) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base + (int64_t)quad * 4 * item_stride);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        // (development, round 4: -DQS_S4B_ABLATE_LOADS / _STORES give the tensor's descriptors a zero range -- loads answer with
        // zeros and stores are dropped without touching memory: the kernel's skeleton, profiles/r04_small_basis_bound.txt)
        int range = 0x7fffffff;
#ifdef QS_S4B_ABLATE_LOADS
        if (base == g.in - 1) range = 0;
#endif
#ifdef QS_S4B_ABLATE_STORES
        if (base == g.out) range = 0;
#endif
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, range, 0x00020000);
    };
    // Fetch layout of a fragment pair (qs_sandwich4.hip): lane = h + 4 row + 16 item; k = 8 m + 2 h, 2 h + 1 -- 16 bytes
    // per lane, 64-byte runs; fragment 2 m + (h >> 1).
    const int f_h = lane & 3, f_row = (lane >> 2) & 3, f_item = lane >> 4, f_par = f_h >> 1, f_k0 = 2 * (f_h & 1);
    auto slot_pos = [](int row, int item, int k, int par) __attribute__((always_inline)) {
        return (unsigned)(16 * k + ((row ^ ((k & 1) << 1)) + 4 * (item ^ ((k >> 1) | (par << 1)))));
    };
    const unsigned rd_even = slot_pos(x, y, z, 0), rd_odd = slot_pos(x, y, z, 1);
    // Per item quad and per pair i of this wave (pair number wave + 4 i): the lane offset of its fetch for interior
    // row quads and for the last two (v[i][0 / 1 / 2]) and where its two elements go in the transit buffer (w[i][0 / 1][c],
    // in doubles; c = bit 1 of the row quad).  A lane whose row / k / item does not exist is parked (the hardware
    // returns zeros); so is every lane of a pair that does not exist.  The last k of an odd L: see qs_sandwich4.hip (the
    // lane fetches 8 bytes earlier, its first half goes to words nobody reads, the missing element's place stays zero).
    // Transit set of row quad r of an item quad: 2 (((r >> 1) & 1) ^ ph) + (r & 1), ph = the parity of the item quad's
    // first pair of row quads in the workgroup's sequence (N4 / 2 may be odd); the (r & 1) part is an immediate, the
    // rest sits in the address registers: w, and rd[parity of the k quad][c] for the reads.
    auto in_offsets = [&](unsigned quad, bool live, unsigned ph, unsigned (&v)[PW][3], unsigned (&w)[PW][2][2],
                          unsigned (&rd)[2][2]) __attribute__((always_inline)) {
        const int64_t el = f_row * g.in_row + f_item * g.in_item + (4 * f_par + f_k0);
        const unsigned base = (unsigned)(el * 8) + 8;
        const bool item_ok = live && quad * 4 + f_item < g.nitems;
        const unsigned p0 = f_par * 64 + slot_pos(f_row, f_item, f_k0, f_par);
        const unsigned p1 = f_par * 64 + slot_pos(f_row, f_item, f_k0 + 1, f_par);
        const int k_last = 8 * (NP - 1) + 4 * f_par + f_k0;      // this lane's first k in the last pair
        const bool e0_ok = k_last < L, e1_ok = k_last + 1 < L;
        const bool shift = e0_ok && !e1_ok;
        const unsigned set_c[2] = {2 * SET * ph, 2 * SET * (ph ^ 1)};
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int m = wave + 4 * i;                            // (wave-uniform)
            const bool exists = m < NP, last = m == NP - 1;
            const unsigned b = !exists || !item_ok ? kParkedB
                               : !last ? base
                               : e0_ok ? (shift ? base - 8 : base) : kParkedB;
            v[i][0] = b;                                           // row quads below N4 - 2: every row exists
            v[i][1] = 4 * (N4 - 2) + f_row < L ? b : kParkedB;     // row quad N4 - 2
            v[i][2] = 4 * (N4 - 1) + f_row < L ? b : kParkedB;     // row quad N4 - 1
            const unsigned at = (exists ? m : 0) * 128;
            const bool divert = !exists || (last && shift);
            const unsigned w0 = at + (divert ? 4 * SET + p0 : p0);
            const unsigned w1 = at + (!exists ? 4 * SET + p1 : (last && shift) ? p0 : p1);
#pragma unroll
            for (int c = 0; c < 2; ++c) { w[i][0][c] = w0 + set_c[c]; w[i][1][c] = w1 + set_c[c]; }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { rd[0][c] = rd_even + set_c[c]; rd[1][c] = rd_odd + set_c[c]; }
    };
    // lane offsets of an Out fragment (D: row z, block y, column x) of row quads below N4 - 2 / of row quad N4 - 2 / of
    // row quad N4 - 1 (a row or item that does not exist is parked; the column is checked per group, see vo)
    auto out_offsets = [&](unsigned quad, unsigned (&v)[3]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)((z * g.out_row + y * g.out_item + x * g.out_col) * 8);
        const bool item_ok = quad * 4 + y < g.nitems;
        v[0] = item_ok ? base : kParkedB;
        v[1] = item_ok && 4 * (N4 - 2) + z < M ? base : kParkedB;
        v[2] = item_ok && 4 * (N4 - 1) + z < M ? base : kParkedB;
    };

    double ring[2][N4];         // In fragments of two row quads in MFMA lane order: parity of the row quad, k quad
    double stg[2][PW][2];       // this wave's pairs of two row quads as fetched: parity of the row quad, pair, element

    auto opaque = [](unsigned v) __attribute__((always_inline)) {
        asm volatile("" : "+s"(v));
        return v;
    };
    // Row quad r of an item quad (r a compile-time number, the quad behind rs / v / w / rd):
    //   load  -- this wave's pairs into the fetch registers of r's parity;
    //   write -- from there into r's transit set (the registers are free for row quad r + 2 afterwards);
    //   read  -- all of its fragments from that set into ring stage r & 1 (after a barrier behind everybody's writes).
    const unsigned pair0 = opaque((unsigned)wave * pair_step);
    auto load_into = [&](auto rs, const unsigned (&v)[PW][3], auto R_, auto I_, double (&dst)[2]) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value, i = decltype(I_)::value;
        const unsigned s_off = opaque(r * ka_step) + pair0 + (unsigned)(4 * i) * pair_step;
        const u32x4b q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v[i][r == N4 - 1 ? 2 : r == N4 - 2 ? 1 : 0], (int)s_off, 0);
        const f64x2b d = __builtin_bit_cast(f64x2b, q);
        dst[0] = d.x;
        dst[1] = d.y;
    };
    auto load_pair = [&](auto rs, const unsigned (&v)[PW][3], auto R_, auto I_) __attribute__((always_inline)) {
        load_into(rs, v, R_, I_, stg[decltype(R_)::value & 1][decltype(I_)::value]);
    };
    auto write_from = [&](const unsigned (&w)[PW][2][2], auto R_, auto I_, auto H_, double val) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value, i = decltype(I_)::value, h = decltype(H_)::value;
        transit[(r & 1) * SET + w[i][h][(r >> 1) & 1]] = val;
    };
    auto write_elem = [&](const unsigned (&w)[PW][2][2], auto R_, auto I_, auto H_) __attribute__((always_inline)) {
        write_from(w, R_, I_, H_, stg[decltype(R_)::value & 1][decltype(I_)::value][decltype(H_)::value]);
    };
    auto read_frag = [&](const unsigned (&rd)[2][2], auto R_, auto KS) __attribute__((always_inline)) {
        constexpr int r = decltype(R_)::value, ks = decltype(KS)::value;
        ring[r & 1][ks] = transit[(r & 1) * SET + ks * 64 + rd[ks & 1][(r >> 1) & 1]];
    };

    // The fragments of R for this wave's groups: the same for every item quad (a wave keeps its groups), read once
    // after the tables are built.
    const int jg0 = Q * wave;                 // first own group (wave-uniform)
    const int jx = 4 * Q + pair;              // the pair's shared group (E == 2)
    double bf[N4][Q];
    double bx[E ? N4 : 1];
    auto load_r_fragments = [&]() __attribute__((always_inline)) {
        unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
            unroll_b<0, Q>([&](auto J) __attribute__((always_inline)) {
                bf[decltype(KS)::value][decltype(J)::value] =
                    rtab[(decltype(KS)::value * N4 + jg0 + decltype(J)::value) * 16 + e_lane];
            });
        });
        if constexpr (E != 0) {
            unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
                bx[decltype(KS)::value] = rtab[(decltype(KS)::value * N4 + jx) * 16 + e_lane];
            });
        }
    };

    // One item quad, in two phases (qs_sandwich4.hip): Y[ka] = In[ka] . R[:, own groups], then Out[pg] = sum_ka
    // Lm[pg][ka] . Y[ka].  Step ka of phase 1: (a barrier before every even one;) behind the MFMAs of row quad ka the
    // fragments of row quad ka + 1 are READ from the transit buffer, this wave's pairs of row quad ka + 3 are WRITTEN
    // there and those of row quad ka + 5 are LOADED into the registers just freed -- row quads beyond the last are the
    // next item quad's (behind rs_nx, v_nx, w_nx, rd_nx).  Between two barriers the waves read row quads 2 D + 1 and
    // 2 D + 2 and write 2 D + 3 and 2 D + 4: four sets.  Every memory instruction behind an MFMA of its own.
    auto quad_pass = [&](auto rs_in, auto rs_out, const unsigned (&v_out)[3], auto rs_nx, const unsigned (&v_nx)[PW][3],
                         const unsigned (&w_nx)[PW][2][2], const unsigned (&rd_nx)[2][2], const unsigned (&v_in)[PW][3],
                         const unsigned (&w_in)[PW][2][2], const unsigned (&rd_in)[2][2]) __attribute__((always_inline)) {
        double Y[N4][Q];
        double Yh[E ? N4 / 2 : 1];       // this wave's half of Y of the shared group: row quads 2 i + mpar
        double lf[2][N4];                // fragments of Lm for row quads pg (stage pg & 1) and pg + 1
        double ov[2][Q];                 // Out fragments of row quads pg and pg - 1
        double ovx[2] = {0.0, 0.0};      // those of the shared group (parity of the row quad; only mpar's is computed here)
        // lane offsets of the groups' stores for the three kinds of row quad (a column that does not exist is parked)
        unsigned vo[3][Q], vox[2][3];
        unroll_b<0, Q>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const bool col_ok = 4 * (jg0 + j) + x < M;
#pragma unroll
            for (int rv = 0; rv < 3; ++rv) vo[rv][j] = col_ok ? v_out[rv] : kParkedB;
        });
        if constexpr (E != 0) {
            const bool col_ok = 4 * jx + x < M;
#pragma unroll
            for (int par = 0; par < 2; ++par)                         // rows of the other parity are the partner's
#pragma unroll
                for (int rv = 0; rv < 3; ++rv) vox[par][rv] = (par == mpar && col_ok) ? v_out[rv] : kParkedB;
        }
        auto store_frag = [&](unsigned vofs, unsigned s_off, double val) __attribute__((always_inline)) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, val), rs_out, (int)vofs, (int)s_off, QS_S4B_STORE_AUX);
        };

        // ---- phase 1
        unroll_b<0, N4>([&](auto KA) __attribute__((always_inline)) {
            constexpr int ka = decltype(KA)::value, st = ka & 1;
            QS_S4B_STAMP(ka)
            // An even step carries the barrier: behind its first two k quads, and all of the step's LDS traffic behind
            // that -- the wait for this wave's own LDS instructions in front of the barrier then finds them done (at
            // the very start of the step the last ring reads of the step before are still on their way).
            constexpr int SH = (ka & 1) ? 0 : 2;
            if constexpr (E != 0) {
                if (QS_S4B_OWNER((ka & 1))) {        // this wave's row quad of the shared group: one chain, a block of its own
                    unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
                        constexpr int ks = decltype(KS)::value;
                        Yh[ka / 2] = mfma4b(ring[st][ks], bx[ks], ks == 0 ? 0.0 : Yh[ka / 2]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            auto read_next = [&](auto F_) __attribute__((always_inline)) {      // fragment F of row quad ka + 1
                if constexpr (ka + 1 >= N4) read_frag(rd_nx, std::integral_constant<int, ka + 1 - N4>{}, F_);
                else read_frag(rd_in, std::integral_constant<int, ka + 1>{}, F_);
            };
            unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                unroll_b<0, Q>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    if constexpr (SH != 0 && ks == SH && j == 0) {
                        lds_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    Y[ka][j] = mfma4b(ring[st][ks], bf[ks][j], ks == 0 ? 0.0 : Y[ka][j]);
                    if constexpr (j == 0 && ks >= SH) {
                        read_next(std::integral_constant<int, (ks >= SH ? ks - SH : 0)>{});
                        // (the last SH fragments ride with the last k quads' reads: two LDS reads behind one MFMA are free)
                        if constexpr (ks >= N4 - SH) read_next(std::integral_constant<int, ks>{});
                    }
                    if constexpr (j == 1 && ks >= SH) {
                        // the writes of row quad ka + 3 (2 PW k quads), then the loads of row quad ka + 5 (PW k quads)
                        constexpr int t = ks >= SH ? ks - SH : 0;
                        if constexpr (t < 2 * PW) {
                            using I_ = std::integral_constant<int, t / 2>;
                            using H_ = std::integral_constant<int, t % 2>;
                            if constexpr (ka + 3 >= N4) write_elem(w_nx, std::integral_constant<int, ka + 3 - N4>{}, I_{}, H_{});
                            else write_elem(w_in, std::integral_constant<int, ka + 3>{}, I_{}, H_{});
                        } else if constexpr (t < 3 * PW) {
                            using I_ = std::integral_constant<int, t - 2 * PW>;
                            if constexpr (ka + 5 >= N4) load_pair(rs_nx, v_nx, std::integral_constant<int, ka + 5 - N4>{}, I_{});
                            else load_pair(rs_in, v_in, std::integral_constant<int, ka + 5>{}, I_{});
                        } else if constexpr (ka == N4 - 1 && Q == 2) {
                            // (two MFMAs per k quad: the first fragments of Lm share the gaps that are left)
                            constexpr int per = (N4 + (N4 - 3 * PW) - 1) / (N4 - 3 * PW), f0 = (ks - 3 * PW) * per;
                            unroll_b<0, per>([&](auto T) __attribute__((always_inline)) {
                                if constexpr (f0 + decltype(T)::value < N4)
                                    lf[0][f0 + decltype(T)::value] = ltab[(f0 + decltype(T)::value) * 16 + e_lane];
                            });
                        }
                    }
                    if constexpr (j == 2 && ka == N4 - 1) lf[0][ks] = ltab[ks * 16 + e_lane];
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        // ---- the pair exchanges its halves of Y of the shared group (D layout = B-operand layout: as they are)
        double Yx[E ? N4 : 1];
        if constexpr (E != 0) {
            double* const mine = xch + pair * SET + mpar * 64 + lane;
            unroll_b<0, N4 / 2>([&](auto I_) __attribute__((always_inline)) {
                mine[decltype(I_)::value * 128] = Yh[decltype(I_)::value];
            });
            lds_barrier();
            const double* const both = xch + pair * SET + lane;
            unroll_b<0, N4>([&](auto KA) __attribute__((always_inline)) {
                Yx[decltype(KA)::value] = both[decltype(KA)::value * 64];
            });
        }
        // ---- phase 2
        unroll_b<0, N4>([&](auto PG) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value, b = pg & 1;
            QS_S4B_STAMP(64 + pg)
            const unsigned s_prev = opaque((pg ? pg - 1 : 0) * pg_step);
            if constexpr (E != 0) {
                if (QS_S4B_OWNER((pg & 1))) {
                    unroll_b<0, N4>([&](auto KA) __attribute__((always_inline)) {
                        constexpr int ka = decltype(KA)::value;
                        ovx[b] = mfma4b(lf[b][ka], Yx[ka], ka == 0 ? 0.0 : ovx[b]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            unroll_b<0, N4>([&](auto KA) __attribute__((always_inline)) {
                constexpr int ka = decltype(KA)::value;
                unroll_b<0, Q>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    ov[b][j] = mfma4b(lf[b][ka], Y[ka][j], ka == 0 ? 0.0 : ov[b][j]);
                    if constexpr (j == 0 && pg + 1 < N4) lf[b ^ 1][ka] = ltab[((pg + 1) * N4 + ka) * 16 + e_lane];
                    // row quad pg - 1 leaves: one store per two row quads of MFMAs
                    if constexpr (j == 1 && pg >= 1 && (ka & 1) && ka / 2 < Q)
                        store_frag(vo[pg - 1 == N4 - 2 ? 1 : 0][(ka / 2) % Q], s_prev + (unsigned)(jg0 + ka / 2) * jg_step,
                                   ov[b ^ 1][(ka / 2) % Q]);
                    if constexpr (E != 0 && j == 1 && pg >= 1 && ka == 2 * Q + 1)
                        store_frag(vox[b ^ 1][pg - 1 == N4 - 2 ? 1 : 0], s_prev + (unsigned)jx * jg_step, ovx[b ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        {
            constexpr int b = (N4 - 1) & 1;
            const unsigned s_row = opaque((N4 - 1) * pg_step);
            unroll_b<0, Q>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                store_frag(vo[2][j], s_row + (unsigned)(jg0 + j) * jg_step, ov[b][j]);
            });
            if constexpr (E != 0) store_frag(vox[b][2], s_row + (unsigned)jx * jg_step, ovx[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        QS_S4B_STAMP(255)
    };

    unsigned v_in[PW][3], v_nx[PW][3], w_in[PW][2][2], w_nx[PW][2][2], rd_in[2][2], rd_nx[2][2], v_out[3];
    unsigned ph = 0;
    auto rs_in = rsrc(g.in - 1, iq, g.in_item);
    in_offsets(iq, true, ph, v_in, w_in, rd_in);
    double first2[PW][2];      // row quad 2 of the first item quad, fetched before the tables like 0 and 1
    {   // tables (qs_sandwich4.hip): thread t takes element t & 15 of the 4 x 4 blocks t / 16, t / 16 + 16, ...
        constexpr int NF = (N4 * N4 + 15) / 16;
        const auto rs_R = rsrc(g.R, 0, 0), rs_L = rsrc(g.Lm, 0, 0);
        const int ez = (tid >> 2) & 3, ex = tid & 3, b0 = tid >> 4;
        int hi = (b0 >= N4) + (b0 >= 2 * N4), lo = b0 - hi * N4;
        const int r_sk = (int)g.r_sk, r_sj = (int)g.r_sj, l_sp = (int)g.l_sp, l_sa = (int)g.l_sa;
        int off_r = (4 * hi + ez) * r_sk + (4 * lo + ex) * r_sj;
        int off_l = (4 * hi + ex) * l_sp + (4 * lo + ez) * l_sa;
        double rv[NF], lv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const bool in = hi < N4;
            const bool r_ok = in && 4 * hi + ez < L && 4 * lo + ex < M, l_ok = in && 4 * hi + ex < M && 4 * lo + ez < L;
            rv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_R, r_ok ? off_r * 8 : (int)kParkedB, 0, 0));
            lv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_L, l_ok ? off_l * 8 : (int)kParkedB, 0, 0));
            hi += 16 / N4; lo += 16 % N4;
            off_r += (16 / N4) * 4 * r_sk + (16 % N4) * 4 * r_sj;
            off_l += (16 / N4) * 4 * l_sp + (16 % N4) * 4 * l_sa;
            const bool wrap = lo >= N4;
            lo -= wrap ? N4 : 0; hi += wrap;
            off_r += wrap ? 4 * r_sk - N4 * 4 * r_sj : 0;
            off_l += wrap ? 4 * l_sp - N4 * 4 * l_sa : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        // the first two row quads go out before the tables are built (behind the table loads: loads return in order)
        unroll_b<0, 2>([&](auto R_) __attribute__((always_inline)) {
            unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) { load_pair(rs_in, v_in, R_, I_); });
        });
        unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) {
            load_into(rs_in, v_in, std::integral_constant<int, 2>{}, I_, first2[decltype(I_)::value]);
        });
        __builtin_amdgcn_sched_barrier(0);
        // the slots of the last k pair start as zeros (the place of a k that does not exist is never written)
        for (int i = tid; i < 4 * 128; i += 256) transit[(i >> 7) * SET + (N4 - 2) * 64 + (i & 127)] = 0.0;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + 256 * i;
            if (f < N4 * N4 * 16) { rtab[f] = rv[i]; ltab[f] = lv[i]; }
        }
    }
    __syncthreads();
    load_r_fragments();
    // row quads 0, 1, 2 into the transit buffer, 3 and 4 into the registers the first two leave, row quad 0 into the ring
    unroll_b<0, 2>([&](auto R_) __attribute__((always_inline)) {
        unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) {
            write_elem(w_in, R_, I_, std::integral_constant<int, 0>{});
            write_elem(w_in, R_, I_, std::integral_constant<int, 1>{});
        });
    });
    unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) {
        write_from(w_in, std::integral_constant<int, 2>{}, I_, std::integral_constant<int, 0>{}, first2[decltype(I_)::value][0]);
        write_from(w_in, std::integral_constant<int, 2>{}, I_, std::integral_constant<int, 1>{}, first2[decltype(I_)::value][1]);
        load_pair(rs_in, v_in, std::integral_constant<int, 3>{}, I_);
        load_pair(rs_in, v_in, std::integral_constant<int, 4>{}, I_);
    });
    lds_barrier();
    QS_S4B_STAMP(254)
    unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) { read_frag(rd_in, std::integral_constant<int, 0>{}, KS); });

    while (iq < u_end) {
        const unsigned nq = iq + slots;
        const bool more = nq < u_end;
        const unsigned ph_nx = ph ^ ((N4 / 2) & 1);
        auto rs_nx = rsrc(g.in - 1, more ? nq : 0, g.in_item);
        in_offsets(nq, more, ph_nx, v_nx, w_nx, rd_nx);
        auto rs_out = rsrc(g.out, iq, g.out_item);
        out_offsets(iq, v_out);
        quad_pass(rs_in, rs_out, v_out, rs_nx, v_nx, w_nx, rd_nx, v_in, w_in, rd_in);
        iq = nq;
        ph = ph_nx;
        rs_in = rs_nx;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            v_in[i][0] = v_nx[i][0]; v_in[i][1] = v_nx[i][1]; v_in[i][2] = v_nx[i][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { w_in[i][0][c] = w_nx[i][0][c]; w_in[i][1][c] = w_nx[i][1][c]; }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { rd_in[0][c] = rd_nx[0][c]; rd_in[1][c] = rd_nx[1][c]; }
    }
    // ---- The TAIL: the item quads of a partly filled last round (56 orbitals: 784 quads = 3 rounds of 256 workgroups
    // + 16 quads, for which every workgroup used to wait a fourth round, 83 -> 113 us against l = 55) are split by column
    // groups over ALL workgroups -- task = (quad, part): the four waves take the groups 4 part ... 4 part + 3, one each,
    // both products.  Nothing of the main loop's machinery is needed: the whole item quad goes into LDS at once (every
    // wave a quarter of the fragment pairs, in MFMA lane order through the same swizzle as the transit buffer: over the
    // fragments of R -- each wave reads its group's first -- and over the transit sets; the fragments of Lm stay), the A
    // operands of the first product come from there, those of the second from the table.  Per-element sums are the same
    // k-ordered chains.
    if (g.tail_parts) {                                           // (kernel-uniform)
        constexpr int RS = N4 / 4;                                // sets that fit over the fragments of R (N4 N4 16 >= RS SET)
        const unsigned task = blockIdx.x;
        const unsigned tq = g.tail_first + task / g.tail_parts;   // (workgroup-uniform)
        if (tq < g.nquads) {
            const int jg = 4 * (int)(task % g.tail_parts) + wave;
            const bool has_group = 4 * jg < M;                    // (wave-uniform; such a wave still fetches and takes the barriers)
            const int jgc = has_group ? jg : 0;
            __syncthreads();                                      // everybody is through with the transit buffer
            double bt[N4];
            unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
                bt[decltype(KS)::value] = rtab[(decltype(KS)::value * N4 + jgc) * 16 + e_lane];
            });
            auto set_of = [&](int r) __attribute__((always_inline)) {
                return lds + (r < RS ? r * SET : 2 * N4 * N4 * 16 + (r - RS) * SET);
            };
            unsigned v_t[PW][3], w_t[PW][2][2], rd_t[2][2];
            in_offsets(tq, true, 0u, v_t, w_t, rd_t);             // (v_t: lane offsets; the positions are this block's own)
            auto rs_t = rsrc(g.in - 1, tq, g.in_item);
            double tl[N4][PW][2];
            unroll_b<0, N4>([&](auto R_) __attribute__((always_inline)) {
                unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) {
                    load_into(rs_t, v_t, R_, I_, tl[decltype(R_)::value][decltype(I_)::value]);
                });
            });
            const unsigned p0 = f_par * 64 + slot_pos(f_row, f_item, f_k0, f_par);
            const unsigned p1 = f_par * 64 + slot_pos(f_row, f_item, f_k0 + 1, f_par);
            const int k_last = 8 * (NP - 1) + 4 * f_par + f_k0;
            const bool shift = k_last < L && !(k_last + 1 < L);   // the last k of an odd L: fetched 8 bytes early (in_offsets)
            lds_barrier();                                        // the fragments of R are in registers: their table may go
            unroll_b<0, N4>([&](auto R_) __attribute__((always_inline)) {
                constexpr int r = decltype(R_)::value;
                double* const set = set_of(r);
                unroll_b<0, PW>([&](auto I_) __attribute__((always_inline)) {
                    constexpr int i = decltype(I_)::value;
                    const int m = wave + 4 * i;                   // (wave-uniform)
                    if (m < NP) {
                        const bool sh = m == NP - 1 && shift;
                        set[m * 128 + p0] = sh ? tl[r][i][1] : tl[r][i][0];
                        set[m * 128 + p1] = sh ? 0.0 : tl[r][i][1];
                    }
                });
            });
            __syncthreads();
            if (has_group) {
                // Y[ka] = In[ka] . R[:, group]: two chains at a time (a single one waits 4 cycles per link)
                double Yt[N4];
                unroll_b<0, N4 / 2>([&](auto H_) __attribute__((always_inline)) {
                    constexpr int ka = 2 * decltype(H_)::value;
                    const double* const s0 = set_of(ka);
                    const double* const s1 = set_of(ka + 1);
                    unroll_b<0, N4>([&](auto KS) __attribute__((always_inline)) {
                        constexpr int ks = decltype(KS)::value;
                        const unsigned rd = (ks & 1) ? rd_odd : rd_even;
                        Yt[ka] = mfma4b(s0[ks * 64 + rd], bt[ks], ks == 0 ? 0.0 : Yt[ka]);
                        Yt[ka + 1] = mfma4b(s1[ks * 64 + rd], bt[ks], ks == 0 ? 0.0 : Yt[ka + 1]);
                    });
                });
                // Out[pg] = sum_ka Lm[pg][ka] . Y[ka], stored as it is finished
                unsigned v_o[3];
                out_offsets(tq, v_o);
                const auto rs_o = rsrc(g.out, tq, g.out_item);
                const bool col_ok = 4 * jg + x < M;
                unroll_b<0, N4 / 2>([&](auto H_) __attribute__((always_inline)) {
                    constexpr int pg = 2 * decltype(H_)::value;
                    double o0 = 0.0, o1 = 0.0;
                    unroll_b<0, N4>([&](auto KA) __attribute__((always_inline)) {
                        constexpr int ka = decltype(KA)::value;
                        o0 = mfma4b(ltab[(pg * N4 + ka) * 16 + e_lane], Yt[ka], ka == 0 ? 0.0 : o0);
                        o1 = mfma4b(ltab[((pg + 1) * N4 + ka) * 16 + e_lane], Yt[ka], ka == 0 ? 0.0 : o1);
                    });
                    const unsigned s_col = (unsigned)jg * jg_step;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, o0), rs_o,
                        (int)(col_ok ? v_o[pg == N4 - 2 ? 1 : 0] : kParkedB), (int)(pg * pg_step + s_col), QS_S4B_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, o1), rs_o,
                        (int)(col_ok ? v_o[pg + 1 == N4 - 1 ? 2 : 0] : kParkedB), (int)((pg + 1) * pg_step + s_col), QS_S4B_STORE_AUX);
                });
            }
        }
    }
    QS_S4B_TRACE_DONE
}

unsigned sandwich4b_tail_quads(unsigned nquads, int M) {
    if (!g_tune.sandwich_tail) return 0;
    const int n_cu = device_cu_count();
    const unsigned wgs = (unsigned)(n_cu - n_cu % 8);
    if (wgs < 8 || nquads <= wgs) return 0;
    const unsigned ntail = nquads % wgs, parts = (unsigned)((cdiv(M, 4) + 3) / 4);
    return ntail * parts <= wgs ? ntail : 0;
}

template <int N4>
static int launch_sandwich4b(const S4Args& g, hipStream_t stream, int dry_run) {
    if (dry_run) return QS_OK;
    const int n_cu = device_cu_count();
    int64_t wgs = n_cu - n_cu % 8;                       // one workgroup (four waves, one per SIMD) per CU
    if (wgs < 8) wgs = 8;
    const int64_t need = ((int64_t)g.nquads + 7) / 8 * 8;
    if (wgs > need) wgs = need;                          // short item lists: no idle workgroups
    size_t lds = sizeof(double) * (2 * N4 * N4 * 16 + 8 * N4 * 64 + ((N4 % 4) ? 2 * N4 * 64 : 0));
    // A last round that is only partly filled: its quads split over all workgroups (the kernel's tail), when one task
    // per workgroup covers them.  (The whole-quad staging of the tail takes N4 - N4 / 4 sets behind the tables.)
    S4Args gt = g;
    if (sandwich4b_tail_quads(g.nquads, g.M)) {
        const size_t lds_tail = sizeof(double) * (2 * N4 * N4 * 16 + (N4 - N4 / 4) * N4 * 64);
        if (lds_tail <= 160 * 1024) {
            gt.tail_first = g.nquads / (unsigned)wgs * (unsigned)wgs;
            gt.tail_parts = (unsigned)((cdiv(g.M, 4) + 3) / 4);
            if (lds_tail > lds) lds = lds_tail;
        }
    }
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)sandwich4b_kernel<N4>, lds, lds_opt_in, "hipFuncSetAttribute(sandwich4b)"))
        return rc;
    hipLaunchKernelGGL(sandwich4b_kernel<N4>, dim3((unsigned)wgs), dim3(256), lds, stream, gt);
    note_dispatch(gt.tail_parts ? "qs::sandwich4b_kernel<%d>+tail" : "qs::sandwich4b_kernel<%d>", N4);
    return launch_status("sandwich4b launch");
}

int sandwich4b_launch(const S4Args& g, int n4, hipStream_t stream, int dry_run) {
    if (g.in_col != 1) return 1;
    switch (n4) {
#ifdef QS_S4_ONLY          // development builds: one instantiation compiles in seconds
        case QS_S4_ONLY: return launch_sandwich4b<QS_S4_ONLY>(g, stream, dry_run);
#else
        case 10: return launch_sandwich4b<10>(g, stream, dry_run);
        case 12: return launch_sandwich4b<12>(g, stream, dry_run);
        case 14: return launch_sandwich4b<14>(g, stream, dry_run);
        case 16: return launch_sandwich4b<16>(g, stream, dry_run);
#endif
        default: return 1;
    }
}

}  // namespace qs

#ifdef QS_S4_TRACE
extern "C" int qs_s4b_trace_reset(void) {
    unsigned zero = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qs::qs_s4b_trace_n), &zero, sizeof(zero));
}
extern "C" int qs_s4b_trace_read(void* dst) {      // dst: device buffer of 4097 x 8 bytes: count, stamps
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(qs::qs_s4b_trace_n), sizeof(n));
    unsigned long long nn = n;
    (void)hipMemcpy(dst, &nn, 8, hipMemcpyHostToDevice);
    void* src = nullptr;
    (void)hipGetSymbolAddress(&src, HIP_SYMBOL(qs::qs_s4b_trace));
    return (int)hipMemcpy((char*)dst + 8, src, 4096 * 8, hipMemcpyDeviceToDevice);
}
#endif
