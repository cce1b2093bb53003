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

constexpr unsigned kParked = 0x80000000u;     // lane offset of an out-of-range lane (>= num_records)

__device__ __forceinline__ double mfma4(double a, double b, double c) {
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
};

// N4 = ceil(L / 4) = ceil(M / 4)
template <int N4>
__global__ __launch_bounds__(256, 1) void sandwich4_kernel(const S4Args g) {
    constexpr int NCH = (N4 + 3) / 4;                       // chunks of column groups, one per wave
    constexpr int NJ_BIG = (N4 + NCH - 1) / NCH;            // the first N4 % NCH chunks (all if it divides) have this many
    constexpr int NJ_SMALL = N4 / NCH;
    constexpr int N_BIG = (N4 % NCH) ? (N4 % NCH) : NCH;
    static_assert(NCH <= 4 && NJ_BIG <= 4, "a workgroup has four waves");

    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* rtab = lds;                    // [ks][jg][16]: R[4 ks + z][4 jg + x]  at z * 4 + x
    double* ltab = lds + N4 * N4 * 16;     // [pg][ka][16]: Lm[4 pg + x][4 ka + z] at z * 4 + x
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int f = tid; f < N4 * N4 * 16; f += 256) {
        const int blk = f >> 4, e = f & 15, hi = blk / N4, lo = blk % N4, z = e >> 2, x = e & 3;
        const int k = 4 * hi + z, j = 4 * lo + x;
        rtab[f] = (k < L && j < M) ? g.R[k * g.r_sk + j * g.r_sj] : 0.0;
        const int p = 4 * hi + x, a = 4 * lo + z;
        ltab[f] = (p < M && a < L) ? g.Lm[p * g.l_sp + a * g.l_sa] : 0.0;
    }
    __syncthreads();
    if (wave >= NCH) return;

    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    const int e_lane = z * 4 + x;
    const int rl = L & 3, rm = M & 3;      // valid rows / columns of the last quad (0 = all four)

    // ---- item quads of this workgroup: every XCD takes a contiguous range, neighbouring workgroups of an
    // XCD take neighbouring quads (they share 128-byte lines of the tensor in the XCD's L2)
    const unsigned n_xcd = 8, xcd = blockIdx.x % n_xcd, slot = blockIdx.x / n_xcd, slots = gridDim.x / n_xcd;
    const unsigned per = (g.nquads + n_xcd - 1) / n_xcd;
    const unsigned q_end = (xcd + 1) * per < g.nquads ? (xcd + 1) * per : g.nquads;
    unsigned iq = xcd * per + slot;
    if (iq >= q_end) return;

    const unsigned ka_step = (unsigned)(4 * g.in_row * 8), ks_step = (unsigned)(4 * g.in_col * 8);
    const unsigned pg_step = (unsigned)(4 * g.out_row * 8), jg_step = (unsigned)(4 * g.out_col * 8);

    auto rsrc = [&](const double* base, unsigned quad, int64_t item_stride) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base + (int64_t)quad * 4 * item_stride);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 0x7fffffff, 0x00020000);
    };
    // lane offsets of an In fragment (A operand: row x, block y, k z).  Variant bit 0: last row quad, bit 1:
    // last k quad.  A lane whose row / k / item does not exist is parked.
    auto in_offsets = [&](unsigned quad, bool live, unsigned (&v)[4]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)((x * g.in_row + y * g.in_item + z * g.in_col) * 8);
        const bool item_ok = live && quad * 4 + y < g.nitems;
        const bool row_ok = rl == 0 || x < rl, k_ok = rl == 0 || z < rl;
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

    double ring[2][N4];       // In fragments of two row quads: stage (step & 1), register ks

    auto load_quad_row = [&](auto rs, const unsigned (&v)[4], auto KA, auto STAGE) __attribute__((always_inline)) {
        constexpr int ka = decltype(KA)::value, st = decltype(STAGE)::value;
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            constexpr int var = (ka == N4 - 1 ? 1 : 0) | (ks == N4 - 1 ? 2 : 0);
            ring[st][ks] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                rs, (int)v[var], (int)(ka * ka_step + ks * ks_step), 0));
        });
    };

    // One chunk: NJ column groups starting at jg0 for the item quad behind (rs_in, v_in); the ring already
    // holds row quads 0 and 1 of it.  While row quad ka multiplies, row quad ka + 2 is fetched -- of this item
    // quad, or (the last two) of the next one, behind (rs_pf, v_pf).  P = parity of the ring stage of ka = 0.
    auto chunk = [&](auto NJC, auto PC, int jg0, auto rs_in, auto rs_out, const unsigned (&v_out)[4],
                     auto rs_pf, const unsigned (&v_pf)[4], const unsigned (&v_in)[4]) __attribute__((always_inline)) {
        constexpr int NJ = decltype(NJC)::value, P = decltype(PC)::value;
        double bf[N4][NJ];
        unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                bf[decltype(KS)::value][decltype(J)::value] =
                    rtab[(decltype(KS)::value * N4 + jg0 + decltype(J)::value) * 16 + e_lane];
            });
        });
        double acc2[N4][NJ];
        unroll<0, N4>([&](auto KA) __attribute__((always_inline)) {
            constexpr int ka = decltype(KA)::value, st = (ka + P) & 1;
            // ---- Y[ka rows][own columns] = In[ka rows][:] . R[:][own columns]
            double acc1[NJ];
            unroll<0, N4>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    acc1[j] = mfma4(ring[st][ks], bf[ks][j], ks == 0 ? 0.0 : acc1[j]);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            // the stage is free: fetch row quad ka + 2 into it
            if constexpr (ka + 2 < N4) {
                load_quad_row(rs_in, v_in, std::integral_constant<int, ka + 2>{}, std::integral_constant<int, st>{});
            } else {
                load_quad_row(rs_pf, v_pf, std::integral_constant<int, ka + 2 - N4>{}, std::integral_constant<int, st>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- Out[:][own columns] += Lm[:][ka rows] . Y[ka rows][own columns]   (Y straight from the accumulators)
            unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
                constexpr int pg = decltype(PG)::value;
                const double lf = ltab[(pg * N4 + ka) * 16 + e_lane];
                unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    acc2[pg][j] = mfma4(lf, acc1[j], ka == 0 ? 0.0 : acc2[pg][j]);
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        unroll<0, N4>([&](auto PG) __attribute__((always_inline)) {
            constexpr int pg = decltype(PG)::value;
            unroll<0, NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const double val = acc2[pg][j];
                const bool last_col = jg0 + j == N4 - 1;                  // wave-uniform
                const unsigned vo = last_col ? v_out[(pg == N4 - 1 ? 1 : 0) | 2] : v_out[pg == N4 - 1 ? 1 : 0];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), rs_out, (int)vo,
                                                      (int)(pg * pg_step + (jg0 + j) * jg_step), 0);
            });
        });
    };

    // chunk c: first column group and size
    auto chunk_first = [&](int c) __attribute__((always_inline)) {
        return c < N_BIG ? c * NJ_BIG : N_BIG * NJ_BIG + (c - N_BIG) * NJ_SMALL;
    };

    unsigned v_in[4], v_nx[4], v_out[4];
    auto rs_in = rsrc(g.in, iq, g.in_item);
    in_offsets(iq, true, v_in);
    load_quad_row(rs_in, v_in, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if constexpr (N4 > 1)
        load_quad_row(rs_in, v_in, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});

    unsigned parity = 0;      // ring stage of row quad 0 of the current item quad (alternates when N4 is odd)
    for (unsigned round = 0; iq < q_end; ++round) {
        const unsigned nq = iq + slots;
        const bool more = nq < q_end;
        auto rs_nx = rsrc(g.in, more ? nq : iq, g.in_item);
        in_offsets(nq, more, v_nx);
        auto rs_out = rsrc(g.out, iq, g.out_item);
        out_offsets(iq, v_out);
        const int c = (int)((wave + round) % NCH);           // chunks rotate over the waves from round to round
        const int jg0 = chunk_first(c);
        const bool big = c < N_BIG;
        if constexpr (N4 % 2 == 0) {
            if (big) chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 0>{}, jg0, rs_in, rs_out,
                           v_out, rs_nx, v_nx, v_in);
            else chunk(std::integral_constant<int, NJ_SMALL>{}, std::integral_constant<int, 0>{}, jg0, rs_in, rs_out,
                       v_out, rs_nx, v_nx, v_in);
        } else {
            if (big) {
                if (parity) chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 1>{}, jg0, rs_in,
                                  rs_out, v_out, rs_nx, v_nx, v_in);
                else chunk(std::integral_constant<int, NJ_BIG>{}, std::integral_constant<int, 0>{}, jg0, rs_in, rs_out,
                           v_out, rs_nx, v_nx, v_in);
            } else {
                if (parity) chunk(std::integral_constant<int, NJ_SMALL>{}, std::integral_constant<int, 1>{}, jg0, rs_in,
                                  rs_out, v_out, rs_nx, v_nx, v_in);
                else chunk(std::integral_constant<int, NJ_SMALL>{}, std::integral_constant<int, 0>{}, jg0, rs_in,
                           rs_out, v_out, rs_nx, v_nx, v_in);
            }
            parity ^= 1;
        }
        iq = nq;
        rs_in = rs_nx;
#pragma unroll
        for (int i = 0; i < 4; ++i) v_in[i] = v_nx[i];
    }
}

template <int N4>
static int launch_sandwich4(const S4Args& g, hipStream_t stream) {
    const int n_cu = device_cu_count();
    int64_t wgs = n_cu - n_cu % 8;                       // one workgroup (four waves, one per SIMD) per CU
    if (wgs < 8) wgs = 8;
    const int64_t need = ((int64_t)g.nquads + 7) / 8 * 8;
    if (wgs > need) wgs = need;                          // short item lists: no idle workgroups
    const size_t lds = sizeof(double) * 2 * N4 * N4 * 16;
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
    switch (n4) {
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
        default: return 1;
    }
}

}  // namespace qs
