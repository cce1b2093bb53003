// Small-coefficient streaming product  C[b] (m x n) = A (m x k) . B[b] (k x n)
// for m, k <= 64 (fp64 and complex128): the c, b and a contractions of a transform with few
// orbitals (BASELINE.json configs[1], l = 55), where A is Ct or C^T and B is
// the tensor.
//
// Why a third form: below l ~ 100 the four unfused contractions are bound by
// their eight passes over the tensor, not by the matrix pipe (l = 55: 73 MB in,
// 73 MB out per pass, 1 GFLOP), and the tiled kernels spend most of a
// 4-k-stage tile on prologue, barriers and address arithmetic
// (profiles/r01_gemm_notes.txt: MFMA pipe 43 % busy, ~600 VALU instructions per
// wave and tile).  Here nothing is staged and nothing is shared:
//   * the whole of A lives in registers as MFMA A-operand fragments
//     (TM row tiles x 16 k-steps = at most 64 doubles per lane), loaded once
//     per wave;
//   * B is streamed: a lane loads its MFMA B-operand fragments straight from
//     global memory (lanes 0-15 read 128 or 256 contiguous bytes of row k,
//     lanes 16-31 of row k+1, ...); the register of k-step ks is refilled with
//     the same k-step of the wave's NEXT column block as soon as it has been
//     consumed, so a full block of loads is always in flight per wave;
//   * a wave owns a block of 16*NT columns and either all row tiles of A or
//     (SPLIT = 2) half of them, the other half belonging to its neighbour wave,
//     which streams the same columns (second reader hits the CU's L1): half
//     the fragment registers per wave -> two waves per SIMD;
//   * no LDS, no barrier, and no branch in the block loop: loads and stores are
//     buffer instructions whose range check does the edge handling -- B rows
//     k >= K lie past num_records and read as 0.0 (A's k >= K and row >= m
//     fragments are 0.0 too), C rows >= m lie past num_records and are dropped,
//     lanes whose column is >= n load a clamped column and store to an offset
//     past num_records.  A column of B only ever reaches the same column of C,
//     so no stray value can leak.  With a branch-free body the compiler's
//     s_waitcnt vmcnt(N) counts are exact (with guarded loads it fell back to
//     vmcnt(0) at every block start and the wave idled on its own stores).
// K is padded to a multiple of 16 (template KQ): the padded k-steps multiply
// zeros and read nothing.  Same k order as the tiled kernels.
// Algorithmic bytes per launch: 8*(k + m)*n*batch; roofline = HBM.

#include <type_traits>

#include "qs_common.h"

namespace qs {

// f(integral_constant<int, I>) for I = 0 .. N-1, unrolled at compile time
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct StreamArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int64_t sb, sc;          // batch strides of B and C (A is shared)
    int m, k;
    int n;
    unsigned blocks_per_batch;   // ceil(n / (16*NT))
    unsigned total_blocks;       // blocks_per_batch * batch
};

template <bool VEC>
struct StreamIO;
template <>
struct StreamIO<true> {
    typedef f64x2 type;
    template <class R>
    static __device__ __forceinline__ type load(R rsrc, unsigned off) {
        return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
    }
    template <class R>
    static __device__ __forceinline__ void store(R rsrc, unsigned off, double a, double b) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f64x2{a, b}), rsrc, (int)off, 0, 0);
    }
};
template <>
struct StreamIO<false> {
    typedef double type;
    template <class R>
    static __device__ __forceinline__ type load(R rsrc, unsigned off) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0));
    }
    template <class R>
    static __device__ __forceinline__ void store(R rsrc, unsigned off, double a) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rsrc, (int)off, 0, 0);
    }
};

// TMW = row tiles per wave, SPLIT = waves sharing a column block (rows 16*TMW*part ...),
// KQ = ceil(k / 16), NT = column tiles per block, VEC = 16-byte accesses
template <int TMW, int SPLIT, int KQ, int NT, bool VEC>
__global__ __launch_bounds__(256, (TMW * KQ > 8 ? 1 : 2))
void gemm_stream_left_kernel(const StreamArgs g) {
    constexpr int NKS = 4 * KQ;                  // k-steps (K padded to a multiple of 16)
    constexpr int NLD = VEC ? NT / 2 : NT;       // loads per lane and k-step
    constexpr int CPL = VEC ? 2 : 1;             // columns per load
    constexpr int BW = 16 * NT;                  // columns per block
    using IO = StreamIO<VEC>;
    typedef typename IO::type ld_t;
    static_assert(NT % 2 == 0, "column tiles come in pairs");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const int part = wave % SPLIT;               // which rows of A this wave multiplies
    const int row0 = 16 * TMW * part;

    // ---- A -> fragments (once); rows >= m and k >= K are zeros
    double af[TMW][NKS];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int row = row0 + 16 * i + c16, kk = 4 * ks + g4;
            af[i][ks] = (row < g.m && kk < g.k) ? g.A[(int64_t)row * g.lda + kk] : 0.0;
        }
    }

    const unsigned w = (blockIdx.x * 4 + wave) / SPLIT, W = gridDim.x * 4 / SPLIT;
    if (w >= g.total_blocks) return;

    // range checks: a batch slice of B holds (k-1)*ldb + n elements, one of C (m-1)*ldc + n
    const unsigned b_room = (unsigned)(((int64_t)(g.k - 1) * g.ldb + g.n) * 8);
    const unsigned c_room = (unsigned)(((int64_t)(g.m - 1) * g.ldc + g.n) * 8);
    const unsigned kstep = (unsigned)(4 * g.ldb * 8);    // bytes per k-step
    const unsigned rstep = (unsigned)(4 * g.ldc * 8);    // bytes per 4 rows of C

    // descriptor of batch slice b of B / C, and the lane offsets inside column block cb
    auto slice = [&](const double* base, int64_t stride, unsigned b, unsigned room) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base + (int64_t)b * stride);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 (int)room, 0x00020000);
    };
    auto b_offsets = [&](unsigned cb, unsigned (&off)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            int col = (int)cb * BW + 16 * CPL * q + CPL * c16;
            col = col < g.n - CPL ? col : g.n - CPL;    // VEC: n is even, pairs stay pairs
            off[q] = (unsigned)(g4 * g.ldb + col) * 8u;
        }
    };

    unsigned blk_b = w / g.blocks_per_batch, blk_cb = w - blk_b * g.blocks_per_batch;
    auto rs_next = slice(g.B, g.sb, blk_b, b_room);
    unsigned off_next[NLD];
    b_offsets(blk_cb, off_next);

    ld_t ring[NKS][NLD];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int q = 0; q < NLD; ++q) ring[ks][q] = IO::load(rs_next, off_next[q] + ks * kstep);

    for (unsigned blk = w; blk < g.total_blocks; blk += W) {
        const unsigned b = blk_b, cb = blk_cb;
        // next block of this wave (the last iteration re-aims at its own block: loads that nobody uses)
        const unsigned nxt = blk + W < g.total_blocks ? blk + W : blk;
        blk_b = nxt / g.blocks_per_batch;
        blk_cb = nxt - blk_b * g.blocks_per_batch;
        rs_next = slice(g.B, g.sb, blk_b, b_room);
        b_offsets(blk_cb, off_next);

        f64x4 acc[TMW][NT];                 // k-step 0 starts every accumulator from the literal zero

#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            // program order is pinned per k-step: multiply with register ks, THEN refill it with the
            // same k-step of the next block (the old value is dead, so the refill lands in the same
            // register: no copies at the loop edge, which would have to wait for every load).  Left to
            // itself the scheduler hoists all refills to the top of the block and then waits on them.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    if constexpr (VEC) {
                        acc[i][2 * q] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][ks], ring[ks][q][0], ks == 0 ? zero4 : acc[i][2 * q], 0, 0, 0);
                        acc[i][2 * q + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][ks], ring[ks][q][1], ks == 0 ? zero4 : acc[i][2 * q + 1], 0, 0, 0);
                    } else {
                        acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][ks], ring[ks][q], ks == 0 ? zero4 : acc[i][q], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NLD; ++q) ring[ks][q] = IO::load(rs_next, off_next[q] + ks * kstep);
        }

        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: reg r of a lane -> row row0 + 16i + g4 + 4r; rows >= m fall past num_records,
        // lanes whose column is >= n start from an offset that is past it for every row
        const auto rs_c = slice(g.C, g.sc, b, c_room);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int col = (int)cb * BW + 16 * CPL * q + CPL * c16;
            const unsigned o0 = col < g.n ? (unsigned)((row0 + g4) * g.ldc + col) * 8u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = o0 + (unsigned)(4 * i + r) * rstep;
                    if constexpr (VEC) IO::store(rs_c, o, acc[i][2 * q][r], acc[i][2 * q + 1][r]);
                    else IO::store(rs_c, o, acc[i][q][r]);
                }
            }
        }
    }
}

// complex128 form: one complex element per lane and load (16 bytes), A fragments as separate re / im
// registers, four MFMAs per fragment pair in the order of the tiled kernels
// (re += ar.br; im += ar.bi; re += (-ai).bi; im += ai.br).  Two column tiles per block.
template <int TMW, int SPLIT, int KQ>
__global__ __launch_bounds__(256, (TMW * KQ > 3 ? 1 : 2))
void gemm_stream_left_cx_kernel(const StreamArgs g) {
    constexpr int NKS = 4 * KQ, NT = 2, BW = 16 * NT;
    using IO = StreamIO<true>;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const int part = wave % SPLIT;
    const int row0 = 16 * TMW * part;

    double ar[TMW][NKS], ai[TMW][NKS];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int row = row0 + 16 * i + c16, kk = 4 * ks + g4;
            const bool ok = row < g.m && kk < g.k;
            const f64x2 v = ok ? *reinterpret_cast<const f64x2*>(g.A + ((int64_t)row * g.lda + kk) * 2) : f64x2{0.0, 0.0};
            ar[i][ks] = v[0];
            ai[i][ks] = v[1];
        }
    }

    const unsigned w = (blockIdx.x * 4 + wave) / SPLIT, W = gridDim.x * 4 / SPLIT;
    if (w >= g.total_blocks) return;

    const unsigned b_room = (unsigned)(((int64_t)(g.k - 1) * g.ldb + g.n) * 16);
    const unsigned c_room = (unsigned)(((int64_t)(g.m - 1) * g.ldc + g.n) * 16);
    const unsigned kstep = (unsigned)(4 * g.ldb * 16);
    const unsigned rstep = (unsigned)(4 * g.ldc * 16);

    auto slice = [&](const double* base, int64_t stride, unsigned b, unsigned room) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base + (int64_t)b * stride * 2);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 (int)room, 0x00020000);
    };
    auto b_offsets = [&](unsigned cb, unsigned (&off)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            int col = (int)cb * BW + 16 * q + c16;
            col = col < g.n - 1 ? col : g.n - 1;
            off[q] = (unsigned)(g4 * g.ldb + col) * 16u;
        }
    };

    unsigned blk_b = w / g.blocks_per_batch, blk_cb = w - blk_b * g.blocks_per_batch;
    auto rs_next = slice(g.B, g.sb, blk_b, b_room);
    unsigned off_next[NT];
    b_offsets(blk_cb, off_next);

    f64x2 ring[NKS][NT];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int q = 0; q < NT; ++q) ring[ks][q] = IO::load(rs_next, off_next[q] + ks * kstep);

    for (unsigned blk = w; blk < g.total_blocks; blk += W) {
        const unsigned b = blk_b, cb = blk_cb;
        const unsigned nxt = blk + W < g.total_blocks ? blk + W : blk;
        blk_b = nxt / g.blocks_per_batch;
        blk_cb = nxt - blk_b * g.blocks_per_batch;
        rs_next = slice(g.B, g.sb, blk_b, b_room);
        b_offsets(blk_cb, off_next);

        f64x4 cr[TMW][NT], ci[TMW][NT];     // k-step 0 starts every accumulator from the literal zero

#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const double br = ring[ks][q][0], bi = ring[ks][q][1];
                    cr[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i][ks], br, ks == 0 ? zero4 : cr[i][q], 0, 0, 0);
                    ci[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i][ks], bi, ks == 0 ? zero4 : ci[i][q], 0, 0, 0);
                    cr[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai[i][ks], bi, cr[i][q], 0, 0, 0);
                    ci[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i][ks], br, ci[i][q], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NT; ++q) ring[ks][q] = IO::load(rs_next, off_next[q] + ks * kstep);
        }

        __builtin_amdgcn_sched_barrier(0);
        const auto rs_c = slice(g.C, g.sc, b, c_room);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = (int)cb * BW + 16 * q + c16;
            const unsigned o0 = col < g.n ? (unsigned)((row0 + g4) * g.ldc + col) * 16u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double zr = cr[i][q][r], zi = ci[i][q][r];
                    IO::store(rs_c, o0 + (unsigned)(4 * i + r) * rstep, zr, zi);
                }
            }
        }
    }
}


template <int TMW, int SPLIT, int KQ, bool VEC>
static int launch_stream(const StreamArgs& g0, int64_t batch, hipStream_t stream) {
    constexpr int NT = 2;   // 32-column blocks: 16 KB of loads in flight per wave either way
    StreamArgs g = g0;
    const int64_t bpb = cdiv(g.n, 16 * NT);
    const int64_t total = bpb * batch;
    if (total <= 0 || total >= (int64_t(1) << 30)) return 1;
    g.blocks_per_batch = (unsigned)bpb;
    g.total_blocks = (unsigned)total;
    const int n_cu = device_cu_count();
    // persistent waves: every wave (pair) walks blocks w, w + W, ...; as many workgroups per CU as the
    // kernel's registers admit (asked once per instantiation)
    static PerDeviceInt occupancy;
    const int wg_per_cu = resident_workgroups((const void*)gemm_stream_left_kernel<TMW, SPLIT, KQ, NT, VEC>, occupancy,
                                              (TMW * KQ > 8) ? 1 : 2);
    int64_t wgs = cdiv(total * SPLIT, 4);
    if (wgs > (int64_t)wg_per_cu * n_cu) wgs = (int64_t)wg_per_cu * n_cu;
    hipLaunchKernelGGL((gemm_stream_left_kernel<TMW, SPLIT, KQ, NT, VEC>), dim3((unsigned)wgs), dim3(256), 0, stream, g);
    note_dispatch("qs::gemm_stream_left_kernel<%d, %d, %d, %d, %s>", TMW, SPLIT, KQ, NT, VEC ? "true" : "false");
    return launch_status("gemm_stream launch");
}

template <int TMW, int SPLIT, int KQ>
static int launch_stream_cx(const StreamArgs& g0, int64_t batch, hipStream_t stream) {
    StreamArgs g = g0;
    const int64_t bpb = cdiv(g.n, 32);
    const int64_t total = bpb * batch;
    if (total <= 0 || total >= (int64_t(1) << 30)) return 1;
    g.blocks_per_batch = (unsigned)bpb;
    g.total_blocks = (unsigned)total;
    const int n_cu = device_cu_count();
    static PerDeviceInt occupancy;
    const int wg_per_cu = resident_workgroups((const void*)gemm_stream_left_cx_kernel<TMW, SPLIT, KQ>, occupancy,
                                              (TMW * KQ > 3) ? 1 : 2);
    int64_t wgs = cdiv(total * SPLIT, 4);
    if (wgs > (int64_t)wg_per_cu * n_cu) wgs = (int64_t)wg_per_cu * n_cu;
    hipLaunchKernelGGL((gemm_stream_left_cx_kernel<TMW, SPLIT, KQ>), dim3((unsigned)wgs), dim3(256), 0, stream, g);
    note_dispatch("qs::gemm_stream_left_cx_kernel<%d, %d, %d>", TMW, SPLIT, KQ);
    return launch_status("gemm_stream_cx launch");
}

// QS_OK / error after launching, 1 = not eligible (caller falls back).
int gemm_stream_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                    int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                    int64_t sb, int64_t sc, int accumulate, hipStream_t stream) {
    if (!g_tune.gemm_stream || accumulate) return 1;
    if (dtype != QS_F64 && dtype != QS_C128) return 1;
    const bool cx = dtype == QS_C128;
    const int64_t esz = cx ? 16 : 8;
    if (m > 64 || k > 64 || m < 1 || k < 1) return 1;
    if (batch > 1 && sa != 0) return 1;                  // A is the shared coefficient matrix
    if (ldb < n || ldc < n) return 1;                    // the range checks assume rows do not overlap
    // 32-bit offsets inside a batch slice, with the "column >= n" marker bit free
    if (((k + 3) * ldb + n) * esz >= (int64_t(1) << 31) || ((m + 63) * ldc + n) * esz >= (int64_t(1) << 31)) return 1;
    if (batch >= (int64_t(1) << 30)) return 1;
    // a stream long enough to keep every wave busy for a few blocks.  Measured against the tiled kernels
    // (profiles/r01_gemm_notes.txt): fp64 wins from l = 32 up (+5...10 %), complex128 only around l = 48
    // (+13 %; below that the tiled kernel's 32 x 32 shape is faster, at 55 and 64 they are level)
    if (n * batch < (int64_t(1) << (cx ? 16 : 15))) return 1;
    if (cx && k > 48) return 1;
    // whole-tile fp64 products (l = 64) run faster on the exact form of the tiled kernel (42.7 vs 41.3 TFLOP/s)
    if (!cx && m % 64 == 0 && n % 128 == 0 && k % 16 == 0) return 1;
    StreamArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sb = sb; g.sc = sc;
    g.m = (int)m; g.k = (int)k; g.n = (int)n;
    g.blocks_per_batch = 0;
    g.total_blocks = 0;
    const int tm = (int)cdiv(m, 16), kq = (int)cdiv(k, 16);
    if (cx) {
        if (!aligned(A, 16) || !aligned(B, 16) || !aligned(C, 16)) return 1;
#define QS_STREAM_CX(TMWV, SPLITV)                                              \
        switch (kq) {                                                           \
            case 1: return launch_stream_cx<TMWV, SPLITV, 1>(g, batch, stream); \
            case 2: return launch_stream_cx<TMWV, SPLITV, 2>(g, batch, stream); \
            case 3: return launch_stream_cx<TMWV, SPLITV, 3>(g, batch, stream); \
            default: return launch_stream_cx<TMWV, SPLITV, 4>(g, batch, stream); \
        }
        if (tm == 1) { QS_STREAM_CX(1, 1) }
        if (tm == 2) { QS_STREAM_CX(2, 1) }
        QS_STREAM_CX(2, 2)
#undef QS_STREAM_CX
    }
    const bool vec = aligned(B, 16) && aligned(C, 16) && !(ldb & 1) && !(ldc & 1) && !(sb & 1) &&
                     !(sc & 1) && !(n & 1);
    const bool split = tm > 2 && g_tune.gemm_stream != 2;
#define QS_STREAM_KQ(TMWV, SPLITV)                                                                         \
    switch (kq) {                                                                                          \
        case 1: return vec ? launch_stream<TMWV, SPLITV, 1, true>(g, batch, stream) : launch_stream<TMWV, SPLITV, 1, false>(g, batch, stream); \
        case 2: return vec ? launch_stream<TMWV, SPLITV, 2, true>(g, batch, stream) : launch_stream<TMWV, SPLITV, 2, false>(g, batch, stream); \
        case 3: return vec ? launch_stream<TMWV, SPLITV, 3, true>(g, batch, stream) : launch_stream<TMWV, SPLITV, 3, false>(g, batch, stream); \
        default: return vec ? launch_stream<TMWV, SPLITV, 4, true>(g, batch, stream) : launch_stream<TMWV, SPLITV, 4, false>(g, batch, stream); \
    }
    if (tm == 1) { QS_STREAM_KQ(1, 1) }
    if (tm == 2) { QS_STREAM_KQ(2, 1) }
    if (split) { QS_STREAM_KQ(2, 2) }
    QS_STREAM_KQ(4, 1)
#undef QS_STREAM_KQ
}

}  // namespace qs
