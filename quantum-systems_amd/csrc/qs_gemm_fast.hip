// Exact-tiling fast path of the batched row-major product (see qs_gemm.hip
// for the general kernel and the contraction -> GEMM map).
//
// Why a second kernel: on gfx950 the fp64 MFMA shares the SIMD's vector issue
// with ordinary VALU work -- every VALU instruction in the K loop takes its
// issue cycles away from the matrix pipe (measured with tools/probe_mix.hip:
// 32 integer adds per 16 MFMAs cost 12 % of the MFMA rate, two co-resident
// waves do not hide it).  The general kernel spends ~130 VALU instructions per
// K step on 64-bit address arithmetic and edge handling.  This kernel is for
// products whose extents are whole multiples of the tile (l = 128, 256, 512,
// ...): the K loop then contains no VALU work at all --
//   * global loads use the scalar-base form (SGPR pointer + one loop-invariant
//     32-bit lane offset); the per-row bases advance on the scalar ALU;
//   * LDS addresses are one VGPR + immediates (the loop is unrolled by two so
//     the stage buffer is a compile-time constant);
//   * no bounds checks, no zero fill.
// Schedule (same as the general kernel's rotated schedule): fragments are
// double-buffered in registers, k-step kk+1 is read while kk multiplies, and
// the last k-step of a stage runs after the stage barrier while the next
// stage's first fragments and the global loads of the stage after are in
// flight, so the MFMA stream does not stop at the barrier.

#include <type_traits>

#include "qs_common.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct FastArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int64_t sa, sb, sc;      // elements
    int nk;                  // K / KT
    int tiles_m, tiles_n;
    int group_along_m;
    int accumulate;
};

__device__ __forceinline__ unsigned xcd_chunked_index_fast(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <bool CX, int TM, int TN>
__global__ __launch_bounds__(256, 2)
void gemm_fast_kernel(const FastArgs g) {
    constexpr int NP = CX ? 2 : 1;
    constexpr int ES = CX ? 2 : 1;
    constexpr int KT = CX ? 8 : 16;
    constexpr int KS = KT / 4;
    constexpr int NT = 256;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    // B rows: complex fragments are ds_read_b64 (row stride = 16 mod 32 doubles is conflict-free);
    // fp64 fragments are ds_read_b128 column PAIRS (row stride = 0 mod 32 is conflict-free)
    constexpr int SA = KT + 2, SB = CX ? BN + 16 : BN;
    constexpr int NA = BM * 8 / NT;                     // 16-byte items per thread, A stage (8 items per row)
    constexpr int IPR_B = CX ? BN : BN / 2;             // 16-byte items per B row
    static_assert(IPR_B % 64 == 0, "a wave must stay inside one B row");
    constexpr int WPR = IPR_B / 64;                     // waves per B row
    constexpr int RPS = NT / IPR_B;                     // B rows covered by one item step
    constexpr int NB = KT * IPR_B / NT;
    constexpr int A_STAGE = NP * BM * SA, B_STAGE = NP * KT * SB;
    constexpr size_t ESZ = 8 * ES;
    static_assert(KS % 2 == 0, "fragment double buffering needs an even k-step count");

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    unsigned w = xcd_chunked_index_fast(blockIdx.x, gridDim.x);
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

    // ---- global addressing: scalar row bases + loop-invariant lane offsets
    const char* a_ptr[NA];
    const char* b_ptr[NB];
    {
        const char* Ab = reinterpret_cast<const char*>(g.A + (b * g.sa + (int64_t)m0 * g.lda) * ES);
        const char* Bb = reinterpret_cast<const char*>(g.B + (b * g.sb + n0) * ES);
#pragma unroll
        for (int i = 0; i < NA; ++i) a_ptr[i] = Ab + (size_t)i * 32 * g.lda * ESZ;
        const int brow0 = wave / WPR;
#pragma unroll
        for (int i = 0; i < NB; ++i) b_ptr[i] = Bb + (size_t)(brow0 + i * RPS) * g.ldb * ESZ;
    }
    const unsigned voff_a = ((unsigned)(tid >> 3) * (unsigned)g.lda + (unsigned)(tid & 7) * (CX ? 1 : 2)) * (unsigned)ESZ;
    const unsigned voff_b = (unsigned)(tid % IPR_B) * 16u;
    const size_t a_step = KT * ESZ;
    const size_t b_step = (size_t)KT * g.ldb * ESZ;

    // ---- LDS addressing: one base per operand and direction, the rest immediates
    double* st_a = As + (tid >> 3) * SA + (tid & 7) * (CX ? 1 : 2);
    double* st_b = Bs + (wave / WPR) * SB + (tid % IPR_B) * (CX ? 1 : 2);
    const double* rd_a = As + (wm * 16 * TM + (lane & 15)) * SA + (lane >> 4);
    // fp64: lane c of n-tile pair (2jp, 2jp+1) owns the ADJACENT columns 32jp + 2c, 32jp + 2c + 1,
    // so one 16-byte LDS read feeds two MFMA tiles and the epilogue stores 16 bytes per lane
    const double* rd_b = Bs + (lane >> 4) * SB + wn * 16 * TN + (CX ? 1 : 2) * (lane & 15);

    f64x2 ra[NA], rb[NB];

    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[i] = *reinterpret_cast<const f64x2*>(a_ptr[i] + voff_a);
            a_ptr[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i] = *reinterpret_cast<const f64x2*>(b_ptr[i] + voff_b);
            b_ptr[i] += b_step;
        }
    };

    auto stash = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            double* d = st_a + buf * A_STAGE + i * 32 * SA;
            if constexpr (CX) { d[0] = ra[i][0]; d[BM * SA] = ra[i][1]; }
            else *reinterpret_cast<f64x2*>(d) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double* d = st_b + buf * B_STAGE + i * RPS * SB;
            if constexpr (CX) { d[0] = rb[i][0]; d[KT * SB] = rb[i][1]; }
            else *reinterpret_cast<f64x2*>(d) = rb[i];
        }
    };

    f64x4 acc[NP][TM][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[p][i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    auto read_frags = [&](auto buf_c, int kk, double (&af)[NP][TM], double (&bf)[NP][TN]) {
        constexpr int buf = decltype(buf_c)::value;
        const double* as = rd_a + buf * A_STAGE;
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

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    const int nk = g.nk;
    fetch();
    stash(B0{});
    __syncthreads();

    double a0[NP][TM], b0[NP][TN], a1[NP][TM], b1[NP][TN];
    if (nk > 1) fetch();
    read_frags(B0{}, 0, a0, b0);

    // one stage: k-steps 0 .. KS-2, [stash next stage], barrier, [fetch the stage
    // after next], [first fragments of the next stage], k-step KS-1.  The three
    // optional parts hang on wave-uniform conditions (scalar branches).
    auto stage = [&](auto cur_c, int t) {
        constexpr int cur = decltype(cur_c)::value;
        using NXT = std::integral_constant<int, cur ^ 1>;
        const bool has_next = t + 1 < nk, has_next2 = t + 2 < nk;
#pragma unroll
        for (int kk = 0; kk + 1 < KS; ++kk) {
            if ((kk & 1) == 0) { read_frags(cur_c, kk + 1, a1, b1); __builtin_amdgcn_sched_barrier(0); mfma_step(a0, b0); }
            else               { read_frags(cur_c, kk + 1, a0, b0); __builtin_amdgcn_sched_barrier(0); mfma_step(a1, b1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) stash(NXT{});
        __syncthreads();
        if (has_next2) fetch();
        if (has_next) read_frags(NXT{}, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int t = 0; t < nk; t += 2) {
        stage(B0{}, t);
        if (t + 1 < nk) stage(B1{}, t + 1);
    }

    // ---- epilogue: reg r of a lane -> row (lane>>4) + 4r of each 16-row block; 16 bytes per lane
    double* __restrict__ C = g.C + (b * g.sc + (int64_t)(m0 + wm * 16 * TM + (lane >> 4)) * g.ldc +
                                    n0 + wn * 16 * TN + (CX ? 1 : 2) * (lane & 15)) * ES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* crow = C + (int64_t)(i * 16 + 4 * r) * g.ldc * ES;
            if constexpr (CX) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f64x2* dst = reinterpret_cast<f64x2*>(crow + 2 * j * 16);
                    f64x2 v = f64x2{acc[0][i][j][r], acc[1][i][j][r]};
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            } else {
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    f64x2* dst = reinterpret_cast<f64x2*>(crow + jp * 32);
                    f64x2 v = f64x2{acc[0][i][2 * jp][r], acc[0][i][2 * jp + 1][r]};
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <bool CX, int TM, int TN>
static int launch_fast(const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb,
                       int64_t sc, int accumulate, int group_along_m, hipStream_t stream) {
    constexpr int KT = CX ? 8 : 16;
    constexpr int NP = CX ? 2 : 1;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    FastArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sa = sa; g.sb = sb; g.sc = sc;
    g.nk = (int)(k / KT);
    g.tiles_m = (int)(m / BM);
    g.tiles_n = (int)(n / BN);
    g.group_along_m = group_along_m;
    g.accumulate = accumulate ? 1 : 0;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * batch;
    if (nwg <= 0 || nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    const size_t lds = sizeof(double) * 2 * NP * (BM * (KT + 2) + KT * (CX ? BN + 16 : BN));
    auto kern = gemm_fast_kernel<CX, TM, TN>;
    static bool lds_opt_in = false;
    if (lds > 64 * 1024 && !lds_opt_in) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm_fast)");
        lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, g);
    return launch_status("gemm_fast launch");
}

int g_gemm_fast = 1;   // tuning knob: 0 routes everything through the general kernel

// Returns QS_OK after launching, or 1 when the product does not qualify for a
// fast shape (caller falls back to the general kernel).
int gemm_fast_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                  int64_t sb, int64_t sc, int accumulate, int group_along_m, hipStream_t stream) {
    if (!g_gemm_fast) return 1;
    const bool cx = dtype == QS_C128;
    const int64_t esz = cx ? 16 : 8;
    if (!aligned(A, 16) || !aligned(B, 16) || !aligned(C, cx ? 16 : 8)) return 1;
    if (!cx && ((lda & 1) || (ldb & 1) || (sa & 1) || (sb & 1))) return 1;   // 16-byte loads
    if (!cx && (!aligned(C, 16) || (ldc & 1) || (sc & 1))) return 1;          // 16-byte stores
    // the lane offset of the A loads is 32-bit: 32 rows of lda elements must fit
    if (32 * lda * esz + 256 >= (int64_t(1) << 32)) return 1;
#define QS_FAST(CXF, TMF, TNF)                                                                      \
    if (m % (32 * TMF) == 0 && n % (32 * TNF) == 0)                                                  \
        return launch_fast<CXF, TMF, TNF>(A, B, C, m, n, k, lda, ldb, ldc, batch, sa, sb, sc,        \
                                          accumulate, group_along_m, stream);
    if (!cx) {
        if (k % 16) return 1;
        QS_FAST(false, 4, 4)     // 128 x 128
        QS_FAST(false, 2, 4)     //  64 x 128
    } else {
        if (k % 8) return 1;
        if (m >= n) { QS_FAST(true, 4, 2) }   // 128 x 64
        QS_FAST(true, 2, 4)      //  64 x 128
        QS_FAST(true, 4, 2)      // 128 x  64
        QS_FAST(true, 2, 2)      //  64 x  64
    }
#undef QS_FAST
    return 1;
}

}  // namespace qs
