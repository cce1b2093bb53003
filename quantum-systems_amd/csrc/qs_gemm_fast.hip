// Exact-tiling fast path of the batched row-major product (see qs_gemm.hip
// for the general kernel and the contraction -> GEMM map).
//
// Why a second kernel: on gfx950 the fp64 MFMA shares the SIMD's vector issue
// with ordinary VALU work -- every VALU instruction in the K loop takes its
// issue cycles away from the matrix pipe (measured with tools/probe_mix.hip:
// 32 integer adds per 16 MFMAs cost 12 % of the MFMA rate, and two co-resident
// waves do not hide it).  The general kernel spends ~130 VALU instructions per
// K step on 64-bit address arithmetic and edge handling.  This kernel is for
// products whose extents are whole multiples of the tile (l = 128, 256, 512,
// ...); its K loop contains no VALU work at all:
//   * global loads use the scalar-base form (SGPR pointer + one loop-invariant
//     32-bit lane offset); the per-row bases advance on the scalar ALU;
//   * LDS addresses are one VGPR + immediates (the loop is unrolled by two so
//     the stage buffer is a compile-time constant);
//   * no bounds checks, no zero fill.
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
    unsigned total;          // tiles_m * tiles_n * batch = size of the virtual grid
    int group_along_m;
    int accumulate;
};

// Work item of virtual block `v` of a `total`-block grid: XCD x (= v % 8) owns
// the x-th contiguous chunk of the work list; bijective for every `total`.
__device__ __forceinline__ unsigned xcd_chunked_index_fast(unsigned v, unsigned total) {
    const unsigned xcd = v & 7u, slot = v >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// value known to be wave-uniform -> scalar registers
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// 16-byte load at scalar base + 32-bit lane offset (buffer form, no range limit in use)
__device__ __forceinline__ f64x2 load16(uint64_t base, unsigned lane_off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), (short)0,
                                                        (int)0xFFFFFFFF, 0x00020000);
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, 0, 0);
    return __builtin_bit_cast(f64x2, raw);
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
    static_assert(CX || TN % 2 == 0, "fp64 n-tiles come in column pairs");

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = g.nk;
    const unsigned P = gridDim.x;                       // persistent grid (multiple of 8 unless P == total)

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

    // ---- fetch cursor: scalar row bases of the tile being loaded + loop-invariant lane offsets
    // Row bases are kept as scalar 64-bit integers and the loads go through buffer
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
        for (int i = 0; i < NA; ++i) a_ptr[i] = uniform64(reinterpret_cast<uint64_t>(Ab + (size_t)i * 32 * g.lda * ESZ));
        const int brow0 = wave / WPR;
#pragma unroll
        for (int i = 0; i < NB; ++i) b_ptr[i] = uniform64(reinterpret_cast<uint64_t>(Bb + (size_t)(brow0 + i * RPS) * g.ldb * ESZ));
    };
    aim(f_v);
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

    // two staging register sets: the data of global stage s waits in set s & 1, so a load has
    // two stages (not one) to arrive from HBM before it is written to LDS
    f64x2 ra[2][NA], rb[2][NB];

    // load the cursor's stage into registers and advance the cursor (to the
    // next tile of this workgroup after the last k-stage)
    auto fetch = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[set][i] = load16(a_ptr[i], voff_a);
            a_ptr[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[set][i] = load16(b_ptr[i], voff_b);
            b_ptr[i] += b_step;
        }
        if (++f_k == nk) {
            f_k = 0;
            f_v += P;
            f_valid = f_v < g.total;
            if (f_valid) aim(f_v);
        }
    };

    auto stash = [&](auto buf_c) {   // stage data of parity `buf` -> LDS stage `buf`
        constexpr int buf = decltype(buf_c)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            double* d = st_a + buf * A_STAGE + i * 32 * SA;
            if constexpr (CX) { d[0] = ra[buf][i][0]; d[BM * SA] = ra[buf][i][1]; }
            else *reinterpret_cast<f64x2*>(d) = ra[buf][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double* d = st_b + buf * B_STAGE + i * RPS * SB;
            if constexpr (CX) { d[0] = rb[buf][i][0]; d[KT * SB] = rb[buf][i][1]; }
            else *reinterpret_cast<f64x2*>(d) = rb[buf][i];
        }
    };

    f64x4 acc[NP][TM][TN];

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
    // each 16-row block; 16 bytes per lane
    auto epilogue = [&](unsigned v) {
        int m0, n0; int64_t b;
        decode(v, m0, n0, b);
        double* __restrict__ C = g.C + (b * g.sc + (int64_t)(m0 + wm * 16 * TM + (lane >> 4)) * g.ldc +
                                        n0 + wn * 16 * TN + (CX ? 1 : 2) * (lane & 15)) * ES;
        auto store_all = [&](auto add_c) {
            constexpr bool add = decltype(add_c)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* crow = C + (int64_t)(i * 16 + 4 * r) * g.ldc * ES;
                    if constexpr (CX) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            f64x2* dst = reinterpret_cast<f64x2*>(crow + 2 * j * 16);
                            f64x2 v2 = f64x2{acc[0][i][j][r], acc[1][i][j][r]};
                            if constexpr (add) v2 += *dst;
                            *dst = v2;
                        }
                    } else {
#pragma unroll
                        for (int jp = 0; jp < TN / 2; ++jp) {
                            f64x2* dst = reinterpret_cast<f64x2*>(crow + jp * 32);
                            f64x2 v2 = f64x2{acc[0][i][2 * jp][r], acc[0][i][2 * jp + 1][r]};
                            if constexpr (add) v2 += *dst;
                            *dst = v2;
                        }
                    }
                }
            }
        };
        if (g.accumulate) store_all(std::true_type{}); else store_all(std::false_type{});
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
        // k-step 0 (fresh accumulators at the start of a tile)
        read_frags(cur_c, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (c_k == 0) mfma_step(a0, b0, T_{}); else mfma_step(a0, b0, F_{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 1; kk + 1 < KS; ++kk) {
            if ((kk & 1) == 0) { read_frags(cur_c, kk + 1, a1, b1); __builtin_amdgcn_sched_barrier(0); mfma_step(a0, b0, F_{}); }
            else               { read_frags(cur_c, kk + 1, a0, b0); __builtin_amdgcn_sched_barrier(0); mfma_step(a1, b1, F_{}); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) stash(NXT{});        // stage gs+1, loaded two stages ago
        __syncthreads();
        if (f_valid) fetch(NXT{});         // stage gs+3 into the set just written out
        if (has_next) read_frags(NXT{}, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a1, b1, F_{});
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

int g_gemm_fast = 1;           // tuning knob: 0 routes everything through the general kernel
int g_gemm_fast_persist = 1;   // tuning knob: 0 one workgroup per tile, 1 automatic, 2 always persistent

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
    const int64_t total = (int64_t)g.tiles_m * g.tiles_n * batch;
    if (total <= 0 || total >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    g.total = (unsigned)total;
    // two persistent workgroups per CU (the LDS and register budget admits exactly two)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            (void)hipGetLastError();
            n_cu = 256;
        } else {
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
    }
    int64_t P = 2 * (int64_t)n_cu;
    P -= P % 8;
    // How many tiles a workgroup walks (cross-tile prefetch hides the next tile's first loads
    // behind the current tile's last stages and epilogue):
    //   short tile lists (<= 8 per resident workgroup): fully persistent grid, +7 % at l = 64;
    //   long lists: 4 tiles per workgroup -- the dispatcher keeps balancing the grid, which a
    //   static split of a long list loses more on than the prefetch wins (-4 % at l = 256 when
    //   fully persistent, +1.5 % with 4 tiles; profiles/r01_gemm_notes.txt).
    // g_gemm_fast_persist: 0 one tile per workgroup, 1 this policy, 2 always fully
    // persistent, >= 3 that many tiles per workgroup.
    int64_t tiles_per_wg = 1;
    bool full = false;
    if (g_gemm_fast_persist == 1) { full = total <= 8 * P; tiles_per_wg = 4; }
    else if (g_gemm_fast_persist == 2) full = true;
    else if (g_gemm_fast_persist >= 3) tiles_per_wg = g_gemm_fast_persist;
    if (!full) {
        P = (total + tiles_per_wg - 1) / tiles_per_wg;
        P = (P + 7) & ~int64_t(7);
    }
    if (P > total) P = total;
    const size_t lds = sizeof(double) * 2 * NP * (BM * (KT + 2) + KT * (CX ? BN + 16 : BN));
    auto kern = gemm_fast_kernel<CX, TM, TN>;
    static bool lds_opt_in = false;
    if (lds > 64 * 1024 && !lds_opt_in) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm_fast)");
        lds_opt_in = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(256), lds, stream, g);
    return launch_status("gemm_fast launch");
}

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
