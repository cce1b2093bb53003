// Two contractions in one pass for small bases:  Z[s] = B^T . X[s] . B  for a stack of
// L x L slabs X[s] (contiguous), B (L x M), L, M <= 64, fp64.
//
// This is the pair (d, c) of the four-index transform,
//     T2[ab][r, s] = sum_c C[c, r] ( sum_d u[ab][c, d] C[d, s] ),
// with X[s] = u[a, b, :, :] and B = C (the c contraction uses C^T, not the bra): the slab u[a, b] is contiguous in
// memory, so the pair reads the tensor once and writes it once instead of twice
// each (basis_set.py:341-344).  Below l ~ 100 the transform is bound by its passes
// over the tensor and by per-tile overhead, not by the matrix pipe
// (profiles/r01_gemm_notes.txt), so a pass saved is time saved.
//
// One wave owns a slab.  What makes the chaining free: the accumulator layout of
// v_mfma_f64_16x16x4_f64 (register r of a lane = row (lane>>4) + 4r of a 16-row
// tile, column lane&15) IS its B-operand layout for k = that row -- register r of
// row tile i of Y = X.B is the B fragment of k-step 4i + r of A.Y.  So Y never
// leaves the accumulators: no LDS round trip, no shuffle.
//   * B lives in LDS once per workgroup as ready-made MFMA fragments (fragment-major:
//     one ds_read_b64 per fragment, lane-linear, conflict-free), zero-padded to whole
//     tiles.  ONE table serves both products: the A-operand fragment of B^T for row
//     tile i and k-step kk (lane -> B^T[16i + c][4kk + g] = B[4kk + g][16i + c]) is
//     the B-operand fragment of B for k-step kk and column tile i;
//   * X fragments come straight from global memory (the A-operand layout wants
//     X[row][k] per lane: 8-byte loads, 16 rows x 32 bytes per instruction, the
//     rest of every line is picked up by the following k-steps from L1/L2); the
//     register of (row tile, k-step) is refilled with the next slab of the wave
//     as soon as it has been consumed;
//   * rows >= L of X lie past the slab's buffer descriptor and read 0, columns >= L
//     (which alias the next row) are zeroed with a select where the fragment is used,
//     so nothing non-finite can leak; Z goes out through a per-slab buffer descriptor whose range check drops
//     rows >= M, lanes with a column >= M store to an offset past it.
// Same k order as the tiled kernels (d ascending, then c ascending).
// Algorithmic bytes per launch: 8 (L^2 + M^2) per slab; roofline = HBM below l ~ 64.

#include "qs_common.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct SlabArgs {
    const double* X;
    const double* B;   // (L, M) row-major
    double* Z;
    int L, M;
    unsigned nslabs;
};

// TL = ceil(L / 16), TM = ceil(M / 16)
template <int TL, int TM>
__global__ __launch_bounds__(256, 1)
void slab_pair_kernel(const SlabArgs g) {
    constexpr int KS = 4 * TL;                    // k-steps of either product (K = L padded to 16 TL)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* bfrag = lds;                          // [KS][TM][64]:  B[4kk + g4][16j + c16]
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};

    for (int f = tid; f < KS * TM * 64; f += 256) {
        const int ln = f & 63, j = (f >> 6) % TM, kk = (f >> 6) / TM;
        const int k = 4 * kk + (ln >> 4), col = 16 * j + (ln & 15);
        bfrag[f] = (k < L && col < M) ? g.B[(int64_t)k * M + col] : 0.0;
    }
    __syncthreads();

    const unsigned w = blockIdx.x * 4 + wave, W = gridDim.x * 4;
    if (w >= g.nslabs) return;

    const unsigned x_slab = (unsigned)(L * L * 8), z_slab = (unsigned)(M * M * 8);
    auto rsrc = [&](const double* base, unsigned slab, unsigned stride, unsigned room) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base) + (uint64_t)slab * stride;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 (int)room, 0x00020000);
    };
    // lane offsets of the X fragments: row 16 i + c16, column 4 kk + g4; tails are zeroed after the load
    unsigned x_off[TL];
#pragma unroll
    for (int i = 0; i < TL; ++i) x_off[i] = (unsigned)((16 * i + c16) * L + g4) * 8u;
    // A slab's descriptor covers the slab only: rows >= L lie past it and read 0.  A column >= L aliases
    // the next row of the slab; it is zeroed with a select where the fragment is USED -- a select at the
    // load would make the wave wait for the load it has just issued.
    auto load_x = [&](auto rs, int i, int kk) __attribute__((always_inline)) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(x_off[i] + kk * 32), 0, 0));
    };

    double xr[TL][KS];
    {
        const auto rs = rsrc(g.X, w, x_slab, x_slab);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int i = 0; i < TL; ++i) xr[i][kk] = load_x(rs, i, kk);
    }

    for (unsigned s = w; s < g.nslabs; s += W) {
        const unsigned nxt = s + W < g.nslabs ? s + W : s;       // the last slab re-reads itself (unused)
        const auto rs_next = rsrc(g.X, nxt, x_slab, x_slab);

        // ---- Y = X . B : row tiles of X, all column tiles of B
        f64x4 Y[TL][TM];      // (k-step 0 starts every accumulator from the literal zero)
        // (fragments of k-step kk + 1 are read from LDS while k-step kk multiplies; k-steps that lie
        // wholly in the padding, 4 kk >= L, are skipped: wave-uniform branches)
        double bf[2][TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) bf[0][j] = bfrag[j * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk == 0 || 4 * kk < L) {          // (k-step 0 always runs: it also zeroes the accumulators)
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < TM; ++j) bf[(kk + 1) & 1][j] = bfrag[((kk + 1) * TM + j) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool k_ok = 4 * kk + g4 < L;
#pragma unroll
                for (int i = 0; i < TL; ++i) {
                    const double xa = k_ok ? xr[i][kk] : 0.0;
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        Y[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, bf[kk & 1][j], kk == 0 ? zero4 : Y[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TL; ++i) xr[i][kk] = load_x(rs_next, i, kk);    // same register, next slab
            }
        }

        // ---- Z = A . Y : register r of row tile i of Y is the B fragment of k-step 4 i + r
        f64x4 Zt[TM][TM];
        double af[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = bfrag[i * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk == 0 || 4 * kk < L) {          // (k-step 0 always runs: it also zeroes the accumulators)
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = bfrag[((kk + 1) * TM + i) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        Zt[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1][i], Y[kk >> 2][j][kk & 3], kk == 0 ? zero4 : Zt[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- store: reg r of a lane -> row 16 i + g4 + 4 r, column 16 j + c16
        const auto rs_z = rsrc(g.Z, s, z_slab, z_slab);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int col = 16 * j + c16;
            const unsigned o0 = col < M ? (unsigned)(g4 * M + col) * 8u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = o0 + (unsigned)((16 * i + 4 * r) * M) * 8u;
                    // (through a scalar temporary: __builtin_bit_cast applied directly to the vector
                    // element Zt[i][j][r] reads element 0 for every r with this compiler)
                    const double z = Zt[i][j][r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, z), rs_z, (int)o, 0, 0);
                }
            }
        }
    }
}

// Two waves per slab: each wave takes half of the COLUMN tiles of B, Y and Z (the two products are
// column-wise independent), reads all of X itself (the second reader hits L1/L2) and keeps only a
// four-k-step ring of X fragments -- about 200 registers, so two waves share a SIMD and one wave's
// waits (loads at a slab's start, the stores at its end) are covered by the other's MFMAs.  The ring
// runs across slab boundaries (k-step kk + 4 of this slab, or kk + 4 - KS of the wave's next one,
// always lands in slot kk & 3 because KS is a multiple of 4).
template <int TL, int TM>
__global__ __launch_bounds__(256, 2)
void slab_pair_split_kernel(const SlabArgs g) {
    static_assert(TM % 2 == 0, "column tiles are split between two waves");
    constexpr int KS = 4 * TL, TH = TM / 2, R = 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* bfrag = lds;                          // [KS][TM][64]
    const int L = g.L, M = g.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const int half = wave & 1, j0 = half * TH;

    for (int f = tid; f < KS * TM * 64; f += 256) {
        const int ln = f & 63, j = (f >> 6) % TM, kk = (f >> 6) / TM;
        const int k = 4 * kk + (ln >> 4), col = 16 * j + (ln & 15);
        bfrag[f] = (k < L && col < M) ? g.B[(int64_t)k * M + col] : 0.0;
    }
    __syncthreads();

    const unsigned w = blockIdx.x * 2 + (wave >> 1), W = gridDim.x * 2;
    if (w >= g.nslabs) return;

    const unsigned x_slab = (unsigned)(L * L * 8), z_slab = (unsigned)(M * M * 8);
    auto rsrc = [&](const double* base, unsigned slab, unsigned stride, unsigned room) __attribute__((always_inline)) {
        const uint64_t p = reinterpret_cast<uint64_t>(base) + (uint64_t)slab * stride;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                                 (int)room, 0x00020000);
    };
    unsigned x_off[TL];
#pragma unroll
    for (int i = 0; i < TL; ++i) x_off[i] = (unsigned)((16 * i + c16) * L + g4) * 8u;
    auto load_x = [&](auto rs, int i, int kk) __attribute__((always_inline)) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(x_off[i] + kk * 32), 0, 0));
    };

    double xr[TL][R];
    {
        const auto rs = rsrc(g.X, w, x_slab, x_slab);
#pragma unroll
        for (int kk = 0; kk < R; ++kk)
#pragma unroll
            for (int i = 0; i < TL; ++i) xr[i][kk] = load_x(rs, i, kk);
    }

    for (unsigned s = w; s < g.nslabs; s += W) {
        const unsigned nxt = s + W < g.nslabs ? s + W : s;
        const auto rs_cur = rsrc(g.X, s, x_slab, x_slab);
        const auto rs_next = rsrc(g.X, nxt, x_slab, x_slab);

        f64x4 Y[TL][TH];      // (k-step 0 starts every accumulator from the literal zero)
        double bf[2][TH];
#pragma unroll
        for (int j = 0; j < TH; ++j) bf[0][j] = bfrag[(j0 + j) * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < KS) {
#pragma unroll
                for (int j = 0; j < TH; ++j) bf[(kk + 1) & 1][j] = bfrag[((kk + 1) * TM + j0 + j) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0 || 4 * kk < L) {               // (a k-step wholly in the padding multiplies nothing)
                const bool k_ok = 4 * kk + g4 < L;
#pragma unroll
                for (int i = 0; i < TL; ++i) {
                    const double xa = k_ok ? xr[i][kk & 3] : 0.0;
#pragma unroll
                    for (int j = 0; j < TH; ++j)
                        Y[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, bf[kk & 1][j], kk == 0 ? zero4 : Y[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot: four k-steps ahead, in this slab or at the start of the wave's next one
#pragma unroll
            for (int i = 0; i < TL; ++i)
                xr[i][kk & 3] = kk + R < KS ? load_x(rs_cur, i, kk + R) : load_x(rs_next, i, kk + R - KS);
        }

        f64x4 Zt[TM][TH];
        double af[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = bfrag[i * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < KS) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = bfrag[((kk + 1) * TM + i) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0 || 4 * kk < L) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TH; ++j)
                        Zt[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1][i], Y[kk >> 2][j][kk & 3], kk == 0 ? zero4 : Zt[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        const auto rs_z = rsrc(g.Z, s, z_slab, z_slab);
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            const int col = 16 * (j0 + j) + c16;
            const unsigned o0 = col < M ? (unsigned)(g4 * M + col) * 8u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = o0 + (unsigned)((16 * i + 4 * r) * M) * 8u;
                    const double z = Zt[i][j][r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, z), rs_z, (int)o, 0, 0);
                }
            }
        }
    }
}


template <int TL, int TM>
static int launch_slab_pair(const SlabArgs& g, hipStream_t stream) {
    const int n_cu = device_cu_count();
    const size_t lds = sizeof(double) * (4 * TL) * TM * 64;
    if constexpr (TM % 2 == 0) {
        if (g_tune.slab_pair != 2) {
            // two waves per slab, two workgroups (four slabs in flight) per CU
            int64_t wgs = cdiv(g.nslabs, 2);
            if (wgs > 2 * (int64_t)n_cu) wgs = 2 * (int64_t)n_cu;
            hipLaunchKernelGGL((slab_pair_split_kernel<TL, TM>), dim3((unsigned)wgs), dim3(256), lds, stream, g);
            note_dispatch("qs::slab_pair_split_kernel<%d, %d>", TL, TM);
            return launch_status("slab_pair_split launch");
        }
    }
    int64_t wgs = cdiv(g.nslabs, 4);
    if (wgs > n_cu) wgs = n_cu;                       // one wave per SIMD, persistent
    hipLaunchKernelGGL((slab_pair_kernel<TL, TM>), dim3((unsigned)wgs), dim3(256), lds, stream, g);
    note_dispatch("qs::slab_pair_kernel<%d, %d>", TL, TM);
    return launch_status("slab_pair launch");
}

// Z[s] = B^T . X[s] . B for s < nslabs; QS_OK / error after launching, 1 = not eligible.
int slab_pair_try(int dtype, const void* X, const void* B, void* Z, int64_t nslabs, int64_t L, int64_t M,
                  hipStream_t stream) {
    if (!g_tune.slab_pair || dtype != QS_F64) return 1;
    if (L < 1 || M < 1 || L > 64 || M > 64) return 1;
    if (nslabs < 256 || nslabs >= (int64_t(1) << 31)) return 1;    // enough slabs to occupy the waves
    SlabArgs g;
    g.X = (const double*)X; g.B = (const double*)B; g.Z = (double*)Z;
    g.L = (int)L; g.M = (int)M; g.nslabs = (unsigned)nslabs;
    const int tl = (int)cdiv(L, 16), tm = (int)cdiv(M, 16);
#define QS_SLAB(TLV)                                                  \
    case TLV:                                                         \
        switch (tm) {                                                 \
            case 1: return launch_slab_pair<TLV, 1>(g, stream);       \
            case 2: return launch_slab_pair<TLV, 2>(g, stream);       \
            case 3: return launch_slab_pair<TLV, 3>(g, stream);       \
            default: return launch_slab_pair<TLV, 4>(g, stream);      \
        }
    switch (tl) {
        QS_SLAB(1)
        QS_SLAB(2)
        QS_SLAB(3)
        default:
            switch (tm) {
                case 1: return launch_slab_pair<4, 1>(g, stream);
                case 2: return launch_slab_pair<4, 2>(g, stream);
                case 3: return launch_slab_pair<4, 3>(g, stream);
                default: return launch_slab_pair<4, 4>(g, stream);
            }
    }
#undef QS_SLAB
}

}  // namespace qs
