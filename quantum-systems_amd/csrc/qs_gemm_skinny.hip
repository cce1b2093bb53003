// Short-and-wide product  C (m x n) = A (m x k) . B (k x n)  with m <= 64 rows
// and n in the millions: the leading-index contraction of the sharded transform
// (sharded.py, replicated-u layout), where every GPU multiplies ITS rows of Ct
// into the whole tensor: m = l / G rows, k = l, n = l^3.
//
// At 8 (m = 32) to 16 (m = 64) flop per streamed byte this product is HBM-bound,
// so it is built as a streaming kernel rather than as a tiled GEMM:
//   * A (m x k, at most 128 KB) is loaded into LDS once per workgroup and stays;
//   * B is never staged in LDS: each lane loads its MFMA B fragments straight
//     from global memory (16 bytes per lane; lanes 0-15 read 256 contiguous
//     bytes of row k, lanes 16-31 of row k+1, ...), eight k-steps ahead of use;
//   * a wave owns all m rows of a 64-column (fp64: two column-pair tiles) or
//     32-column (complex128) strip, so every byte of B is read exactly once.
// Algorithmic bytes: e*k*n read + e*m*n written; roofline = HBM.

#include "qs_common.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct SkinnyArgs {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;   // elements
    int k;                   // multiple of 4
    int64_t n;               // multiple of the workgroup strip
};

// TMS = m / 16 (1..4).  fp64: wave strip = 64 columns (4 n-tiles as 2 column pairs);
// complex128: wave strip = 32 columns (2 n-tiles).
template <bool CX, int TMS>
__global__ __launch_bounds__(256, 2)
void gemm_skinny_kernel(const SkinnyArgs g) {
    constexpr int NP = CX ? 2 : 1;
    constexpr int ES = CX ? 2 : 1;
    constexpr int M = 16 * TMS;
    constexpr int NTL = CX ? 2 : 4;            // n-tiles per wave
    constexpr int WSTRIP = 16 * NTL;           // columns per wave
    constexpr int NLD = 2;                     // 16-byte loads per lane per k-step
    constexpr int DEPTH = 8;                   // k-steps of B kept in flight

    extern __shared__ __attribute__((aligned(16))) double smem[];   // [NP][M][SA]
    const int K = g.k;
    const int SA = K + 2;                      // (SA/2) odd for K % 4 == 0: conflict-free fragment reads

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- A -> LDS (once)
    for (int idx = tid; idx < M * K; idx += 256) {
        const int r = idx / K, c = idx % K;
        if constexpr (CX) {
            const f64x2 v = *reinterpret_cast<const f64x2*>(g.A + ((int64_t)r * g.lda + c) * 2);
            smem[r * SA + c] = v[0];
            smem[M * SA + r * SA + c] = v[1];
        } else {
            smem[r * SA + c] = g.A[(int64_t)r * g.lda + c];
        }
    }
    __syncthreads();

    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * WSTRIP;   // first column of this wave
    const double* a_rd = smem + (lane & 15) * SA + (lane >> 4);

    // lane's B pointers: row (lane>>4), 16 bytes at column n0 + (fp64: 32*jp + 2c | complex: 16*j + c)
    const double* bp[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int64_t col = CX ? n0 + 16 * q + (lane & 15) : n0 + 32 * q + 2 * (lane & 15);
        bp[q] = g.B + ((int64_t)(lane >> 4) * g.ldb + col) * ES;
    }
    const int64_t bstep = 4 * g.ldb * ES;     // doubles per k-step

    f64x4 acc[NP][TMS][NTL];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TMS; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[p][i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    f64x2 ring[DEPTH][NLD];
    const int nks = K / 4;
    auto issue = [&](int slot) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            ring[slot][q] = *reinterpret_cast<const f64x2*>(bp[q]);
            bp[q] += bstep;
        }
    };
    auto consume = [&](int slot, int ks) {
        double af[NP][TMS];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TMS; ++i) af[p][i] = a_rd[p * M * SA + i * 16 * SA + ks * 4];
#pragma unroll
        for (int i = 0; i < TMS; ++i) {
            if constexpr (!CX) {
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    acc[0][i][2 * q] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], ring[slot][q][0], acc[0][i][2 * q], 0, 0, 0);
                    acc[0][i][2 * q + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], ring[slot][q][1], acc[0][i][2 * q + 1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    const double br = ring[slot][j][0], bi = ring[slot][j][1];
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], br, acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bi, acc[1][i][j], 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-af[1][i], bi, acc[0][i][j], 0, 0, 0);
                    acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][i], br, acc[1][i][j], 0, 0, 0);
                }
            }
        }
    };

    // prime the ring, then consume slot s of each round while refilling it DEPTH steps ahead
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
        if (s < nks) issue(s);
    for (int base = 0; base < nks; base += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int ks = base + s;
            if (ks < nks) {
                consume(s, ks);
                if (ks + DEPTH < nks) issue(s);
            }
        }
    }

    // ---- epilogue: reg r of a lane -> row (lane>>4) + 4r of each 16-row block, 16 bytes per lane
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + (lane >> 4) + 4 * r;
            double* crow = g.C + (int64_t)row * g.ldc * ES;
            if constexpr (CX) {
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    *reinterpret_cast<f64x2*>(crow + (n0 + 16 * j + (lane & 15)) * 2) =
                        f64x2{acc[0][i][j][r], acc[1][i][j][r]};
            } else {
#pragma unroll
                for (int q = 0; q < NLD; ++q)
                    *reinterpret_cast<f64x2*>(crow + n0 + 32 * q + 2 * (lane & 15)) =
                        f64x2{acc[0][i][2 * q][r], acc[0][i][2 * q + 1][r]};
            }
        }
    }
}

template <bool CX, int TMS>
static int launch_skinny(const double* A, const double* B, double* C, int64_t n, int64_t k,
                         int64_t lda, int64_t ldb, int64_t ldc, hipStream_t stream) {
    constexpr int NP = CX ? 2 : 1;
    constexpr int M = 16 * TMS;
    constexpr int WGSTRIP = 4 * (CX ? 32 : 64);
    SkinnyArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k = (int)k; g.n = n;
    const int64_t nwg = n / WGSTRIP;
    if (nwg <= 0 || nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    const size_t lds = sizeof(double) * NP * M * (k + 2);
    auto kern = gemm_skinny_kernel<CX, TMS>;
    static PerDeviceLds lds_opt_in;
    if (int rc = opt_in_dynamic_lds((const void*)kern, lds, lds_opt_in, "hipFuncSetAttribute(gemm_skinny)")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, g);
    note_dispatch("qs::gemm_skinny_kernel<%s, %d>", CX ? "true" : "false", TMS);
    return launch_status("gemm_skinny launch");
}


// QS_OK / error after launching, 1 = not eligible (caller falls back).
int gemm_skinny_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                    int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int accumulate,
                    hipStream_t stream) {
    if (!g_tune.gemm_skinny || batch != 1 || accumulate) return 1;
    const bool cx = dtype == QS_C128;
    // m = 64 is MFMA-bound (16 flop/B) and runs better on the tiled kernel (9.8 vs 15.3 ms at
    // l = 256); the streaming form is used where the product is HBM-bound: m <= 32 rows
    if (m > 32 || (m & 15) || (k & 3) || k < 4) return 1;
    if (n < (int64_t(1) << 16)) return 1;                       // only worth it for a long stream
    if (n % (cx ? 128 : 256)) return 1;
    if (!aligned(B, 16) || !aligned(C, 16) || !aligned(A, cx ? 16 : 8)) return 1;
    if (!cx && ((ldb & 1) || (ldc & 1))) return 1;
    const int64_t lds = (cx ? 16 : 8) * m * (k + 2);
    if (lds > 150 * 1024) return 1;
    switch (m / 16) {
        case 1: return cx ? launch_skinny<true, 1>(A, B, C, n, k, lda, ldb, ldc, stream)
                          : launch_skinny<false, 1>(A, B, C, n, k, lda, ldb, ldc, stream);
        case 2: return cx ? launch_skinny<true, 2>(A, B, C, n, k, lda, ldb, ldc, stream)
                          : launch_skinny<false, 2>(A, B, C, n, k, lda, ldb, ldc, stream);
        case 3: return cx ? launch_skinny<true, 3>(A, B, C, n, k, lda, ldb, ldc, stream)
                          : launch_skinny<false, 3>(A, B, C, n, k, lda, ldb, ldc, stream);
        case 4: return cx ? launch_skinny<true, 4>(A, B, C, n, k, lda, ldb, ldc, stream)
                          : launch_skinny<false, 4>(A, B, C, n, k, lda, ldb, ldc, stream);
        default: return 1;
    }
}

}  // namespace qs
