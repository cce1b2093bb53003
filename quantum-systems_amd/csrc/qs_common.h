// Shared host-side helpers for the gfx950 basis-transformation library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qs_amd.h"

namespace qs {

// Record the text of a failed HIP call for qs_last_hip_error().
void note_hip_error(hipError_t e, const char* what);

inline int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return QS_OK;
    note_hip_error(e, what);
    return QS_ERR_HIP;
}

// Launch check: kernels are asynchronous, so this only catches launch-time
// failures (bad configuration, missing code object); it never synchronises.
inline int launch_status(const char* what) {
    return hip_status(hipGetLastError(), what);
}

inline bool aligned(const void* p, size_t a) {
    return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
}

inline size_t elem_size(int dtype) { return dtype == QS_C128 ? 16 : 8; }

inline bool dtype_ok(int dtype) { return dtype == QS_F64 || dtype == QS_C128; }

// Compute units of the current device (cached: one process drives one GPU; the property query
// costs far more than a kernel launch).
int device_cu_count();

// ceil division for positive operands
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Internal launchers (defined in qs_gemm_f64.hip / qs_gemm_c128.hip).
int gemm_f64(const double* A, const double* B, double* C, int64_t m, int64_t n,
             int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch,
             int64_t sa, int64_t sb, int64_t sc, int accumulate, hipStream_t stream);
int gemm_c128(const double* A, const double* B, double* C, int64_t m, int64_t n,
              int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch,
              int64_t sa, int64_t sb, int64_t sc, int accumulate, hipStream_t stream);

// VALU-free fast path, exact and edge forms (qs_gemm_fast.hip): QS_OK / error after launching, 1 = not eligible.
int gemm_fast_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                  int64_t sb, int64_t sc, int accumulate, int group_along_m, hipStream_t stream);

// Small-coefficient streaming product, m, k <= 64 (qs_gemm_stream.hip): same return convention.
int gemm_stream_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                    int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                    int64_t sb, int64_t sc, int accumulate, hipStream_t stream);

// Fused pair of contractions on contiguous L x L slabs, L, M <= 64 (qs_slab_pair.hip): Z[s] = B^T.X[s].B.
int slab_pair_try(int dtype, const void* X, const void* B, void* Z, int64_t nslabs, int64_t L, int64_t M,
                  hipStream_t stream);

// Short-and-wide streaming product (qs_gemm_skinny.hip): same return convention.
int gemm_skinny_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                    int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int accumulate,
                    hipStream_t stream);

// out (cols, rows) = in (rows, cols)^T, element = 8 or 16 bytes (tiny helper
// for the coefficient matrices).
int transpose_small(int dtype, const void* in, void* out, int64_t rows,
                    int64_t cols, hipStream_t stream);

// Bandwidth kernels (qs_permute.hip).
int antisymmetrize(int dtype, const void* u, void* out, int64_t npq, int64_t l, hipStream_t stream);
int spin_expand(int in_dtype, int out_dtype, const void* u, void* out, int64_t l, int64_t p_lo,
                int64_t p_hi, int as, hipStream_t stream);
int kron_eye2(int in_dtype, int out_dtype, const void* h, void* out, int64_t nmat, int64_t l,
              hipStream_t stream);
int spin2_two_body(const void* S, void* out, int64_t n, int64_t p_lo, int64_t p_hi, int as,
                   hipStream_t stream);

// Tuning knobs (qs_tuning_set).
extern int g_gemm_f64_cfg;
extern int g_gemm_c128_cfg;
extern int g_gemm_pipe;
extern int g_gemm_fast;
extern int g_gemm_fast_persist;
extern int g_gemm_fast_shape;
extern int g_gemm_skinny;
extern int g_gemm_stream;
extern int g_slab_pair;

}  // namespace qs
