// extern "C" surface of libqs_amd.so (include/qs_amd.h): argument checking,
// workspace carving and the contraction schedule.  No kernels live here.
//
// contraction -> GEMM map for  out = Ct Ct u C C  (basis_set.py:341-348), all
// operands row-major, L = old size, M = new size:
//   d:  T1[(abc), s]   = sum_d u[(abc), d]  C[d, s]        m=rows*L^2 n=M   k=L
//   c:  T2[ab][r, s]   = sum_c CT[r, c]     T1[ab][c, s]   batch rows*L, m=n=M, k=L
//   b:  T3[a][q, (rs)] = sum_b Ct[q, b]     T2[a][b, (rs)] batch rows,   m=M n=M^2 k=L
//   a:  out[p, (qrs)]  = sum_a Ct[p, a]     T3[a, (qrs)]   m=M n=M^3 k=L
// CT = C^T is materialised once (L*M elements) so that every product is the
// same row-major kernel with the tensor streamed along its contiguous axis.

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "qs_common.h"

namespace qs {

static thread_local char g_hip_err[256] = "";

void note_hip_error(hipError_t e, const char* what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return (dev >= 0 && dev < kMaxDevices) ? dev : 0;
}

int device_cu_count() {
    static PerDeviceInt n_cu;
    const int dev = current_device();
    int n = n_cu.v[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
            n = prop.multiProcessorCount;
        } else {
            (void)hipGetLastError();
            n = 256;   // MI355X
        }
        n_cu.v[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int opt_in_dynamic_lds(const void* kern, size_t lds_bytes, PerDeviceLds& seen, const char* what) {
    if (lds_bytes <= 64 * 1024) return QS_OK;
    std::atomic<uint32_t>& top = seen.bytes[current_device()];
    if (top.load(std::memory_order_acquire) >= lds_bytes) return QS_OK;
    // slow path (a first launch, or one that needs more than any before it): serialised, so that the attribute and the
    // record of it cannot end up in different orders when two threads raise the limit at once
    static std::mutex raise;
    std::lock_guard<std::mutex> hold(raise);
    if (top.load(std::memory_order_relaxed) >= lds_bytes) return QS_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return hip_status(e, what);
    top.store((uint32_t)lds_bytes, std::memory_order_release);
    return QS_OK;
}

int resident_workgroups(const void* kern, PerDeviceInt& cache, int fallback) {
    const int dev = current_device();
    int n = cache.v[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = fallback;
        }
        n = nb > 4 ? 4 : nb;
        cache.v[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

thread_local Tuning g_tune;

static thread_local char g_dispatch[4096] = "";
static thread_local size_t g_dispatch_len = 0;

void dispatch_reset() {
    g_dispatch[0] = 0;
    g_dispatch_len = 0;
}

void note_dispatch(const char* fmt, ...) {
    char name[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof(name), fmt, ap);
    va_end(ap);
    // "name xN" run-length form: the four contractions of a large transform are one kernel
    const size_t nl = strlen(name);
    char* last = strrchr(g_dispatch, ';');
    last = last ? last + 1 : g_dispatch;
    if (g_dispatch_len && !strncmp(last, name, nl) && (last[nl] == 0 || !strncmp(last + nl, " x", 2))) {
        const int count = last[nl] ? atoi(last + nl + 2) + 1 : 2;
        const size_t room = sizeof(g_dispatch) - (size_t)(last - g_dispatch);
        if (nl + 16 < room) {
            snprintf(last + nl, room - nl, " x%d", count);
            g_dispatch_len = strlen(g_dispatch);
        }
        return;
    }
    if (g_dispatch_len + nl + 2 >= sizeof(g_dispatch)) return;
    if (g_dispatch_len) g_dispatch[g_dispatch_len++] = ';';
    memcpy(g_dispatch + g_dispatch_len, name, nl + 1);
    g_dispatch_len += nl;
}

static int gemm(int dtype, const void* A, const void* B, void* C, int64_t m, int64_t n, int64_t k,
                int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb,
                int64_t sc, hipStream_t s, int accumulate = 0) {
    if (dtype == QS_F64)
        return gemm_f64((const double*)A, (const double*)B, (double*)C, m, n, k, lda, ldb, ldc, batch,
                        sa, sb, sc, accumulate, s);
    return gemm_c128((const double*)A, (const double*)B, (double*)C, m, n, k, lda, ldb, ldc, batch,
                     sa, sb, sc, accumulate, s);
}

int matmul_real_by_complex(const void* A, const void* B, void* out, int64_t m, int64_t n, int64_t k, int64_t lda,
                           int64_t ldb, int64_t ldc, hipStream_t stream) {
    return gemm_f64((const double*)A, (const double*)B, (double*)out, m, 2 * n, k, lda, 2 * ldb, 2 * ldc, 1, 0, 0, 0, 0,
                    stream);
}

// the d contraction: T1[(abc), s] = u[(abc), d] C[d, s]; `in_dtype` is the tensor's type, `dtype` that of C and T1
static int gemm_d(int in_dtype, int dtype, const void* u, const void* C, void* T1, int64_t rows3, int64_t L, int64_t M,
                  hipStream_t s) {
    if (in_dtype == dtype) return gemm(dtype, u, C, T1, rows3, M, L, L, M, M, 1, 0, 0, 0, s);
    return matmul_real_by_complex(u, C, T1, rows3, M, L, L, M, M, s);
}

static inline int64_t even_up(int64_t x) { return (x + 1) & ~int64_t(1); }

// Extents for which every product keeps its n and grid inside 32 bits.
static bool extents_ok(int64_t L, int64_t M) {
    if (L <= 0 || M <= 0) return false;
    if (L > 4096 || M > 1024) return false;   // M^3 < 2^31 needs M <= 1290
    return true;
}

static inline char* at(void* base, int64_t elems, size_t es) { return (char*)base + (size_t)elems * es; }

// contractions d, c, b on `rows` leading-index rows.  `CT` is scratch for C^T: the c contraction of
// the tiled path multiplies with it; the fused small-basis pass reads C itself and needs no transpose.
static int contract_dcb(int in_dtype, int dtype, const void* u, const void* C, void* CT, const void* Ct,
                        void* T1, void* T2, void* T3, int64_t rows, int64_t L, int64_t M,
                        hipStream_t s) {
    // small bases: d and c in one pass over the tensor (each slab u[a, b] is contiguous):
    //   T2[ab] = C^T . u[ab] . C   on the 4-wide matrix instruction, else on the 16-wide one
    int rc = 1;
    if (in_dtype == dtype && (g_tune.sandwich == 1 || g_tune.sandwich == 2 || g_tune.sandwich == 4 || g_tune.sandwich == 5))
        rc = sandwich4_try(dtype, u, T2, C, M, 1, C, 1, M, rows * L, L, M, L * L, L, 1, M * M, M, 1, s);
    if (rc == 1 && in_dtype == dtype) rc = slab_pair_try(dtype, u, C, T2, rows * L, L, M, s);
    if (rc == 1) {
        rc = transpose_small(dtype, C, CT, L, M, s);
        if (rc) return rc;
        rc = gemm_d(in_dtype, dtype, u, C, T1, rows * L * L, L, M, s);
        if (rc) return rc;
        rc = gemm(dtype, CT, T1, T2, M, M, L, L, M, M, rows * L, 0, L * M, M * M, s);
    }
    if (rc) return rc;
    return gemm(dtype, Ct, T2, T3, M, M * M, L, L, M * M, M * M, rows, 0, L * M * M, M * M * M, s);
}

}  // namespace qs

using namespace qs;

extern "C" {

int qs_abi_version(void) { return QS_ABI_VERSION; }

const char* qs_error_string(int code) {
    switch (code) {
        case QS_OK: return "ok";
        case QS_ERR_BAD_EXTENT: return "bad extent (non-positive, inconsistent or too large)";
        case QS_ERR_NULL_POINTER: return "null pointer";
        case QS_ERR_MISALIGNED: return "pointer not aligned to its element size";
        case QS_ERR_WORKSPACE: return "workspace too small";
        case QS_ERR_HIP: return "HIP runtime error";
        case QS_ERR_BAD_DTYPE: return "unsupported dtype code";
        case QS_ERR_ALIAS: return "output aliases an input";
        case QS_ERR_COMM: return "RCCL error";
        default: return "unknown error code";
    }
}

const char* qs_last_hip_error(void) { return g_hip_err; }

const char* qs_last_dispatch(void) { return g_dispatch; }

int qs_tuning_reset(void) {
    g_tune = Tuning();
    return QS_OK;
}

int qs_tuning_set(const char* key, int64_t value) {
    if (!key) return QS_ERR_NULL_POINTER;
    if (!strcmp(key, "gemm_f64_cfg")) { g_tune.gemm_f64_cfg = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_c128_cfg")) { g_tune.gemm_c128_cfg = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_pipe")) { g_tune.gemm_pipe = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_fast")) { g_tune.gemm_fast = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_fit")) { g_tune.gemm_fit = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_skinny")) { g_tune.gemm_skinny = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_stream")) { g_tune.gemm_stream = (int)value; return QS_OK; }
    if (!strcmp(key, "slab_pair")) { g_tune.slab_pair = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_fast_persist")) { g_tune.gemm_fast_persist = (int)value; return QS_OK; }
    if (!strcmp(key, "sandwich_tail")) { g_tune.sandwich_tail = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_fast_shape")) { g_tune.gemm_fast_shape = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_pick")) { g_tune.gemm_pick = (int)value; return QS_OK; }
    if (!strcmp(key, "small4")) { g_tune.small4 = (int)value; return QS_OK; }
    if (!strcmp(key, "quad4s")) { g_tune.quad4s = (int)value; return QS_OK; }
    if (!strcmp(key, "pair4c_stream")) { g_tune.pair4c_stream = (int)value; return QS_OK; }
    if (!strcmp(key, "pair4c")) { g_tune.pair4c = (int)value; return QS_OK; }
    if (!strcmp(key, "sandwich")) { g_tune.sandwich = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_strip")) { g_tune.gemm_strip = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_strip_w")) { g_tune.gemm_strip_w = (int)value; return QS_OK; }
    if (!strcmp(key, "gemm_fast_unaligned")) { g_tune.gemm_fast_unaligned = (int)value; return QS_OK; }
    if (!strcmp(key, "comm_drop_wait")) { g_tune.comm_drop_wait = (int)value; return QS_OK; }
    if (!strcmp(key, "sandwich_mode")) { g_tune.sandwich_mode = (int)value; return QS_OK; }
    if (!strcmp(key, "sandwich_t2")) { g_tune.sandwich_t2 = (int)value; return QS_OK; }
    if (!strcmp(key, "sandwich_v2")) { g_tune.sandwich_v2 = (int)value; return QS_OK; }
    return QS_ERR_BAD_EXTENT;
}

int qs_matmul(int dtype, const void* A, const void* B, void* out, int64_t m, int64_t n, int64_t k,
              int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t stride_a,
              int64_t stride_b, int64_t stride_c, int accumulate, void* stream) {
    dispatch_reset();
    return matmul_checked(dtype, A, B, out, m, n, k, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, accumulate,
                          (hipStream_t)stream);
}

}  // extern "C"

namespace qs {
// qs_matmul without the reset of the dispatch record (entry points that issue several products: qs_comm.hip)
int matmul_checked(int dtype, const void* A, const void* B, void* out, int64_t m, int64_t n, int64_t k, int64_t lda,
                   int64_t ldb, int64_t ldc, int64_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                   int accumulate, hipStream_t stream) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!A || !B || !out) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(A, es) || !aligned(B, es) || !aligned(out, es)) return QS_ERR_MISALIGNED;
    if (stride_a < 0 || stride_b < 0 || stride_c < 0) return QS_ERR_BAD_EXTENT;
    return gemm(dtype, A, B, out, m, n, k, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, stream, accumulate);
}
}  // namespace qs

extern "C" {

int64_t qs_transform_two_body_workspace(int dtype, int64_t L, int64_t M) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M)) return QS_ERR_BAD_EXTENT;
    const int64_t wa = (L * L * L * M > L * M * M * M) ? L * L * L * M : L * M * M * M;
    const int64_t wb = (M < L) ? L * L * M * M : 0;   // otherwise T2 lives in `out`
    return (even_up(L * M) + wa + wb) * (int64_t)elem_size(dtype);
}

}  // extern "C"

namespace qs {
// qs_transform_two_body / qs_transform_two_body_mixed: `in_dtype` is the type of u, `dtype` that of C, Ct and out

// where the streamed fp64 kernel measured faster than the other small-basis paths (profiles/r03_quad4s.txt)
// (9 ... 16 orbitals: 1.06-1.14x over qs_small4.hip; 17 ... 32: 1.12-1.21x over it / the 16-wide kernels, l = 20 10.2 -> 9.1 us, 32 22.4 -> 19.8; from 33
// the hand-scheduled qs_sandwich4*.hip stay ahead, 0.72-0.86x)
// 65 ... 95 orbitals (two workgroups per item quad): 1.05-1.42x over the tiled kernels, every size (l = 66 444 -> 331 us, 78 740 -> 585,
// 91 1406 -> 1113); 96 itself level (0.98x): the tiled kernels keep it.
static bool quad4s_wins(int64_t L, int64_t M) {
    if (L >= 9 && M >= 9 && L <= 32 && M <= 32) return true;
    // (round 4: a basis of exactly 80 = 5 x 16 orbitals is the strip kernels' -- 597 against 678 us, found by tools/dispatch_guard.py)
    if (L == 80 && M == 80) return false;
    return L >= 65 && M >= 65 && L <= 96 && M <= 96 && !(L == 96 && M == 96);
}

static int transform_two_body_impl(int in_dtype, int dtype, const void* u, const void* C, const void* Ct, void* out,
                                   void* work, int64_t work_bytes, int64_t L, int64_t M, void* stream) {
    dispatch_reset();
    if (!dtype_ok(dtype) || !dtype_ok(in_dtype) || (in_dtype == QS_C128 && dtype == QS_F64)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M)) return QS_ERR_BAD_EXTENT;
    if (!u || !C || !Ct || !out || !work) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(u, elem_size(in_dtype)) || !aligned(C, es) || !aligned(Ct, es) || !aligned(out, es) ||
        !aligned(work, 16))
        return QS_ERR_MISALIGNED;
    if (out == u || out == work) return QS_ERR_ALIAS;
    if (work_bytes < qs_transform_two_body_workspace(dtype, L, M)) return QS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;

    void* CT = work;
    void* WA = at(work, even_up(L * M), es);
    const int64_t wa = (L * L * L * M > L * M * M * M) ? L * L * L * M : L * M * M * M;
    void* WB = (M < L) ? at(WA, wa, es) : out;

    // up to 32 orbitals, both dtypes: two launches of the LDS-staged kernel (what is left below ~33 orbitals is launches, not
    // work: the 16-wide kernels need three to five); T2 (L, L, M, M) in WA
    // (same-box sweep with the launches of a transform captured in one graph, profiles/r03_small4.txt: 1.8-2.9x up to 15
    // orbitals for both dtypes, 1.1-1.4x for complex128 up to 24.  Since the streamed kernels exist (below: the loads of the
    // next items under the products of these) it is the automatic choice only up to 8 orbitals (fp64) / 4 (complex128) -- what the kernel waits for there is the one round trip of its loads and the
    // drain of its stores, with one workgroup per CU and nothing to overlap them with.  g_tune.small4 == 2: wherever it exists)
    const int64_t n4s = cdiv(L, 4);
    if (in_dtype == dtype && g_tune.small4 && L <= 32 && M <= 32 && n4s == cdiv(M, 4) &&
        (g_tune.small4 == 2 || n4s <= (dtype == QS_C128 ? 1 : 2))) {
        const int64_t MM = M * M;
        int rc1 = small4_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, MM, M, 1, 0, s);
        if (rc1 == QS_OK)
            rc1 = small4_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1, M * MM, MM, 1, s);
        if (rc1 != 1) return rc1;
    }

    // fp64, 5 ... 96 orbitals: the two passes on the streamed kernel (qs_quad4s.h) where it measured faster than what
    // follows (profiles/r03_quad4s.txt).  g_tune.quad4s == 2: wherever it exists.
    if (in_dtype == dtype && dtype == QS_F64 && g_tune.quad4s && L >= 5 && M >= 5 && L <= 96 && M <= 96 && n4s == cdiv(M, 4) &&
        (g_tune.quad4s == 2 || (quad4s_wins(L, M) && g_tune.sandwich < 4))) {       // (sandwich >= 4: tuning runs of those kernels)
        const int64_t MM = M * M;
        int rc1 = quad4s_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, MM, M, 1, s);
        if (rc1 == QS_OK)
            rc1 = quad4s_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1, M * MM, MM, s);
        if (rc1 != 1) return rc1;
    }

    // complex128 up to 56 orbitals: the same two passes with two items per matrix instruction (qs_pair4c.hip).  Automatic
    // where it measured faster than qs_small4.hip / the four 16-wide passes (same-box sweeps, profiles/r03_pair4c.txt): from 5
    // orbitals, all of them in the STREAMED form (section 5: item pairs through a ring of row quads, fetched ahead): 1.07-1.56x
    // (l = 20 17.2 -> 12.0 us, l = 55 409 -> 351 us = 45.9 TFLOP/s); 57 ... 64 orbitals exist but spill and lose.
    // g_tune.pair4c == 2: wherever it exists.
    if (in_dtype == dtype && dtype == QS_C128 && g_tune.pair4c && L <= 64 && M <= 64 && n4s == cdiv(M, 4) &&
        (g_tune.pair4c == 2 || (n4s >= 2 && n4s <= 14))) {
        const int64_t MM = M * M;
        int rc1 = pair4c_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, MM, M, 1, 0, s);
        if (rc1 == QS_OK)
            rc1 = pair4c_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1, M * MM, MM, 1, s);
        if (rc1 != 1) return rc1;
    }

    // small bases: two passes over the tensor instead of four -- (d, c) per slab u[a, b], then (b, a) per
    // column (r, s): out[:, :, rs] = Ct . T2[:, :, rs] . Ct^T (the same k-ordered sums, element for element)
    // a REAL tensor against complex coefficients (qs_transform_two_body_mixed), 5 ... 56 orbitals: the streamed pair kernel with
    // real items for the first pass (its first product is one MFMA per fragment instead of two), the complex one for the second
    // (tools/mixed_small.py: the tiled route took 22.7 us at l = 20 where a complex tensor takes 13.3).
    if (in_dtype == QS_F64 && dtype == QS_C128 && g_tune.pair4c && L >= 5 && M >= 5 && L <= 56 && M <= 56 &&
        n4s == cdiv(M, 4)) {
        const int64_t MM = M * M;
        int rc1 = pair4m_try(u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, MM, M, 1, s);
        if (rc1 == QS_OK)
            rc1 = pair4c_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1, M * MM, MM, 1, s);
        if (rc1 != 1) return rc1;
    }

    if (in_dtype == dtype && (g_tune.sandwich == 1 || g_tune.sandwich == 3 || g_tune.sandwich == 4 || g_tune.sandwich == 6)) {
        // T2 (L, L, M, M) goes to WA; the eligibility of the second pass is known before the first runs
        const int64_t MM = M * M;
        const int64_t n4 = cdiv(L, 4);
        // the policy (where two fused passes measured faster) ...
        const bool wanted = dtype == QS_F64 && L <= 64 && M <= 64 && n4 == cdiv(M, 4) && MM >= 1024 &&
                            (4 * n4) * L * MM * 2 * 8 < (int64_t(1) << 31) &&
                            (g_tune.sandwich >= 4 ? n4 >= 6 : n4 >= 9) &&
                            // (15: both passes on slabs through the balanced kernel's instantiation for 16, or not at all)
                            (n4 != 15 || (g_tune.sandwich_v2 != 0 && g_tune.sandwich_t2 != 0 && g_tune.sandwich != 3 &&
                                          g_tune.sandwich != 6));
        // ... and the kernel's own eligibility rule, asked of the kernel (dry run of the very calls made below), so the
        // two can never disagree after the first pass has run: T2 transposed only if both passes take that layout
        const bool second_nat = wanted && sandwich4_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1,
                                                        M * MM, MM, s, 1) == QS_OK;
        const bool second_tr = wanted && g_tune.sandwich != 3 && g_tune.sandwich != 6 &&
                               sandwich4_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, 1, M * L * L, L * L, s, 1) == QS_OK &&
                               sandwich4_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, L * L, L, 1, 1, M * MM, MM, s, 1) == QS_OK;
        const bool second_ok = second_nat || second_tr;
        if (second_ok) {
            void* T1s = (M < L) ? at(WA, wa, es) : out;     // scratch of the unfused fall-back of the first pass
            int rc2 = 1;
            // T2 transposed, (r, s, a, b): the second pass then fetches slabs like the first (64-byte runs, the four
            // waves of a workgroup on the same lines) instead of columns (32-byte runs), and the first one stores the
            // way the second does.  Same-box sweep (profiles/r02_small_basis_sweep.txt): 4-10 % faster for
            // ceil(l/4) in {10, 13, 14, 16}, faster than the second pass alone for 11, slower for 9 and l = 48 (45 ... 47:
            // faster since round 3, gpurun_out r03y tail_ab*); 15 has
            // slab passes only (the balanced kernel's instantiation for 16).
            const bool want_t2 = second_tr && (!second_nat || (g_tune.sandwich_t2 >= 0 ? g_tune.sandwich_t2 != 0
                                                                                         : (n4 == 10 || n4 == 11 || n4 >= 13 ||
                                                                                            (n4 == 12 && L % 4 != 0) ||
                                                                                            (n4 == 9 && L == 36 && M == 36 && g_tune.sandwich_tail))));
            bool t2_transposed = false;
            if (g_tune.sandwich != 3 && g_tune.sandwich != 6) {
                if (want_t2)
                    rc2 = sandwich4_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, 1, M * L * L, L * L, s);
                else if (second_nat)
                    rc2 = sandwich4_try(dtype, u, WA, C, M, 1, C, 1, M, L * L, L, M, L * L, L, 1, MM, M, 1, s);
                t2_transposed = want_t2 && rc2 != 1;
            }
            if (rc2 == 1 && second_nat) {
                // first pass on the 16-wide kernels (T1 in the spare buffer, T2 into WA, natural layout)
                rc2 = slab_pair_try(dtype, u, C, WA, L * L, L, M, s);
                if (rc2 == 1) {
                    rc2 = transpose_small(dtype, C, CT, L, M, s);
                    if (rc2) return rc2;
                    rc2 = gemm(dtype, u, C, T1s, L * L * L, M, L, L, M, M, 1, 0, 0, 0, s);
                    if (rc2) return rc2;
                    rc2 = gemm(dtype, CT, T1s, WA, M, M, L, L, M, M, L * L, 0, L * M, MM, s);
                }
            }
            if (rc2) return rc2;
            if (t2_transposed)
                rc2 = sandwich4_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, L * L, L, 1, 1, M * MM, MM, s);
            else
                rc2 = sandwich4_try(dtype, WA, out, Ct, 1, L, Ct, L, 1, MM, L, M, 1, L * MM, MM, 1, M * MM, MM, s);
            if (rc2 != 1) return rc2;
            // not reached: the dry runs above ARE the kernel's eligibility rule for these very calls
            snprintf(g_hip_err, sizeof(g_hip_err), "qs_transform_two_body: second fused pass refused after its dry run");
            return QS_ERR_HIP;
        }
    }

    int rc = contract_dcb(in_dtype, dtype, u, C, CT, Ct, /*T1*/ WA, /*T2*/ WB, /*T3*/ WA, L, L, M, s);
    if (rc) return rc;
    return gemm(dtype, Ct, WA, out, M, M * M * M, L, L, M * M * M, M * M * M, 1, 0, 0, 0, s);
}
}  // namespace qs

extern "C" {

int qs_transform_two_body(int dtype, const void* u, const void* C, const void* Ct, void* out,
                          void* work, int64_t work_bytes, int64_t L, int64_t M, void* stream) {
    return transform_two_body_impl(dtype, dtype, u, C, Ct, out, work, work_bytes, L, M, stream);
}

/* real fp64 u against complex128 coefficients -> complex128 result, without a complex copy of u: the d contraction
 * reads the real tensor (8 bytes per element) and runs as a real product against C seen as an (L, 2M) real matrix; the
 * other three contractions are complex.  Workspace: qs_transform_two_body_workspace(QS_C128, L, M). */
int qs_transform_two_body_mixed(const void* u_f64, const void* C, const void* Ct, void* out, void* work,
                                int64_t work_bytes, int64_t L, int64_t M, void* stream) {
    return transform_two_body_impl(QS_F64, QS_C128, u_f64, C, Ct, out, work, work_bytes, L, M, stream);
}

int64_t qs_transform_two_body_inplace_workspace(int dtype, int64_t L, int64_t M) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M) || M > L) return QS_ERR_BAD_EXTENT;
    return (even_up(L * M) + L * L * L * M) * (int64_t)elem_size(dtype);
}

int qs_transform_two_body_inplace(int dtype, void* u, const void* C, const void* Ct, void* work,
                                  int64_t work_bytes, int64_t L, int64_t M, void* stream) {
    dispatch_reset();
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M) || M > L) return QS_ERR_BAD_EXTENT;
    if (!u || !C || !Ct || !work) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(u, es) || !aligned(C, es) || !aligned(Ct, es) || !aligned(work, 16)) return QS_ERR_MISALIGNED;
    if (work == u) return QS_ERR_ALIAS;
    if (work_bytes < qs_transform_two_body_inplace_workspace(dtype, L, M)) return QS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    void* CT = work;
    void* B = at(work, even_up(L * M), es);
    // the four contractions ping-pong between the tensor's own storage (A) and ONE spare buffer (B): every
    // product reads one and writes the other, and each intermediate fits where it goes (M <= L):
    //   d: A (L^4) -> B (L^3 M)   c: B -> A (L^2 M^2)   b: A -> B (L M^3)   a: B -> A (M^4)
    int rc = transpose_small(dtype, C, CT, L, M, s);
    if (rc) return rc;
    rc = gemm(dtype, u, C, B, L * L * L, M, L, L, M, M, 1, 0, 0, 0, s);
    if (rc) return rc;
    rc = gemm(dtype, CT, B, u, M, M, L, L, M, M, L * L, 0, L * M, M * M, s);
    if (rc) return rc;
    rc = gemm(dtype, Ct, u, B, M, M * M, L, L, M * M, M * M, L, 0, L * M * M, M * M * M, s);
    if (rc) return rc;
    return gemm(dtype, Ct, B, u, M, M * M * M, L, L, M * M * M, M * M * M, 1, 0, 0, 0, s);
}

int64_t qs_transform_two_body_partial_workspace(int dtype, int64_t L, int64_t M, int64_t rows) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M) || rows <= 0 || rows > L) return QS_ERR_BAD_EXTENT;
    return (even_up(L * M) + rows * L * L * M + rows * L * M * M) * (int64_t)elem_size(dtype);
}

int qs_transform_two_body_partial(int dtype, const void* u_slab, const void* C, const void* Ct,
                                  void* v_slab, void* work, int64_t work_bytes, int64_t L,
                                  int64_t M, int64_t rows, void* stream) {
    dispatch_reset();
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (!extents_ok(L, M) || rows <= 0 || rows > L) return QS_ERR_BAD_EXTENT;
    if (!u_slab || !C || !Ct || !v_slab || !work) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(u_slab, es) || !aligned(C, es) || !aligned(Ct, es) || !aligned(v_slab, es) ||
        !aligned(work, 16))
        return QS_ERR_MISALIGNED;
    if (v_slab == u_slab || v_slab == work) return QS_ERR_ALIAS;
    if (work_bytes < qs_transform_two_body_partial_workspace(dtype, L, M, rows)) return QS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    void* CT = work;
    void* T1 = at(work, even_up(L * M), es);
    void* T2 = at(T1, rows * L * L * M, es);
    return contract_dcb(dtype, dtype, u_slab, C, CT, Ct, T1, T2, v_slab, rows, L, M, s);
}

int qs_transform_one_body(int dtype, const void* h, const void* C, const void* Ct, void* out,
                          void* work, int64_t work_bytes, int64_t nmat, int64_t L, int64_t M,
                          void* stream) {
    dispatch_reset();
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (L <= 0 || M <= 0 || nmat <= 0 || L > 65536 || M > 65536) return QS_ERR_BAD_EXTENT;
    if (!h || !C || !Ct || !out || !work) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(h, es) || !aligned(C, es) || !aligned(Ct, es) || !aligned(out, es) || !aligned(work, 16))
        return QS_ERR_MISALIGNED;
    if (out == h || out == work) return QS_ERR_ALIAS;
    if (work_bytes < nmat * L * M * (int64_t)es) return QS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // T[(i,a), q] = sum_b h[i][a, b] C[b, q]
    int rc = gemm(dtype, h, C, work, nmat * L, M, L, L, M, M, 1, 0, 0, 0, s);
    if (rc) return rc;
    // out[i][p, q] = sum_a Ct[p, a] T[i][a, q]
    return gemm(dtype, Ct, work, out, M, M, L, L, M, M, nmat, 0, L * M, M * M, s);
}

int qs_antisymmetrize(int dtype, const void* u, void* out, int64_t npq, int64_t l, void* stream) {
    dispatch_reset();
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (npq <= 0 || l <= 0 || l > 65536) return QS_ERR_BAD_EXTENT;
    if (!u || !out) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    if (!aligned(u, es) || !aligned(out, es)) return QS_ERR_MISALIGNED;
    return antisymmetrize(dtype, u, out, npq, l, (hipStream_t)stream);
}

int qs_spin_expand_two_body(int in_dtype, int out_dtype, const void* u, void* out, int64_t l,
                            int64_t p_lo, int64_t p_hi, int antisymmetrize_flag, void* stream) {
    dispatch_reset();
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return QS_ERR_BAD_DTYPE;
    if (in_dtype == QS_C128 && out_dtype == QS_F64) return QS_ERR_BAD_DTYPE;
    if (l <= 0 || l > 32768 || p_lo < 0 || p_hi > l || p_lo >= p_hi) return QS_ERR_BAD_EXTENT;
    if (!u || !out) return QS_ERR_NULL_POINTER;
    if (!aligned(u, elem_size(in_dtype)) || !aligned(out, elem_size(out_dtype))) return QS_ERR_MISALIGNED;
    if (out == u) return QS_ERR_ALIAS;
    return spin_expand(in_dtype, out_dtype, u, out, l, l, p_lo, p_hi, antisymmetrize_flag ? 1 : 0,
                       (hipStream_t)stream);
}

int qs_spin_expand_two_body_block(int in_dtype, int out_dtype, const void* u, void* out, int64_t l,
                                  int64_t np, int64_t nq, int antisymmetrize_flag, void* stream) {
    dispatch_reset();
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return QS_ERR_BAD_DTYPE;
    if (in_dtype == QS_C128 && out_dtype == QS_F64) return QS_ERR_BAD_DTYPE;
    if (l <= 0 || l > 32768 || np <= 0 || np > l || nq <= 0 || nq > l) return QS_ERR_BAD_EXTENT;
    if (!u || !out) return QS_ERR_NULL_POINTER;
    if (!aligned(u, elem_size(in_dtype)) || !aligned(out, elem_size(out_dtype))) return QS_ERR_MISALIGNED;
    if (out == u) return QS_ERR_ALIAS;
    return spin_expand(in_dtype, out_dtype, u, out, l, nq, 0, np, antisymmetrize_flag ? 1 : 0,
                       (hipStream_t)stream);
}

int qs_add_spin_one_body(int in_dtype, int out_dtype, const void* h, void* out, int64_t nmat,
                         int64_t l, void* stream) {
    dispatch_reset();
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return QS_ERR_BAD_DTYPE;
    if (in_dtype == QS_C128 && out_dtype == QS_F64) return QS_ERR_BAD_DTYPE;
    if (l <= 0 || nmat <= 0 || l > (1 << 20)) return QS_ERR_BAD_EXTENT;
    if (!h || !out) return QS_ERR_NULL_POINTER;
    if (!aligned(h, elem_size(in_dtype)) || !aligned(out, elem_size(out_dtype))) return QS_ERR_MISALIGNED;
    if (out == h) return QS_ERR_ALIAS;
    return kron_eye2(in_dtype, out_dtype, h, out, nmat, l, (hipStream_t)stream);
}

int qs_spin_squared_two_body(const void* S, void* out, int64_t n, int64_t p_lo, int64_t p_hi,
                             int antisymmetrize_flag, void* stream) {
    dispatch_reset();
    if (n <= 0 || n > 65536 || p_lo < 0 || p_hi > n || p_lo >= p_hi) return QS_ERR_BAD_EXTENT;
    if (!S || !out) return QS_ERR_NULL_POINTER;
    if (!aligned(S, 16) || !aligned(out, 16)) return QS_ERR_MISALIGNED;
    return spin2_two_body(S, out, n, p_lo, p_hi, antisymmetrize_flag ? 1 : 0, (hipStream_t)stream);
}

}  // extern "C"
