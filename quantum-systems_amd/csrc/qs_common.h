// Shared host-side helpers for the gfx950 basis-transformation library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

// Library state is keyed by DEVICE, never by process: one process may drive several GPUs
// (hipFuncSetAttribute, occupancy and the CU count are per-device facts).
constexpr int kMaxDevices = 64;

// Ordinal of the calling thread's current device (0 when the query fails).
int current_device();

// Compute units of the current device (cached per device: the property query costs far more
// than a kernel launch).
int device_cu_count();

// Largest dynamic-LDS size one kernel instantiation has been opted in to, per device (0 = never).
struct PerDeviceLds {
    std::atomic<uint32_t> bytes[kMaxDevices];
    PerDeviceLds() { for (auto& x : bytes) x.store(0); }
};

// Opt a kernel in to more than 64 KB of dynamic LDS on the current device.  The attribute is raised again whenever a
// launch asks for more than any earlier one did (kernels whose LDS size depends on run-time extents: spin2_tb,
// gemm_skinny), and never lowered.
int opt_in_dynamic_lds(const void* kern, size_t lds_bytes, PerDeviceLds& once, const char* what);

// Workgroups of `kern` (256 threads, no dynamic LDS) resident per CU on the current device,
// asked once per device and clamped to [1, 4]; `fallback` when the query fails.
struct PerDeviceInt {
    std::atomic<int> v[kMaxDevices];
    PerDeviceInt() { for (auto& x : v) x.store(0); }
};
int resident_workgroups(const void* kern, PerDeviceInt& cache, int fallback);

// Which kernels the calling thread's most recent entry point launched (qs_last_dispatch()):
// every launcher appends the name rocprofv3 shows for its instantiation.
void dispatch_reset();
void note_dispatch(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// ceil division for positive operands
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Internal launchers (defined in qs_gemm_f64.hip / qs_gemm_c128.hip).
int gemm_f64(const double* A, const double* B, double* C, int64_t m, int64_t n,
             int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch,
             int64_t sa, int64_t sb, int64_t sc, int accumulate, hipStream_t stream);
int gemm_c128(const double* A, const double* B, double* C, int64_t m, int64_t n,
              int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch,
              int64_t sa, int64_t sb, int64_t sc, int accumulate, hipStream_t stream);

// qs_matmul's body without the reset of the dispatch record (qs_api.hip)
int matmul_checked(int dtype, const void* A, const void* B, void* out, int64_t m, int64_t n, int64_t k, int64_t lda,
                   int64_t ldb, int64_t ldc, int64_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                   int accumulate, hipStream_t stream);

// A real (m x k, fp64) times B complex (k x n) -> out complex (m x n), row-major, leading dimensions in elements of each
// operand's own type.  Interleaved complex storage makes this EXACTLY the real product A . [B as k x 2n] -> [out as
// m x 2n]: the real kernels run it with 2 MFMAs per fragment pair and 8 bytes read per element of A -- no complex copy
// of A (the d contraction of a real u against complex coefficients, basis_set.py:341-342 with NumPy's promotion).
int matmul_real_by_complex(const void* A, const void* B, void* out, int64_t m, int64_t n, int64_t k, int64_t lda,
                           int64_t ldb, int64_t ldc, hipStream_t stream);

// VALU-free fast path, exact and edge forms (qs_gemm_fast.hip): QS_OK / error after launching, 1 = not eligible.
// general_cost: the general kernel's estimated time for this product (its best shape, in this kernel's units): the edge
// form runs when its own estimate is not worse.
int gemm_fast_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                  int64_t sb, int64_t sc, int accumulate, int group_along_m, double general_cost, hipStream_t stream);

// Small-coefficient streaming product, m, k <= 64 (qs_gemm_stream.hip): same return convention.
int gemm_stream_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n,
                    int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int64_t batch, int64_t sa,
                    int64_t sb, int64_t sc, int accumulate, hipStream_t stream);

// Fused pair of contractions on contiguous L x L slabs, L, M <= 64 (qs_slab_pair.hip): Z[s] = B^T.X[s].B.
int slab_pair_try(int dtype, const void* X, const void* B, void* Z, int64_t nslabs, int64_t L, int64_t M,
                  hipStream_t stream);

// Fused pair of contractions on the 4-wide matrix instruction (qs_sandwich4.hip): Out_t = Lm . In_t . R for a batch
// of L x L matrices with arbitrary element strides; L, M <= 64, ceil(L/4) == ceil(M/4).  dry_run: launch nothing,
// QS_OK = the call would launch.
int sandwich4_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
                  int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
                  int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, hipStream_t stream,
                  int dry_run = 0);

// Fused pair of contractions for SMALL bases, L, M <= 32, ceil(L/4) == ceil(M/4), fp64 and complex128 (qs_small4.hip): the
// same product on item quads staged in LDS.  tensor_is_b: complex only -- in the 16-wide kernels' call for the first
// product the tensor is the B operand (the b contraction), which fixes the order of the two imaginary-part products.
int small4_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, int tensor_is_b, hipStream_t stream);

// ... and for REAL items against complex R and Lm (the first pass of a real tensor against complex coefficients), streamed form only
int pair4m_try(const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm, int64_t l_sp, int64_t l_sa,
               int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row, int64_t in_col, int64_t out_item,
               int64_t out_row, int64_t out_col, hipStream_t stream);
// fp64, 17 ... 64 orbitals, streamed (qs_quad4s.hip): item quads through a ring of row quads, one wave per column group.
int quad4s_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, hipStream_t stream);
// The same for complex128, 5 ... 64 orbitals (qs_pair4s.h, qs_pair4c.hip): two items per matrix instruction, blocks = (item, re | im), streamed.
int pair4c_try(int dtype, const void* in, void* out, const void* R, int64_t r_sk, int64_t r_sj, const void* Lm,
               int64_t l_sp, int64_t l_sa, int64_t nitems, int64_t L, int64_t M, int64_t in_item, int64_t in_row,
               int64_t in_col, int64_t out_item, int64_t out_row, int64_t out_col, int tensor_is_b, hipStream_t stream);

// Strip kernels (qs_gemm_strip.hip): same return convention; general_cost as for gemm_fast_try.
int gemm_strip_try(int dtype, const double* A, const double* B, double* C, int64_t m, int64_t n, int64_t k, int64_t lda,
                   int64_t ldb, int64_t ldc, int64_t batch, int64_t sa, int64_t sb, int64_t sc, int accumulate,
                   double general_cost, hipStream_t stream);

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
int spin_expand(int in_dtype, int out_dtype, const void* u, void* out, int64_t l, int64_t nq, int64_t p_lo,
                int64_t p_hi, int as, hipStream_t stream);
int kron_eye2(int in_dtype, int out_dtype, const void* h, void* out, int64_t nmat, int64_t l,
              hipStream_t stream);
int spin2_two_body(const void* S, void* out, int64_t n, int64_t p_lo, int64_t p_hi, int as,
                   hipStream_t stream);

// Tuning knobs (qs_tuning_set / qs_tuning_reset): state of the CALLING THREAD only, so a tuning
// run or a test cannot change the dispatch of another thread's calls; every thread starts from the
// automatic policy.
struct Tuning {
    int gemm_f64_cfg = 0;        // tile shape of the general kernel, 0 = automatic
    int gemm_c128_cfg = 0;
    int gemm_pipe = 1;           // 1: rotated K-loop schedule, 0: plain schedule (A/B reference)
    int gemm_fast = 1;           // 0 general kernel only, 1 automatic, 2 exact form only, 3 edge form wherever legal
    int gemm_fast_persist = 1;   // 0 one workgroup per tile, 1 automatic, 2 always persistent, >= 3 tiles per workgroup
    int gemm_pick = 1;           // tile shape of the general kernel: 1 by rounds over the resident workgroups x tile work, 0 by padded area
    int sandwich_tail = 1;       // balanced small-basis kernel: the quads of a partly filled last round split over all workgroups
    int gemm_fast_shape = 0;     // forces edge-form shape 1..N (0 = by padded-work cost)
    int gemm_fit = 1;            // fitted tile shapes of the general kernel (the basis size covered by one tile, to the next multiple of
                                 // 16): 1 by estimated time, 2 wherever they exist, 0 off
    int gemm_skinny = 1;         // 0 disables the short-and-wide streaming product
    int gemm_stream = 1;         // 0 disables the small-coefficient streaming product, 2 = never split rows over two waves
    int slab_pair = 1;           // 0 disables the fused (d, c) pass, 2 = one wave per slab always
    int sandwich_t2 = -1;        // the intermediate of the two fused passes stored transposed, (r, s, a, b): -1 automatic, 0 never, 1 always
    int sandwich_v2 = -1;        // the balanced small-basis kernel with a cooperative fetch (qs_sandwich4b.hip): -1 automatic, 0 never, 1 wherever it exists,
                                 // 2 also every odd ceil(l/4) on the instantiation for the next even one
    int sandwich_mode = -1;      // work split of the fused passes: -1 automatic, 0 one item quad per workgroup, 1 four adjacent quads, 3 + step barrier
    int small4 = 1;              // both fused passes of a basis of <= 32 orbitals on the LDS-staged 4-wide kernel (qs_small4.hip), fp64 and
                                 // complex128: 1 automatic (fp64 up to 16, complex128 up to 24 orbitals), 2 wherever it exists (up to 32), 0 off
    int quad4s = 1;              // fp64 17 ... 64 orbitals on the streamed fused kernel (qs_quad4s.hip): 0 never, 1 where measured faster, 2 wherever it exists
    int pair4c_stream = 1;       // (kept for old tuning scripts: the whole-pair form it switched to is gone)
    int pair4c = 1;              // complex128 up to 56 orbitals: both fused passes on the two-items-per-instruction kernel (qs_pair4c.hip):
                                 // 1 automatic, 2 wherever it exists, 0 off
    int gemm_strip = 1;          // strip kernels (qs_gemm_strip.hip: the small extent of a product, <= 256, covered by ONE tile to the next
                                 // multiple of 16, eight waves): 0 never, 1 by estimated time, 2 wherever they exist
    int gemm_strip_w = 0;        // (tuning runs: relative rate of the strip kernels in percent, 0 = the built-in weights)
    int gemm_fast_unaligned = 1; // 16-byte items of the VALU-free kernel's edge form also at odd strides / extents (0: 8-byte items there)
    int comm_drop_wait = 0;      // TEST HOOK (qs_comm.hip): bit mask of stream waits of the sharded entry points to leave out -- the negative
                                 // test of the asynchronous stand-in transport; never set outside tests/
    int sandwich = 1;            // 4-wide fused passes of a small-basis transform: 0 off, 1 both (d, c) and (b, a), 2 (d, c) only, 3 (b, a) only;
                                 // tuning runs, wherever the kernel exists (not only where it measured faster): 4 both, 5 (d, c) only, 6 (b, a) only
};
extern thread_local Tuning g_tune;

}  // namespace qs
