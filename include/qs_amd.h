/*
 * qs_amd.h -- C ABI of the MI355X (gfx950) integral basis-transformation path.
 *
 * The reference (HyQD/quantum-systems v0.2.6) is pure Python and has no FFI:
 * its seam is the injected array module (`np=` / `change_module`,
 * quantum_systems/basis_set.py:32-38, :268-296).  The entry points below are
 * what a binding for that seam calls in place of the NumPy calls on the hot
 * path; each one names the reference call it replaces.  `INTEGRATION.md` shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor
 *     storage), 8-byte aligned (16 for complex), row-major, contiguous;
 *   - complex128 is interleaved (re, im) doubles, as NumPy / torch store it;
 *   - conjugation is resolved by the caller (`Ct` is passed explicitly);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it
 *     and the call returns without synchronising;
 *   - the library allocates nothing: outputs and workspace are the caller's;
 *   - no process-global mutable state: caches are keyed by device ordinal, the
 *     error / dispatch / tuning records are thread-local; calls are re-entrant
 *     for distinct streams and devices (the current device must be the one
 *     that owns the pointers and the stream);
 *   - return value: QS_OK (0) or a negative QS_ERR_* code, never throws.
 *
 * dtype codes: QS_F64 = real fp64, QS_C128 = complex128.
 */
#ifndef QS_AMD_H
#define QS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_ABI_VERSION 4

enum {
    QS_OK = 0,
    QS_ERR_BAD_EXTENT = -1,   /* non-positive or overflowing dimension      */
    QS_ERR_NULL_POINTER = -2, /* required pointer is NULL                    */
    QS_ERR_MISALIGNED = -3,   /* pointer not aligned to its element size     */
    QS_ERR_WORKSPACE = -4,    /* workspace smaller than qs_*_workspace says  */
    QS_ERR_HIP = -5,          /* a HIP runtime call or kernel launch failed  */
    QS_ERR_BAD_DTYPE = -6,    /* dtype code not QS_F64 / QS_C128             */
    QS_ERR_ALIAS = -7,        /* output aliases an input where not allowed   */
    QS_ERR_COMM = -8          /* RCCL call failed / librccl not loadable     */
};

enum { QS_F64 = 0, QS_C128 = 1 };

/* ABI version of the loaded library (QS_ABI_VERSION it was built with). */
int qs_abi_version(void);

/* Text for a QS_ERR_* code (static storage). */
const char* qs_error_string(int code);

/* Last HIP error string recorded by this thread's most recent QS_ERR_HIP. */
const char* qs_last_hip_error(void);

/*
 * Row-major (batched) matrix product  out[b] = A[b] . B[b]
 *   A[b] : (m, k), leading dimension lda, batch stride stride_a (0 = shared)
 *   B[b] : (k, n), leading dimension ldb, batch stride stride_b (0 = shared)
 *   out[b]: (m, n), leading dimension ldc, batch stride stride_c
 * Strides and leading dimensions are in ELEMENTS of the dtype.
 * accumulate != 0 computes out[b] += A[b] . B[b] (used to close a contraction
 * whose summed index arrives in several slabs, see qs_transform_two_body_partial).
 * Replaces np.dot / np.tensordot over one index:
 *   transform_spf / transform_bra_spf   basis_set.py:321-327
 *   transform_one_body_elements         basis_set.py:329-334
 * and is the building block of qs_transform_two_body.
 */
int qs_matmul(int dtype, const void* A, const void* B, void* out,
              int64_t m, int64_t n, int64_t k,
              int64_t lda, int64_t ldb, int64_t ldc,
              int64_t batch, int64_t stride_a, int64_t stride_b,
              int64_t stride_c, int accumulate, void* stream);

/*
 * Bytes of workspace qs_transform_two_body needs for u:(L,L,L,L) -> (M,M,M,M).
 * Returns a negative QS_ERR_* on bad extents.
 */
int64_t qs_transform_two_body_workspace(int dtype, int64_t L, int64_t M);

/*
 * Four-index transform
 *   out[p,q,r,s] = sum_abcd Ct[p,a] Ct[q,b] u[a,b,c,d] C[c,r] C[d,s]
 * evaluated as the reference does, one index at a time in the order d, c, b, a.
 *   u   : (L,L,L,L)   C : (L,M)   Ct : (M,L)   out : (M,M,M,M)
 * `u` is not modified; `out` must not alias `u` or the workspace.
 * Replaces BasisSet.transform_two_body_elements, basis_set.py:336-350.
 */
int qs_transform_two_body(int dtype, const void* u, const void* C,
                          const void* Ct, void* out, void* work,
                          int64_t work_bytes, int64_t L, int64_t M,
                          void* stream);

/*
 * The same transform for a REAL fp64 tensor against complex128 coefficients
 * (NumPy's promotion at basis_set.py:341-342; the per-step call of a
 * time-dependent solver on a real quantum-dot `u`, system.py:222-225):
 *   u_f64 (L,L,L,L) fp64;  C (L,M), Ct (M,L), out (M,M,M,M) complex128.
 * No complex copy of `u` is made: the d contraction reads the real tensor
 * (8 bytes per element) and runs on the real matrix instruction against C seen
 * as an (L, 2M) real matrix -- half the MFMA work of the promoted product --
 * and the c, b, a contractions are complex.  Workspace:
 * qs_transform_two_body_workspace(QS_C128, L, M).
 */
int qs_transform_two_body_mixed(const void* u_f64, const void* C, const void* Ct,
                                void* out, void* work, int64_t work_bytes,
                                int64_t L, int64_t M, void* stream);

/*
 * The same transform IN PLACE, for a caller that drops the old tensor anyway
 * (BasisSet.change_basis rebinds self.u, basis_set.py:374-377): `u` (L,L,L,L) is
 * overwritten and the result (M,M,M,M), M <= L, is left at the START of its
 * storage.  Peak memory is the tensor plus ONE L^3 M spare buffer instead of
 * tensor + workspace + result: the four contractions ping-pong between the two.
 * Same arithmetic, same order; l = 256 fp64 needs 69 GB instead of 103 GB, and
 * the largest fp64 basis one MI355X holds grows from 320 to ~360 orbitals.
 */
int64_t qs_transform_two_body_inplace_workspace(int dtype, int64_t L, int64_t M);
int qs_transform_two_body_inplace(int dtype, void* u, const void* C,
                                  const void* Ct, void* work, int64_t work_bytes,
                                  int64_t L, int64_t M, void* stream);

/*
 * Same transform restricted to rows [a_lo, a_hi) of the leading index of `u`
 * for the contractions over d, c, b only:
 *   v[a,q,r,s] = sum_bcd Ct[q,b] u[a,b,c,d] C[c,r] C[d,s],  a in [a_lo,a_hi)
 *   u_slab : (a_hi-a_lo, L, L, L)     v_slab : (a_hi-a_lo, M, M, M)
 * The slab-local half of the sharded transform; the contraction
 * over `a` is a plain qs_matmul on the exchanged slabs.
 */
int64_t qs_transform_two_body_partial_workspace(int dtype, int64_t L, int64_t M,
                                                int64_t rows);
int qs_transform_two_body_partial(int dtype, const void* u_slab, const void* C,
                                  const void* Ct, void* v_slab, void* work,
                                  int64_t work_bytes, int64_t L, int64_t M,
                                  int64_t rows, void* stream);

/*
 * One-body transform of a stack of matrices:  out[i] = Ct . (h[i] . C)
 *   h : (nmat, L, L)   out : (nmat, M, M)   work : nmat*L*M elements
 * Replaces BasisSet.transform_one_body_elements, basis_set.py:329-334, as
 * applied to h, s, position[i], momentum[i] (:358-406).
 */
int qs_transform_one_body(int dtype, const void* h, const void* C,
                          const void* Ct, void* out, void* work,
                          int64_t work_bytes, int64_t nmat, int64_t L,
                          int64_t M, void* stream);

/*
 * Anti-symmetrisation  out[p,q,r,s] = u[p,q,r,s] - u[p,q,s,r]
 *   u, out : (npq, l, l) with npq = number of leading (p,q) pairs handled
 *            (l*l for a full tensor, fewer for a p-slab).
 * `out` may equal `u` (in place).  Exact (one subtraction per element).
 * Replaces BasisSet.anti_symmetrize_u, basis_set.py:776-778.
 */
int qs_antisymmetrize(int dtype, const void* u, void* out, int64_t npq,
                      int64_t l, void* stream);

/*
 * Spin doubling of the two-body tensor, optionally fused with the
 * anti-symmetrisation and the cast to complex128, for spatial rows
 * p in [p_lo, p_hi):
 *   out[2p+s1, 2q+s2, 2r+s3, 2s+s4] =
 *        d(s1,s3) d(s2,s4) u[p,q,r,s]  - antisym * d(s1,s4) d(s2,s3) u[p,q,s,r]
 *   u   : (l,l,l,l) of in_dtype (full tensor, indexed by absolute p)
 *   out : (2*(p_hi-p_lo), 2l, 2l, 2l) of out_dtype (slab, first row = 2*p_lo)
 * in_dtype QS_F64 may be combined with out_dtype QS_C128 (imaginary part 0).
 * Replaces add_spin_two_body (basis_set.py:772-774) + anti_symmetrize_u
 * (:776-778) + cast_to_complex (:298-319) inside
 * change_to_general_orbital_basis (:530-636).
 */
int qs_spin_expand_two_body(int in_dtype, int out_dtype, const void* u,
                            void* out, int64_t l, int64_t p_lo, int64_t p_hi,
                            int antisymmetrize, void* stream);

/*
 * The same expansion for a BLOCK of the tensor, the form the sharded layouts
 * use (no rank holds the whole tensor):
 *   u   : (np, nq, l, l)  = u[p0:p0+np, q0:q0+nq, :, :], contiguous
 *   out : (2 np, 2 nq, 2l, 2l) = out[2 p0 : 2(p0+np), 2 q0 : 2(q0+nq), :, :]
 * np = rows of a leading-index slab with nq = l, or np = l with nq = the rows
 * of a second-index slab.  Slab-local: output element (2p+s1, 2q+s2, ., .)
 * needs input matrix (p, q) only.
 */
int qs_spin_expand_two_body_block(int in_dtype, int out_dtype, const void* u,
                                  void* out, int64_t l, int64_t np, int64_t nq,
                                  int antisymmetrize, void* stream);

/*
 * kron(h, I2) for a stack of matrices: out[i, 2p+s, 2q+t] = d(s,t) h[i,p,q]
 *   h : (nmat, l, l) in_dtype     out : (nmat, 2l, 2l) out_dtype
 * Replaces BasisSet.add_spin_one_body, basis_set.py:768-770.
 */
int qs_add_spin_one_body(int in_dtype, int out_dtype, const void* h, void* out,
                         int64_t nmat, int64_t l, void* stream);

/*
 * Two-body part of S^2:
 *   out[p,q,r,s] = sum_i S_i[p,r] S_i[q,s]  - antisym * S_i[p,s] S_i[q,r]
 *   S : (3, n, n) complex128 (spin_x, spin_y, spin_z), out : rows
 *   p in [p_lo, p_hi) of the (n,n,n,n) complex128 tensor.
 * Replaces the einsum("pr,qs->pqrs") accumulation of
 * setup_spin_squared_operator, basis_set.py:745-747 (+ :525-526).
 */
int qs_spin_squared_two_body(const void* S, void* out, int64_t n, int64_t p_lo,
                             int64_t p_hi, int antisymmetrize, void* stream);

/*
 * Coulomb matrix elements of the two-dimensional harmonic oscillator (quantum
 * dot) in the Fock-Darwin basis, omega = 1:
 *   out[p - p_lo, q, r, s] = <pq|u|rs>,  p in [p_lo, p_hi),  out : fp64
 * orbitals ordered by shell as two_dim_helper.py:111-166 (index p <-> (n, m)).
 * Replaces _get_coulomb_elements, quantum_dots/two_dim/two_dim_helper.py:250-268
 * and coulomb_ho, quantum_dots/two_dim/coulomb_elements.py:6-92 (the input
 * generator of TwoDimensionalHarmonicOscillator, two_dim_ho.py:84-95).
 */
int qs_tdho_coulomb_elements(void* out, int64_t l, int64_t p_lo, int64_t p_hi,
                             void* stream);

/*
 * The same elements for an explicit orbital table: nm_table is a device array
 * of 2*l int32, n of every orbital followed by m of every orbital; max_shell =
 * max(2 n + |m| + 1) over the table (the caller knows it; the log-factorial
 * tables cover max_shell <= 30).  Replaces get_coulomb_elements_B,
 * two_dim_helper.py:284-301 (orbitals ordered by their energy in a magnetic
 * field, TwoDimHarmonicOscB, two_dim_ho.py:213-276).
 */
int qs_tdho_coulomb_elements_nm(void* out, const void* nm_table, int64_t l,
                                int64_t max_shell, int64_t p_lo, int64_t p_hi,
                                void* stream);

/*
 * ---- several GPUs of one node: one process per GPU, RCCL over xGMI ------------
 * (the reference itself knows one device only.)
 *
 * qs_comm_unique_id: 128 bytes that identify a communicator; ONE rank calls it,
 *   the host distributes the bytes to the other ranks by its own means (MPI
 *   broadcast, a file, a socket) -- exactly ncclGetUniqueId's contract.
 * qs_comm_init: collective over the `world` ranks; binds the communicator to
 *   the calling thread's current device and creates the stream the exchange
 *   runs on.  The handle is the only persistent object the library owns.
 * qs_comm_destroy: frees it.  qs_last_comm_error: text of the calling thread's
 *   most recent QS_ERR_COMM.
 * RCCL is loaded at run time (librccl.so.1; the copy already in the process
 * when there is one): single-GPU users have no link-time dependency on it.
 * (Development / test hook: the environment variable QS_AMD_RCCL_LIB names a
 * library to load instead -- the test suite's file-based stand-in, which lets
 * several ranks share one GPU.)
 */
#define QS_UNIQUE_ID_BYTES 128
int qs_comm_unique_id(void* id /* QS_UNIQUE_ID_BYTES */);
int qs_comm_init(void** comm, int rank, int world, const void* unique_id);
int qs_comm_destroy(void* comm);
/* Tear down without waiting for outstanding operations (ncclCommAbort): the
 * only thing left to do after a sharded call returned an error in the middle
 * of its exchange (the handle then refuses further work: peers may be blocked
 * in a group this rank never completed) or when a peer died. */
int qs_comm_abort(void* comm);
int qs_comm_rank(void* comm);
int qs_comm_world(void* comm);
const char* qs_last_comm_error(void);
/* Per-handle options (every rank of the communicator must choose the same):
 *   "rows_coalesce" = 1: qs_transform_two_body_sharded_rows exchanges ONE
 *     message per peer and step -- the peer's block of the send buffer as it
 *     is; the received block goes through a staging area at the end of the
 *     workspace and is put in place by one strided copy on the communicator's
 *     stream -- instead of one message per peer and result row that lands in
 *     place (0, the default).  Same results bit for bit; trades
 *     jl (world - 1) chunk_rows M^2 elements of workspace and one extra pass
 *     over the received rows for (world - 1) instead of jl (world - 1)
 *     messages per step and direction.
 * Unknown key: QS_ERR_BAD_EXTENT. */
int qs_comm_set_option(void* comm, const char* key, int64_t value);

/*
 * STATUS of the sharded entry points below: EXPERIMENTAL.  On REAL RCCL they
 * have run with ONE rank only (the development boxes hold one GPU).  Their
 * multi-rank branches are executed by tests/test_gpu_mock_rccl_ranks.py: 2-5
 * rank processes on one GPU, every ncclSend / ncclRecv the library posts
 * carried by a file-based stand-in for librccl (tests/cabi/mock_rccl.cpp: same
 * pairing and size rules, no asynchrony), results bit-identical to the
 * single-GPU transform, and by tests/test_gpu_async_transport.py: the ranks as
 * THREADS of one process over a stream-ordered, asynchronous stand-in
 * (tests/cabi/mock_rccl_async.cpp: ncclGroupEnd returns before anything has
 * moved, every transfer is a device copy behind events of both sides, with an
 * optional delay), which fails -- and is tested to fail -- when any one of
 * the stream waits between the caller's stream and the communicator's stream
 * is left out; in addition worlds of 1..8 ranks are covered by CPU
 * replays of the exchange plans (qs_sharded_exchange_plan,
 * qs_sharded_rows_exchange_plan) and by the same algorithms driven through
 * torch.distributed in the Python layer.  Unrun until an 8-GPU node: RCCL's own
 * transport and the overlap of the two streams.
 *
 * Four-index transform of a tensor sharded over the ranks of `comm`:
 *   u_bslab   : u[:, b_lo:b_hi, :, :]  (L, bl, L, L), this rank's share of the
 *               SECOND index; balanced split: the first L % world ranks hold
 *               L / world + 1 rows (same rule for the result)
 *   out_pslab : out[p_lo:p_hi]  (pc, M, M, M), this rank's share of the LEADING
 *               index of the result
 *   C (L, M), Ct (M, L) replicated on every rank.
 * d, c and a are contracted on the slab, ONE exchange re-shards
 * [p, b_loc] -> [p_loc, b] -- (world-1)/world^2 of the tensor leaves every rank,
 * as grouped ncclSend / ncclRecv pairs so that every peer's xGMI link carries
 * its share at once -- and the contraction over b closes on the received rows.
 * The exchange is issued in `nchunks` pieces (1..16; <= 0 selects 4) on the
 * communicator's stream and overlaps the products on `stream`; on return,
 * `stream` is ordered behind all of it.  Replaces transform_two_body_elements
 * (basis_set.py:336-350) for a tensor that one device does not hold;
 * collective: every rank of `comm` must call it with the same L, M, nchunks.
 */
int64_t qs_transform_two_body_sharded_workspace(int dtype, int64_t L, int64_t M,
                                                int world, int rank);
int qs_transform_two_body_sharded(void* comm, int dtype, const void* u_bslab,
                                  const void* C, const void* Ct, void* out_pslab,
                                  void* work, int64_t work_bytes, int64_t L,
                                  int64_t M, int nchunks, void* stream);

/*
 * The exchange plan of qs_transform_two_body_sharded for one rank, as numbers
 * (pure index arithmetic: no GPU, no RCCL).  Test hook: the CPU suite replays
 * the plans of all ranks of a world with NumPy and checks that every row ends
 * up where the closing product reads it.
 *   header  : {b_lo, bl, p_lo, pc, row_x, row_r, nchunks}
 *   ct_rows : M entries, the row of Ct multiplied in slot i of X
 *   chunks  : nchunks x {first slot, slots, first result row (relative), result rows}
 *   table   : one row {chunk, peer, kind, x_off, r_off, count, rows} per
 *             operation, kind 0 send / 1 receive / 2 own rows (X -> R)
 * Returns the number of operations (<= table_rows) or a negative QS_ERR_*.
 */
int qs_sharded_exchange_plan(int64_t L, int64_t M, int world, int rank,
                             int nchunks, int64_t* header, int64_t* ct_rows,
                             int64_t* chunks, int64_t* table, int64_t table_rows);

/*
 * The memory-lean sharded transform: rows of ONE leading index in, rows of the
 * OTHER one out, everything else O(chunk_rows * l^3).  Replaces
 * transform_two_body_elements (basis_set.py:336-350) for the per-step call on
 * a resident sharded `u` (system.py:222-225) and inside change_basis
 * (basis_set.py:374-382) when two slabs are all a GPU can hold (l = 512
 * complex128 on 8 GPUs: 128 GiB in + 128 GiB out per rank).
 *   rows       : (il, L, L, L) of `in_dtype` -- rows[i][j] = u[i_lo + i, j]
 *                for a leading-index sharding, u[j, i_lo + i] for a
 *                second-index sharding (the transform is symmetric under
 *                swapping its two leading index pairs, so one routine serves
 *                both); `in_starts` (world + 1 offsets, 0 ... L) gives every
 *                rank's first row, NULL = the balanced split.  in_dtype
 *                QS_F64 with dtype QS_C128 is the mixed product above.
 *   out_buffer : qs_transform_two_body_sharded_rows_out_bytes(); on return its
 *                first jl * M^3 elements are out[j'_loc][i'][r][s], this
 *                rank's rows j' (balanced split of M) of the other
 *                transformed leading index.  The rest of the buffer held the
 *                received rows: the closing contraction runs row by row
 *                inside it, no second slab.
 *   chunk_rows : input rows per exchange step (<= 0: qs_sharded_rows_default_chunk,
 *                at least four steps within a fixed scratch budget).  Per step
 *                d, c and the contraction over the whole leading index run on
 *                `stream`; the grouped ncclSend / ncclRecv of the step (one
 *                contiguous message per peer and result row, landing in
 *                place) run on the communicator's stream under the next
 *                step's products.
 * Collective: every rank passes the same L, M, chunk_rows, in_starts.
 */
int64_t qs_sharded_rows_default_chunk(int dtype, int64_t L, int64_t M, int world,
                                      const int64_t* in_starts);
int64_t qs_transform_two_body_sharded_rows_out_bytes(int dtype, int64_t L, int64_t M,
                                                     int world, int rank);
int64_t qs_transform_two_body_sharded_rows_workspace(int dtype, int64_t L, int64_t M,
                                                     int64_t chunk_rows);
/* The same for THIS handle: adds the staging area when the handle's option
 * "rows_coalesce" is set (this is the size the call checks work_bytes against). */
int64_t qs_comm_rows_workspace(void* comm, int dtype, int64_t L, int64_t M,
                               int64_t chunk_rows);
int qs_transform_two_body_sharded_rows(void* comm, int in_dtype, int dtype,
                                       const void* rows, const int64_t* in_starts,
                                       const void* C, const void* Ct,
                                       void* out_buffer, int64_t out_bytes,
                                       void* work, int64_t work_bytes, int64_t L,
                                       int64_t M, int64_t chunk_rows, void* stream);
/* Its exchange plan for one rank as numbers (test hook, no GPU, no RCCL):
 *   header : {i_start, il, jl, il_max, r0, out_elems, chunk_rows, nsteps}
 *   table  : {step, peer, kind, w_off, buf_off, count, rows} per operation,
 *            kind 0 send (offset into the step's send block W[j'][i][(r,s)]),
 *            1 receive (offset into out_buffer), 2 own rows (W -> out_buffer,
 *            `rows` pieces of `count` elements, pitches n*M*M and L*M*M).
 * Returns the number of operations (<= table_rows) or a negative QS_ERR_*. */
int qs_sharded_rows_exchange_plan(int64_t L, int64_t M, int world, int rank,
                                  const int64_t* in_starts, int64_t chunk_rows,
                                  int64_t* header, int64_t* table, int64_t table_rows);
/* ... of the coalesced exchange ("rows_coalesce"): kind 0 send (one per peer:
 * the peer's whole block of W), 3 receive into the staging area (buf_off =
 * offset into it), 4 staging -> out_buffer behind the group (w_off = offset
 * into the staging area, `rows` pieces of `count` elements, pitches count and
 * L*M*M), 2 own rows as above. */
int qs_sharded_rows_exchange_plan_coalesced(int64_t L, int64_t M, int world, int rank,
                                            const int64_t* in_starts, int64_t chunk_rows,
                                            int64_t* header, int64_t* table,
                                            int64_t table_rows);

/*
 * Which kernels the calling thread's most recent compute entry point launched,
 * as the names rocprofv3 prints for them, ';'-separated, repeated launches of
 * one instantiation folded to "name xN" (static thread-local storage; empty
 * before the first call).  bench.py puts this string into its `roofline.kernel`
 * field so that a bench line names the kernel that actually ran.
 */
const char* qs_last_dispatch(void);

/*
 * Auxiliary entry points (no reference counterpart).  NOT part of the product
 * path: tuning runs, tests and bench probes only.
 *   qs_tuning_set / qs_tuning_reset: override a kernel choice FOR THE CALLING
 *     THREAD (thread-local state: the library has no process-global mutable
 *     state; every thread starts from the automatic policy and
 *     qs_tuning_reset() returns the calling thread to it).  Keys
 *     "gemm_f64_cfg", "gemm_c128_cfg" (tile shape of the general
 *     kernel, 0 = automatic), "gemm_pipe" (1 = rotated K-loop schedule, 0 = plain),
 *     "gemm_fast" (0 = general kernel only, 1 = automatic, 2 = exact form of
 *     the VALU-free kernel only, 3 = its edge form wherever it is legal), "gemm_fast_shape" (edge-form tile
 *     shape 1..4, 0 = automatic), "gemm_fast_persist"
 *     (0 one workgroup per tile, 1 automatic, 2 always persistent, >= 3 tiles
 *     per workgroup),
 *     "gemm_skinny" (0 = never use the streaming short-and-wide kernel),
 *     "gemm_stream" (0 = never use the small-coefficient streaming kernel,
 *     2 = never split the rows of A over two waves), "slab_pair" (0 = never fuse
 *     the d and c contractions of a small-basis transform into one pass, 2 = one
 *     wave per slab always), "sandwich" (the two fused passes of a small-basis
 *     transform on the 4-wide fp64 matrix instruction: 0 = off, 1 = both,
 *     2 = the (d, c) pass only, 3 = the (b, a) pass only; 4 / 5 / 6 = both /
 *     (d, c) only / (b, a) only wherever the kernel is legal, not only where it
 *     measures faster), "sandwich_mode"
 *     (work split of those passes: -1 automatic, 0 one item quad per workgroup,
 *     1 four adjacent quads per workgroup, 3 the same with a barrier per step),
 *     "sandwich_t2" (the intermediate between those passes stored transposed,
 *     (r, s, a, b), so that the second pass fetches slabs too: -1 automatic,
 *     0 never, 1 always), "sandwich_v2" (the balanced form of those passes:
 *     -1 automatic, 0 never, 1 wherever it exists, 2 also odd quad counts on
 *     the next even instantiation), "sandwich_tail" (0 = never split the item
 *     quads of a partly filled last round over all workgroups), "small4"
 *     (the whole-quad kernel of up to 32 orbitals: 0 never, 1 automatic,
 *     2 wherever it exists), "quad4s" (the streamed fp64 kernel of 5-64
 *     orbitals: 0 never, 1 automatic, 2 wherever it exists), "pair4c" (the
 *     streamed complex128 / real-tensor-complex-coefficients kernel of 5-64
 *     orbitals: 0 never, 1 automatic, 2 wherever it exists), "gemm_fit"
 *     (fitted tile shapes of the general kernel: 0 never, 1 automatic, 2 always),
 *     "gemm_pick" (0 = tile shape by padded area only), "gemm_strip" (the strip
 *     kernels, which cover the small extent of a product with one tile to the
 *     next multiple of 16: 0 never, 1 by estimated time, 2 wherever they
 *     exist), "gemm_strip_w" (tuning runs: their relative rate in percent,
 *     0 = built-in weights), "gemm_fast_unaligned" (0 = 8-byte global items at
 *     odd strides as in rounds 1-3, 1 = 16-byte items at any 8-byte-aligned
 *     address), "comm_drop_wait" (TEST HOOK of the sharded entry points: a bit
 *     mask of stream waits between the caller's stream and the communicator's
 *     to leave out -- the negative control of the asynchronous stand-in
 *     transport, tests/test_gpu_async_transport.py; never set it elsewhere).
 *   qs_probe_mfma_f64: register-resident fp64 MFMA loop, `blocks` workgroups
 *     of 4 waves, each wave issuing iters*8 v_mfma_f64_16x16x4_f64
 *     (flops = blocks*4*iters*8*2048); `sink` is a device scratch of
 *     8 + 16*blocks bytes: after the dummy first word, per block the deltas of
 *     the shader clock and of the 100 MHz counter around the loop (their ratio
 *     x 100 MHz is the clock the chip holds under pure MFMA load).
 *   qs_probe_stream_copy: 16-byte-per-lane device copy (moves 2*bytes).
 * bench.py uses the probes to print measured ceilings beside datasheet ones.
 */
int qs_tuning_set(const char* key, int64_t value);
int qs_tuning_reset(void);
int qs_probe_mfma_f64(void* sink, int64_t blocks, int64_t iters, void* stream);
int qs_probe_stream_copy(const void* src, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QS_AMD_H */
