// Multi-GPU entry points of the C ABI (include/qs_amd.h): an RCCL communicator behind an opaque handle and the
// four-index transform of a tensor sharded over the GPUs of one node, one process per GPU.
//
// The reference has no notion of a second device; SURVEY 8(b)/(e) asks for these entry points so that a
// host that is not Python can shard the path.  Layout (the one sharded.transform_two_body_sharded uses): `u` is
// sharded over its SECOND index (rank g holds u[:, b_lo:b_hi, :, :], balanced split), the result over its LEADING
// index (rank g gets out[p_lo:p_hi]).
//
//   local      d, c, then a (local in this layout):  X[p, b_loc, r, s] = Ct[p,a] u[a,b,c,d] C[c,r] C[d,s]
//   exchange   row p of X goes to the owner of p: (G-1)/G^2 of the tensor leaves every rank, one xGMI link per peer
//   close      out[p_loc][q, (r,s)] = Ct[q, b] R[p_loc][b, (r,s)]   with R[p_loc] = the rows received for p_loc
//
// RCCL is used directly: grouped ncclSend / ncclRecv pairs, one pair per row and peer, so that all seven links of a
// rank carry traffic at once (a ring all-to-all would be bound by one link).  The exchange is CHUNKED and runs on the
// communicator's own stream: the rows of Ct are taken in an order in which every chunk holds rows of EVERY peer;
// while chunk c travels, the contraction over a of chunk c + 1 and the closing contraction of chunk c - 1 run on the
// caller's stream.  No packing anywhere: a row of X is one contiguous message, and it lands at its final place
// R[p_loc][b_lo(sender) ...], from where the closing product reads it with K = L in one batched GEMM per chunk.
//
// RCCL is loaded at run time (dlopen of librccl.so.1: the copy PyTorch has already loaded when there is one), so the
// single-GPU library has no link-time dependency on it.

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <new>

#include "qs_common.h"

namespace qs {

namespace {

// ---- the few RCCL symbols used, resolved once per process (the functions are process-wide facts, not state)
typedef void* nccl_comm_t;
struct NcclUniqueId { char internal[QS_UNIQUE_ID_BYTES]; };
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, NcclUniqueId, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_comm_abort)(nccl_comm_t);
typedef int (*fn_group)(void);
typedef int (*fn_send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclFloat64 = 8;      // ncclFloat64 (rccl.h); complex128 travels as pairs of doubles

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_abort comm_abort = nullptr;      // optional
    fn_group group_start = nullptr, group_end = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_error_string error_string = nullptr;
    bool ok = false;
};

static thread_local char g_comm_err[256] = "";

const Rccl& rccl() {
    static const Rccl lib = [] {
        Rccl r;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        // development / test hook: QS_AMD_RCCL_LIB names the library to load instead (the file-based stand-in of
        // tests/cabi/mock_rccl.cpp, which lets several ranks share one GPU, also inside a process that has PyTorch's RCCL)
        if (const char* forced = getenv("QS_AMD_RCCL_LIB")) {
            if (forced[0]) r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        }
        for (const char* n : names) {      // a copy that is already in the process (PyTorch's) first
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) return r;
        r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
        r.comm_abort = (fn_comm_abort)dlsym(r.handle, "ncclCommAbort");
        r.group_start = (fn_group)dlsym(r.handle, "ncclGroupStart");
        r.group_end = (fn_group)dlsym(r.handle, "ncclGroupEnd");
        r.send = (fn_send)dlsym(r.handle, "ncclSend");
        r.recv = (fn_recv)dlsym(r.handle, "ncclRecv");
        r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.group_start && r.group_end && r.send &&
               r.recv && r.error_string;
        return r;
    }();
    return lib;
}

int rccl_status(int code, const char* what) {
    if (code == 0) return QS_OK;
    snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, rccl().error_string ? rccl().error_string(code) : "?");
    return QS_ERR_COMM;
}

constexpr int kMaxChunks = 16;

// balanced split of n rows over `world` ranks: the same rule as sharded.SlabPartition
inline int64_t part_lo(int64_t n, int world, int r) {
    const int64_t base = n / world, extra = n % world;
    return r * base + (r < extra ? r : extra);
}

// ---- the exchange plan: pure index arithmetic, shared by the executor below and by qs_sharded_exchange_plan (which
// lets a CPU test replay the plan of every rank with NumPy and check that the rows end up where the closing product
// reads them -- the part of this file that a one-GPU box cannot exercise with more than one rank)
struct PlanOp {
    int chunk, peer, kind;         // kind 0 = send (offset into X), 1 = receive (offset into R), 2 = own rows X -> R
    int64_t x_off, r_off, count;   // element offsets / count (own rows: count per row, `rows` rows, pitches below)
    int64_t rows;
};
struct Plan {
    int64_t b_lo, bl, p_lo, pc, row_x, row_r;       // row_x = bl*M*M elements of an X row, row_r = L*M*M of an R row
    int64_t chunk_slot0[kMaxChunks + 1];            // X rows (in exchange order) of chunk k: [slot0[k], slot0[k+1])
    int64_t close_lo[kMaxChunks], close_n[kMaxChunks];   // our result rows completed by chunk k (relative to p_lo)
    int64_t ct_row[1024];                           // global row of Ct in X slot i
    int nops;
    PlanOp ops[2 * kMaxChunks * 1024 / 1 > 65536 ? 65536 : 2 * kMaxChunks * 1024];
};

// rows of chunk k that belong to peer g: [rows_lo(g, k), rows_lo(g, k + 1))
inline int64_t rows_lo(int64_t M, int G, int nchunks, int g, int k) {
    const int64_t n = part_lo(M, G, g + 1) - part_lo(M, G, g);
    return part_lo(M, G, g) + part_lo(n, nchunks, k);
}

int build_plan(Plan& pl, int64_t L, int64_t M, int G, int me, int nchunks) {
    const int64_t MM = M * M;
    pl.b_lo = part_lo(L, G, me); pl.bl = part_lo(L, G, me + 1) - pl.b_lo;
    pl.p_lo = part_lo(M, G, me); pl.pc = part_lo(M, G, me + 1) - pl.p_lo;
    pl.row_x = pl.bl * MM; pl.row_r = L * MM;
    pl.nops = 0;
    int64_t slot = 0;
    for (int k = 0; k < nchunks; ++k) {
        pl.chunk_slot0[k] = slot;
        pl.close_lo[k] = rows_lo(M, G, nchunks, me, k) - pl.p_lo;
        pl.close_n[k] = rows_lo(M, G, nchunks, me, k + 1) - rows_lo(M, G, nchunks, me, k);
        for (int g = 0; g < G; ++g) {
            const int64_t lo = rows_lo(M, G, nchunks, g, k), n_send = rows_lo(M, G, nchunks, g, k + 1) - lo;
            const int64_t gb_lo = part_lo(L, G, g), gbl = part_lo(L, G, g + 1) - gb_lo;    // b range of rank g
            for (int64_t i = 0; i < n_send; ++i) pl.ct_row[slot + i] = lo + i;
            if (g == me) {
                if (n_send > 0 && pl.bl > 0) {
                    if (pl.nops >= (int)(sizeof(pl.ops) / sizeof(pl.ops[0]))) return QS_ERR_BAD_EXTENT;
                    pl.ops[pl.nops++] = PlanOp{k, g, 2, slot * pl.row_x, (pl.close_lo[k] * L + pl.b_lo) * MM, pl.row_x, n_send};
                }
            } else {
                // what we computed for peer g goes out row by row; what peer g computed for us comes in row by row
                for (int64_t i = 0; i < n_send && pl.bl > 0; ++i) {
                    if (pl.nops >= (int)(sizeof(pl.ops) / sizeof(pl.ops[0]))) return QS_ERR_BAD_EXTENT;
                    pl.ops[pl.nops++] = PlanOp{k, g, 0, (slot + i) * pl.row_x, 0, pl.row_x, 1};
                }
                for (int64_t i = 0; i < pl.close_n[k] && gbl > 0; ++i) {
                    if (pl.nops >= (int)(sizeof(pl.ops) / sizeof(pl.ops[0]))) return QS_ERR_BAD_EXTENT;
                    pl.ops[pl.nops++] = PlanOp{k, g, 1, 0, ((pl.close_lo[k] + i) * L + gb_lo) * MM, gbl * MM, 1};
                }
            }
            slot += n_send;
        }
    }
    pl.chunk_slot0[nchunks] = slot;
    return QS_OK;
}

}  // namespace

struct Comm {
    nccl_comm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;              // the exchange runs here, beside the caller's stream
    hipEvent_t x_ready[kMaxChunks], r_ready[kMaxChunks], idle, done;
    std::unique_ptr<Plan> plan;                // exchange plan of the most recent call (1.5 MB: on the heap, per handle)
    bool broken = false;                       // a call failed after its exchange had started: peers may be waiting in a
                                               // group this rank never completed -- only qs_comm_abort / destroy are left
    int rows_coalesce = 0;                     // qs_comm_set_option("rows_coalesce"): the rows exchange as ONE message per peer and step
};

// TEST HOOK (tuning key "comm_drop_wait", a bit mask, thread-local like every tuning key): leave out one of the waits
// that order the caller's stream and the communicator's stream.  The asynchronous stand-in transport of the test suite
// (tests/cabi/mock_rccl_async.cpp) must then produce WRONG results -- the proof that it would catch such a mistake.
//   1  rows: products of step t do not wait for the exchange of step t - 2 to have read their send block
//   2  rows: the exchange of a step does not wait for the step's products
//   4  rows: the closing products do not wait for the exchange
//   8  slab entry: the closing product of a chunk does not wait for the chunk's rows
//   16 slab entry: the exchange of a chunk does not wait for the chunk's products
inline bool dropped(int bit) { return (g_tune.comm_drop_wait & bit) != 0; }

// A failure once send / receive operations of this call may have been posted: the peers can be blocked in a group that
// this rank will never complete.  The handle is marked, so that every later call fails at once instead of dead-locking
// too, and the caller tears the communicator down (qs_comm_abort).
static int fail_mid_exchange(Comm* c, int rc) {
    c->broken = true;
    return rc;
}

}  // namespace qs

using namespace qs;

extern "C" {

const char* qs_last_comm_error(void) { return g_comm_err; }

int qs_comm_unique_id(void* id) {
    if (!id) return QS_ERR_NULL_POINTER;
    if (!rccl().ok) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so.1 could not be loaded: %s", dlerror());
        return QS_ERR_COMM;
    }
    NcclUniqueId uid;
    if (int rc = rccl_status(rccl().get_unique_id(&uid), "ncclGetUniqueId")) return rc;
    memcpy(id, &uid, sizeof(uid));
    return QS_OK;
}

int qs_comm_init(void** comm, int rank, int world, const void* unique_id) {
    if (!comm || !unique_id) return QS_ERR_NULL_POINTER;
    if (world < 1 || rank < 0 || rank >= world) return QS_ERR_BAD_EXTENT;
    if (!rccl().ok) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so.1 could not be loaded");
        return QS_ERR_COMM;
    }
    Comm* c = new Comm;
    c->rank = rank;
    c->world = world;
    c->device = current_device();
    NcclUniqueId uid;
    memcpy(&uid, unique_id, sizeof(uid));
    if (int rc = rccl_status(rccl().comm_init_rank(&c->nccl, world, uid, rank), "ncclCommInitRank")) {
        delete c;
        return rc;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < kMaxChunks && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->x_ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->r_ready[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->idle, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        rccl().comm_destroy(c->nccl);
        delete c;
        return hip_status(e, "qs_comm_init: stream / events");
    }
    *comm = c;
    return QS_OK;
}

int qs_comm_destroy(void* comm) {
    if (!comm) return QS_ERR_NULL_POINTER;
    Comm* c = (Comm*)comm;
    (void)hipStreamSynchronize(c->stream);
    int rc = rccl_status(rccl().comm_destroy(c->nccl), "ncclCommDestroy");
    for (int i = 0; i < kMaxChunks; ++i) {
        (void)hipEventDestroy(c->x_ready[i]);
        (void)hipEventDestroy(c->r_ready[i]);
    }
    (void)hipEventDestroy(c->idle);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
}

/* Exchange plan of one rank as numbers (no GPU, no RCCL): tests replay it on the CPU.  table: nops rows of
 * {chunk, peer, kind, x_off, r_off, count, rows}; header: {b_lo, bl, p_lo, pc, row_x, row_r, nchunks}; ct_rows: M;
 * chunks: per chunk {slot0, rows, close_lo, close_n}.  Returns the number of operations or a negative error. */
int qs_sharded_exchange_plan(int64_t L, int64_t M, int world, int rank, int nchunks, int64_t* header, int64_t* ct_rows,
                             int64_t* chunks, int64_t* table, int64_t table_rows) {
    if (L <= 0 || M <= 0 || M > 1024 || world < 1 || rank < 0 || rank >= world) return QS_ERR_BAD_EXTENT;
    if (!header || !ct_rows || !chunks || !table) return QS_ERR_NULL_POINTER;
    if (nchunks < 1) nchunks = 4;
    if (nchunks > kMaxChunks) nchunks = kMaxChunks;
    std::unique_ptr<Plan> holder(new (std::nothrow) Plan);
    if (!holder) return QS_ERR_WORKSPACE;
    Plan& plan = *holder;
    if (int rc = build_plan(plan, L, M, world, rank, nchunks)) return rc;
    if (plan.nops > table_rows) return QS_ERR_WORKSPACE;
    const int64_t h[7] = {plan.b_lo, plan.bl, plan.p_lo, plan.pc, plan.row_x, plan.row_r, nchunks};
    memcpy(header, h, sizeof(h));
    for (int64_t i = 0; i < M; ++i) ct_rows[i] = plan.ct_row[i];
    for (int k = 0; k < nchunks; ++k) {
        chunks[4 * k] = plan.chunk_slot0[k];
        chunks[4 * k + 1] = plan.chunk_slot0[k + 1] - plan.chunk_slot0[k];
        chunks[4 * k + 2] = plan.close_lo[k];
        chunks[4 * k + 3] = plan.close_n[k];
    }
    for (int i = 0; i < plan.nops; ++i) {
        const PlanOp& o = plan.ops[i];
        const int64_t row[7] = {o.chunk, o.peer, o.kind, o.x_off, o.r_off, o.count, o.rows};
        memcpy(table + 7 * i, row, sizeof(row));
    }
    return plan.nops;
}

/* Tear a communicator down WITHOUT waiting for outstanding operations (ncclCommAbort): what is left to do after a call
 * returned an error in the middle of its exchange, or when a peer died.  Frees the handle. */
int qs_comm_abort(void* comm) {
    if (!comm) return QS_ERR_NULL_POINTER;
    Comm* c = (Comm*)comm;
    int rc = QS_OK;
    if (rccl().comm_abort) rc = rccl_status(rccl().comm_abort(c->nccl), "ncclCommAbort");
    else rc = rccl_status(rccl().comm_destroy(c->nccl), "ncclCommDestroy");
    for (int i = 0; i < kMaxChunks; ++i) {
        (void)hipEventDestroy(c->x_ready[i]);
        (void)hipEventDestroy(c->r_ready[i]);
    }
    (void)hipEventDestroy(c->idle);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
}

int qs_comm_rank(void* comm) { return comm ? ((Comm*)comm)->rank : QS_ERR_NULL_POINTER; }
int qs_comm_world(void* comm) { return comm ? ((Comm*)comm)->world : QS_ERR_NULL_POINTER; }

// workspace: Ct rows in exchange order | C^T | T1 (reused as X) | T2 | R
int64_t qs_transform_two_body_sharded_workspace(int dtype, int64_t L, int64_t M, int world, int rank) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (L <= 0 || M <= 0 || L > 4096 || M > 1024 || world < 1 || rank < 0 || rank >= world) return QS_ERR_BAD_EXTENT;
    const int64_t bl = part_lo(L, world, rank + 1) - part_lo(L, world, rank);
    const int64_t pc = part_lo(M, world, rank + 1) - part_lo(M, world, rank);
    const int64_t t1 = L * bl * L * M, x = M * bl * M * M, t2 = L * bl * M * M, r = pc * L * M * M;
    const int64_t elems = 2 * ((L * M + 1) & ~int64_t(1)) + (t1 > x ? t1 : x) + t2 + r + 8;
    return elems * (int64_t)elem_size(dtype);
}

int qs_transform_two_body_sharded(void* comm, int dtype, const void* u_bslab, const void* C, const void* Ct,
                                  void* out_pslab, void* work, int64_t work_bytes, int64_t L, int64_t M, int nchunks,
                                  void* stream) {
    dispatch_reset();
    if (!comm) return QS_ERR_NULL_POINTER;
    Comm* c = (Comm*)comm;
    const int G = c->world, me = c->rank;
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (L <= 0 || M <= 0 || L > 4096 || M > 1024) return QS_ERR_BAD_EXTENT;
    if (!C || !Ct || !work) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype);
    const int64_t b_lo = part_lo(L, G, me), bl = part_lo(L, G, me + 1) - b_lo;
    const int64_t p_lo = part_lo(M, G, me), pc = part_lo(M, G, me + 1) - p_lo;
    if ((bl > 0 && !u_bslab) || (pc > 0 && !out_pslab)) return QS_ERR_NULL_POINTER;
    if (!aligned(C, es) || !aligned(Ct, es) || !aligned(work, 16) || (u_bslab && !aligned(u_bslab, es)) ||
        (out_pslab && !aligned(out_pslab, es)))
        return QS_ERR_MISALIGNED;
    if (out_pslab && (out_pslab == u_bslab || out_pslab == work)) return QS_ERR_ALIAS;
    if (work_bytes < qs_transform_two_body_sharded_workspace(dtype, L, M, G, me)) return QS_ERR_WORKSPACE;
    if (current_device() != c->device) return QS_ERR_BAD_EXTENT;      // the communicator belongs to another device
    if (nchunks < 1) nchunks = 4;
    if (nchunks > kMaxChunks) nchunks = kMaxChunks;
    hipStream_t s = (hipStream_t)stream;
    const int64_t MM = M * M, width = dtype == QS_C128 ? 2 : 1;

    auto at = [&](void* base, int64_t elems) { return (void*)((char*)base + (size_t)elems * es); };
    const int64_t lm = (L * M + 1) & ~int64_t(1);
    void* CtX = work;                      // rows of Ct in exchange order (chunk-major, then peer)
    void* CT = at(CtX, lm);
    void* T1 = at(CT, lm);                 // (L*bl*L, M); later X (M rows in exchange order) x (bl*M*M)
    const int64_t t1 = L * bl * L * M, xs = M * bl * MM;
    void* T2 = at(T1, t1 > xs ? t1 : xs);  // (L*bl, M, M)
    void* R = at(T2, L * bl * MM);         // (pc, L, M*M): row p_loc, columns b of every sender
    void* X = T1;

    if (c->broken) {
        snprintf(g_comm_err, sizeof(g_comm_err), "the communicator was left broken by an earlier failed call: abort / destroy it");
        return QS_ERR_COMM;
    }
    if (!c->plan) c->plan.reset(new (std::nothrow) Plan);      // (large: neither on the stack nor in thread-local storage)
    if (!c->plan) return QS_ERR_WORKSPACE;
    Plan& plan = *c->plan;
    if (M > 1024) return QS_ERR_BAD_EXTENT;
    if (int prc = build_plan(plan, L, M, G, me, nchunks)) return prc;
    // ---- Ct rows in exchange order (device-to-device row copies on the caller's stream: runs of consecutive rows)
    for (int64_t i = 0; i < M;) {
        int64_t n = 1;
        while (i + n < M && plan.ct_row[i + n] == plan.ct_row[i] + n) ++n;
        hipError_t ce = hipMemcpyAsync(at(CtX, i * L), (const char*)Ct + (size_t)(plan.ct_row[i] * L) * es,
                                       (size_t)(n * L) * es, hipMemcpyDeviceToDevice, s);
        if (ce != hipSuccess) return hip_status(ce, "qs_transform_two_body_sharded: Ct rows");
        i += n;
    }
    int rc = QS_OK;
    if (bl > 0) {
        rc = transpose_small(dtype, C, CT, L, M, s);
        if (rc) return rc;
        // d:  T1[(a,b,c), s] = u[(a,b,c), d] C[d, s]
        rc = matmul_checked(dtype, u_bslab, C, T1, L * bl * L, M, L, L, M, M, 1, 0, 0, 0, 0, s);
        if (rc) return rc;
        // c:  T2[(a,b)][r, s] = CT[r, c] T1[(a,b)][c, s]
        rc = matmul_checked(dtype, CT, T1, T2, M, M, L, L, M, M, L * bl, 0, L * M, MM, 0, s);
        if (rc) return rc;
    }
    // the exchange must not start before earlier work on the caller's stream that R / X might still be read by
    hipError_t e = hipEventRecord(c->idle, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->idle, 0);
    if (e != hipSuccess) return hip_status(e, "qs_transform_two_body_sharded: stream order");

    const int64_t row_x = plan.row_x;
    auto close_chunk = [&](int k) -> int {
        // out[p][q, (r,s)] = Ct[q, :] . R[p][:, (r,s)] for our rows of chunk k, once they are complete
        const int64_t lo = plan.close_lo[k], n = plan.close_n[k];
        if (n <= 0) return QS_OK;
        hipError_t ee = dropped(8) ? hipSuccess : hipStreamWaitEvent(s, c->r_ready[k], 0);
        if (ee != hipSuccess) return hip_status(ee, "qs_transform_two_body_sharded: wait for rows");
        return matmul_checked(dtype, Ct, at(R, lo * L * MM), at(out_pslab, lo * M * MM), M, MM, L, L, MM, MM, n, 0, L * MM,
                         M * MM, 0, s);
    };
    int op = 0;
    for (int k = 0; k < nchunks; ++k) {
        const int64_t slot0 = plan.chunk_slot0[k], rows_k = plan.chunk_slot0[k + 1] - slot0;
        // a:  X[slot, (b,r,s)] = CtX[slot, a] T2[a, (b,r,s)]   for the rows of chunk k
        if (bl > 0 && rows_k > 0) {
            rc = matmul_checked(dtype, at(CtX, slot0 * L), T2, at(X, slot0 * row_x), rows_k, row_x, L, L, row_x, row_x, 1, 0, 0,
                           0, 0, s);
            if (rc) return k ? fail_mid_exchange(c, rc) : rc;
        }
        e = hipEventRecord(c->x_ready[k], s);
        if (e == hipSuccess && !dropped(16)) e = hipStreamWaitEvent(c->stream, c->x_ready[k], 0);
        if (e != hipSuccess) {
            rc = hip_status(e, "qs_transform_two_body_sharded: chunk ready");
            return k ? fail_mid_exchange(c, rc) : rc;
        }
        // exchange of chunk k on the communicator's stream: one message per row and peer, all peers in one group
        if (int grc = rccl_status(rccl().group_start(), "ncclGroupStart")) return k ? fail_mid_exchange(c, grc) : grc;
        for (; op < plan.nops && plan.ops[op].chunk == k; ++op) {
            const PlanOp& o = plan.ops[op];
            int orc = QS_OK;
            if (o.kind == 2) {
                // our own rows: straight into R (strided copy: a row of X is bl*MM long, a row of R is L*MM long)
                hipError_t me_e = hipMemcpy2DAsync(at(R, o.r_off), (size_t)plan.row_r * es, at(X, o.x_off), (size_t)row_x * es,
                                                   (size_t)o.count * es, (size_t)o.rows, hipMemcpyDeviceToDevice, c->stream);
                if (me_e != hipSuccess) orc = hip_status(me_e, "own rows");
            } else if (o.kind == 0) {
                orc = rccl_status(rccl().send(at(X, o.x_off), (size_t)(o.count * width), kNcclFloat64, o.peer, c->nccl,
                                              c->stream), "ncclSend");
            } else {
                orc = rccl_status(rccl().recv(at(R, o.r_off), (size_t)(o.count * width), kNcclFloat64, o.peer, c->nccl,
                                              c->stream), "ncclRecv");
            }
            if (orc) { rccl().group_end(); return fail_mid_exchange(c, orc); }
        }
        if (int grc = rccl_status(rccl().group_end(), "ncclGroupEnd")) return fail_mid_exchange(c, grc);
        e = hipEventRecord(c->r_ready[k], c->stream);
        if (e != hipSuccess) return fail_mid_exchange(c, hip_status(e, "qs_transform_two_body_sharded: rows ready"));
        // while chunk k travels: close chunk k - 1 (its rows have arrived or are about to)
        if (k > 0) { rc = close_chunk(k - 1); if (rc) return fail_mid_exchange(c, rc); }
    }
    rc = close_chunk(nchunks - 1);
    if (rc) return rc;
    // the caller's stream ends behind everything the exchange stream did (workspace and R are free after `stream`)
    e = hipEventRecord(c->done, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, c->done, 0);
    if (e != hipSuccess) return hip_status(e, "qs_transform_two_body_sharded: join");
    note_dispatch("rccl grouped send/recv (%d chunks)", nchunks);
    return QS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Rows in, rows out: the memory-lean form (what sharded.transform_two_body_rows does with torch.distributed, as ONE
// call on RCCL).  This rank holds rows[i][j][c][d], its rows i of ONE leading index of u with the other one whole
// (u[a_lo + i, j] for a leading-index sharding, u[j, b_lo + i] for a second-index sharding: the transform is symmetric
// under swapping its two leading index pairs), and gets the rows j' it owns of the OTHER transformed leading index,
// out[j'_loc][i'][r][s].  Apart from those two only O(chunk_rows * l^3) exists at any time:
//
//   per step of chunk_rows input rows, on the caller's stream:
//     d, c   t2[i][j][r, s] = C[c, r] rows[i][j][c, d] C[d, s]
//     J      W[j', i, (r,s)] = Ct[j', j] t2[i][j, (r,s)]         W is stored [j'][i][(r,s)]: a peer's share is one block
//   on the communicator's stream, overlapping the NEXT step's products (W is double-buffered):
//     grouped ncclSend / ncclRecv, one message per (peer, j'): n*M*M contiguous elements on both sides -- no packing,
//     no staging -- landing at R[j'_loc][i_global ...][(r,s)] inside the result buffer
//   after the last step, on the caller's stream, row by row inside the result buffer:
//     I      out[j'_loc][i', (r,s)] = Ct[i', i] R[j'_loc][i, (r,s)]
//
// Result rows are packed from the start of the buffer; received rows sit behind a gap of one row (plus the growth
// (M - L) M^2 per row when M > L), so the product of row p never reaches a received row that is still to be read.
// ------------------------------------------------------------------------------------------------------------------

namespace qs {
namespace {

constexpr int64_t kRowsBudgetBytes = int64_t(8) << 30;      // scratch per rank (as sharded.STREAM_BUDGET_BYTES)

struct RowsGeom {
    int G, me;
    int64_t L, M, MM, il, jl, il_max, i_start;      // my input rows [i_start, i_start + il), my result rows jl
    int64_t r0;                                     // element offset of received row 0 in the result buffer
    int64_t out_elems;
};

// start of rank g's input rows: the caller's table (world + 1 entries) or the balanced split
inline int64_t in_lo(const int64_t* in_starts, int64_t L, int G, int g) { return in_starts ? in_starts[g] : part_lo(L, G, g); }

int rows_geometry(RowsGeom& q, int64_t L, int64_t M, int G, int me, const int64_t* in_starts) {
    if (L <= 0 || M <= 0 || L > 4096 || M > 1024 || G < 1 || me < 0 || me >= G) return QS_ERR_BAD_EXTENT;
    if (in_starts) {
        if (in_starts[0] != 0 || in_starts[G] != L) return QS_ERR_BAD_EXTENT;
        for (int g = 0; g < G; ++g) if (in_starts[g + 1] < in_starts[g]) return QS_ERR_BAD_EXTENT;
    }
    q.G = G; q.me = me; q.L = L; q.M = M; q.MM = M * M;
    q.i_start = in_lo(in_starts, L, G, me);
    q.il = in_lo(in_starts, L, G, me + 1) - q.i_start;
    q.jl = part_lo(M, G, me + 1) - part_lo(M, G, me);
    q.il_max = 0;
    for (int g = 0; g < G; ++g) {
        const int64_t n = in_lo(in_starts, L, G, g + 1) - in_lo(in_starts, L, G, g);
        if (n > q.il_max) q.il_max = n;
    }
    q.r0 = q.jl * (M > L ? M - L : 0) * q.MM + L * q.MM;
    q.out_elems = q.jl * (M > L ? M : L) * q.MM + L * q.MM;
    return QS_OK;
}

inline int64_t clamp_rows(int64_t have, int64_t i0, int64_t ni) {
    const int64_t n = have - i0;
    return n < 0 ? 0 : (n > ni ? ni : n);
}

// The operations of one step, in the order both sides of every pair post them (peer ascending; per peer the sends in
// ascending j', the receives in ascending j'_loc).  kind 0 send (offset into W), 1 receive (offset into the result
// buffer), 2 own rows (W -> buffer, `rows` pieces of `count` elements, pitches n*MM and L*MM).
// COALESCED form (one message per peer and step instead of one per peer and result row): a peer's share of W is one
// contiguous block anyway (W is stored [j'][i][(r,s)]), so its send is kind 0 with count = jc*n*MM; what the peer
// computed for us arrives as ONE block [j'_loc][i][(r,s)] in the staging area (kind 3: r_off = offset into the staging
// area) and is put in place by one strided copy on the communicator's stream behind the group (kind 4: x_off = offset
// into the staging area, r_off = offset into the result buffer, `rows` pieces of `count` elements, pitches count and L*MM).
template <typename F>
int rows_step_ops(const RowsGeom& q, const int64_t* in_starts, int64_t ni, int64_t step, bool coalesce, F&& emit) {
    const int64_t i0 = step * ni, n = clamp_rows(q.il, i0, ni), MM = q.MM;
    int64_t stage = 0;
    for (int g = 0; g < q.G; ++g) {
        const int64_t j_lo = part_lo(q.M, q.G, g), jc = part_lo(q.M, q.G, g + 1) - j_lo;
        const int64_t g_start = in_lo(in_starts, q.L, q.G, g);
        const int64_t ng = clamp_rows(in_lo(in_starts, q.L, q.G, g + 1) - g_start, i0, ni);   // rows rank g brings
        if (g == q.me) {
            if (n > 0 && q.jl > 0)
                if (int rc = emit(PlanOp{(int)step, g, 2, j_lo * n * MM, q.r0 + (q.i_start + i0) * MM, n * MM, q.jl})) return rc;
            continue;
        }
        if (coalesce) {
            if (jc > 0 && n > 0)
                if (int rc = emit(PlanOp{(int)step, g, 0, j_lo * n * MM, 0, jc * n * MM, 1})) return rc;
            if (q.jl > 0 && ng > 0) {
                if (int rc = emit(PlanOp{(int)step, g, 3, 0, stage, q.jl * ng * MM, 1})) return rc;
                stage += q.jl * ng * MM;
            }
            continue;
        }
        for (int64_t j = 0; j < jc && n > 0; ++j)
            if (int rc = emit(PlanOp{(int)step, g, 0, (j_lo + j) * n * MM, 0, n * MM, 1})) return rc;
        for (int64_t j = 0; j < q.jl && ng > 0; ++j)
            if (int rc = emit(PlanOp{(int)step, g, 1, 0, q.r0 + (j * q.L + g_start + i0) * MM, ng * MM, 1})) return rc;
    }
    if (coalesce) {      // behind the group: the received blocks into place
        stage = 0;
        for (int g = 0; g < q.G; ++g) {
            if (g == q.me) continue;
            const int64_t g_start = in_lo(in_starts, q.L, q.G, g);
            const int64_t ng = clamp_rows(in_lo(in_starts, q.L, q.G, g + 1) - g_start, i0, ni);
            if (q.jl > 0 && ng > 0) {
                if (int rc = emit(PlanOp{(int)step, g, 4, stage, q.r0 + (g_start + i0) * MM, ng * MM, q.jl})) return rc;
                stage += q.jl * ng * MM;
            }
        }
    }
    return QS_OK;
}

inline int64_t rows_default_chunk(int64_t L, int64_t M, int64_t il_max, size_t es) {
    int64_t unit = L * L * M;
    if (L * M * M > unit) unit = L * M * M;
    if (M * M * M > unit) unit = M * M * M;
    int64_t ni = kRowsBudgetBytes / (5 * unit * (int64_t)es);
    const int64_t quarter = (il_max + 3) / 4;          // at least four steps, so that the exchange has products to hide under
    if (ni > quarter) ni = quarter;
    return ni < 1 ? 1 : ni;
}

}  // namespace
}  // namespace qs

extern "C" {

/* chunk_rows the library picks when the caller passes <= 0 (a function of the GLOBAL extents only: every rank gets
 * the same number) */
int64_t qs_sharded_rows_default_chunk(int dtype, int64_t L, int64_t M, int world, const int64_t* in_starts) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    RowsGeom q;
    if (int rc = rows_geometry(q, L, M, world, 0, in_starts)) return rc;
    return rows_default_chunk(L, M, q.il_max, elem_size(dtype));
}

/* bytes of the result buffer (the result rows are its first jl * M^3 elements) */
int64_t qs_transform_two_body_sharded_rows_out_bytes(int dtype, int64_t L, int64_t M, int world, int rank) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    RowsGeom q;
    if (int rc = rows_geometry(q, L, M, world, rank, nullptr)) return rc;
    return q.out_elems * (int64_t)elem_size(dtype);
}

/* workspace: C^T | t1 | t2 | W0 | W1 */
int64_t qs_transform_two_body_sharded_rows_workspace(int dtype, int64_t L, int64_t M, int64_t chunk_rows) {
    if (!dtype_ok(dtype)) return QS_ERR_BAD_DTYPE;
    if (L <= 0 || M <= 0 || L > 4096 || M > 1024 || chunk_rows < 1 || chunk_rows > L) return QS_ERR_BAD_EXTENT;
    const int64_t lm = (L * M + 1) & ~int64_t(1), ni = chunk_rows, MM = M * M;
    return (lm + ni * L * L * M + ni * L * MM + 2 * M * ni * MM + 8) * (int64_t)elem_size(dtype);
}

/* ... for THIS handle: with the option "rows_coalesce" the staging area of the coalesced exchange follows (what the
 * peers send this rank in one step: at most jl (world - 1) chunk_rows M^2 elements, about one more send block) */
int64_t qs_comm_rows_workspace(void* comm, int dtype, int64_t L, int64_t M, int64_t chunk_rows) {
    if (!comm) return QS_ERR_NULL_POINTER;
    const Comm* c = (const Comm*)comm;
    const int64_t base = qs_transform_two_body_sharded_rows_workspace(dtype, L, M, chunk_rows);
    if (base < 0 || !c->rows_coalesce) return base;
    const int64_t jl = part_lo(M, c->world, c->rank + 1) - part_lo(M, c->world, c->rank);
    return base + jl * (c->world - 1) * chunk_rows * M * M * (int64_t)elem_size(dtype);
}

/* Per-handle options.  "rows_coalesce" = 1: qs_transform_two_body_sharded_rows exchanges ONE message per peer and step
 * (the peer's block of the send buffer as it is; the received block goes through a staging area of the workspace and is
 * put in place by one strided copy on the communicator's stream) instead of one message per peer and result row that
 * lands in place.  Same results bit for bit; every rank of the communicator must choose the same. */
int qs_comm_set_option(void* comm, const char* key, int64_t value) {
    if (!comm || !key) return QS_ERR_NULL_POINTER;
    Comm* c = (Comm*)comm;
    if (!strcmp(key, "rows_coalesce")) { c->rows_coalesce = value != 0; return QS_OK; }
    return QS_ERR_BAD_EXTENT;
}

/* The exchange plan of one rank as numbers (no GPU, no RCCL): the CPU suite replays the plans of every rank of a world
 * with NumPy.  header: {i_start, il, jl, il_max, r0, out_elems, chunk_rows, nsteps}; table: one row
 * {step, peer, kind, w_off, buf_off, count, rows} per operation.  Returns the number of operations or an error. */
static int rows_exchange_plan(int64_t L, int64_t M, int world, int rank, const int64_t* in_starts, int64_t chunk_rows,
                              bool coalesce, int64_t* header, int64_t* table, int64_t table_rows) {
    if (!header || !table) return QS_ERR_NULL_POINTER;
    RowsGeom q;
    if (int rc = rows_geometry(q, L, M, world, rank, in_starts)) return rc;
    const int64_t ni = chunk_rows >= 1 ? (chunk_rows > q.il_max ? q.il_max : chunk_rows) : rows_default_chunk(L, M, q.il_max, 8);
    const int64_t nsteps = (q.il_max + ni - 1) / ni;
    const int64_t h[8] = {q.i_start, q.il, q.jl, q.il_max, q.r0, q.out_elems, ni, nsteps};
    memcpy(header, h, sizeof(h));
    int64_t nops = 0;
    for (int64_t t = 0; t < nsteps; ++t) {
        int rc = rows_step_ops(q, in_starts, ni, t, coalesce, [&](const PlanOp& o) -> int {
            if (nops >= table_rows) return QS_ERR_WORKSPACE;
            const int64_t row[7] = {o.chunk, o.peer, o.kind, o.x_off, o.r_off, o.count, o.rows};
            memcpy(table + 7 * nops, row, sizeof(row));
            ++nops;
            return QS_OK;
        });
        if (rc) return rc;
    }
    return (int)nops;
}

int qs_sharded_rows_exchange_plan(int64_t L, int64_t M, int world, int rank, const int64_t* in_starts, int64_t chunk_rows,
                                  int64_t* header, int64_t* table, int64_t table_rows) {
    return rows_exchange_plan(L, M, world, rank, in_starts, chunk_rows, false, header, table, table_rows);
}

/* ... of the coalesced exchange (qs_comm_set_option "rows_coalesce"): kind 0 send (one per peer), 3 receive into the
 * staging area (buf_off = offset into it), 4 staging -> out_buffer behind the group (w_off = offset into the staging
 * area, `rows` pieces of `count` elements, pitches count and L*M*M), 2 own rows as before. */
int qs_sharded_rows_exchange_plan_coalesced(int64_t L, int64_t M, int world, int rank, const int64_t* in_starts,
                                            int64_t chunk_rows, int64_t* header, int64_t* table, int64_t table_rows) {
    return rows_exchange_plan(L, M, world, rank, in_starts, chunk_rows, true, header, table, table_rows);
}

int qs_transform_two_body_sharded_rows(void* comm, int in_dtype, int dtype, const void* rows, const int64_t* in_starts,
                                       const void* C, const void* Ct, void* out_buffer, int64_t out_bytes, void* work,
                                       int64_t work_bytes, int64_t L, int64_t M, int64_t chunk_rows, void* stream) {
    dispatch_reset();
    if (!comm) return QS_ERR_NULL_POINTER;
    Comm* c = (Comm*)comm;
    if (!dtype_ok(dtype) || !dtype_ok(in_dtype) || (in_dtype == QS_C128 && dtype == QS_F64)) return QS_ERR_BAD_DTYPE;
    RowsGeom q;
    if (int rc = rows_geometry(q, L, M, c->world, c->rank, in_starts)) return rc;
    if (!C || !Ct || !work || !out_buffer) return QS_ERR_NULL_POINTER;
    if (q.il > 0 && !rows) return QS_ERR_NULL_POINTER;
    const size_t es = elem_size(dtype), ies = elem_size(in_dtype);
    if (!aligned(C, es) || !aligned(Ct, es) || !aligned(work, 16) || !aligned(out_buffer, 16) || (rows && !aligned(rows, ies)))
        return QS_ERR_MISALIGNED;
    if (out_buffer == rows || out_buffer == work) return QS_ERR_ALIAS;
    const int64_t ni = chunk_rows >= 1 ? (chunk_rows > q.il_max ? q.il_max : chunk_rows)
                                       : rows_default_chunk(L, M, q.il_max, es);
    if (out_bytes < q.out_elems * (int64_t)es) return QS_ERR_WORKSPACE;
    if (work_bytes < qs_comm_rows_workspace(comm, dtype, L, M, ni)) return QS_ERR_WORKSPACE;
    if (current_device() != c->device) return QS_ERR_BAD_EXTENT;      // the communicator belongs to another device
    if (c->broken) {
        snprintf(g_comm_err, sizeof(g_comm_err), "the communicator was left broken by an earlier failed call: abort / destroy it");
        return QS_ERR_COMM;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t MM = q.MM, width = dtype == QS_C128 ? 2 : 1;
    const int64_t nsteps = (q.il_max + ni - 1) / ni;
    auto at = [&](void* base, int64_t elems) { return (void*)((char*)base + (size_t)elems * es); };
    const int64_t lm = (L * M + 1) & ~int64_t(1);
    void* CT = work;
    void* T1 = at(CT, lm);
    void* T2 = at(T1, ni * L * L * M);
    void* W[2] = {at(T2, ni * L * MM), at(T2, ni * L * MM + M * ni * MM)};
    void* S = at(T2, ni * L * MM + 2 * M * ni * MM);     // staging of the coalesced exchange
    const bool coalesce = c->rows_coalesce != 0;
    bool posted = false;                                 // anything handed to RCCL yet?
    auto fail = [&](int rc) { return posted ? fail_mid_exchange(c, rc) : rc; };

    int rc = transpose_small(dtype, C, CT, L, M, s);
    if (rc) return rc;
    for (int64_t t = 0; t < nsteps; ++t) {
        const int w = (int)(t & 1);
        const int64_t i0 = t * ni, n = clamp_rows(q.il, i0, ni);
        hipError_t e = hipSuccess;
        // W[w] was last read by the exchange of step t - 2
        if (t >= 2 && !dropped(1)) e = hipStreamWaitEvent(s, c->r_ready[w], 0);
        if (e != hipSuccess) return fail(hip_status(e, "qs_transform_two_body_sharded_rows: send buffer free"));
        if (n > 0) {
            const void* src = (const char*)rows + (size_t)(i0 * L * L * L) * ies;
            // d:  t1[(i,j,c), s] = rows[(i,j,c), d] C[d, s]        (a real tensor against complex coefficients: the mixed product)
            rc = in_dtype == dtype ? matmul_checked(dtype, src, C, T1, n * L * L, M, L, L, M, M, 1, 0, 0, 0, 0, s)
                                   : matmul_real_by_complex(src, C, T1, n * L * L, M, L, L, M, M, s);
            if (rc) return fail(rc);
            // c:  t2[(i,j)][r, s] = CT[r, c] t1[(i,j)][c, s]
            rc = matmul_checked(dtype, CT, T1, T2, M, M, L, L, M, M, n * L, 0, L * M, MM, 0, s);
            if (rc) return fail(rc);
            // J:  W[j', i, (r,s)] = Ct[j', j] t2[i][j, (r,s)]     one product per row i, rows of W n*MM apart
            rc = matmul_checked(dtype, Ct, T2, W[w], M, MM, L, L, MM, n * MM, n, 0, L * MM, MM, 0, s);
            if (rc) return fail(rc);
        }
        e = hipEventRecord(c->x_ready[w], s);
        if (e == hipSuccess && !dropped(2)) e = hipStreamWaitEvent(c->stream, c->x_ready[w], 0);
        if (e != hipSuccess) return fail(hip_status(e, "qs_transform_two_body_sharded_rows: chunk ready"));
        if (int grc = rccl_status(rccl().group_start(), "ncclGroupStart")) return fail(grc);
        posted = true;
        bool group_open = true;
        rc = rows_step_ops(q, in_starts, ni, t, coalesce, [&](const PlanOp& o) -> int {
            if (o.kind == 2) {
                hipError_t ce = hipMemcpy2DAsync(at(out_buffer, o.r_off), (size_t)(L * MM) * es, at(W[w], o.x_off),
                                                 (size_t)o.count * es, (size_t)o.count * es, (size_t)o.rows,
                                                 hipMemcpyDeviceToDevice, c->stream);
                return ce == hipSuccess ? QS_OK : hip_status(ce, "qs_transform_two_body_sharded_rows: own rows");
            }
            if (o.kind == 0)
                return rccl_status(rccl().send(at(W[w], o.x_off), (size_t)(o.count * width), kNcclFloat64, o.peer, c->nccl,
                                               c->stream), "ncclSend");
            if (o.kind == 1)
                return rccl_status(rccl().recv(at(out_buffer, o.r_off), (size_t)(o.count * width), kNcclFloat64, o.peer, c->nccl,
                                               c->stream), "ncclRecv");
            if (o.kind == 3)
                return rccl_status(rccl().recv(at(S, o.r_off), (size_t)(o.count * width), kNcclFloat64, o.peer, c->nccl,
                                               c->stream), "ncclRecv");
            // kind 4: the group is complete (these follow every send / receive of the step): received blocks into place
            if (group_open) {
                group_open = false;
                if (int grc = rccl_status(rccl().group_end(), "ncclGroupEnd")) return grc;
            }
            hipError_t ce = hipMemcpy2DAsync(at(out_buffer, o.r_off), (size_t)(L * MM) * es, at(S, o.x_off), (size_t)o.count * es,
                                             (size_t)o.count * es, (size_t)o.rows, hipMemcpyDeviceToDevice, c->stream);
            return ce == hipSuccess ? QS_OK : hip_status(ce, "qs_transform_two_body_sharded_rows: staged rows");
        });
        if (rc) { if (group_open) rccl().group_end(); return fail(rc); }
        if (group_open)
            if (int grc = rccl_status(rccl().group_end(), "ncclGroupEnd")) return fail(grc);
        e = hipEventRecord(c->r_ready[w], c->stream);
        if (e != hipSuccess) return fail(hip_status(e, "qs_transform_two_body_sharded_rows: chunk sent"));
    }
    // every received row is complete only now: the closing products follow the whole exchange
    hipError_t e = hipEventRecord(c->done, c->stream);
    if (e == hipSuccess && !dropped(4)) e = hipStreamWaitEvent(s, c->done, 0);
    if (e != hipSuccess) return fail(hip_status(e, "qs_transform_two_body_sharded_rows: join"));
    // I:  out[p][i', (r,s)] = Ct[i', i] R[p][i, (r,s)], row by row, packed from the start of the buffer
    for (int64_t p = 0; p < q.jl; ++p) {
        rc = matmul_checked(dtype, Ct, at(out_buffer, q.r0 + p * L * MM), at(out_buffer, p * M * MM), M, MM, L, L, MM, MM, 1,
                            0, 0, 0, 0, s);
        if (rc) return rc;
    }
    note_dispatch("rccl grouped send/recv (%lld steps of %lld rows%s)", (long long)nsteps, (long long)ni,
                  coalesce ? ", one message per peer and step" : "");
    return QS_OK;
}

}  // extern "C"
