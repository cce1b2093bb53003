// A stand-in for librccl.so.1 for ONE measurement: one rank of an N-rank job alone on a one-GPU box (tools/config4_one_rank.py).
// The peers do not exist: a send is dropped, a receive fills its buffer with zeros on the stream it was posted on -- which is what
// the exchange would deliver if every other rank held rows of zeros, so the result of the call is still checkable (the transform
// of a tensor whose only non-zero leading rows are this rank's).  Same entry points as tests/cabi/mock_rccl.cpp; no pairing, no links.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

namespace {
struct UniqueId { char internal[128]; };
struct Comm { int rank, world; };
size_t type_bytes(int dtype) {      // ncclDataType_t: 0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double, 9 bfloat16
    switch (dtype) { case 0: case 1: return 1; case 6: case 9: return 2; case 2: case 3: case 7: return 4; default: return 8; }
}
uint64_t g_sent = 0, g_received = 0;
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId* id) { memset(id, 0, sizeof(*id)); id->internal[0] = 'a'; return 0; }
int ncclCommInitRank(void** comm, int nranks, UniqueId, int rank) { *comm = new Comm{rank, nranks}; return 0; }
int ncclCommDestroy(void* comm) { delete (Comm*)comm; return 0; }
int ncclCommAbort(void* comm) { delete (Comm*)comm; return 0; }
int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }
int ncclSend(const void*, size_t count, int dtype, int, void*, hipStream_t) { g_sent += count * type_bytes(dtype); return 0; }
int ncclRecv(void* buf, size_t count, int dtype, int, void*, hipStream_t stream) {
    g_received += count * type_bytes(dtype);
    return hipMemsetAsync(buf, 0, count * type_bytes(dtype), stream) == hipSuccess ? 0 : 1;
}
const char* ncclGetErrorString(int code) { return code ? "absent-peers stand-in: hipMemsetAsync failed" : "no error"; }
// what the rank would have put on / taken off its links (bytes since load)
uint64_t absent_peers_sent_bytes(void) { return g_sent; }
uint64_t absent_peers_received_bytes(void) { return g_received; }
}  // extern "C"
