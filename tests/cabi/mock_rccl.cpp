// A stand-in for librccl.so.1, for TESTS ONLY: the handful of entry points qs_comm.hip resolves at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommAbort, ncclGroupStart / End, ncclSend, ncclRecv,
// ncclGetErrorString) with a transport made of files in a temporary directory, so that the MULTI-RANK branches of the
// library's sharded entry points -- which a one-GPU development box can never run on real RCCL (one rank per device) --
// execute with several processes on one GPU: every ncclSend / ncclRecv the library posts is carried out, with the
// matching rules RCCL enforces (the k-th send of rank a to rank b pairs with the k-th receive of b from a; sizes must
// agree), on the library's real device buffers.  What it does NOT model: asynchrony (every operation completes before
// the call that ends its group returns; the stream it was posted on is drained first), links, performance.
// Built by tests/test_gpu_mock_rccl_ranks.py into a private directory that only that test puts on the loader path.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };

struct Comm {
    int rank, world;
    std::string dir;
    std::vector<uint64_t> sent, received;      // per peer: messages so far (the pairing order)
};

struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

enum { kSuccess = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

std::string path_of(const Comm* c, int src, int dst, uint64_t seq) {
    char name[64];
    snprintf(name, sizeof(name), "/m_%d_%d_%llu", src, dst, (unsigned long long)seq);
    return c->dir + name;
}

int do_send(const Op& o) {
    if (hipStreamSynchronize(o.stream) != hipSuccess) return kUnhandled;
    std::vector<char> host(o.bytes);
    if (o.bytes && hipMemcpy(host.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandled;
    const std::string final_path = path_of(o.comm, o.comm->rank, o.peer, o.comm->sent[o.peer]++), tmp = final_path + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return kSystem;
    const uint64_t n = o.bytes;
    bool ok = fwrite(&n, 8, 1, f) == 1 && (o.bytes == 0 || fwrite(host.data(), 1, o.bytes, f) == o.bytes);
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), final_path.c_str()) != 0) return kSystem;      // (rename: the receiver never sees half a file)
    return kSuccess;
}

int do_recv(const Op& o) {
    const std::string p = path_of(o.comm, o.peer, o.comm->rank, o.comm->received[o.peer]++);
    FILE* f = nullptr;
    for (int waited_ms = 0; !(f = fopen(p.c_str(), "rb")); waited_ms += 2) {
        if (waited_ms > 120000) { fprintf(stderr, "mock rccl: rank %d waited 120 s for %s\n", o.comm->rank, p.c_str()); return kSystem; }
        usleep(2000);
    }
    uint64_t n = 0;
    std::vector<char> host(o.bytes);
    bool ok = fread(&n, 8, 1, f) == 1;
    if (ok && n != o.bytes) {      // RCCL: send and receive of a pair must agree in size
        fprintf(stderr, "mock rccl: rank %d expects %zu bytes from %d, the message has %llu\n", o.comm->rank, o.bytes, o.peer,
                (unsigned long long)n);
        fclose(f);
        return kInvalidUsage;
    }
    ok = ok && (o.bytes == 0 || fread(host.data(), 1, o.bytes, f) == o.bytes);
    fclose(f);
    unlink(p.c_str());
    if (!ok) return kSystem;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return kUnhandled;
    if (o.bytes && hipMemcpy(o.buf, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandled;
    return kSuccess;
}

int run(std::vector<Op>& ops) {
    int rc = kSuccess;
    for (const Op& o : ops) if (o.send && rc == kSuccess) rc = do_send(o);        // all sends first: nobody blocks on a peer
    for (const Op& o : ops) if (!o.send && rc == kSuccess) rc = do_recv(o);
    ops.clear();
    return rc;
}

int post(bool send, const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return kInvalidArgument;
    if (dtype != 8) return kInvalidArgument;                                      // ncclFloat64: all the library sends
    g_ops.push_back(Op{send, const_cast<void*>(buf), count * 8, peer, c, stream});
    return g_depth ? kSuccess : run(g_ops);
}

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
    memset(id, 0, sizeof(*id));
    const char* base = getenv("QS_MOCK_RCCL_DIR");
    snprintf(id->internal, sizeof(id->internal), "%s/c_%d_%ld", base ? base : "/tmp", (int)getpid(), (long)time(nullptr));
    return kSuccess;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[127] = 0;
    Comm* c = new Comm{rank, nranks, std::string(id.internal), std::vector<uint64_t>(nranks, 0), std::vector<uint64_t>(nranks, 0)};
    mkdir(c->dir.c_str(), 0700);                                                   // (every rank tries; one wins)
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(void* comm) { delete (Comm*)comm; return kSuccess; }
int ncclCommAbort(void* comm) { delete (Comm*)comm; return kSuccess; }
int ncclGroupStart(void) { ++g_depth; return kSuccess; }
int ncclGroupEnd(void) {
    if (g_depth <= 0) return kInvalidUsage;
    return --g_depth ? kSuccess : run(g_ops);
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    return post(true, buf, count, dtype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    return post(false, buf, count, dtype, peer, comm, stream);
}
const char* ncclGetErrorString(int code) {
    switch (code) {
        case kSuccess: return "no error";
        case kInvalidUsage: return "mock rccl: invalid usage (sizes of a send / receive pair differ)";
        case kInvalidArgument: return "mock rccl: invalid argument";
        case kSystem: return "mock rccl: transport error";
        default: return "mock rccl: error";
    }
}

}  // extern "C"
