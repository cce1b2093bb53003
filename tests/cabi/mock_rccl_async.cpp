// A STREAM-ORDERED, ASYNCHRONOUS stand-in for librccl.so.1, for TESTS ONLY (VERDICT r03 "next" 1a).
//
// tests/cabi/mock_rccl.cpp carries the library's ncclSend / ncclRecv calls between rank PROCESSES through files and
// finishes every operation on the host before ncclGroupEnd returns: it proves pairing, sizes and offsets, and nothing about
// the stream and event logic that lets the exchange of one step run under the products of the next
// (qs_comm.hip: the waits on r_ready / x_ready / done).  This one models what RCCL does with that logic:
//
//   * the ranks are THREADS of one process (the library's state is thread-local or per handle), all on the same device;
//   * ncclGroupEnd returns BEFORE anything has moved.  It only (1) records an event on every stream the group was posted
//     on ("everything enqueued on this stream before the group is done": the send buffer is ready, the receive buffer is
//     free), (2) pairs the k-th send of rank a to rank b with the k-th receive of b from a -- waiting ON THE HOST for the
//     peer thread to POST its half, never for device work -- and (3) makes the posting stream wait for the transfer's
//     completion event;
//   * a transfer is hipMemcpyAsync(device to device) on a transfer stream of its own (one per ordered pair of ranks), behind
//     the events of BOTH sides, optionally behind a delay kernel (QS_MOCK_RCCL_DELAY_US: the link is slow, the copy reads
//     its source late), followed by the completion event.
//
// So the only thing that keeps a send buffer intact until it has been read, or a received row unread until it has arrived,
// is the caller's own stream ordering -- exactly as with RCCL.  tests/test_gpu_async_transport.py shows that the suite
// FAILS when one of the library's waits is left out (tuning key "comm_drop_wait"), and passes with all of them.
// Not modelled: links, performance, RCCL's internal channels and protocols.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };
enum { kSuccess = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

// the link is slow: hold the transfer stream for `ticks` of the 100 MHz counter (an exit condition every lane reaches)
__global__ void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

struct Transfer {
    const void* src = nullptr;
    void* dst = nullptr;
    size_t send_bytes = 0, recv_bytes = 0;
    hipEvent_t src_ready = nullptr, dst_ready = nullptr, done = nullptr;
    bool have_send = false, have_recv = false, launched = false, bad = false;
};

struct World;
struct Comm {
    int rank = 0, world = 0;
    std::shared_ptr<World> w;
    std::vector<hipEvent_t> events;                // everything this rank created: destroyed with the communicator
};

struct World {
    int n = 0, joined = 0, left = 0;
    long long delay_ticks = 0;
    std::mutex m;
    std::condition_variable cv;
    // per ordered pair (src, dst): the transfers in pairing order, how many each side has posted, the pair's stream
    std::vector<std::vector<std::shared_ptr<Transfer>>> q;
    std::vector<size_t> n_sent, n_recv;
    std::vector<hipStream_t> xfer;
    std::atomic<long long> transfers{0};
};

std::mutex g_registry_mutex;
std::map<std::string, std::shared_ptr<World>> g_registry;
std::atomic<int> g_id_counter{0};

struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int launch(World& w, int pair, Transfer& t) {      // (under w.m) both halves are there: enqueue the copy
    if (t.send_bytes != t.recv_bytes) {
        fprintf(stderr, "mock rccl (async): pair %d -> %d: send of %zu bytes meets a receive of %zu\n", pair / w.n, pair % w.n,
                t.send_bytes, t.recv_bytes);
        t.bad = true;
        t.launched = true;
        return kInvalidUsage;
    }
    hipStream_t x = w.xfer[pair];
    hipError_t e = hipStreamWaitEvent(x, t.src_ready, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(x, t.dst_ready, 0);
    if (e == hipSuccess && w.delay_ticks > 0) {
        hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, x, w.delay_ticks);
        e = hipGetLastError();
    }
    if (e == hipSuccess && t.send_bytes) e = hipMemcpyAsync(t.dst, t.src, t.send_bytes, hipMemcpyDeviceToDevice, x);
    if (e == hipSuccess) e = hipEventRecord(t.done, x);
    t.launched = true;
    t.bad = e != hipSuccess;
    ++w.transfers;
    return e == hipSuccess ? kSuccess : kUnhandled;
}

int run(std::vector<Op>& ops) {
    if (ops.empty()) return kSuccess;
    Comm* c = ops[0].comm;
    World& w = *c->w;
    int rc = kSuccess;
    // (1) one "ready" event per stream of the group
    std::vector<std::pair<hipStream_t, hipEvent_t>> ready;
    auto ready_of = [&](hipStream_t s) -> hipEvent_t {
        for (auto& p : ready) if (p.first == s) return p.second;
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, s) != hipSuccess) { rc = kUnhandled; return nullptr; }
        c->events.push_back(ev);
        ready.push_back({s, ev});
        return ev;
    };
    // (2) post every operation of the group, launching a transfer when its second half arrives
    std::vector<std::shared_ptr<Transfer>> mine;
    {
        std::unique_lock<std::mutex> lock(w.m);
        for (const Op& o : ops) {
            hipEvent_t ev = ready_of(o.stream);
            if (!ev) break;
            const int pair = o.send ? c->rank * w.n + o.peer : o.peer * w.n + c->rank;
            size_t& k = o.send ? w.n_sent[pair] : w.n_recv[pair];
            if (w.q[pair].size() <= k) w.q[pair].resize(k + 1);
            if (!w.q[pair][k]) w.q[pair][k] = std::make_shared<Transfer>();
            Transfer& t = *w.q[pair][k];
            if (o.send) { t.src = o.buf; t.send_bytes = o.bytes; t.src_ready = ev; t.have_send = true; }
            else {
                t.dst = o.buf; t.recv_bytes = o.bytes; t.dst_ready = ev; t.have_recv = true;
                if (hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) { rc = kUnhandled; break; }
                c->events.push_back(t.done);
            }
            mine.push_back(w.q[pair][k]);
            ++k;
            if (t.have_send && t.have_recv) {
                const int lrc = launch(w, pair, t);
                if (lrc != kSuccess && rc == kSuccess) rc = lrc;
            }
        }
        w.cv.notify_all();
        // (3) the peers' halves: wait on the HOST until every transfer of this group has been enqueued (the peer thread
        // has posted its group), never for the device
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        for (auto& t : mine) {
            while (!t->launched) {
                if (w.cv.wait_until(lock, deadline) == std::cv_status::timeout && !t->launched) {
                    fprintf(stderr, "mock rccl (async): rank %d waited 120 s for a peer to post its half of a transfer\n", c->rank);
                    ops.clear();
                    return kSystem;
                }
            }
            if (t->bad && rc == kSuccess) rc = kInvalidUsage;
        }
    }
    // (4) the posting streams continue behind their transfers (a send buffer may be reused, a received row read)
    if (rc == kSuccess)
        for (size_t i = 0; i < ops.size(); ++i)
            if (hipStreamWaitEvent(ops[i].stream, mine[i]->done, 0) != hipSuccess) rc = kUnhandled;
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
    snprintf(id->internal, sizeof(id->internal), "async_%d_%d", (int)getpid(), g_id_counter.fetch_add(1));
    return kSuccess;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[127] = 0;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> g(g_registry_mutex);
        auto& slot = g_registry[std::string(id.internal)];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->n = nranks;
            slot->q.resize((size_t)nranks * nranks);
            slot->n_sent.assign((size_t)nranks * nranks, 0);
            slot->n_recv.assign((size_t)nranks * nranks, 0);
            slot->xfer.assign((size_t)nranks * nranks, nullptr);
            if (const char* d = getenv("QS_MOCK_RCCL_DELAY_US")) slot->delay_ticks = atoll(d) * 100;
            for (auto& s : slot->xfer)
                if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return kUnhandled;
        }
        w = slot;
        if (w->n != nranks) return kInvalidArgument;
    }
    Comm* c = new Comm;
    c->rank = rank; c->world = nranks; c->w = w;
    {      // collective, as ncclCommInitRank is: return when every rank has joined
        std::unique_lock<std::mutex> lock(w->m);
        ++w->joined;
        w->cv.notify_all();
        if (!w->cv.wait_for(lock, std::chrono::seconds(120), [&] { return w->joined >= nranks; })) { delete c; return kSystem; }
    }
    *comm = c;
    return kSuccess;
}

static int leave(Comm* c, bool wait) {
    if (!c) return kInvalidArgument;
    World& w = *c->w;
    if (wait) (void)hipDeviceSynchronize();
    bool last;
    {
        std::lock_guard<std::mutex> lock(w.m);
        last = ++w.left == w.n;
    }
    if (last) {      // the last rank out frees what the world shares (transfers of the others may still hold events of this rank until then)
        (void)hipDeviceSynchronize();
        for (auto& s : w.xfer) if (s) (void)hipStreamDestroy(s);
        if (getenv("QS_MOCK_RCCL_VERBOSE")) fprintf(stderr, "mock rccl (async): %lld transfers\n", w.transfers.load());
        std::lock_guard<std::mutex> g(g_registry_mutex);
        for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
            if (it->second.get() == &w) { g_registry.erase(it); break; }
    }
    // events are tiny and a peer's transfer may still name them: they live until the process ends
    delete c;
    return kSuccess;
}

int ncclCommDestroy(void* comm) { return leave((Comm*)comm, true); }
int ncclCommAbort(void* comm) { return leave((Comm*)comm, false); }
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
        case kInvalidUsage: return "mock rccl (async): invalid usage (sizes of a send / receive pair differ)";
        case kInvalidArgument: return "mock rccl (async): invalid argument";
        case kSystem: return "mock rccl (async): a peer never posted its half";
        default: return "mock rccl (async): error";
    }
}

}  // extern "C"
