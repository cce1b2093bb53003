// A stand-in for librccl.so.1 for ONE measurement: one rank of an N-rank job alone on a one-GPU box (tools/config4_one_rank.py).
// The peers do not exist: a send is dropped, a receive fills its buffer with zeros on the stream it was posted on -- which is what
// the exchange would deliver if every other rank held rows of zeros, so the result of the call is still checkable (the transform
// of a tensor whose only non-zero leading rows are this rank's).  Same entry points as tests/cabi/mock_rccl.cpp; no pairing.
// Links: none by default.  With ABSENT_PEERS_LINK_GBS = x every group of sends / receives holds the stream it was posted on for
// (the largest number of bytes any ONE peer sends or receives in the group) / (x GB/s) -- the peers' links are separate and full
// duplex (xGMI is point to point), so the slowest pair bounds the group -- behind the zero fills.  A model of the link TIME only
// (no protocol, no channels, no contention with the products for anything but the stream order): what it shows is how much of
// that time the library's stream pipeline hides.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>

namespace {
struct UniqueId { char internal[128]; };
struct Comm { int rank, world; };
size_t type_bytes(int dtype) {      // ncclDataType_t: 0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double, 9 bfloat16
    switch (dtype) { case 0: case 1: return 1; case 6: case 9: return 2; case 2: case 3: case 7: return 4; default: return 8; }
}
uint64_t g_sent = 0, g_received = 0;
// the open group: bytes per peer and direction, the stream of its operations
thread_local std::map<int, uint64_t> g_out, g_in;
thread_local hipStream_t g_stream = nullptr;
thread_local int g_depth = 0;
double link_bytes_per_tick() {      // bytes per tick of the 100 MHz counter; 0 = no link model
    static const double v = [] { const char* e = getenv("ABSENT_PEERS_LINK_GBS"); return e ? atof(e) * 1e9 / 1e8 : 0.0; }();
    return v;
}
// hold the stream for `ticks` of the 100 MHz counter (an exit condition every lane reaches)
__global__ void link_time_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(512) void link_time_kernel_wide(long long ticks) {
    __shared__ double pad[4096];      // 32 KB: what a channel's staging takes
    pad[threadIdx.x] = 0.0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (pad[threadIdx.x] != 0.0) __builtin_trap();
}
int close_group() {
    uint64_t worst = 0;
    for (auto& kv : g_out) if (kv.second > worst) worst = kv.second;
    for (auto& kv : g_in) if (kv.second > worst) worst = kv.second;
    g_out.clear(); g_in.clear();
    if (worst == 0 || link_bytes_per_tick() <= 0) return 0;
    const long long ticks = (long long)(worst / link_bytes_per_tick());
    // ABSENT_PEERS_HOST_DELAY: the stream is held by a host function that sleeps instead of a wave that spins -- the link time without
    // ANY use of the GPU (a transfer engine); the default, one spinning wave, takes a wave slot and a few registers on one CU the way a
    // (much larger) RCCL kernel does, which a product kernel that fills every CU to its register limit notices
    static const bool host_delay = getenv("ABSENT_PEERS_HOST_DELAY") != nullptr;
    if (host_delay)
        return hipLaunchHostFunc(g_stream, [](void* t) { usleep((useconds_t)((long long)(intptr_t)t / 100)); }, (void*)(intptr_t)ticks) == hipSuccess ? 0 : 1;
    // ABSENT_PEERS_SPIN_WGS = n: the link time as n workgroups of 512 threads with 32 KB of LDS each -- the footprint of a communication
    // kernel with n channels -- instead of one wave: does a product kernel that fills the chip lose more than the slots they take?
    static const int wgs = [] { const char* e = getenv("ABSENT_PEERS_SPIN_WGS"); return e ? atoi(e) : 0; }();
    if (wgs > 0) hipLaunchKernelGGL(link_time_kernel_wide, dim3(wgs), dim3(512), 0, g_stream, ticks);
    else hipLaunchKernelGGL(link_time_kernel, dim3(1), dim3(64), 0, g_stream, ticks);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId* id) { memset(id, 0, sizeof(*id)); id->internal[0] = 'a'; return 0; }
int ncclCommInitRank(void** comm, int nranks, UniqueId, int rank) { *comm = new Comm{rank, nranks}; return 0; }
int ncclCommDestroy(void* comm) { delete (Comm*)comm; return 0; }
int ncclCommAbort(void* comm) { delete (Comm*)comm; return 0; }
int ncclGroupStart(void) { ++g_depth; return 0; }
int ncclGroupEnd(void) { return --g_depth == 0 ? close_group() : 0; }
int ncclSend(const void*, size_t count, int dtype, int peer, void*, hipStream_t stream) {
    g_sent += count * type_bytes(dtype);
    g_out[peer] += count * type_bytes(dtype);
    g_stream = stream;
    return g_depth == 0 ? close_group() : 0;
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void*, hipStream_t stream) {
    g_received += count * type_bytes(dtype);
    g_in[peer] += count * type_bytes(dtype);
    g_stream = stream;
    static const bool fill = getenv("ABSENT_PEERS_NO_FILL") == nullptr;      // (timing experiments: nothing delivered, results wrong)
    if (fill && hipMemsetAsync(buf, 0, count * type_bytes(dtype), stream) != hipSuccess) return 1;
    return g_depth == 0 ? close_group() : 0;
}
const char* ncclGetErrorString(int code) { return code ? "absent-peers stand-in: hipMemsetAsync failed" : "no error"; }
// what the rank would have put on / taken off its links (bytes since load)
uint64_t absent_peers_sent_bytes(void) { return g_sent; }
uint64_t absent_peers_received_bytes(void) { return g_received; }
}  // extern "C"
