// The ranks of a multi-rank run of the library's sharded entry points as THREADS of one process, all on one device
// (no Python, no torch) -- the host side of tests/cabi/mock_rccl_async.cpp, the stream-ordered asynchronous stand-in
// for librccl:
//     sharded_threads_demo <world> <L> <M> <complex 0|1|2> <chunk_rows> <coalesce 0|1> <drop mask> <stall us> <entry 0|1>
// (complex 2 = real tensor against complex coefficients; entry 0 = qs_transform_two_body_sharded_rows on rows of the
// leading index, 1 = the slab entry qs_transform_two_body_sharded).  Every rank thread keeps its share of the same seeded
// tensor, calls the entry point with its own stream and communicator handle, and compares what it gets with ITS part of
// the single-GPU transform of the whole tensor: bit for bit (entry 0), to 1e-12 (entry 1).
//   drop mask : handed to qs_tuning_set("comm_drop_wait") in every rank thread -- the negative tests: with one of the
//               library's stream waits left out the results must be WRONG over this transport;
//   stall us  : a kernel that holds the rank's stream for that long is launched right before the call, so that the host
//               posts the whole exchange while the products have not even started (what a busy GPU looks like).
// Buffers the exchange writes or reads are filled with NaN patterns before the call: data that is read before it was
// produced, or after it was overwritten, cannot compare equal by accident.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "qs_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define QS_CALL(x) do { int rc_ = (x); if (rc_ != QS_OK) { std::printf("rank %d: %s -> %d (%s; %s; %s)\n", rank, #x, rc_, qs_error_string(rc_), qs_last_hip_error(), qs_last_comm_error()); return 3; } } while (0)

__global__ void stall_kernel(long long ticks) {      // holds a stream; every lane leaves after `ticks` of the 100 MHz counter
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static double uniform(unsigned& s) { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); }
static int64_t lo_of(int64_t n, int G, int r) { return r * (n / G) + (r < n % G ? r : n % G); }

struct Job {
    int G; int64_t L, M; int mode; int64_t chunk; int coalesce, drop; long long stall_us; int entry;
    const std::vector<double>*u, *C, *Ct, *full;
    double scale;
    unsigned char id[QS_UNIQUE_ID_BYTES];
};

static std::mutex g_print;

static int rank_body(const Job& job, int rank) {
    const int G = job.G;
    const int64_t L = job.L, M = job.M;
    const bool cx_u = job.mode == 1, cx = job.mode != 0;
    const int es_u = cx_u ? 2 : 1, es = cx ? 2 : 1;
    const int dt = cx ? QS_C128 : QS_F64, dt_u = cx_u ? QS_C128 : QS_F64;
    HIP_OK(hipSetDevice(0));
    void* comm = nullptr;
    QS_CALL(qs_comm_init(&comm, rank, G, job.id));
    QS_CALL(qs_comm_set_option(comm, "rows_coalesce", job.coalesce));
    QS_CALL(qs_tuning_set("comm_drop_wait", job.drop));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    double *d_C, *d_Ct;
    HIP_OK(hipMalloc(&d_C, job.C->size() * 8)); HIP_OK(hipMalloc(&d_Ct, job.Ct->size() * 8));
    HIP_OK(hipMemcpy(d_C, job.C->data(), job.C->size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_Ct, job.Ct->data(), job.Ct->size() * 8, hipMemcpyHostToDevice));
    const int64_t j_lo = lo_of(M, G, rank), jl = lo_of(M, G, rank + 1) - j_lo;
    const std::vector<double>& u = *job.u;
    const std::vector<double>& full = *job.full;
    std::vector<double> got((size_t)(jl > 0 ? jl : 1) * M * M * M * es);
    size_t bad_blocks = 0;
    double worst = 0;
    if (job.entry == 0) {
        const int64_t i_lo = lo_of(L, G, rank), il = lo_of(L, G, rank + 1) - i_lo;
        const int64_t ni = job.chunk > 0 ? job.chunk : qs_sharded_rows_default_chunk(dt, L, M, G, nullptr);
        const int64_t ob = qs_transform_two_body_sharded_rows_out_bytes(dt, L, M, G, rank);
        const int64_t wb = qs_comm_rows_workspace(comm, dt, L, M, ni < L ? ni : L);
        if (ob < 0 || wb < 0) { std::printf("rank %d: size queries %lld %lld\n", rank, (long long)ob, (long long)wb); return 4; }
        double* d_rows;
        void *d_buf, *d_w;
        HIP_OK(hipMalloc(&d_rows, (size_t)(il > 0 ? il : 1) * L * L * L * es_u * 8));
        HIP_OK(hipMalloc(&d_buf, (size_t)ob)); HIP_OK(hipMalloc(&d_w, (size_t)wb));
        HIP_OK(hipMemcpy(d_rows, u.data() + (size_t)i_lo * L * L * L * es_u, (size_t)il * L * L * L * es_u * 8, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {      // twice: the second call meets the events and buffers the first one left behind
            HIP_OK(hipMemsetAsync(d_buf, 0xFF, (size_t)ob, stream));
            HIP_OK(hipMemsetAsync(d_w, 0xFF, (size_t)wb, stream));
            if (job.stall_us > 0) hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, stream, job.stall_us * 100);
            QS_CALL(qs_transform_two_body_sharded_rows(comm, dt_u, dt, d_rows, nullptr, d_C, d_Ct, d_buf, ob, d_w, wb, L, M, ni, stream));
            HIP_OK(hipMemcpyAsync(got.data(), d_buf, (size_t)jl * M * M * M * es * 8, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            // out_rows[q_loc][p][r][s] == full[p][q_lo + q_loc][r][s], bit for bit (contractions in the order d, c, b, a)
            for (int64_t q = 0; q < jl; ++q) for (int64_t p = 0; p < M; ++p)
                if (std::memcmp(&got[(size_t)((q * M + p) * M * M) * es], &full[(size_t)(((p * M) + j_lo + q) * M * M) * es],
                                (size_t)M * M * es * 8) != 0) ++bad_blocks;
        }
        (void)hipFree(d_rows); (void)hipFree(d_buf); (void)hipFree(d_w);
    } else {
        if (job.mode == 2) { std::printf("rank %d: the slab entry takes one dtype\n", rank); return 4; }
        const int64_t b_lo = lo_of(L, G, rank), bl = lo_of(L, G, rank + 1) - b_lo;
        std::vector<double> slab((size_t)L * (bl > 0 ? bl : 1) * L * L * es);
        for (int64_t a = 0; a < L; ++a) for (int64_t b = 0; b < bl; ++b)
            std::memcpy(&slab[(size_t)((a * bl + b) * L * L) * es], &u[(size_t)((a * L + b_lo + b) * L * L) * es], (size_t)L * L * es * 8);
        double *d_slab, *d_out;
        void* d_w;
        const int64_t wb = qs_transform_two_body_sharded_workspace(dt, L, M, G, rank);
        const size_t out_bytes = (size_t)(jl > 0 ? jl : 1) * M * M * M * es * 8;
        HIP_OK(hipMalloc(&d_slab, slab.size() * 8)); HIP_OK(hipMalloc(&d_out, out_bytes)); HIP_OK(hipMalloc(&d_w, (size_t)wb));
        HIP_OK(hipMemcpy(d_slab, slab.data(), slab.size() * 8, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipMemsetAsync(d_out, 0xFF, out_bytes, stream));
            HIP_OK(hipMemsetAsync(d_w, 0xFF, (size_t)wb, stream));
            if (job.stall_us > 0) hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, stream, job.stall_us * 100);
            QS_CALL(qs_transform_two_body_sharded(comm, dt, d_slab, d_C, d_Ct, d_out, d_w, wb, L, M, (int)job.chunk, stream));
            HIP_OK(hipMemcpyAsync(got.data(), d_out, (size_t)jl * M * M * M * es * 8, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            for (size_t i = 0; i < (size_t)jl * M * M * M * es; ++i) {
                const double d = std::fabs(got[i] - full[(size_t)j_lo * M * M * M * es + i]);
                worst = std::isnan(d) ? INFINITY : std::fmax(worst, d);
            }
        }
        (void)hipFree(d_slab); (void)hipFree(d_out); (void)hipFree(d_w);
    }
    QS_CALL(qs_tuning_reset());
    QS_CALL(qs_comm_destroy(comm));
    (void)hipFree(d_C); (void)hipFree(d_Ct); (void)hipStreamDestroy(stream);
    const bool ok = bad_blocks == 0 && worst <= 1e-12 * job.scale;
    std::lock_guard<std::mutex> g(g_print);
    std::printf("rank %d/%d L=%lld M=%lld mode=%d chunk=%lld coalesce=%d drop=%d stall=%lld entry=%d: differing blocks %zu, worst %.2e -> %s\n",
                rank, G, (long long)L, (long long)M, job.mode, (long long)job.chunk, job.coalesce, job.drop, job.stall_us, job.entry,
                bad_blocks, worst / job.scale, ok ? "RANK_OK" : "RANK_WRONG");
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc < 10) { std::printf("usage: world L M complex chunk_rows coalesce drop_mask stall_us entry\n"); return 64; }
    const int rank = -1;
    Job job;
    job.G = std::atoi(argv[1]); job.L = std::atoi(argv[2]); job.M = std::atoi(argv[3]); job.mode = std::atoi(argv[4]);
    job.chunk = std::atoi(argv[5]); job.coalesce = std::atoi(argv[6]); job.drop = std::atoi(argv[7]);
    job.stall_us = std::atoll(argv[8]); job.entry = std::atoi(argv[9]);
    const int64_t L = job.L, M = job.M;
    const bool cx_u = job.mode == 1, cx = job.mode != 0;
    const int es_u = cx_u ? 2 : 1, es = cx ? 2 : 1;
    const int dt = cx ? QS_C128 : QS_F64;
    unsigned seed = 777u;
    const size_t nu = (size_t)L * L * L * L, nout = (size_t)M * M * M * M;
    std::vector<double> u(nu * es_u), C((size_t)L * M * es), Ct((size_t)M * L * es), full(nout * es);
    for (auto& x : u) x = uniform(seed) - 0.5;
    for (auto& x : C) x = (uniform(seed) - 0.5) / std::sqrt((double)L);
    for (auto& x : Ct) x = (uniform(seed) - 0.5) / std::sqrt((double)L);
    {      // the single-GPU transform of the whole tensor, once
        HIP_OK(hipSetDevice(0));
        double *d_u, *d_C, *d_Ct, *d_full;
        void* d_work;
        const int64_t wfull = qs_transform_two_body_workspace(dt, L, M);
        HIP_OK(hipMalloc(&d_u, u.size() * 8)); HIP_OK(hipMalloc(&d_C, C.size() * 8)); HIP_OK(hipMalloc(&d_Ct, Ct.size() * 8));
        HIP_OK(hipMalloc(&d_full, full.size() * 8)); HIP_OK(hipMalloc(&d_work, (size_t)wfull));
        HIP_OK(hipMemcpy(d_u, u.data(), u.size() * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_C, C.data(), C.size() * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_Ct, Ct.data(), Ct.size() * 8, hipMemcpyHostToDevice));
        if (job.mode == 2) QS_CALL(qs_transform_two_body_mixed(d_u, d_C, d_Ct, d_full, d_work, wfull, L, M, nullptr));
        else QS_CALL(qs_transform_two_body(dt, d_u, d_C, d_Ct, d_full, d_work, wfull, L, M, nullptr));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(full.data(), d_full, full.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_u); (void)hipFree(d_C); (void)hipFree(d_Ct); (void)hipFree(d_full); (void)hipFree(d_work);
    }
    job.scale = 0;
    for (double v : full) job.scale = std::fmax(job.scale, std::fabs(v));
    job.u = &u; job.C = &C; job.Ct = &Ct; job.full = &full;
    QS_CALL(qs_comm_unique_id(job.id));
    std::vector<int> rcs(job.G, -1);
    std::vector<std::thread> threads;
    for (int r = 0; r < job.G; ++r) threads.emplace_back([&, r] { rcs[r] = rank_body(job, r); });
    for (auto& t : threads) t.join();
    int worst_rc = 0;
    for (int rc : rcs) if (rc > worst_rc) worst_rc = rc;
    std::printf("%s\n", worst_rc == 0 ? "ALL_RANKS_OK" : worst_rc == 1 ? "SOME_RANK_WRONG" : "ERROR");
    return worst_rc;
}
