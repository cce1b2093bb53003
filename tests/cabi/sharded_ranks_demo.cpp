// One RANK of a multi-process run of the library's sharded entry points (no Python, no torch):
//     sharded_ranks_demo <rank> <world> <id file> <L> <M> <complex 0|1|2> <chunk_rows>
// (complex 2 = real tensor against complex coefficients).  Every rank builds the same seeded tensor, keeps its share,
// calls qs_transform_two_body_sharded_rows (balanced rows of the leading index, then the doubled partition of spin
// doubling when L is even) and qs_transform_two_body_sharded (second-index slab), and compares what it gets with ITS
// part of the single-GPU transform of the whole tensor (qs_transform_two_body) computed on the same device: bit for bit
// where the order of the contractions is the same (rows of the leading index), to 1e-12 otherwise.
// With tests/cabi/mock_rccl.cpp on the loader path this runs several ranks on ONE GPU (tests/test_gpu_mock_rccl_ranks.py);
// on a node with real RCCL the same program runs one rank per GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <string>
#include <vector>

#include "qs_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define QS_CALL(x) do { int rc_ = (x); if (rc_ != QS_OK) { std::printf("rank %d: %s -> %d (%s; %s; %s)\n", g_rank, #x, rc_, qs_error_string(rc_), qs_last_hip_error(), qs_last_comm_error()); return 3; } } while (0)

static int g_rank = 0;

static double uniform(unsigned& s) { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); }
static int64_t lo_of(int64_t n, int G, int r) { return r * (n / G) + (r < n % G ? r : n % G); }

int main(int argc, char** argv) {
    if (argc < 8) { std::printf("usage: rank world idfile L M complex chunk_rows\n"); return 1; }
    const int rank = g_rank = std::atoi(argv[1]), G = std::atoi(argv[2]);
    const char* idfile = argv[3];
    const int64_t L = std::atoi(argv[4]), M = std::atoi(argv[5]);
    const int mode = std::atoi(argv[6]);
    const int64_t chunk = std::atoi(argv[7]);
    const bool cx_u = mode == 1, cx = mode != 0;
    const int es_u = cx_u ? 2 : 1, es = cx ? 2 : 1;                         // doubles per element
    const int dt = cx ? QS_C128 : QS_F64, dt_u = cx_u ? QS_C128 : QS_F64;

    // ---- the communicator: rank 0 draws the id, the others read it from the file
    unsigned char id[QS_UNIQUE_ID_BYTES];
    if (rank == 0) {
        QS_CALL(qs_comm_unique_id(id));
        std::string tmp = std::string(idfile) + ".part";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 1;
        std::fclose(f);
        std::rename(tmp.c_str(), idfile);
    } else {
        FILE* f = nullptr;
        for (int i = 0; i < 30000 && !(f = std::fopen(idfile, "rb")); ++i) usleep(2000);
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) { std::printf("rank %d: no id\n", rank); return 1; }
        std::fclose(f);
    }
    void* comm = nullptr;
    QS_CALL(qs_comm_init(&comm, rank, G, id));

    // ---- the same tensor and coefficients on every rank
    unsigned seed = 4242u;
    const size_t nu = (size_t)L * L * L * L, nout = (size_t)M * M * M * M;
    std::vector<double> u(nu * es_u), C((size_t)L * M * es), Ct((size_t)M * L * es);
    for (auto& x : u) x = uniform(seed) - 0.5;
    for (auto& x : C) x = (uniform(seed) - 0.5) / std::sqrt((double)L);
    for (auto& x : Ct) x = (uniform(seed) - 0.5) / std::sqrt((double)L);
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    double *d_u, *d_C, *d_Ct, *d_full;
    void* d_work;
    HIP_OK(hipMalloc(&d_u, u.size() * 8)); HIP_OK(hipMalloc(&d_C, C.size() * 8)); HIP_OK(hipMalloc(&d_Ct, Ct.size() * 8));
    HIP_OK(hipMalloc(&d_full, nout * es * 8));
    HIP_OK(hipMemcpy(d_u, u.data(), u.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_C, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_Ct, Ct.data(), Ct.size() * 8, hipMemcpyHostToDevice));
    const int64_t wfull = qs_transform_two_body_workspace(dt, L, M);
    HIP_OK(hipMalloc(&d_work, (size_t)wfull));
    if (mode == 2) QS_CALL(qs_transform_two_body_mixed(d_u, d_C, d_Ct, d_full, d_work, wfull, L, M, stream));
    else QS_CALL(qs_transform_two_body(dt, d_u, d_C, d_Ct, d_full, d_work, wfull, L, M, stream));
    std::vector<double> full(nout * es);
    HIP_OK(hipMemcpyAsync(full.data(), d_full, full.size() * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    double scale = 0;
    for (double v : full) scale = std::fmax(scale, std::fabs(v));

    const int64_t j_lo = lo_of(M, G, rank), jl = lo_of(M, G, rank + 1) - j_lo;
    const int64_t ni = chunk > 0 ? chunk : qs_sharded_rows_default_chunk(dt, L, M, G, nullptr);
    const int64_t ob = qs_transform_two_body_sharded_rows_out_bytes(dt, L, M, G, rank);
    const int64_t wb = qs_transform_two_body_sharded_rows_workspace(dt, L, M, ni < L ? ni : L);
    void *d_buf, *d_w2;
    HIP_OK(hipMalloc(&d_buf, (size_t)ob)); HIP_OK(hipMalloc(&d_w2, (size_t)wb));
    std::vector<double> got((size_t)(jl > 0 ? jl : 1) * M * M * M * es);
    size_t bad_bits = 0;
    // ---- rows of the LEADING index in (balanced, then the doubled partition), rows of the second transformed index out:
    // out_rows[q_loc][p][r][s] == full[p][q_lo + q_loc][r][s], bit for bit (contractions in the order d, c, b, a)
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int64_t> starts(G + 1);
        if (pass == 1) {
            if (L % 2) break;
            for (int g = 0; g <= G; ++g) starts[g] = 2 * lo_of(L / 2, G, g);
        }
        const int64_t i_lo = pass ? starts[rank] : lo_of(L, G, rank), i_hi = pass ? starts[rank + 1] : lo_of(L, G, rank + 1);
        const double* rows = d_u + (size_t)i_lo * L * L * L * es_u;
        (void)i_hi;
        HIP_OK(hipMemsetAsync(d_buf, 0xFF, (size_t)ob, stream));                   // (NaNs: nothing may be read before it arrived)
        QS_CALL(qs_transform_two_body_sharded_rows(comm, dt_u, dt, rows, pass ? starts.data() : nullptr, d_C, d_Ct, d_buf, ob, d_w2, wb,
                                                   L, M, ni, stream));
        HIP_OK(hipMemcpyAsync(got.data(), d_buf, (size_t)jl * M * M * M * es * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int64_t q = 0; q < jl; ++q) for (int64_t p = 0; p < M; ++p)
            if (std::memcmp(&got[(size_t)((q * M + p) * M * M) * es], &full[(size_t)(((p * M) + j_lo + q) * M * M) * es],
                            (size_t)M * M * es * 8) != 0) ++bad_bits;
    }
    // ---- rows of the SECOND index in (rows[i][j] = u[j, i_lo + i]): out_rows[p_loc][q] == full[p_lo + p_loc][q] to rounding
    // (contractions in the order d, c, a, b)
    double worst2 = 0;
    {
        const int64_t i_lo = lo_of(L, G, rank), il = lo_of(L, G, rank + 1) - i_lo;
        std::vector<double> tr((size_t)(il > 0 ? il : 1) * L * L * L * es_u);
        for (int64_t i = 0; i < il; ++i) for (int64_t j = 0; j < L; ++j)
            std::memcpy(&tr[(size_t)((i * L + j) * L * L) * es_u], &u[(size_t)((j * L + i_lo + i) * L * L) * es_u], (size_t)L * L * es_u * 8);
        double* d_tr;
        HIP_OK(hipMalloc(&d_tr, tr.size() * 8));
        HIP_OK(hipMemcpy(d_tr, tr.data(), tr.size() * 8, hipMemcpyHostToDevice));
        QS_CALL(qs_transform_two_body_sharded_rows(comm, dt_u, dt, d_tr, nullptr, d_C, d_Ct, d_buf, ob, d_w2, wb, L, M, ni, stream));
        HIP_OK(hipMemcpyAsync(got.data(), d_buf, (size_t)jl * M * M * M * es * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < (size_t)jl * M * M * M * es; ++i)
            worst2 = std::fmax(worst2, std::fabs(got[i] - full[(size_t)j_lo * M * M * M * es + i]));
        (void)hipFree(d_tr);
    }
    // ---- the round-2 entry (second-index slab in its (L, bl, L, L) form, result rows of the leading index), same dtype only
    double worst3 = 0;
    if (mode != 2) {
        const int64_t b_lo = lo_of(L, G, rank), bl = lo_of(L, G, rank + 1) - b_lo;
        std::vector<double> slab((size_t)L * (bl > 0 ? bl : 1) * L * L * es);
        for (int64_t a = 0; a < L; ++a) for (int64_t b = 0; b < bl; ++b)
            std::memcpy(&slab[(size_t)((a * bl + b) * L * L) * es], &u[(size_t)((a * L + b_lo + b) * L * L) * es], (size_t)L * L * es * 8);
        double *d_slab, *d_out;
        void* d_w3;
        const int64_t w3 = qs_transform_two_body_sharded_workspace(dt, L, M, G, rank);
        HIP_OK(hipMalloc(&d_slab, slab.size() * 8)); HIP_OK(hipMalloc(&d_out, (size_t)(jl > 0 ? jl : 1) * M * M * M * es * 8));
        HIP_OK(hipMalloc(&d_w3, (size_t)w3));
        HIP_OK(hipMemcpy(d_slab, slab.data(), slab.size() * 8, hipMemcpyHostToDevice));
        QS_CALL(qs_transform_two_body_sharded(comm, dt, d_slab, d_C, d_Ct, d_out, d_w3, w3, L, M, 3, stream));
        HIP_OK(hipMemcpyAsync(got.data(), d_out, (size_t)jl * M * M * M * es * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < (size_t)jl * M * M * M * es; ++i)
            worst3 = std::fmax(worst3, std::fabs(got[i] - full[(size_t)j_lo * M * M * M * es + i]));
        (void)hipFree(d_slab); (void)hipFree(d_out); (void)hipFree(d_w3);
    }
    QS_CALL(qs_comm_destroy(comm));
    const bool ok = bad_bits == 0 && worst2 <= 1e-12 * scale && worst3 <= 1e-12 * scale;
    std::printf("rank %d/%d L=%lld M=%lld mode=%d chunk=%lld: rows(leading) differing blocks %zu, rows(second) %.2e, slab entry %.2e -> %s\n",
                rank, G, (long long)L, (long long)M, mode, (long long)ni, bad_bits, worst2 / scale, worst3 / scale, ok ? "RANK_OK" : "RANK_FAILED");
    return ok ? 0 : 1;
}
