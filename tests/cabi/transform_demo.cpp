// Stand-alone host program for the C ABI of libqs_amd.so: no Python, no torch.
// Built and run by tests/test_gpu_cabi_native.py (hipcc, links the in-tree library).
//
// It does what a C/C++ host of the reference's hot path would do: allocate device buffers,
// ask for the workspace size, call qs_transform_two_body / qs_antisymmetrize /
// qs_spin_expand_two_body on a stream, and check the results against plain loops
// (out[pqrs] = sum_abcd Ct[pa] Ct[qb] u[abcd] C[cr] C[ds], basis_set.py:336-350).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "qs_amd.h"

#define HIP_OK(x)                                                                 \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
    } while (0)
#define QS_CALL(x)                                                                \
    do {                                                                          \
        int rc_ = (x);                                                            \
        if (rc_ != QS_OK) { std::printf("%s -> %d (%s; %s)\n", #x, rc_, qs_error_string(rc_), qs_last_hip_error()); return 3; } \
    } while (0)

static double uniform(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (s >> 8) * (1.0 / 16777216.0);
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? std::atoi(argv[1]) : 14, M = argc > 2 ? std::atoi(argv[2]) : 11;
    if (qs_abi_version() != QS_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }
    unsigned seed = 12345u;
    std::vector<double> u((size_t)L * L * L * L), C((size_t)L * M), Ct((size_t)M * L);
    for (auto& x : u) x = uniform(seed) - 0.5;
    for (auto& x : C) x = uniform(seed) - 0.5;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < M; ++j) Ct[(size_t)j * L + i] = C[(size_t)i * M + j];   // real case: Ct = C^T

    // reference by plain loops, one index at a time in the order d, c, b, a
    auto idx4 = [](int n1, int n2, int n3, int a, int b, int c, int d) { return (((size_t)a * n1 + b) * n2 + c) * n3 + d; };
    std::vector<double> t1((size_t)L * L * L * M), t2((size_t)L * L * M * M), t3((size_t)L * M * M * M),
        ref((size_t)M * M * M * M);
    for (int a = 0; a < L; ++a) for (int b = 0; b < L; ++b) for (int c = 0; c < L; ++c) for (int s = 0; s < M; ++s) {
        double acc = 0; for (int d = 0; d < L; ++d) acc += u[idx4(L, L, L, a, b, c, d)] * C[(size_t)d * M + s];
        t1[idx4(L, L, M, a, b, c, s)] = acc; }
    for (int a = 0; a < L; ++a) for (int b = 0; b < L; ++b) for (int r = 0; r < M; ++r) for (int s = 0; s < M; ++s) {
        double acc = 0; for (int c = 0; c < L; ++c) acc += C[(size_t)c * M + r] * t1[idx4(L, L, M, a, b, c, s)];
        t2[idx4(L, M, M, a, b, r, s)] = acc; }
    for (int a = 0; a < L; ++a) for (int q = 0; q < M; ++q) for (int r = 0; r < M; ++r) for (int s = 0; s < M; ++s) {
        double acc = 0; for (int b = 0; b < L; ++b) acc += Ct[(size_t)q * L + b] * t2[idx4(L, M, M, a, b, r, s)];
        t3[idx4(M, M, M, a, q, r, s)] = acc; }
    for (int p = 0; p < M; ++p) for (int q = 0; q < M; ++q) for (int r = 0; r < M; ++r) for (int s = 0; s < M; ++s) {
        double acc = 0; for (int a = 0; a < L; ++a) acc += Ct[(size_t)p * L + a] * t3[idx4(M, M, M, a, q, r, s)];
        ref[idx4(M, M, M, p, q, r, s)] = acc; }

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    double *d_u, *d_C, *d_Ct, *d_out, *d_as, *d_spin;
    void* d_work;
    const int64_t work_bytes = qs_transform_two_body_workspace(QS_F64, L, M);
    if (work_bytes < 0) { std::printf("workspace query failed: %lld\n", (long long)work_bytes); return 1; }
    const size_t n_out = (size_t)M * M * M * M, n_spin = 16 * n_out;
    HIP_OK(hipMalloc(&d_u, u.size() * 8)); HIP_OK(hipMalloc(&d_C, C.size() * 8)); HIP_OK(hipMalloc(&d_Ct, Ct.size() * 8));
    HIP_OK(hipMalloc(&d_out, n_out * 8)); HIP_OK(hipMalloc(&d_as, n_out * 8)); HIP_OK(hipMalloc(&d_spin, n_spin * 16));
    HIP_OK(hipMalloc(&d_work, (size_t)work_bytes));
    HIP_OK(hipMemcpyAsync(d_u, u.data(), u.size() * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_C, C.data(), C.size() * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_Ct, Ct.data(), Ct.size() * 8, hipMemcpyHostToDevice, stream));

    QS_CALL(qs_transform_two_body(QS_F64, d_u, d_C, d_Ct, d_out, d_work, work_bytes, L, M, stream));
    QS_CALL(qs_antisymmetrize(QS_F64, d_out, d_as, (int64_t)M * M, M, stream));
    // spin doubling fused with anti-symmetrisation and the complex cast: (M)^4 fp64 -> (2M)^4 complex128
    QS_CALL(qs_spin_expand_two_body(QS_F64, QS_C128, d_out, d_spin, M, 0, M, 1, stream));

    std::vector<double> out(n_out), as(n_out), spin(2 * n_spin);
    HIP_OK(hipMemcpyAsync(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(as.data(), d_as, n_out * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(spin.data(), d_spin, n_spin * 16, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    double worst = 0, scale = 0;
    for (size_t i = 0; i < n_out; ++i) { worst = std::fmax(worst, std::fabs(out[i] - ref[i])); scale = std::fmax(scale, std::fabs(ref[i])); }
    size_t bad_as = 0, bad_spin = 0;
    for (int p = 0; p < M; ++p) for (int q = 0; q < M; ++q) for (int r = 0; r < M; ++r) for (int s = 0; s < M; ++s)
        if (as[idx4(M, M, M, p, q, r, s)] != out[idx4(M, M, M, p, q, r, s)] - out[idx4(M, M, M, p, q, s, r)]) ++bad_as;
    const int N = 2 * M;
    for (int P = 0; P < N; ++P) for (int Q = 0; Q < N; ++Q) for (int R = 0; R < N; ++R) for (int S = 0; S < N; ++S) {
        const int p = P / 2, q = Q / 2, r = R / 2, s = S / 2;
        const double direct = (P % 2 == R % 2 && Q % 2 == S % 2) ? out[idx4(M, M, M, p, q, r, s)] : 0.0;
        const double exch = (P % 2 == S % 2 && Q % 2 == R % 2) ? out[idx4(M, M, M, p, q, s, r)] : 0.0;
        const size_t i = idx4(N, N, N, P, Q, R, S);
        if (spin[2 * i] != direct - exch || spin[2 * i + 1] != 0.0) ++bad_spin;
    }
    // ---- a REAL tensor against COMPLEX coefficients, the tensor read as it is (qs_transform_two_body_mixed): with
    // C_c = C e^{i phi} and an explicit C~_c = C^T e^{i theta} the result is the real one times e^{2 i (phi + theta)}
    const double phi = 0.3, theta = -0.7;
    std::vector<double> Cc(2 * C.size()), Ctc(2 * Ct.size());
    for (size_t i = 0; i < C.size(); ++i) { Cc[2 * i] = C[i] * std::cos(phi); Cc[2 * i + 1] = C[i] * std::sin(phi); }
    for (size_t i = 0; i < Ct.size(); ++i) { Ctc[2 * i] = Ct[i] * std::cos(theta); Ctc[2 * i + 1] = Ct[i] * std::sin(theta); }
    double *d_Cc, *d_Ctc, *d_outc;
    void* d_workc;
    const int64_t work_c = qs_transform_two_body_workspace(QS_C128, L, M);
    HIP_OK(hipMalloc(&d_Cc, Cc.size() * 8)); HIP_OK(hipMalloc(&d_Ctc, Ctc.size() * 8));
    HIP_OK(hipMalloc(&d_outc, n_out * 16)); HIP_OK(hipMalloc(&d_workc, (size_t)work_c));
    HIP_OK(hipMemcpyAsync(d_Cc, Cc.data(), Cc.size() * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_Ctc, Ctc.data(), Ctc.size() * 8, hipMemcpyHostToDevice, stream));
    QS_CALL(qs_transform_two_body_mixed(d_u, d_Cc, d_Ctc, d_outc, d_workc, work_c, L, M, stream));
    std::vector<double> outc(2 * n_out);
    HIP_OK(hipMemcpyAsync(outc.data(), d_outc, n_out * 16, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    double worst_c = 0;
    const double pr = std::cos(2 * (phi + theta)), pi = std::sin(2 * (phi + theta));
    for (size_t i = 0; i < n_out; ++i)
        worst_c = std::fmax(worst_c, std::fmax(std::fabs(outc[2 * i] - ref[i] * pr), std::fabs(outc[2 * i + 1] - ref[i] * pi)));

    // ---- the memory-lean sharded transform through a ONE-rank RCCL communicator (rows in, rows out; on a node every rank
    // makes the same call): result rows [q][p][r][s] at the start of the result buffer
    int rows_state = 0;                  // 0 skipped (librccl not loadable here), 1 ok, -1 failed
    double worst_rows = 0;
    unsigned char uid[QS_UNIQUE_ID_BYTES];
    void* comm = nullptr;
    if (qs_comm_unique_id(uid) == QS_OK && qs_comm_init(&comm, 0, 1, uid) == QS_OK) {
        const int64_t ni = 3, ob = qs_transform_two_body_sharded_rows_out_bytes(QS_F64, L, M, 1, 0),
                      wb = qs_transform_two_body_sharded_rows_workspace(QS_F64, L, M, ni);
        void *d_buf, *d_w2;
        HIP_OK(hipMalloc(&d_buf, (size_t)ob)); HIP_OK(hipMalloc(&d_w2, (size_t)wb));
        QS_CALL(qs_transform_two_body_sharded_rows(comm, QS_F64, QS_F64, d_u, nullptr, d_C, d_Ct, d_buf, ob, d_w2, wb, L, M, ni, stream));
        std::vector<double> rows(n_out);
        HIP_OK(hipMemcpyAsync(rows.data(), d_buf, n_out * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int p = 0; p < M; ++p) for (int q = 0; q < M; ++q) for (int r = 0; r < M; ++r) for (int s = 0; s < M; ++s)
            worst_rows = std::fmax(worst_rows, std::fabs(rows[idx4(M, M, M, q, p, r, s)] - ref[idx4(M, M, M, p, q, r, s)]));
        rows_state = worst_rows <= 1e-12 * scale ? 1 : -1;
        (void)hipFree(d_buf); (void)hipFree(d_w2);
        if (qs_comm_destroy(comm) != QS_OK) rows_state = -1;
    } else {
        std::printf("RCCL not loadable here (%s): sharded rows call skipped\n", qs_last_comm_error());
    }

    // argument checking: the library reports, it never throws or aborts
    const int rc_small = qs_transform_two_body(QS_F64, d_u, d_C, d_Ct, d_out, d_work, 16, L, M, stream);
    const int rc_null = qs_transform_two_body(QS_F64, nullptr, d_C, d_Ct, d_out, d_work, work_bytes, L, M, stream);
    const int rc_alias = qs_transform_two_body(QS_F64, d_u, d_C, d_Ct, d_u, d_work, work_bytes, L, M, stream);

    std::printf("L=%d M=%d rel_err=%.3e mixed_rel_err=%.3e sharded_rows=%d (%.3e) antisym_mismatches=%zu spin_mismatches=%zu rc_small=%d rc_null=%d rc_alias=%d\n",
                L, M, worst / scale, worst_c / scale, rows_state, worst_rows / scale, bad_as, bad_spin, rc_small, rc_null, rc_alias);
    const bool ok = worst <= 1e-12 * scale && worst_c <= 1e-12 * scale && rows_state >= 0 && bad_as == 0 && bad_spin == 0 &&
                    rc_small == QS_ERR_WORKSPACE && rc_null == QS_ERR_NULL_POINTER && rc_alias == QS_ERR_ALIAS;
    for (void* ptr : {(void*)d_Cc, (void*)d_Ctc, (void*)d_outc, d_workc}) (void)hipFree(ptr);
    for (void* ptr : {(void*)d_u, (void*)d_C, (void*)d_Ct, (void*)d_out, (void*)d_as, (void*)d_spin, d_work}) (void)hipFree(ptr);
    (void)hipStreamDestroy(stream);
    std::printf(ok ? "CABI_DEMO_OK\n" : "CABI_DEMO_FAILED\n");
    return ok ? 0 : 1;
}
