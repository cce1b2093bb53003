// Host-side sanitizer run of the C ABI (AddressSanitizer + UBSan, CPU build: no GPU involved).  Every entry point is
// driven through its argument checks -- null pointers, misaligned pointers, bad extents and dtypes, aliases, short
// workspaces -- which all return before the first HIP call, plus the pure host logic: workspace queries, error
// strings, tuning state, the dispatch record and the exchange plan of the sharded transform for several worlds.
// Built and run by tests/test_cabi_sanitizers.py (hipcc --offload-host-only -fsanitize=address,undefined).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "qs_amd.h"

static int fails = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); ++fails; } \
    } while (0)

int main() {
    EXPECT(qs_abi_version() == QS_ABI_VERSION);
    for (int code = -9; code <= 0; ++code) EXPECT(std::strlen(qs_error_string(code)) > 0);
    EXPECT(std::strlen(qs_last_dispatch()) == 0);

    alignas(16) static double buf[64];
    void* p = buf;
    void* odd = (char*)buf + 4;
    // qs_matmul
    EXPECT(qs_matmul(7, p, p, p, 4, 4, 4, 4, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_BAD_DTYPE);
    EXPECT(qs_matmul(QS_F64, nullptr, p, p, 4, 4, 4, 4, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_matmul(QS_F64, odd, p, p, 4, 4, 4, 4, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_MISALIGNED);
    EXPECT(qs_matmul(QS_C128, (char*)buf + 8, p, p, 4, 4, 4, 4, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_MISALIGNED);
    EXPECT(qs_matmul(QS_F64, p, p, p, 4, 4, 4, 4, 4, 4, 1, -1, 0, 0, 0, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_matmul(QS_F64, p, p, p, 0, 4, 4, 4, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_matmul(QS_F64, p, p, p, 4, 4, 4, 2, 4, 4, 1, 0, 0, 0, 0, nullptr) == QS_ERR_BAD_EXTENT);   // lda < k
    // workspace queries
    EXPECT(qs_transform_two_body_workspace(QS_F64, 256, 256) == 8 * (256LL * 256 + 256LL * 256 * 256 * 256));
    EXPECT(qs_transform_two_body_workspace(QS_F64, 10, 4) == 8 * (10 * 4 + 10LL * 10 * 10 * 4 + 10 * 10 * 4 * 4));
    EXPECT(qs_transform_two_body_workspace(QS_F64, 0, 4) < 0 && qs_transform_two_body_workspace(5, 4, 4) < 0);
    EXPECT(qs_transform_two_body_workspace(QS_C128, 5000, 4) < 0);
    EXPECT(qs_transform_two_body_inplace_workspace(QS_F64, 8, 8) == 8 * (64 + 8 * 8 * 8 * 8));
    EXPECT(qs_transform_two_body_inplace_workspace(QS_F64, 8, 9) < 0);
    EXPECT(qs_transform_two_body_partial_workspace(QS_F64, 8, 8, 9) < 0);
    EXPECT(qs_transform_two_body_sharded_workspace(QS_F64, 8, 8, 2, 2) < 0);
    EXPECT(qs_transform_two_body_sharded_workspace(QS_F64, 8, 8, 2, 1) > 0);
    // transforms
    EXPECT(qs_transform_two_body(QS_F64, p, p, p, p, p, 1 << 20, 4, 4, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_transform_two_body(QS_F64, p, p, p, buf + 32, buf + 16, 8, 2, 2, nullptr) == QS_ERR_WORKSPACE);
    EXPECT(qs_transform_two_body(QS_F64, p, nullptr, p, buf + 32, buf + 16, 1 << 20, 2, 2, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_transform_two_body(QS_F64, p, p, p, buf + 32, buf + 16, 1 << 20, 0, 2, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_transform_two_body_inplace(QS_F64, p, p, p, p, 1 << 20, 4, 4, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_transform_two_body_inplace(QS_F64, p, p, p, buf + 16, 1 << 20, 4, 5, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_transform_two_body_inplace(QS_F64, p, p, p, buf + 16, 8, 2, 2, nullptr) == QS_ERR_WORKSPACE);
    EXPECT(qs_transform_two_body_partial(QS_F64, p, p, p, p, buf + 16, 1 << 20, 4, 4, 2, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_transform_two_body_partial(QS_F64, p, p, p, buf + 32, buf + 16, 1 << 20, 4, 4, 5, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_transform_one_body(QS_F64, p, p, p, p, buf + 16, 1 << 20, 1, 4, 4, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_transform_one_body(QS_F64, p, p, p, buf + 32, buf + 16, 8, 1, 4, 4, nullptr) == QS_ERR_WORKSPACE);
    EXPECT(qs_transform_one_body(QS_F64, p, p, p, buf + 32, odd, 1 << 20, 1, 4, 4, nullptr) == QS_ERR_MISALIGNED);
    // bandwidth kernels
    EXPECT(qs_antisymmetrize(QS_F64, p, p, 0, 4, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_antisymmetrize(9, p, p, 1, 4, nullptr) == QS_ERR_BAD_DTYPE);
    EXPECT(qs_antisymmetrize(QS_F64, nullptr, p, 1, 4, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_spin_expand_two_body(QS_C128, QS_F64, p, buf + 32, 2, 0, 1, 0, nullptr) == QS_ERR_BAD_DTYPE);
    EXPECT(qs_spin_expand_two_body(QS_F64, QS_F64, p, p, 2, 0, 1, 0, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_spin_expand_two_body(QS_F64, QS_F64, p, buf + 32, 2, 1, 1, 0, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_spin_expand_two_body_block(QS_F64, QS_F64, p, buf + 32, 2, 3, 1, 0, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_spin_expand_two_body_block(QS_F64, QS_C128, p, buf + 33, 2, 1, 1, 0, nullptr) == QS_ERR_MISALIGNED);
    EXPECT(qs_add_spin_one_body(QS_F64, QS_F64, p, p, 1, 2, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_add_spin_one_body(QS_F64, QS_F64, p, buf + 32, 0, 2, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_spin_squared_two_body(p, buf + 32, 4, 2, 2, 0, nullptr) == QS_ERR_BAD_EXTENT);
    EXPECT(qs_spin_squared_two_body((char*)buf + 8, buf + 32, 4, 0, 1, 0, nullptr) == QS_ERR_MISALIGNED);
    EXPECT(qs_tdho_coulomb_elements(nullptr, 6, 0, 1, nullptr) < 0);
    EXPECT(qs_tdho_coulomb_elements_nm(p, nullptr, 6, 3, 0, 1, nullptr) < 0);
    // communicator entry points
    EXPECT(qs_comm_unique_id(nullptr) == QS_ERR_NULL_POINTER);
    void* comm = nullptr;
    EXPECT(qs_comm_init(&comm, 2, 2, buf) == QS_ERR_BAD_EXTENT && comm == nullptr);
    EXPECT(qs_comm_init(nullptr, 0, 1, buf) == QS_ERR_NULL_POINTER);
    EXPECT(qs_comm_destroy(nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_comm_rank(nullptr) < 0 && qs_comm_world(nullptr) < 0);
    EXPECT(qs_transform_two_body_sharded(nullptr, QS_F64, p, p, p, buf + 32, buf + 16, 1 << 20, 4, 4, 4, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_comm_abort(nullptr) == QS_ERR_NULL_POINTER);
    // the mixed transform (real u, complex coefficients) and the rows-in / rows-out sharded transform (round 3)
    EXPECT(qs_transform_two_body_mixed(p, p, p, p, buf + 16, 1 << 20, 4, 4, nullptr) == QS_ERR_ALIAS);
    EXPECT(qs_transform_two_body_mixed(nullptr, p, p, buf + 32, buf + 16, 1 << 20, 2, 2, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_transform_two_body_mixed(p, (char*)buf + 8, p, buf + 32, buf + 16, 1 << 20, 2, 2, nullptr) == QS_ERR_MISALIGNED);
    EXPECT(qs_transform_two_body_mixed(p, p, p, buf + 32, buf + 16, 8, 2, 2, nullptr) == QS_ERR_WORKSPACE);
    EXPECT(qs_transform_two_body_sharded_rows(nullptr, QS_F64, QS_F64, p, nullptr, p, p, buf + 32, 1 << 20, buf + 16, 1 << 20, 4, 4, 1,
                                              nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_transform_two_body_sharded_rows_workspace(QS_F64, 8, 8, 0) < 0);
    EXPECT(qs_transform_two_body_sharded_rows_workspace(QS_F64, 8, 8, 9) < 0);
    EXPECT(qs_transform_two_body_sharded_rows_workspace(QS_F64, 8, 8, 2) == 8 * (64 + 2 * 8 * 8 * 8 + 2 * 8 * 64 + 2 * 8 * 2 * 64 + 8));
    EXPECT(qs_transform_two_body_sharded_rows_out_bytes(QS_C128, 8, 6, 2, 0) == 16 * (3 * 8 * 36 + 8 * 36));
    EXPECT(qs_transform_two_body_sharded_rows_out_bytes(QS_C128, 6, 8, 2, 1) == 16 * (4 * 8 * 64 + 6 * 64));
    EXPECT(qs_transform_two_body_sharded_rows_out_bytes(9, 8, 8, 2, 0) == QS_ERR_BAD_DTYPE);
    EXPECT(qs_sharded_rows_default_chunk(QS_F64, 256, 256, 8, nullptr) == 8);
    EXPECT(qs_sharded_rows_default_chunk(QS_C128, 512, 512, 8, nullptr) == 1);
    {
        const int64_t bad_starts[3] = {0, 5, 7};             // does not end at L
        EXPECT(qs_sharded_rows_default_chunk(QS_F64, 8, 8, 2, bad_starts) == QS_ERR_BAD_EXTENT);
        const int64_t back[3] = {0, 9, 8};                   // goes backwards
        EXPECT(qs_sharded_rows_default_chunk(QS_F64, 8, 8, 2, back) == QS_ERR_BAD_EXTENT);
    }
    // its exchange plan (pure host arithmetic): balanced, uneven and spin-doubled partitions, several chunk sizes
    for (int world : {1, 2, 3, 8}) {
        for (int rank = 0; rank < world; ++rank) {
            const int64_t L = 18, M = 13;
            std::vector<int64_t> header(8), table(7 * 4096), starts(world + 1);
            for (int g = 0; g <= world; ++g) {               // twice the balanced offsets of 9 spatial rows
                const int64_t base = 9 / world, extra = 9 % world;
                starts[g] = 2 * (g * base + (g < extra ? g : extra));
            }
            for (int64_t chunk : {0, 1, 2, 5, 64}) {
                for (const int64_t* st : {(const int64_t*)nullptr, (const int64_t*)starts.data()}) {
                    const int n = qs_sharded_rows_exchange_plan(L, M, world, rank, st, chunk, header.data(), table.data(), 4096);
                    EXPECT(n >= 0);
                    const int64_t jl = header[2], r0 = header[4], out_elems = header[5], ni = header[6], nsteps = header[7];
                    EXPECT(ni >= 1 && nsteps * ni >= header[3] && (nsteps - 1) * ni < header[3]);
                    EXPECT(out_elems == jl * 18 * M * M + L * M * M && r0 + jl * L * M * M == out_elems);
                    for (int i = 0; i < n; ++i) {            // every operation stays inside the send block / the result buffer
                        const int64_t* op = &table[7 * i];
                        EXPECT(op[0] >= 0 && op[0] < nsteps && op[1] >= 0 && op[1] < world && op[5] > 0);
                        if (op[2] != 0) EXPECT(op[4] >= r0 && op[4] + (op[6] - 1) * L * M * M + op[5] <= out_elems);
                        if (op[2] != 1) EXPECT(op[3] >= 0 && op[3] + op[5] * op[6] <= M * ni * M * M);
                    }
                    if (n > 0) EXPECT(qs_sharded_rows_exchange_plan(L, M, world, rank, st, chunk, header.data(), table.data(), 0) == QS_ERR_WORKSPACE);
                }
            }
        }
    }
    // tuning state: thread-local, reset
    EXPECT(qs_tuning_set("gemm_fast", 0) == QS_OK && qs_tuning_set("sandwich_mode", 3) == QS_OK && qs_tuning_set("small4", 2) == QS_OK);
    EXPECT(qs_tuning_set("no_such_knob", 1) == QS_ERR_BAD_EXTENT && qs_tuning_set(nullptr, 1) == QS_ERR_NULL_POINTER);
    EXPECT(qs_tuning_reset() == QS_OK);
    EXPECT(qs_probe_mfma_f64(nullptr, 1, 1, nullptr) == QS_ERR_NULL_POINTER);
    EXPECT(qs_probe_stream_copy(p, buf + 32, 24, nullptr) == QS_ERR_BAD_EXTENT);
    // exchange plan of the sharded transform (pure host arithmetic) for several worlds, incl. uneven splits
    for (int world : {1, 2, 3, 8}) {
        for (int rank = 0; rank < world; ++rank) {
            const int64_t L = 19, M = 13;
            std::vector<int64_t> header(7), ct(M), chunks(4 * 16), table(7 * (4 * M + 64));
            const int n = qs_sharded_exchange_plan(L, M, world, rank, 5, header.data(), ct.data(), chunks.data(),
                                                   table.data(), (int64_t)table.size() / 7);
            EXPECT(n >= 0);
            int64_t rows = 0;
            for (int k = 0; k < 5; ++k) rows += chunks[4 * k + 1];
            EXPECT(rows == M);
            EXPECT(qs_sharded_exchange_plan(L, M, world, rank, 5, header.data(), ct.data(), chunks.data(), table.data(), 0) ==
                   (n == 0 ? 0 : QS_ERR_WORKSPACE));
        }
    }
    EXPECT(qs_sharded_exchange_plan(4, 2000, 1, 0, 4, nullptr, nullptr, nullptr, nullptr, 0) == QS_ERR_BAD_EXTENT);
    std::printf(fails ? "%d checks FAILED\n" : "all argument checks ok (%d)\n", fails);
    return fails ? 1 : 0;
}
