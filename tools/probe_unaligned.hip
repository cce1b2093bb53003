// Do 16-byte buffer loads / global stores at addresses that are only 8-byte aligned work on gfx950, and at what rate?
// (Odd basis sizes give every second row of the tensor such an address; the VALU-free GEMM kernel stages them with 8-byte
// items today.)  Build: hipcc -O3 --offload-arch=gfx950 tools/probe_unaligned.hip -o /tmp/probe_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows of `pitch` doubles; every thread moves 2 adjacent doubles of a row per step (one 16-byte item)
__global__ void copy_rows(const double* src, double* dst, long rows, long pitch, long items_per_row) {
    const long total = rows * items_per_row;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / items_per_row, c = i % items_per_row;
        const double* p = src + r * pitch + 2 * c;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), (short)0, 16, 0x00020000);
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 0, 0, 0);
        f64x2 v = __builtin_bit_cast(f64x2, raw);
        *reinterpret_cast<f64x2*>(dst + r * pitch + 2 * c) = v;
    }
}
__global__ void copy_rows8(const double* src, double* dst, long rows, long pitch, long items_per_row) {
    const long total = rows * items_per_row * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (2 * items_per_row), c = i % (2 * items_per_row);
        dst[r * pitch + c] = src[r * pitch + c];
    }
}
int main() {
    for (long pitch : {256L, 254L, 253L, 255L}) {
        const long rows = 1 << 18, ipr = pitch / 2;       // (the last element of an odd row is left out)
        std::vector<double> h(rows * pitch + 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)(i % 1000003) * 0.5;
        double *s, *d;
        hipMalloc(&s, h.size() * 8); hipMalloc(&d, h.size() * 8);
        hipMemcpy(s, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (int form = 0; form < 2; ++form) {
            hipMemset(d, 0, h.size() * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto run = [&] { if (form == 0) hipLaunchKernelGGL(copy_rows, dim3(4096), dim3(256), 0, 0, s, d, rows, pitch, ipr);
                             else hipLaunchKernelGGL(copy_rows8, dim3(4096), dim3(256), 0, 0, s, d, rows, pitch, ipr); };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int k = 0; k < 5; ++k) run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<double> back(h.size());
            hipMemcpy(back.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            long bad = 0;
            for (long r = 0; r < rows; ++r) for (long c = 0; c < 2 * ipr; ++c) if (back[r * pitch + c] != h[r * pitch + c]) ++bad;
            printf("pitch %ld %s: %s, %.0f GB/s (read + write), hip error: %s\n", pitch, form ? " 8-byte items" : "16-byte items",
                   bad ? "WRONG" : "exact", 5.0 * rows * 2 * ipr * 16 / (ms * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
        }
        hipFree(s); hipFree(d);
    }
    return 0;
}
