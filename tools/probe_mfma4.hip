// Microbenchmark: issue rate of v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks, 512 flop) next to
// v_mfma_f64_16x16x4_f64 (2048 flop) on gfx950.  Question behind it: a 55-orbital basis pads to 64 on every
// 16-wide MFMA dimension (1.38x the work of a product with two such dimensions) but only to 56 on 4-wide ones.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma4.hip -o /tmp/probe_mfma4 && /tmp/probe_mfma4
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k16(double* sink, int iters) {
    f64x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f64x4{0, 0, 0, 0};
    const double a = 0.5 + 1e-3 * threadIdx.x, b = 1.0 - 1e-3 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    double r = 0;
    for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
    if (r == 12345.678) sink[0] = r;
}

template <int NACC>
__global__ __launch_bounds__(256) void k4(double* sink, int iters) {
    double c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = 0.0;
    const double a = 0.5 + 1e-3 * threadIdx.x, b = 1.0 - 1e-3 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    double r = 0;
    for (int i = 0; i < NACC; ++i) r += c[i];
    if (r == 12345.678) sink[0] = r;
}

// dependent chain: latency of one 4x4x4 (back-to-back on the same accumulator)
__global__ __launch_bounds__(64) void k4dep(double* sink, int iters, unsigned long long* cyc) {
    double c = 0.0;
    const double a = 0.5 + 1e-3 * threadIdx.x, b = 1.0 - 1e-3 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (c == 12345.678) sink[0] = c;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static double timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    double* sink; unsigned long long* cyc;
    hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
    const int blocks = 256 * 8, iters = 4000;
    double t = timeit([&] { hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, sink, iters); });
    printf("v_mfma_f64_16x16x4_f64     : %.2f TFLOP/s\n", (double)blocks * 4 * iters * 8 * 2048 / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k4<8>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
    printf("v_mfma_f64_4x4x4_4b  8 acc : %.2f TFLOP/s\n", (double)blocks * 4 * iters * 8 * 512 / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k4<16>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
    printf("v_mfma_f64_4x4x4_4b 16 acc : %.2f TFLOP/s\n", (double)blocks * 4 * iters * 16 * 512 / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k4<32>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
    printf("v_mfma_f64_4x4x4_4b 32 acc : %.2f TFLOP/s\n", (double)blocks * 4 * iters * 32 * 512 / t / 1e12);
    hipLaunchKernelGGL(k4dep, dim3(1), dim3(64), 0, 0, sink, 1000, cyc);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent 4x4x4 chain      : %.1f s_memtime ticks per MFMA (ticks are 100 MHz-based on gfx950: see ratio)\n", (double)h / 8000.0);
    return 0;
}
