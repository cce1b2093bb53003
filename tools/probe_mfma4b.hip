// Second look at v_mfma_f64_4x4x4_4b_f64: the first probe (tools/probe_mfma4.hip) multiplied the SAME two operand
// registers all the time.  Here the operands change from instruction to instruction, as in a real product:
// NA A-registers x NB B-registers -> NA*NB accumulators (VGPR or AGPR), issue order a-major.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma4b.hip -o tools/probe_mfma4b
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NA, int NB, bool ACC_AGPR>
__global__ __launch_bounds__(256) void kblock(double* sink, int iters) {
    double c[NA][NB];
    double a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = 0.5 + 1e-3 * (threadIdx.x + i);
    for (int j = 0; j < NB; ++j) b[j] = 1.0 - 1e-3 * (threadIdx.x + 3 * j);
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) c[i][j] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (ACC_AGPR) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(c[i][j]) : "v"(a[i]), "v"(b[j]));
                else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c[i][j]) : "v"(a[i]), "v"(b[j]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    double r = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) r += c[i][j];
    if (r == 12345.678) sink[0] = r;
}

// the sandwich kernel's first product: one A register per k step against NB B registers that CHANGE every k step
// (NK * NB different B registers), NB accumulators -- dependent distance NB
template <int NK, int NB, bool B_AGPR>
__global__ __launch_bounds__(256) void kchain(double* sink, int iters) {
    double c[NB];
    double a[NK], b[NK][NB];
    for (int k = 0; k < NK; ++k) { a[k] = 0.5 + 1e-3 * (threadIdx.x + k); for (int j = 0; j < NB; ++j) b[k][j] = 1.0 - 1e-3 * (threadIdx.x + 3 * j + k); }
    for (int j = 0; j < NB; ++j) c[j] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (B_AGPR) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a[k]), "a"(b[k][j]));
                else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a[k]), "v"(b[k][j]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    double r = 0;
    for (int j = 0; j < NB; ++j) r += c[j];
    if (r == 12345.678) sink[0] = r;
}

template <typename F>
static double timeit(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

#define RUN(label, kern, per_iter)                                                                          \
    {                                                                                                       \
        double t = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, sink, iters); });   \
        printf("%-58s %6.2f TFLOP/s\n", label, (double)blocks * 4 * iters * (per_iter) * 512 / t / 1e12);  \
    }

int main() {
    double* sink;
    (void)hipMalloc(&sink, 64);
    const int blocks = 256 * 4, iters = 2000;
    RUN("1 A x 8 B, accumulators in VGPRs", (kblock<1, 8, false>), 8)
    RUN("4 A x 4 B, accumulators in VGPRs", (kblock<4, 4, false>), 16)
    RUN("4 A x 4 B, accumulators in AGPRs", (kblock<4, 4, true>), 16)
    RUN("8 A x 4 B, accumulators in AGPRs", (kblock<8, 4, true>), 32)
    RUN("14 A x 4 B, accumulators in AGPRs (second product)", (kblock<14, 4, true>), 56)
    RUN("14 A x 3 B, accumulators in AGPRs", (kblock<14, 3, true>), 42)
    RUN("14 k x 4 B chains, B in VGPRs (first product)", (kchain<14, 4, false>), 56)
    RUN("14 k x 4 B chains, B in AGPRs", (kchain<14, 4, true>), 56)
    RUN("14 k x 3 B chains, B in VGPRs", (kchain<14, 3, false>), 42)
    RUN("14 k x 8 B chains, B in VGPRs", (kchain<14, 8, false>), 112)
    return 0;
}
