// Stand-alone micro-benchmark (diagnostic, not part of the library): which
// instruction class, mixed into a stream of independent fp64 MFMAs, costs
// matrix-pipe time on gfx950?  Build: hipcc -O3 --offload-arch=gfx950 probe_mix.hip -o probe_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// FLAGS bit0: 8 ds_read_b64 per 16 MFMAs feeding the operands
//       bit1: 32 VALU adds per 16 MFMAs
//       bit2: s_barrier per 16 MFMAs
//       bit3: 2 global_load_dwordx4 + 2 ds_write_b128 per 16 MFMAs
//       bit4: 64 MFMAs between barriers instead of 16 (with bit2)
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void mix_kernel(const double* __restrict__ src, double* sink, int iters) {
    __shared__ __attribute__((aligned(16))) double lds[8192];
    f64x4 c[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = f64x4{0, 0, 0, 0};
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = 0.5 + 1e-3 * (threadIdx.x + i); b[i] = 1.0 - 1e-3 * (threadIdx.x + 7 * i); }
    const int lane = threadIdx.x & 63;
    int v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7;
    const f64x2* gp = reinterpret_cast<const f64x2*>(src) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 1) {
            const volatile double* l = lds + ((it & 7) * 512) + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = l[i * 64]; b[i] = l[256 + i * 64]; }
        }
        f64x2 g0, g1;
        if (FLAGS & 8) { g0 = gp[(it & 15) * 512]; g1 = gp[(it & 15) * 512 + 256]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[i][j]) : "v"(a[i]), "v"(b[j]));
                if (FLAGS & 2) {
                    asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2" : "+v"(v0), "+v"(v1) : "v"(v2));
                }
            }
        }
        if (FLAGS & 8) {
            f64x2* w = reinterpret_cast<f64x2*>(lds + 4096) + threadIdx.x;
            w[0] = g0; w[256] = g1;
        }
        if (FLAGS & 4) { if (!(FLAGS & 16) || (it & 3) == 3) __syncthreads(); }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    double r = v0 + v1 + v3;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r += c[i][j][0] + c[i][j][3];
    if (r == 12345.6789) sink[0] = r;
}

template <int FLAGS> void run(const double* src, double* sink, int blocks, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mix_kernel<FLAGS>, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("flags=%2d blocks=%4d %-44s %6.1f TFLOP/s\n", FLAGS, blocks, what, blocks * 4.0 * iters * 16 * 2048 / best / 1e9);
}

int main() {
    double *src, *sink; hipMalloc(&src, 1 << 24); hipMalloc(&sink, 64); hipMemset(src, 0, 1 << 24);
    for (int blocks : {256, 512}) {
        run<0>(src, sink, blocks, "MFMA only");
        run<1>(src, sink, blocks, "+ 8 ds_read_b64 / 16 MFMA");
        run<2>(src, sink, blocks, "+ 32 VALU / 16 MFMA");
        run<3>(src, sink, blocks, "+ ds_read + VALU");
        run<4>(src, sink, blocks, "+ barrier / 16 MFMA");
        run<5>(src, sink, blocks, "+ ds_read + barrier / 16 MFMA");
        run<21>(src, sink, blocks, "+ ds_read + barrier / 64 MFMA");
        run<8>(src, sink, blocks, "+ 2 gload + 2 ds_write / 16 MFMA");
        run<9>(src, sink, blocks, "+ ds_read + gload + ds_write");
        run<29>(src, sink, blocks, "+ ds_read + gload + ds_write + barrier/64");
        run<31>(src, sink, blocks, "+ all (barrier/64)");
    }
    return 0;
}
