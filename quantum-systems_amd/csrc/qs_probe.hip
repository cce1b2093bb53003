// Machine probes used by bench.py to state the measured ceilings next to the
// datasheet ones: a register-resident fp64 MFMA loop (matrix-core peak at the
// clock the chip holds) and a 16-byte-per-lane streaming copy (HBM ceiling).
// Diagnostics only; nothing on the transform path calls them.

#include "qs_common.h"

namespace qs {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// 8 independent accumulators per wave, operands in registers, no memory traffic.
__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(double* sink, int iters, double seed) {
    f64x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double a = seed + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.6789) sink[0] = r;   // keep the chain alive, never true in practice
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const f64x2* __restrict__ src,
                                                          f64x2* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

}  // namespace qs

using namespace qs;

extern "C" {

// Launch `blocks` workgroups of 4 waves, each wave issuing iters*8 MFMAs.
// flops = blocks * 4 * iters * 8 * 2048.
int qs_probe_mfma_f64(void* sink, int64_t blocks, int64_t iters, void* stream) {
    if (!sink) return QS_ERR_NULL_POINTER;
    if (blocks <= 0 || iters <= 0 || blocks > (1 << 20) || iters > (1 << 24)) return QS_ERR_BAD_EXTENT;
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (double*)sink, (int)iters, 0.5);
    return launch_status("mfma probe launch");
}

// dst[0:bytes] = src[0:bytes], bytes a multiple of 16.  Moves 2*bytes.
int qs_probe_stream_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (!src || !dst) return QS_ERR_NULL_POINTER;
    if (bytes <= 0 || (bytes & 15)) return QS_ERR_BAD_EXTENT;
    if (!aligned(src, 16) || !aligned(dst, 16)) return QS_ERR_MISALIGNED;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream,
                       (const f64x2*)src, (f64x2*)dst, bytes / 16);
    return launch_status("stream copy launch");
}

}  // extern "C"
