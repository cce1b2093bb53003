// What does a SCATTERED 8-byte-per-lane buffer store cost the wave that issues it on gfx950, as a function of how many separate runs
// its 64 lanes write, how long the runs are, how far apart, and how large the footprint of all stores is?  (The small-basis kernels
// write their transposed result in runs of 32 bytes, sixteen per store instruction: profiles/r04_small_basis_store_positions.txt.)
// The setting of qs_sandwich4b.hip: one workgroup of four waves per CU (one wave per SIMD), every wave a chain of fp64 4x4x4 MFMAs with
// one store every MFMAS_PER_STORE instructions.  Reported: ns per loop iteration with the store, without it (stores to a zero-range
// descriptor: issued and dropped), and the difference = the price of the store.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_scattered_stores.hip -o /tmp/probe_scattered_stores
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kMfmasPerStore = 14;

// lanes_per_run lanes write one run of 8 * lanes_per_run bytes; the runs of an instruction are run_stride bytes apart; successive
// stores of a wave move on by iter_stride bytes inside a window of `window` bytes per wave (the footprint = waves x window)
__global__ __launch_bounds__(256, 1) void probe(double* out, long window, long run_stride, long iter_stride, int lanes_per_run, int iters,
                                                 int range, int shared_window) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave_id = (long)blockIdx.x * 4 + wave;
    double* base = out + (shared_window ? 0 : wave_id * (window / 8));      // shared: every wave writes the same window (a footprint that stays in cache)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, range ? (int)window : 0, 0x00020000);
    const unsigned voff = (unsigned)((lane / lanes_per_run) * run_stride + (lane % lanes_per_run) * 8);
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    unsigned soff = 0;
    const unsigned span = (unsigned)(64 / lanes_per_run) * (unsigned)run_stride;      // bytes an instruction spans
    const unsigned wrap = (unsigned)window - span;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < kMfmasPerStore; k += 3) {
            acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc1, 0, 0, 0);
            if (k + 2 < kMfmasPerStore) acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, acc2, 0, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc0), rsrc, (int)voff, (int)soff, 0);
        soff += (unsigned)iter_stride;
        if (soff >= wrap) soff -= wrap;
    }
    if (acc0 + acc1 + acc2 == 12345.678) out[0] = acc1 + acc2;      // (keeps the chains)
}

int main(int argc, char** argv) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    const int iters = 4000;
    struct Case { const char* name; int lanes_per_run; long run_stride; long iter_stride; long window; int shared; };
    const long MB = 1 << 20;
    // run_stride 24200 = 55 * 55 * 8 (the stride between the runs of a store at l = 55), 32768 = 64 * 64 * 8
    const Case cases[] = {
        {"1 run of 512 B, cached footprint", 64, 512, 512, 1 * MB, 1},
        {"1 run of 512 B, streaming footprint", 64, 512, 4096, 16 * MB, 0},
        {"4 runs of 128 B, 24200 apart, cached", 16, 24200, 128, 1 * MB, 1},
        {"4 runs of 128 B, 24200 apart, streaming", 16, 24200, 97000, 16 * MB, 0},
        {"8 runs of 64 B, 24200 apart, cached", 8, 24200, 64, 1 * MB, 1},
        {"8 runs of 64 B, 24200 apart, streaming", 8, 24200, 194000, 16 * MB, 0},
        {"16 runs of 32 B, 24200 apart, cached", 4, 24200, 32, 1 * MB, 1},
        {"16 runs of 32 B, 24200 apart, streaming", 4, 24200, 388000, 16 * MB, 0},
        {"16 runs of 32 B, 32768 apart (l = 64), streaming", 4, 32768, 524320, 16 * MB, 0},
        {"16 runs of 32 B, 4096 apart, streaming", 4, 4096, 65600, 16 * MB, 0},
        {"64 runs of 8 B, 24200 apart, streaming", 1, 24200, 1552000, 32 * MB, 0},
        // whole, aligned 128-byte lines (24192 = 189 * 128): what a tiled layout of the intermediate gives
        {"4 runs of 128 B, 24192 apart (whole lines), streaming", 16, 24192, 96768, 16 * MB, 0},
        {"2 runs of 256 B, 24192 apart (whole lines), streaming", 32, 24192, 48384, 16 * MB, 0},
        {"8 runs of 64 B, 24192 apart (half lines), streaming", 8, 24192, 193536, 16 * MB, 0},
        {"16 runs of 32 B, 24192 apart (quarter lines), streaming", 4, 24192, 387072, 16 * MB, 0},
        // quarter lines whose other three quarters are written by the same wave's next three stores (iter_stride 32: the line is
        // complete after four stores) -- what neighbouring item quads do for each other in the kernel, in the best case
        {"16 runs of 32 B, 24192 apart, lines completed by the next 3 stores", 4, 24192, 32, 16 * MB, 0},
    };
    const long waves = (long)cus * 4;
    size_t bytes = 0;
    for (const Case& c : cases) if ((size_t)c.window * (c.shared ? 1 : waves) > bytes) bytes = (size_t)c.window * (c.shared ? 1 : waves);
    double* out = nullptr;
    if (hipMalloc(&out, bytes + 4096) != hipSuccess) { printf("hipMalloc of %zu bytes failed\n", bytes); return 1; }
    hipMemset(out, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %d CUs, one workgroup of four waves each, %d MFMAs (fp64 4x4x4) per store, %d stores per wave\n", cus, kMfmasPerStore, iters);
    printf("# case | ns per iteration with the store | with the store dropped (zero range) | price of the store (ns, cycles at 2.4 GHz)\n");
    for (const Case& c : cases) {
        float ms[2] = {0, 0};
        for (int range = 1; range >= 0; --range) {
            for (int rep = 0; rep < 2; ++rep) {      // (first repetition: warm-up)
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(cus), dim3(256), 0, 0, out, c.window, c.run_stride, c.iter_stride, c.lanes_per_run, iters, range, c.shared);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[range], e0, e1);
            }
        }
        const double with = ms[1] * 1e6 / iters, without = ms[0] * 1e6 / iters;
        printf("%-58s | %7.1f | %7.1f | %6.1f ns = %5.0f cycles   (hip: %s)\n", c.name, with, without, with - without, (with - without) * 2.4,
               hipGetErrorString(hipGetLastError()));
    }
    hipFree(out);
    return 0;
}
