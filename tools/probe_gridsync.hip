// What a grid-wide barrier costs inside one launch (the l <= 32 transform wants its two passes in ONE launch):
// (a) empty kernel, (b) cooperative launch + cooperative_groups grid sync, (c) plain launch + a hand-written barrier on a
// self-resetting counter (agent-scope release / acquire), each for a few grid sizes.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_empty(double* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

__global__ void k_coop(double* p) {
    if (threadIdx.x == 0) p[blockIdx.x] = blockIdx.x;
    cg::this_grid().sync();
    if (threadIdx.x == 0) p[gridDim.x + blockIdx.x] = p[(blockIdx.x + 1) % gridDim.x];
}

// counters[0]: arrivals, counters[1]: departures; the last workgroup to leave resets both (stream order makes the next
// launch find zeros).  A bounded spin: every wave leaves the loop whatever happens.
__global__ void k_hand(double* p, unsigned* counters) {
    if (threadIdx.x == 0) p[blockIdx.x] = blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&counters[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p[gridDim.x + blockIdx.x] = p[(blockIdx.x + 1) % gridDim.x];
        if (__hip_atomic_fetch_add(&counters[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(&counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&counters[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    double* p; unsigned* c;
    hipMalloc(&p, 1 << 20); hipMalloc(&c, 64); hipMemset(c, 0, 64); hipMemset(p, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int grid : {16, 64, 128, 256, 512}) {
        float t[3] = {0, 0, 0};
        for (int which = 0; which < 3; ++which) {
            for (int r = -20; r < reps; ++r) {
                if (r == 0) hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, p);
                else if (which == 1) { void* args[] = {&p}; if (hipLaunchCooperativeKernel((void*)k_coop, dim3(grid), dim3(256), args, 0, 0) != hipSuccess) { printf("coop launch failed at grid %d\n", grid); break; } }
                else hipLaunchKernelGGL(k_hand, dim3(grid), dim3(256), 0, 0, p, c);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&t[which], e0, e1);
        }
        double chk[2]; hipMemcpy(chk, p + grid, 16, hipMemcpyDeviceToHost);
        printf("grid %4d: empty %.2f us, cooperative + grid.sync %.2f us, hand-written barrier %.2f us per launch (back to back; check %.0f %.0f)\n",
               grid, 1e3 * t[0] / reps, 1e3 * t[1] / reps, 1e3 * t[2] / reps, chk[0], chk[1]);
    }
    return 0;
}
