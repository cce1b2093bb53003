// Which lanes does a 64-bit LDS WRITE / READ of a wave serve in the same pass on gfx950?  THROUGHPUT probe: every lane issues long runs
// of independent ds_write_b64 / ds_read_b64 (eight in flight per iteration, no dependency between them); each 16-lane group g of the
// wave accesses 16 consecutive doubles starting at base[g] (32 doubles = all 64 banks).  Cycles per instruction tell which groups
// collide.  Build: hipcc -O3 --offload-arch=gfx950 tools/probe_lds64.hip -o tools/bin/probe_lds64
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int b0, int b1, int b2, int b3, int write, long long* out, double* sink) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int base = g == 0 ? b0 : g == 1 ? b1 : g == 2 ? b2 : b3;
    double* p = lds + base + (lane & 15);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    double v = lane, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 4000; ++it) {
        if (write) {
            p[0] = v; p[512] = v; p[1024] = v; p[1536] = v; p[2048] = v; p[2560] = v; p[3072] = v; p[3584] = v;
            v += 1.0;
            asm volatile("" ::: "memory");
        } else {
            a0 += p[0]; a1 += p[512]; a2 += p[1024]; a3 += p[1536]; a4 += p[2048]; a5 += p[2560]; a6 += p[3072]; a7 += p[3584];
            asm volatile("" ::: "memory");
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x] + v;
}
int main() {
    long long* d_out; double* d_sink;
    (void)hipMalloc(&d_out, 8 * 64); (void)hipMalloc(&d_sink, 8 * 256);
    struct { const char* name; int b[4]; } pats[] = {
        {"A 0,16,32,48   groups (0,1) and (2,3) disjoint; 0=2, 1=3 mod 32", {0, 16, 32, 48}},
        {"B 0,32,16,48   groups (0,2) and (1,3) disjoint; 0=1, 2=3 mod 32", {0, 32, 16, 48}},
        {"C 0,16,48,32   groups (0,1) disjoint; 0=3, 1=2 mod 32", {0, 16, 48, 32}},
        {"D 0,32,64,96   all four groups on the same 16 doubles", {0, 32, 64, 96}},
        {"E 0,8,16,24    overlapping halves", {0, 8, 16, 24}},
        {"F 0,10,40,50   padded complex rows {0,1,4,5} of stride 10", {0, 10, 40, 50}},
        {"G 0,8,32,40    swizzled complex rows {0,1,4,5} of stride 8", {0, 8, 32, 40}},
        {"H 0,8,16,24+.. rows {0,1,2,3} of stride 8: 0,8,16,24", {0, 8, 16, 24}},
    };
    for (int w = 0; w < 2; ++w)
        for (auto& p : pats) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 65536, 0, p.b[0], p.b[1], p.b[2], p.b[3], w, d_out, d_sink);
            (void)hipDeviceSynchronize();
            long long t; (void)hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
            printf("%s %-72s %6.2f cycles per wave-instruction (16 waves on one CU)\n", w ? "ds_write_b64" : "ds_read_b64 ", p.name, (double)t / 32000.0 / 16.0);
        }
    return 0;
}
