// Which lane holds which element of v_mfma_f64_4x4x4_4b_f64?  Four independent 4x4x4 products (blocks):
// D_b[i][j] = sum_k A_b[i][k] B_b[k][j].  Every operand is one f64 per lane; a lane index splits into three
// base-4 digits (x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4).  The probe runs one MFMA on random lane
// values and tries all 6 x 6 x 6 assignments of (block, row, k / col) to the digits.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma4_layout.hip -o tools/probe_mfma4_layout
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void one(const double* a, const double* b, double* d) {
    const int lane = threadIdx.x;
    d[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], 0.0, 0, 0, 0);
}

static const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
static const char* DIG = "xyz";

int main() {
    double ha[64], hb[64], hd[64];
    srand(7);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX; hb[i] = rand() / (double)RAND_MAX; }
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice);
    hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    int found = 0;
    // operand X indexed by three roles (r0, r1, r2) placed on digits PERM[p]: lane = sum role_value << (2 * digit)
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        double worst = 0;
        for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) {
                const int la = (blk << (2 * PERM[pa][0])) | (i << (2 * PERM[pa][1])) | (k << (2 * PERM[pa][2]));
                const int lb = (blk << (2 * PERM[pb][0])) | (k << (2 * PERM[pb][1])) | (j << (2 * PERM[pb][2]));
                s = fma(ha[la], hb[lb], s);
            }
            const int ld = (blk << (2 * PERM[pd][0])) | (i << (2 * PERM[pd][1])) | (j << (2 * PERM[pd][2]));
            const double e = fabs(s - hd[ld]);
            if (e > worst) worst = e;
        }
        if (worst < 1e-12) {
            ++found;
            printf("MATCH  A: block=%c row=%c k=%c   B: block=%c k=%c col=%c   D: block=%c row=%c col=%c   (max err %.1e; bitwise-equal to a k-ordered fma chain: %s)\n",
                   DIG[PERM[pa][0]], DIG[PERM[pa][1]], DIG[PERM[pa][2]], DIG[PERM[pb][0]], DIG[PERM[pb][1]], DIG[PERM[pb][2]],
                   DIG[PERM[pd][0]], DIG[PERM[pd][1]], DIG[PERM[pd][2]], worst, worst == 0 ? "yes" : "no");
        }
    }
    printf("%d matching assignment(s); digits: x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4\n", found);
    return found ? 0 : 1;
}
