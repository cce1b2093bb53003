// Bandwidth-bound kernels of the path: anti-symmetrisation, spin doubling
// (Kronecker scatter) fused with anti-symmetrisation and complex cast, the
// two-body S^2 outer products, kron(h, I2) and a small-matrix transpose.
//
// All of them are HBM-bound byte movers.  Common shape: a (p,q) pair selects an
// l x l matrix over (r,s); a workgroup takes one 32 x 32 tile of it together
// with the mirrored tile, stages both in LDS (rows read coalesced), and writes
// whole contiguous runs of the output.  Every input element is read once and
// every output element written once; the (r,s) <-> (s,r) exchange happens in
// LDS (row stride 33 elements: conflict-free column reads).
//
// Algorithmic bytes: antisymmetrise 2*e*l^4; spin expansion of a
// real tensor into complex128: 8*l^4 read + 16*(2l)^4 written = 264*l^4.

#include "qs_common.h"

namespace qs {

typedef double f64x2 __attribute__((ext_vector_type(2)));

// Store that will not be read again soon: bypasses the write-back path of L2 (measured on the
// anti-symmetrisation: 5.41 -> 5.77 TB/s at l = 256; the spin expansion, whose lanes fill a line from
// several instructions, LOSES 4 % with it and keeps plain stores).
template <typename T> __device__ __forceinline__ void stream_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
#ifndef QS_SPIN_NT
#define QS_SPIN_NT 0          // non-temporal stores in the spin-expansion / two-body S^2 kernels (A/B switch, profiles/r03_*)
#endif

static constexpr int PT = 32;        // tile edge (elements)
static constexpr int PS = PT + 1;    // LDS row stride

template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ double zero_of<double>() { return 0.0; }
template <> __device__ __forceinline__ f64x2 zero_of<f64x2>() { return f64x2{0.0, 0.0}; }

template <typename TO, typename TI> __device__ __forceinline__ TO widen(TI v);
template <> __device__ __forceinline__ double widen<double, double>(double v) { return v; }
template <> __device__ __forceinline__ f64x2 widen<f64x2, f64x2>(f64x2 v) { return v; }
template <> __device__ __forceinline__ f64x2 widen<f64x2, double>(double v) { return f64x2{v, 0.0}; }

// (ti, tj) with ti <= tj from a linear index over the upper triangle of an
// nt x nt tile grid, row by row.
__device__ __forceinline__ void tile_pair(int idx, int nt, int& ti, int& tj) {
    int i = 0;
    while (idx >= nt - i) { idx -= nt - i; ++i; }
    ti = i; tj = i + idx;
}

// Load a PT x PT tile of the l x l matrix `m` (row r0.., col c0..) into LDS.
template <typename T>
__device__ __forceinline__ void load_tile(T (*t)[PS], const T* m, int l, int r0, int c0) {
    const int tx = threadIdx.x & (PT - 1), ty = threadIdx.x / PT;   // 256 threads: ty in 0..7
#pragma unroll
    for (int rr = ty; rr < PT; rr += 8) {
        const int r = r0 + rr, c = c0 + tx;
        t[rr][tx] = (r < l && c < l) ? m[(int64_t)r * l + c] : zero_of<T>();
    }
}

// ----------------------------------------------------------------------------
// out[pq][r][s] = u[pq][r][s] - u[pq][s][r]         (basis_set.py:776-778)
// ----------------------------------------------------------------------------
// `out` may be `u` itself (in-place form): a workgroup owns a tile and its mirror tile, reads both
// into LDS, and only stores after the barrier -- so the pointers carry no __restrict__.
template <typename T>
__global__ __launch_bounds__(256) void antisym_kernel(const T* u, T* out, int l, int nt, int npairs) {
    __shared__ T t1[PT][PS];
    __shared__ T t2[PT][PS];
    const int64_t pq = blockIdx.x / npairs;
    const int pair = blockIdx.x % npairs;
    int ti, tj;
    tile_pair(pair, nt, ti, tj);
    const T* m = u + pq * (int64_t)l * l;
    T* o = out + pq * (int64_t)l * l;
    load_tile(t1, m, l, ti * PT, tj * PT);
    if (ti != tj) load_tile(t2, m, l, tj * PT, ti * PT);
    __syncthreads();
    const int tx = threadIdx.x & (PT - 1), ty = threadIdx.x / PT;
    if (ti == tj) {
#pragma unroll
        for (int rr = ty; rr < PT; rr += 8) {
            const int r = ti * PT + rr, c = tj * PT + tx;
            if (r < l && c < l) stream_store(&o[(int64_t)r * l + c], (T)(t1[rr][tx] - t1[tx][rr]));
        }
    } else {
#pragma unroll
        for (int rr = ty; rr < PT; rr += 8) {
            int r = ti * PT + rr, c = tj * PT + tx;
            if (r < l && c < l) stream_store(&o[(int64_t)r * l + c], (T)(t1[rr][tx] - t2[tx][rr]));
            r = tj * PT + rr; c = ti * PT + tx;
            if (r < l && c < l) stream_store(&o[(int64_t)r * l + c], (T)(t2[rr][tx] - t1[tx][rr]));
        }
    }
}

// ----------------------------------------------------------------------------
// Spin doubling (+ anti-symmetrisation + cast).  P = 2p+s1, Q = 2q+s2,
// R = 2r+s3, S = 2s+s4:
//   out[P,Q,R,S] = [s1==s3][s2==s4] u[p,q,r,s] - as [s1==s4][s2==s3] u[p,q,s,r]
// (basis_set.py:772-778).  One workgroup = one (p,q) and one tile pair; it
// writes, for each of the 4 (s1,s2) output matrices, the two 64 x 64 output
// blocks fed by its input tiles, zeros included.
// ----------------------------------------------------------------------------
template <typename TI, typename TO>
__device__ __forceinline__ void spin_write_block(TO* __restrict__ o, int n2, const TI (*ta)[PS],
                                                 const TI (*tb)[PS], int l, int r0, int c0,
                                                 int s1, int s2, bool as) {
    // o: (2l x 2l) output matrix of this (P,Q); ta = u tile [r][s], tb = u tile [s][r]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sc = lane >> 1, s4 = lane & 1;              // local column, spin of S
    const int c = c0 + sc;
    // 64 output rows (32 local r x 2 spins), 16 per wave
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int orow = wave * 16 + k;                   // 0..63
        const int rr = orow >> 1, s3 = orow & 1;
        const int r = r0 + rr;
        if (r < l && c < l) {
            TI v = zero_of<TI>();
            if (s1 == s3 && s2 == s4) v = ta[rr][sc];
            if (as && s1 == s4 && s2 == s3) v = v - tb[sc][rr];
#if QS_SPIN_NT
            stream_store(&o[(int64_t)(2 * r + s3) * n2 + (2 * c + s4)], widen<TO, TI>(v));
#else
            o[(int64_t)(2 * r + s3) * n2 + (2 * c + s4)] = widen<TO, TI>(v);
#endif
        }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void spin_expand_kernel(const TI* __restrict__ u, TO* __restrict__ out,
                                                          int l, int nq, int nt, int npairs, int p_lo, int as) {
    // u is a block (rows, nq, l, l) of the tensor: nq = l for a leading-index slab, fewer for a slab of
    // the second index; the output block is (2 rows, 2 nq, 2l, 2l)
    __shared__ TI t1[PT][PS];
    __shared__ TI t2[PT][PS];
    const int64_t pq = blockIdx.x / npairs;                // local (p - p_lo) * nq + q
    const int pair = blockIdx.x % npairs;
    int ti, tj;
    tile_pair(pair, nt, ti, tj);
    const int64_t pl = pq / nq, q = pq % nq;
    const TI* m = u + ((pl + p_lo) * (int64_t)nq + q) * (int64_t)l * l;
    load_tile(t1, m, l, ti * PT, tj * PT);
    if (ti != tj) load_tile(t2, m, l, tj * PT, ti * PT);
    __syncthreads();
    const int n2 = 2 * l;
    const int64_t mat = (int64_t)n2 * n2;
#pragma unroll
    for (int s12 = 0; s12 < 4; ++s12) {
        const int s1 = s12 >> 1, s2 = s12 & 1;
        TO* o = out + ((2 * pl + s1) * (int64_t)(2 * nq) + (2 * q + s2)) * mat;
        if (ti == tj) {
            spin_write_block<TI, TO>(o, n2, t1, t1, l, ti * PT, tj * PT, s1, s2, as != 0);
        } else {
            spin_write_block<TI, TO>(o, n2, t1, t2, l, ti * PT, tj * PT, s1, s2, as != 0);
            spin_write_block<TI, TO>(o, n2, t2, t1, l, tj * PT, ti * PT, s1, s2, as != 0);
        }
    }
}

// ----------------------------------------------------------------------------
// out[i, 2p+s, 2q+t] = [s==t] h[i,p,q]                 (basis_set.py:768-770)
// ----------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void kron_eye2_kernel(const TI* __restrict__ h, TO* __restrict__ out, int64_t nmat, int l) {
    const int64_t n2 = 2 * (int64_t)l;
    const int64_t total = nmat * n2 * n2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t mat = i / (n2 * n2), rem = i % (n2 * n2);
        const int P = (int)(rem / n2), Q = (int)(rem % n2);
        TI v = zero_of<TI>();
        if ((P & 1) == (Q & 1)) v = h[(mat * l + (P >> 1)) * l + (Q >> 1)];
        out[i] = widen<TO, TI>(v);
    }
}

// ----------------------------------------------------------------------------
// Two-body S^2 (basis_set.py:745-747, :525-526), complex128, n spin-orbitals:
//   out[p,q,r,s] = sum_i S_i[p,r] S_i[q,s] - as * S_i[p,s] S_i[q,r]
// accumulated in the reference's order i = x, y, z.  One workgroup per (p,q)
// and a chunk of rows r (all of them when there are enough (p,q) pairs to fill the chip); rows p and q of the
// three matrices are staged in LDS.
// ----------------------------------------------------------------------------
__device__ __forceinline__ f64x2 cmul(f64x2 a, f64x2 b) {
    return f64x2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]};
}

__global__ __launch_bounds__(1024) void spin2_tb_kernel(const f64x2* __restrict__ S, f64x2* __restrict__ out,
                                                        int n, int p_lo, int rchunks, int rows_per_chunk, int cw, int as) {
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    f64x2* sp = reinterpret_cast<f64x2*>(smem_raw);   // [3][n] rows p
    f64x2* sq = sp + 3 * n;                           // [3][n] rows q
    const int64_t blk = blockIdx.x;
    const int rc = (int)(blk % rchunks);
    const int64_t pq = blk / rchunks;
    const int q = (int)(pq % n);
    const int64_t pl = pq / n;
    const int p = (int)pl + p_lo;
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) {
        const int k = i / n, c = i % n;
        sp[i] = S[((int64_t)k * n + p) * n + c];
        sq[i] = S[((int64_t)k * n + q) * n + c];
    }
    __syncthreads();
    f64x2* o = out + (pl * n + q) * (int64_t)n * n;
    const int r_end = min(n, (rc + 1) * rows_per_chunk);
    // The workgroup is a (rows x cw columns) grid of threads: cw = the columns rounded up to whole waves (at most the
    // workgroup), so a small n does not leave most lanes idle.  A lane keeps its column s for all its rows: S_i[q, s]
    // (and S_i[p, s] for the exchange term) are read from LDS once per column, only the row factors S_i[p, r] /
    // S_i[q, r] (one address per wave: a broadcast) per row.
    const int ts = threadIdx.x % cw, tr = threadIdx.x / cw, rg = blockDim.x / cw;
    if (tr >= rg) return;     // cw does not divide the workgroup: the surplus lanes would repeat rows of row group 0
    for (int s = ts; s < n; s += cw) {
        f64x2 qs[3], ps[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { qs[k] = sq[k * n + s]; ps[k] = sp[k * n + s]; }
        for (int r = rc * rows_per_chunk + tr; r < r_end; r += rg) {
            f64x2 v = f64x2{0.0, 0.0}, w = f64x2{0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v = v + cmul(sp[k * n + r], qs[k]);
                if (as) w = w + cmul(ps[k], sq[k * n + r]);
            }
#if QS_SPIN_NT
            stream_store(&o[(int64_t)r * n + s], as ? (v - w) : v);
#else
            o[(int64_t)r * n + s] = as ? (v - w) : v;
#endif
        }
    }
}

// out (cols x rows) = in (rows x cols)^T, small matrices only.
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t rows, int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / rows, r = i % rows;   // out index i = c*rows + r
        out[i] = in[r * cols + c];
    }
}

// ------------------------------------------------------------------ launchers

static inline unsigned stream_grid(int64_t total, int block) {
    const int64_t want = cdiv(total, block);
    return (unsigned)(want < 1 ? 1 : (want > 256 * 32 ? 256 * 32 : want));
}

int transpose_small(int dtype, const void* in, void* out, int64_t rows, int64_t cols, hipStream_t stream) {
    const unsigned grid = stream_grid(rows * cols, 256);
    if (dtype == QS_F64)
        hipLaunchKernelGGL(transpose_kernel<double>, dim3(grid), dim3(256), 0, stream,
                           (const double*)in, (double*)out, rows, cols);
    else
        hipLaunchKernelGGL(transpose_kernel<f64x2>, dim3(grid), dim3(256), 0, stream,
                           (const f64x2*)in, (f64x2*)out, rows, cols);
    return launch_status("transpose_small launch");
}

int antisymmetrize(int dtype, const void* u, void* out, int64_t npq, int64_t l, hipStream_t stream) {
    const int nt = (int)cdiv(l, PT);
    const int64_t npairs = (int64_t)nt * (nt + 1) / 2;
    const int64_t nwg = npq * npairs;
    if (nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    if (dtype == QS_F64)
        hipLaunchKernelGGL(antisym_kernel<double>, dim3((unsigned)nwg), dim3(256), 0, stream,
                           (const double*)u, (double*)out, (int)l, nt, (int)npairs);
    else
        hipLaunchKernelGGL(antisym_kernel<f64x2>, dim3((unsigned)nwg), dim3(256), 0, stream,
                           (const f64x2*)u, (f64x2*)out, (int)l, nt, (int)npairs);
    note_dispatch("qs::antisym_kernel<%s>", dtype == QS_F64 ? "double" : "f64x2");
    return launch_status("antisymmetrize launch");
}

int spin_expand(int in_dtype, int out_dtype, const void* u, void* out, int64_t l, int64_t nq, int64_t p_lo,
                int64_t p_hi, int as, hipStream_t stream) {
    const int nt = (int)cdiv(l, PT);
    const int64_t npairs = (int64_t)nt * (nt + 1) / 2;
    const int64_t nwg = (p_hi - p_lo) * nq * npairs;
    if (nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    const dim3 grid((unsigned)nwg), block(256);
    if (in_dtype == QS_F64 && out_dtype == QS_F64)
        hipLaunchKernelGGL((spin_expand_kernel<double, double>), grid, block, 0, stream,
                           (const double*)u, (double*)out, (int)l, (int)nq, nt, (int)npairs, (int)p_lo, as);
    else if (in_dtype == QS_F64 && out_dtype == QS_C128)
        hipLaunchKernelGGL((spin_expand_kernel<double, f64x2>), grid, block, 0, stream,
                           (const double*)u, (f64x2*)out, (int)l, (int)nq, nt, (int)npairs, (int)p_lo, as);
    else if (in_dtype == QS_C128 && out_dtype == QS_C128)
        hipLaunchKernelGGL((spin_expand_kernel<f64x2, f64x2>), grid, block, 0, stream,
                           (const f64x2*)u, (f64x2*)out, (int)l, (int)nq, nt, (int)npairs, (int)p_lo, as);
    else
        return QS_ERR_BAD_DTYPE;
    note_dispatch("qs::spin_expand_kernel<%s, %s>", in_dtype == QS_F64 ? "double" : "f64x2",
                  out_dtype == QS_F64 ? "double" : "f64x2");
    return launch_status("spin_expand launch");
}

int kron_eye2(int in_dtype, int out_dtype, const void* h, void* out, int64_t nmat, int64_t l,
              hipStream_t stream) {
    const unsigned grid = stream_grid(nmat * 4 * l * l, 256);
    if (in_dtype == QS_F64 && out_dtype == QS_F64)
        hipLaunchKernelGGL((kron_eye2_kernel<double, double>), dim3(grid), dim3(256), 0, stream,
                           (const double*)h, (double*)out, nmat, (int)l);
    else if (in_dtype == QS_F64 && out_dtype == QS_C128)
        hipLaunchKernelGGL((kron_eye2_kernel<double, f64x2>), dim3(grid), dim3(256), 0, stream,
                           (const double*)h, (f64x2*)out, nmat, (int)l);
    else if (in_dtype == QS_C128 && out_dtype == QS_C128)
        hipLaunchKernelGGL((kron_eye2_kernel<f64x2, f64x2>), dim3(grid), dim3(256), 0, stream,
                           (const f64x2*)h, (f64x2*)out, nmat, (int)l);
    else
        return QS_ERR_BAD_DTYPE;
    note_dispatch("qs::kron_eye2_kernel");
    return launch_status("kron_eye2 launch");
}

int spin2_two_body(const void* S, void* out, int64_t n, int64_t p_lo, int64_t p_hi, int as,
                   hipStream_t stream) {
    // one workgroup stages rows p and q of the three matrices (96 n bytes) and writes rows_per_chunk * n elements:
    // whole (p, q) matrices per workgroup unless that leaves the chip short of workgroups (the first version
    // always took 8 rows: 48 KB staged per 64 KB written at n = 512)
    const int64_t npq = (p_hi - p_lo) * n;
    int64_t want = cdiv(2048, npq);
    if (want < 1) want = 1;
    if (want > cdiv(n, 8)) want = cdiv(n, 8);
    const int rows_per_chunk = (int)cdiv(n, want);
    const int rchunks = (int)cdiv(n, rows_per_chunk);
    const int64_t nwg = npq * rchunks;
    if (nwg >= (int64_t(1) << 31)) return QS_ERR_BAD_EXTENT;
    const size_t lds = sizeof(double) * 2 * 6 * n;       // rows p and q of S_x, S_y, S_z: 96 n bytes
    if (lds > 160 * 1024) return QS_ERR_BAD_EXTENT;      // n <= 1706 spin orbitals (LDS of one CU)
    static PerDeviceLds lds_opt_in;                     // beyond 64 KB (n > 682) the kernel opts in, per device
    if (int rc = opt_in_dynamic_lds((const void*)spin2_tb_kernel, lds, lds_opt_in, "hipFuncSetAttribute(spin2_tb)"))
        return rc;
    // beyond 64 KB of LDS one workgroup fits a CU: make it 1024 threads so that the CU still has 16 waves in flight
    const int threads = lds > 64 * 1024 ? 1024 : 256;
    int cw = (int)(cdiv(n, 64) * 64);
    if (cw > threads) cw = threads;
    hipLaunchKernelGGL(spin2_tb_kernel, dim3((unsigned)nwg), dim3(threads), lds, stream,
                       (const f64x2*)S, (f64x2*)out, (int)n, (int)p_lo, rchunks, rows_per_chunk, cw, as);
    note_dispatch("qs::spin2_tb_kernel");
    return launch_status("spin2_two_body launch");
}

}  // namespace qs
