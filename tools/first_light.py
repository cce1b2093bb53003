"""GPU bring-up script (not a test, not the bench): probes, GEMM spot checks
against torch fp64 matmul, and first timings of the four-index transform.

    python tools/first_light.py [lmax]
"""

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import quantum_systems_amd as qsa
from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
lib = qsa._lib.load()


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def probe():
    blocks, iters = 256 * 8, 4000
    sink = torch.zeros(1 + 2 * blocks, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    med, best = timeit(lambda: lib.qs_probe_mfma_f64(sink.data_ptr(), blocks, iters, st))
    fl = blocks * 4 * iters * 8 * 2048
    print(f"mfma f64 probe: median {fl/med/1e12:.1f} TFLOP/s best {fl/best/1e12:.1f}")
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    med, best = timeit(lambda: lib.qs_probe_stream_copy(a.data_ptr(), b.data_ptr(), n, st))
    print(f"stream copy: median {2*n/med/1e12:.2f} TB/s best {2*n/best/1e12:.2f}")


def gemm_checks():
    torch.manual_seed(0)
    shapes = [(128, 128, 128), (100, 70, 33), (55, 55, 55), (257, 129, 66), (20, 20, 20),
              (1000, 18, 10), (18, 1000, 10), (64, 64, 64), (256, 256, 256), (31, 47, 5)]
    for dt in (torch.float64, torch.complex128):
        for (m, n, k) in shapes:
            A = torch.randn(m, k, dtype=dt, device=dev)
            B = torch.randn(k, n, dtype=dt, device=dev)
            C = K.matmul(A, B)
            ref = A @ B
            err = (C - ref).abs().max().item() / ref.abs().max().item()
            print(f"gemm {dt} {m}x{n}x{k}: rel err {err:.2e}")
            assert err < 1e-13, "gemm mismatch"
        # batched, shared A
        A = torch.randn(40, 24, dtype=dt, device=dev)
        B = torch.randn(7, 24, 50, dtype=dt, device=dev)
        C = K.matmul(A, B)
        ref = torch.matmul(A, B)
        err = (C - ref).abs().max().item() / ref.abs().max().item()
        print(f"gemm batched {dt}: rel err {err:.2e}")
        assert err < 1e-13


def transform_checks():
    torch.manual_seed(1)
    for dt in (torch.float64, torch.complex128):
        for (L, M) in [(6, 6), (7, 10), (20, 20), (10, 18), (33, 20), (55, 55)]:
            u = torch.randn(L, L, L, L, dtype=dt, device=dev)
            C = torch.randn(L, M, dtype=dt, device=dev)
            Ct = torch.randn(M, L, dtype=dt, device=dev)
            out = K.transform_two_body(u, C, Ct)
            ref = torch.einsum("pa,qb,abcd,cr,ds->pqrs", Ct, Ct, u, C, C)
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            print(f"transform {dt} {L}->{M}: rel err {err:.2e}")
            assert err < 1e-12


def transform_timing(lmax):
    for l in (64, 128, 192, 256):
        if l > lmax:
            break
        for dt, kf in ((torch.float64, 1), (torch.complex128, 4)):
            if dt == torch.complex128 and l > 192:
                continue
            u = torch.rand(l, l, l, l, dtype=dt, device=dev)
            C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev))
            Ct = C.conj().T.contiguous()
            out = torch.empty_like(u)
            med, best = timeit(lambda: K.transform_two_body(u, C, Ct, out=out), reps=3, warm=1)
            fl = kf * 8 * l**5
            print(f"transform {dt} l={l}: median {med*1e3:.1f} ms  {fl/med/1e12:.2f} TFLOP/s "
                  f"(best {fl/best/1e12:.2f})", flush=True)
            # randomised identity check
            x, y, z, w = (torch.randn(l, dtype=dt, device=dev) for _ in range(4))
            lhs = torch.einsum("pqrs,p,q,r,s->", out, x, y, z, w)
            rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
            print(f"   identity check rel diff {abs(lhs-rhs).item()/abs(rhs).item():.2e}")
            del u, out
            K.workspace.release()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print(torch.cuda.get_device_name(0), flush=True)
    probe()
    gemm_checks()
    transform_checks()
    transform_timing(lmax)
