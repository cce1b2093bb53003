"""Upper bound on what fusing two contractions in a tile could gain: the same product
Out[l, N] = Ct[l, l] . T[l, N] timed with N = l^3 (operand and result stream through HBM, as in the transform) and with
N small enough that both stay in the caches (no HBM traffic at all), many repetitions.  Fusing d + c removes ONE of the four
intermediate write + read pairs, i.e. a quarter of the difference."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def rate(l, n, batch, cached, reps):
    """batch products of (l x l) . (l x n); cached: every batch entry reads the SAME operand and writes the SAME result"""
    Ct = torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)
    nb = 1 if cached else batch
    T = torch.randn((nb, l, n), dtype=torch.float64, device=dev, generator=g)
    out = torch.empty_like(T)
    st = 0 if cached else l * n

    def run():
        K.gemm_raw(torch.float64, Ct, T, out, l, n, l, l, n, n, batch=batch, sa=0, sb=st, sc=st)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return 2.0 * l * l * n * batch / best / 1e9, best, K.last_dispatch()[:60]


print("l  n x batch  operand  TFLOP/s  ms  kernel")
for l in (128, 256):
    for n in (l * l // 2, 8192):
        batch = l**3 // n
        for cached in (False, True):
            r, ms, k = rate(l, n, batch, cached, 10 if l == 128 else 3)
            mb = 16.0 * l * n * (1 if cached else batch) / 2**20
            print(f"{l:4d} n={n:6d} x {batch:5d} {'same' if cached else 'own '} operand/result per entry ({mb:8.1f} MiB touched) "
                  f"{r:7.2f} TFLOP/s {ms:9.4f} ms  {k}", flush=True)
