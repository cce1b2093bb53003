"""Headline GEMM kernel: time against K at a fixed tile count (fixed m, n): t = a + b K splits the per-tile fixed cost
(prologue, epilogue, tile switch) from the steady state.   python tools/k_sweep.py [n] [m]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_systems_amd import kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 256 * 64
g = torch.Generator(device="cuda").manual_seed(1)
rows = []
for k in (64, 128, 256, 384, 512, 768, 1024):
    A = torch.rand(m, k, dtype=torch.float64, device="cuda", generator=g) - 0.5
    B = torch.rand(k, n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    out = torch.empty(m, n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        K.matmul(A, B, out=out)
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            K.matmul(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    tf = 2 * m * n * k / best / 1e9
    rows.append((k, best, tf))
    print(f"K={k:5d}  {best:8.3f} ms  {tf:6.2f} TFLOP/s  {K.last_dispatch()}", flush=True)
    del A, B, out
# least squares over K >= 256
import numpy as np
ks = np.array([r[0] for r in rows if r[0] >= 256], float); ts = np.array([r[1] for r in rows if r[0] >= 256])
b, a = np.polyfit(ks, ts, 1)
print(f"fit over K >= 256: t = {a:.3f} ms + {b * 256:.3f} ms per 256 of K;  steady state {2 * m * n * 256 / (b * 256) / 1e9:.2f} TFLOP/s, "
      f"fixed part {a / (a + b * 256) * 100:.1f} % of the K=256 time")
