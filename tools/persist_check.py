"""Headline GEMM: tiles per workgroup (tuning key gemm_fast_persist), alternating runs at the l = 256 shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_systems_amd import kernels as K  # noqa: E402

n = 256
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 256 * 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.rand(m, k, dtype=torch.float64, device="cuda", generator=g) - 0.5
B = torch.rand(k, n, dtype=torch.float64, device="cuda", generator=g) - 0.5
out = torch.empty(m, n, dtype=torch.float64, device="cuda")
for rnd in range(2):
    for mode in (1, 0, 3, 4, 6, 8, 12, 16, 32, 64, 2):
        K.tuning_reset(); K.tuning_set("gemm_fast_persist", mode)
        K.matmul(A, B, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            K.matmul(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 3
        print(f"persist {mode:3d}: {t:8.3f} ms  {2 * m * n * k / t / 1e9:6.2f} TFLOP/s", flush=True)
K.tuning_reset()
