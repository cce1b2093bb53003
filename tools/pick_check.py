"""General GEMM kernel: tile shape picked by padded area (gemm_pick = 0) against rounds x tile work (1), on the small
products of a change_basis (alternating runs, bit-compared)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_systems_amd import kernels as K  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
cases = [("spf 55 x 55 . 55 x 10201 c128", 55, 55, 10201, torch.complex128),
         ("spf 55 x 55 . 55 x 10201 f64", 55, 55, 10201, torch.float64),
         ("h 55 x 55 . 55 x 55 f64", 55, 55, 55, torch.float64),
         ("l=20 d-contraction 8000 x 20 . 20 x 20 c128", 8000, 20, 20, torch.complex128),
         ("l=32 d-contraction 32768 x 32 . 32 x 32 f64", 32768, 32, 32, torch.float64),
         ("l=64 c128 d-contraction 262144 x 64 . 64 x 64", 262144, 64, 64, torch.complex128),
         ("grid 200 x 200 . 200 x 40000 f64", 200, 200, 40000, torch.float64)]
for name, m, k, n, dt in cases:
    A = torch.randn(m, k, dtype=dt, device="cuda", generator=g)
    B = torch.randn(k, n, dtype=dt, device="cuda", generator=g)
    out = torch.empty(m, n, dtype=dt, device="cuda")
    res = {}
    for rnd in range(2):
        for pick in (0, 1):
            K.tuning_reset(); K.tuning_set("gemm_pick", pick)
            K.matmul(A, B, out=out); torch.cuda.synchronize()
            disp = K.last_dispatch()
            ref = res.setdefault("ref", out.clone())
            same = torch.equal(out, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                K.matmul(A, B, out=out)
            e1.record(); torch.cuda.synchronize()
            res[pick] = min(res.get(pick, 1e9), e0.elapsed_time(e1) / 50 * 1e3)
            res[("d", pick)] = disp
    K.tuning_reset()
    print(f"{name:50s} area {res[0]:7.1f} us {res[('d', 0)]:45s} | rounds {res[1]:7.1f} us {res[('d', 1)]}  {'same' if same else 'DIFFERS'}")
