"""API-level cost of change_basis at small l: wall time per call against the time of the u kernels alone.
    python tools/api_overhead.py [l] [random|dot]      (dot: TwoDimensionalHarmonicOscillator, BASELINE.json configs[1])"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_systems_amd as qs  # noqa: E402
from quantum_systems_amd import kernels as K  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 55
kind = sys.argv[2] if len(sys.argv) > 2 else "dot"
t0 = time.perf_counter()
if kind == "dot":
    bs = qs.TwoDimensionalHarmonicOscillator(l, 5.0, 101, np=qs.hip)
else:
    bs = qs.RandomBasisSet(l, 2, np=qs.hip)
torch.cuda.synchronize()
t_setup = time.perf_counter() - t0
ut = torch.as_tensor(bs.u)
g = torch.Generator(device="cuda").manual_seed(3)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda", generator=g))
C = C.contiguous().to(ut.dtype) if ut.is_complex() else C.contiguous()
for _ in range(3):
    bs.change_basis(C)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    bs.change_basis(C)
torch.cuda.synchronize()
t_api = (time.perf_counter() - t0) / n
t_plan = None
if getattr(bs, "spf", None) is None:
    plan = bs.change_basis_plan()
    for _ in range(3):
        plan(C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan(C)
    torch.cuda.synchronize()
    t_plan = (time.perf_counter() - t0) / n
ut = torch.as_tensor(bs.u).clone()
Ct = C.conj().T.contiguous()
out = torch.empty_like(ut)
for _ in range(3):
    K.transform_two_body(ut, C, Ct, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    K.transform_two_body(ut, C, Ct, out=out)
torch.cuda.synchronize()
t_k = (time.perf_counter() - t0) / n
print(f"{kind} l={l} ({ut.dtype}): setup {t_setup:.2f} s; change_basis {t_api * 1e6:.0f} us per call; "
      + (f"ChangeBasisPlan {t_plan * 1e6:.0f} us per call; " if t_plan is not None else "") + "the two-body "
      f"transform alone {t_k * 1e6:.0f} us | {K.last_dispatch()}")
