"""Largest single-GPU fp64 size through change_basis: the dropped tensor donates its storage
(qs_transform_two_body_inplace: tensor + one spare buffer), so l = 352 (2 x 123 GB of the 288 GB) fits where the
out-of-place transform stops at l = 320.  The randomised contraction identity is evaluated in row blocks against
partial sums taken BEFORE the transform (the input no longer exists afterwards)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantum_systems_amd as qsa
from quantum_systems_amd import hip, kernels as K
dev = torch.device("cuda:0")
l = int(sys.argv[1]) if len(sys.argv) > 1 else 352
g = torch.Generator(device=dev).manual_seed(5)

def contract4(t, va, vb, vc, vd):
    A, B, C, D = t.shape
    parts = []
    for a0 in range(0, A, 2):
        x = t[a0:a0 + 2].reshape(-1, D) @ vd
        x = x.reshape(-1, C) @ vc
        parts.append(x.reshape(-1, B) @ vb)
    return torch.cat(parts) @ va

bs = qsa.BasisSet(l, 1, np=hip)
u = torch.empty((l, l, l, l), dtype=torch.float64, device=dev)
for lo in range(0, l, 4):
    u[lo:lo + 4] = torch.rand((min(4, l - lo), l, l, l), dtype=torch.float64, device=dev, generator=g)
bs.h = hip.asarray(torch.eye(l, dtype=torch.float64, device=dev))
bs.s = hip.asarray(torch.eye(l, dtype=torch.float64, device=dev))
bs.u = hip.asarray(u)
del u
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device=dev, generator=g))
C = C.contiguous(); Ct = C.T.contiguous()
x, y, z, w = (torch.randn(l, dtype=torch.float64, device=dev, generator=g) for _ in range(4))
rhs = contract4(torch.as_tensor(bs.u), Ct.T @ x, Ct.T @ y, C @ z, C @ w)
ptr = torch.as_tensor(bs.u).data_ptr()
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bs.change_basis(hip.asarray(C)); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
same = torch.as_tensor(bs.u).data_ptr() == ptr
print(f"l={l}: change_basis {t*1e3:.1f} ms (first call), {8*l**5/t/1e12:.2f} TFLOP/s, peak memory {torch.cuda.max_memory_allocated()/1e9:.0f} GB, "
      f"storage reused: {same}, dispatch: {K.last_dispatch()}", flush=True)
lhs = contract4(torch.as_tensor(bs.u), x, y, z, w)
print(f"randomised identity: relative difference {abs(lhs - rhs).item() / abs(rhs).item():.2e}")
e0.record(); bs.change_basis(hip.asarray(C.T.contiguous())); e1.record(); torch.cuda.synchronize()      # and back
t = e0.elapsed_time(e1) * 1e-3
print(f"l={l}: second change_basis {t*1e3:.1f} ms, {8*l**5/t/1e12:.2f} TFLOP/s")
