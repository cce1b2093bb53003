"""Largest single-GPU fp64 size: l = 320 (u, out and workspace = 3 x 84 GB of the 288 GB).  Checks the
randomised contraction identity (O(l^4)) and reports the rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
l = int(sys.argv[1]) if len(sys.argv) > 1 else 320
g = torch.Generator(device=dev).manual_seed(5)
u = torch.empty((l, l, l, l), dtype=torch.float64, device=dev)
for lo in range(0, l, 4):
    u[lo:lo + 4] = torch.rand((min(4, l - lo), l, l, l), dtype=torch.float64, device=dev, generator=g)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device=dev, generator=g))
Ct = C.T.contiguous()
out = torch.empty_like(u)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.transform_two_body(u, C, Ct, out=out); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    print(f"l={l}: {t*1e3:.1f} ms, {8*l**5/t/1e12:.2f} TFLOP/s, memory {torch.cuda.max_memory_allocated()/1e9:.0f} GB", flush=True)
x, y, z, w = (torch.randn(l, dtype=torch.float64, device=dev, generator=g) for _ in range(4))
lhs = torch.einsum("pqrs,p,q,r,s->", out, x, y, z, w)
rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
print(f"randomised identity: relative difference {abs(lhs - rhs).item() / abs(rhs).item():.2e}")
