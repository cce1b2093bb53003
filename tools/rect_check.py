"""Rectangular transforms at sizes that go through the fast kernels: randomised contraction identity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for (L, M, dt) in [(192, 128, torch.float64), (128, 192, torch.float64), (160, 96, torch.complex128), (96, 136, torch.complex128), (200, 56, torch.float64)]:
    u = torch.rand(L, L, L, L, dtype=torch.float64, device=dev, generator=g).to(dt)
    C = (torch.randn(L, M, dtype=torch.float64, device=dev, generator=g) / L**0.5).to(dt)
    Ct = (torch.randn(M, L, dtype=torch.float64, device=dev, generator=g) / L**0.5).to(dt)
    if dt.is_complex:
        C = C + 1j * (torch.randn(L, M, dtype=torch.float64, device=dev, generator=g) / L**0.5)
    out = K.transform_two_body(u, C, Ct)
    x, y, z, w = (torch.randn(M, dtype=torch.float64, device=dev, generator=g).to(dt) for _ in range(4))
    lhs = torch.einsum("pqrs,p,q,r,s->", out, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    print(f"L={L} M={M} {str(dt)[6:]}: out {tuple(out.shape)}, identity rel diff {abs(lhs - rhs).item() / abs(rhs).item():.2e}", flush=True)
    del u, out; K.workspace.release(); torch.cuda.empty_cache()
