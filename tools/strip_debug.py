import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def crand(*shape):
    return torch.complex(torch.randn(shape, dtype=torch.float64, device=dev, generator=g), torch.randn(shape, dtype=torch.float64, device=dev, generator=g))
l = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m, n, k = l**3, l, l
A = crand(m, k); B = crand(k, n)
ref = A @ B
def run(strip):
    K.tuning_set("gemm_fast", 0); K.tuning_set("gemm_strip", strip); K.tuning_set("gemm_stream", 0); K.tuning_set("gemm_skinny", 0)
    out = K.matmul(A, B); d = K.last_dispatch(); K.tuning_reset(); return out, d
for rep in range(3):
    gen, d0 = run(0)
    got, d1 = run(2)
    for name, x in (("general", gen), ("strip", got)):
        diff = (x - ref).abs()
        bad = (diff > 1e-9).nonzero()
        cols = sorted(set(bad[:, 1].tolist()))
        print(rep, name, "max diff vs torch", diff.max().item(), "elements off by > 1e-9:", bad.shape[0], "columns", cols[:20], "rows mod 128", sorted(set((bad[:, 0] % 128).tolist()))[:20])
print(d0, "|", d1)
