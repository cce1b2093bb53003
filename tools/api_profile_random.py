"""cProfile of BasisSet.change_basis for RandomBasisSet(l, 2) (host-side cost per call; the reference's own test input)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_systems_amd as qs  # noqa: E402

l = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bs = qs.RandomBasisSet(l, 2, np=qs.hip)
g = torch.Generator(device="cuda").manual_seed(3)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda", generator=g))
C = C.contiguous().to(torch.complex128)
for _ in range(5):
    bs.change_basis(C)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    bs.change_basis(C)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
