"""change_basis on a spin basis of 2l spin orbitals (GeneralOrbitalSystem), timed with spin_2_tb kept as its
recipe (three transformed spin matrices; this round) and with spin_2_tb as a tensor that goes through its own
four-index transform (the reference's route, basis_set.py:379-382; what round 2 did).  Usage: python tools/gos_change_basis_time.py [l]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import quantum_systems_amd as qsa
from quantum_systems_amd import hip

l = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def system():
    bs = qsa.BasisSet(l, 1, np=hip)
    h = torch.rand(l, l, dtype=torch.float64, device=dev, generator=g)
    bs.h = hip.asarray(h + h.T)
    bs.s = hip.asarray(torch.eye(l, dtype=torch.float64, device=dev))
    bs.u = hip.asarray(torch.rand(l, l, l, l, dtype=torch.float64, device=dev, generator=g))
    return qsa.GeneralOrbitalSystem(2, bs)


n = 2 * l
C = torch.linalg.qr(torch.complex(torch.randn(n, n, dtype=torch.float64, device=dev, generator=g),
                                  torch.randn(n, n, dtype=torch.float64, device=dev, generator=g)))[0]
for route in ("recipe", "tensor"):
    gos = system()
    if route == "tensor":
        tb = gos.spin_2_tb
        tb[0, 0, 0, 0] += 0.0           # an in-place write: the recipe no longer stands for the tensor
        del tb
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    gos.change_basis(hip.asarray(C))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    peak = torch.cuda.max_memory_allocated() / 1e9
    tb = gos.spin_2_tb
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"2l={n} complex128 change_basis, spin_2_tb as {route}: {1e3 * (t1 - t0):.1f} ms, peak {peak:.1f} GB; "
          f"first access of spin_2_tb afterwards {1e3 * (t2 - t1):.1f} ms "
          f"(tensor {16 * n**4 / 1e9:.1f} GB, recipe kept: {gos._basis_set._spin_2_tb_recipe is not None})")
    del gos, tb
    torch.cuda.empty_cache()
