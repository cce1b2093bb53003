"""Fock matrix / reference energy on a resident tensor, timed against the bytes they touch (VERDICT r03 "next" 6): are they
really off the throughput path?  l = 256 fp64 (whole tensor) and a 32-row slab of l = 512 complex128 (what one of 16 ranks
would hold); n_occ occupied orbitals.  HIP events, best of 5."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import sharded

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("case | n_occ | Fock rows ms, GB/s of the bytes touched | reference energy ms, GB/s | one transform of the same tensor (ms)")
for (rows, l, dt, name, t_transform) in ((256, 256, torch.float64, "l=256 fp64, whole tensor", 130.0),
                                          (32, 512, torch.complex128, "l=512 complex128, 32-row slab", None)):
    es = 16 if dt.is_complex else 8
    u = torch.empty((rows, l, l, l), dtype=dt, device=dev)
    u.view(torch.float64).normal_()
    h = torch.randn((l, l), dtype=torch.float64, device=dev).to(dt)
    for n_occ in (16, 64, 128):
        for spin in (False, True):
            # bytes the formulas touch: u[p, i, q, i] (+ u[p, i, i, q]) for the Fock rows; u[i, j, i, j] (+ u[i, j, j, i]) for the energy
            fock_bytes = rows * n_occ * l * es * (1 if spin else 2)     # diagonal in the two i's: one element per (p, i, q)
            line_bytes = rows * n_occ * l * 128                          # ... but every element sits in its own 128-byte line
            t_f = timed(lambda: sharded.fock_rows(h, u, n_occ, 0, spin_orbitals=spin))
            t_e = timed(lambda: sharded.reference_energy_partial(h, u, n_occ, 0, spin_orbitals=spin))
            print(f"{name} | {n_occ:3d} {'spin' if spin else 'spatial'} | {t_f:8.3f} ms, {fock_bytes / t_f / 1e6:8.1f} GB/s "
                  f"({line_bytes / t_f / 1e6:8.1f} GB/s of cache lines) | {t_e:8.3f} ms | {t_transform}", flush=True)
    del u
    torch.cuda.empty_cache()
