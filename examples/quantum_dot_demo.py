"""End-to-end example on one MI355X: the 2-D quantum dot of BASELINE.json configs[1].

    python examples/quantum_dot_demo.py [shells]

Builds the Fock-Darwin basis (Coulomb elements generated on the GPU), moves it to the device,
rotates it into the eigenbasis of a double-well one-body Hamiltonian (the four-index transform),
spin-doubles and anti-symmetrises it, and evaluates the reference energy and the Fock matrix --
every tensor stays in HBM between the steps.  Same API as HyQD/quantum-systems.
"""

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import quantum_systems_amd as qs
from quantum_systems_amd import hip
from quantum_systems_amd.two_dim_ho import get_double_well_one_body_elements


def main():
    shells = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    l = shells * (shells + 1) // 2                       # 10 shells -> 55 orbitals
    n = 2

    t0 = time.perf_counter()
    basis = qs.TwoDimensionalHarmonicOscillator(l, 6.0, 61, omega=1.0, np=hip)
    torch.cuda.synchronize()
    print(f"{l} orbitals ({shells} shells): basis with {l}^4 Coulomb elements in {time.perf_counter() - t0:.2f} s")

    system = qs.SpatialOrbitalSystem(n, basis)
    h_dw = get_double_well_one_body_elements(l, 1.0, 1.0, 2.0, dtype=np.complex128, axis=0)
    eps, C = np.linalg.eigh(h_dw)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    system.change_basis(hip.asarray(C))                  # h, s, u, position, spf on the device
    torch.cuda.synchronize()
    print(f"change_basis: {(time.perf_counter() - t0) * 1e3:.2f} ms "
          f"(u alone is {8 * l**5 * 4 / 1e9:.1f} GFLOP as complex128)")

    gos = system.construct_general_orbital_system()     # fused spin expansion + anti-symmetrisation
    e_ref = complex(qs.array_module.to_host(gos.compute_reference_energy()))
    f = gos.construct_fock_matrix(gos.h, gos.u)
    print(f"{gos.l} spin orbitals, u is {tuple(gos.u.shape)} {gos.u.dtype}")
    print(f"reference energy of the two-electron determinant: {e_ref.real:.8f}")
    print(f"Fock matrix: {tuple(f.shape)}, lowest diagonal element {float(torch.as_tensor(f).diagonal().real.min()):.6f}")
    u = torch.as_tensor(gos.u)
    print("anti-symmetry |u_pqrs + u_pqsr| max:", float((u + u.transpose(2, 3)).abs().max()))


if __name__ == "__main__":
    main()
