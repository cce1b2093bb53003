"""``SpatialOrbitalSystem``: closed-shell system over spatial orbitals
(reference: quantum_systems/spatial_orbital_system.py)."""

import copy

from .general_orbital_system import GeneralOrbitalSystem
from .system import QuantumSystem


class SpatialOrbitalSystem(QuantumSystem):
    """``n`` particles in doubly occupied spatial orbitals; internally the
    number of occupied basis functions is ``n // 2``
    (spatial_orbital_system.py:40-50).

    >>> from quantum_systems_amd import SpatialOrbitalSystem, BasisSet
    >>> SpatialOrbitalSystem(4, BasisSet(20, 2)).n
    2
    """

    def __init__(self, n, basis_set, **kwargs):
        assert n % 2 == 0, "n must be divisable by 2 to be a closed-shell system"
        assert not basis_set.includes_spin, (
            f"{self.__class__.__name__} only supports basis sets without "
            + "spin-dependence."
        )
        super().__init__(n // 2, basis_set, **kwargs)

    def construct_general_orbital_system(self, a=[1, 0], b=[0, 1], anti_symmetrize=True):
        """Spin-doubled copy of this system: every spatial orbital yields an
        alpha and a beta spin orbital, ``2n`` occupied, two-body elements
        anti-symmetrised by default.  This system is left intact
        (spatial_orbital_system.py:52-104)."""
        gos = GeneralOrbitalSystem(
            self.n * 2, self._basis_set.copy_basis(), a=a, b=b, anti_symmetrize=anti_symmetrize
        )
        if self._time_evolution_operator is not None:
            gos.set_time_evolution_operator(copy.deepcopy(self._time_evolution_operator))
        return gos

    def compute_reference_energy(self, h=None, u=None):
        """E0 = 2 h_ii + 2 u_ijij - u_ijji + E_nuc
        (spatial_orbital_system.py:106-150)."""
        o = self.o
        np = self.np
        h = self.h if h is None else h
        u = self.u if u is None else u
        return (
            2 * np.trace(h[o, o])
            + 2 * np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=3))
            - np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=2))
            + self.nuclear_repulsion_energy
        )

    def construct_fock_matrix(self, h, u, f=None):
        """Restricted closed-shell Fock matrix
        f_pq = h_pq + 2 u_piqi - u_piiq (spatial_orbital_system.py:152-190)."""
        np = self.np
        o = self.o
        if f is None:
            f = np.zeros_like(h)
        f.fill(0)
        f += h
        f += 2 * np.einsum("piqi -> pq", u[:, o, :, o])
        f -= np.einsum("piiq -> pq", u[:, o, o, :])
        return f

    def change_to_hf_basis(self, *args, **kwargs):
        raise NotImplementedError("There is currently no RHF implementation")
