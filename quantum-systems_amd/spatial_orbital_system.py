"""``SpatialOrbitalSystem``: closed-shell system over spatial orbitals
(reference: quantum_systems/spatial_orbital_system.py)."""

import copy

from . import sharded_basis
from .general_orbital_system import GeneralOrbitalSystem
from .sharded_module import is_sharded
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
        # The reference deep-copies the whole basis set first (spatial_orbital_system.py:89-91,
        # basis_set.py:784-805) so that this system stays intact.  The spin doubling never modifies the
        # spatial ``u`` -- it reads it and writes the 16x larger spin tensor -- so the copy of ``u`` (34 GB at
        # l = 256, and the one thing that does not fit when ``u`` is sharded over the node) is elided: the
        # copy shares the tensor, everything O(l^2) is copied as upstream.
        bs = self._basis_set
        u = bs._u
        bs._u = None
        try:
            twin = bs.copy_basis()
        finally:
            bs._u = u
        twin._u = u
        gos = GeneralOrbitalSystem(self.n * 2, twin, a=a, b=b, anti_symmetrize=anti_symmetrize)
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
        if is_sharded(u):       # slab-local sums + one all-reduce of a single number
            return sharded_basis.compute_reference_energy(h, u, self.n, False, self.nuclear_repulsion_energy)
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
        if is_sharded(u):       # slab-local rows / partial sums + l*l numbers over the node
            return sharded_basis.construct_fock_matrix(h, u, self.n, False, f=f)
        if f is None:
            f = np.zeros_like(h)
        f.fill(0)
        f += h
        f += 2 * np.einsum("piqi -> pq", u[:, o, :, o])
        f -= np.einsum("piiq -> pq", u[:, o, o, :])
        return f

    def change_to_hf_basis(self, *args, **kwargs):
        raise NotImplementedError("There is currently no RHF implementation")
