"""``GeneralOrbitalSystem``: matrix elements over general spin orbitals
(reference: quantum_systems/general_orbital_system.py)."""

from . import sharded_basis
from .sharded_module import is_sharded
from .system import QuantumSystem


class GeneralOrbitalSystem(QuantumSystem):
    """System in a spin-orbital basis.  A basis set without spin is
    spin-doubled first (``BasisSet.change_to_general_orbital_basis`` with the
    spinors ``a``, ``b``), and ``u`` is anti-symmetrised unless
    ``anti_symmetrize=False`` (general_orbital_system.py:39-53)."""

    def __init__(self, n, basis_set, a=[1, 0], b=[0, 1], anti_symmetrize=True, **kwargs):
        if not basis_set.includes_spin:
            basis_set = basis_set.change_to_general_orbital_basis(
                a=a, b=b, anti_symmetrize=anti_symmetrize
            )
        if anti_symmetrize:
            # covers spin bases handed in with a plain (not anti-symmetric) u;
            # a no-op when the doubling above already did it
            basis_set.anti_symmetrize_two_body_elements()
        super().__init__(n, basis_set, **kwargs)

    spin_x = property(lambda self: self._basis_set.spin_x)
    spin_y = property(lambda self: self._basis_set.spin_y)
    spin_z = property(lambda self: self._basis_set.spin_z)
    spin_2 = property(lambda self: self._basis_set.spin_2)
    spin_2_tb = property(lambda self: self._basis_set.spin_2_tb)

    def compute_reference_energy(self, h=None, u=None):
        """E0 = h_ii + 1/2 u_ijij + E_nuc over occupied i, j
        (general_orbital_system.py:75-121)."""
        o = self.o
        h = self.h if h is None else h
        u = self.u if u is None else u
        np = self.np
        if is_sharded(u):
            return sharded_basis.compute_reference_energy(h, u, self.n, True, self.nuclear_repulsion_energy)
        return (
            np.trace(h[o, o])
            + 0.5 * np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=3))
            + self.nuclear_repulsion_energy
        )

    def construct_fock_matrix(self, h, u, f=None):
        """f_pq = h_pq + u_piqi with anti-symmetric u
        (general_orbital_system.py:123-159)."""
        np = self.np
        o = self.o
        if is_sharded(u):
            return sharded_basis.construct_fock_matrix(h, u, self.n, True, f=f)
        if f is None:
            f = np.zeros_like(h)
        f.fill(0)
        f += h
        f += np.einsum("piqi -> pq", u[:, o, :, o])
        return f

    def change_to_hf_basis(self, *args, **kwargs):
        raise NotImplementedError("There is currently no GHF implementation")
