"""``QuantumSystem``: a ``BasisSet`` plus a particle number.

Thin forwarding layer with the interface of the reference's abstract base
(quantum_systems/system.py): it owns the basis set, tracks ``n``, ``l``,
``m = l - n`` and the occupied / virtual slices ``o`` / ``v``, and forwards the
basis-change and module-change calls.  The heavy work happens in
``BasisSet`` (HIP kernels)."""

import abc
import copy
from collections.abc import Iterable


class QuantumSystem(metaclass=abc.ABCMeta):
    """``n`` occupied basis functions on top of ``basis_set`` (system.py:20-30)."""

    def __init__(self, n, basis_set):
        self._basis_set = basis_set
        assert n <= self._basis_set.l
        self.np = self._basis_set.np
        self.set_system_size(n, self._basis_set.l)
        self._time_evolution_operator = []
        self._add_h_0 = True
        self._add_u_0 = True

    def set_system_size(self, n, l):
        """Record sizes and the index ranges they imply (system.py:32-51)."""
        assert n <= l
        self.n, self.l = n, l
        self.m = l - n
        self.o = slice(0, n)
        self.v = slice(n, l)

    # -- to be provided by the concrete system kinds
    @abc.abstractmethod
    def construct_fock_matrix(self, h, u, f=None):
        pass

    @abc.abstractmethod
    def change_to_hf_basis(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def compute_reference_energy(self, h=None, u=None):
        pass

    # -- forwarding (system.py:57-71, :217-225)
    def change_module(self, np):
        self.np = np
        self._basis_set.change_module(np)

    def change_basis(self, C, C_tilde=None):
        self._basis_set.change_basis(C, C_tilde)
        self.set_system_size(self.n, self._basis_set.l)

    def change_basis_plan(self, C_tilde_given=False):
        """``change_basis`` with a square ``C`` captured as a HIP graph (``basis_set.ChangeBasisPlan``): ``plan(C)`` instead of
        ``system.change_basis(C)`` in loops over a small basis; ``n``, ``l``, ``o`` and ``v`` do not change."""
        return self._basis_set.change_basis_plan(C_tilde_given)

    def transform_one_body_elements(self, h, C, C_tilde=None):
        return self._basis_set.transform_one_body_elements(h, C, np=self.np, C_tilde=C_tilde)

    def transform_two_body_elements(self, u, C, C_tilde=None):
        return self._basis_set.transform_two_body_elements(u, C, np=self.np, C_tilde=C_tilde)

    def compute_particle_density(self, rho_qp, C=None, C_tilde=None):
        return self._basis_set.compute_particle_density(rho_qp, C=C, C_tilde=C_tilde)

    # -- read-only views of the basis set (system.py:86-142)
    dim = property(lambda self: self._basis_set.dim)
    grid = property(lambda self: self._basis_set.grid)
    h = property(lambda self: self._basis_set.h, doc="one-body Hamiltonian")
    u = property(lambda self: self._basis_set.u, doc="two-body Hamiltonian")
    s = property(lambda self: self._basis_set.s, doc="overlap matrix")
    position = property(lambda self: self._basis_set.position)
    momentum = property(lambda self: self._basis_set.momentum)
    dipole_moment = property(lambda self: self._basis_set.dipole_moment)
    spf = property(lambda self: self._basis_set.spf)
    bra_spf = property(lambda self: self._basis_set.bra_spf)
    nuclear_repulsion_energy = property(lambda self: self._basis_set.nuclear_repulsion_energy)
    particle_charge = property(lambda self: self._basis_set.particle_charge)

    # -- time-dependent Hamiltonian assembly (system.py:144-215)
    def set_time_evolution_operator(self, time_evolution_operator, add_h_0=True, add_u_0=True):
        if not isinstance(time_evolution_operator, Iterable):
            time_evolution_operator = [time_evolution_operator]
        self._add_h_0 = add_h_0
        self._add_u_0 = add_u_0
        self._time_evolution_operator = [op.set_system(self) for op in time_evolution_operator]

    @property
    def has_one_body_time_evolution_operator(self):
        return any(op.is_one_body_operator for op in self._time_evolution_operator)

    @property
    def has_two_body_time_evolution_operator(self):
        return any(op.is_two_body_operator for op in self._time_evolution_operator)

    def h_t(self, current_time):
        h_0 = self._basis_set.h if self._add_h_0 else self.np.zeros_like(self._basis_set.h)
        if not self.has_one_body_time_evolution_operator:
            return h_0
        return h_0 + sum(op.h_t(current_time) for op in self._time_evolution_operator)

    def u_t(self, current_time):
        u_0 = self._basis_set.u if self._add_u_0 else self.np.zeros_like(self._basis_set.u)
        if not self.has_two_body_time_evolution_operator:
            return u_0
        return u_0 + sum(op.u_t(current_time) for op in self._time_evolution_operator)

    def copy_system(self):
        """Independent deep copy (system.py:227-250); modules are detached
        during the copy because module objects cannot be deep-copied."""
        np = self.np
        self.np = None
        self._basis_set.np = None
        try:
            new = copy.deepcopy(self)
        finally:
            self.np = np
            self._basis_set.np = np
        new.np = np
        new._basis_set.np = np
        return new
