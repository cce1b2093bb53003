"""One-dimensional sinc-DVR basis: ``ODSincDVR``
(reference: quantum_systems/sinc_dvr/one_dim/sinc_dvr.py:22-263).

In a DVR basis a local interaction is diagonal, ``u[a,b,c,d] = K[a,b]
delta_ac delta_bd``, so the class can keep ``u`` as the l x l matrix ``K``
("2d" representation) and only the FIRST change of basis produces a rank-4
tensor -- through ``kernels.two_body_from_grid`` (two GEMMs on the HIP kernels,
optional fused anti-symmetrisation) instead of the reference's five-operand
einsums (:238-256).

Mirrors what the reference class does, including what it does not do:
``set_u_repr`` builds the other representation and does not store it (:118-138),
``change_module`` is refused for the 2-d form (:190-198), and
``change_to_general_orbital_basis`` takes no spinor arguments (:180-188), so
``GeneralOrbitalSystem(n, ODSincDVR(...))`` raises ``TypeError`` exactly as it
does upstream.
"""

import warnings

import numpy

from . import kernels
from .array_module import convert
from .basis_set import BasisSet, _deliver, _stage
from .one_dim_potentials import (
    AsymmetricDWPotential,
    AtomicPotential,
    DWPotential,
    DWPotentialSmooth,
    GaussianPotential,
    HOPotential,
    SymmetricDWPotential,
)
from .one_dim_qd import shielded_coulomb


class ODSincDVR(BasisSet):
    """``l`` sinc functions on ``linspace(-grid_length, grid_length, l)``."""

    HOPotential = HOPotential
    DWPotential = DWPotential
    DWPotentialSmooth = DWPotentialSmooth
    SymmetricDWPotential = SymmetricDWPotential
    AsymmetricDWPotential = AsymmetricDWPotential
    GaussianPotential = GaussianPotential
    AtomicPotential = AtomicPotential

    def __init__(self, l, grid_length, a=0.25, alpha=1.0, beta=0, potential=None, u_repr="2d", **kwargs):
        if u_repr not in ("2d", "4d"):
            raise ValueError("Invalid u_repr value: '{}'".format(u_repr))
        super().__init__(l, dim=1, **kwargs)
        self.alpha, self.a, self.beta = alpha, a, beta
        self.grid_length = grid_length
        self.grid = numpy.linspace(-grid_length, grid_length, self.l)
        self.num_grid_points = self.l
        self.potential = HOPotential(0.25) if potential is None else potential
        self.setup_basis(u_repr)

    # ------------------------------------------------------------ representation
    @property
    def u_repr(self):
        ndim = len(self.u.shape)
        return {2: "2d", 4: "4d"}.get(ndim, "unknown")

    @property
    def sparse_repr(self):
        return self.u_repr == "2d"

    def set_u_repr(self, new_repr):
        """Builds the requested representation and -- like the reference
        (:118-138) -- does not store it."""
        if new_repr == self.u_repr:
            print("u repr is already {}, doing nothing".format(new_repr))
        elif new_repr not in ("2d", "4d"):
            raise ValueError("'{}' is not a valid representation".format(new_repr))

    # ------------------------------------------------------------------- setup
    def setup_basis(self, u_repr):
        """sinc_dvr.py:88-116: kinetic matrix of the sinc basis + potential on
        the diagonal, unit overlap, the sinc functions on their own grid,
        Coulomb kernel, diagonal position; everything complex128 at the end."""
        np, l, x = self.np, self.l, self.grid
        self.dx = x[1] - x[0]
        idx = numpy.arange(l)
        diff = idx[:, None] - idx[None, :]
        h = numpy.zeros((l, l), dtype=numpy.complex128)
        off = diff != 0
        h[off] = (-1.0) ** diff[off] / (self.dx**2 * diff[off] ** 2)
        h[idx, idx] = numpy.pi**2 / (6 * self.dx**2) + self.potential(x)
        self.h = convert(h, np)
        self.s = convert(self.construct_s(), np)
        self.spf = convert(self.construct_sinc_grid(), np)
        self.u = convert(self.construct_coulomb_elements(u_repr), np)
        self.construct_position_integrals()
        self.cast_to_complex()

    def construct_s(self):
        return numpy.eye(self.l)

    def construct_sinc_grid(self):
        x = self.grid
        return numpy.sinc((x - x[:, None]) / self.dx) / numpy.sqrt(self.dx)

    def construct_position_integrals(self):
        position = numpy.zeros((1, self.l, self.l), dtype=numpy.complex128)
        position[0] = numpy.diag(self.grid + self.beta * self.grid**2)
        self.position = convert(position, self.np)

    def construct_coulomb_elements(self, u_repr="4d"):
        """``K[p,q] = alpha / sqrt((x_p - x_q)^2 + a^2)`` as an l x l matrix
        ("2d") or scattered onto ``u[p,q,p,q]`` ("4d") (:146-173)."""
        x = self.grid
        K = shielded_coulomb(x[:, None], x[None, :], self.alpha, self.a)
        if u_repr == "2d":
            return K
        u = numpy.zeros((self.l,) * 4)
        idx = numpy.arange(self.l)
        u[idx[:, None], idx[None, :], idx[:, None], idx[None, :]] = K
        return u

    # --------------------------------------------------------------- overrides
    def change_to_general_orbital_basis(self, anti_symmetrize=True):
        if anti_symmetrize and self.u_repr == "2d":
            if self.l > 100:
                warnings.warn("Warning, l large. Change to gos with anti_symmetrize=True forces 4d u.")
            self.set_u_repr("4d")
        return super().change_to_general_orbital_basis(anti_symmetrize=anti_symmetrize)

    def change_module(self, np):
        if self.sparse_repr:
            self.np = np
            warnings.warn("change_module not implemented for sparse u, doing nothing")
        else:
            return super().change_module(np)

    @staticmethod
    def add_spin_two_body(u, np):
        """2-d form: spin symmetry coincides with the DVR symmetry, every
        element is kept: ``kron(u, ones(2,2))`` (:200-208)."""
        if len(u.shape) == 2:
            return np.kron(u, np.ones((2, 2)))
        return BasisSet.add_spin_two_body(u, np)

    @staticmethod
    def anti_symmetrize_u(_u):
        if len(_u.shape) == 2:
            return _u
        return BasisSet.anti_symmetrize_u(_u)

    def transform_two_body_elements(self, u, C, np, anti_symmetrize=False, C_tilde=None):
        """Rank-4 transformed elements from either representation; the 2-d
        one may be anti-symmetrised on the fly (:217-263)."""
        if self.u_repr == "2d":
            Ct = None if C_tilde is None else _stage(C_tilde)
            out = kernels.two_body_from_grid(_stage(u), _stage(C), Ct, antisymmetrize=anti_symmetrize)
            return _deliver(out, np)
        assert not anti_symmetrize, "antisymmetrize only valid for sparse storage of u"
        return BasisSet.transform_two_body_elements(u, C, np, C_tilde=C_tilde)
