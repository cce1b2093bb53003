"""One-dimensional quantum dot on a uniform grid: ``ODQD``
(reference: quantum_systems/quantum_dots/one_dim/one_dim_qd.py:181-289).

The single-particle functions are the lowest ``l`` eigenvectors of the
three-point finite-difference Hamiltonian on the interior grid points (host,
``scipy.linalg.eigh_tridiagonal`` -- an l x N problem, as in the reference);
the two-body elements are the grid quadrature of the shielded Coulomb kernel,

    u[a,b,c,d] = sum_pq C[p,a] C[q,b] C[p,c] C[q,d] K[p,q],
    K[p,q] = alpha / sqrt((x_p - x_q)^2 + a^2),

which the reference evaluates with a five-operand ``einsum``
(one_dim_qd.py:272-280) and this module with two GEMMs on the HIP kernels
(``kernels.two_body_from_grid``; SURVEY 8f #4: same contraction family as the
four-index transform).

Not built: ``ODHO`` (analytic oscillator functions with numba trapezoid
integrals, one_dim_qd.py:71-178) -- ``ODQD`` with ``HOPotential`` covers the
same physics on the grid.
"""

import numpy
import scipy.linalg
import torch

from . import kernels
from .array_module import convert
from .basis_set import BasisSet
from .one_dim_potentials import (
    AsymmetricDWPotential,
    AtomicPotential,
    DWPotential,
    DWPotentialSmooth,
    GaussianPotential,
    HOPotential,
    SymmetricDWPotential,
)


def shielded_coulomb(x_1, x_2, alpha, a):
    """``alpha / sqrt((x_1 - x_2)^2 + a^2)`` (one_dim_qd.py:30-32)."""
    return alpha / numpy.sqrt((x_1 - x_2) ** 2 + a**2)


class ODQD(BasisSet):
    """``l`` grid eigenfunctions of a 1-D well with a shielded Coulomb
    interaction.  Same constructor, attributes (``grid``, ``potential``,
    ``eigen_energies``, ``spf``, ``h``, ``s``, ``u``, ``position``) and dtypes
    as the reference: ``h`` and ``spf`` complex128, ``u`` float64.

    >>> odqd = ODQD(20, 11, 201, potential=ODQD.HOPotential(omega=1))   # doctest: +SKIP
    >>> odqd.l
    20
    """

    HOPotential = HOPotential
    DWPotential = DWPotential
    DWPotentialSmooth = DWPotentialSmooth
    SymmetricDWPotential = SymmetricDWPotential
    AsymmetricDWPotential = AsymmetricDWPotential
    GaussianPotential = GaussianPotential
    AtomicPotential = AtomicPotential

    def __init__(self, l, grid_length, num_grid_points, a=0.25, alpha=1.0, beta=0, potential=None,
                 **kwargs):
        super().__init__(l, dim=1, **kwargs)
        self.a = a
        self.alpha = alpha
        self.grid_length = grid_length
        self.num_grid_points = num_grid_points
        self.grid = numpy.linspace(-grid_length, grid_length, num_grid_points)
        self.beta = beta
        if potential is None:
            potential = HOPotential(0.25)    # the reference's default frequency (:249-253)
        self.potential = potential
        self.setup_basis()

    def setup_basis(self):
        """one_dim_qd.py:258-289."""
        np = self.np
        x = self.grid[1:-1]                  # the functions vanish on the two end points
        dx = self.grid[1] - self.grid[0]
        diagonal = 1.0 / dx**2 + self.potential(x)
        off_diagonal = numpy.full(self.num_grid_points - 3, -1.0 / (2 * dx**2))
        eps, C = scipy.linalg.eigh_tridiagonal(
            diagonal, off_diagonal, select="i", select_range=(0, self.l - 1)
        )
        spf = numpy.zeros((self.l, self.num_grid_points), dtype=numpy.complex128)
        spf[:, 1:-1] = C.T / numpy.sqrt(dx)
        self.eigen_energies = eps

        self.spf = convert(spf, np)
        self.h = convert(numpy.diag(eps).astype(numpy.complex128), np)
        self.s = convert(numpy.eye(self.l), np)

        K = shielded_coulomb(x[None, :], x[:, None], self.alpha, self.a)
        if not torch.cuda.is_available():
            raise RuntimeError("the two-body elements are contracted on the GPU only")
        dev = torch.device("cuda", torch.cuda.current_device())
        Cd = torch.from_numpy(numpy.ascontiguousarray(C)).to(dev)
        u = kernels.two_body_from_grid(torch.from_numpy(K).to(dev), Cd, C_tilde=Cd.transpose(0, 1).contiguous())
        self.u = convert(u, np)

        position = numpy.zeros((1, self.l, self.l), dtype=numpy.complex128)
        position[0] = (C.T * (x + self.beta * x**2)) @ C
        self.position = convert(position, np)
