"""Two-dimensional harmonic-oscillator quantum dot: the basis set of
BASELINE.json configs[1] (reference: quantum_systems/quantum_dots/two_dim/
two_dim_ho.py:23-139 and two_dim_helper.py:111-268).

Scope of this module: what feeds the basis transformation -- the orbital index
map, the one-body energies, the overlap and the Coulomb elements ``u`` (HIP
generator ``qs_tdho_coulomb_elements``).  The reference class additionally
tabulates the single-particle functions on a polar grid and integrates the
dipole elements numerically (two_dim_ho.py:97-139, scipy based); those are
not part of the transform path and are not built here: ``spf`` and
``position`` stay ``None``.
"""

import numpy
import torch

from . import _lib
from .array_module import convert
from .basis_set import BasisSet


def get_index_p(n, m):
    """Orbital index of the Fock-Darwin state (n, m): shells hold 1, 2, 3, ...
    states, ordered by increasing m (two_dim_helper.py:111-129)."""
    shell = 2 * n + abs(m) + 1
    previous = shell * (shell - 1) // 2
    if m == 0:
        return 0 if n == 0 else previous + shell // 2
    return previous + n if m < 0 else previous + shell - (n + 1)


def get_indices_nm(p):
    """(n, m) of orbital p -- inverse of ``get_index_p``
    (two_dim_helper.py:132-166)."""
    shell = 1
    while shell * (shell + 1) // 2 <= p:
        shell += 1
    previous = shell * (shell - 1) // 2
    k = p - previous                      # position inside the shell, m ascending
    m = -(shell - 1) + 2 * k
    return (shell - 1 - abs(m)) // 2, m


def get_shell_energy(n, m):
    """two_dim_helper.py:169-171."""
    return 2 * n + abs(m) + 1


def get_one_body_elements(num_orbitals):
    """diag(shell energies), omega = 1 (two_dim_helper.py:174-182)."""
    h = numpy.zeros((num_orbitals, num_orbitals))
    for p in range(num_orbitals):
        h[p, p] = get_shell_energy(*get_indices_nm(p))
    return h


def get_coulomb_elements(num_orbitals, p_lo=0, p_hi=None, device=None):
    """Coulomb elements ``u[p, q, r, s]`` (omega = 1) as a device tensor,
    rows ``p_lo:p_hi`` (two_dim_helper.py:185-268; generated on the GPU)."""
    if not torch.cuda.is_available():
        raise RuntimeError("the Coulomb-element generator runs on the GPU only")
    p_hi = num_orbitals if p_hi is None else p_hi
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    out = torch.empty((p_hi - p_lo,) + (num_orbitals,) * 3, dtype=torch.float64, device=device)
    _lib.check(
        _lib.load().qs_tdho_coulomb_elements(
            out.data_ptr(), num_orbitals, p_lo, p_hi, torch.cuda.current_stream().cuda_stream
        ),
        "qs_tdho_coulomb_elements",
    )
    return out


class TwoDimensionalHarmonicOscillator(BasisSet):
    """``l`` Fock-Darwin orbitals of a 2-D parabolic dot of frequency ``omega``:
    ``h = omega * diag(shell energies)``, ``s = 1``,
    ``u = sqrt(omega) * Coulomb elements`` (two_dim_ho.py:84-95).  The
    constructor keeps the reference's signature; ``radius_length`` and
    ``num_grid_points`` only define ``radius`` / ``theta`` (no spf table here)."""

    def __init__(self, l, radius_length, num_grid_points, omega=1, mass=1, verbose=False, **kwargs):
        super().__init__(l, dim=2, **kwargs)
        self.omega, self.mass, self.verbose = omega, mass, verbose
        self.radius_length, self.num_grid_points = radius_length, num_grid_points
        self.radius = numpy.linspace(0, radius_length, num_grid_points)
        self.theta = numpy.linspace(0, 2 * numpy.pi, num_grid_points)
        self.setup_basis()

    def get_indices_nm(self, p):
        return get_indices_nm(p)

    def setup_basis(self):
        np = self.np
        self._h = convert(self.omega * get_one_body_elements(self.l), np)
        self._s = convert(numpy.eye(self.l), np)
        self._u = convert(numpy.sqrt(self.omega) * get_coulomb_elements(self.l), np)
