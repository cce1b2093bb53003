"""Two-dimensional harmonic-oscillator quantum dot: the basis set of
BASELINE.json configs[1] (reference: quantum_systems/quantum_dots/two_dim/
two_dim_ho.py:23-190 and two_dim_helper.py:16-339).

``TwoDimensionalHarmonicOscillator``: orbital index map, one-body energies,
overlap, Coulomb elements ``u`` (HIP generator ``qs_tdho_coulomb_elements``),
the single-particle functions on the polar grid and the dipole elements.
``TwoDimensionalDoubleWell``: the same basis with the one-body Hamiltonian of
an oscillator split by a barrier ``-omega^2 b |x| / 2 + omega^2 b^2 / 8``.
The radial integrals, which the reference hands to sympy
(two_dim_helper.py:50-67), are evaluated in closed form here: both factors are
``(a r)^|m| L_n^|m|(a^2 r^2) exp(-a^2 r^2 / 2)``, so with t = a^2 r^2 every
integral is a finite sum of Gamma functions.

``TwoDimHarmonicOscB``: the oscillator in a perpendicular magnetic field --
orbitals ordered by their field-split energies (the reference builds a pandas
frame for this, two_dim_helper.py:380-420; a ``lexsort`` here) and Coulomb
elements generated for that orbital table.

Not built: ``TwoDimSmoothDoubleWell`` (its constructor reads ``self.a`` before
setting it and raises upstream, two_dim_ho.py:190-210; the element generator is
here as a function).
"""

import math

import numpy
import scipy.special
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


def get_coulomb_elements(num_orbitals, p_lo=0, p_hi=None, device=None, nm=None):
    """Coulomb elements ``u[p, q, r, s]`` (omega = 1) as a device tensor,
    rows ``p_lo:p_hi`` (two_dim_helper.py:185-268; generated on the GPU).
    ``nm``: optional list of (n, m) per orbital replacing the shell order
    (``get_coulomb_elements_B``, two_dim_helper.py:284-301)."""
    if not torch.cuda.is_available():
        raise RuntimeError("the Coulomb-element generator runs on the GPU only")
    p_hi = num_orbitals if p_hi is None else p_hi
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    out = torch.empty((p_hi - p_lo,) + (num_orbitals,) * 3, dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    if nm is None:
        _lib.check(_lib.load().qs_tdho_coulomb_elements(out.data_ptr(), num_orbitals, p_lo, p_hi, stream),
                   "qs_tdho_coulomb_elements")
        return out
    nm = numpy.asarray(nm, dtype=numpy.int32)
    if nm.shape != (num_orbitals, 2):
        raise ValueError("nm must list (n, m) for every orbital")
    table = torch.from_numpy(numpy.ascontiguousarray(nm.T)).to(device)        # n of all, then m of all
    max_shell = int((2 * nm[:, 0] + abs(nm[:, 1]) + 1).max())
    _lib.check(
        _lib.load().qs_tdho_coulomb_elements_nm(out.data_ptr(), table.data_ptr(), num_orbitals, max_shell,
                                                p_lo, p_hi, stream),
        "qs_tdho_coulomb_elements_nm",
    )
    return out


def get_shell_energy_B(n, m, omega_c=0, omega=1):
    """Fock-Darwin level in a perpendicular field (two_dim_helper.py:271-272)."""
    return omega * (2 * n + abs(m) + 1) - (omega_c * m) / 2


def construct_level_table(n_array, m_array, omega_c=0, omega=1):
    """Orbitals (n, m) over ``n_array x m_array`` ordered by their energy in the
    field, ties by m -- the rows of the reference's pandas frame
    (two_dim_helper.py:380-420; its degeneracy columns and the capping of the
    frame do not enter any matrix element).  Returns ``(nm, E)``: an (N, 2) int
    array and the N energies."""
    n, m = numpy.meshgrid(numpy.asarray(n_array, dtype=float), numpy.asarray(m_array, dtype=float), indexing="ij")
    n, m = n.ravel(), m.ravel()
    E = get_shell_energy_B(n, m, omega_c=omega_c, omega=omega)
    order = numpy.lexsort((m, E))                      # by E, then by m; stable
    return numpy.stack([n[order], m[order]], axis=1).astype(int), E[order]


def bohr_radius(mass, omega):
    """``sqrt(mass * omega)`` (two_dim_helper.py:36-37; an inverse length)."""
    return numpy.sqrt(mass * omega)


def spf_norm(n, m, mass, omega):
    """Normalisation of the Fock-Darwin state (n, m) (two_dim_helper.py:26-33)."""
    return bohr_radius(mass, omega) * numpy.sqrt(
        scipy.special.factorial(n) / (numpy.pi * scipy.special.factorial(n + abs(m)))
    )


def spf_radial(r, n, m, mass, omega):
    """``(a r)^|m| L_n^|m|(a^2 r^2) exp(-a^2 r^2 / 2)`` (two_dim_helper.py:44-50)."""
    a = bohr_radius(mass, omega)
    return (a * r) ** abs(m) * scipy.special.assoc_laguerre(a**2 * r**2, n, abs(m)) * numpy.exp(-(a**2) * r**2 / 2.0)


def spf_state(r, theta, p, mass, omega, indices_nm=None):
    """Orbital p on the polar grid (two_dim_helper.py:16-23)."""
    n, m = (indices_nm or get_indices_nm)(p)
    return spf_norm(n, m, mass, omega) * numpy.exp(1j * m * theta) * spf_radial(r, n, m, mass, omega)


def _laguerre_coefficients(n, alpha):
    """Exact c_i of L_n^alpha(t) = sum_i c_i t^i (rationals)."""
    from fractions import Fraction

    return [Fraction((-1) ** i * math.comb(n + alpha, n - i), math.factorial(i)) for i in range(n + 1)]


def radial_integral(n_p, m_p, n_q, m_q, mass, omega, order=1):
    """``int_0^inf r^(1+order) R_p(r) R_q(r) dr`` of two radial functions
    (two_dim_helper.py:53-67, sympy there).  With t = a^2 r^2 the integrand is
    a polynomial in t times t^((order+|m_p|+|m_q|)/2) exp(-t): a finite sum of
    Gamma(base + i + j) = Gamma(base) (base)_(i+j).  The alternating sum is
    done in exact rational arithmetic (it cancels to zero for many orbital
    pairs; in floating point the terms of ~1e12 would leave ~1e-9 behind) and
    only the common factor Gamma(base) is a float."""
    from fractions import Fraction

    a = bohr_radius(mass, omega)
    mp, mq = abs(m_p), abs(m_q)
    base = Fraction(order + mp + mq, 2) + 1
    cp, cq = _laguerre_coefficients(n_p, mp), _laguerre_coefficients(n_q, mq)
    rising = [Fraction(1)]                               # (base)_k = base (base+1) ... (base+k-1)
    for k in range(len(cp) + len(cq) - 2):
        rising.append(rising[-1] * (base + k))
    total = sum(ci * cj * rising[i + j] for i, ci in enumerate(cp) for j, cj in enumerate(cq))
    return float(total) * scipy.special.gamma(float(base)) / (2.0 * a ** (2 + order))


def theta_1_integral(m_p, m_q):
    """``int exp(-i m_p t) cos t exp(i m_q t) dt`` (two_dim_helper.py:78-82)."""
    return numpy.pi if abs(m_p - m_q) == 1 else 0


def theta_2_integral(m_p, m_q):
    """sin t in place of cos t (two_dim_helper.py:85-89)."""
    return -(m_p - m_q) * 1j * numpy.pi if abs(m_p - m_q) == 1 else 0


def theta_1_tilde_integral(m_p, m_q):
    """|cos t| in place of cos t (two_dim_helper.py:92-98)."""
    d = m_p - m_q
    if abs(d) % 2 == 1:
        return 0
    sign = 1 if (abs(d) // 2) % 2 == 0 else -1
    return sign * 4 / (1 - d**2)


def theta_2_tilde_integral(m_p, m_q):
    """|sin t| (two_dim_helper.py:101-105)."""
    d = m_p - m_q
    return 0 if abs(d) % 2 == 1 else 4 / (1 - d**2)


def get_double_well_one_body_elements(num_orbitals, omega, mass, barrier_strength, dtype=numpy.float64, axis=0):
    """One-body Hamiltonian of the oscillator with the barrier
    ``omega^2 (b^2/8 - b |x_axis| / 2)`` in the Fock-Darwin basis
    (two_dim_helper.py:304-339)."""
    h = numpy.zeros((num_orbitals, num_orbitals), dtype=dtype)
    theta = theta_1_tilde_integral if axis == 0 else theta_2_tilde_integral
    nm = [get_indices_nm(p) for p in range(num_orbitals)]
    for p, (n_p, m_p) in enumerate(nm):
        h[p, p] += omega * get_shell_energy(n_p, m_p) + omega**2 * barrier_strength**2 / 8.0
        for q, (n_q, m_q) in enumerate(nm):
            if abs(m_p - m_q) == 1:
                continue
            h[p, q] -= (
                0.5 * omega**2 * barrier_strength
                * spf_norm(n_p, m_p, mass, omega) * spf_norm(n_q, m_q, mass, omega)
                * radial_integral(n_p, m_p, n_q, m_q, mass, omega)
                * theta(m_p, m_q)
            )
    return h


def get_smooth_double_well_one_body_elements(num_orbitals, omega, mass, a=2, b=2, dtype=numpy.float64):
    """two_dim_helper.py:342-379 (quartic / quadratic radial moments)."""
    h = numpy.zeros((num_orbitals, num_orbitals), dtype=dtype)
    prefactor = omega**2 / 4
    nm = [get_indices_nm(p) for p in range(num_orbitals)]
    for p, (n_p, m_p) in enumerate(nm):
        h[p, p] += omega * get_shell_energy(n_p, m_p) + omega**2 * a**2 / 64
        for q, (n_q, m_q) in enumerate(nm):
            norm = spf_norm(n_p, m_p, mass, omega) * spf_norm(n_q, m_q, mass, omega)
            parity = (-1) ** (abs(m_p - m_q) % 2)
            h[p, q] += prefactor / a**2 * norm * radial_integral(n_p, m_p, n_q, m_q, mass, omega, order=4) * parity * (3 * numpy.pi / 4)
            h[p, q] -= prefactor * (5 * b / 2) * norm * radial_integral(n_p, m_p, n_q, m_q, mass, omega, order=2) * parity * numpy.pi
    return h


class TwoDimensionalHarmonicOscillator(BasisSet):
    """``l`` Fock-Darwin orbitals of a 2-D parabolic dot of frequency ``omega``:
    ``h = omega * diag(shell energies)``, ``s = 1``,
    ``u = sqrt(omega) * Coulomb elements``, the orbitals on the polar grid
    ``radius x theta`` and the dipole elements (two_dim_ho.py:60-139).  Same
    constructor as the reference."""

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
        self.setup_spf()
        self.construct_position_integrals()

    def setup_spf(self):
        """Orbitals on ``meshgrid(radius, theta)`` (two_dim_ho.py:96-108)."""
        self.R, self.T = numpy.meshgrid(self.radius, self.theta)
        spf = numpy.empty((self.l, self.num_grid_points, self.num_grid_points), dtype=numpy.complex128)
        for p in range(self.l):
            spf[p] = spf_state(self.R, self.T, p, self.mass, self.omega, self.get_indices_nm)
        self._spf = convert(spf, self.np)

    def construct_position_integrals(self):
        """<p| x |q>, <p| y |q>: non-zero for |m_p - m_q| = 1 only
        (two_dim_ho.py:113-139)."""
        position = numpy.zeros((2, self.l, self.l), dtype=numpy.complex128)
        nm = [self.get_indices_nm(p) for p in range(self.l)]
        for p, (n_p, m_p) in enumerate(nm):
            for q, (n_q, m_q) in enumerate(nm):
                if abs(m_p - m_q) != 1:
                    continue
                amp = (spf_norm(n_p, m_p, self.mass, self.omega) * spf_norm(n_q, m_q, self.mass, self.omega)
                       * radial_integral(n_p, m_p, n_q, m_q, self.mass, self.omega))
                position[0, p, q] = amp * theta_1_integral(m_p, m_q)
                position[1, p, q] = amp * theta_2_integral(m_p, m_q)
        self._position = convert(position, self.np)


class TwoDimensionalDoubleWell(TwoDimensionalHarmonicOscillator):
    """The oscillator basis with the double-well one-body Hamiltonian
    (two_dim_ho.py:142-190): ``barrier_strength`` b, barrier along ``axis``
    (0 = x, 1 = y)."""

    def __init__(self, *args, barrier_strength=1, axis=0, **kwargs):
        self.barrier_strength = barrier_strength
        self.axis = axis
        super().__init__(*args, **kwargs)

    def setup_basis(self):
        super().setup_basis()
        self._h = convert(
            get_double_well_one_body_elements(self.l, self.omega, self.mass, self.barrier_strength,
                                              dtype=numpy.complex128, axis=self.axis),
            self.np,
        )
        self.change_module(self.np)


class TwoDimHarmonicOscB(TwoDimensionalHarmonicOscillator):
    """The 2-D oscillator in a homogeneous perpendicular magnetic field of
    cyclotron frequency ``omega_c`` (two_dim_ho.py:213-276): effective frequency
    ``sqrt(omega^2 + omega_c^2 / 4)``, orbitals ordered by their energy in the
    field, ``h = diag(E)``, Coulomb elements for that orbital table
    (``qs_tdho_coulomb_elements_nm``).  Everything is cast to complex128."""

    def __init__(self, *args, omega_c=0, **kwargs):
        self.omega_c = omega_c
        super().__init__(*args, **kwargs)

    def setup_basis(self):
        np = self.np
        self.omega = numpy.sqrt(self.omega**2 + self.omega_c**2 / 4)
        l = self.l
        self.level_nm, self.level_energy = construct_level_table(
            numpy.arange(l), numpy.arange(-l - 5, l + 6), omega_c=self.omega_c, omega=self.omega)
        self._h = convert(numpy.diag(self.level_energy[:l]), np)
        self._s = convert(numpy.eye(l), np)
        self._u = convert(numpy.sqrt(self.omega) * get_coulomb_elements(l, nm=self.level_nm[:l]), np)
        self.setup_spf()
        self.construct_position_integrals()
        self.cast_to_complex()
        self.change_module(np)

    def get_indices_nm(self, p):
        n, m = self.level_nm[p]
        return int(n), int(m)
