"""One-dimensional confining potentials for ``ODQD`` / ``ODSincDVR``
(reference: quantum_systems/quantum_dots/one_dim/one_dim_potentials.py).

Each potential is a callable ``V(x)`` on a grid (NumPy array or scalar) with an
optional ``derivative(x)``.  Same class names, constructor arguments and
defaults as the reference.
"""

import abc

import numpy


class OneDimPotential(metaclass=abc.ABCMeta):
    """Interface (one_dim_potentials.py:5-11)."""

    @abc.abstractmethod
    def __call__(self, x):
        ...

    def derivative(self, x):
        raise NotImplementedError()


class _PolynomialPotential(OneDimPotential):
    """V(x) = sum_k c_k x^k with coefficients from ``_powers()`` ({power: c})."""

    def _powers(self):
        raise NotImplementedError()

    def __call__(self, x):
        return sum(c * x**k for k, c in self._powers().items())

    def derivative(self, x):
        return sum(k * c * x ** (k - 1) for k, c in self._powers().items() if k)


class HOPotential(_PolynomialPotential):
    """Harmonic well ``omega^2 x^2 / 2`` (one_dim_potentials.py:14-22)."""

    def __init__(self, omega):
        self.omega = omega

    def _powers(self):
        return {2: 0.5 * self.omega**2}


class DWPotential(HOPotential):
    """Two harmonic wells a distance ``l`` apart joined by a cusp at the
    origin: ``omega^2/2 (x^2 + l^2/4 - l |x|)`` (one_dim_potentials.py:25-43)."""

    def __init__(self, omega, l):
        super().__init__(omega)
        self.l = l

    def __call__(self, x):
        shift = 0.25 * self.l**2 - self.l * abs(x)
        return super().__call__(x) + 0.5 * self.omega**2 * shift

    def derivative(self, x):
        # d|x|/dx written with the Heaviside step so that the cusp gets 0
        half_sign = numpy.heaviside(x, 0.5) - 0.5
        return super().derivative(x) - self.l * self.omega**2 * half_sign


class DWPotentialSmooth(OneDimPotential):
    """Quartic double well with minima at +-a/2:
    ``(x + a/2)^2 (x - a/2)^2 / (2 a^2)`` (one_dim_potentials.py:46-74)."""

    def __init__(self, a=4):
        self.a = a

    def _factors(self, x):
        return x + 0.5 * self.a, x - 0.5 * self.a

    def __call__(self, x):
        up, down = self._factors(x)
        return (1.0 / (2 * self.a**2)) * up**2 * down**2

    def derivative(self, x):
        up, down = self._factors(x)
        return 1 / self.a**2 * (up * down**2 + down * up**2)


class SymmetricDWPotential(_PolynomialPotential):
    """``a x^6 + b x^4 + c x^2`` (one_dim_potentials.py:77-92)."""

    def __init__(self, a=0.5, b=1, c=-7):
        self.a, self.b, self.c = a, b, c

    def _powers(self):
        return {6: self.a, 4: self.b, 2: self.c}

    def derivative(self, x):
        # the reference differentiates the quartic term as 3 b x^3 (:91-92); kept as is
        return super().derivative(x) - self.b * x**3


class AsymmetricDWPotential(_PolynomialPotential):
    """``a x^4 + b x^3 + c x^2`` (one_dim_potentials.py:95-110)."""

    def __init__(self, a=1, b=1, c=-2.5):
        self.a, self.b, self.c = a, b, c

    def _powers(self):
        return {4: self.a, 3: self.b, 2: self.c}


def _gaussian_well(x, weight, center, deviation, exp):
    """-w exp(-(x - x0)^2 / (2 sigma^2)) with the caller's ``exp``."""
    z = x - center
    return -weight * exp(-(z**2) / (2.0 * deviation**2))


class GaussianPotential(OneDimPotential):
    """Gaussian well of depth ``weight`` and width ``deviation`` around
    ``center``; the array module comes last, as in the reference
    (one_dim_potentials.py:113-127)."""

    def __init__(self, weight, center, deviation, np):
        self.weight, self.center, self.deviation, self.np = weight, center, deviation, np

    def __call__(self, x):
        return _gaussian_well(x, self.weight, self.center, self.deviation, self.np.exp)

    def derivative(self, x):
        return -(x - self.center) / self.deviation**2 * self(x)


class GaussianPotentialHardWall(OneDimPotential):
    """The same well inside ``|x| <= x_wall`` and 1e5 added outside, which
    forces the unbound states to vanish at the grid ends
    (one_dim_potentials.py:130-150)."""

    def __init__(self, weight, center, deviation, x_wall):
        self.weight, self.center, self.deviation, self.x_wall = weight, center, deviation, x_wall

    def __call__(self, x):
        x = numpy.asarray(x)
        outside = abs(x) > self.x_wall
        return _gaussian_well(x, self.weight, self.center, self.deviation, numpy.exp) + 1e5 * outside


class AtomicPotential(OneDimPotential):
    """Soft-Coulomb attraction of a nucleus of charge ``Za``:
    ``-Za / sqrt(x^2 + c)`` (one_dim_potentials.py:153-159)."""

    def __init__(self, Za=2, c=0.54878464):
        self.Za, self.c = Za, c

    def __call__(self, x):
        return -self.Za / numpy.sqrt(x**2 + self.c)

    def derivative(self, x):
        softened = x**2 + self.c
        return self.Za * x / softened ** (3 / 2)
