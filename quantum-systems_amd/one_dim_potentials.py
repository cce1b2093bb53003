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


class HOPotential(OneDimPotential):
    """Harmonic well ``omega^2 x^2 / 2`` (one_dim_potentials.py:14-22)."""

    def __init__(self, omega):
        self.omega = omega

    def __call__(self, x):
        return 0.5 * self.omega**2 * x**2

    def derivative(self, x):
        return self.omega**2 * x


class DWPotential(HOPotential):
    """Two harmonic wells a distance ``l`` apart joined by a cusp:
    ``omega^2/2 (x^2 + l^2/4 - l |x|)`` (one_dim_potentials.py:25-43)."""

    def __init__(self, omega, l):
        super().__init__(omega)
        self.l = l

    def __call__(self, x):
        return super().__call__(x) + 0.5 * self.omega**2 * (0.25 * self.l**2 - self.l * abs(x))

    def derivative(self, x):
        # sign(x)/2 written with the Heaviside step: defined (as 0) at the cusp
        return super().derivative(x) - self.l * self.omega**2 * (numpy.heaviside(x, 0.5) - 0.5)


class DWPotentialSmooth(OneDimPotential):
    """Quartic double well ``(x + a/2)^2 (x - a/2)^2 / (2 a^2)``
    (one_dim_potentials.py:46-74)."""

    def __init__(self, a=4):
        self.a = a

    def __call__(self, x):
        return (x + 0.5 * self.a) ** 2 * (x - 0.5 * self.a) ** 2 / (2 * self.a**2)

    def derivative(self, x):
        lo, hi = x - 0.5 * self.a, x + 0.5 * self.a
        return (hi * lo**2 + lo * hi**2) / self.a**2


class SymmetricDWPotential(OneDimPotential):
    """``a x^6 + b x^4 + c x^2`` (one_dim_potentials.py:77-92; the reference's
    ``derivative`` uses 3 b x^3 for the quartic term and so does this one)."""

    def __init__(self, a=0.5, b=1, c=-7):
        self.a, self.b, self.c = a, b, c

    def __call__(self, x):
        return self.a * x**6 + self.b * x**4 + self.c * x**2

    def derivative(self, x):
        return 6 * self.a * x**5 + 3 * self.b * x**3 + 2 * self.c * x


class AsymmetricDWPotential(OneDimPotential):
    """``a x^4 + b x^3 + c x^2`` (one_dim_potentials.py:95-110)."""

    def __init__(self, a=1, b=1, c=-2.5):
        self.a, self.b, self.c = a, b, c

    def __call__(self, x):
        return self.a * x**4 + self.b * x**3 + self.c * x**2

    def derivative(self, x):
        return 4 * self.a * x**3 + 3 * self.b * x**2 + 2 * self.c * x


class GaussianPotential(OneDimPotential):
    """Gaussian well ``-w exp(-(x - x0)^2 / (2 sigma^2))``; takes the array
    module as its last argument like the reference (one_dim_potentials.py:113-127)."""

    def __init__(self, weight, center, deviation, np):
        self.weight, self.center, self.deviation, self.np = weight, center, deviation, np

    def __call__(self, x):
        return -self.weight * self.np.exp(-((x - self.center) ** 2) / (2.0 * self.deviation**2))

    def derivative(self, x):
        return -(x - self.center) / self.deviation**2 * self(x)


class GaussianPotentialHardWall(OneDimPotential):
    """Gaussian well plus a 1e5 wall outside ``|x| > x_wall``
    (one_dim_potentials.py:130-150)."""

    def __init__(self, weight, center, deviation, x_wall):
        self.weight, self.center, self.deviation, self.x_wall = weight, center, deviation, x_wall

    def __call__(self, x):
        x = numpy.asarray(x)
        wall = numpy.where(abs(x) > self.x_wall, 1e5, 0.0)
        return -self.weight * numpy.exp(-((x - self.center) ** 2) / (2.0 * self.deviation**2)) + wall


class AtomicPotential(OneDimPotential):
    """Soft-Coulomb attraction ``-Za / sqrt(x^2 + c)`` (one_dim_potentials.py:153-159)."""

    def __init__(self, Za=2, c=0.54878464):
        self.Za, self.c = Za, c

    def __call__(self, x):
        return -self.Za / numpy.sqrt(x**2 + self.c)

    def derivative(self, x):
        return self.Za * x / (x**2 + self.c) ** 1.5
