"""``RandomBasisSet``: synthetic matrix elements with the symmetries of
second-quantised integrals (reference: quantum_systems/random_basis.py).
The benchmark / test input generator."""

from .basis_set import BasisSet


class RandomBasisSet(BasisSet):
    """Random complex ``h``, ``s`` (Hermitian), ``u`` (u_pqrs = u_qpsr) and
    ``position`` (Hermitian per axis) drawn from ``np.random`` of the array
    module in the reference's order h, s, u, position, nuclear repulsion energy,
    charge (random_basis.py:21-35) -- with NumPy as the module and a seeded
    global stream the arrays equal the reference's bit for bit.

    >>> rbs = RandomBasisSet(8, 3)
    >>> rbs.dipole_moment.shape
    (3, 8, 8)
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.setup_basis()

    def setup_basis(self):
        """Draw order matters for reproducing a seeded stream: h, s, u,
        position, then the two scalars (random_basis.py:21-35)."""
        l = self.l
        draws = {}
        for name, shape in (("h", (l, l)), ("s", (l, l)), ("u", (l,) * 4), ("position", (self.dim, l, l))):
            draws[name] = self.get_random_elements(shape, self.np)
        self.h = self.make_hermitian(draws["h"])
        self.s = self.make_hermitian(draws["s"])
        self.u = self.make_two_body_symmetry(draws["u"])
        self.position = self.make_position_elements_hermitian(draws["position"])
        rng = self.np.random
        self.nuclear_repulsion_energy = rng.random()
        self.charge = rng.choice([-1, 1])

    @staticmethod
    def make_hermitian(h):
        """(h + h^dagger) / 2"""
        return (h + h.conj().T) * 0.5

    @staticmethod
    def make_position_elements_hermitian(position):
        """Every axis made Hermitian, in place (the argument is returned)."""
        position[...] = (position + position.conj().transpose(0, 2, 1)) * 0.5
        return position

    @staticmethod
    def make_two_body_symmetry(u):
        """u_pqrs = u_qpsr: particle-exchange symmetry."""
        return (u + u.transpose(1, 0, 3, 2)) * 0.5

    @staticmethod
    def get_random_elements(shape, np):
        """Complex array, real and imaginary parts uniform on [0, 1), real part
        drawn first (random_basis.py:52-69)."""
        real = np.random.random(shape)
        return real + np.random.random(shape) * 1j
