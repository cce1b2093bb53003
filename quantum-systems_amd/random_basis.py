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
        np, l = self.np, self.l
        self.h = self.make_hermitian(self.get_random_elements((l, l), np))
        self.s = self.make_hermitian(self.get_random_elements((l, l), np))
        self.u = self.make_two_body_symmetry(self.get_random_elements((l, l, l, l), np))
        self.position = self.make_position_elements_hermitian(
            self.get_random_elements((self.dim, l, l), np)
        )
        self.nuclear_repulsion_energy = np.random.random()
        self.charge = np.random.choice([-1, 1])

    @staticmethod
    def make_hermitian(h):
        return 0.5 * (h + h.conj().T)

    @staticmethod
    def make_position_elements_hermitian(position):
        for axis in range(len(position)):
            position[axis] = RandomBasisSet.make_hermitian(position[axis])
        return position

    @staticmethod
    def make_two_body_symmetry(u):
        return 0.5 * (u + u.transpose(1, 0, 3, 2))

    @staticmethod
    def get_random_elements(shape, np):
        """Complex array, real and imaginary parts uniform on [0, 1), real part
        drawn first (random_basis.py:52-69)."""
        return np.random.random(shape) + 1j * np.random.random(shape)
