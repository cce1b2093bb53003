"""Time-dependent Hamiltonian assembly (system.py:144-215 and
time_evolution_operators/operator.py of the reference), restated from the
reference's tests/test_time_evolution_operators.py.  Host logic only: runs on
CPU with NumPy as the array module (no transform is involved)."""

import numpy as np
import pytest

from quantum_systems_amd import BasisSet, GeneralOrbitalSystem, RandomBasisSet, SpatialOrbitalSystem
from quantum_systems_amd.time_evolution_operators import (
    AdiabaticSwitching, CustomOneBodyOperator, DipoleFieldInteraction, TimeEvolutionOperator,
)


def spin_system(n, l, dim):
    # a spin-carrying, already anti-symmetric basis: GeneralOrbitalSystem then needs no GPU work
    rbs = RandomBasisSet(l, dim)
    bs = BasisSet(l, dim, includes_spin=True, anti_symmetrized_u=True)
    bs.h, bs.s, bs.u, bs.position = rbs.h, rbs.s, rbs.u, rbs.position
    return GeneralOrbitalSystem(n, bs)


@pytest.fixture
def systems():
    np.random.seed(0)
    return SpatialOrbitalSystem(4, RandomBasisSet(10, 3)), spin_system(4, 10, 3)


def test_no_operators(systems):
    # reference tests/test_time_evolution_operators.py:15-47
    for sys_ in systems:
        assert not sys_.has_one_body_time_evolution_operator
        assert not sys_.has_two_body_time_evolution_operator
        np.testing.assert_allclose(sys_.h_t(10), sys_.h)
        np.testing.assert_allclose(sys_.u_t(10), sys_.u)
        sys_.set_time_evolution_operator([], add_h_0=False, add_u_0=False)
        np.testing.assert_allclose(sys_.h_t(0), np.zeros_like(sys_.h))
        np.testing.assert_allclose(sys_.u_t(0), np.zeros_like(sys_.u))


def test_dipole_length_gauge(systems):
    # h(t) = h - E(t) eps . dipole, default polarisation along x
    for sys_ in systems:
        field = lambda t: 0.3 * np.sin(2.0 * t)  # noqa: E731
        sys_.set_time_evolution_operator(DipoleFieldInteraction(field))
        assert sys_.has_one_body_time_evolution_operator
        assert not sys_.has_two_body_time_evolution_operator
        for t in (0.0, 0.4, 1.3):
            np.testing.assert_allclose(sys_.h_t(t), sys_.h - field(t) * sys_.dipole_moment[0])
            np.testing.assert_allclose(sys_.u_t(t), sys_.u)
        pol = np.array([0.0, 1.0, 1.0]) / np.sqrt(2)
        sys_.set_time_evolution_operator(DipoleFieldInteraction(0.7, polarization_vector=pol), add_h_0=False)
        expect = -0.7 * np.tensordot(pol, sys_.dipole_moment, axes=(0, 0))
        np.testing.assert_allclose(sys_.h_t(5.0), expect)


def test_dipole_velocity_gauge(systems):
    spas, _ = systems
    mom = np.random.random((3, 10, 10)) + 1j * np.random.random((3, 10, 10))
    spas._basis_set.momentum = mom
    A = lambda t: 0.2 * t  # noqa: E731
    spas.set_time_evolution_operator(DipoleFieldInteraction(A, gauge="velocity"))
    np.testing.assert_allclose(spas.h_t(2.0), spas.h + A(2.0) * mom[0] + 0.5 * A(2.0) ** 2 * np.eye(10))
    spas.set_time_evolution_operator(DipoleFieldInteraction(A, gauge="velocity", quadratic_term=False))
    np.testing.assert_allclose(spas.h_t(2.0), spas.h + A(2.0) * mom[0])
    with pytest.raises(AssertionError):
        DipoleFieldInteraction(1.0, gauge="coulomb")


def test_adiabatic_switching_and_custom_operator(systems):
    for sys_ in systems:
        ramp = lambda t: 1 - np.exp(-t)  # noqa: E731
        op = np.random.random((10, 10))
        sys_.set_time_evolution_operator(
            [AdiabaticSwitching(ramp), CustomOneBodyOperator(lambda t: t**2, op)], add_u_0=False
        )
        assert sys_.has_one_body_time_evolution_operator and sys_.has_two_body_time_evolution_operator
        np.testing.assert_allclose(sys_.u_t(0.5), ramp(0.5) * sys_.u)
        np.testing.assert_allclose(sys_.h_t(3.0), sys_.h + 9.0 * op)
        # constants are accepted in place of functions of time
        sys_.set_time_evolution_operator([AdiabaticSwitching(0.25), CustomOneBodyOperator(2.0, op)])
        np.testing.assert_allclose(sys_.u_t(1.0), 1.25 * sys_.u)
        np.testing.assert_allclose(sys_.h_t(1.0), sys_.h + 2.0 * op)


def test_base_class_contributes_nothing(systems):
    spas, _ = systems
    spas.set_time_evolution_operator(TimeEvolutionOperator())
    assert not spas.has_one_body_time_evolution_operator
    np.testing.assert_allclose(spas.h_t(1.0), spas.h)
    copy = spas.copy_system()                 # operators survive the deep copy
    assert len(copy._time_evolution_operator) == 1
