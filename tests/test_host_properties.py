"""Property tests of the host-side logic (hypothesis): partitions, orbital index maps, level tables,
potentials.  CPU only."""

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from quantum_systems_amd import one_dim_potentials as pot
from quantum_systems_amd import two_dim_ho as td
from quantum_systems_amd.sharded import SlabPartition


@given(st.integers(1, 5000), st.integers(1, 64))
@settings(max_examples=200, deadline=None)
def test_slab_partition_is_a_balanced_cover(n, world):
    part = SlabPartition(n, world)
    bounds = [part.bounds(r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    counts = [hi - lo for lo, hi in bounds]
    assert min(counts) >= 0 and max(counts) - min(counts) <= 1 and sum(counts) == n


@given(st.integers(0, 5000))
@settings(max_examples=300, deadline=None)
def test_orbital_index_map_is_a_bijection_over_shells(p):
    # two_dim_helper.py:111-166: p <-> (n, m), shells of 1, 2, 3, ... states ordered by m
    n, m = td.get_indices_nm(p)
    assert n >= 0 and td.get_index_p(n, m) == p
    shell = 2 * n + abs(m) + 1
    assert shell * (shell - 1) // 2 <= p < shell * (shell + 1) // 2
    if p:
        n0, m0 = td.get_indices_nm(p - 1)
        s0 = 2 * n0 + abs(m0) + 1
        assert (s0, m0) < (shell, m)                     # ordered by shell, then by m


@given(st.integers(1, 12), st.floats(0.0, 3.0), st.floats(0.2, 3.0))
@settings(max_examples=60, deadline=None)
def test_level_table_is_sorted_and_complete(l, omega_c, omega):
    n_array, m_array = np.arange(l), np.arange(-l - 5, l + 6)
    nm, E = td.construct_level_table(n_array, m_array, omega_c=omega_c, omega=omega)
    assert len(E) == len(n_array) * len(m_array) == len({tuple(r) for r in nm})
    assert np.all(np.diff(E) >= 0)
    ties = np.diff(E) == 0
    assert np.all(np.diff(nm[:, 1])[ties] >= 0)            # equal energies: by m
    np.testing.assert_allclose(E, td.get_shell_energy_B(nm[:, 0], nm[:, 1], omega_c=omega_c, omega=omega))


@given(st.floats(-3.0, 3.0), st.floats(0.3, 2.0))
@settings(max_examples=100, deadline=None)
def test_potential_derivatives_match_finite_differences(x, w):
    h = 1e-6
    cases = [pot.HOPotential(w), pot.DWPotentialSmooth(a=2 + w), pot.AsymmetricDWPotential(),
             pot.GaussianPotential(w, 0.1, 1.5, np), pot.AtomicPotential()]
    if abs(x) > 1e-3:
        cases.append(pot.DWPotential(w, 2.0))              # the cusp at 0 has no derivative
    for V in cases:
        fd = (V(x + h) - V(x - h)) / (2 * h)
        assert abs(V.derivative(x) - fd) <= 1e-5 * max(1.0, abs(fd)), type(V).__name__
    # the reference's SymmetricDWPotential.derivative uses 3 b x^3 for the quartic term; mirrored as is
    V = pot.SymmetricDWPotential()
    assert np.isclose(V.derivative(x), 6 * V.a * x**5 + 3 * V.b * x**3 + 2 * V.c * x)


@given(st.integers(0, 6), st.integers(-6, 6), st.integers(0, 6), st.integers(-6, 6), st.sampled_from([1, 2, 4]))
@settings(max_examples=80, deadline=None)
def test_radial_integral_is_symmetric_and_matches_quadrature(n_p, m_p, n_q, m_q, order):
    a = td.radial_integral(n_p, m_p, n_q, m_q, 1.0, 0.9, order=order)
    b = td.radial_integral(n_q, m_q, n_p, m_p, 1.0, 0.9, order=order)
    assert np.isclose(a, b, rtol=1e-12, atol=1e-14)
    r = np.linspace(0, 40, 400001)
    f = r ** (1 + order) * td.spf_radial(r, n_p, m_p, 1.0, 0.9) * td.spf_radial(r, n_q, m_q, 1.0, 0.9)
    # (the quadrature's own rounding grows with the size of the integrand: a selection-rule zero of the closed
    # form comes out of it as ~1e-9 when int |f| ~ 1e3 -- found by this very test in round 2)
    np.testing.assert_allclose(a, np.trapezoid(f, r), rtol=1e-6, atol=1e-9 + 1e-11 * np.trapezoid(np.abs(f), r))
