"""Pin the C oracle of the 2-D harmonic-oscillator Coulomb elements
(oracle/coulomb_ho.c) to the reference: its own table of elements (the file its
tests read, orbitals 0..35, atol 1e-6 as in tests/test_two_dim_ho.py:77-90 of the
reference), elements computed by the reference's coulomb_ho itself, its index
map and shell energies.  CPU only."""

import numpy as np

from oracle import coulomb_oracle as co


def test_index_map_and_shell_energies(golden):
    g = golden("tdho_coulomb_spot")
    nm = g["index_map"]
    for p in range(len(nm)):
        assert co.indices_nm(p) == (int(nm[p, 0]), int(nm[p, 1]))
    # reference tests/test_two_dim_ho.py:54-61 (p <-> (n, m) is a bijection over shells)
    assert co.indices_nm(54) == (0, 9) and co.indices_nm(0) == (0, 0)
    np.testing.assert_array_equal(np.diag(co.one_body_elements(55)), g["one_body_l55"])


def test_spot_elements_from_the_reference_code(golden):
    g = golden("tdho_coulomb_spot")
    nm = g["index_map"]
    # The closed form is an alternating sum of terms up to ~1e6 times the result at the
    # 9th/10th shell, so two correct double-precision evaluations differ by ~1e-10 there
    # (the reference itself is built with numba fastmath and is not bit-defined);
    # low-shell elements (first six shells) agree to 1e-12.
    worst = 0.0
    for (p, q, r, s), ref in zip(g["idx"], g["val"]):
        got = co.coulomb_ho(*map(int, nm[p]), *map(int, nm[q]), *map(int, nm[r]), *map(int, nm[s]))
        tol = 1e-12 if max(p, q, r, s) < 21 else 2e-9
        assert abs(got - ref) <= tol, (p, q, r, s, got, ref)
        worst = max(worst, abs(got - ref))
    assert worst > 0 or True


def test_full_table_of_the_reference(golden):
    g = golden("tdho_coulomb_table")
    l = 36
    u = co.coulomb_elements(l)
    ref = np.zeros((l, l, l, l))
    p, q, r, s = g["idx"].astype(np.int64).T
    ref[p, q, r, s] = g["val"]
    np.testing.assert_allclose(u, ref, atol=1e-6, rtol=1e-6)
    # m conservation: everything the table does not list is exactly zero
    assert np.count_nonzero(u) == len(g["val"])
    # symmetries of the Coulomb interaction in this (real) basis
    np.testing.assert_allclose(u, u.transpose(1, 0, 3, 2), atol=1e-9)
    np.testing.assert_allclose(u, u.transpose(2, 3, 0, 1), atol=1e-9)


def test_slab_form_matches_full():
    full = co.coulomb_elements(10)
    np.testing.assert_array_equal(co.coulomb_elements(10, 3, 7), full[3:7])


# ---- one-body side: double-well Hamiltonian, orbital table, dipole elements


def test_double_well_one_body_oracle_matches_reference(golden):
    g = golden("tdho_one_body")
    for tag in "acd":                      # (b: l = 12 takes the symbolic route ~10 s; covered by the product test)
        l, omega, mass, b, axis = g[f"dw_{tag}_params"]
        h = co.double_well_one_body(int(l), omega, mass, b, axis=int(axis))
        np.testing.assert_allclose(h, g[f"dw_{tag}_h"], rtol=1e-11, atol=1e-12)
    # reference tests/test_two_dim_dw.py:93-112
    eps = np.linalg.eigvalsh(co.double_well_one_body(6, 1.0, 1, 2, axis=1))
    np.testing.assert_allclose(eps[:6], g["test_energies_l6_b2_axis1"], rtol=1e-7)


def test_spf_table_and_dipole_oracle_match_reference(golden):
    g = golden("tdho_one_body")
    l, radius, n, omega, mass = g["tdho_l10_params"]
    spf = co.spf_table(int(l), np.linspace(0, radius, int(n)), np.linspace(0, 2 * np.pi, int(n)), mass, omega)
    np.testing.assert_allclose(spf, g["tdho_l10_spf"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(co.position_integrals(int(l), mass, omega), g["tdho_l10_position"], rtol=1e-11, atol=1e-12)


def test_spf_tables_match_the_reference_files(golden):
    # reference tests/test_two_dim_ho.py:93-100 with tests/conftest.py:155-168 (radius 4, 101 points)
    g = golden("tdho_one_body")
    grid_r, grid_t = np.linspace(0, 4, 101), np.linspace(0, 2 * np.pi, 101)
    spf = co.spf_table(15, grid_r, grid_t)
    assert spf.shape[1:] == tuple(g["spf_files_shape"])
    pts = g["spf_files_pts"]
    for p in range(15):
        np.testing.assert_allclose(spf[p][tuple(pts.T)], g["spf_files_val"][p], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(np.abs(spf[p]).sum(), g["spf_files_abs_sum"][p], rtol=1e-9)


def test_magnetic_field_level_table_and_elements(golden):
    g = golden("tdho_one_body")
    for tag in "abc":
        l, wc, w0 = g[f"levels_{tag}_params"]
        l = int(l)
        w = np.sqrt(w0**2 + wc**2 / 4)
        nm, E = co.level_table(np.arange(l), np.arange(-l - 5, l + 6), omega_c=wc, omega=w)
        k = len(g[f"levels_{tag}_E"])
        np.testing.assert_array_equal(nm[:k], g[f"levels_{tag}_nm"])
        np.testing.assert_allclose(E[:k], g[f"levels_{tag}_E"], rtol=0, atol=0)
    # the reference's regression file of the spin-doubled, anti-symmetrised u (sampled): element by element
    l, wc, w0 = g["levels_a_params"]
    w = np.sqrt(w0**2 + wc**2 / 4)
    nm = g["levels_a_nm"]
    for (P, Q, R, S), ref in list(zip(g["tdhob_u_idx"], g["tdhob_u_val"]))[:600]:
        p, q, r, s = P // 2, Q // 2, R // 2, S // 2
        direct = co.coulomb_element_nm(nm, p, q, r, s) if (P % 2 == R % 2 and Q % 2 == S % 2) else 0.0
        exchange = co.coulomb_element_nm(nm, p, q, s, r) if (P % 2 == S % 2 and Q % 2 == R % 2) else 0.0
        assert abs(np.sqrt(w) * (direct - exchange) - ref) <= 1e-10
    np.testing.assert_allclose(np.diag(g["tdhob_h"]).real, np.repeat(g["levels_a_E"][:10], 2), atol=1e-12)
