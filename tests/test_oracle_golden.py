"""Pin the CPU oracle (oracle/qs_oracle.py) to the reference.

Two kinds of pin:
* the committed golden vectors, produced by the reference's own code
  (tests/golden/make_golden.py);
* restatements of the reference's analytic tests for this path
  (reference tests/test_helper.py:14-147, tests/test_custom_system.py:27-68),
  with the same tolerances.
CPU only; no GPU, no HIP library.
"""

import numpy as np
import pytest

from oracle import qs_oracle as orc

TIGHT = dict(rtol=1e-13, atol=1e-13)


def eq_exact(a, b):
    """Value equality (-0.0 == +0.0), shape and dtype kind included."""
    assert a.shape == b.shape
    assert np.array_equal(a, b)


# ------------------------------------------------------------------ golden


@pytest.mark.parametrize(
    "name",
    [
        "transform_c128_square",
        "transform_c128_rect_ctilde",
        "transform_c128_shrink",
        "transform_f64_orthogonal",
        "transform_f64_rect",
        "transform_mixed_real_u_complex_C",
    ],
)
def test_transforms_match_reference(golden, name):
    g = golden(name)
    Ct = g.get("C_tilde")
    u = orc.transform_two_body(g["u"], g["C"], Ct)
    h = orc.transform_one_body(g["h"], g["C"], Ct)
    assert u.dtype == g["u_out"].dtype and h.dtype == g["h_out"].dtype
    # same NumPy calls in the same order -> bitwise equal on the same machine;
    # keep a hair of slack for a different BLAS build.
    np.testing.assert_allclose(u, g["u_out"], **TIGHT)
    np.testing.assert_allclose(h, g["h_out"], **TIGHT)
    np.testing.assert_allclose(
        orc.transform_two_body_einsum(g["u"], g["C"], Ct), g["u_out"],
        rtol=1e-11, atol=1e-11,
    )
    # the sampled form (what the l = 256 GPU test compares against) reproduces the reference's output too
    M = g["u_out"].shape[0]
    pairs = [(0, 0), (M - 1, 0), (M // 2, M - 1), (1 % M, 2 % M)]
    np.testing.assert_allclose(
        orc.transform_two_body_pq_samples(g["u"], g["C"], Ct, pairs),
        np.stack([g["u_out"][p, q] for (p, q) in pairs]), rtol=1e-11, atol=1e-11,
    )


def test_spf_transforms_match_reference(golden):
    g = golden("transform_spf")
    np.testing.assert_allclose(orc.transform_spf(g["spf"], g["C"]), g["spf_out"], **TIGHT)
    np.testing.assert_allclose(
        orc.transform_bra_spf(g["bra_spf"], g["C_tilde"]), g["bra_out"], **TIGHT
    )


@pytest.mark.parametrize("name", ["spin_statics_f64", "spin_statics_c128"])
def test_spin_statics_match_reference(golden, name):
    g = golden(name)
    us = orc.add_spin_two_body(g["u"])
    eq_exact(us, g["u_spin"])
    # signed zeros are reproduced too: bitwise identical
    assert us.tobytes() == g["u_spin"].tobytes()
    eq_exact(orc.anti_symmetrize_u(us), g["u_spin_as"])
    eq_exact(orc.anti_symmetrize_u(g["u"]), g["u_as"])
    eq_exact(orc.spin_two_body_index_law(g["u"]), g["u_spin_as"])
    if "h" in g:
        eq_exact(orc.add_spin_one_body(g["h"]), g["h_spin"])
        eq_exact(orc.add_spin_spf(g["spf"]), g["spf_spin"])


def test_random_basis_draw_order(golden):
    g = golden("random_basis_seed1234_l4_dim3")
    np.random.seed(1234)
    st = orc.random_basis(4, 3)
    for k in ("h", "s", "u", "position"):
        eq_exact(st[k], g[k])
    assert st["nuclear_repulsion_energy"] == float(g["nuclear_repulsion_energy"])
    assert st["charge"] == int(g["charge"])


def test_config1_change_basis(golden):
    g = golden("config1_l20_change_basis")
    np.random.seed(int(g["seed"]))
    st = orc.random_basis(20, 2)
    A = np.random.random((20, 20)) + 1j * np.random.random((20, 20))
    C, _ = np.linalg.qr(A)
    np.testing.assert_allclose(C, g["C"], **TIGHT)
    eq_exact(st["u"][::3, 1::4, 2::5, ::2], g["u_in_sample"])
    orc.change_basis(st, C)
    assert st["l"] == int(g["l"])
    for k in ("h", "s", "u", "position"):
        np.testing.assert_allclose(st[k], g[k], rtol=1e-12, atol=1e-12)
    # and the spin doubling of the transformed system
    s = golden("config1_l20_gos_sampled")
    orc.change_to_general_orbital_basis(st)
    assert st["l"] == int(s["l"]) == 40
    p, q, r, t = s["idx"].T
    np.testing.assert_allclose(st["u"][p, q, r, t], s["u_samples"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(
        st["spin_2_tb"][p, q, r, t], s["spin_2_tb_samples"], rtol=1e-12, atol=1e-12
    )
    np.testing.assert_allclose(np.linalg.norm(st["u"]), s["u_fro"], rtol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(st["spin_2_tb"]), s["spin_2_tb_fro"], rtol=1e-12)
    np.testing.assert_allclose(st["u"][3, :8], s["u_slab_p3"], rtol=1e-12, atol=1e-12)
    for k in ("h", "s", "spin_x", "spin_y", "spin_z", "spin_2", "position"):
        np.testing.assert_allclose(st[k], s[k], rtol=1e-12, atol=1e-12)


def _state_from(g, prefix, l, dim, **flags):
    st = orc.new_state(l, dim, **flags)
    for k in orc._FIELDS:
        if prefix + k in g:
            st[k] = g[prefix + k].copy()
    return st


def test_gos_default_spinors_and_spin_change_basis(golden):
    g = golden("gos_l5_default_spinors")
    st = _state_from(g, "in_", 5, 2)
    orc.change_to_general_orbital_basis(st)
    assert st["l"] == 10
    for k in ("h", "s", "u", "position", "spin_x", "spin_y", "spin_z", "spin_2", "spin_2_tb"):
        assert st[k].dtype == np.complex128
        eq_exact(st[k], g["gos_" + k])
    for k in ("sigma_x", "sigma_y", "sigma_z"):
        eq_exact(st[k], g["gos_" + k])
    # rectangular change of basis on the spin basis
    orc.change_basis(st, g["C"])
    assert st["l"] == int(g["l_after"]) == 8
    for k in ("h", "s", "u", "position", "spin_2_tb"):
        np.testing.assert_allclose(st[k], g["cb_" + k], rtol=1e-12, atol=1e-12)
    # quirk 0.6: spin one-body operators keep their OLD shape and values
    for k in ("spin_x", "spin_y", "spin_z", "spin_2"):
        assert st[k].shape == (10, 10)
        eq_exact(st[k], g["cb_" + k])


def test_gos_custom_spinors_without_antisymmetrisation(golden):
    g = golden("gos_l4_custom_spinors_no_as")
    st = _state_from(g, "in_", 4, 1)
    orc.change_to_general_orbital_basis(st, a=g["a"], b=g["b"], anti_symmetrize=False)
    assert not st["anti_symmetrized_u"]
    for k in ("h", "s", "u", "position", "momentum", "spf", "spin_x", "spin_y",
              "spin_z", "spin_2", "spin_2_tb", "sigma_x", "sigma_y", "sigma_z"):
        np.testing.assert_allclose(st[k], g["gos_" + k], **TIGHT)
    eq_exact(st["u"], g["gos_u"])


def test_change_basis_with_spf_and_ctilde(golden):
    g = golden("change_basis_l5_to_7_spf_ctilde")
    st = _state_from(g, "in_", 5, 2)
    orc.change_basis(st, g["C"], g["C_tilde"])
    assert st["l"] == 7
    for k in ("h", "s", "u", "position", "momentum", "spf", "bra_spf"):
        np.testing.assert_allclose(st[k], g["out_" + k], **TIGHT)


# ------------------------------------------- the reference's analytic tests


def test_one_body_vs_einsum():
    # reference tests/test_helper.py:14-35 (l=10, complex, atol 1e-10)
    rng = np.random.default_rng(0)
    l = 10
    h = rng.random((l, l)) + 1j * rng.random((l, l))
    C = rng.random((l, l)) + 1j * rng.random((l, l))
    ref = np.einsum("ip, jq, ij", C.conj(), C, h, optimize=True)
    np.testing.assert_allclose(ref, orc.transform_one_body(h, C), atol=1e-10)
    np.testing.assert_allclose(
        orc.transform_one_body(h, C), orc.transform_one_body(h, C, C.conj().T)
    )


def test_two_body_vs_einsum():
    # reference tests/test_helper.py:38-69
    rng = np.random.default_rng(1)
    l = 10
    u = rng.random((l, l, l, l)) + 1j * rng.random((l, l, l, l))
    C = rng.random((l, l)) + 1j * rng.random((l, l))
    ref = np.einsum("ls, kr, jq, ip, ijkl -> pqrs", C, C, C.conj(), C.conj(), u, optimize=True)
    np.testing.assert_allclose(ref, orc.transform_two_body(u, C), atol=1e-10)
    np.testing.assert_allclose(
        orc.transform_two_body(u, C), orc.transform_two_body(u, C, C.conj().T)
    )


def test_spin_delta_law():
    # reference tests/test_helper.py:6-11
    for p in range(40):
        for q in range(40):
            assert orc.spin_delta(p, q) == ((p % 2) == (q % 2))


def test_antisymmetry_properties():
    # reference tests/test_helper.py:138-147
    rng = np.random.default_rng(2)
    u = rng.random((6, 6, 6, 6))
    u = u + u.transpose(1, 0, 3, 2)
    u = orc.anti_symmetrize_u(orc.add_spin_two_body(u))
    np.testing.assert_allclose(u, -u.transpose(0, 1, 3, 2), atol=1e-10)
    np.testing.assert_allclose(u, -u.transpose(1, 0, 2, 3), atol=1e-10)
    np.testing.assert_allclose(u, u.transpose(1, 0, 3, 2), atol=1e-10)


def test_rectangular_change_of_basis_like_reference():
    # reference tests/test_custom_system.py:38-68 (atol = rtol = 1e-12)
    np.random.seed(3)
    n, l, dim = 2, 10, 2
    new_l = 2 * l - n
    st = orc.random_basis(l, dim)
    C = np.random.random((l, new_l)) + 1j * np.random.random((l, new_l))
    h_ref = np.einsum("ap,bq,ab->pq", C.conj(), C, st["h"], optimize=True)
    u_ref = np.einsum(
        "ap,bq,gr,ds,abgd->pqrs", C.conj(), C.conj(), C, C, st["u"], optimize=True
    )
    orc.change_basis(st, C)
    assert st["l"] == new_l and st["u"].shape == (new_l,) * 4
    np.testing.assert_allclose(h_ref, st["h"], atol=1e-12, rtol=1e-12)
    np.testing.assert_allclose(u_ref, st["u"], atol=1e-12, rtol=1e-12)


# ------------------------------------------------------------------ f2: Fock matrix / reference energy


def _fock_state(g, prefix=""):
    l = int(g[prefix + "l"])
    st = orc.new_state(l, 2)
    st["h"], st["u"] = g[prefix + "h"].copy(), g[prefix + "u"].copy()
    st["s"] = g["s"].copy() if not prefix else np.eye(l, dtype=np.complex128)
    st["nuclear_repulsion_energy"] = float(g[prefix + "e_nuc"])
    return st


def test_fock_and_energy_match_reference_classes(golden):
    # SpatialOrbitalSystem / GeneralOrbitalSystem of the reference on a seeded RandomBasisSet
    # (spatial_orbital_system.py:106-190, general_orbital_system.py:75-159), before and after change_basis
    g = golden("fock_energy_random_basis")
    n = int(g["n"]) // 2          # n particles -> n // 2 doubly occupied spatial orbitals (spatial_orbital_system.py:50)
    st = _fock_state(g)
    e = orc.reference_energy(st["h"], st["u"], n, st["nuclear_repulsion_energy"])
    np.testing.assert_allclose(e, g["spas_energy"], **TIGHT)
    np.testing.assert_allclose(orc.fock_matrix(st["h"], st["u"], n), g["spas_fock"], **TIGHT)
    o = slice(0, n)
    np.testing.assert_allclose(
        orc.reference_energy(st["h"][o, o], st["u"][o, o, o, o], n, st["nuclear_repulsion_energy"]),
        g["spas_energy_occ_block"], **TIGHT)
    gos = orc.change_to_general_orbital_basis(_fock_state(g))
    np.testing.assert_allclose(gos["h"], g["gos_h"], **TIGHT)
    np.testing.assert_allclose(
        orc.reference_energy(gos["h"], gos["u"], 2 * n, gos["nuclear_repulsion_energy"], spin_orbitals=True),
        g["gos_energy"], **TIGHT)
    np.testing.assert_allclose(orc.fock_matrix(gos["h"], gos["u"], 2 * n, spin_orbitals=True), g["gos_fock"], **TIGHT)
    orc.change_basis(st, g["C"])
    np.testing.assert_allclose(
        orc.reference_energy(st["h"], st["u"], n, st["nuclear_repulsion_energy"]), g["spas_cb_energy"], **TIGHT)
    np.testing.assert_allclose(orc.fock_matrix(st["h"], st["u"], n), g["spas_cb_fock"], **TIGHT)
    orc.change_basis(gos, g["C_gos"])
    np.testing.assert_allclose(
        orc.reference_energy(gos["h"], gos["u"], 2 * n, gos["nuclear_repulsion_energy"], spin_orbitals=True),
        g["gos_cb_energy"], **TIGHT)
    np.testing.assert_allclose(orc.fock_matrix(gos["h"], gos["u"], 2 * n, spin_orbitals=True), g["gos_cb_fock"], **TIGHT)
    # second case: rectangular (shrinking) change of basis
    sb = _fock_state(g, "b_")
    orc.change_basis(sb, g["b_C"])
    nb = int(g["b_n"]) // 2
    np.testing.assert_allclose(
        orc.reference_energy(sb["h"], sb["u"], nb, sb["nuclear_repulsion_energy"]), g["b_cb_energy"], **TIGHT)
    np.testing.assert_allclose(orc.fock_matrix(sb["h"], sb["u"], nb), g["b_cb_fock"], **TIGHT)


# ------------------------------------------------------------------ sizes beyond the committed tensors


def test_formula_inputs_are_the_same_from_numpy_and_torch():
    import torch

    import _lattice_inputs as li

    for cplx in (False, True):
        a = li.tensor_np(11, 5, cplx)
        b = li.tensor_torch(11, 5, cplx).numpy()
        assert a.dtype == b.dtype and np.array_equal(a, b)
        k = a.real * 32768                                  # every value is an integer / 32768: nothing to round anywhere
        assert np.array_equal(k, np.round(k)) and np.abs(k).max() <= 32768
    assert len({tuple(p) for p in li.sample_positions(100, li.N_SAMPLES, 2)}) > li.N_SAMPLES - 8


@pytest.mark.parametrize("name", ["f64_78", "f64_100", "c128_72"])
def test_oracle_matches_reference_samples_at_mid_size(golden, name):
    # tests/golden/mid_size_sampled.npz: the REFERENCE's transform of formula inputs at 78 ... 180 orbitals, sampled; the
    # oracle is checked on the cases that take it seconds (the GPU tests check the HIP path on all of them)
    import _lattice_inputs as li

    g = golden("mid_size_sampled")
    _, L, M, ucplx, ccplx, salt = next(c for c in li.CASES if c[0] == name)
    C, Ct = li.case_inputs_np(L, M, ucplx, ccplx, salt)
    out = orc.transform_two_body(li.tensor_np(L, salt, ucplx), C, Ct)
    pos = g[name + "_pos"]
    assert np.array_equal(pos, li.sample_positions(M, li.N_SAMPLES, salt))
    got = out[pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3]]
    scale = float(g[name + "_max_abs"])
    assert got.dtype == g[name + "_val"].dtype
    assert np.abs(got - g[name + "_val"]).max() <= 1e-12 * scale
    assert abs(out.sum() - g[name + "_sum"]) <= 1e-12 * float(g[name + "_abs_sum"])
    assert abs(np.abs(out).sum() - g[name + "_abs_sum"]) <= 1e-12 * float(g[name + "_abs_sum"])
