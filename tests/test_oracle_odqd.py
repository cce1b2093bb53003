"""Oracle for the grid / DVR contractions against golden vectors made by the
reference's own ODQD and ODSincDVR classes and against the reference's regression files."""

import os

import numpy as np
import pytest

from oracle import qs_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

POTENTIALS = {
    "ho": lambda x: 0.5 * x**2,
    "dw": lambda x: 0.5 * x**2 + 0.5 * (0.25 * 25.0 - 5.0 * abs(x)),
}


@pytest.mark.parametrize("tag", ["ho", "dw"])
def test_odqd_setup_matches_reference_class(tag):
    g = np.load(os.path.join(GOLD, f"odqd_small_{tag}.npz"))
    l, length, n, a, alpha, beta = g["params"]
    st = orc.odqd_setup(int(l), length, int(n), POTENTIALS[tag], a=a, alpha=alpha, beta=beta)
    np.testing.assert_allclose(st["grid"], g["grid"], atol=0)
    np.testing.assert_allclose(st["eigen_energies"], g["eigen_energies"], rtol=1e-12)
    np.testing.assert_allclose(st["h"], g["h"], atol=1e-12)
    np.testing.assert_allclose(st["s"], g["s"], atol=0)
    # eigenvectors are defined up to a sign: compare as the reference's own test does
    np.testing.assert_allclose(np.abs(st["spf"]), np.abs(g["spf"]), atol=1e-10)
    np.testing.assert_allclose(np.abs(st["u"]), np.abs(g["u"]), atol=1e-10)
    np.testing.assert_allclose(np.abs(st["position"]), np.abs(g["position"]), atol=1e-10)
    assert st["u"].dtype == g["u"].dtype == np.float64 and st["h"].dtype == np.complex128


def test_two_body_from_grid_matches_sinc_dvr_transform():
    g = np.load(os.path.join(GOLD, "sinc_dvr_small.npz"))
    K, C, Ct = g["u2d"], g["C"], g["C_tilde"]
    np.testing.assert_allclose(orc.two_body_from_grid(K, C), g["u_default_bra"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(orc.two_body_from_grid(K, C, Ct), g["u_ctilde"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(orc.two_body_from_grid(K, C, Ct, antisymmetrize=True), g["u_ctilde_as"],
                               rtol=1e-12, atol=1e-12)
    # the 2-d form is the diagonal of the 4-d one: both routes of the reference agree
    l = K.shape[0]
    u4 = np.zeros((l,) * 4, dtype=K.dtype)
    idx = np.arange(l)
    u4[idx[:, None], idx[None, :], idx[:, None], idx[None, :]] = K
    np.testing.assert_allclose(orc.transform_two_body(u4, C, Ct), g["u_ctilde"], rtol=1e-12, atol=1e-12)


REGRESSION = {
    # fixtures of tests/test_one_dim_qd.py:8-118: GeneralOrbitalSystem(2, ODQD(10, length, 1001, potential))
    "odho": (5, lambda x: 0.5 * x**2),
    "oddw": (6, lambda x: 0.5 * x**2 + 0.5 * (0.25 * 25.0 - 5.0 * abs(x))),
    "odgauss": (20, lambda x: -np.exp(-(x**2) / (2.0 * 2.5**2))),
    "oddw_smooth": (5, lambda x: (x + 2.5) ** 2 * (x - 2.5) ** 2 / 50.0),
}


@pytest.mark.parametrize("name", list(REGRESSION))
def test_oracle_reproduces_the_reference_regression_files(name):
    g = np.load(os.path.join(GOLD, "odqd_reference_regression_files.npz"))
    length, pot = REGRESSION[name]
    st = orc.odqd_setup(10, length, 1001, pot)
    state = orc.new_state(10, 1)
    for k in ("h", "s", "u", "spf", "position"):
        state[k] = st[k]
    gos = orc.change_to_general_orbital_basis(state)
    np.testing.assert_allclose(np.abs(g[f"{name}_dipole_moment"]), np.abs(gos["position"]), atol=1e-9)
    np.testing.assert_allclose(g[f"{name}_h"], gos["h"], atol=1e-10)
    ui = g[f"{name}_u_idx"]
    assert tuple(g[f"{name}_u_shape"]) == gos["u"].shape
    np.testing.assert_allclose(np.abs(g[f"{name}_u_val"]), np.abs(gos["u"][tuple(ui.T)]), atol=1e-10)
    np.testing.assert_allclose(g[f"{name}_u_abs_sum"], np.abs(gos["u"]).sum(), rtol=1e-9)
    si = g[f"{name}_spf_idx"]
    assert tuple(g[f"{name}_spf_shape"]) == gos["spf"].shape
    np.testing.assert_allclose(np.abs(g[f"{name}_spf_val"]), np.abs(gos["spf"][tuple(si.T)]), atol=1e-10)
    np.testing.assert_allclose(g[f"{name}_spf_abs_sum"], np.abs(gos["spf"]).sum(), rtol=1e-9)
