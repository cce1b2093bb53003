"""1-D quantum dot (ODQD) and the grid / DVR contraction on the GPU: the product
classes against golden vectors made by the reference's ODQD / ODSincDVR and against the
reference's regression files (tests/test_one_dim_qd.py:127-142 restated)."""

import os

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def H(x):
    from quantum_systems_amd.array_module import to_host

    return np.asarray(to_host(x))


@pytest.mark.parametrize("use_device_module", [False, True])
@pytest.mark.parametrize("tag", ["ho", "dw"])
def test_odqd_matches_reference_class(tag, use_device_module):
    import quantum_systems_amd as qsa

    g = np.load(os.path.join(GOLD, f"odqd_small_{tag}.npz"))
    l, length, n, a, alpha, beta = g["params"]
    pot = qsa.ODQD.HOPotential(1.0) if tag == "ho" else qsa.ODQD.DWPotential(1.0, 5.0)
    kw = dict(np=qsa.hip) if use_device_module else {}
    bs = qsa.ODQD(int(l), length, int(n), a=a, alpha=alpha, beta=beta, potential=pot, **kw)
    assert bs.l == int(l) and bs.dim == 1
    np.testing.assert_allclose(bs.eigen_energies, g["eigen_energies"], rtol=1e-12)
    np.testing.assert_allclose(H(bs.h), g["h"], atol=1e-12)
    np.testing.assert_allclose(H(bs.s), g["s"], atol=0)
    np.testing.assert_allclose(np.abs(H(bs.spf)), np.abs(g["spf"]), atol=1e-10)
    np.testing.assert_allclose(np.abs(H(bs.position)), np.abs(g["position"]), atol=1e-10)
    u = H(bs.u)
    assert u.dtype == np.float64 and H(bs.h).dtype == np.complex128
    np.testing.assert_allclose(np.abs(u), np.abs(g["u"]), atol=1e-10)
    # same eigenvector signs as the oracle on this host: element-wise, no absolute values
    st = orc.odqd_setup(int(l), length, int(n), pot, a=a, alpha=alpha, beta=beta)
    np.testing.assert_allclose(u, st["u"], rtol=1e-11, atol=1e-13)
    # symmetries of a real local interaction
    np.testing.assert_allclose(u, u.transpose(1, 0, 3, 2), atol=1e-12)
    np.testing.assert_allclose(u, u.transpose(2, 1, 0, 3), atol=1e-12)


REGRESSION = {
    "odho": (5, lambda q: q.ODQD.HOPotential(1)),
    "oddw": (6, lambda q: q.ODQD.DWPotential(1, 5)),
    "odgauss": (20, lambda q: q.ODQD.GaussianPotential(1, 0, 2.5, np=np)),
    "oddw_smooth": (5, lambda q: q.ODQD.DWPotentialSmooth(a=5)),
}


@pytest.mark.parametrize("name", list(REGRESSION))
def test_odqd_systems_like_reference(name):
    # tests/test_one_dim_qd.py:127-142 on sampled entries of the reference's own files
    import quantum_systems_amd as qsa

    g = np.load(os.path.join(GOLD, "odqd_reference_regression_files.npz"))
    length, make = REGRESSION[name]
    odqd = qsa.GeneralOrbitalSystem(2, qsa.ODQD(10, length, 1001, potential=make(qsa)))
    assert odqd.l == 20
    np.testing.assert_allclose(np.abs(g[f"{name}_dipole_moment"]), np.abs(H(odqd.position)), atol=1e-9)
    np.testing.assert_allclose(g[f"{name}_h"], H(odqd.h), atol=1e-10)
    u = H(odqd.u)
    ui = g[f"{name}_u_idx"]
    np.testing.assert_allclose(np.abs(g[f"{name}_u_val"]), np.abs(u[tuple(ui.T)]), atol=1e-10)
    np.testing.assert_allclose(g[f"{name}_u_abs_sum"], np.abs(u).sum(), rtol=1e-9)
    spf = H(odqd.spf)
    si = g[f"{name}_spf_idx"]
    np.testing.assert_allclose(np.abs(g[f"{name}_spf_val"]), np.abs(spf[tuple(si.T)]), atol=1e-10)
    # anti-symmetry of the spin-orbital elements (tests/test_one_dim_qd.py:145-172)
    np.testing.assert_allclose(u, -u.transpose(0, 1, 3, 2), atol=1e-8)
    np.testing.assert_allclose(u, -u.transpose(1, 0, 2, 3), atol=1e-8)
    np.testing.assert_allclose(u, u.transpose(1, 0, 3, 2), atol=1e-8)


def test_two_body_from_grid_matches_sinc_dvr_transform():
    from quantum_systems_amd import kernels as K

    g = np.load(os.path.join(GOLD, "sinc_dvr_small.npz"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    u2d, C, Ct = dev(g["u2d"]), dev(g["C"]), dev(g["C_tilde"])
    for got, key in (
        (K.two_body_from_grid(u2d, C), "u_default_bra"),
        (K.two_body_from_grid(u2d, C, Ct), "u_ctilde"),
        (K.two_body_from_grid(u2d, C, Ct, antisymmetrize=True), "u_ctilde_as"),
    ):
        ref = g[key]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("cplx", [False, True])
def test_two_body_from_grid_vs_oracle_larger(cplx):
    from quantum_systems_amd import kernels as K

    rng = np.random.default_rng(3 + cplx)
    N, M = 301, 24
    x = np.linspace(-8, 8, N)
    Kmat = 1.0 / np.sqrt((x[:, None] - x[None, :]) ** 2 + 0.0625)
    C = rng.standard_normal((N, M)) / np.sqrt(N)
    if cplx:
        C = C + 1j * rng.standard_normal((N, M)) / np.sqrt(N)
    ref = orc.two_body_from_grid(Kmat, C)
    got = K.two_body_from_grid(torch.from_numpy(Kmat).cuda(), torch.from_numpy(C).cuda()).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_sinc_dvr_change_basis_like_reference():
    # SpatialOrbitalSystem(n, ODSincDVR).change_basis: the 2-d u becomes the rank-4 tensor of
    # sinc_dvr.py:238-246; golden vectors from the reference class itself
    import quantum_systems_amd as qsa

    g = np.load(os.path.join(GOLD, "sinc_dvr_small.npz"))
    C, Ct = g["C"], g["C_tilde"]
    for bra, key in ((None, "u_default_bra"), (Ct, "u_ctilde")):
        spas = qsa.SpatialOrbitalSystem(2, qsa.ODSincDVR(12, 6.0, potential=qsa.ODSincDVR.HOPotential(0.5)))
        spas.change_basis(C, C_tilde=bra)
        assert spas.l == 7 and H(spas.u).shape == (7, 7, 7, 7)
        ref = g[key]
        assert np.abs(H(spas.u) - ref).max() <= 1e-12 * np.abs(ref).max()
        h_ref = (C.conj().T if bra is None else bra) @ g["h"] @ C
        np.testing.assert_allclose(H(spas.h), h_ref, rtol=1e-12, atol=1e-12)
    dvr = qsa.ODSincDVR(12, 6.0, potential=qsa.ODSincDVR.HOPotential(0.5))
    got = dvr.transform_two_body_elements(dvr.u, C, np, anti_symmetrize=True, C_tilde=Ct)
    ref = g["u_ctilde_as"]
    assert np.abs(H(got) - ref).max() <= 1e-12 * np.abs(ref).max()
    # the 4-d representation goes through the ordinary four-index transform and agrees
    d4 = qsa.ODSincDVR(12, 6.0, potential=qsa.ODSincDVR.HOPotential(0.5), u_repr="4d")
    got4 = d4.transform_two_body_elements(d4.u, C, np, C_tilde=Ct)
    assert np.abs(H(got4) - g["u_ctilde"]).max() <= 1e-12 * np.abs(g["u_ctilde"]).max()
    with pytest.raises(AssertionError):
        d4.transform_two_body_elements(d4.u, C, np, anti_symmetrize=True)


def test_sinc_dvr_spin_doubling_of_the_2d_form_like_reference():
    # The only upstream route to a spin DVR basis: ODSincDVR(2-d u).change_to_general_orbital_basis goes
    # through the class's OWN add_spin_two_body (kron(K, ones(2, 2))) and anti_symmetrize_u (identity on a
    # matrix), reached via self as basis_set.py:576 / :523 do -- not through the rank-4 kernels.  Golden
    # vectors from the reference class (tests/golden/make_golden.py: case_sinc_dvr_spin)
    import quantum_systems_amd as qsa

    g = np.load(os.path.join(GOLD, "sinc_dvr_spin_doubling.npz"))
    for tag, anti in (("as", True), ("noas", False)):
        dvr = qsa.ODSincDVR(6, 4.0, potential=qsa.ODSincDVR.HOPotential(0.5))
        assert H(dvr.u).shape == (6, 6)
        assert dvr.change_to_general_orbital_basis(anti_symmetrize=anti) is dvr
        assert dvr.l == 12 and H(dvr.u).shape == (12, 12) and H(dvr.u).dtype == np.complex128
        assert np.array_equal(H(dvr.u), g[tag + "_u"])                  # kron(K, ones(2,2)), never K - K^T
        for k in ("h", "s", "position", "spf"):
            np.testing.assert_allclose(H(getattr(dvr, k)), g[tag + "_" + k], rtol=1e-13, atol=1e-13)
        flags = [dvr.includes_spin, dvr.anti_symmetrized_u, dvr.spin_2_tb is None]
        assert flags == list(g[tag + "_flags"])
    # a spin-carrying DVR basis handed the driver directly: the 2-d interaction is left alone (upstream
    # anti_symmetrize_u returns the matrix unchanged), not replaced by K - K^T = 0
    dvr = qsa.ODSincDVR(6, 4.0, potential=qsa.ODSincDVR.HOPotential(0.5), includes_spin=True)
    assert np.array_equal(H(dvr.u), g["spin_ctor_u_before"])
    dvr.anti_symmetrize_two_body_elements()
    assert dvr.anti_symmetrized_u and np.array_equal(H(dvr.u), g["spin_ctor_u_after"])
    assert np.abs(H(dvr.u)).max() > 0
    # the 4-d representation takes the same route through the class's statics and agrees with the oracle
    d4 = qsa.ODSincDVR(6, 4.0, potential=qsa.ODSincDVR.HOPotential(0.5), u_repr="4d")
    u4 = H(d4.u).copy()
    d4.change_to_general_orbital_basis(anti_symmetrize=True)
    ref = orc.anti_symmetrize_u(orc.add_spin_two_body(u4)).astype(np.complex128)
    assert np.array_equal(H(d4.u), ref) and d4.anti_symmetrized_u


def test_spin_squared_like_reference():
    # tests/test_spin.py:136-177 (the part of the reference test that is active): singlet / triplet
    # expectation values of the two-spin S^2 built from the spinors stored on the spin-doubled ODQD basis;
    # plus the basis set's one- and two-body spin operators against the oracle's for the same basis
    import quantum_systems_amd as qsa

    spas = qsa.SpatialOrbitalSystem(2, qsa.ODQD(2, 8, 1001, potential=qsa.ODQD.HOPotential(1)))
    gos = spas.construct_general_orbital_system(a=[1, 0], b=[0, 1])
    a, b = H(gos._basis_set.a), H(gos._basis_set.b)
    aa, ab, ba, bb = np.kron(a, a), np.kron(a, b), np.kron(b, a), np.kron(b, b)
    S_sq_spin = np.zeros((4, 4))
    S_sq_spin[0, 0] = S_sq_spin[3, 3] = 2
    S_sq_spin[1, 1] = S_sq_spin[2, 2] = S_sq_spin[1, 2] = S_sq_spin[2, 1] = 1
    for trip in (aa, (ab + ba) / np.sqrt(2), bb):
        np.testing.assert_allclose(trip.T @ S_sq_spin @ trip, 2)
    singlet = (ab - ba) / np.sqrt(2)
    np.testing.assert_allclose(singlet.T @ S_sq_spin @ singlet, 0, atol=1e-15)

    st = orc.new_state(2, 1)
    for k in ("h", "s", "u", "spf", "position"):
        st[k] = H(getattr(spas, k))
    ref = orc.change_to_general_orbital_basis(st, a=[1, 0], b=[0, 1])
    for name in ("spin_x", "spin_y", "spin_z", "spin_2", "spin_2_tb"):
        np.testing.assert_allclose(H(getattr(gos, name)), ref[name], rtol=1e-12, atol=1e-13)
