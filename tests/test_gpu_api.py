"""API-level parity on the GPU: the reference's own system-level tests restated
against this package (tests/test_custom_system.py, test_spin.py, test_copy.py of
the reference) and the golden vectors produced by the reference's
``BasisSet`` / ``QuantumSystem`` classes.  Each case runs with NumPy as the
array module (inputs staged through the GPU) and with the device module (arrays
resident in HBM)."""

import warnings

import numpy as np
import pytest
import torch

import quantum_systems_amd as qsa
from oracle import qs_oracle as orc
from quantum_systems_amd import (
    BasisSet, GeneralOrbitalSystem, RandomBasisSet, SpatialOrbitalSystem,
    construct_custom_system, hip, setup_basis_set,
)
from quantum_systems_amd.array_module import to_host
from quantum_systems_amd.system_helper import spin_delta

pytestmark = pytest.mark.gpu

MODULES = ["numpy", "hip"]


def module(name):
    return np if name == "numpy" else hip


def H(a):
    return to_host(a)


def change_basis_h(h, c):
    return np.einsum("ap,bq,ab->pq", c.conj(), c, h, optimize=True)


def change_basis_u(u, c):
    return np.einsum("ap,bq,gr,ds,abgd->pqrs", c.conj(), c.conj(), c, c, u, optimize=True)


def seeded_spas(n, l, dim, mod, seed=0):
    np.random.seed(seed)
    spas = SpatialOrbitalSystem(n, RandomBasisSet(l, dim))   # NumPy stream = reference's
    if mod is not np:
        spas.change_module(mod)
    return spas


@pytest.mark.parametrize("mod", MODULES)
def test_setters_l_doubles(mod):
    # reference tests/test_custom_system.py:27-35
    spas = seeded_spas(2, 10, 3, module(mod))
    gos = spas.construct_general_orbital_system()
    assert gos.l == 2 * spas.l and gos.n == 2 * spas.n
    assert tuple(gos.u.shape) == (20,) * 4
    assert spas.l == 10 and tuple(spas.u.shape) == (10,) * 4     # source intact


@pytest.mark.parametrize("mod", MODULES)
def test_change_of_basis_rectangular(mod):
    # reference tests/test_custom_system.py:38-68, atol = rtol = 1e-12
    m = module(mod)
    n, l, dim = 2, 10, 2
    new_l = 2 * l - n
    spas = seeded_spas(n, l, dim, m, seed=5)
    gos = spas.construct_general_orbital_system()
    C_spas = RandomBasisSet.get_random_elements((spas.l, new_l), np)
    C_gos = RandomBasisSet.get_random_elements((gos.l, new_l), np)
    h_spas = change_basis_h(H(spas.h), C_spas)
    u_spas = change_basis_u(H(spas.u), C_spas)
    h_gos = change_basis_h(H(gos.h), C_gos)
    u_gos = change_basis_u(H(gos.u), C_gos)
    spas.change_basis(m.asarray(C_spas))
    gos.change_basis(m.asarray(C_gos))
    assert spas.l == new_l and gos.l == new_l
    assert all(new_l == s for s in spas.h.shape) and all(new_l == s for s in spas.u.shape)
    assert all(new_l == s for s in gos.h.shape) and all(new_l == s for s in gos.u.shape)
    np.testing.assert_allclose(h_spas, H(spas.h), atol=1e-12, rtol=1e-12)
    np.testing.assert_allclose(u_spas, H(spas.u), atol=1e-12, rtol=1e-12)
    np.testing.assert_allclose(h_gos, H(gos.h), atol=1e-12, rtol=1e-12)
    np.testing.assert_allclose(u_gos, H(gos.u), atol=1e-12, rtol=1e-12)
    assert spas.m == new_l - spas.n and spas.v == slice(spas.n, new_l)


@pytest.mark.parametrize("mod", MODULES)
def test_config1_through_the_api(mod, golden):
    # BASELINE.json configs[0]: RandomBasisSet(n=2, l=20), change_basis(unitary C)
    m = module(mod)
    g = golden("config1_l20_change_basis")
    np.random.seed(int(g["seed"]))
    spas = SpatialOrbitalSystem(2, RandomBasisSet(20, 2))
    A = RandomBasisSet.get_random_elements((20, 20), np)
    C, _ = np.linalg.qr(A)
    np.testing.assert_allclose(C, g["C"], rtol=1e-13, atol=1e-13)
    if m is not np:
        spas.change_module(m)
    spas.change_basis(m.asarray(C))
    assert (spas.n, spas.l) == (int(g["n"]), int(g["l"]))
    for k in ("h", "s", "u", "position"):
        np.testing.assert_allclose(H(getattr(spas, k)), g[k], rtol=1e-10, atol=1e-12)
    s = golden("config1_l20_gos_sampled")
    gos = spas.construct_general_orbital_system()
    assert (gos.n, gos.l) == (int(s["n"]), int(s["l"]))
    p, q, r, t = s["idx"].T
    u = H(gos.u)
    assert u.dtype == np.complex128
    np.testing.assert_allclose(u[p, q, r, t], s["u_samples"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(u), s["u_fro"], rtol=1e-12)
    np.testing.assert_allclose(u[3, :8], s["u_slab_p3"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(u, -u.transpose(0, 1, 3, 2), atol=1e-12)
    tb = H(gos.spin_2_tb)
    np.testing.assert_allclose(tb[p, q, r, t], s["spin_2_tb_samples"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(tb), s["spin_2_tb_fro"], rtol=1e-12)
    for k in ("h", "s", "spin_x", "spin_y", "spin_z", "spin_2", "position"):
        np.testing.assert_allclose(H(getattr(gos._basis_set, k)), s[k], rtol=1e-10, atol=1e-12)


def basis_from(g, prefix, l, dim, mod, **flags):
    bs = BasisSet(l, dim, np=mod, **flags)
    for k in ("h", "s", "u", "position", "momentum", "spf"):
        if prefix + k in g:
            setattr(bs, k, mod.asarray(g[prefix + k]))
    return bs


@pytest.mark.parametrize("mod", MODULES)
def test_gos_golden_then_change_basis_on_spin_basis(mod, golden):
    m = module(mod)
    g = golden("gos_l5_default_spinors")
    bs = basis_from(g, "in_", 5, 2, m)
    gos = SpatialOrbitalSystem(4, bs).construct_general_orbital_system()
    assert gos.n == int(g["n_gos"]) and gos.l == 10
    assert bs.l == 5 and not bs.includes_spin                      # original untouched
    gb = gos._basis_set
    assert gb.includes_spin and gb.anti_symmetrized_u
    # the scatter is value-exact (array_equal: -0.0 == +0.0)
    for k in ("h", "s", "u", "position"):
        got = H(getattr(gb, k))
        assert got.dtype == np.complex128
        assert np.array_equal(got, g["gos_" + k]), k
    for k in ("spin_x", "spin_y", "spin_z", "sigma_x", "sigma_y", "sigma_z"):
        assert np.array_equal(H(getattr(gb, k)), g["gos_" + k]), k
    np.testing.assert_allclose(H(gb.spin_2), g["gos_spin_2"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(H(gos.spin_2_tb), g["gos_spin_2_tb"], rtol=1e-13, atol=1e-14)
    # change of basis on the spin basis: u AND spin_2_tb transformed (:374-382),
    # spin one-body operators left alone (:368-372)
    gos.change_basis(m.asarray(g["C"]))
    assert gos.l == int(g["l_after"])
    if mod == "hip":
        # spin_2_tb went through change_basis as its RECIPE (three transformed spin matrices), not as a second
        # four-index transform of a (2l)^4 tensor; the tensor read above was dropped and is rebuilt on access
        assert gb._spin_2_tb is None and gb._spin_2_tb_recipe is not None
        assert tuple(gb._spin_2_tb_recipe[0].shape) == (3, 8, 8) and gb._spin_2_tb_recipe[1] is True
    for k in ("h", "s", "u", "position", "spin_2_tb"):
        np.testing.assert_allclose(H(getattr(gb, k)), g["cb_" + k], rtol=1e-10, atol=1e-12)
    for k in ("spin_x", "spin_y", "spin_z", "spin_2"):
        got = H(getattr(gb, k))
        assert got.shape == (10, 10)
        np.testing.assert_allclose(got, g["cb_" + k], rtol=1e-13, atol=1e-14)


def test_spin_2_tb_recipe_gives_way_to_a_tensor_somebody_wrote_into(golden):
    # the recipe stands for the tensor only while nobody has changed the tensor: after an in-place write the
    # reference's route (:379-382, the four-index transform of the tensor itself) is taken and the write survives
    from quantum_systems_amd import kernels as K

    g = golden("gos_l5_default_spinors")
    bs = basis_from(g, "in_", 5, 2, hip)
    gos = SpatialOrbitalSystem(4, bs).construct_general_orbital_system()
    gb = gos._basis_set
    tb = gos.spin_2_tb
    assert gb._spin_2_tb_recipe is not None and gb._spin_2_tb_recipe_valid()
    tb[1, 2, 3, 4] += 0.5
    changed = H(tb).copy()
    gos.change_basis(hip.asarray(g["C"]))
    assert gb._spin_2_tb_recipe is None and gb._spin_2_tb is not None
    np.testing.assert_allclose(H(gos.spin_2_tb), orc.transform_two_body(changed, g["C"]), rtol=1e-10, atol=1e-12)
    # a basis whose spin_2_tb was never read: recipe in, recipe out, no (2l)^4 tensor ever built -- and the
    # rectangular C of the fixture (10 -> 8) lands in the recipe's shape
    bs = basis_from(g, "in_", 5, 2, hip)
    gos = SpatialOrbitalSystem(4, bs).construct_general_orbital_system()
    gb = gos._basis_set
    calls, real = [], (K.transform_two_body, K.transform_two_body_)

    def counted(fn):
        def inner(*a, **k):
            calls.append(fn.__name__)
            return fn(*a, **k)
        return inner

    K.transform_two_body, K.transform_two_body_ = counted(real[0]), counted(real[1])
    try:
        gos.change_basis(hip.asarray(g["C"]))
    finally:
        K.transform_two_body, K.transform_two_body_ = real
    assert len(calls) == 1                                    # u only: ONE four-index transform per change_basis
    assert gb._spin_2_tb is None and tuple(gb._spin_2_tb_recipe[0].shape) == (3, 8, 8)
    np.testing.assert_allclose(H(gos.spin_2_tb), g["cb_spin_2_tb"], rtol=1e-10, atol=1e-12)
    # copies keep working on their own
    twin = gos.copy_system()
    np.testing.assert_allclose(H(twin.spin_2_tb), g["cb_spin_2_tb"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("mod", MODULES)
def test_gos_custom_spinors_without_antisymmetrisation(mod, golden):
    m = module(mod)
    g = golden("gos_l4_custom_spinors_no_as")
    bs = basis_from(g, "in_", 4, 1, m)
    ret = bs.change_to_general_orbital_basis(a=g["a"], b=g["b"], anti_symmetrize=False)
    assert ret is bs and bs.l == 8 and bs.includes_spin and not bs.anti_symmetrized_u
    assert np.array_equal(H(bs.u), g["gos_u"])
    for k in ("h", "s", "position", "momentum", "spf", "spin_x", "spin_y", "spin_z",
              "spin_2", "spin_2_tb", "sigma_x", "sigma_y", "sigma_z"):
        got = H(getattr(bs, k))
        np.testing.assert_allclose(got, g["gos_" + k], rtol=1e-13, atol=1e-14, err_msg=k)
    assert H(bs.spf).dtype == np.complex128
    # a later explicit anti-symmetrisation covers u and spin_2_tb exactly once
    u0, tb0 = H(bs.u), H(bs.spin_2_tb)
    bs.anti_symmetrize_two_body_elements()
    assert bs.anti_symmetrized_u
    assert np.array_equal(H(bs.u), u0 - u0.transpose(0, 1, 3, 2))
    assert np.array_equal(H(bs.spin_2_tb), tb0 - tb0.transpose(0, 1, 3, 2))
    u1 = H(bs.u)
    bs.anti_symmetrize_two_body_elements()
    assert np.array_equal(H(bs.u), u1)


@pytest.mark.parametrize("mod", MODULES)
def test_change_basis_with_spf_and_explicit_bra(mod, golden):
    m = module(mod)
    g = golden("change_basis_l5_to_7_spf_ctilde")
    bs = basis_from(g, "in_", 5, 2, m)
    bs.change_basis(m.asarray(g["C"]), C_tilde=m.asarray(g["C_tilde"]))
    assert bs.l == 7
    for k in ("h", "s", "u", "position", "momentum", "spf", "bra_spf"):
        np.testing.assert_allclose(H(getattr(bs, k)), g["out_" + k], rtol=1e-10, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("mod", MODULES)
def test_static_transforms_return_new_arrays(mod):
    # reference tests/test_helper.py:14-69 through the static API
    m = module(mod)
    rng = np.random.default_rng(0)
    l = 10
    u = rng.random((l,) * 4) + 1j * rng.random((l,) * 4)
    h = rng.random((l, l)) + 1j * rng.random((l, l))
    C = rng.random((l, l)) + 1j * rng.random((l, l))
    du, dh, dC = m.asarray(u), m.asarray(h), m.asarray(C)
    ht = BasisSet.transform_one_body_elements(dh, dC, np=m)
    np.testing.assert_allclose(H(ht), np.einsum("ip,jq,ij", C.conj(), C, h, optimize=True), atol=1e-10)
    ut = BasisSet.transform_two_body_elements(du, dC, np=m)
    ref = np.einsum("ls,kr,jq,ip,ijkl->pqrs", C, C, C.conj(), C.conj(), u, optimize=True)
    np.testing.assert_allclose(H(ut), ref, atol=1e-10)
    ut2 = BasisSet.transform_two_body_elements(du, dC, m, C_tilde=m.asarray(C.conj().T.copy()))
    np.testing.assert_allclose(H(ut), H(ut2))
    assert np.array_equal(H(du), u) and ut is not du
    assert type(ut) is type(du)
    # system-level forwards (system.py:217-225)
    np.random.seed(1)
    spas = seeded_spas(2, l, 2, m, seed=1)
    fwd = spas.transform_two_body_elements(du, dC)
    np.testing.assert_allclose(H(fwd), ref, atol=1e-10)
    np.testing.assert_allclose(H(spas._basis_set.get_transformed_h(dC)),
                               change_basis_h(H(spas.h), C), atol=1e-10)


@pytest.mark.parametrize("mod", MODULES)
def test_spin_statics_like_reference(mod):
    # reference tests/test_helper.py:72-135
    m = module(mod)
    rng = np.random.default_rng(2)
    lh = 6
    h = rng.random((lh, lh))
    u = rng.random((lh,) * 4)
    u = u + u.transpose(1, 0, 3, 2)
    l = 2 * lh
    h_spin = np.zeros((l, l))
    for p in range(l):
        for q in range(l):
            h_spin[p, q] = spin_delta(p, q) * h[p // 2, q // 2]
    np.testing.assert_allclose(h_spin, H(BasisSet.add_spin_one_body(m.asarray(h), np=m)), atol=1e-10)
    us = BasisSet.add_spin_two_body(m.asarray(u), np=m)
    ua = H(BasisSet.anti_symmetrize_u(us))
    P, Q, R, S = np.meshgrid(*(np.arange(l),) * 4, indexing="ij")
    same = lambda a, b: ((a & 1) == (b & 1)).astype(float)  # noqa: E731
    ref = same(P, R) * same(Q, S) * u[P // 2, Q // 2, R // 2, S // 2]
    np.testing.assert_allclose(H(us), ref, atol=1e-10)
    ref_as = ref - same(P, S) * same(Q, R) * u[P // 2, Q // 2, S // 2, R // 2]
    np.testing.assert_allclose(ua, ref_as, atol=1e-10)
    np.testing.assert_allclose(ua, -ua.transpose(1, 0, 2, 3), atol=1e-10)
    np.testing.assert_allclose(ua, ua.transpose(1, 0, 3, 2), atol=1e-10)


@pytest.mark.parametrize("mod", MODULES)
def test_add_spin_spf_interleaves_rows(mod):
    # reference tests/test_spin.py:15-55
    m = module(mod)
    rng = np.random.default_rng(3)
    spf = rng.random((4, 11)) + 1j * rng.random((4, 11))
    out = H(BasisSet.add_spin_spf(m.asarray(spf), m))
    assert out.shape == (8, 11)
    assert np.array_equal(out[::2], spf) and np.array_equal(out[1::2], spf)
    assert BasisSet.add_spin_bra_spf(None, m) is None


@pytest.mark.parametrize("mod", MODULES)
def test_spin_matrices_are_half_kron_overlap_sigma(mod):
    # reference tests/test_spin.py:58-114
    m = module(mod)
    spas = seeded_spas(2, 6, 2, m, seed=7)
    s = H(spas.s)
    gos = spas.construct_general_orbital_system()
    a = np.array([1, 0]).reshape(-1, 1)
    b = np.array([0, 1]).reshape(-1, 1)
    sx, sy, sz = BasisSet.setup_pauli_matrices(a, b, np)
    for got, sig in ((gos.spin_x, sx), (gos.spin_y, sy), (gos.spin_z, sz)):
        np.testing.assert_allclose(H(got), 0.5 * np.kron(s, sig))
    s_up = H(gos.spin_x) + 1j * H(gos.spin_y)
    s_down = H(gos.spin_x) - 1j * H(gos.spin_y)
    np.testing.assert_allclose(s_up, np.kron(s, [[0, 1], [0, 0]]), atol=1e-14)
    np.testing.assert_allclose(s_down, np.kron(s, [[0, 0], [1, 0]]), atol=1e-14)


def test_change_module_moves_arrays_both_ways():
    spas = seeded_spas(2, 6, 2, np, seed=8)
    ref = {k: getattr(spas, k).copy() for k in ("h", "s", "u", "position")}
    spas.change_module(hip)
    assert spas.np is hip and spas._basis_set.np is hip
    for k in ref:
        arr = getattr(spas, k)
        assert isinstance(arr, torch.Tensor) and arr.is_cuda
    # NumPy-flavoured methods on device arrays
    ut = spas.u.transpose(0, 1, 3, 2)
    assert np.array_equal(H(ut), ref["u"].transpose(0, 1, 3, 2))
    assert np.array_equal(H(spas.h.copy()), ref["h"])
    assert spas.h.astype(np.complex128).dtype == torch.complex128
    with pytest.raises(TypeError):
        np.asarray(spas.h)
    spas.change_module(np)
    for k in ref:
        arr = getattr(spas, k)
        assert isinstance(arr, np.ndarray) and np.array_equal(arr, ref[k])


def test_copy_system_on_device_is_independent():
    # reference tests/test_copy.py:6-18 with resident arrays
    spas = seeded_spas(2, 5, 2, hip, seed=9)
    other = spas.copy_system()
    assert other.np is hip and other._basis_set.np is hip and spas.np is hip
    assert other.u.data_ptr() != spas.u.data_ptr()
    assert torch.equal(other.u, spas.u)
    other.h[0, 0] = 5.0
    assert spas.h[0, 0] != 5.0


@pytest.mark.parametrize("mod", MODULES)
def test_construct_custom_system_applies_C(mod):
    # custom_system.py:92-93
    m = module(mod)
    rng = np.random.default_rng(10)
    l = 6
    h = rng.random((l, l)); h = h + h.T
    s = np.eye(l)
    u = rng.random((l,) * 4); u = u + u.transpose(1, 0, 3, 2)
    C = np.linalg.qr(rng.standard_normal((l, l)))[0]
    sys_ = construct_custom_system(2, l, m.asarray(s), m.asarray(h), m.asarray(u), np=m,
                                   system_type="spatial", C=m.asarray(C))
    np.testing.assert_allclose(H(sys_.u), change_basis_u(u, C), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(H(sys_.s), np.eye(l), atol=1e-12)
    gos = construct_custom_system(2, l, m.asarray(s), m.asarray(h), m.asarray(u), np=m)
    assert gos.l == 2 * l and H(gos.u).dtype == np.complex128


@pytest.mark.parametrize("mod", MODULES)
def test_reference_energy_and_fock_consistency(mod):
    # the Fock matrix / energy of the spin-doubled system equal the closed-shell ones
    m = module(mod)
    spas = seeded_spas(4, 6, 2, m, seed=11)
    gos = spas.construct_general_orbital_system()
    e_s = complex(H(spas.compute_reference_energy()))
    e_g = complex(H(gos.compute_reference_energy()))
    assert abs(e_s - e_g) <= 1e-10 * abs(e_s)
    f_s = H(spas.construct_fock_matrix(spas.h, spas.u))
    f_g = H(gos.construct_fock_matrix(gos.h, gos.u))
    np.testing.assert_allclose(f_g, np.kron(f_s, np.eye(2)), rtol=1e-10, atol=1e-12)


def test_double_doubling_warns_on_device():
    spas = seeded_spas(2, 4, 1, hip, seed=12)
    gos = spas.construct_general_orbital_system()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert gos._basis_set.change_to_general_orbital_basis() is None
    assert len(w) == 1 and gos.l == 8


def test_change_basis_reuses_the_storage_of_the_tensor_it_drops():
    # the reference rebinds self.u and lets the old array go (basis_set.py:374-377); on the device the old
    # storage becomes the result's storage (in-place transform, one spare buffer) -- unless anybody else can
    # still see the old array, in which case it stays intact exactly as in NumPy
    import quantum_systems_amd as qsa
    from quantum_systems_amd import kernels as K

    l = 24
    rng = np.random.default_rng(8)
    u = rng.standard_normal((l,) * 4)
    h = rng.standard_normal((l, l))
    C = np.linalg.qr(rng.standard_normal((l, l)))[0]
    ref = orc.transform_two_body(u, C)

    def system():
        bs = qsa.BasisSet(l, 1, np=hip)
        bs.h, bs.s, bs.u = hip.asarray(h), hip.asarray(np.eye(l)), hip.asarray(u)
        bs.donate_u_from = 8                                   # (default: from 96 orbitals up)
        return bs

    bs = system()
    ptr = torch.as_tensor(bs.u).data_ptr()
    bs.change_basis(hip.asarray(C))
    assert K.last_dispatch() != "" and torch.as_tensor(bs.u).data_ptr() == ptr          # same storage
    np.testing.assert_allclose(H(bs.u), ref, rtol=1e-11, atol=1e-12)
    # a second reference to the old array: no reuse, the old array survives unchanged
    bs = system()
    keep = bs.u
    bs.change_basis(hip.asarray(C))
    assert torch.as_tensor(bs.u).data_ptr() != keep.data_ptr()
    assert np.array_equal(H(keep), u)
    np.testing.assert_allclose(H(bs.u), ref, rtol=1e-11, atol=1e-12)
    # a view of the old array counts as well
    bs = system()
    view = bs.u[0]
    bs.change_basis(hip.asarray(C))
    assert np.array_equal(H(view), u[0])
    # ... and so does a tensor built on the same storage without being a view
    bs = system()
    raw = torch.as_tensor(bs.u)
    alias = torch.empty(0, dtype=raw.dtype, device=raw.device).set_(raw.untyped_storage(), 0, tuple(raw.shape))
    del raw
    bs.change_basis(hip.asarray(C))
    assert np.array_equal(alias.cpu().numpy(), u)
    del alias
    # the reuse can be switched off
    bs = system()
    bs.donate_u_from = None
    ptr = torch.as_tensor(bs.u).data_ptr()
    bs.change_basis(hip.asarray(C))
    assert torch.as_tensor(bs.u).data_ptr() != ptr
    # a refusal before anything is launched leaves the tensor with the basis set
    bs = system()
    ptr = torch.as_tensor(bs.u).data_ptr()
    real = K.transform_two_body_

    def refuse(*a, **k):
        raise RuntimeError("no workspace")

    K.transform_two_body_ = refuse
    try:
        with pytest.raises(RuntimeError, match="no workspace"):
            bs.change_basis(hip.asarray(C))
    finally:
        K.transform_two_body_ = real
    assert bs.u is not None and torch.as_tensor(bs.u).data_ptr() == ptr and np.array_equal(H(bs.u), u)
    # shrinking basis: result at the start of the old storage; growing basis / complex C on a real u: ordinary path
    bs = system()
    ptr = torch.as_tensor(bs.u).data_ptr()
    bs.change_basis(hip.asarray(C[:, :17].copy()))
    assert bs.l == 17 and torch.as_tensor(bs.u).data_ptr() == ptr
    np.testing.assert_allclose(H(bs.u), orc.transform_two_body(u, C[:, :17]), rtol=1e-11, atol=1e-12)
    bs = system()
    Cc = C.astype(np.complex128) * np.exp(0.3j)
    bs.change_basis(hip.asarray(Cc))
    np.testing.assert_allclose(H(bs.u), orc.transform_two_body(u, Cc), rtol=1e-11, atol=1e-12)


def test_inplace_transform_kernel_wrapper():
    from quantum_systems_amd import kernels as K

    rng = np.random.default_rng(81)
    for (L, M, cplx) in ((20, 20, False), (130, 128, False), (18, 11, True)):
        u = rng.standard_normal((L,) * 4)
        C = rng.standard_normal((L, M)) / np.sqrt(L)
        Ct = rng.standard_normal((M, L)) / np.sqrt(L)
        if cplx:
            u = u + 1j * rng.standard_normal((L,) * 4)
            C = C + 1j * rng.standard_normal((L, M))
            Ct = Ct + 1j * rng.standard_normal((M, L))
        du = torch.from_numpy(u).cuda()
        ref = K.transform_two_body(du, torch.from_numpy(C).cuda(), torch.from_numpy(Ct).cuda())
        got = K.transform_two_body_(du, torch.from_numpy(C).cuda(), torch.from_numpy(Ct).cuda())
        assert got.data_ptr() == du.data_ptr() and tuple(got.shape) == (M,) * 4
        assert torch.equal(got, ref)                           # same products in the same order: bit-identical
    with pytest.raises(ValueError):
        K.transform_two_body_(torch.zeros((4,) * 4, dtype=torch.float64, device="cuda"),
                              torch.zeros((4, 6), dtype=torch.float64, device="cuda"))      # growing basis


@pytest.mark.parametrize("l,spin", [(12, False), (20, False), (6, True)])
def test_change_basis_plan_replays_change_basis_bit_for_bit(l, spin):
    # round 4: change_basis with a square C captured as a HIP graph (two buffer sets, a call = copy C + one replay + rebinding
    # the attributes): the same kernels as change_basis, so the same bits -- h, s, position, u, and on a spin-doubled basis
    # the spin_2_tb recipe; explicit bra coefficients as a second captured form
    np.random.seed(11)
    make = (lambda: SpatialOrbitalSystem(2, RandomBasisSet(l, 2))) if not spin else \
        (lambda: SpatialOrbitalSystem(2, RandomBasisSet(l, 2)).construct_general_orbital_system())
    state = np.random.get_state()
    a = make()
    np.random.set_state(state)
    b = make()
    a.change_module(hip)
    b.change_module(hip)
    n = a.l
    rng = np.random.default_rng(5)
    Cs = [np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0] for _ in range(4)]
    Cs = [torch.from_numpy(c).cuda() for c in Cs]
    plan = b.change_basis_plan()
    for c in Cs[:3]:
        a.change_basis(c)
        assert plan(c) is b._basis_set
        for name in ("h", "s", "u", "position"):
            assert torch.equal(torch.as_tensor(getattr(a, name)), torch.as_tensor(getattr(b, name))), name
    if spin:
        assert torch.equal(torch.as_tensor(a.spin_2_tb), torch.as_tensor(b.spin_2_tb))
        assert torch.equal(torch.as_tensor(a.spin_x), torch.as_tensor(b.spin_x))      # (left alone by both, quirk 0.6)
    assert abs(complex(H(a.compute_reference_energy())) - complex(H(b.compute_reference_energy()))) == 0
    # explicit bra coefficients
    Ct = torch.linalg.inv(Cs[3])
    plan2 = b.change_basis_plan(C_tilde_given=True)
    a.change_basis(Cs[3], C_tilde=Ct)
    plan2(Cs[3], Ct)
    for name in ("h", "s", "u", "position"):
        assert torch.equal(torch.as_tensor(getattr(a, name)), torch.as_tensor(getattr(b, name))), name
    with pytest.raises(ValueError):
        plan2(Cs[3])
    # what a plan does not do
    rect = torch.from_numpy(rng.standard_normal((n, n - 1))).cuda()
    with pytest.raises(ValueError):
        plan(rect.to(torch.complex128))
    with pytest.raises(ValueError):
        plan(Cs[0][0])                                  # a vector would broadcast under copy_: refused


def test_change_basis_plan_on_real_arrays_refuses_complex_coefficients():
    # the dtype of a plan's arrays is fixed when it is captured: a complex C for real arrays is an error, not a silent cast;
    # a real C replays change_basis bit for bit, numpy coefficients are taken as they are
    l = 10
    rng = np.random.default_rng(21)
    h, u = rng.standard_normal((l, l)), rng.standard_normal((l,) * 4)

    def system():
        bs = qsa.BasisSet(l, 1, np=hip)
        bs.h, bs.s, bs.u = hip.asarray(h), hip.asarray(np.eye(l)), hip.asarray(u)
        return bs

    a, b = system(), system()
    plan = b.change_basis_plan()
    C = np.linalg.qr(rng.standard_normal((l, l)))[0]
    a.change_basis(hip.asarray(C))
    plan(C)
    assert torch.equal(torch.as_tensor(a.u), torch.as_tensor(b.u)) and torch.equal(torch.as_tensor(a.h), torch.as_tensor(b.h))
    with pytest.raises(TypeError):
        plan(C + 0.5j)
    assert torch.equal(torch.as_tensor(a.u), torch.as_tensor(b.u))      # the refused call changed nothing
