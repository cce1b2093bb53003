"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container, where /root/reference is mounted:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference package (HyQD/quantum-systems v0.2.6), drives its own
``BasisSet`` / ``QuantumSystem`` API on small seeded inputs and stores inputs
and outputs as ``.npz``.  The fixtures are data (arrays); no reference source
is written anywhere.

The reference imports ``numba`` at package import time for two trivial helper
functions that are not on the transform path (system_helper.py:4-11) and for
its quantum-dot generators.  numba is not installed here, so an identity
``njit`` decorator is registered below for the duration of this script; every
array produced below comes from plain NumPy code in the reference.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    nb = types.ModuleType("numba")

    def passthrough(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    nb.njit = passthrough
    nb.jit = passthrough
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    sys.path.insert(0, "/root/reference")
    import quantum_systems

    return quantum_systems


qs = _import_reference()
BasisSet = qs.BasisSet


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e3:.1f} kB")


def crand(rng, *shape):
    return rng.random(shape) + 1j * rng.random(shape)


def case_transforms():
    rng = np.random.default_rng(20241101)
    # complex, square, non-unitary C (as tests/test_helper.py:38-69)
    l = 6
    u = crand(rng, l, l, l, l)
    h = crand(rng, l, l)
    C = crand(rng, l, l)
    save(
        "transform_c128_square",
        u=u, h=h, C=C,
        u_out=BasisSet.transform_two_body_elements(u, C, np),
        h_out=BasisSet.transform_one_body_elements(h, C, np),
    )
    # complex, rectangular L=6 -> M=9, explicit C_tilde != C^dagger
    C = crand(rng, l, 9)
    Ct = crand(rng, 9, l)
    save(
        "transform_c128_rect_ctilde",
        u=u, h=h, C=C, C_tilde=Ct,
        u_out=BasisSet.transform_two_body_elements(u, C, np, C_tilde=Ct),
        h_out=BasisSet.transform_one_body_elements(h, C, np, C_tilde=Ct),
    )
    # complex, rectangular shrinking L=6 -> M=4, default C_tilde
    C = crand(rng, l, 4)
    save(
        "transform_c128_shrink",
        u=u, h=h, C=C,
        u_out=BasisSet.transform_two_body_elements(u, C, np),
        h_out=BasisSet.transform_one_body_elements(h, C, np),
    )
    # real fp64, orthogonal C, odd l (unaligned rows)
    l = 7
    u = rng.standard_normal((l, l, l, l))
    u = 0.5 * (u + u.transpose(1, 0, 3, 2))
    h = rng.standard_normal((l, l))
    C, _ = np.linalg.qr(rng.standard_normal((l, l)))
    save(
        "transform_f64_orthogonal",
        u=u, h=h, C=C,
        u_out=BasisSet.transform_two_body_elements(u, C, np),
        h_out=BasisSet.transform_one_body_elements(h, C, np),
    )
    # real fp64 rectangular 7 -> 10
    C = rng.standard_normal((l, 10))
    save(
        "transform_f64_rect",
        u=u, h=h, C=C,
        u_out=BasisSet.transform_two_body_elements(u, C, np),
        h_out=BasisSet.transform_one_body_elements(h, C, np),
    )
    # mixed: real u, complex C (NumPy promotes to complex128)
    C = crand(rng, l, l)
    save(
        "transform_mixed_real_u_complex_C",
        u=u, h=h, C=C,
        u_out=BasisSet.transform_two_body_elements(u, C, np),
        h_out=BasisSet.transform_one_body_elements(h, C, np),
    )
    # spf transforms (basis_set.py:321-327)
    spf = crand(rng, l, 5, 3)
    bra = crand(rng, l, 5, 3)
    Ct = crand(rng, l, l)
    save(
        "transform_spf",
        spf=spf, bra_spf=bra, C=C, C_tilde=Ct,
        spf_out=BasisSet.transform_spf(spf, C, np),
        bra_out=BasisSet.transform_bra_spf(bra, Ct, np),
    )


def case_spin_statics():
    rng = np.random.default_rng(7)
    l = 5
    u = rng.standard_normal((l, l, l, l))  # negative entries -> signed zeros
    h = rng.standard_normal((l, l))
    spf = crand(rng, l, 4)
    us = BasisSet.add_spin_two_body(u, np)
    save(
        "spin_statics_f64",
        u=u, h=h, spf=spf,
        h_spin=BasisSet.add_spin_one_body(h, np),
        u_spin=us,
        u_spin_as=BasisSet.anti_symmetrize_u(us),
        u_as=BasisSet.anti_symmetrize_u(u),
        spf_spin=BasisSet.add_spin_spf(spf, np),
    )
    uc = crand(rng, l, l, l, l) - (0.5 + 0.5j)
    usc = BasisSet.add_spin_two_body(uc, np)
    save(
        "spin_statics_c128",
        u=uc,
        u_spin=usc,
        u_spin_as=BasisSet.anti_symmetrize_u(usc),
        u_as=BasisSet.anti_symmetrize_u(uc),
    )


def _basis_fields(bs):
    out = {}
    for name in (
        "h", "s", "u", "position", "momentum", "spf", "bra_spf",
        "spin_x", "spin_y", "spin_z", "spin_2", "spin_2_tb",
        "sigma_x", "sigma_y", "sigma_z",
    ):
        val = getattr(bs, name)
        if val is not None:
            out[name] = np.asarray(val)
    return out


def case_random_basis_stream():
    # pins the draw order of RandomBasisSet (random_basis.py:21-35)
    np.random.seed(1234)
    rbs = qs.RandomBasisSet(4, 3)
    save(
        "random_basis_seed1234_l4_dim3",
        nuclear_repulsion_energy=np.float64(rbs.nuclear_repulsion_energy),
        charge=np.int64(rbs.charge),
        **_basis_fields(rbs),
    )


def case_config1():
    # BASELINE.json configs[0]: RandomBasisSet(l=20, dim=2), n=2, unitary C
    np.random.seed(2024)
    rbs = qs.RandomBasisSet(20, 2)
    spas = qs.SpatialOrbitalSystem(2, rbs)
    A = qs.RandomBasisSet.get_random_elements((20, 20), np)
    C, _ = np.linalg.qr(A)
    u_in_sample = spas.u[::3, 1::4, 2::5, ::2].copy()
    spas.change_basis(C)
    save(
        "config1_l20_change_basis",
        seed=np.int64(2024), C=C,
        u_in_sample=u_in_sample,
        n=np.int64(spas.n), l=np.int64(spas.l),
        h=spas.h, s=spas.s, u=spas.u, position=spas.position,
    )
    # then the spin doubling of the transformed system: l=40, sampled
    gos = spas.construct_general_orbital_system()
    idx = np.random.default_rng(5).integers(0, 40, size=(4096, 4))
    p, q, r, s = idx.T
    save(
        "config1_l20_gos_sampled",
        idx=idx,
        u_samples=gos.u[p, q, r, s],
        spin_2_tb_samples=gos.spin_2_tb[p, q, r, s],
        u_fro=np.float64(np.linalg.norm(gos.u)),
        spin_2_tb_fro=np.float64(np.linalg.norm(gos.spin_2_tb)),
        u_slab_p3=gos.u[3, :8],
        h=gos.h, s=gos.s, spin_x=gos.spin_x, spin_y=gos.spin_y,
        spin_z=gos.spin_z, spin_2=gos.spin_2, position=gos.position,
        n=np.int64(gos.n), l=np.int64(gos.l),
    )


def case_gos_small():
    # SpatialOrbitalSystem -> GeneralOrbitalSystem, full tensors, l=5 -> 10,
    # then a rectangular change_basis on the spin basis (10 -> 8), which
    # transforms u AND spin_2_tb (basis_set.py:374-382) and leaves the spin
    # one-body operators untouched (:368-372).
    np.random.seed(99)
    rbs = qs.RandomBasisSet(5, 2)
    spas = qs.SpatialOrbitalSystem(4, rbs)
    spatial = _basis_fields(rbs)
    gos = spas.construct_general_orbital_system()
    doubled = _basis_fields(gos._basis_set)
    C = qs.RandomBasisSet.get_random_elements((10, 8), np)
    gos.change_basis(C)
    after = _basis_fields(gos._basis_set)
    save(
        "gos_l5_default_spinors",
        C=C,
        **{"in_" + k: v for k, v in spatial.items()},
        **{"gos_" + k: v for k, v in doubled.items()},
        **{"cb_" + k: v for k, v in after.items()},
        n_gos=np.int64(gos.n), l_after=np.int64(gos.l),
    )
    # custom spinor basis, no anti-symmetrisation, real-valued input with spf
    rng = np.random.default_rng(3)
    l = 4
    bs = qs.BasisSet(l, dim=1, np=np)
    bs.h = rng.standard_normal((l, l))
    bs.s = np.eye(l) + 0.1 * rng.standard_normal((l, l))
    bs.s = 0.5 * (bs.s + bs.s.T)
    bs.u = rng.standard_normal((l, l, l, l))
    bs.position = rng.standard_normal((1, l, l))
    bs.momentum = rng.standard_normal((1, l, l))
    bs.spf = crand(rng, l, 6)
    spatial = _basis_fields(bs)
    a = np.array([1, 1j]) / np.sqrt(2)
    b = np.array([1, -1j]) / np.sqrt(2)
    bs.change_to_general_orbital_basis(a=a, b=b, anti_symmetrize=False)
    doubled = _basis_fields(bs)
    save(
        "gos_l4_custom_spinors_no_as",
        a=a, b=b,
        **{"in_" + k: v for k, v in spatial.items()},
        **{"gos_" + k: v for k, v in doubled.items()},
    )


def case_change_basis_with_spf():
    rng = np.random.default_rng(11)
    l, m = 5, 7
    bs = qs.BasisSet(l, dim=2, np=np)
    bs.h = crand(rng, l, l)
    bs.s = crand(rng, l, l)
    bs.u = crand(rng, l, l, l, l)
    bs.position = crand(rng, 2, l, l)
    bs.momentum = crand(rng, 2, l, l)
    bs.spf = crand(rng, l, 4, 3)
    before = _basis_fields(bs)
    C = crand(rng, l, m)
    Ct = crand(rng, m, l)
    bs.change_basis(C, C_tilde=Ct)
    after = _basis_fields(bs)
    save(
        "change_basis_l5_to_7_spf_ctilde",
        C=C, C_tilde=Ct,
        **{"in_" + k: v for k, v in before.items()},
        **{"out_" + k: v for k, v in after.items()},
    )


def case_tdho_coulomb():
    """2-D harmonic-oscillator Coulomb elements: (i) the reference's own table of
    elements (tests/dat/two_dim_quantum_dots_coulomb_elements.dat, orbitals 0..35,
    ~8 significant digits, used by its tests with atol 1e-6) repacked as arrays --
    data, not source; (ii) elements computed here by the reference's coulomb_ho in
    interpreter mode (numba absent), incl. the high shells of the l=55 basis; (iii) its
    index map and shell energies."""
    from quantum_systems.quantum_dots.two_dim.coulomb_elements import coulomb_ho
    from quantum_systems.quantum_dots.two_dim.two_dim_helper import (
        get_indices_nm, get_one_body_elements,
    )

    rows = np.loadtxt("/root/reference/tests/dat/two_dim_quantum_dots_coulomb_elements.dat")
    save("tdho_coulomb_table", idx=rows[:, :4].astype(np.uint8), val=rows[:, 4])
    nm = np.array([get_indices_nm(p) for p in range(66)], dtype=np.int64)
    rng = np.random.default_rng(2)
    picks = []
    # m-conserving index quadruples: low shells exhaustively sampled, high shells sparsely
    while len(picks) < 60:
        hi = 55 if len(picks) >= 40 else 21
        p, q, r = rng.integers(0, hi, size=3)
        target = nm[p, 1] + nm[q, 1] - nm[r, 1]
        cand = [s for s in range(hi) if nm[s, 1] == target]
        if not cand:
            continue
        s = int(rng.choice(cand))
        if len(picks) >= 40 and max(nm[p, 0], nm[q, 0], nm[r, 0], nm[s, 0]) > 2:
            continue  # keep the interpreted evaluation to seconds per element
        picks.append((p, q, r, s))
    picks = np.array(picks, dtype=np.int64)
    vals = np.array([
        coulomb_ho(nm[p, 0], nm[p, 1], nm[q, 0], nm[q, 1], nm[r, 0], nm[r, 1], nm[s, 0], nm[s, 1])
        for p, q, r, s in picks
    ])
    save("tdho_coulomb_spot", idx=picks, val=vals, index_map=nm,
         one_body_l55=np.diag(get_one_body_elements(55)))


def case_odqd():
    """1-D quantum dot.  (i) the reference class itself on small grids, all
    fields; (ii) the reference's own regression files tests/dat/od*_{h,u,spf,dipole_moment}.npy
    (GeneralOrbitalSystem(2, ODQD(10, ..., 1001, potential)), 20 spin orbitals): h and the dipole
    in full, u and spf as a seeded sample of entries plus their absolute sums (the files are
    2.5 MB each; the reference compares them by absolute value, tests/test_one_dim_qd.py:127-142);
    (iii) the 2-d-u transform of ODSincDVR on a small case."""
    ODQD = qs.ODQD
    for tag, pot, args in (
        ("ho", ODQD.HOPotential(1.0), dict(l=6, grid_length=5, num_grid_points=101)),
        ("dw", ODQD.DWPotential(1.0, 5.0), dict(l=5, grid_length=6, num_grid_points=152, a=0.3, alpha=0.9, beta=0.1)),
    ):
        bs = ODQD(potential=pot, **args)
        save(f"odqd_small_{tag}", grid=bs.grid, eigen_energies=bs.eigen_energies, spf=bs.spf, h=bs.h,
             s=bs.s, u=bs.u, position=bs.position,
             params=np.array([args["l"], args["grid_length"], args["num_grid_points"],
                              args.get("a", 0.25), args.get("alpha", 1.0), args.get("beta", 0.0)]))
    rng = np.random.default_rng(7)
    dat = "/root/reference/tests/dat"
    out = {}
    for name in ("odho", "oddw", "odgauss", "oddw_smooth"):
        h = np.load(os.path.join(dat, f"{name}_h.npy"))
        dip = np.load(os.path.join(dat, f"{name}_dipole_moment.npy"))
        u = np.load(os.path.join(dat, f"{name}_u.npy"))
        spf = np.load(os.path.join(dat, f"{name}_spf.npy"))
        ui = rng.integers(0, u.shape[0], size=(3000, 4))
        si = np.stack([rng.integers(0, spf.shape[0], 2000), rng.integers(0, spf.shape[1], 2000)], axis=1)
        out.update({
            f"{name}_h": h, f"{name}_dipole_moment": dip,
            f"{name}_u_idx": ui, f"{name}_u_val": u[tuple(ui.T)], f"{name}_u_abs_sum": np.abs(u).sum(),
            f"{name}_u_shape": np.array(u.shape),
            f"{name}_spf_idx": si, f"{name}_spf_val": spf[tuple(si.T)], f"{name}_spf_abs_sum": np.abs(spf).sum(),
            f"{name}_spf_shape": np.array(spf.shape),
        })
    save("odqd_reference_regression_files", **out)
    # sinc-DVR: 2-d u transformed with a complex C, with and without the fused anti-symmetrisation
    from quantum_systems.sinc_dvr.one_dim.sinc_dvr import ODSincDVR

    dvr = ODSincDVR(12, 6.0, potential=ODSincDVR.HOPotential(0.5))
    C = crand(rng, 12, 7)
    Ct = crand(rng, 7, 12)
    save("sinc_dvr_small", h=dvr.h, s=dvr.s, spf=dvr.spf, u2d=dvr.u, position=dvr.position, grid=dvr.grid,
         C=C, C_tilde=Ct,
         u_default_bra=dvr.transform_two_body_elements(dvr.u, C, np),
         u_ctilde=dvr.transform_two_body_elements(dvr.u, C, np, C_tilde=Ct),
         u_ctilde_as=dvr.transform_two_body_elements(dvr.u, C, np, anti_symmetrize=True, C_tilde=Ct))


def case_sinc_dvr_spin():
    """ODSincDVR with its 2-d u through the spin doubling (the only upstream route to a spin DVR basis):
    add_spin_two_body / anti_symmetrize_u are overridden there (sinc_dvr.py:200-215) and
    change_to_general_orbital_basis reaches them through self (basis_set.py:576, :523)."""
    from quantum_systems.sinc_dvr.one_dim.sinc_dvr import ODSincDVR

    out = {}
    for tag, anti in (("as", True), ("noas", False)):
        dvr = ODSincDVR(6, 4.0, potential=ODSincDVR.HOPotential(0.5))
        assert dvr.u.shape == (6, 6)
        ret = dvr.change_to_general_orbital_basis(anti_symmetrize=anti)
        assert ret is dvr
        out[tag + "_u"] = dvr.u
        out[tag + "_h"], out[tag + "_s"] = dvr.h, dvr.s
        out[tag + "_position"], out[tag + "_spf"] = dvr.position, dvr.spf
        out[tag + "_flags"] = np.array([dvr.includes_spin, dvr.anti_symmetrized_u, dvr.spin_2_tb is None])
    # includes_spin=True at construction, then the explicit driver call on the 2-d u: K stays K
    dvr = ODSincDVR(6, 4.0, potential=ODSincDVR.HOPotential(0.5), includes_spin=True)
    out["spin_ctor_u_before"] = dvr.u.copy()
    dvr.anti_symmetrize_two_body_elements()
    out["spin_ctor_u_after"] = dvr.u.copy()
    out["spin_ctor_l"] = np.int64(dvr.l)
    save("sinc_dvr_spin_doubling", **out)


def case_tdho_one_body():
    """One-body side of the 2-D dots from the reference's own functions: double-well Hamiltonians,
    the orbital table and dipole elements of a small oscillator (the class methods are run on a bare
    instance so that the interpreter-mode Coulomb generator is not needed), the eigenvalues its test
    quotes (tests/test_two_dim_dw.py:93-112) and its double-well regression files
    (tests/dat/tddw_{h,u,dipole_moment}.npy; u as a seeded sample + absolute sum)."""
    from quantum_systems.quantum_dots.two_dim import two_dim_helper as hlp
    from quantum_systems.quantum_dots.two_dim.two_dim_ho import TwoDimensionalHarmonicOscillator as TDHO

    out = {}
    for tag, (l, omega, mass, b, axis) in {"a": (10, 0.8, 1, 3, 0), "b": (12, 1.0, 1, 2, 1), "c": (6, 1.0, 1, 2, 1),
                                            "d": (8, 0.5, 2.0, 1.5, 0)}.items():
        out[f"dw_{tag}_params"] = np.array([l, omega, mass, b, axis], dtype=float)
        out[f"dw_{tag}_h"] = hlp.get_double_well_one_body_elements(l, omega, mass, b, dtype=np.complex128, axis=axis)
    out["smooth_params"] = np.array([6, 1.0, 1.0, 2.0, 2.0])
    out["smooth_h"] = hlp.get_smooth_double_well_one_body_elements(6, 1.0, 1.0, a=2.0, b=2.0, dtype=np.complex128)
    out["test_energies_l6_b2_axis1"] = np.array([0.81129823, 1.37162083, 1.93581042, 2.21403823, 2.37162083, 2.93581042])
    bare = object.__new__(TDHO)
    bare.l, bare.mass, bare.omega, bare.num_grid_points, bare.np = 10, 1, 0.8, 21, np
    bare.radius = np.linspace(0, 4, 21)
    bare.theta = np.linspace(0, 2 * np.pi, 21)
    TDHO.setup_spf(bare)
    TDHO.construct_position_integrals(bare)
    out["tdho_l10_spf"], out["tdho_l10_position"] = bare._spf, bare._position
    out["tdho_l10_params"] = np.array([10, 4.0, 21, 0.8, 1.0])
    rng = np.random.default_rng(11)
    dat = "/root/reference/tests/dat"
    u = np.load(os.path.join(dat, "tddw_u.npy"))
    ui = rng.integers(0, u.shape[0], size=(4000, 4))
    out.update(tddw_h=np.load(os.path.join(dat, "tddw_h.npy")),
               tddw_dipole_moment=np.load(os.path.join(dat, "tddw_dipole_moment.npy")),
               tddw_u_idx=ui, tddw_u_val=u[tuple(ui.T)], tddw_u_abs_sum=np.abs(u).sum(),
               tddw_u_shape=np.array(u.shape))
    # the reference's orbital tables tests/dat/2d-ho-qd-spf-p=*.dat (radius 4, 101 x 101 points,
    # tests/conftest.py:155-168): 600 seeded grid points per orbital + the absolute sum of each table
    pts = np.stack([rng.integers(0, 101, 600), rng.integers(0, 101, 600)], axis=1)
    vals, sums = [], []
    for p in range(15):
        tab = np.loadtxt(os.path.join(dat, f"2d-ho-qd-spf-p={p}.dat")).view(complex)
        vals.append(tab[tuple(pts.T)])
        sums.append(np.abs(tab).sum())
    out.update(spf_files_pts=pts, spf_files_val=np.array(vals), spf_files_abs_sum=np.array(sums),
               spf_files_shape=np.array(tab.shape))
    # magnetic-field dot: the reference's level table (pandas frame) for a few parameter sets and its
    # regression files tests/dat/tdhob_{h,u,dipole_moment}.npy (GeneralOrbitalSystem(2, TwoDimHarmonicOscB(
    # 10, 5, 201, omega_c=0.5)), tests/test_two_dim_ho_b_field.py:11-40)
    for tag, (l, wc, w0) in {"a": (10, 0.5, 1.0), "b": (6, 0.0, 1.0), "c": (12, 1.3, 0.7)}.items():
        w = np.sqrt(w0**2 + wc**2 / 4)
        df = hlp.construct_dataframe(np.arange(l), np.arange(-l - 5, l + 6), omega_c=wc, omega=w)
        out[f"levels_{tag}_params"] = np.array([l, wc, w0])
        out[f"levels_{tag}_nm"] = df[["n", "m"]].values.astype(int)
        out[f"levels_{tag}_E"] = df["E"].values
    ub = np.load(os.path.join(dat, "tdhob_u.npy"))
    bi = rng.integers(0, ub.shape[0], size=(4000, 4))
    out.update(tdhob_h=np.load(os.path.join(dat, "tdhob_h.npy")),
               tdhob_dipole_moment=np.load(os.path.join(dat, "tdhob_dipole_moment.npy")),
               tdhob_u_idx=bi, tdhob_u_val=ub[tuple(bi.T)], tdhob_u_abs_sum=np.abs(ub).sum(),
               tdhob_u_shape=np.array(ub.shape))
    save("tdho_one_body", **out)


def case_fock_energy():
    """Fock matrix and reference energy of the reference's own system classes
    (spatial_orbital_system.py:106-190, general_orbital_system.py:75-159) on seeded
    RandomBasisSets: before and after a change of basis, closed-shell and spin-orbital."""
    out = {}
    np.random.seed(4242)
    l, n = 8, 4
    bs = qs.RandomBasisSet(l, 2)
    spas = qs.SpatialOrbitalSystem(n, bs)
    out["n"], out["l"] = np.int64(n), np.int64(l)
    out["h"], out["u"], out["s"] = spas.h.copy(), spas.u.copy(), spas.s.copy()
    out["e_nuc"] = np.float64(spas.nuclear_repulsion_energy)
    out["spas_energy"] = np.complex128(spas.compute_reference_energy())
    out["spas_fock"] = spas.construct_fock_matrix(spas.h, spas.u)
    # custom (h, u) arguments: occupied block only, as the docstring allows
    o = spas.o
    out["spas_energy_occ_block"] = np.complex128(
        spas.compute_reference_energy(h=spas.h[o, o], u=spas.u[o, o, o, o]))
    gos = spas.construct_general_orbital_system()
    out["gos_energy"] = np.complex128(gos.compute_reference_energy())
    out["gos_fock"] = gos.construct_fock_matrix(gos.h, gos.u)
    out["gos_h"] = gos.h.copy()
    rng = np.random.default_rng(77)
    C, _ = np.linalg.qr(crand(rng, l, l))
    out["C"] = C
    spas.change_basis(C)
    out["spas_cb_energy"] = np.complex128(spas.compute_reference_energy())
    out["spas_cb_fock"] = spas.construct_fock_matrix(spas.h, spas.u)
    C2, _ = np.linalg.qr(crand(rng, 2 * l, 2 * l))
    out["C_gos"] = C2
    gos.change_basis(C2)
    out["gos_cb_energy"] = np.complex128(gos.compute_reference_energy())
    out["gos_cb_fock"] = gos.construct_fock_matrix(gos.h, gos.u)
    # f buffer argument: filled in place and returned
    f = np.ones_like(gos.h)
    ret = gos.construct_fock_matrix(gos.h, gos.u, f=f)
    assert ret is f
    # a second, larger closed-shell case with a rectangular (shrinking) change of basis
    np.random.seed(99)
    l2, n2 = 10, 6
    spas2 = qs.SpatialOrbitalSystem(n2, qs.RandomBasisSet(l2, 1))
    out["b_n"], out["b_l"] = np.int64(n2), np.int64(l2)
    out["b_h"], out["b_u"] = spas2.h.copy(), spas2.u.copy()
    out["b_e_nuc"] = np.float64(spas2.nuclear_repulsion_energy)
    Cr = crand(rng, l2, 8)
    out["b_C"] = Cr
    spas2.change_basis(Cr)
    out["b_cb_energy"] = np.complex128(spas2.compute_reference_energy())
    out["b_cb_fock"] = spas2.construct_fock_matrix(spas2.h, spas2.u)
    save("fock_energy_random_basis", **out)


def case_mid_size_sampled():
    """Sizes that run through the streamed and strip kernels (78 ... 180 orbitals): the tensors are too large to commit, so the
    inputs come from a closed integer formula (tests/_lattice_inputs.py) and the fixture holds sampled outputs of the
    reference's transform plus two whole-tensor sums."""
    sys.path.insert(0, os.path.dirname(HERE))
    import _lattice_inputs as li

    arrays = {}
    for name, L, M, ucplx, ccplx, salt in li.CASES:
        u = li.tensor_np(L, salt, ucplx)
        C, Ct = li.case_inputs_np(L, M, ucplx, ccplx, salt)
        out = BasisSet.transform_two_body_elements(u, C, np, C_tilde=Ct)
        del u
        pos = li.sample_positions(M, li.N_SAMPLES, salt)
        arrays[name + "_pos"] = pos
        arrays[name + "_val"] = out[pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3]]
        arrays[name + "_sum"] = np.array(out.sum())
        arrays[name + "_abs_sum"] = np.array(np.abs(out).sum())
        arrays[name + "_max_abs"] = np.array(np.abs(out).max())
        print(name, out.shape, out.dtype, float(arrays[name + "_max_abs"]))
        del out
    save("mid_size_sampled", **arrays)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "midsize":
        case_mid_size_sampled()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sincspin":
        case_sinc_dvr_spin()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fock":
        case_fock_energy()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tdho1":
        case_tdho_one_body()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tdho":
        case_tdho_coulomb()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "odqd":
        case_odqd()
        sys.exit(0)
    case_transforms()
    case_spin_statics()
    case_random_basis_stream()
    case_config1()
    case_gos_small()
    case_change_basis_with_spf()
    case_tdho_coulomb()
    case_odqd()
    case_tdho_one_body()
    case_fock_energy()
    case_sinc_dvr_spin()
    case_mid_size_sampled()
