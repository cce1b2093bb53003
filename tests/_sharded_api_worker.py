"""Worker for tests/test_sharded_api_gloo.py (torch.distributed.run, gloo, CPU): the reference's API flow
SpatialOrbitalSystem -> GeneralOrbitalSystem -> change_basis -> Fock matrix / reference energy with the
SHARDED array module, every rank holding one slab of the rank-4 tensors, against tensors the reference
itself produced (tests/golden/gos_l5_default_spinors.npz, fock_energy_random_basis.npz).

The slab-local arithmetic is done by an oracle-backed stand-in for the HIP engine (no GPU here); what is
under test is everything around it: which rows a rank keeps, the exchange inside the transforms, the
flip of the sharded index, the elided copy of u, the lazy rows of spin_2_tb, the reductions."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _dist_worker import OracleEngine as _Products  # noqa: E402
from oracle import qs_oracle as orc  # noqa: E402
import quantum_systems_amd as qsa  # noqa: E402
from quantum_systems_amd import sharded  # noqa: E402


HIP = os.environ.get("QS_WORKER_ENGINE") == "hip"      # real HIP engine, every rank on cuda:0 (one-GPU rehearsal)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    """Host NumPy copy of a tensor / device array / NumPy array."""
    if isinstance(t, np.ndarray):
        return t
    return torch.as_tensor(t).detach().cpu().resolve_conj().numpy()


class OracleEngine(_Products):
    """CPU test double with the full interface of sharded.HipEngine."""

    @staticmethod
    def transform_one_body(h, C, Ct):
        hn, Cn, Ctn = h.numpy(), C.numpy(), Ct.numpy()
        if hn.ndim == 2:
            return _t(orc.transform_one_body(hn, Cn, Ctn))
        return _t(np.asarray([orc.transform_one_body(m, Cn, Ctn) for m in hn]))

    @staticmethod
    def add_spin_one_body(h, out_dtype=None):
        hn = h.numpy()
        out = orc.add_spin_one_body(hn) if hn.ndim == 2 else np.asarray([orc.add_spin_one_body(m) for m in hn])
        return _t(out).to(out_dtype or h.dtype)

    @staticmethod
    def spin_expand_block(u_block, antisymmetrize=False, out_dtype=None):
        out = orc.add_spin_two_body(u_block.numpy())          # kron with the delta tensor: any leading extents
        if antisymmetrize:
            out = orc.anti_symmetrize_u(out)
        return _t(out).to(out_dtype or u_block.dtype)

    @staticmethod
    def antisymmetrize(u_block, in_place=False):
        return _t(orc.anti_symmetrize_u(u_block.numpy()))

    @staticmethod
    def spin_squared_two_body(S, antisymmetrize=False, p_lo=0, p_hi=None):
        Sn = S.numpy()
        tb = sum(np.einsum("pr,qs->pqrs", Sn[k], Sn[k]) for k in range(3))      # basis_set.py:745-747
        if antisymmetrize:
            tb = orc.anti_symmetrize_u(tb)
        return _t(tb[p_lo:p_hi])


def golden(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def rows_of(full, t):
    """What this rank must hold of the whole reference tensor."""
    return full[t.lo:t.hi] if t.axis == 0 else full[:, t.lo:t.hi]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if HIP:
        torch.cuda.set_device(0)
        mod = qsa.ShardedDeviceModule(rank, world, device="cuda:0")             # engine: libqs_amd.so
    else:
        mod = qsa.ShardedDeviceModule(rank, world, device="cpu", engine=OracleEngine)

    # ---- SpatialOrbitalSystem l=5 -> GeneralOrbitalSystem l=10 -> change_basis 10 -> 8 (reference fixture)
    g = golden("gos_l5_default_spinors")
    bs = qsa.BasisSet(5, 2, np=np)
    bs.h, bs.s, bs.u, bs.position = g["in_h"], g["in_s"], g["in_u"], g["in_position"]
    bs.nuclear_repulsion_energy = float(g["in_nuclear_repulsion_energy"]) if "in_nuclear_repulsion_energy" in g else 0
    spas = qsa.SpatialOrbitalSystem(4, bs)
    spas.change_module(mod)                                   # u becomes this rank's slab; the rest is replicated
    u0 = spas.u
    assert isinstance(u0, qsa.ShardedTensor4) and u0.axis == 0 and tuple(u0.shape) == (5, 5, 5, 5)
    assert np.array_equal(N(u0.local), g["in_u"][u0.lo:u0.hi])
    gos = spas.construct_general_orbital_system()
    assert spas.u is u0 and spas.l == 5 and not spas._basis_set.includes_spin     # original intact, u not copied
    gb = gos._basis_set
    assert gos.l == 10 and gos.n == int(g["n_gos"]) and gb.includes_spin and gb.anti_symmetrized_u
    ug = gos.u
    # spin rows 2p, 2p+1 stay where spatial row p is: 5 rows over 2 ranks = 3+2 -> 6+4 (not the balanced 5+5)
    assert ug.axis == 0 and ug.dtype == torch.complex128 and (ug.lo, ug.hi) == (2 * u0.lo, 2 * u0.hi)
    for k in ("h", "s", "position", "spin_x", "spin_y", "spin_z", "sigma_x", "sigma_y", "sigma_z"):
        assert np.array_equal(N(getattr(gb, k)), g["gos_" + k]), k
    assert np.array_equal(N(ug.local), g["gos_u"][ug.lo:ug.hi])            # value-exact scatter
    np.testing.assert_allclose(N(gb.spin_2), g["gos_spin_2"], rtol=1e-13, atol=1e-14)
    s2 = gos.spin_2_tb                                                           # built lazily: this rank's rows only
    np.testing.assert_allclose(N(s2.local), rows_of(g["gos_spin_2_tb"], s2), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(N(ug.gather()), g["gos_u"], rtol=0, atol=0)  # the all-gather of the slabs

    gos.change_basis(mod.asarray(g["C"]))                     # u AND spin_2_tb transformed (:374-382)
    assert gos.l == int(g["l_after"])
    # ... spin_2_tb as its recipe (three transformed spin matrices): no second all-to-all, no second slab; the rows
    # read above were dropped and are rebuilt on access
    assert gb._spin_2_tb is None and tuple(gb._spin_2_tb_recipe[0].shape) == (3, 8, 8)
    uc = gos.u
    assert uc.axis == 1                                       # one all-to-all: the sharded index flipped
    np.testing.assert_allclose(N(uc.local), rows_of(g["cb_u"], uc), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(gos.spin_2_tb.local), rows_of(g["cb_spin_2_tb"], gos.spin_2_tb), rtol=1e-10, atol=1e-12)
    for k in ("h", "s", "position"):
        np.testing.assert_allclose(N(getattr(gb, k)), g["cb_" + k], rtol=1e-10, atol=1e-12)
    for k in ("spin_x", "spin_y", "spin_z", "spin_2"):        # left alone by the reference (:368-372)
        np.testing.assert_allclose(N(getattr(gb, k)), g["cb_" + k], rtol=1e-13, atol=1e-14)
    back = uc.reshard(0)                                      # and back to leading-index slabs
    np.testing.assert_allclose(N(back.local), rows_of(g["cb_u"], back), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(N(uc.gather()), g["cb_u"], rtol=1e-10, atol=1e-12)
    # leaving the sharded module gathers
    gos.change_module(np)
    np.testing.assert_allclose(gos.u, g["cb_u"], rtol=1e-10, atol=1e-12)

    # ---- Fock matrix / reference energy through the system classes, both shardings (reference values)
    f = golden("fock_energy_random_basis")
    l, n = int(f["l"]), int(f["n"])
    spas = qsa.construct_custom_system(n, l, f["s"], f["h"], f["u"], dim=2, np=np, system_type="spatial",
                                       nuclear_repulsion_energy=float(f["e_nuc"]))
    spas.change_module(mod)
    np.testing.assert_allclose(complex(spas.compute_reference_energy()), f["spas_energy"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(N(spas.construct_fock_matrix(spas.h, spas.u)), f["spas_fock"],
                               rtol=1e-12, atol=1e-12)
    gos = spas.construct_general_orbital_system()
    np.testing.assert_allclose(complex(gos.compute_reference_energy()), f["gos_energy"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(N(gos.construct_fock_matrix(gos.h, gos.u)), f["gos_fock"],
                               rtol=1e-12, atol=1e-12)
    spas.change_basis(mod.asarray(f["C"]))                    # leading -> second index
    assert spas.u.axis == 1
    np.testing.assert_allclose(complex(spas.compute_reference_energy()), f["spas_cb_energy"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(N(spas.construct_fock_matrix(spas.h, spas.u)), f["spas_cb_fock"],
                               rtol=1e-11, atol=1e-11)
    gos.change_basis(mod.asarray(f["C_gos"]))
    np.testing.assert_allclose(complex(gos.compute_reference_energy()), f["gos_cb_energy"], rtol=1e-11, atol=1e-11)
    buf = mod.zeros_like(gos.h) + 1
    assert gos.construct_fock_matrix(gos.h, gos.u, f=buf) is buf
    np.testing.assert_allclose(N(buf), f["gos_cb_fock"], rtol=1e-11, atol=1e-11)
    spas.change_basis(mod.asarray(np.eye(l)))                 # second -> leading index again (identity rotation)
    assert spas.u.axis == 0
    np.testing.assert_allclose(complex(spas.compute_reference_energy()), f["spas_cb_energy"], rtol=1e-11, atol=1e-11)
    # the functional per-step call of a solver (system.py:222-225) on the resident sharded u
    out = spas.transform_two_body_elements(spas.u, mod.asarray(f["C"]))
    ref = orc.transform_two_body(N(spas.u.gather()), f["C"])
    np.testing.assert_allclose(N(out.local), rows_of(ref, out), rtol=1e-10, atol=1e-11)
    # element-wise algebra of u_t (system.py:206-215; operator.py:193-196 scales u)
    twice = 2.0 * spas.u + spas.u * 0.5
    np.testing.assert_allclose(N(twice.local), 2.5 * N(spas.u.local), rtol=1e-14, atol=0)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
