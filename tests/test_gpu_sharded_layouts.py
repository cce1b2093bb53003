"""The sharded layouts driven by the HIP engine on one GPU (world = 1 and the
per-rank pieces of larger worlds, where no collective is needed): strided
batched products on sub-blocks, exchange and last contraction inside the output buffer."""

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cplx", [False, True])
def test_inplace_layout_world1_matches_oracle(cplx):
    from quantum_systems_amd import sharded

    rng = np.random.default_rng(5 + cplx)
    l = 20
    u = rng.standard_normal((l,) * 4)
    C = rng.standard_normal((l, l))
    Ct = rng.standard_normal((l, l))
    if cplx:
        u = u + 1j * rng.standard_normal((l,) * 4)
        C = C + 1j * rng.standard_normal((l, l))
        Ct = Ct + 1j * rng.standard_normal((l, l))
    ref = orc.transform_two_body(u, C, Ct)
    du = torch.from_numpy(u).cuda()
    out = sharded.transform_two_body_sharded_inplace(du, torch.from_numpy(C).cuda(), torch.from_numpy(Ct).cuda())
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
    # the resident slab is untouched: everything happens inside the output buffer
    assert np.array_equal(du.cpu().numpy(), u)
    # the buffer reused across steps (time loop: u resident, new C every step)
    keep = torch.empty((l + 1, l, l, l), dtype=du.dtype, device="cuda")
    for _ in range(2):
        again = sharded.transform_two_body_sharded_inplace(
            du, torch.from_numpy(C).cuda(), torch.from_numpy(Ct).cuda(), out=keep)
        assert again.data_ptr() == keep.data_ptr() and torch.equal(again, out)
    with pytest.raises(ValueError):
        sharded.transform_two_body_sharded_inplace(torch.from_numpy(u).cuda(), torch.from_numpy(C[:, :10].copy()).cuda())


def test_out_of_place_sharded_layout_world1(golden=None):
    from quantum_systems_amd import sharded

    rng = np.random.default_rng(9)
    L, M = 14, 9
    u = rng.standard_normal((L,) * 4) + 1j * rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, M)) + 1j * rng.standard_normal((L, M))
    ref = orc.transform_two_body(u, C)
    out = sharded.transform_two_body_sharded(torch.from_numpy(u).cuda(), torch.from_numpy(C).cuda())
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-10 * np.abs(ref).max()


def test_sharded_fock_and_energy_match_the_system_classes():
    # the first consumers of a transformed, p-sharded u.  Rows / partial sums
    # computed per (emulated) rank from its slab reproduce what the system classes compute
    # from the whole tensor (spatial_orbital_system.py:106-190, general_orbital_system.py:75-159)
    import quantum_systems_amd as qsa
    from quantum_systems_amd import hip, sharded

    np.random.seed(31)
    spas = qsa.SpatialOrbitalSystem(4, qsa.RandomBasisSet(10, 2))
    spas.change_module(hip)
    gos = spas.construct_general_orbital_system()
    for system, spin in ((spas, False), (gos, True)):
        h, u = torch.as_tensor(system.h), torch.as_tensor(system.u)
        f_ref = torch.as_tensor(system.construct_fock_matrix(system.h, system.u))
        e_ref = complex(torch.as_tensor(system.compute_reference_energy()).cpu())
        for world in (1, 2, 4):
            rows = []
            for rank in range(world):
                lo, hi = sharded.SlabPartition(system.l, world).bounds(rank)
                rows.append(sharded.fock_rows(h, u[lo:hi], system.n, lo, spin_orbitals=spin))
            f = torch.cat(rows)
            assert (f - f_ref).abs().max().item() <= 1e-12 * max(1.0, f_ref.abs().max().item())
        e = complex(sharded.reference_energy_sharded(
            h, u, system.n, spin_orbitals=spin,
            nuclear_repulsion_energy=system.nuclear_repulsion_energy).cpu())
        assert abs(e - e_ref) <= 1e-10 * abs(e_ref)
