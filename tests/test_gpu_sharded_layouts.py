"""The sharded layouts driven by the HIP engine on one GPU (world = 1 and the
per-rank pieces of larger worlds, where no collective is needed): strided
batched products on sub-blocks, in-place overwrite of the input slab."""

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
    # the input slab was consumed (overwritten by the intermediate X)
    assert not np.array_equal(du.cpu().numpy(), u)
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
