"""The multi-GPU entry points of the C ABI with real RCCL, as far as a one-GPU box goes: a one-rank communicator
(qs_comm_unique_id / qs_comm_init / qs_comm_destroy) and qs_transform_two_body_sharded through it -- local
contractions, the chunked exchange on the communicator's stream (own rows only at world = 1), closing contraction --
against the oracle.  The multi-rank index logic is replayed on the CPU (tests/test_sharded_plan.py)."""

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    from quantum_systems_amd import kernels as K

    uid = K.RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    with K.RcclComm(0, 1, uid) as c:
        yield c


@pytest.mark.parametrize("L,M,cplx,nchunks", [(12, 12, False, 4), (14, 9, True, 3), (9, 14, False, 1), (64, 64, False, 4),
                                               (20, 20, True, 16)])
def test_sharded_transform_through_the_cabi_one_rank(comm, L, M, cplx, nchunks):
    rng = np.random.default_rng(L * 100 + M)
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, M)) / np.sqrt(L)
    Ct = rng.standard_normal((M, L)) / np.sqrt(L)
    if cplx:
        u = u + 1j * rng.standard_normal((L,) * 4)
        C = C + 1j * rng.standard_normal((L, M)) / np.sqrt(L)
        Ct = Ct + 1j * rng.standard_normal((M, L)) / np.sqrt(L)
    ref = orc.transform_two_body(u, C, Ct)
    du, dC, dCt = (torch.from_numpy(a).cuda() for a in (u, C, Ct))
    out = comm.transform_two_body(du, dC, dCt, nchunks=nchunks)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
    # again into a caller-supplied buffer, back to back on the same stream (workspace and R reused)
    buf = torch.empty_like(out)
    for _ in range(3):
        comm.transform_two_body(du, dC, dCt, out=buf, nchunks=nchunks)
    assert torch.equal(buf, out)
    assert np.array_equal(du.cpu().numpy(), u)                     # the resident slab is untouched


def test_comm_argument_errors(comm):
    from quantum_systems_amd import _lib, kernels as K

    lib = _lib.load()
    assert lib.qs_comm_rank(comm._handle) == 0 and lib.qs_comm_world(comm._handle) == 1
    assert lib.qs_transform_two_body_sharded_workspace(0, 8, 8, 2, 2) < 0          # rank outside the world
    assert lib.qs_transform_two_body_sharded_workspace(7, 8, 8, 1, 0) < 0          # bad dtype
    u = torch.zeros((8, 8, 8, 8), dtype=torch.float64, device="cuda")
    C = torch.eye(8, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        comm.transform_two_body(u[:, :4].contiguous(), C)                           # not this rank's slab
    tiny = torch.empty(16, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(u)
    rc = lib.qs_transform_two_body_sharded(comm._handle, 0, u.data_ptr(), C.data_ptr(), C.data_ptr(), out.data_ptr(),
                                           tiny.data_ptr(), 16, 8, 8, 4, torch.cuda.current_stream().cuda_stream)
    assert rc == -4                                                                 # QS_ERR_WORKSPACE
    with pytest.raises(ValueError):
        K.RcclComm(0, 1, b"short")
