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


def test_one_rank_of_many_with_absent_peers_tool():
    # tools/config4_one_rank.py: one rank's share of a sharded transform as rank r of G through the C entry, the peers replaced by
    # a stand-in that drops sends and delivers zeros (tests/cabi/absent_peers_rccl.cpp) -- the tool that ran configs[4]'s rank at
    # its size (profiles/r04_config4_one_rank_of_8_at_size.txt).  Here small: the result must be the transform of the tensor whose
    # only non-zero rows are the rank's (the tool's own check), both dtypes, uneven split.
    import json
    import os
    import shutil
    import subprocess
    import sys

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["--orbitals", "26", "--world", "4", "--rank", "2"], ["--orbitals", "21", "--world", "8", "--rank", "7", "--dtype", "f64"]):
        env = {k: v for k, v in os.environ.items() if k != "QS_AMD_RCCL_LIB"}
        res = subprocess.run([sys.executable, os.path.join(root, "tools", "config4_one_rank.py"), "--steps", "1"] + args,
                             capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["parity_ok"] and line["max_rel_err_sampled_planes"] <= 1e-12
        assert line["exchange_gb_per_step"]["sent"] >= 0 and line["rows_in"] >= 1
