"""The sharded array module with the REAL engine (libqs_amd.so): the API flow of
tests/_sharded_api_worker.py -- SpatialOrbitalSystem -> GeneralOrbitalSystem -> change_basis -> Fock / energy
on slabs, against the reference's tensors -- with one rank, and with two and three ranks rehearsed on this one
GPU (every rank on cuda:0, gloo carrying the collectives; on a node each rank has its own GPU and RCCL)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_api_flow_with_the_hip_engine(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", QS_WORKER_ENGINE="hip")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
        f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
        "--master-port", str(29570 + world), os.path.join(ROOT, "tests", "_sharded_api_worker.py"),
    ]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(" ok") == world


def test_sharded_example_agrees_with_the_single_gpu_example():
    # examples/sharded_quantum_dot.py with 1 and 3 ranks on this one GPU: the reference energy is the same number
    def run(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", QS_EXAMPLE_ONE_DEVICE="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(29580 + world),
               os.path.join(ROOT, "examples", "sharded_quantum_dot.py"), "5"]
        res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("reference energy")][0]
        return float(line.split()[2].rstrip(","))

    e1, e3 = run(1), run(3)
    assert abs(e1 - e3) <= 1e-10 * abs(e1)
