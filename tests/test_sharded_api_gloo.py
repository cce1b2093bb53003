"""The sharded array module behind the reference's API (BASELINE.json configs[3] as named:
SpatialOrbitalSystem -> GeneralOrbitalSystem, u as one slab per rank), multi-process on CPU over gloo with
world sizes 1, 2 and 3 (uneven slabs), against the reference's own tensors."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_api_flow_under_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
        f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
        "--master-port", str(29560 + world), os.path.join(ROOT, "tests", "_sharded_api_worker.py"),
    ]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(" ok") == world
