"""Multi-process CPU tests of the sharded transform (gloo, world_size 2 and 3):
partition arithmetic, the replicated layout, the all-gather of the result and
the one-all-to-all layout, each against the oracle's full transform."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_balanced_and_covers():
    from quantum_systems_amd.sharded import SlabPartition

    for n in (1, 5, 8, 20, 256, 257):
        for w in (1, 2, 3, 8):
            part = SlabPartition(n, w)
            counts = [part.count(r) for r in range(w)]
            assert sum(counts) == n and max(counts) - min(counts) <= 1
            assert part.bounds(0)[0] == 0 and part.bounds(w - 1)[1] == n
            for r in range(w - 1):
                assert part.bounds(r)[1] == part.bounds(r + 1)[0]


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_transform_under_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
        f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
        "--master-port", str(29540 + world), os.path.join(ROOT, "tests", "_dist_worker.py"),
    ]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(" ok") == world
