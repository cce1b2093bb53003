"""The MULTI-RANK branches of the C ABI's sharded entry points, executed: several rank processes on this one GPU, the
library's ncclSend / ncclRecv calls carried by a file-based stand-in for librccl (tests/cabi/mock_rccl.cpp: same pairing
and size rules as RCCL, no asynchrony).  Each rank is the plain C++ program tests/cabi/sharded_ranks_demo.cpp (no Python,
no torch in the process, so the library's dlopen finds the stand-in): qs_transform_two_body_sharded_rows for leading-index
rows (balanced and spin-doubled partitions: bit-identical to the single-GPU transform), second-index rows, a real tensor
against complex coefficients, and the round-2 entry qs_transform_two_body_sharded.  What stays unrun until the driver's
8-GPU node: real RCCL and the overlap of its streams."""

import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    libdir = os.path.join(ROOT, "quantum-systems_amd")
    assert os.path.exists(os.path.join(libdir, "libqs_amd.so")), "build the library first (__graft_entry__.build)"
    d = tmp_path_factory.mktemp("mock_rccl")
    mock = d / "librccl.so.1"
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
                    os.path.join(ROOT, "tests", "cabi", "mock_rccl.cpp"), "-o", str(mock)], check=True, capture_output=True, timeout=300)
    exe = d / "sharded_ranks_demo"
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi", "sharded_ranks_demo.cpp"), "-L", libdir, "-l:libqs_amd.so",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True, capture_output=True, timeout=300)
    return d, str(exe)


@pytest.mark.parametrize("world,L,M,mode,chunk", [
    (2, 12, 12, 0, 2), (3, 14, 9, 1, 1), (2, 10, 14, 1, 3), (3, 16, 16, 2, 0), (4, 18, 18, 0, 1), (2, 40, 40, 1, 0),
    (5, 6, 6, 0, 1),              # more ranks than some partitions have rows (2 x 3 spatial rows doubled: ranks with no rows)
])
def test_sharded_entry_points_with_several_ranks_on_one_device(built, world, L, M, mode, chunk):
    d, exe = built
    run_dir = d / f"run_{world}_{L}_{M}_{mode}_{chunk}"
    run_dir.mkdir()
    env = dict(os.environ, LD_LIBRARY_PATH=f"{d}:" + os.environ.get("LD_LIBRARY_PATH", ""), QS_MOCK_RCCL_DIR=str(run_dir))
    idfile = str(run_dir / "unique_id")
    procs = [subprocess.Popen([exe, str(r), str(world), idfile, str(L), str(M), str(mode), str(chunk)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                  # exactly the processes started above
    text = "".join(outs)
    assert all(p.returncode == 0 for p in procs), text
    assert text.count("RANK_OK") == world, text
    assert "mock rccl" not in text, text          # no complaint of the transport (sizes of every pair agreed)


def test_bench_rccl_legs_with_two_ranks_over_the_stand_in(built):
    # bench.py's two RCCL legs (the whole step ONE C-ABI call per rank), two ranks on this one GPU: the unique id broadcast
    # over the job's process group, the communicator, the timed steps, the parity property summed over the ranks
    import json
    import sys

    d, _ = built
    for layout in ("rows_rccl", "rows_rccl_coalesced", "rccl"):
        run_dir = d / f"bench_{layout}"
        run_dir.mkdir()
        env = dict(os.environ, QS_BENCH_SINGLE_DEVICE="1", QS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
                   QS_AMD_RCCL_LIB=str(d / "librccl.so.1"), QS_MOCK_RCCL_DIR=str(run_dir))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--orbitals", "48", "--layout", layout, "--no-cpu-baseline", "--no-probes", "--chunk-rows", "6"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
        line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")][0])
        assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["parity"]["ok"] is True
        assert "rccl grouped send/recv" in line["roofline"]["dispatch"]
        assert ("one message per peer and step" in line["roofline"]["dispatch"]) == (layout == "rows_rccl_coalesced")


def test_sharded_module_through_the_c_entry_with_three_ranks(built):
    # ShardedDeviceModule(exchange="rccl"): the API-level sharded transform is ONE C-ABI call per tensor on every rank
    # (tests/_rows_worker.py with QS_ROWS_WORKER_RCCL=1: three ranks on this GPU, gloo for the process group)
    import sys

    d, _ = built
    run_dir = d / "module_rccl"
    run_dir.mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", QS_ROWS_WORKER_RCCL="1",
               QS_AMD_RCCL_LIB=str(d / "librccl.so.1"), QS_MOCK_RCCL_DIR=str(run_dir))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", "29599", os.path.join(ROOT, "tests", "_rows_worker.py")]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(" ok") == 3 and res.stdout.count("module on the C entry") == 3
