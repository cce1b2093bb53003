"""bench.py itself (the driver's measuring instrument), run as a child process at a small size:
one JSON line on stdout with every field of the contract, for the transform and for the two
bandwidth workloads."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def run_bench(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *args],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout                        # exactly one JSON line
    return json.loads(lines[0])


def test_transform_line_has_the_contract_fields():
    d = run_bench("--orbitals", "64", "--cpu-l", "48")
    for key in CONTRACT + ["cpu_baseline"]:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "TFLOP/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["randomised_identity_rel_diff"] <= d["parity"]["bound"]
    assert abs(d["value"] - 8 * 64**5 * 3 / (d["ms_per_step"] * 3e-3) / 1e12) < 1e-6 * d["value"]


@pytest.mark.parametrize("workload", ["spin_expand", "antisymmetrize"])
def test_bandwidth_lines(workload):
    d = run_bench("--workload", workload, "--orbitals", "48")
    for key in CONTRACT:
        assert key in d, key
    assert d["unit"] == "GB/s" and d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert d["parity"]["value_exact_vs_definition"] is True


@pytest.mark.parametrize("layout", ["replicated", "sharded"])
def test_two_rank_launch_on_one_device(layout):
    # the N > 1 code path as the driver launches it (torch.distributed.run, one process per rank), rehearsed
    # on ONE GPU: both ranks on cuda:0 over gloo (QS_BENCH_SINGLE_DEVICE / QS_BENCH_BACKEND are rehearsal
    # hooks; the driver's real run uses one GPU per rank and RCCL)
    env = dict(os.environ, QS_BENCH_SINGLE_DEVICE="1", QS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29671" if layout == "replicated" else "29672",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--orbitals", "64",
           "--layout", layout, "--no-cpu-baseline", "--no-probes"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    if layout == "replicated":
        assert d["parity"]["randomised_identity_rel_diff"] <= 1e-10
