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
    d = run_bench("--orbitals", "128", "--cpu-l", "48")
    for key in CONTRACT + ["cpu_baseline"]:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "TFLOP/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["randomised_identity_rel_diff"] <= d["parity"]["bound"] and d["parity"]["ok"] is True
    assert abs(d["value"] - 8 * 128**5 * 3 / (d["ms_per_step"] * 3e-3) / 1e12) < 1e-6 * d["value"]
    # the kernel named in the line is the one the dispatcher took for this size (not a constant)
    assert rf["kernel"] == "qs::gemm_fast_kernel<false, 4, 4, true, false>" and rf["launches_per_step"] == 4
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] and d["n_ranks_seen"] == 1
    pts = {p["l"]: p for p in cb["points"]}
    assert 55 in pts and pts[55]["einsum_optimize_s"] > 0 and pts[55]["tensordot_x4_s"] > 0


def test_transform_line_names_the_small_basis_kernels():
    d = run_bench("--orbitals", "55", "--no-cpu-baseline", "--no-probes")
    # config 2 runs as two fused passes on the 4-wide matrix instruction; the line names that kernel
    assert d["roofline"]["kernel"] == "qs::sandwich4b_kernel<14>" and d["roofline"]["launches_per_step"] == 2
    assert d["parity"]["ok"] is True


def test_complex_time_evolution_pattern():
    # BASELINE.json configs[4] call pattern: u resident, a new complex C(t) every step, C_tilde derived inside
    d = run_bench("--orbitals", "64", "--dtype", "c128", "--no-cpu-baseline", "--no-probes")
    assert d["dtype"] == "c128" and "a new C every step" in d["config"]["workload"] and d["parity"]["ok"] is True
    assert d["roofline"]["kernel"].startswith("qs::gemm_fast_kernel<true")


@pytest.mark.parametrize("workload", ["spin_expand", "antisymmetrize"])
def test_bandwidth_lines(workload):
    d = run_bench("--workload", workload, "--orbitals", "48")
    for key in CONTRACT:
        assert key in d, key
    assert d["unit"] == "GB/s" and d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert d["parity"]["value_exact_vs_definition"] is True


REHEARSAL = dict(QS_BENCH_SINGLE_DEVICE="1", QS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")


def test_two_rank_launch_by_the_external_launcher():
    # the N > 1 code path as the driver launches it (torch.distributed.run, one process per rank), rehearsed
    # on ONE GPU: both ranks on cuda:0 over gloo (QS_BENCH_SINGLE_DEVICE / QS_BENCH_BACKEND are rehearsal
    # hooks; the driver's real run uses one GPU per rank and RCCL)
    env = dict(os.environ, **REHEARSAL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29671",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--orbitals", "64",
           "--layout", "replicated", "--no-cpu-baseline", "--no-probes"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["parity"]["randomised_identity_rel_diff"] <= 1e-10


@pytest.mark.parametrize("layout", ["auto", "replicated", "sharded", "inplace"])
def test_two_rank_self_launch(layout):
    # `python bench.py --gpus 2` with NO external launcher and no WORLD_SIZE: bench.py starts its two rank
    # processes itself (before touching the GPU), waits, relays rank 0's single line -- the form the driver
    # uses when it does not go through torch.distributed.run
    env = dict(os.environ, **REHEARSAL)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--orbitals", "64", "--layout", layout, "--no-cpu-baseline", "--no-probes"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    if layout == "auto":
        # a COMPLETE provisional line after every leg (whatever happens to the legs still to come, the last line on stdout
        # carries everything measured so far), the final line last and without the flag
        every = [json.loads(ln) for ln in lines]
        assert len(every) == 5 and all(e.get("provisional") is True for e in every[:-1]) and "provisional" not in every[-1]
        assert [len(e["legs"]) for e in every] == [1, 2, 3, 4, 5] and every[0]["legs_pending"] == ["rows", "rows_rccl", "rows_rccl_coalesced", "rccl"]
        assert all(e["value"] > 0 and e["parity"]["ok"] for e in every)
    else:
        assert len(lines) == 1, res.stdout[-2000:]                 # exactly one line on the parent's stdout
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["parity"]["ok"] is True and d["parity"]["randomised_identity_rel_diff"] <= 1e-10
    if layout == "replicated":
        assert d["collective"]["in_value"] == "none" and d["with_all_gather"]["value"] > 0
        assert d["with_all_gather"]["value"] <= d["value"] * 1.05
    if layout == "auto":
        # the driver's one command: every layout measured as its own group of rank processes, ONE line; `value` is the
        # best leg whose time includes a collective, and it is named
        legs = d["legs"]
        assert list(legs) == ["replicated", "rows", "rows_rccl", "rows_rccl_coalesced", "rccl"]
        assert legs["replicated"]["status"] == "ok" and legs["rows"]["status"] == "ok"
        assert legs["replicated"]["collective"] == "none" and legs["replicated"]["with_all_gather"]["value"] > 0
        assert "all-to-all per chunk" in legs["rows"]["collective"] and legs["rows"]["parity"]["ok"] is True
        # RCCL driven directly needs one GPU per rank: in the one-device rehearsal those legs are recorded as failed
        # and cost nothing else
        assert all(legs[x]["status"].startswith("failed") for x in ("rows_rccl", "rows_rccl_coalesced", "rccl"))
        assert d["config"]["chosen_leg"] in ("rows", "replicated + all-gather")
        assert d["collective"]["in_value"] != "none"
        best = max(legs["rows"]["value"], legs["replicated"]["with_all_gather"]["value"])
        assert abs(d["value"] - best) <= 1e-9 * best
    if layout == "inplace":
        assert "in place" in d["config"]["layout"] and "per rank" in d["data"]


def test_default_layout_under_the_external_launcher_runs_every_leg():
    # `python -m torch.distributed.run ... bench.py --gpus 2` exactly as the driver issues it (no --layout): every rank
    # process starts one child per leg and never touches the GPU itself; rank 0 prints the ONE composed line
    env = dict(os.environ, **REHEARSAL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29681",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--orbitals", "48",
           "--no-cpu-baseline", "--no-probes", "--legs", "replicated,rows,rows_rccl"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 3, res.stdout[-2000:]                 # a provisional line after each of the first two legs, then the final one
    assert all(json.loads(ln).get("provisional") is True for ln in lines[:-1])
    d = json.loads(lines[-1])
    assert "provisional" not in d
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["parity"]["ok"] is True
    assert d["legs"]["replicated"]["status"] == "ok" and d["legs"]["rows"]["status"] == "ok"
    assert d["legs"]["rows_rccl"]["status"].startswith("failed")
    assert d["config"]["chosen_leg"] in ("rows", "replicated + all-gather") and d["value"] > 0


def test_a_hanging_leg_costs_only_itself_and_the_run_stays_inside_its_budget():
    # VERDICT r03 "next" 1c: the legs that drive RCCL directly have never run on more than one rank before the driver's node.
    # Rehearsal of the worst case: a leg that HANGS (QS_BENCH_HANG_LEG, a test hook: the leg's rank processes sleep instead of
    # measuring).  Its ranks are killed at --leg-timeout, the legs measured before it are already on stdout as a complete
    # provisional line, the legs after it still run, the final line carries all of them, and nothing runs past --total-budget.
    import time

    env = dict(os.environ, **REHEARSAL, QS_BENCH_HANG_LEG="rows_rccl")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--orbitals", "48",
           "--no-cpu-baseline", "--no-probes", "--legs", "replicated,rows_rccl,rows", "--leg-timeout", "20", "--total-budget", "300"]
    t0 = time.monotonic()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    wall = time.monotonic() - t0
    assert res.returncode == 0, res.stderr[-3000:]
    assert wall < 300, wall                       # (inside --total-budget; a warm box takes ~45 s)
    lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 3 and lines[0]["provisional"] is True and list(lines[0]["legs"]) == ["replicated"]
    assert lines[0]["value"] > 0 and lines[0]["parity"]["ok"] is True          # what stands if everything after it is lost
    d = lines[-1]
    assert "provisional" not in d and list(d["legs"]) == ["replicated", "rows_rccl", "rows"]
    assert "timed out" in d["legs"]["rows_rccl"]["status"] and 18 <= d["legs"]["rows_rccl"]["wall_s"] <= 40
    assert d["legs"]["replicated"]["status"] == "ok" and d["legs"]["rows"]["status"] == "ok"
    assert d["config"]["chosen_leg"] in ("rows", "replicated + all-gather") and d["parity"]["ok"] is True


def test_rows_layouts_with_a_one_rank_group():
    # the rows layouts (what ShardedDeviceModule runs) as bench legs: torch.distributed-driven and as ONE C-ABI call on
    # a real RCCL communicator (one rank here)
    for layout, port in (("rows", "29692"), ("rows_rccl", "29693")):
        env = dict(os.environ, QS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--orbitals", "64", "--layout", layout, "--no-cpu-baseline", "--no-probes", "--chunk-rows", "16"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert res.returncode == 0, res.stderr[-3000:]
        d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")][0])
        assert "rows in, rows out" in d["config"]["layout"] and d["parity"]["ok"] is True and d["n_ranks_seen"] == 1
        if layout == "rows_rccl":
            assert "rccl grouped send/recv (4 steps of 16 rows)" in d["roofline"]["dispatch"]


def test_config_presets_and_the_mixed_dtype():
    # --config 3 / 4: BASELINE.json configs[3] and [4] without flag knowledge
    d = run_bench("--config", "3", "--steps", "1", "--warmup", "0")
    assert "configs[3]" in d["config"]["workload"] and d["config"]["l"] == 256 and d["unit"] == "GB/s"
    d = run_bench("--config", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-probes")
    # one GPU holds input rows + result rows of l = 256 complex128 (l = 512 needs the 8-GPU node)
    assert d["config"]["l"] == 256 and d["dtype"] == "c128" and "a new C every step" in d["config"]["workload"]
    assert "rows in, rows out" in d["config"]["layout"] and d["parity"]["ok"] is True
    # --dtype mixed: a real u against complex coefficients, natively and through the complex copy of rounds 1-2
    a = run_bench("--orbitals", "64", "--dtype", "mixed", "--no-cpu-baseline", "--no-probes")
    b = run_bench("--orbitals", "64", "--dtype", "mixed", "--mixed-route", "cast", "--no-cpu-baseline", "--no-probes")
    for d in (a, b):
        assert d["parity"]["ok"] is True and d["config"]["flops_per_step"] == 28 * 64**5
    assert "read as it is" in a["config"]["workload"] and "complex copy" in b["config"]["workload"]
    assert "<false" in a["roofline"]["dispatch"] and "<false" not in b["roofline"]["dispatch"]


def test_self_launch_bandwidth_workload_two_ranks():
    env = dict(os.environ, **REHEARSAL)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--orbitals", "32", "--workload", "spin_expand"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert d["n_gpus"] == 2 and d["parity"]["value_exact_vs_definition"] is True
    assert d["roofline"]["kernel"] == "qs::spin_expand_kernel<double, f64x2>"


def test_rccl_layout_with_a_one_rank_group():
    # --layout rccl: the whole sharded step is one C-ABI call on a real RCCL communicator (one rank here; the
    # driver's node gives it one GPU per rank)
    env = dict(os.environ, QS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29691",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--orbitals", "64", "--layout", "rccl", "--no-cpu-baseline", "--no-probes"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert "RCCL" in d["config"]["layout"] and d["parity"]["ok"] is True and d["n_ranks_seen"] == 1


def test_failed_parity_exits_nonzero(monkeypatch):
    # the parity bound is enforced, not just printed: an impossible bound makes the run fail
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "PARITY_BOUND", 0.0)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = bench.parse(["--orbitals", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-probes"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.run_rank(args) == 3
