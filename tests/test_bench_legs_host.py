"""bench.py at N > 1, host logic only (no GPU): one line composed from the legs measured so far, provisional after every leg,
the final one last; a leg that timed out or was skipped costs only itself (VERDICT r03 "next" 1c)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def leg_line(layout, value, collective, gather=None):
    d = {"metric": "4-index u transform TFLOP/s (fp64) at L=256 orbitals", "value": value, "unit": "TFLOP/s", "n_gpus": 8,
         "n_ranks_seen": 8, "steps": 10, "warmup": 2, "ms_per_step": 8.8e3 / value / 1e3, "ms_per_step_median": 1.0,
         "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
         "config": {"workload": "w", "l": 256, "layout": layout}, "roofline": {"kernel": "k", "frac": 0.5},
         "parity": {"randomised_identity_rel_diff": 1e-15, "bound": 1e-10, "ok": True}, "collective": {"in_value": collective}}
    if gather:
        d["with_all_gather"] = {"value": gather, "ms_per_step": 1.0, "steps": 3}
    return d


def compose(legs, records, provisional, pending):
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.compose_legs(SimpleNamespace(), 8, legs, records, provisional=provisional, pending=pending)
    lines = [json.loads(ln) for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    return rc, lines[0]


def test_provisional_line_after_the_first_leg_is_complete():
    rec = {"replicated": {"status": "ok", "wall_s": 30.0, "line": leg_line("replicated", 430.0, "none", gather=180.0)}}
    rc, line = compose(["replicated"], rec, True, ["rows", "rows_rccl"])
    assert rc == 0 and line["provisional"] is True and line["legs_pending"] == ["rows", "rows_rccl"]
    # the only candidate whose time includes a collective: replicated + the all-gather of the result
    assert line["value"] == 180.0 and line["config"]["chosen_leg"] == "replicated + all-gather"
    assert line["legs"]["replicated"]["value"] == 430.0 and line["parity"]["ok"] is True
    assert {"metric", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"} <= set(line)


def test_final_line_takes_the_best_leg_with_a_collective_and_keeps_the_failed_ones():
    rec = {
        "replicated": {"status": "ok", "wall_s": 30.0, "line": leg_line("replicated", 430.0, "none", gather=180.0)},
        "rows": {"status": "ok", "wall_s": 28.0, "line": leg_line("rows", 380.0, "one all-to-all per chunk of input rows")},
        "rows_rccl": {"status": "failed (exit code 124, timed out after 100 s)", "wall_s": 100.0},
        "rows_rccl_coalesced": {"status": "ok", "wall_s": 25.0, "line": leg_line("rows_rccl", 405.0, "one all-to-all per chunk of input rows")},
        "rccl": {"status": "skipped (out of time: --total-budget)", "wall_s": 0.0},
    }
    legs = list(rec)
    rc, line = compose(legs, rec, False, [])
    assert rc == 0 and "provisional" not in line
    assert line["value"] == 405.0 and line["config"]["chosen_leg"] == "rows_rccl_coalesced"
    assert list(line["legs"]) == legs and line["legs"]["rows_rccl"]["status"].startswith("failed")
    assert line["legs"]["rccl"]["status"].startswith("skipped") and "value" not in line["legs"]["rccl"]


def test_a_leg_without_parity_is_never_the_value():
    bad = leg_line("rows", 900.0, "one all-to-all per chunk of input rows")
    bad["parity"]["ok"] = False
    rec = {"replicated": {"status": "ok", "wall_s": 1.0, "line": leg_line("replicated", 430.0, "none", gather=180.0)},
           "rows": {"status": "ok", "wall_s": 1.0, "line": bad}}
    rc, line = compose(["replicated", "rows"], rec, False, [])
    assert rc == 0 and line["value"] == 180.0 and line["legs"]["rows"]["parity"]["ok"] is False
