"""Headline benchmark: four-index transform of the two-body integrals.

    python bench.py [--gpus N --steps K --warmup W] [--orbitals 256] [--dtype f64|c128|mixed] [--config 2|3|4]
                    [--layout auto|replicated|rows|rows_rccl|sharded|inplace|rccl] [--gather] [--fresh-c auto|on|off]
                    [--workload transform|spin_expand|antisymmetrize] [--no-cpu-baseline]

Metric (BASELINE.json): "4-index u transform TFLOP/s (fp64) at L orbitals".
A step is ONE full transform out = Ct Ct u C C of a synthetic RandomBasisSet-
shaped tensor resident in HBM (BASELINE.json configs[2]: real fp64, l=256,
u = 34.4 GB, C = real orthogonal).  The work figure is the algorithmic
8 l^5 flops (32 l^5 for complex128), whatever the kernels actually do.

N > 1: one process per GPU over RCCL.  Either the driver launches the ranks
(`python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE come
from the environment) or `python bench.py --gpus N` alone: then THIS process
starts the N rank processes itself -- fresh children, before anything here has
touched the GPU or imported torch -- waits for them and relays rank 0's JSON
line.  The SAME l=256 problem at every N ("strong" scaling), result sharded
over its leading index p.  Layouts:
  replicated  u resident on every GPU, no collective on the data path (default
              when u fits: 34 GB of 288);
  sharded     u sharded over its second index, one all-to-all, out of place;
  inplace     the same exchange inside the output buffer: input slab + output
              slab + O(l^3) per GPU -- the form BASELINE.json configs[4]
              (l=512 complex128, 1.1 TB) needs; slabs are generated per rank;
  rccl        the sharded layout in ONE C-ABI call per step (qs_transform_two_body_sharded:
              RCCL driven directly, grouped send/recv per peer, chunked exchange on its own
              stream overlapped with the products);
  rows        what ShardedDeviceModule does behind the API: rows of one leading index in,
              rows of the other out, streamed (input rows + result rows + O(l^3) per GPU),
              one all-to-all per chunk of rows through torch.distributed;
  rows_rccl   the same as ONE C-ABI call (qs_transform_two_body_sharded_rows), the exchange
              on the communicator's stream under the next chunk's products: one message per
              (peer, result row), landing in place;
  rows_rccl_coalesced   the same call with ONE message per peer and step (the handle's option
              "rows_coalesce": staging area + one strided copy on the communicator's stream).
The north star's single all-gather (replicating the p-sharded result) is timed
as a second leg and reported next to the no-collective value; `--gather` makes
it part of `value`.

`--layout auto` at N > 1 (what the driver's one command runs) measures EVERY layout that
fits, one after the other, each as its own group of fresh rank processes (a leg that fails or
hangs is recorded and killed, the others stand): `legs` holds each leg's TFLOP/s, ms/step and
parity.  After EVERY leg rank 0 prints a complete line composed from the legs measured so far
(`"provisional": true`); the final line comes last.  A leg is killed after --leg-timeout
seconds and nothing runs past --total-budget, so a hanging leg can neither take the measured
legs with it nor push the run over the driver's limit.  `value` is the best leg whose time INCLUDES a collective (replicated + all-gather of
the result, or a sharded layout with its all-to-all), and `config.layout` names it.
`--config 3` = BASELINE.json configs[3] (spin expansion l=256 -> 512), `--config 4` =
configs[4] (l=512 complex128, slabs generated per rank, rows layout, a new C(t) per step;
below 8 GPUs the largest l whose two slabs fit).

Prints ONE JSON line on rank 0; exits non-zero if the parity property fails.
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F64_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz (MI355X_MICROARCH.md clocks)
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)
PARITY_BOUND = 1e-10          # BASELINE.json north_star: <= 1e-10 relative fp64
HBM_BYTES = 288 * 2**30      # hipMemGetInfo on the box: 309.2e9 bytes


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--orbitals", "-l", dest="l", type=int, default=256,
                    help="number of orbitals l (named so that torch.distributed.run does not read it as --l*)")
    ap.add_argument("--dtype", choices=["f64", "c128", "mixed"], default="f64",
                    help="mixed = real fp64 u against complex128 coefficients (the time-propagation call on a real "
                         "quantum-dot u): complex128 result")
    ap.add_argument("--mixed-route", choices=["native", "cast"], default="native",
                    help="native: the real tensor is read as it is (qs_transform_two_body_mixed); cast: rounds 1-2, a "
                         "complex copy of the tensor first (A/B)")
    ap.add_argument("--layout", choices=["auto", "replicated", "sharded", "inplace", "rccl", "rows", "rows_rccl",
                                         "rows_rccl_coalesced"], default="auto")
    ap.add_argument("--config", type=int, choices=[2, 3, 4], default=None,
                    help="preset: BASELINE.json configs[2] (default), [3] spin expansion l=256 -> 512, "
                         "[4] l=512 complex128 time-evolution pattern on per-rank slabs")
    ap.add_argument("--legs", default="replicated,rows,rows_rccl,rows_rccl_coalesced,rccl",
                    help="layouts measured by --layout auto at N > 1, in this order (the torch.distributed legs first: the "
                         "legs that drive RCCL directly have never run on more than one rank before the driver's node)")
    ap.add_argument("--leg-timeout", type=float, default=100.0,
                    help="seconds before a leg is given up and killed (the first leg gets twice that)")
    ap.add_argument("--total-budget", type=float, default=520.0,
                    help="seconds from the start of this process after which no leg is started or kept alive: the line of "
                         "the legs measured so far stands (the driver gives a run 600 s)")
    ap.add_argument("--chunk-rows", type=int, default=0, help="input rows per exchange step of the rows layouts (0 = automatic)")
    ap.add_argument("--gather", action="store_true", help="make the all-gather of the result part of `value`")
    ap.add_argument("--no-gather-leg", action="store_true", help="skip the second, gather-inclusive timing leg at N > 1")
    ap.add_argument("--fresh-c", choices=["auto", "on", "off"], default="auto",
                    help="a new coefficient matrix every step, C_tilde derived inside the call (the per-step caller "
                         "system.py:222-225); auto = on for complex128 (time evolution), off for fp64 (one change_basis)")
    ap.add_argument("--staging-rows", type=int, default=1, help="rows per all-to-all of the in-place exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-l", type=int, default=192, help="largest size of the CPU-baseline sample")
    ap.add_argument("--cpu-full", choices=["auto", "on", "off"], default="auto", nargs="?", const="on",
                    help="time the CPU baseline at the FULL l too (one repetition, ~100 s and ~105 GB of host memory at l = 256): "
                         "auto = when the host has >= 110 GB available (SURVEY 8d)")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--workload", choices=["transform", "spin_expand", "antisymmetrize"], default="transform",
                    help="transform = the headline metric; the other two are the HBM-bound kernels of "
                         "BASELINE.json configs[3] (own metric names, same JSON shape)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without an external launcher
# ----------------------------------------------------------------------------------------------

def launch_group(n, argv, timeout=None, extra_env=None):
    """Start the N rank processes of ONE group (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as
    torch.distributed.run would set them), wait (at most `timeout` seconds), return (exit code, rank 0's stdout
    lines).  This parent never imports torch and never touches the GPU; the children are fresh processes (no exec
    of a GPU-initialised process).  Only the PIDs started here are ever signalled."""
    import socket
    import threading

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), QS_BENCH_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update(extra_env or {})
    procs, rank0_out = [], []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))

    def drain():
        for ln in procs[0].stdout:
            rank0_out.append(ln)

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    failed = 0
    live = set(range(n))
    t0 = time.monotonic()
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and not failed:
                failed = rc
                print(f"bench.py: rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()          # exactly the PIDs started above
        if live and timeout is not None and time.monotonic() - t0 > timeout:
            print(f"bench.py: group not finished after {timeout:.0f} s; stopping its ranks", file=sys.stderr)
            failed = failed or 124
            for o in live:
                procs[o].kill()
            timeout = None
        time.sleep(0.05)
    reader.join(timeout=10)
    return failed, rank0_out


def self_launch(n, argv=None):
    """`python bench.py --gpus N` without an external launcher: one group, rank 0's line relayed."""
    failed, rank0_out = launch_group(n, sys.argv[1:] if argv is None else argv)
    for ln in rank0_out:
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
    sys.stdout.flush()
    return failed


def _without_option(argv, name, has_value=True):
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == name:
            skip = has_value
            continue
        if a.startswith(name + "="):
            continue
        out.append(a)
    return out


T_START = time.monotonic()


def warm_import(timeout):
    """Page the image's torch in (1-2 minutes on a fresh box, seconds afterwards) in a child of its own, so that the first
    leg's clock measures the leg.  Never touches the GPU."""
    try:
        subprocess.run([sys.executable, "-c", "import torch, numpy"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout)
    except (subprocess.TimeoutExpired, OSError):
        pass


def run_legs(args, n, under_launcher):
    """`--layout auto` at N > 1: every layout of `--legs` measured as its OWN group of fresh rank processes, one after
    the other; rank 0 composes ONE line -- and prints a provisional one after every leg.  Without an external launcher
    this process starts each group itself (`launch_group`).  Under `torch.distributed.run` this process IS rank RANK of
    the driver's launch: it then starts one child per leg -- its rank of that leg's own process group (own rendezvous
    port) -- and never touches the GPU itself, so a leg whose communicator code fails or hangs (the RCCL entry points of
    the C ABI have never run on more than one rank before the driver's node) costs that leg only: the child is killed at
    `--leg-timeout`, the leg is recorded as failed and the line still carries every other leg.  No leg is started, or
    kept alive, past `--total-budget` seconds after the start of this process."""
    legs = [x for x in args.legs.split(",") if x]
    argv = _without_option(sys.argv[1:], "--layout")
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    deadline = T_START + args.total_budget
    warm_import(min(240.0, max(10.0, deadline - time.monotonic() - 60.0)))
    records = {}
    rc_all = 1
    for k, leg in enumerate(legs):
        leg_argv = argv + ["--layout", leg]
        t0 = time.monotonic()
        limit = min(args.leg_timeout * (2 if k == 0 else 1), deadline - t0 - 2.0)
        if limit < 15.0:
            records[leg] = {"status": "skipped (out of time: --total-budget)", "wall_s": 0.0}
            continue
        if under_launcher:
            env = {kk: v for kk, v in os.environ.items() if not kk.startswith("TORCHELASTIC")}
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1 + k)
            env["QS_BENCH_LEG"] = leg
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__)] + leg_argv, env=env,
                                     stdout=subprocess.PIPE, stderr=sys.stderr, text=True, timeout=limit)
                rc, lines = res.returncode, res.stdout.splitlines()
            except subprocess.TimeoutExpired as exc:          # (run() has killed and reaped the child)
                rc, lines = 124, (exc.stdout or "").splitlines() if isinstance(exc.stdout, str) else []
        else:
            rc, lines = launch_group(n, leg_argv, timeout=limit, extra_env={"QS_BENCH_LEG": leg})
        rec = {"status": "ok" if rc == 0 else f"failed (exit code {rc}{', timed out after %.0f s' % limit if rc == 124 else ''})",
               "wall_s": time.monotonic() - t0}
        for ln in lines:
            if ln.lstrip().startswith("{"):
                try:
                    rec["line"] = json.loads(ln)
                except ValueError:
                    pass
        if rc == 0 and "line" not in rec and rank == 0:
            rec["status"] = "failed (no result line)"
        records[leg] = rec
        if rank == 0:
            # a COMPLETE line after every leg: whatever happens to the legs still to come, the last line on stdout
            # carries everything measured so far (the final one, without the flag, comes last)
            done = [x for x in legs if x in records]
            rc_all = compose_legs(args, n, done, records, provisional=k + 1 < len(legs), pending=legs[k + 1:])
    if rank != 0:
        return 0
    if legs and legs[-1] in records and records[legs[-1]]["status"].startswith("skipped"):
        rc_all = compose_legs(args, n, [x for x in legs if x in records], records, provisional=False, pending=[])
    return rc_all


def compose_legs(args, n, legs, records, provisional=False, pending=()):
    """ONE line from the legs: `value` = the best leg whose time includes a collective."""
    summary, candidates = {}, []
    extra = {"provisional": True, "legs_pending": list(pending)} if provisional else {}
    for leg in legs:
        rec = records[leg]
        d = rec.get("line")
        ent = {"status": rec["status"], "wall_s": round(rec["wall_s"], 1)}
        if d:
            ent.update({"value": d["value"], "ms_per_step": d["ms_per_step"], "ms_per_step_median": d.get("ms_per_step_median"),
                        "layout": d["config"]["layout"], "parity": d["parity"], "collective": d.get("collective", {}).get("in_value"),
                        "kernel": d["roofline"]["kernel"], "n_ranks_seen": d.get("n_ranks_seen")})
            ok = rec["status"] == "ok" and d["parity"]["ok"]
            if "with_all_gather" in d:
                ent["with_all_gather"] = d["with_all_gather"]
                if ok:
                    candidates.append((d["with_all_gather"]["value"], leg + " + all-gather", d, d["with_all_gather"]))
            if ok and d.get("collective", {}).get("in_value") not in (None, "none"):
                candidates.append((d["value"], leg, d, None))
        summary[leg] = ent
    if not candidates:
        if not provisional:
            print("bench.py: no leg with a collective in its time finished", file=sys.stderr)
        fallback = [records[leg].get("line") for leg in legs if records[leg].get("line")]
        if not fallback:
            return 1
        line = dict(fallback[0], **extra)
        line["legs"] = summary
        print(json.dumps(line), flush=True)
        return 0 if line["parity"]["ok"] else 3
    value, name, d, gathered = max(candidates, key=lambda c: c[0])
    line = dict(d, **extra)
    line["legs"] = summary
    line["config"] = dict(d["config"], chosen_leg=name,
                          choice="best of the legs whose time includes a collective (see `legs`; the no-collective "
                                 "replicated figure is legs.replicated.value)")
    if gathered is not None:
        # the replicated layout followed by the north star's all-gather of the result: its own (shorter) timing leg
        line["value"] = gathered["value"]
        line["ms_per_step"] = gathered["ms_per_step"]
        line["steps"] = gathered["steps"]
        line["config"]["layout"] = d["config"]["layout"] + ", + all-gather of the result"
        line["collective"] = {"in_value": "all-gather of the result"}
        for k in ("ms_per_step_median", "ms_per_step_min", "value_at_median_step", "value_at_min_step", "with_all_gather"):
            line.pop(k, None)
        line["config"]["frac_of_mfma_peak"] = gathered["value"] / (MFMA_F64_PEAK_TFLOPS * n)
    print(json.dumps(line), flush=True)
    return 0


# ----------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------

def make_unitary(torch, l, dtype, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randn((l, l), dtype=torch.float64, device=device, generator=g)
    if dtype.is_complex:
        a = torch.complex(a, torch.randn((l, l), dtype=torch.float64, device=device, generator=g))
    C, _ = torch.linalg.qr(a)
    return C.contiguous()


def make_inputs(torch, l, dtype, device, seed=1234):
    """RandomBasisSet-shaped synthetic u (uniform [0,1), symmetrised
    u_pqrs = u_qpsr like random_basis.py:46-50) generated on the device slab
    by slab, and a unitary C from a seeded QR."""
    g = torch.Generator(device=device).manual_seed(seed)
    u = torch.empty((l, l, l, l), dtype=dtype, device=device)
    rdt = torch.float64
    step = max(1, min(l, (1 << 28) // (l * l * l)))
    for lo in range(0, l, step):
        hi = min(l, lo + step)
        blk = torch.rand((hi - lo, l, l, l), dtype=rdt, device=device, generator=g)
        if dtype.is_complex:
            blk = torch.complex(blk, torch.rand((hi - lo, l, l, l), dtype=rdt, device=device, generator=g))
        u[lo:hi] = blk
    # symmetrise in place pair by pair: u[p,q] <- (u[p,q] + u[q,p]^T)/2 over (r,s)
    for p in range(l):
        a = u[p, p:]                                  # (l-p, l, l)
        b = u[p:, p].transpose(1, 2)                  # u[q,p,s,r] as [q][r][s]
        sym = 0.5 * (a + b)
        u[p, p:] = sym
        u[p:, p] = sym.transpose(1, 2)
    a = torch.randn((l, l), dtype=rdt, device=device, generator=g)
    if dtype.is_complex:
        a = torch.complex(a, torch.randn((l, l), dtype=rdt, device=device, generator=g))
    C, _ = torch.linalg.qr(a)
    C = C.contiguous()
    Ct = C.conj().transpose(0, 1).resolve_conj().contiguous()
    return u, C, Ct


def make_u_slab(torch, l, dtype, device, axis, lo, hi, seed=1234, centre=False):
    """Rows [lo, hi) of axis 0 or 1 of a synthetic (l,l,l,l) tensor, generated ON THIS RANK ONLY (no rank
    ever allocates the whole tensor: l=512 complex128 is 1.1 TB).  Uniform [0,1) re and im like
    random_basis.py:52-69; the (p,q,r,s)<->(q,p,s,r) symmetrisation needs the partner slab of another rank and
    is skipped -- the transform and its parity property do not depend on it.  Element
    values depend on (seed, leading index, slab bounds) only."""
    n = hi - lo
    shape = (n, l, l, l) if axis == 0 else (l, n, l, l)
    u = torch.empty(shape, dtype=dtype, device=device)
    rdt = torch.float64
    per_row = shape[1] * l * l
    step = max(1, (1 << 27) // per_row)
    for a0 in range(0, shape[0], step):
        a1 = min(shape[0], a0 + step)
        first = (lo + a0) if axis == 0 else (a0 + 7919 * lo)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + first)
        sub = (a1 - a0,) + shape[1:]
        blk = torch.rand(sub, dtype=rdt, device=device, generator=g)
        if centre:
            blk -= 0.5
        if dtype.is_complex:
            blk = torch.complex(blk, torch.rand(sub, dtype=rdt, device=device, generator=g))
        u[a0:a1] = blk
    return u


def contract4(t, va, vb, vc, vd):
    """sum_abcd t[a,b,c,d] va[a] vb[b] vc[c] vd[d] as matrix-vector products over blocks of leading rows
    (no tensor-sized temporary; blocks of <= 2^18 rows: rocBLAS gemv rejects the 1.7e7-row call)."""
    import torch

    A, B, C, D = t.shape
    step = max(1, (1 << 18) // (B * C))
    parts = []
    for a0 in range(0, A, step):
        blk = t[a0:a0 + step]
        if blk.dtype != vd.dtype:
            blk = blk.to(vd.dtype)                      # (a real tensor against complex vectors: block-wise, never whole)
        x = blk.reshape(-1, D) @ vd
        x = x.reshape(-1, C) @ vc
        parts.append(x.reshape(-1, B) @ vb)
    return torch.cat(parts) @ va


def identity_parts(torch, u_part, out_rows, C, Ct, p_lo, b_lo=None, seed=7):
    """Size-independent parity property: contract both sides with random vectors x, y, z, w,
        sum out[pqrs] x_p y_q z_r w_s  ==  sum u[abcd] (Ct^T x)_a (Ct^T y)_b (C z)_c (C w)_d,
    O(l^4) each.  Returns this rank's partial sums (lhs over its rows p of `out`; rhs over the part of u it
    holds: everything with x restricted to its rows when u is whole, its b-slab when u is b-sharded); the sums
    over ranks are compared."""
    l, dt = C.shape[0], out_rows.dtype
    g = torch.Generator(device=out_rows.device).manual_seed(seed)
    x, y, z, w = [torch.randn(l, dtype=torch.float64, device=out_rows.device, generator=g).to(dt) for _ in range(4)]
    pc = out_rows.shape[0]
    C, Ct = C.to(dt), Ct.to(dt)
    lhs = contract4(out_rows, x[p_lo:p_lo + pc], y, z, w)
    yb, zc, wd = Ct.transpose(0, 1) @ y, C @ z, C @ w
    if b_lo is None:      # whole u on this rank: restrict the x-contraction to this rank's rows of Ct
        xa = Ct[p_lo:p_lo + pc].transpose(0, 1) @ x[p_lo:p_lo + pc]
        rhs = contract4(u_part, xa, yb, zc, wd)
    else:                 # u[:, b_lo:b_hi] on this rank
        xa = Ct.transpose(0, 1) @ x
        rhs = contract4(u_part, xa, yb[b_lo:b_lo + u_part.shape[1]], zc, wd)
    return lhs, rhs


def identity_parts_rows(torch, u_rows, out_rows, C, Ct, a_lo, q_lo, seed=7):
    """The same property for the rows layouts: this rank holds u[a_lo:a_hi] and out[:, q_lo:q_hi] stored with q leading
    (out_rows[q, p, r, s]); lhs over its q rows, rhs over its a rows, both summed over the ranks."""
    l, dt = C.shape[0], out_rows.dtype
    g = torch.Generator(device=out_rows.device).manual_seed(seed)
    x, y, z, w = [torch.randn(l, dtype=torch.float64, device=out_rows.device, generator=g).to(dt) for _ in range(4)]
    C, Ct = C.to(dt), Ct.to(dt)
    lhs = contract4(out_rows, y[q_lo:q_lo + out_rows.shape[0]], x, z, w)
    xa, yb, zc, wd = Ct.transpose(0, 1) @ x, Ct.transpose(0, 1) @ y, C @ z, C @ w
    rhs = contract4(u_rows, xa[a_lo:a_lo + u_rows.shape[0]], yb, zc, wd)
    return lhs, rhs


# ----------------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed on the host cores of this box)
# ----------------------------------------------------------------------------------------------

def host_mem_available():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline(l_max, budget_s=15.0, full_l=None):
    """The oracle (NumPy restatement of basis_set.py:341-348: tensordot x4) on the host cores of this box on
    a bounded sample of the same workload.  Fixed points at l=55 (BASELINE.json configs[1]) and l=128 with the
    5-operand numpy.einsum(optimize=True) form the reference's tests use (tests/test_custom_system.py:19-24)
    beside them, then a sample grown until it is worth ~`budget_s` seconds (time ~ l^5), capped at l_max;
    `value` is the largest tensordot sample.  `full_l`: one repetition at the full size (opt-in, ~1 min at 256)."""
    import numpy as np

    from oracle import qs_oracle as orc

    try:
        from threadpoolctl import threadpool_info

        infos = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        threads = max([i.get("num_threads", 1) for i in infos] or [1])
        blas = ",".join(sorted({str(i.get("internal_api")) for i in infos})) or "unknown"
    except Exception:
        threads, blas = os.cpu_count() or 1, "unknown"

    def inputs(l):
        rng = np.random.default_rng(0)
        if l >= 200:      # the full workload: 34 GB at l = 256 -- a repeated random block (the time does not depend on the values),
            u = np.resize(rng.random(1 << 22), (l, l, l, l))      # no symmetrised copy (three tensors of host memory less)
            C, _ = np.linalg.qr(rng.standard_normal((l, l)))
            return u, C
        u = rng.random((l, l, l, l))
        u = 0.5 * (u + u.transpose(1, 0, 3, 2))
        C, _ = np.linalg.qr(rng.standard_normal((l, l)))
        return u, C

    def run(l, einsum=False):
        u, C = inputs(l)
        t0 = time.perf_counter()
        out = orc.transform_two_body(u, C)
        dt = time.perf_counter() - t0
        assert out.shape == (l, l, l, l)
        dte = None
        if einsum:
            Ct = C.conj().T
            t0 = time.perf_counter()
            ref = np.einsum("pa,qb,abcd,cr,ds->pqrs", Ct, Ct, u, C, C, optimize=True)
            dte = time.perf_counter() - t0
            assert np.abs(ref - out).max() <= 1e-10 * np.abs(ref).max()
        return dt, dte

    run(32)                                   # BLAS thread pool warm-up
    points = []
    for lp in (55, 128):
        if lp <= max(l_max, 55):
            dt, dte = run(lp, einsum=True)
            fl = orc.transform_flops(lp, lp)
            points.append({"l": lp, "tensordot_x4_s": dt, "tensordot_x4_tflops": fl / dt / 1e12,
                           "einsum_optimize_s": dte, "einsum_optimize_tflops": fl / dte / 1e12})
    l, dt = (points[-1]["l"], points[-1]["tensordot_x4_s"]) if points else (48, run(48)[0])
    if full_l and full_l > l:                 # the workload itself, one repetition (instead of the grown sample)
        l, dt = full_l, run(full_l)[0]
    while dt < 0.6 * budget_s and l < l_max:  # grow the sample until it is worth 10-20 s
        nxt = int(min(l_max, max(l + 16, l * (budget_s / max(dt, 1e-3)) ** 0.2)))
        nxt -= nxt % 8
        if nxt <= l:
            break
        l, dt = nxt, run(nxt)[0]
    flops = orc.transform_flops(l, l)
    res = {
        "value": flops / dt / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
        "sample": f"one fp64 transform at l={l} ({flops/1e9:.1f} GFLOP, {dt:.1f} s), numpy "
                  f"{np.__version__} tensordot x4 (oracle/qs_oracle.py"
                  + (", the FULL workload, one repetition" if full_l and l == full_l else "") + f") on {blas} with {threads} threads "
                  f"(os.cpu_count={os.cpu_count()}); fixed points at l=55 and l=128 with "
                  f"numpy.einsum(optimize=True) beside them in `points`",
        "points": points,
    }
    if full_l:
        res["full_size"] = {"l": l, "tensordot_x4_s": dt, "tensordot_x4_tflops": flops / dt / 1e12}
    return res


# ----------------------------------------------------------------------------------------------
# timing helpers
# ----------------------------------------------------------------------------------------------

class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when the communicator is created; the
    bench contract is one JSON line on stdout, so fd 1 points at stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def timed_steps(torch, step, steps, warmup, barrier):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides.
    Returns (wall seconds for the K steps, device seconds between the first and last event, per-step device
    milliseconds from HIP events recorded on the launch stream, last result).  Steps of a millisecond or more get an
    event each; shorter ones (the l = 55 transform is two 50 us kernels) one event per ten steps, the per-step figures
    then being tenths of a group: an event between two such steps costs ~10 % of the step."""
    res = None
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for i in range(warmup):
        res = step(i)
    torch.cuda.synchronize()
    est = (time.perf_counter() - w0) / max(warmup, 1)
    stride = 1 if (warmup == 0 or est >= 1e-3 or steps < 20) else 10
    barrier()
    marks = list(range(0, steps, stride)) + [steps]
    evs = [torch.cuda.Event(enable_timing=True) for _ in marks]
    t0 = time.perf_counter()
    evs[0].record()
    nxt = 1
    for i in range(steps):
        res = step(warmup + i)
        if i + 1 == marks[nxt]:
            evs[nxt].record()
            nxt += 1
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = []
    for a in range(len(marks) - 1):
        n = marks[a + 1] - marks[a]
        per_step += [evs[a].elapsed_time(evs[a + 1]) / n] * n
    return elapsed, evs[0].elapsed_time(evs[-1]) * 1e-3, per_step, res


def median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def dominant_kernel(log):
    """('name', launches per step, full dispatch text) from the per-call dispatch records of one step."""
    counts, order = {}, []
    for rec in log:
        for item in rec.split(";"):
            if not item:
                continue
            name, _, rep = item.rpartition(" x")
            if not name or not rep.isdigit():
                name, rep = item, ""
            name = name.replace("+tail", "")     # (a launch whose last round is split: the same kernel symbol in a profile)
            k = int(rep) if rep else 1
            if name not in counts:
                order.append(name)
            counts[name] = counts.get(name, 0) + k
    if not counts:
        return "unknown", 1, ""
    heavy = [n for n in order if "transpose" not in n and not n.startswith("rccl")] or order
    total = sum(counts[n] for n in heavy)
    text = "; ".join(f"{n} x{counts[n]}" for n in order)
    return max(heavy, key=lambda n: counts[n]), total, text


# ----------------------------------------------------------------------------------------------
# HBM-bound workloads (BASELINE.json configs[3])
# ----------------------------------------------------------------------------------------------

def bandwidth_workload(args, torch, dist, kernels, sharded, device, rank, world, use_dist, ranks_seen):
    """BASELINE.json configs[3]: spatial real fp64 u at l (default 256), p-slab per rank (generated per rank),
    fused spin expansion + anti-symmetrisation + complex cast to 2l spin orbitals
    (264*l^4 algorithmic bytes), or the stand-alone anti-symmetrisation (16*l^4).  The spin
    tensor (1.1 TB at l=256) is streamed slab by slab through one reused output buffer."""
    l = args.l
    part = sharded.SlabPartition(l, world)
    p_lo, p_hi = part.bounds(rank)
    pc = p_hi - p_lo
    u = make_u_slab(torch, l, torch.float64, device, 0, p_lo, p_hi, seed=4321, centre=True)   # rows p_lo:p_hi only
    if args.workload == "spin_expand":
        rows = max(1, min(pc, int(40e9 // (2 * (2 * l) ** 3 * 16))))
        buf = torch.empty((2 * rows, 2 * l, 2 * l, 2 * l), dtype=torch.complex128, device=device)

        def step(_i):
            for r0 in range(0, pc, rows):
                r1 = min(pc, r0 + rows)
                kernels.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128,
                                             p_lo=r0, p_hi=r1, out=buf[: 2 * (r1 - r0)])
        step_bytes = (8 * l**3 + 16 * 2 * (2 * l) ** 3) * l
        launches = -(-pc // rows)
        name = "fused spin-expand + antisymmetrise + complex cast"
    else:
        out = torch.empty_like(u)

        def step(_i):
            kernels.antisymmetrize(u, out=out)
        step_bytes = 16 * l**4
        launches = 1
        name = "anti-symmetrise u"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, dev_elapsed, per_step, _ = timed_steps(torch, step, args.steps, args.warmup, barrier)
    kernels.dispatch_log = []
    step(0)
    kernel, _, dispatch = dominant_kernel(kernels.dispatch_log)
    kernels.dispatch_log = None
    if use_dist:
        t = torch.tensor([elapsed, dev_elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_elapsed = t[0].item(), t[1].item()
    # parity (value-exact): the first local p row against the definition written with torch ops
    row = u[:1]
    if args.workload == "spin_expand":
        got = kernels.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128, p_lo=0, p_hi=1)
        eye = torch.eye(2, dtype=torch.float64, device=device)
        ref = torch.einsum("pqrs,ac,bd->paqbrcsd", row, eye, eye).reshape(2, 2 * l, 2 * l, 2 * l)
        ref = (ref - ref.transpose(2, 3)).to(torch.complex128)
    else:
        got = kernels.antisymmetrize(row)
        ref = row - row.transpose(2, 3)
    ok = torch.tensor([1.0 if torch.equal(got, ref) else 0.0], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    exact = bool(ok.item() == 1.0)
    if rank == 0:
        gbps = step_bytes * args.steps / elapsed / 1e9
        per_launch = dev_elapsed / (args.steps * launches)
        achieved = step_bytes / world / launches / per_launch / 1e9
        traffic = traffic_source = None
        prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world == 1 and os.path.exists(prof):          # HBM bytes per (full-size) launch from this round's PMC passes
            try:
                with open(prof) as f:
                    for ent in json.load(f):
                        if ent.get("workload") == args.workload and ent.get("l") == l and ent.get("kernel") == kernel:
                            traffic, traffic_source = ent["hbm_bytes_per_launch"], ent.get("source")
            except Exception:
                pass
        line = {
            "metric": f"{name} GB/s (algorithmic bytes) at l={l} spatial orbitals",
            "value": gbps, "unit": "GB/s", "n_gpus": world, "n_ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": median(per_step),
            "ms_per_step_min": min(per_step), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic (p-slab generated per rank)",
            "config": {"workload": f"BASELINE.json configs[3]: spatial real fp64 u l={l} -> "
                                   f"{2*l} spin orbitals complex128, p-slab per GPU, output streamed "
                                   f"through a reused slab buffer" if args.workload == "spin_expand"
                                   else f"anti-symmetrisation of real fp64 u l={l}, p-slab per GPU",
                       "l": l, "bytes_per_step": step_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "dispatch": dispatch,
                         "bytes_per_launch": step_bytes / world / launches, "avg_launch_ms": per_launch * 1e3},
            "parity": {"value_exact_vs_definition": exact},
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if exact else 3


# ----------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------

def run_rank(args):
    if os.environ.get("QS_BENCH_HANG_LEG") and os.environ.get("QS_BENCH_HANG_LEG") == os.environ.get("QS_BENCH_LEG"):
        time.sleep(3600)      # test hook (tests/test_gpu_bench_script.py): a leg that hangs before it has touched the GPU
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    # rehearsal hooks for a one-GPU box (never set by the driver): all ranks on cuda:0 over
    # gloo, or a one-rank RCCL group, to walk the multi-rank code paths without 8 GPUs
    single_dev = os.environ.get("QS_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("QS_BENCH_BACKEND", "nccl")
    dev_index = 0 if single_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("QS_BENCH_FORCE_DIST") == "1"
    ranks_seen = 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            warm = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(warm)          # creates the communicator (and its banner) now; counts the ranks
            torch.cuda.synchronize()
            ranks_seen = int(warm.item())
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

    from quantum_systems_amd import _lib, kernels, sharded

    lib = _lib.load()
    if args.workload != "transform":
        return bandwidth_workload(args, torch, dist, kernels, sharded, device, rank, world, use_dist, ranks_seen)
    l = args.l
    mixed = args.dtype == "mixed"
    dtype = torch.float64 if args.dtype == "f64" else torch.complex128          # of C and of the result
    u_dtype = torch.float64 if mixed else dtype
    es = 8 if args.dtype == "f64" else 16
    kf = 1 if args.dtype == "f64" else 4
    # algorithmic flops: 8 l^5 real, 32 l^5 complex; real u x complex C: the d contraction is a real-by-complex product
    # (4 flops per multiply-add instead of 8): 4 l^5 + 3 * 8 l^5
    flops = 28 * l**5 if mixed else kf * 8 * l**5
    kernels.mixed_real_u = args.mixed_route == "native"
    fresh_c = args.fresh_c == "on" or (args.fresh_c == "auto" and args.dtype in ("c128", "mixed"))
    part = sharded.SlabPartition(l, world)
    p_lo, p_hi = part.bounds(rank)

    layout_kind = args.layout
    if world == 1 and not use_dist:
        layout_kind = "single" if args.layout in ("auto", "replicated") else args.layout
    elif args.layout == "auto":
        # replicated needs u + four slab-sized buffers per GPU
        layout_kind = "replicated" if l**4 * es * (1 + 4 / world) < 0.85 * HBM_BYTES else "inplace"
    if layout_kind == "inplace" and l % world:
        raise SystemExit("--layout inplace needs l divisible by the number of GPUs")
    coalesced = layout_kind == "rows_rccl_coalesced"
    if coalesced:
        layout_kind = "rows_rccl"
    if layout_kind in ("rccl", "rows_rccl") and world > 1 and (single_dev or backend != "nccl") \
            and not os.environ.get("QS_AMD_RCCL_LIB"):       # (the test suite's stand-in transport lets ranks share a GPU)
        raise SystemExit(f"--layout {layout_kind} drives RCCL directly: one GPU per rank (not available in the one-device rehearsal)")
    if mixed and layout_kind not in ("single", "rows", "rows_rccl"):
        raise SystemExit("--dtype mixed: single GPU or the rows layouts")

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs.  Coefficients: one unitary (fp64 default: a single change_basis) or a new one per step
    n_c = (args.steps + args.warmup) if fresh_c else 1
    Cs = [make_unitary(torch, l, dtype, device, 99 + i) for i in range(n_c)]
    b_lo = a_lo = None
    if layout_kind in ("single", "replicated"):
        u, C0, _ = make_inputs(torch, l, u_dtype, device)
        if not fresh_c:
            Cs = [C0]
        data = "synthetic"
    elif layout_kind in ("rows", "rows_rccl"):
        a_lo, a_hi = part.bounds(rank)
        u = make_u_slab(torch, l, u_dtype, device, 0, a_lo, a_hi)
        data = "synthetic (rows of the leading index generated per rank, not symmetrised)"
    else:
        b_lo, b_hi = part.bounds(rank)
        u = make_u_slab(torch, l, dtype, device, 1, b_lo, b_hi)
        data = "synthetic (b-slab generated per rank, not symmetrised)"
    if use_dist:
        for c in Cs:                          # QR on different devices may round differently: rank 0's C everywhere
            cr = torch.view_as_real(c) if c.is_complex() else c
            dist.broadcast(cr, src=0)
    Cts = [c.conj().transpose(0, 1).resolve_conj().contiguous() for c in Cs]
    torch.cuda.synchronize()

    def coeff(i):
        return (Cs[i % n_c], None if fresh_c else Cts[i % n_c])

    full_buf = [None]

    def gather(o):
        if full_buf[0] is None:
            full_buf[0] = torch.empty((l, l, l, l), dtype=dtype, device=device)
        return sharded.all_gather_slabs(o, l, rank, world, full=full_buf[0])

    rccl_comm = None
    if layout_kind in ("rccl", "rows_rccl"):
        ids = [kernels.RcclComm.unique_id() if rank == 0 else None]
        if use_dist:
            dist.broadcast_object_list(ids, src=0)         # 128 bytes, by the job's existing rendezvous
        with _StdoutToStderr():
            rccl_comm = kernels.RcclComm(rank, world, ids[0], rows_coalesce=coalesced)

    if layout_kind == "single":
        out = torch.empty(u.shape, dtype=dtype, device=device)

        def local_step(i):
            C, Ct = coeff(i)
            return kernels.transform_two_body(u, C, Ct, out=out)
        layout = "single GPU"
    elif layout_kind == "replicated":
        def local_step(i):
            C, Ct = coeff(i)
            return sharded.transform_two_body_replicated(u, C, Ct, rank, world)
        layout = "u replicated, out p-sharded, no collective on the data path"
    elif layout_kind == "sharded":
        def local_step(i):
            C, Ct = coeff(i)
            return sharded.transform_two_body_sharded(u, C, Ct, rank, world)
        layout = "u b-sharded, one all-to-all, out p-sharded (out of place)"
    elif layout_kind in ("rows", "rows_rccl"):
        keep = torch.empty(sharded.rows_buffer_elems(l, l, p_hi - p_lo), dtype=dtype, device=device)
        if layout_kind == "rows":
            def local_step(i):
                C, Ct = coeff(i)
                return sharded.transform_two_body_rows(u, C, Ct, rank, world, chunk_rows=args.chunk_rows or None, out=keep)
            layout = ("u sharded over its leading index and resident, out over its second (rows in, rows out): streamed, "
                      "input rows + result rows + O(l^3) per GPU, one all-to-all per chunk of rows (torch.distributed)")
        else:
            def local_step(i):
                C, Ct = coeff(i)
                return rccl_comm.transform_two_body_rows(u, C, Ct, chunk_rows=args.chunk_rows, out=keep)
            layout = ("u sharded over its leading index and resident, out over its second (rows in, rows out): ONE C-ABI "
                      "call, input rows + result rows + O(l^3) per GPU, RCCL grouped send/recv per chunk of rows on its "
                      "own stream under the next chunk's products"
                      + (", ONE message per peer and step (staging + strided copy)" if coalesced
                         else ", one message per (peer, result row) landing in place"))
    elif layout_kind == "rccl":
        out_slab = torch.empty((p_hi - p_lo, l, l, l), dtype=dtype, device=device)

        def local_step(i):
            C, Ct = coeff(i)
            return rccl_comm.transform_two_body(u, C, Ct, out=out_slab)
        layout = "u b-sharded, one C-ABI call: RCCL grouped send/recv, chunked exchange overlapped with the products"
    else:
        keep = torch.empty((l // world + 1, l, l, l), dtype=dtype, device=device)

        def local_step(i):
            C, Ct = coeff(i)
            return sharded.transform_two_body_sharded_inplace(u, C, Ct, rank, world, staging_rows=args.staging_rows,
                                                              out=keep)
        layout = "u b-sharded and resident, exchange + last contraction inside the output buffer (in place)"

    with_gather = args.gather and world > 1
    step = (lambda i: gather(local_step(i))) if with_gather else local_step
    if with_gather:
        layout += ", + all-gather of the result"

    elapsed, dev_elapsed, per_step, res = timed_steps(torch, step, args.steps, args.warmup, barrier)
    last_i = args.warmup + args.steps - 1

    # second leg at N > 1: the same steps followed by the north star's all-gather of the result
    gather_leg = None
    if world > 1 and not with_gather and not args.no_gather_leg and l**4 * es * (2 + 4 / world) < 0.85 * HBM_BYTES \
            and layout_kind in ("replicated", "sharded", "rccl"):
        k2 = max(1, min(args.steps, 3))
        e2, _, ps2, _ = timed_steps(torch, lambda i: gather(local_step(i)), k2, 1, barrier)
        gather_leg = (e2, k2, ps2)
        full_buf[0] = None

    # which kernels ran (one extra untimed step with the dispatch record on)
    kernels.dispatch_log = []
    res = local_step(last_i)
    kernel, launches_per_step, dispatch = dominant_kernel(kernels.dispatch_log)
    kernels.dispatch_log = None
    torch.cuda.synchronize()

    if use_dist:
        t = torch.tensor([elapsed, dev_elapsed] + ([gather_leg[0]] if gather_leg else []), dtype=torch.float64,
                         device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_elapsed = t[0].item(), t[1].item()
        if gather_leg:
            gather_leg = (t[2].item(),) + gather_leg[1:]
        ps = torch.tensor(per_step, dtype=torch.float64, device=device)
        dist.all_reduce(ps, op=dist.ReduceOp.MAX)
        per_step = ps.tolist()

    # parity property at full size: partial sums of both sides on this rank, summed over the ranks
    C_last = Cs[last_i % n_c]
    Ct_last = Cts[last_i % n_c]
    rows = res[p_lo:p_hi] if layout_kind == "single" else res
    if layout_kind in ("rows", "rows_rccl"):
        lhs, rhs = identity_parts_rows(torch, u, rows, C_last, Ct_last, a_lo, p_lo)
    else:
        lhs, rhs = identity_parts(torch, u, rows, C_last, Ct_last, p_lo, b_lo=b_lo)
    pair = torch.stack([lhs, rhs]).to(torch.complex128)
    if use_dist:
        pr = torch.view_as_real(pair).contiguous()
        dist.all_reduce(pr)
        pair = torch.view_as_complex(pr)
    rel = (abs(pair[0] - pair[1]) / abs(pair[1])).item()
    parity_ok = rel <= PARITY_BOUND

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return 0 if parity_ok else 3

    ms_per_step = elapsed / args.steps * 1e3
    value = flops * args.steps / elapsed / 1e12
    # dominant kernel = the MFMA GEMM family: `launches_per_step` launches carry all the
    # flops of a step; everything else on the stream is microseconds.
    per_launch_s = dev_elapsed / (args.steps * launches_per_step)
    achieved = (flops / world / launches_per_step) / per_launch_s / 1e12
    roofline = {
        "bound": "mfma", "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / MFMA_F64_PEAK_TFLOPS, "traffic": None,
        "kernel": kernel, "dispatch": dispatch, "launches_per_step": launches_per_step,
        "flops_per_launch": flops / world / launches_per_step,
        "avg_launch_ms": per_launch_s * 1e3,
    }
    prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if world == 1 and os.path.exists(prof):
        try:
            with open(prof) as f:
                tr = json.load(f)
            for ent in (tr if isinstance(tr, list) else [tr]):       # one entry per (size, dtype, kernel) profiled
                if ent.get("l") == l and ent.get("dtype") == args.dtype and ent.get("kernel", kernel) == kernel:
                    roofline["traffic"] = ent["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = ent.get("source")
        except Exception:
            pass

    probes = {}
    if not args.no_probes and world == 1:   # single-GPU ceilings; at N > 1 the other ranks would only wait
        st = torch.cuda.current_stream().cuda_stream
        blocks, iters = 256 * 8, 4000
        sink = torch.zeros(1 + 2 * blocks, dtype=torch.int64, device=device)
        lib.qs_probe_mfma_f64(sink.data_ptr(), blocks, iters, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.qs_probe_mfma_f64(sink.data_ptr(), blocks, iters, st)
        e1.record()
        torch.cuda.synchronize()
        probes["mfma_f64_register_loop_tflops"] = blocks * 4 * iters * 8 * 2048 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        stamps = sink[1:].reshape(blocks, 2).to(torch.float64)
        probes["mfma_f64_register_loop_clock_ghz"] = (stamps[:, 0] / stamps[:, 1]).median().item() * 0.1
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
        lib.qs_probe_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, st)
        e0.record()
        lib.qs_probe_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, st)
        e1.record()
        torch.cuda.synchronize()
        probes["hbm_stream_copy_tbps"] = 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12
        del src, dst

    med, mn = median(per_step), min(per_step)
    line = {
        "metric": f"4-index u transform TFLOP/s ({'fp64' if kf == 1 else ('real u x complex C' if mixed else 'complex128')}) "
                  f"at L={l} orbitals",
        "value": value, "unit": "TFLOP/s", "n_gpus": world, "n_ranks_seen": ranks_seen,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": med, "ms_per_step_min": mn,
        "value_at_median_step": flops / (med * 1e-3) / 1e12, "value_at_min_step": flops / (mn * 1e-3) / 1e12,
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64 u, c128 C and result" if mixed else args.dtype, "data": data,
        "config": {
            "workload": f"RandomBasisSet-shaped l={l} {'real-fp64-u x complex128-C' if mixed else args.dtype} four-index transform "
                        f"(BASELINE.json configs[{(1 if l == 55 else 2) if kf == 1 else 4}]"
                        + (" call pattern on a real u" if mixed else "") + "), u resident in HBM, C unitary"
                        + (", a new C every step (C_tilde derived inside the call)" if fresh_c else "")
                        + (f"; route: {'the real tensor read as it is (d contraction real x complex)' if args.mixed_route == 'native' else 'complex copy of u first (rounds 1-2)'}; flops counted 4 l^5 + 24 l^5" if mixed else ""),
            "l": l, "flops_per_step": flops, "layout": layout,
            "frac_of_mfma_peak": value / (MFMA_F64_PEAK_TFLOPS * world),
        },
        "roofline": roofline,
        "parity": {"randomised_identity_rel_diff": rel, "bound": PARITY_BOUND, "ok": parity_ok},
        "probes": probes,
    }
    if world > 1:
        line["collective"] = {"in_value": "all-gather of the result" if with_gather else (
            "none" if layout_kind == "replicated" else (
                "one all-to-all per chunk of input rows (re-shard of the intermediate)" if layout_kind in ("rows", "rows_rccl")
                else "one all-to-all (re-shard of the intermediate)"))}
        if gather_leg:
            e2, k2, _ = gather_leg
            line["with_all_gather"] = {"value": flops * k2 / e2 / 1e12, "ms_per_step": e2 / k2 * 1e3, "steps": k2,
                                       "note": "same steps followed by the all-gather that replicates the "
                                               "p-sharded result on every GPU (north star's single collective)"}
    if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0 would stall the others)
        del res, rows
        full = args.cpu_full == "on" or (args.cpu_full == "auto" and args.dtype == "f64" and 4.2 * 8 * l**4 < 0.9 * host_mem_available()
                                         and host_mem_available() >= 110e9 and l >= 200)
        line["cpu_baseline"] = cpu_baseline(args.cpu_l, full_l=l if full else None)
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if parity_ok else 3


def apply_config(args):
    """`--config N`: the workload BASELINE.json configs[N] names, without flag knowledge."""
    if args.config == 3:
        args.workload, args.l = "spin_expand", 256
    elif args.config == 4:
        n = int(os.environ.get("WORLD_SIZE", args.gpus))
        # input rows + result rows (+ one row, scratch) per rank within ~85 % of the HBM; 512 at 8 GPUs
        l = 512
        while l > 64 and 2 * l**4 * 16 / n + 8 * l**3 * 16 > 0.85 * HBM_BYTES:
            l -= 64
        args.l, args.dtype, args.fresh_c = l, "c128", "on"
        if args.layout == "auto":
            args.layout = "rows_rccl" if (n > 1 and os.environ.get("QS_BENCH_BACKEND", "nccl") == "nccl") else "rows"
    return args


def main():
    args = apply_config(parse())
    under_launcher = "WORLD_SIZE" in os.environ
    n = int(os.environ["WORLD_SIZE"]) if under_launcher else args.gpus
    if (n > 1 and args.layout == "auto" and args.workload == "transform" and "QS_BENCH_LEG" not in os.environ
            and os.environ.get("QS_BENCH_SELF_LAUNCHED") != "1"):
        return run_legs(args, n, under_launcher)
    if args.gpus > 1 and not under_launcher:
        return self_launch(args.gpus)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
