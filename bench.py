"""Headline benchmark: four-index transform of the two-body integrals.

    python bench.py [--gpus N --steps K --warmup W] [--orbitals 256] [--dtype f64|c128]
                    [--layout replicated|sharded] [--gather] [--no-cpu-baseline]

Metric (BASELINE.json): "4-index u transform TFLOP/s (fp64) at L orbitals".
A step is ONE full transform out = Ct Ct u C C of a synthetic RandomBasisSet-
shaped tensor resident in HBM (BASELINE.json configs[2]: real fp64, l=256,
u = 34.4 GB, C = real orthogonal).  The work figure is the algorithmic
8 l^5 flops (32 l^5 for complex128), whatever the kernels actually do.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the SAME
l=256 problem, output sharded over its leading index p ("strong" scaling).
Default layout: u replicated on every GPU, no collective on the data path
(SURVEY 8e); --layout sharded keeps u sharded over its second index and does
one all-to-all; --gather adds the all-gather that replicates the result.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F64_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz (MI355X_MICROARCH.md clocks)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--orbitals", "-l", dest="l", type=int, default=256,
                    help="number of orbitals l (named so that torch.distributed.run does not read it as --l*)")
    ap.add_argument("--dtype", choices=["f64", "c128"], default="f64")
    ap.add_argument("--layout", choices=["replicated", "sharded"], default="replicated")
    ap.add_argument("--gather", action="store_true", help="include the all-gather of the result")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-l", type=int, default=192, help="largest size of the CPU-baseline sample")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--workload", choices=["transform", "spin_expand", "antisymmetrize"], default="transform",
                    help="transform = the headline metric; the other two are the HBM-bound kernels of "
                         "BASELINE.json configs[3] (own metric names, same JSON shape)")
    return ap.parse_args()


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


def identity_check(torch, u, out_rows, C, Ct, p_lo, seed=7):
    """Size-independent parity property: contract both sides with
    random vectors; O(l^4) each.  Returns the relative difference."""
    l, dt = C.shape[0], u.dtype
    g = torch.Generator(device=u.device).manual_seed(seed)
    vs = [torch.randn(l, dtype=torch.float64, device=u.device, generator=g).to(dt) for _ in range(4)]
    x, y, z, w = vs
    pc = out_rows.shape[0]
    lhs = torch.einsum("pqrs,p,q,r,s->", out_rows, x[p_lo:p_lo + pc], y, z, w)
    # restrict the x-contraction to this rank's rows of Ct
    xa = Ct[p_lo:p_lo + pc].transpose(0, 1) @ x[p_lo:p_lo + pc]
    rhs = torch.einsum("abcd,a,b,c,d->", u, xa, Ct.transpose(0, 1) @ y, C @ z, C @ w)
    return lhs, rhs


def cpu_baseline(l_max, budget_s=15.0):
    """The oracle (NumPy restatement of basis_set.py:341-348) timed on the
    host cores of this box on a bounded sample of the same workload: a pilot
    at l=48 sizes the sample so that it takes roughly `budget_s` seconds
    (time ~ l^5), capped at l_max."""
    import numpy as np

    from oracle import qs_oracle as orc

    try:
        from threadpoolctl import threadpool_info

        infos = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        threads = max([i.get("num_threads", 1) for i in infos] or [1])
        blas = ",".join(sorted({str(i.get("internal_api")) for i in infos})) or "unknown"
    except Exception:
        threads, blas = os.cpu_count() or 1, "unknown"

    def run(l):
        rng = np.random.default_rng(0)
        u = rng.random((l, l, l, l))
        u = 0.5 * (u + u.transpose(1, 0, 3, 2))
        C, _ = np.linalg.qr(rng.standard_normal((l, l)))
        t0 = time.perf_counter()
        out = orc.transform_two_body(u, C)
        dt = time.perf_counter() - t0
        assert out.shape == (l, l, l, l)
        return dt

    run(32)                                   # BLAS thread pool warm-up
    l, dt = 64, run(64)
    while dt < 0.6 * budget_s and l < l_max:  # grow the sample until it is worth 10-20 s
        nxt = int(min(l_max, max(l + 16, l * (budget_s / max(dt, 1e-3)) ** 0.2)))
        nxt -= nxt % 8
        if nxt <= l:
            break
        l, dt = nxt, run(nxt)
    flops = orc.transform_flops(l, l)
    return {
        "value": flops / dt / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
        "sample": f"one fp64 transform at l={l} ({flops/1e9:.1f} GFLOP, {dt:.1f} s), numpy "
                  f"{np.__version__} tensordot x4 (oracle/qs_oracle.py) on {blas} with {threads} threads "
                  f"(os.cpu_count={os.cpu_count()})",
    }


HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)


def bandwidth_workload(args, torch, dist, kernels, sharded, device, rank, world, use_dist):
    """BASELINE.json configs[3]: spatial real fp64 u at l (default 256), p-slab per rank,
    fused spin expansion + anti-symmetrisation + complex cast to 2l spin orbitals
    (264*l^4 algorithmic bytes), or the stand-alone anti-symmetrisation (16*l^4).  The spin
    tensor (1.1 TB at l=256) is streamed slab by slab through one reused output buffer."""
    l = args.l
    part = sharded.SlabPartition(l, world)
    p_lo, p_hi = part.bounds(rank)
    g = torch.Generator(device=device).manual_seed(4321)
    u = torch.empty((l, l, l, l), dtype=torch.float64, device=device)
    for lo in range(0, l, 8):
        u[lo:lo + 8] = torch.rand((min(8, l - lo), l, l, l), dtype=torch.float64, device=device, generator=g) - 0.5
    if args.workload == "spin_expand":
        rows = max(1, min(p_hi - p_lo, int(40e9 // (2 * (2 * l) ** 3 * 16))))
        buf = torch.empty((2 * rows, 2 * l, 2 * l, 2 * l), dtype=torch.complex128, device=device)

        def step():
            for p0 in range(p_lo, p_hi, rows):
                p1 = min(p_hi, p0 + rows)
                kernels.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128,
                                             p_lo=p0, p_hi=p1, out=buf[: 2 * (p1 - p0)])
        step_bytes = (8 * l**3 + 16 * 2 * (2 * l) ** 3) * l
        launches = -(-(p_hi - p_lo) // rows)
        name = "fused spin-expand + antisymmetrise + complex cast"
        kernel = "qs::spin_expand_kernel<double, f64x2>"
    else:
        src = u[p_lo:p_hi]
        out = torch.empty_like(src)

        def step():
            kernels.antisymmetrize(src, out=out)
        step_bytes = 16 * l**4
        launches = 1
        name = "anti-symmetrise u"
        kernel = "qs::antisym_kernel<double>"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_elapsed = e0.elapsed_time(e1) * 1e-3
    if use_dist:
        t = torch.tensor([elapsed, dev_elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_elapsed = t[0].item(), t[1].item()
    # parity (value-exact): the first local p row against the definition written with torch ops
    row = u[p_lo:p_lo + 1]
    if args.workload == "spin_expand":
        got = kernels.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128, p_lo=p_lo, p_hi=p_lo + 1)
        eye = torch.eye(2, dtype=torch.float64, device=device)
        ref = torch.einsum("pqrs,ac,bd->paqbrcsd", row, eye, eye).reshape(2, 2 * l, 2 * l, 2 * l)
        ref = (ref - ref.transpose(2, 3)).to(torch.complex128)
    else:
        got = kernels.antisymmetrize(row)
        ref = row - row.transpose(2, 3)
    exact = bool(torch.equal(got, ref))
    if rank == 0:
        gbps = step_bytes * args.steps / elapsed / 1e9
        per_launch = dev_elapsed / (args.steps * launches)
        achieved = step_bytes / world / launches / per_launch / 1e9
        line = {
            "metric": f"{name} GB/s (algorithmic bytes) at l={l} spatial orbitals",
            "value": gbps, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[3]: spatial real fp64 u l={l} -> "
                                   f"{2*l} spin orbitals complex128, p-slab per GPU, output streamed "
                                   f"through a reused slab buffer" if args.workload == "spin_expand"
                                   else f"anti-symmetrisation of real fp64 u l={l}, p-slab per GPU",
                       "l": l, "bytes_per_step": step_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kernel,
                         "bytes_per_launch": step_bytes / world / launches, "avg_launch_ms": per_launch * 1e3},
            "parity": {"value_exact_vs_definition": exact},
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


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


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
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
            dist.all_reduce(warm)          # creates the communicator (and its banner) now
            torch.cuda.synchronize()

    from quantum_systems_amd import _lib, kernels, sharded

    lib = _lib.load()
    if args.workload != "transform":
        return bandwidth_workload(args, torch, dist, kernels, sharded, device, rank, world, use_dist)
    l = args.l
    dtype = torch.float64 if args.dtype == "f64" else torch.complex128
    kf = 1 if args.dtype == "f64" else 4
    flops = kf * 8 * l**5

    u, C, Ct = make_inputs(torch, l, dtype, device)
    part = sharded.SlabPartition(l, world)
    p_lo, p_hi = part.bounds(rank)

    if world == 1 and not use_dist:
        out = torch.empty_like(u)

        def step():
            return kernels.transform_two_body(u, C, Ct, out=out)
        launches_per_step = 4
        layout = "single"
    elif args.layout == "replicated":
        def step():
            o = sharded.transform_two_body_replicated(u, C, Ct, rank, world)
            if args.gather:
                o = sharded.all_gather_slabs(o, l, rank, world)
            return o
        launches_per_step = 4
        layout = "u replicated, out p-sharded" + (", +all-gather" if args.gather else ", no collective")
    else:
        b_lo, b_hi = part.bounds(rank)
        ub = u[:, b_lo:b_hi].contiguous()
        del u
        torch.cuda.empty_cache()
        u = None

        def step():
            o = sharded.transform_two_body_sharded(ub, C, Ct, rank, world)
            if args.gather:
                o = sharded.all_gather_slabs(o, l, rank, world)
            return o
        launches_per_step = 3 + world
        layout = "u b-sharded, one all-to-all, out p-sharded" + (", +all-gather" if args.gather else "")

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        res = step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_elapsed = ev0.elapsed_time(ev1) * 1e-3       # same stream as the kernels
    if use_dist:
        t = torch.tensor([elapsed, dev_elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_elapsed = t[0].item(), t[1].item()

    # parity property at full size on this rank's rows
    if u is not None:
        rows = res[p_lo:p_hi] if ((world == 1 and not use_dist) or args.gather) else res
        lhs, rhs = identity_check(torch, u, rows, C, Ct, p_lo)
        pair = torch.stack([lhs, rhs]).to(torch.complex128)
        if use_dist:
            pr = torch.view_as_real(pair).contiguous()
            dist.all_reduce(pr)
            pair = torch.view_as_complex(pr)
        rel = (abs(pair[0] - pair[1]) / abs(pair[1])).item()
    else:
        rel = None

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = flops * args.steps / elapsed / 1e12
    # dominant kernel = the MFMA GEMM: `launches_per_step` launches carry all the
    # flops of a step; everything else on the stream is microseconds.
    per_launch_s = dev_elapsed / (args.steps * launches_per_step)
    achieved = (flops / world / launches_per_step) / per_launch_s / 1e12
    roofline = {
        "bound": "mfma", "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / MFMA_F64_PEAK_TFLOPS, "traffic": None,
        "kernel": "qs::gemm_fast_kernel<false, 4, 4, true, false> (v_mfma_f64_16x16x4_f64; exact 128x128 tiles at l = 256)",
        "flops_per_launch": flops / world / launches_per_step,
        "avg_launch_ms": per_launch_s * 1e3,
    }
    prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if world == 1 and os.path.exists(prof):
        try:
            with open(prof) as f:
                tr = json.load(f)
            if tr.get("l") == l and tr.get("dtype") == args.dtype:
                roofline["traffic"] = tr["hbm_bytes_per_launch"]
                roofline["traffic_source"] = tr.get("source")
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

    line = {
        "metric": f"4-index u transform TFLOP/s ({'fp64' if kf == 1 else 'complex128'}) at L={l} orbitals",
        "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": f"RandomBasisSet-shaped l={l} {args.dtype} four-index transform "
                        f"(BASELINE.json configs[2]), u resident in HBM, C unitary",
            "l": l, "flops_per_step": flops, "layout": layout,
            "frac_of_mfma_peak": value / (MFMA_F64_PEAK_TFLOPS * world),
        },
        "roofline": roofline,
        "parity": {"randomised_identity_rel_diff": rel, "bound": 1e-10},
        "probes": probes,
    }
    if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0 would stall the others)
        del res
        line["cpu_baseline"] = cpu_baseline(args.cpu_l)
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
