"""ONE rank's share of BASELINE.json configs[4] (l = 512 complex128, u sharded over 8 GPUs by leading-index rows) AT ITS SIZE on a
one-GPU box: 64 rows of 2 GiB in, 64 rows out, the C-ABI call `qs_transform_two_body_sharded_rows` as rank r of 8.

The seven peers do not exist: librccl is replaced by tests/cabi/absent_peers_rccl.cpp (sends dropped, receives deliver zeros --
what the exchange would deliver if every other rank held rows of zeros).  What this measures: that the rank's buffers fit the
GPU (266-270 GiB of 288), the rank's products at their real extents, its share of the exchange as bytes.
What it cannot measure: the links -- `--link-gbs x` models their TIME (every exchange step holds the communicator's stream for
the bytes of one peer / x GB/s; seven separate full-duplex links), which shows how much of it the stream pipeline hides.  The result is checked: it must be the transform of the tensor whose only non-zero
leading rows are this rank's -- sampled (q', r') planes against a dense contraction in torch.

    python tools/config4_one_rank.py [--orbitals 512] [--world 8] [--rank 0] [--steps 2] [--dtype c128|f64]
"""
import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_transport(tmp):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = os.path.join(tmp, "librccl_absent_peers.so")
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
                    os.path.join(ROOT, "tests", "cabi", "absent_peers_rccl.cpp"), "-o", so], check=True, capture_output=True, timeout=300)
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orbitals", type=int, default=512)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--dtype", default="c128")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--transport", default="", help="a prebuilt absent-peers stand-in (default: built into a temporary directory)")
    ap.add_argument("--build-transport", default="", help="only build the stand-in to this path and exit")
    ap.add_argument("--coalesce", action="store_true", help="the exchange with one message per peer and step (qs_comm_set_option rows_coalesce)")
    ap.add_argument("--link-gbs", type=float, default=0.0,
                    help="model the link TIME: every exchange step holds the communicator's stream for (bytes of one peer) / this many GB/s")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="absent_peers_")
    if a.build_transport:
        shutil.copy(build_transport(tmp), a.build_transport)
        return 0
    so = a.transport or build_transport(tmp)
    os.environ["QS_AMD_RCCL_LIB"] = so
    if a.link_gbs > 0:
        os.environ["ABSENT_PEERS_LINK_GBS"] = str(a.link_gbs)

    import torch

    from quantum_systems_amd import kernels as K

    l, G, r = a.orbitals, a.world, a.rank
    dt = torch.complex128 if a.dtype == "c128" else torch.float64
    es = 16 if dt.is_complex else 8
    dev = torch.device("cuda:0")
    free0, total = torch.cuda.mem_get_info()
    base, extra = divmod(l, G)
    il = base + (1 if r < extra else 0)
    lo = r * base + min(r, extra)
    print(f"# rank {r} of {G}, l = {l} {a.dtype}: {il} rows of {l ** 3 * es / 2 ** 30:.2f} GiB; device memory {total / 2 ** 30:.1f} GiB, free {free0 / 2 ** 30:.1f} GiB", flush=True)
    g = torch.Generator(device=dev).manual_seed(512 + r)
    rows = torch.empty((il, l, l, l), dtype=dt, device=dev)
    for i in range(il):                      # (a row at a time: the generator's temporaries stay at one row)
        if dt.is_complex:
            rows[i] = torch.complex(torch.randn((l, l, l), dtype=torch.float64, device=dev, generator=g),
                                    torch.randn((l, l, l), dtype=torch.float64, device=dev, generator=g))
        else:
            rows[i] = torch.randn((l, l, l), dtype=torch.float64, device=dev, generator=g)
    c = torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)
    C = torch.complex(c, torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)) if dt.is_complex else c
    C = torch.linalg.qr(C)[0].contiguous()
    Ct = C.conj().T.contiguous()
    torch.cuda.empty_cache()
    comm = K.RcclComm(r, G, K.RcclComm.unique_id(), rows_coalesce=a.coalesce)
    lib = ctypes.CDLL(so)
    lib.absent_peers_sent_bytes.restype = ctypes.c_uint64
    lib.absent_peers_received_bytes.restype = ctypes.c_uint64
    from quantum_systems_amd import _lib

    out_bytes = K.check(_lib.load().qs_transform_two_body_sharded_rows_out_bytes(K.dtype_code(dt), l, l, G, r), "size query")
    out_flat = torch.empty(out_bytes // es, dtype=dt, device=dev)      # (reused by every step, as in a time loop)
    times = []
    for step in range(a.steps + 1):          # (the first call is the warm-up: workspace allocation, first launches)
        torch.cuda.synchronize()
        s0, r0 = lib.absent_peers_sent_bytes(), lib.absent_peers_received_bytes()
        t0 = time.perf_counter()
        out = comm.transform_two_body_rows(rows, C, Ct, chunk_rows=a.chunk_rows, out=out_flat)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        if step:
            times.append(dt_s)
        sent, received = lib.absent_peers_sent_bytes() - s0, lib.absent_peers_received_bytes() - r0
        print(f"step {step}: {dt_s * 1e3:9.1f} ms; posted {sent / 1e9:.2f} GB of sends, {received / 1e9:.2f} GB of receives; route {K.last_dispatch()[:160]}", flush=True)
    peak = torch.cuda.max_memory_allocated()
    free1, _ = torch.cuda.mem_get_info()
    jl = out.shape[0]
    jlo = r * (l // G) + min(r, l % G)
    # the check.  The call returns out[j'_loc][i', r, s] = T[i', j_lo + j'_loc, r, s] -- the sharded index flips from the first to the
    # second leading index (sharded.transform_two_body_rows) -- and with peers that hold zeros T is the transform of the tensor whose
    # only non-zero leading rows are this rank's: out[j'_loc][i', r', s'] = sum_{i own} Ct[i', i] X[i, s'],
    # X[i, s'] = sum_{j, c, d} Ct[j', j] u[i, j, c, d] C[c, r'] C[d, s'].  Planes (j', r') sampled; per plane one pass over the rows.
    rng = torch.Generator().manual_seed(7)
    worst = 0.0
    for _ in range(a.samples):
        jloc, rp = int(torch.randint(0, jl, (1,), generator=rng)), int(torch.randint(0, l, (1,), generator=rng))
        plane = torch.empty((il, l), dtype=dt, device=dev)
        for i in range(il):
            t = torch.tensordot(Ct[jlo + jloc], rows[i], dims=([0], [0]))   # sum_j Ct[j', j] u[i, j, c, d] -> (c, d)
            plane[i] = (C[:, rp] @ t) @ C                                   # sum_c C[c, r'] ... then sum_d C[d, s']
        ref = Ct[:, lo:lo + il] @ plane                                     # (i', s')
        got = out[jloc, :, rp, :]
        worst = max(worst, ((got - ref).abs().max() / ref.abs().max()).item())
    ms = min(times) * 1e3
    flops = (32 if dt.is_complex else 8) * l ** 5 / G
    line = {"what": f"one rank's share of the l = {l} {a.dtype} transform sharded over {G} GPUs by leading-index rows, peers absent", "rank": r,
            "world": G, "link_gbs_per_peer_and_direction": a.link_gbs or None, "coalesced": a.coalesce, "rows_in": il, "rows_out": jl, "ms_per_step": round(ms, 1), "rank_tflops": round(flops / ms / 1e9, 2),
            "job_tflops_if_links_hidden": round(flops * G / ms / 1e9, 1), "peak_allocated_gib": round(peak / 2 ** 30, 1),
            "device_gib": round(total / 2 ** 30, 1), "free_after_gib": round(free1 / 2 ** 30, 1),
            "exchange_gb_per_step": {"sent": round(sent / 1e9, 2), "received": round(received / 1e9, 2)},
            "max_rel_err_sampled_planes": worst, "parity_ok": bool(worst <= 1e-10)}
    print(json.dumps(line))
    comm.close()
    shutil.rmtree(tmp, ignore_errors=True)
    return 0 if line["parity_ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
