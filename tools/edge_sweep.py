"""Whole-transform rate for arbitrary l: general kernel (gemm_fast=2 keeps only the exact
form of the fast kernel) against the edge form of the fast kernel, automatic and per shape.
    python tools/edge_sweep.py [l ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")


def bench(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return min(ts)


def run(l, dt):
    u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev).to(dt)
    C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev))
    Ct = C.conj().T.contiguous()
    out = torch.empty_like(u)
    kf = 4 if dt.is_complex else 1
    res = []
    ref = None
    variants = [("general", 2, 0), ("edge auto", 3, 0)] + [(f"edge s{s}", 3, s) for s in (1, 2, 3, 4)]
    if os.environ.get("QS_SWEEP") == "slab":
        variants = [("two products", 1, 0, 1, 0), ("fused 1 wave/slab", 1, 0, 1, 2), ("fused 2 waves/slab", 1, 0, 1, 1)]
    if os.environ.get("QS_SWEEP") == "stream":
        variants = [("tiled", 1, 0, 0), ("stream", 1, 0, 1), ("stream no-split", 1, 0, 2)]
    for label, fast, shape, *rest in variants:
        if dt.is_complex and shape == 4:
            continue
        K.tuning_set("gemm_stream", rest[0] if rest else 0)
        K.tuning_set("slab_pair", rest[1] if len(rest) > 1 else 1)
        K.tuning_set("gemm_fast", fast)
        K.tuning_set("gemm_fast_shape", shape)
        reps = 20 if l <= 64 else 5
        t = bench(lambda: [K.transform_two_body(u, C, Ct, out=out) for _ in range(reps)]) / reps
        if ref is None:
            ref = out.clone()
            err = 0.0
        else:
            err = ((out - ref).abs().max() / ref.abs().max()).item()
        res.append(f"{label} {kf*8*l**5/t/1e12:6.2f} ({err:.0e})")
    K.tuning_set("gemm_fast", 1)
    K.tuning_set("gemm_fast_shape", 0)
    K.tuning_set("gemm_stream", 1)
    print(f"l={l:4d} {str(dt)[6:]:>10}: " + " | ".join(res), flush=True)
    del u, out, ref
    K.workspace.release()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ls = [int(x) for x in sys.argv[1:]] or [20, 32, 40, 55, 64, 72, 96, 100, 128, 160, 192, 200]
    for l in ls:
        run(l, torch.float64)
    for l in ls:
        if l <= 160:
            run(l, torch.complex128)
