"""Whole fp64 transform at l = 17 ... 64: the streamed fused kernel (qs_quad4s.hip, forced wherever it exists) against the automatic
choice without it (tuning quad4s = 0: qs_small4.hip up to 20 orbitals, the 16-wide kernels up to 32, qs_sandwich4*.hip above), same
process, same tensors; bit-equality checked.  GPU time per transform from a graph of back-to-back transforms."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def rnd(shape):
    return torch.randn(shape, dtype=torch.float64, device=dev, generator=g)


def timed(u, C, Ct, out, reps):
    for _ in range(2):
        K.transform_two_body(u, C, Ct, out=out)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        K.transform_two_body(u, C, Ct, out=out)
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                K.transform_two_body(u, C, Ct, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


sizes = [int(v) for v in os.environ["QS_SWEEP_L"].split(",")] if os.environ.get("QS_SWEEP_L") else \
    [17, 20, 21, 24, 25, 28, 29, 32, 33, 36, 37, 40, 41, 44, 45, 48, 49, 52, 53, 55, 56, 57, 60, 61, 64]
print("l  quad4s_us  other_us  speedup  TFLOP/s(quad4s)  TFLOP/s(other)  bit-equal  kernels(other)")
for l in sizes:
    u, C = rnd((l,) * 4), rnd((l, l))
    C = torch.linalg.qr(C)[0].contiguous()
    Ct = C.T.contiguous()
    out = torch.empty_like(u)
    reps = 40 if l <= 32 else 10 if l <= 64 else 4
    K.tuning_set("quad4s", 2)
    K.tuning_set("small4", 0)
    a = K.transform_two_body(u, C, Ct).clone()
    name_a = K.last_dispatch()
    t_new = timed(u, C, Ct, out, reps)
    K.tuning_reset()
    K.tuning_set("quad4s", 0)
    b = K.transform_two_body(u, C, Ct).clone()
    name_b = K.last_dispatch()
    t_old = timed(u, C, Ct, out, reps)
    K.tuning_reset()
    fl = 8 * l**5
    print(f"{l:3d} {t_new:9.2f} {t_old:9.2f} {t_old / t_new:8.2f} {fl / t_new / 1e6:10.2f} {fl / t_old / 1e6:10.2f}"
          f"  {torch.equal(a, b)}  {name_a[:24]} | {name_b[:50]}", flush=True)
