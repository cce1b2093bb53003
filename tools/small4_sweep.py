"""Whole transform at l = 2 ... 36, fp64 and complex128: the two launches of the LDS-staged small-basis kernel
(qs_small4.hip) against the path of rounds 1-2 (tuning small4 = 0), same process, same tensors; bit-equality checked.
GPU time per transform from HIP events around a graph of 40 back-to-back transforms (no host time in the figure)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def rnd(shape, cx):
    a = torch.randn(shape, dtype=torch.float64, device=dev, generator=g)
    return torch.complex(a, torch.randn(shape, dtype=torch.float64, device=dev, generator=g)) if cx else a


def timed(u, C, Ct, out, reps=40):
    """GPU time per transform: `reps` transforms captured as ONE HIP graph (no host time between the launches, the
    dependencies between them as in the eager call), the best of five replays."""
    for _ in range(3):
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
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


print("l  dtype  small4_us  rounds12_us  speedup  TFLOP/s(small4)  bit-equal  kernels")
for cx in (False, True):
    sizes = [int(v) for v in os.environ["QS_SWEEP_L"].split(",")] if os.environ.get("QS_SWEEP_L") else list(range(2, 33)) + [33, 36]
    for l in sizes:
        u, C = rnd((l,) * 4, cx), rnd((l, l), cx)
        C = torch.linalg.qr(C)[0].contiguous()
        Ct = C.conj().T.contiguous()
        out = torch.empty_like(u)
        K.tuning_set("small4", 2)
        a = K.transform_two_body(u, C, Ct).clone()
        name = K.last_dispatch()
        t_new = timed(u, C, Ct, out)
        K.tuning_set("small4", 0)
        b = K.transform_two_body(u, C, Ct).clone()
        t_old = timed(u, C, Ct, out)
        K.tuning_reset()
        fl = (32 if cx else 8) * l**5
        print(f"{l:3d} {'c128' if cx else 'f64 '} {t_new:9.2f} {t_old:11.2f} {t_old / t_new:8.2f} {fl / t_new / 1e6:10.2f}"
              f"  {torch.equal(a, b)}  {name[:60]}", flush=True)
