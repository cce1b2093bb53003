"""A/B of the balanced small-basis kernel's tail (sandwich_tail = 0 / 1) in one process: whole transform, event-timed, a different
C per call and rotating result buffers as in bench.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
sizes = [int(v) for v in os.environ.get("QS_TAIL_L", "46,47,48,56,57").split(",")]
for l in sizes:
    nbuf = max(2, int(600e6 // (8 * l**4)))          # more than the 256 MB memory-side cache holds
    us = [torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g) for _ in range(nbuf)]
    outs = [torch.empty_like(us[0]) for _ in range(nbuf)]
    Cs = [torch.linalg.qr(torch.randn((l, l), dtype=torch.float64, device=dev, generator=g))[0].contiguous() for _ in range(4)]
    Cts = [c.T.contiguous() for c in Cs]
    res = {}
    for rep in range(2):
        for tail in (0, 1):
            K.tuning_reset()
            K.tuning_set("sandwich_tail", tail)
            for kv in os.environ.get("QS_TAIL_TUNE", "").split(","):       # e.g. QS_TAIL_TUNE=sandwich_t2=1
                if "=" in kv:
                    K.tuning_set(kv.split("=")[0], int(kv.split("=")[1]))
            for i in range(8):
                K.transform_two_body(us[i % nbuf], Cs[i % 4], Cts[i % 4], out=outs[i % nbuf])
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(200):
                    K.transform_two_body(us[i % nbuf], Cs[i % 4], Cts[i % 4], out=outs[i % nbuf])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
            res[tail] = min(res.get(tail, 1e9), best)
            disp = K.last_dispatch()
    print(f"l={l}: tail off {res[0]:7.1f} us, on {res[1]:7.1f} us  ({disp})", flush=True)
K.tuning_reset()
