"""TFLOP/s of the whole transform over l (fp64 and complex128), to find cliffs in the dispatch.  QS_SWEEP_L=a,b,... overrides."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from quantum_systems_amd import kernels as K

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
sizes = [int(v) for v in os.environ["QS_SWEEP_L"].split(",")] if os.environ.get("QS_SWEEP_L") else \
    [57, 60, 63, 64, 65, 66, 70, 72, 79, 80, 88, 95, 96, 97, 100, 104, 111, 112, 120, 127, 128, 129, 130, 136, 144, 150, 160, 176, 191,
     192, 193, 200, 208, 224, 240, 255, 256]
for kv in os.environ.get("QS_SWEEP_TUNE", "").split(","):          # e.g. QS_SWEEP_TUNE=gemm_fit=2,gemm_fast=0
    if "=" in kv:
        K.tuning_set(kv.split("=")[0], int(kv.split("=")[1]))
print("l dtype us TFLOP/s kernels")
dtypes = [d == "c128" for d in os.environ.get("QS_SWEEP_DTYPES", "f64,c128").split(",")]
for cx in dtypes:
    for l in sizes:
        if cx and l > 224:
            continue
        a = torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g)
        u = torch.complex(a, torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g)) if cx else a
        del a
        c = torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)
        C = torch.complex(c, torch.randn((l, l), dtype=torch.float64, device=dev, generator=g)) if cx else c
        C = torch.linalg.qr(C)[0].contiguous()
        Ct = C.conj().T.contiguous()
        out = torch.empty_like(u)
        for _ in range(2):
            K.transform_two_body(u, C, Ct, out=out)
        torch.cuda.synchronize()
        reps = max(2, min(20, int(2e5 / (l / 64) ** 5 / 160)))
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                K.transform_two_body(u, C, Ct, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        fl = (32 if cx else 8) * l**5
        print(f"{l:4d} {'c128' if cx else 'f64 '} {best:10.1f} {fl / best / 1e6:7.2f}  {K.last_dispatch()[:110]}", flush=True)
        del u, out
        K.workspace.release()
        torch.cuda.empty_cache()
K.tuning_reset()
