"""Host-side cost of a transform call at small l: eager calls against replays of the captured
HIP graph (kernels.TransformPlan)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
for l in [int(x) for x in sys.argv[1:]] or [8, 16, 24, 32, 40, 55, 64]:
    u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev)
    C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device=dev))
    Ct = C.T.contiguous()
    out = torch.empty_like(u)
    plan = K.TransformPlan(u, C, Ct)
    def timed(fn, reps=300):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    te = timed(lambda: K.transform_two_body(u, C, Ct, out=out))
    tg = timed(plan.replay)
    print(f"l={l:3d}: eager {te*1e6:7.1f} us  graph {tg*1e6:7.1f} us  ({te/tg:.2f}x)  {8*l**5/tg/1e12:.2f} TFLOP/s", flush=True)
