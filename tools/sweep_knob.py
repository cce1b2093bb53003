"""Sweep one tuning knob over several values on the whole transform.
    python tools/sweep_knob.py <knob> <l> v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
knob, l = sys.argv[1], int(sys.argv[2]); vals = [int(x) for x in sys.argv[3:]]
dt = torch.float64
u = torch.rand(l, l, l, l, dtype=dt, device=dev)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev)); Ct = C.T.contiguous()
out = torch.empty_like(u)
res = {v: [] for v in vals}
for r in range(4):
    for v in vals:
        K.tuning_set(knob, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.transform_two_body(u, C, Ct, out=out); e1.record(); torch.cuda.synchronize()
        if r: res[v].append(e0.elapsed_time(e1))
for v in vals:
    ts = sorted(res[v]); print(f"l={l} {knob}={v}: median {ts[len(ts)//2]:.2f} ms {8*l**5/ts[len(ts)//2]/1e9:.2f} TFLOP/s", flush=True)
