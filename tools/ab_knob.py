"""Interleaved A/B of one tuning knob on the whole transform (rule 24).
    python tools/ab_knob.py <knob> <l> [l ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
knob = sys.argv[1]
def run(l, dt, rounds=5):
    u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev).to(dt)
    C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev))
    Ct = C.conj().T.contiguous()
    out = torch.empty_like(u)
    kf = 4 if dt.is_complex else 1
    res = {0: [], 1: []}
    for r in range(rounds + 1):
        for v in (0, 1):
            K.tuning_set(knob, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); K.transform_two_body(u, C, Ct, out=out); e1.record()
            torch.cuda.synchronize()
            if r: res[v].append(e0.elapsed_time(e1))
    K.tuning_set(knob, 1)
    for v in (0, 1):
        ts = sorted(res[v]); med = ts[len(ts)//2]
        print(f"l={l} {str(dt)[6:]} {knob}={v}: median {med:.2f} ms {kf*8*l**5/med/1e9:.2f} TFLOP/s  min {ts[0]:.2f} ms", flush=True)
    del u, out; K.workspace.release(); torch.cuda.empty_cache()
for l in [int(x) for x in sys.argv[2:]] or [256, 128]:
    run(l, torch.float64)
    if l <= 160: run(l, torch.complex128)
