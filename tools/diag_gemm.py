"""Steady-state rate of the GEMM kernels: large-K products where prologue,
epilogue and workgroup launch are negligible, vs K=256 (the l=256 transform)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
for dt in (torch.float64, torch.complex128):
    for (m, n, k) in [(8192, 8192, 128), (8192, 8192, 256), (8192, 8192, 512), (8192, 8192, 2048), (8192, 8192, 8192)]:
        A = torch.rand(m, k, dtype=torch.float64, device=dev).to(dt)
        B = torch.rand(k, n, dtype=torch.float64, device=dev).to(dt)
        C = torch.empty(m, n, dtype=dt, device=dev)
        kf = 4 if dt.is_complex else 1
        for fast in (0, 1):
            K.tuning_set("gemm_fast", fast)
            ms = t(lambda: K.matmul(A, B, out=C))
            print(f"{str(dt)[6:]:10s} {m}x{n}x{k:5d} fast={fast}: {ms:8.2f} ms {kf*2*m*n*k/ms/1e9:6.2f} TFLOP/s", flush=True)
        del A, B, C
K.tuning_set("gemm_fast", 1)
