"""Per-rank time of the replicated-u layout at world = 1, 2, 4, 8 measured on ONE
GPU (rank 0's share only; the layout has no collective on the data path, so this
is the rank's whole step).  Predicts the strong-scaling curve of bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K, sharded
dev = torch.device("cuda:0")
l = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = torch.complex128 if len(sys.argv) > 2 and sys.argv[2] == "c128" else torch.float64
kf = 4 if dt.is_complex else 1
u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev).to(dt)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=dt, device=dev))
Ct = C.conj().T.contiguous()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e-3)
    return min(ts)
base = None
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        s = t(lambda: sharded.transform_two_body_replicated(u, C, Ct, rank, world))
        if base is None: base = s
        # split: the a-contraction alone
        lo, hi = sharded.SlabPartition(l, world).bounds(rank)
        rows = Ct[lo:hi].contiguous()
        sa = t(lambda: K.matmul(rows, u.reshape(l, l**3)))
        print(f"world={world} rank={rank}: {s*1e3:8.2f} ms  -> aggregate {kf*8*l**5/s/1e12:7.1f} TFLOP/s  speedup {base/s:5.2f}x"
              f"   (a-contraction {sa*1e3:.2f} ms = {(l**4*(16 if kf==4 else 8))/sa/1e12:.2f} TB/s of u)", flush=True)
