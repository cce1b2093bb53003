"""Repeat the l=55 (config 2) transform for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K
l = int(sys.argv[1]) if len(sys.argv) > 1 else 55
dev = torch.device("cuda:0")
u = torch.rand(l, l, l, l, dtype=torch.float64, device=dev)
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device=dev)); Ct = C.T.contiguous()
out = torch.empty_like(u)
if os.environ.get("QS_SANDWICH"):       # 0 old path, 2 = (d, c) pass only, 3 = (b, a) pass only, 1 both (default)
    K.tuning_set("sandwich", int(os.environ["QS_SANDWICH"]))
for _ in range(30):
    K.transform_two_body(u, C, Ct, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    K.transform_two_body(u, C, Ct, out=out)
e1.record(); torch.cuda.synchronize()
print(f"l={l}: {e0.elapsed_time(e1)/100*1e3:.1f} us per transform, {8*l**5/(e0.elapsed_time(e1)/100*1e-3)/1e12:.2f} TFLOP/s")
