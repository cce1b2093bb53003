"""Per-step shader-clock stamps of one wave of the sandwich kernel (library built with -DQS_S4_TRACE=<block>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import _lib, kernels as K
lib = _lib.load()
l = int(sys.argv[1]) if len(sys.argv) > 1 else 55
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda")
C, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda")); Ct = C.T.contiguous()
out = torch.empty_like(u)
K.tuning_set("sandwich", mode)
v2 = os.environ.get("QS_V2") == "1"
if v2:
    K.tuning_set("sandwich_v2", 1)
reset, read = (lib.qs_s4b_trace_reset, lib.qs_s4b_trace_read) if v2 else (lib.qs_s4_trace_reset, lib.qs_s4_trace_read)
for _ in range(3):
    K.transform_two_body(u, C, Ct, out=out)
torch.cuda.synchronize()
buf = torch.zeros(4097, dtype=torch.int64, device="cuda")
reset()
K.transform_two_body(u, C, Ct, out=out)
torch.cuda.synchronize()
read(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
n = int(buf[0]); st = buf[1:1 + n].tolist()
real = int(buf[4096])
print(f"wave lifetime: {(st[-1] >> 8) - (st[0] >> 8)} shader cycles in {real * 10} ns -> {((st[-1] >> 8) - (st[0] >> 8)) / (real * 10):.3f} GHz")
t0 = st[0] >> 8
prev = t0
for v in st:
    t, tag = v >> 8, v & 255
    name = {253: "entry", 254: "tables ready", 255: "chunk end"}.get(tag, ("P1 ka=%d" % tag) if tag < 64 else ("P2 ka=%d" % (tag - 64)))
    print(f"{t - t0:9d} (+{t - prev:6d})  {name}")
    prev = t
