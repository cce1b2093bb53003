"""Bandwidth kernels: anti-symmetrisation and fused spin expansion, GB/s of
ALGORITHMIC bytes on synthetic tensors.
    python tools/bench_bw.py [l_antisym] [l_spin]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import kernels as K, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e-3)
    ts.sort(); return ts[len(ts)//2]
la = int(sys.argv[1]) if len(sys.argv) > 1 else 192
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 96
st = torch.cuda.current_stream().cuda_stream
n = 1 << 31
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
s = t(lambda: lib.qs_probe_stream_copy(a.data_ptr(), b.data_ptr(), n, st))
print(f"stream copy probe: {2*n/s/1e12:.2f} TB/s"); 
s = t(lambda: b.copy_(a)); print(f"torch copy_: {2*n/s/1e12:.2f} TB/s")
del a, b
for dt, e in ((torch.float64, 8), (torch.complex128, 16)):
    u = torch.rand(la, la, la, la, dtype=torch.float64, device=dev).to(dt)
    out = torch.empty_like(u)
    s = t(lambda: K.antisymmetrize(u, out=out)); print(f"antisymmetrize {str(dt)[6:]} l={la}: {s*1e3:.2f} ms  {2*e*la**4/s/1e12:.2f} TB/s")
    s = t(lambda: K.antisymmetrize(u, out=u)); print(f"antisymmetrize in place {str(dt)[6:]} l={la}: {s*1e3:.2f} ms  {2*e*la**4/s/1e12:.2f} TB/s")
    del u, out
for (idt, odt, ei, eo) in ((torch.float64, torch.complex128, 8, 16), (torch.float64, torch.float64, 8, 8), (torch.complex128, torch.complex128, 16, 16)):
    u = torch.rand(ls, ls, ls, ls, dtype=torch.float64, device=dev).to(idt)
    out = torch.empty((2*ls,)*4, dtype=odt, device=dev)
    for anti in (True, False):
        s = t(lambda: K.spin_expand_two_body(u, antisymmetrize=anti, out_dtype=odt, out=out))
        by = ei*ls**4 + eo*(2*ls)**4
        print(f"spin_expand {str(idt)[6:]}->{str(odt)[6:]} antisym={anti} l={ls}: {s*1e3:.2f} ms  {by/s/1e12:.2f} TB/s")
    del u, out
n2 = 2*ls
S = torch.randn(3, n2, n2, dtype=torch.complex128, device=dev)
s = t(lambda: K.spin_squared_two_body(S, antisymmetrize=True)); print(f"spin_squared_two_body n={n2}: {s*1e3:.2f} ms  {16*n2**4/s/1e12:.2f} TB/s written")
