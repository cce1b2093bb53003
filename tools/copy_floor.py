"""How fast can ONE pass over an l^4 fp64 tensor be?  Times a plain device copy (read + write of
8*l^4 bytes each) back to back -- the floor for one unfused contraction pass at small l."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantum_systems_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for l in [int(x) for x in sys.argv[1:]] or [32, 40, 55, 64, 96, 128]:
    a = torch.rand(l, l, l, l, dtype=torch.float64, device=dev)
    b = torch.empty_like(a); c = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    def chain_torch():
        b.copy_(a); c.copy_(b); b.copy_(c); c.copy_(b)
    def chain_probe():
        n = a.numel() * 8
        lib.qs_probe_stream_copy(a.data_ptr(), b.data_ptr(), n, st)
        lib.qs_probe_stream_copy(b.data_ptr(), c.data_ptr(), n, st)
        lib.qs_probe_stream_copy(c.data_ptr(), b.data_ptr(), n, st)
        lib.qs_probe_stream_copy(b.data_ptr(), c.data_ptr(), n, st)
    for name, fn in (("torch copy_", chain_torch), ("16-byte copy probe", chain_probe)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 200
        print(f"l={l:4d} {name:>20}: {t*1e6:8.1f} us per pass ({a.numel()*8/1e6:.0f} MB in + out), {2*a.numel()*8/t/1e12:.2f} TB/s", flush=True)
