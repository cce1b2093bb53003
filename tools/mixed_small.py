import os, sys
sys.path.insert(0, "/root/repo")
import torch
from quantum_systems_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for l in (12, 20, 28, 36, 44, 55, 64, 72):
    u = torch.randn((l,) * 4, dtype=torch.float64, device=dev, generator=g)
    C = torch.complex(torch.randn((l, l), dtype=torch.float64, device=dev, generator=g), torch.randn((l, l), dtype=torch.float64, device=dev, generator=g))
    C = torch.linalg.qr(C)[0].contiguous(); Ct = C.conj().T.contiguous()
    uc = u.to(torch.complex128)
    out = torch.empty_like(uc)
    t_mixed = timed(lambda: K.transform_two_body(u, C, Ct, out=out), 50); d1 = K.last_dispatch()[:70]
    t_c = timed(lambda: K.transform_two_body(uc, C, Ct, out=out), 50); d2 = K.last_dispatch()[:40]
    t_cast = timed(lambda: u.to(torch.complex128), 50)
    print(f"l={l}: mixed {t_mixed:8.1f} us ({d1}) | complex {t_c:8.1f} us ({d2}) | cast {t_cast:6.1f} us", flush=True)
