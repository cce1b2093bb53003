"""Rate of the two-body S^2 kernel (16 n^4 bytes written): rows of a (n, n, n, n) complex128 tensor."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from quantum_systems_amd import kernels as K
for n, rows in ((512, 16), (1024, 2), (110, 110)):
    g = torch.Generator(device="cuda").manual_seed(n)
    S = torch.complex(torch.randn(3, n, n, dtype=torch.float64, device="cuda", generator=g),
                      torch.randn(3, n, n, dtype=torch.float64, device="cuda", generator=g))
    for anti in (False, True):
        for _ in range(2):
            out = K.spin_squared_two_body(S, antisymmetrize=anti, p_lo=0, p_hi=rows)
        del out; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = K.spin_squared_two_body(S, antisymmetrize=anti, p_lo=0, p_hi=rows)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        print(f"n={n} rows={rows} antisymmetrize={anti}: {t*1e3:.2f} ms  {16 * rows * n**3 / t / 1e12:.2f} TB/s")
