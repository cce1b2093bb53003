"""PCIe-inclusive rate of the np=numpy route (host arrays in, host arrays out):
upload -> HIP transform -> download.  Reported in DESIGN.md; never the bench value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_systems_amd as qsa
for l, cplx in ((128, False), (128, True), (160, False)):
    rng = np.random.default_rng(0)
    u = rng.random((l,)*4); C = np.linalg.qr(rng.standard_normal((l, l)))[0]
    if cplx: u = u + 1j*rng.random((l,)*4); C = C.astype(complex)
    qsa.BasisSet.transform_two_body_elements(u, C, np)      # warm-up (allocator, page faults)
    t0 = time.perf_counter(); out = qsa.BasisSet.transform_two_body_elements(u, C, np); t = time.perf_counter() - t0
    du, dC = qsa.hip.asarray(u), qsa.hip.asarray(C)
    torch.cuda.synchronize(); t0 = time.perf_counter(); qsa.BasisSet.transform_two_body_elements(du, dC, qsa.hip); torch.cuda.synchronize(); tr = time.perf_counter() - t0
    k = 4 if cplx else 1
    print(f"l={l} {'c128' if cplx else 'f64'}: host->host {t*1e3:.0f} ms = {k*8*l**5/t/1e12:.2f} TFLOP/s ({u.nbytes*2/t/1e9:.1f} GB/s of PCIe payload); resident {tr*1e3:.1f} ms = {k*8*l**5/tr/1e12:.1f} TFLOP/s")
