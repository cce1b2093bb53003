"""Element-wise parity at the headline size (BASELINE.json configs[2], l = 256 fp64) against the
NumPy restatement of the reference on the SAME tensors: u is generated on the device, copied to
the host once, transformed by the oracle (four tensordots, ~2 minutes on the box's host cores,
~140 GB of host memory) and compared with the HIP result slab by slab.

Opt-in (QS_FULL_SIZE=1, optionally QS_FULL_SIZE_L=<l>): far too heavy for the default suite, which
checks the same property through the randomised contraction identity.  Result of the round-1 run:
profiles/r01_full_size_parity.txt."""

import os
import time

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(os.environ.get("QS_FULL_SIZE") != "1", reason="opt-in: QS_FULL_SIZE=1"),
]


def test_full_size_elementwise_parity_vs_numpy():
    from quantum_systems_amd import kernels as K

    l = int(os.environ.get("QS_FULL_SIZE_L", "256"))
    cplx = os.environ.get("QS_FULL_SIZE_DTYPE", "f64") == "c128"
    dt = torch.complex128 if cplx else torch.float64
    g = torch.Generator(device="cuda:0").manual_seed(99)
    u = torch.empty((l, l, l, l), dtype=dt, device="cuda:0")
    for lo in range(0, l, 8):
        blk = torch.rand((min(8, l - lo), l, l, l), dtype=torch.float64, device="cuda:0", generator=g)
        if cplx:
            blk = torch.complex(blk, torch.rand(blk.shape, dtype=torch.float64, device="cuda:0", generator=g))
        u[lo:lo + 8] = blk
    a = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g)
    if cplx:
        a = torch.complex(a, torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g))
    C, _ = torch.linalg.qr(a)
    t0 = time.perf_counter()
    out = K.transform_two_body(u, C)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    u_host = u.cpu().numpy()
    del u
    K.workspace.release()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    ref = orc.transform_two_body(u_host, C.cpu().numpy())
    t_cpu = time.perf_counter() - t0
    del u_host
    scale = float(np.abs(ref[0]).max())
    worst = 0.0
    differing = 0
    for p in range(l):
        d = np.abs(out[p].cpu().numpy() - ref[p])
        assert np.isfinite(d).all()                  # (a NaN would slip through max())
        worst = max(worst, float(d.max()))
        differing += int(np.count_nonzero(d))
        scale = max(scale, float(np.abs(ref[p]).max()))
    rel = worst / scale
    print(f"\nelements that differ at all: {differing} of {l**4}")
    print(f"\nfull-size parity l={l} {'complex128' if cplx else 'fp64'}: max|diff| = {worst:.3e}, max|ref| = {scale:.3e}, relative {rel:.3e}; "
          f"GPU {t_gpu:.2f} s (first call), NumPy {t_cpu:.1f} s")
    assert rel <= 1e-10
