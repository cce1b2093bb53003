"""The HIP transform against the REFERENCE at 72 ... 180 orbitals -- the sizes served by the streamed quads and the strip kernels
(one block column, 256-wide tiles, several tiles along the basis extent, the complex form, real tensor x complex coefficients).

tests/golden/mid_size_sampled.npz holds 1536 sampled elements and two whole-tensor sums of the reference's own
`BasisSet.transform_two_body_elements` (quantum_systems/basis_set.py:336-350, run by tests/golden/make_golden.py) on inputs
given by a closed integer formula (tests/_lattice_inputs.py), rebuilt here on the device.  Tolerance: 1e-10 relative to the
largest element (BASELINE.json north_star); measured ~1e-15.
"""

import numpy as np
import pytest
import torch

import _lattice_inputs as li

pytestmark = pytest.mark.gpu

RTOL = 1e-10

# the kernel family the automatic dispatch is expected to take (a fixture that no longer reaches the kernel it was made for
# should say so); None = not pinned
ROUTE = {
    "f64_78": "quad4s_kernel",
    "f64_100": "gemm_strip_kernel<false",
    "f64_130": "gemm_strip_kernel<false",
    "f64_150_to_120": None,
    "f64_180": "gemm_strip_kernel<false",
    "c128_72": "gemm_strip_kernel<true",
    "c128_100": "gemm_strip_kernel<true",
    "c128_140": "gemm_strip_kernel<true",
    "mixed_90": None,
}


@pytest.mark.parametrize("case", li.CASES, ids=[c[0] for c in li.CASES])
def test_transform_matches_reference_samples(golden, case):
    from quantum_systems_amd import kernels as K

    name, L, M, ucplx, ccplx, salt = case
    g = golden("mid_size_sampled")
    C, Ct = li.case_inputs_np(L, M, ucplx, ccplx, salt)
    u = li.tensor_torch(L, salt, ucplx, device="cuda:0")
    out = K.transform_two_body(u, torch.from_numpy(C).cuda(), torch.from_numpy(Ct).cuda())
    ran = K.last_dispatch()
    assert out.shape == (M,) * 4 and out.dtype == torch.from_numpy(g[name + "_val"]).dtype
    if ROUTE[name] is not None:
        assert ROUTE[name] in ran, (name, ran)
    pos = torch.from_numpy(g[name + "_pos"]).cuda()
    got = out[pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3]].cpu().numpy()
    scale = float(g[name + "_max_abs"])
    assert np.abs(got - g[name + "_val"]).max() <= RTOL * scale, (name, ran)
    abs_sum = float(g[name + "_abs_sum"])
    assert abs(out.sum().item() - g[name + "_sum"].item()) <= RTOL * abs_sum
    assert abs(out.abs().sum().item() - abs_sum) <= RTOL * abs_sum
    assert abs(out.abs().max().item() - scale) <= RTOL * scale
    # the input is untouched (the functional transform never donates its argument)
    assert torch.equal(u[L // 2], li.tensor_torch(L, salt, ucplx, device="cuda:0")[L // 2])
