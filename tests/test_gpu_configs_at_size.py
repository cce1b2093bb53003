"""BASELINE.json configs[2] and configs[3] at THEIR size inside the default GPU suite (no opt-in):

* configs[2], l = 256 real fp64 (u = 34 GB): the HIP transform against the oracle on sampled output
  matrices out[p, q, :, :] -- element-wise, O(l^4) per sample on the host from ONE download of u --
  plus the randomised contraction identity over the whole result;
* configs[3], l = 256 -> 512 spin orbitals: fused spin expansion + anti-symmetrisation + complex cast on
  sampled p-slabs, `np.array_equal` against oracle.add_spin_two_body / anti_symmetrize_u
  (quantum_systems/basis_set.py:772-778), through the slab form a p-sharded rank uses.

The opt-in test_gpu_full_size.py compares all 4.3e9 elements (2 min of host time)."""

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu
L = 256


def _rand_u(l, seed, centre=False):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    u = torch.empty((l, l, l, l), dtype=torch.float64, device="cuda:0")
    for lo in range(0, l, 8):
        blk = torch.rand((min(8, l - lo), l, l, l), dtype=torch.float64, device="cuda:0", generator=g)
        u[lo:lo + 8] = blk - 0.5 if centre else blk
    return u


def test_config2_l256_sampled_rows_and_identity_vs_oracle():
    from quantum_systems_amd import kernels as K

    u = _rand_u(L, 2024)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    C, _ = torch.linalg.qr(torch.randn(L, L, dtype=torch.float64, device="cuda:0", generator=g))
    C = C.contiguous()
    out = K.transform_two_body(u, C)
    assert K.last_dispatch() == "qs::gemm_fast_kernel<false, 4, 4, true, false> x4"     # the headline kernel
    # randomised identity over the whole result: sum out x y z w == sum u (Ct^T x)(Ct^T y)(C z)(C w)
    x, y, z, w = [torch.randn(L, dtype=torch.float64, device="cuda:0", generator=g) for _ in range(4)]

    def contract(t, a, b, c, d):          # blocks of 4 leading rows: rocBLAS gemv rejects a 1.7e7-row call
        parts = []
        for a0 in range(0, L, 4):
            v = t[a0:a0 + 4].reshape(-1, L) @ d
            v = v.reshape(-1, L) @ c
            parts.append(v.reshape(-1, L) @ b)
        return torch.cat(parts) @ a

    Ct = C.t().contiguous()
    lhs = contract(out, x, y, z, w)
    rhs = contract(u, Ct.t() @ x, Ct.t() @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()
    # element-wise on sampled (p, q): corners, a tile edge of the 128 x 128 kernel, interior
    pairs = [(0, 0), (0, 255), (255, 0), (255, 255), (127, 128), (128, 127), (64, 191), (200, 13), (31, 32)]
    pairs += [(97, q) for q in range(0, L, 17)]          # a column of matrices of one leading index
    got = np.stack([out[p, q].cpu().numpy() for (p, q) in pairs])
    del out
    K.workspace.release()
    torch.cuda.empty_cache()
    u_host = u.cpu().numpy()                           # one download of the 34 GB tensor
    del u
    ref = orc.transform_two_body_pq_samples(u_host, C.cpu().numpy(), None, pairs)
    scale = np.abs(ref).max()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * scale


def test_config3_l256_spin_expand_antisym_cast_sampled_slabs_value_exact():
    from quantum_systems_amd import kernels as K

    u = _rand_u(L, 77, centre=True)                    # signed values: the reference's kron leaves -0.0 behind
    # (a) rows of the WHOLE tensor, absolute indexing (single-GPU form)
    for (p_lo, p_hi) in [(0, 1), (255, 256), (97, 99)]:
        got = K.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128, p_lo=p_lo, p_hi=p_hi)
        assert K.last_dispatch() == "qs::spin_expand_kernel<double, f64x2>"
        rows = u[p_lo:p_hi].cpu().numpy()
        ref = orc.anti_symmetrize_u(orc.add_spin_two_body(rows)).astype(np.complex128)   # basis_set.py:772-778, :634
        assert ref.shape == (2 * (p_hi - p_lo), 2 * L, 2 * L, 2 * L)
        assert np.array_equal(got.cpu().numpy(), ref)
        del got, ref
    # (b) the slab form of a p-sharded rank (rank 5 of 8 holds rows 160:192 only), no anti-symmetrisation
    slab = u[160:192].clone()
    del u
    got = K.spin_expand_two_body(slab, antisymmetrize=False, p_lo=30, p_hi=32)
    ref = orc.add_spin_two_body(slab[30:32].cpu().numpy())
    assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), ref)
    # (c) stand-alone anti-symmetrisation of a slab, in place
    ref = orc.anti_symmetrize_u(slab[:3].cpu().numpy())
    part = slab[:3].clone()
    K.antisymmetrize(part, out=part)
    assert np.array_equal(part.cpu().numpy(), ref)
