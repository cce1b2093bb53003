"""Index arithmetic beyond 32 bits: the l = 512 shapes of BASELINE.json
configs[4] cannot be held on one GPU as a whole tensor (1.1 TB), but their
LEADING DIMENSIONS can be exercised on slabs: the last contraction streams an
operand whose rows are l^3 = 134 M elements (1 or 2 GiB) apart, and slabs of the
d, c, b contractions use l = 512 extents.  Checked against torch's own fp64 GEMM
on the GPU (size-independent property: agreement of two independent
implementations), and against the oracle's formula on a sub-block."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from quantum_systems_amd import kernels

    return kernels


def free_gb():
    free, _ = torch.cuda.mem_get_info()
    return free / 2**30


def test_fast_kernel_with_gigabyte_row_stride_f64(K):
    # out[p, (qrs)] = Ct[p, a] T3[a, (qrs)] with M = 512: n = 512^3 columns, rows 1 GiB apart
    torch.cuda.empty_cache()
    if free_gb() < 170:
        pytest.skip("needs ~155 GB of HBM (B 17 GB, out 137 GB)")
    n, k, m = 512**3, 16, 128
    g = torch.Generator(device="cuda:0").manual_seed(1)
    B = torch.rand(k, n, dtype=torch.float64, device="cuda:0", generator=g)       # 17 GB
    A = torch.randn(m, k, dtype=torch.float64, device="cuda:0", generator=g)
    C = K.matmul(A, B)                                                             # m*n*8 = 137 GB
    # spot-check column blocks spread over the whole 1 GiB-stride range
    for c0 in (0, 2**27 - 256, 2**26 + 12345 * 128, n - 128):
        ref = A @ B[:, c0:c0 + 128]
        err = (C[:, c0:c0 + 128] - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 1e-13, (c0, err)
    del C, B
    torch.cuda.empty_cache()


def test_skinny_kernel_with_two_gigabyte_row_stride_c128(K):
    # the sharded layout's leading contraction at l = 512 complex: m = l/G rows, n = l^3
    if free_gb() < 90:
        pytest.skip("needs ~80 GB of HBM")
    n, k, m = 512**3, 8, 16
    g = torch.Generator(device="cuda:0").manual_seed(2)
    B = torch.view_as_complex(torch.rand(k, n, 2, dtype=torch.float64, device="cuda:0", generator=g))   # 17 GB
    A = torch.view_as_complex(torch.randn(m, k, 2, dtype=torch.float64, device="cuda:0", generator=g))
    C = K.matmul(A, B)                                                             # 34 GB
    for c0 in (0, 2**27 - 256, 2**26 + 777 * 128, n - 128):
        ref = A @ B[:, c0:c0 + 128]
        err = (C[:, c0:c0 + 128] - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 1e-13, (c0, err)
    del C, B


def test_partial_transform_slab_at_l512_complex(K):
    # d, c, b contractions of two leading-index rows of an l = 512 complex tensor
    if free_gb() < 60:
        pytest.skip("needs ~50 GB of HBM")
    l, rows = 512, 2
    g = torch.Generator(device="cuda:0").manual_seed(3)
    u = torch.view_as_complex(torch.rand(rows, l, l, l, 2, dtype=torch.float64, device="cuda:0", generator=g))
    C = torch.view_as_complex(torch.randn(l, l, 2, dtype=torch.float64, device="cuda:0", generator=g)) / l**0.5
    Ct = C.conj().T.contiguous()
    v = K.transform_two_body_partial(u, C, Ct)
    # randomised identity per row: sum_qrs v[a,q,r,s] y_q z_r w_s = sum_bcd u[a,b,c,d] (Ct^T y)_b (C z)_c (C w)_d
    y, z, w = (torch.view_as_complex(torch.randn(l, 2, dtype=torch.float64, device="cuda:0", generator=g)) for _ in range(3))
    lhs = torch.einsum("aqrs,q,r,s->a", v, y, z, w)
    rhs = torch.einsum("abcd,b,c,d->a", u, Ct.T @ y, C @ z, C @ w)
    assert ((lhs - rhs).abs() / rhs.abs()).max().item() <= 1e-10
    # and one explicit sub-block against the definition
    ref = torch.einsum("qb,bcd,cr,ds->qrs", Ct[:4], u[1], C[:, :3], C[:, :5])
    assert (v[1, :4, :3, :5] - ref).abs().max().item() <= 1e-10 * ref.abs().max().item()
