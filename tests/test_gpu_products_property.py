"""Property test of the product dispatch (hypothesis, runs on the GPU): random extents, leading
dimensions, batch strides, dtypes and the accumulate flag through qs_matmul -- whichever kernel the
dispatcher picks (fast exact / edge, skinny, stream, general) -- against NumPy's matmul on the same
strided views."""

import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu


@st.composite
def product_case(draw):
    kind = draw(st.sampled_from(["small", "tile", "stream", "wide"]))
    if kind == "small":
        m, n, k = draw(st.integers(1, 40)), draw(st.integers(1, 40)), draw(st.integers(1, 40))
    elif kind == "tile":
        m = draw(st.sampled_from([64, 100, 128, 130, 192, 200, 256]))
        n = draw(st.sampled_from([64, 100, 128, 136, 192, 258]))
        k = draw(st.sampled_from([16, 17, 48, 55, 64, 96, 100]))
    elif kind == "stream":
        m, k = draw(st.integers(1, 64)), draw(st.integers(1, 64))
        n = draw(st.sampled_from([40000, 65536, 70001, 33000]))
    else:
        m, k = draw(st.sampled_from([16, 32])), draw(st.sampled_from([8, 20, 64]))
        n = draw(st.sampled_from([65536, 131072]))
    batch = 1 if kind in ("stream", "wide") and n * m > 3_000_000 else draw(st.integers(1, 3))
    pad_a, pad_b, pad_c = draw(st.integers(0, 3)), draw(st.integers(0, 3)), draw(st.integers(0, 3))
    shared_a = draw(st.booleans())
    cplx = draw(st.booleans())
    accumulate = draw(st.booleans())
    seed = draw(st.integers(0, 2**31 - 1))
    return m, n, k, batch, pad_a, pad_b, pad_c, shared_a, cplx, accumulate, seed


@given(product_case())
@settings(max_examples=int(os.environ.get("QS_HYP_EXAMPLES", "70")), deadline=None, suppress_health_check=list(HealthCheck))
def test_random_products_match_numpy(case):
    from quantum_systems_amd import kernels as K

    m, n, k, batch, pad_a, pad_b, pad_c, shared_a, cplx, accumulate, seed = case
    rng = np.random.default_rng(seed)
    lda, ldb, ldc = k + pad_a, n + pad_b, n + pad_c
    na = 1 if shared_a else batch

    def rnd(*shape):
        x = rng.standard_normal(shape)
        return x + 1j * rng.standard_normal(shape) if cplx else x

    A, B, C0 = rnd(na, m, lda), rnd(batch, k, ldb), rnd(batch, m, ldc)
    dt = torch.complex128 if cplx else torch.float64
    dA, dB, dC = (torch.from_numpy(x).cuda() for x in (A, B, C0))
    K.gemm_raw(dt, dA, dB, dC, m, n, k, lda, ldb, ldc, batch,
               0 if shared_a else m * lda, k * ldb, m * ldc, accumulate)
    got = dC.cpu().numpy()
    prod = np.matmul(A[:, :, :k] if not shared_a else A[0, :, :k], B[:, :, :n])
    ref = C0.copy()
    ref[:, :, :n] = (C0[:, :, :n] + prod) if accumulate else prod
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= 1e-11 * scale * max(1, k) ** 0.5
    # the padding columns of C (ldc > n) are never touched
    assert np.array_equal(got[:, :, n:], C0[:, :, n:])


@st.composite
def transform_case(draw):
    # (up to 96 orbitals: the streamed kernels end there and the strip kernels begin; the oracle takes ~1 s at 96)
    L = draw(st.one_of(st.integers(1, 70), st.integers(1, 96)))
    M = draw(st.one_of(st.just(L), st.integers(1, 96 if L > 70 else 70)))
    cplx_u, cplx_c = draw(st.booleans()), draw(st.booleans())
    explicit_bra = draw(st.booleans())
    rows = draw(st.integers(1, L))
    seed = draw(st.integers(0, 2**31 - 1))
    return L, M, cplx_u, cplx_c, explicit_bra, rows, seed


@given(transform_case())
@settings(max_examples=int(os.environ.get("QS_HYP_EXAMPLES", "40")), deadline=None,
          suppress_health_check=list(HealthCheck))
def test_random_transforms_match_oracle(case):
    # whole transforms and leading-index slabs, square and rectangular, real / complex / mixed, default
    # and explicit bra coefficients -- through whatever kernels the sizes select (fused pass, streaming,
    # tiled) -- against the NumPy restatement of basis_set.py:336-350
    from oracle import qs_oracle as orc
    from quantum_systems_amd import kernels as K

    L, M, cplx_u, cplx_c, explicit_bra, rows, seed = case
    rng = np.random.default_rng(seed)

    def rnd(cplx, *shape):
        x = rng.standard_normal(shape)
        return x + 1j * rng.standard_normal(shape) if cplx else x

    u, C = rnd(cplx_u, L, L, L, L), rnd(cplx_c, L, M) / np.sqrt(L)
    Ct = rnd(cplx_c, M, L) / np.sqrt(L) if explicit_bra else None
    dev = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    ref = orc.transform_two_body(u, C, Ct)
    got = K.transform_two_body(dev(u), dev(C), dev(Ct)).cpu().numpy()
    assert got.dtype == ref.dtype and got.shape == ref.shape
    scale = max(1e-300, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= 1e-12 * scale
    lo = (L - rows) // 2
    Ct_eff = C.conj().T.copy() if Ct is None else Ct
    part = K.transform_two_body_partial(dev(u[lo:lo + rows]), dev(C), dev(Ct_eff)).cpu().numpy()
    ref_part = orc.transform_two_body_dcb(u[lo:lo + rows], C, Ct_eff)
    assert np.abs(part - ref_part).max() <= 1e-12 * max(1e-300, float(np.abs(ref_part).max()))


@given(st.integers(1, 40), st.booleans(), st.booleans(), st.booleans(), st.integers(0, 2**31 - 1), st.data())
@settings(max_examples=int(os.environ.get("QS_HYP_EXAMPLES", "40")), deadline=None,
          suppress_health_check=list(HealthCheck))
def test_random_spin_expansions_are_value_exact(l, cplx_in, cplx_out, anti, seed, data):
    # the bandwidth kernels against their definitions (basis_set.py:772-778, :634), value-exact:
    # any l, any p-slab, real/complex in and out, with and without the fused anti-symmetrisation,
    # and the stand-alone anti-symmetrisation in and out of place
    from quantum_systems_amd import kernels as K

    if cplx_in and not cplx_out:
        cplx_out = True
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((l,) * 4)
    if cplx_in:
        u = u + 1j * rng.standard_normal((l,) * 4)
    p_lo = data.draw(st.integers(0, l - 1))
    p_hi = data.draw(st.integers(p_lo + 1, l))
    eye = np.eye(2)
    ref = np.einsum("pqrs,ac,bd->paqbrcsd", u[p_lo:p_hi], eye, eye).reshape(2 * (p_hi - p_lo), 2 * l, 2 * l, 2 * l)
    if anti:
        ref = ref - ref.transpose(0, 1, 3, 2)
    if cplx_out:
        ref = ref.astype(np.complex128)
    du = torch.from_numpy(u).cuda()
    got = K.spin_expand_two_body(du, antisymmetrize=anti, out_dtype=torch.complex128 if cplx_out else None,
                                 p_lo=p_lo, p_hi=p_hi).cpu().numpy()
    assert got.dtype == ref.dtype and np.array_equal(got, ref)
    a_ref = u - u.transpose(0, 1, 3, 2)
    assert np.array_equal(K.antisymmetrize(du).cpu().numpy(), a_ref)
    dv = du.clone()
    K.antisymmetrize(dv, out=dv)
    assert np.array_equal(dv.cpu().numpy(), a_ref)
