"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
committed golden vectors.  Needs a real MI355X: ``-m gpu``.

Tolerances: floating-point transforms <= 1e-10 relative (BASELINE.json
north_star; in practice ~1e-15), index/scatter work value-exact
(``np.array_equal``: -0.0 == +0.0).
"""

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from quantum_systems_amd import kernels

    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    return t.cpu().numpy()


def relerr(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)


def crand(rng, *shape):
    return rng.random(shape) + 1j * rng.random(shape)


# ------------------------------------------------------------------ golden


@pytest.mark.parametrize(
    "name",
    [
        "transform_c128_square",
        "transform_c128_rect_ctilde",
        "transform_c128_shrink",
        "transform_f64_orthogonal",
        "transform_f64_rect",
        "transform_mixed_real_u_complex_C",
    ],
)
def test_transform_golden(K, golden, name):
    g = golden(name)
    Ct = dev(g["C_tilde"]) if "C_tilde" in g else None
    u = host(K.transform_two_body(dev(g["u"]), dev(g["C"]), Ct))
    h = host(K.transform_one_body(dev(g["h"]), dev(g["C"]), Ct))
    assert u.dtype == g["u_out"].dtype and u.shape == g["u_out"].shape
    assert h.dtype == g["h_out"].dtype
    assert relerr(u, g["u_out"]) <= RTOL
    assert relerr(h, g["h_out"]) <= RTOL


def test_mixed_real_u_complex_C_runs_without_a_complex_copy_of_u(K, golden):
    # a real tensor against complex coefficients (NumPy's promotion, basis_set.py:341-342) goes through
    # qs_transform_two_body_mixed -- the d contraction is a REAL product on the real tensor (C seen as an (L, 2M) real
    # matrix), the rest complex -- instead of a complex copy of the whole tensor followed by the complex transform.
    g = golden("transform_mixed_real_u_complex_C")
    assert g["u"].dtype == np.float64 and g["C"].dtype == np.complex128
    u, C = dev(g["u"]), dev(g["C"])
    got = K.transform_two_body(u, C)
    ran = K.last_dispatch().split(";")
    if "pair4s" in ran[0]:           # small bases: the streamed fused kernel with REAL items for the first pass
        assert ran[0].endswith(", true>") and "pair4s" in ran[1] and ran[1].endswith(", false>"), ran
        with K.tuning(pair4c=0):     # ... bit-identical to the tiled route (real product for d, then complex)
            tiled = K.transform_two_body(u, C)
            assert "pair4s" not in K.last_dispatch()
        assert torch.equal(tiled, got)
    else:
        first = [r for r in ran if "gemm" in r][0]
        assert "<false" in first or "gemm_kernel<" in first or "stream_left_kernel" in first, ran     # the real product first
    assert relerr(host(got), g["u_out"]) <= RTOL
    K.mixed_real_u = False
    try:
        cast = K.transform_two_body(u, C)
    finally:
        K.mixed_real_u = True
    assert (got - cast).abs().max().item() <= 1e-13 * cast.abs().max().item()
    # oracle sweep through the new route: odd sizes, rectangular both ways, explicit C~, whole tiles, the fused sizes
    rng = np.random.default_rng(33)
    for (L, M) in ((5, 5), (9, 14), (21, 13), (13, 15), (20, 20), (31, 29), (40, 40), (47, 45), (64, 64), (55, 55), (56, 53), (70, 66),
                   (128, 128)):
        un = rng.standard_normal((L,) * 4)
        Cn = (rng.standard_normal((L, M)) + 1j * rng.standard_normal((L, M))) / np.sqrt(L)
        Ctn = (rng.standard_normal((M, L)) + 1j * rng.standard_normal((M, L))) / np.sqrt(L)
        d_u = dev(un)
        got = K.transform_two_body(d_u, dev(Cn), dev(Ctn))
        assert got.dtype == torch.complex128
        fused = 5 <= min(L, M) and max(L, M) <= 56 and -(-L // 4) == -(-M // 4)
        assert ("pair4s" in K.last_dispatch()) == fused, (L, M, K.last_dispatch())
        if fused:                    # the streamed kernel against the tiled route: the same chains
            with K.tuning(pair4c=0):
                assert torch.equal(K.transform_two_body(d_u, dev(Cn), dev(Ctn)), got), (L, M)
        if L <= 70:
            assert relerr(host(got), orc.transform_two_body(un, Cn, Ctn)) <= RTOL, (L, M)
        else:
            K.mixed_real_u = False
            try:
                cast = K.transform_two_body(d_u, dev(Cn), dev(Ctn))
            finally:
                K.mixed_real_u = True
            assert (got - cast).abs().max().item() <= 1e-12 * cast.abs().max().item()
        assert np.array_equal(host(d_u), un)                          # the real tensor is read, not converted
    # argument checks of the entry point itself
    from quantum_systems_amd import _lib
    lib = _lib.load()
    s0 = torch.cuda.current_stream().cuda_stream
    w = torch.empty(lib.qs_transform_two_body_workspace(1, 4, 4), dtype=torch.uint8, device="cuda")
    u4, C4 = dev(rng.standard_normal((4,) * 4)), dev(crand(rng, 4, 4))
    o4 = torch.empty((4,) * 4, dtype=torch.complex128, device="cuda")
    assert lib.qs_transform_two_body_mixed(u4.data_ptr(), C4.data_ptr(), C4.data_ptr(), o4.data_ptr(), w.data_ptr(), 16, 4, 4, s0) == -4
    assert lib.qs_transform_two_body_mixed(None, C4.data_ptr(), C4.data_ptr(), o4.data_ptr(), w.data_ptr(), w.numel(), 4, 4, s0) == -2
    assert lib.qs_transform_two_body_mixed(u4.data_ptr(), C4.data_ptr(), C4.data_ptr(), o4.data_ptr(), w.data_ptr(), w.numel(), 4, 4, s0) == 0


@pytest.mark.parametrize("cplx", [False, True])
def test_small_basis_kernel_is_bit_identical_to_the_16_wide_path(K, cplx):
    # qs_small4.hip: both fused passes of a basis of <= 32 orbitals, fp64 AND complex128, two launches.  Every element is
    # the same chain of fused multiply-adds as on the 16-wide kernels (for complex128 including the order of the two
    # imaginary-part products of each contraction), so the results are bit-identical; against the oracle <= 1e-10.
    rng = np.random.default_rng(40 + cplx)
    shapes = [(l, l) for l in (1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 21, 24, 27, 28, 29, 31, 32)]
    shapes += [(5, 7), (8, 6), (18, 20), (20, 17), (30, 32), (32, 29)]           # rectangular within one quad count
    for (L, M) in shapes:
        u = rng.standard_normal((L,) * 4)
        C = rng.standard_normal((L, M)) / np.sqrt(L)
        Ct = rng.standard_normal((M, L)) / np.sqrt(L)
        if cplx:
            u = u + 1j * rng.standard_normal((L,) * 4)
            C = C + 1j * rng.standard_normal((L, M)) / np.sqrt(L)
            Ct = Ct + 1j * rng.standard_normal((M, L)) / np.sqrt(L)
        du, dC, dCt = dev(u), dev(C), dev(Ct)
        auto = max(L, M) <= (4 if cplx else 8)            # (above: the streamed kernels, qs_quad4s.hip / pair4s_kernel)
        if not auto:
            K.tuning_set("small4", 2)                      # ... and wherever it exists
        try:
            got = K.transform_two_body(du, dC, dCt)
            ran = K.last_dispatch()
        finally:
            K.tuning_reset()
        assert ran == f"qs::small4_kernel<{'true' if cplx else 'false'}, {-(-L // 4)}> x2", (L, M, ran)
        if not auto:
            K.transform_two_body(du, dC, dCt)
            assert "small4" not in K.last_dispatch(), (L, M)
        K.tuning_set("small4", 0)
        try:
            wide = K.transform_two_body(du, dC, dCt)
            assert "small4" not in K.last_dispatch()
        finally:
            K.tuning_reset()
        assert torch.equal(got, wide), (L, M)
        if L <= 20:
            assert relerr(host(got), orc.transform_two_body(u, C, Ct)) <= RTOL, (L, M)
    # a different quad count for L and M, more than 32 orbitals, a real tensor against complex coefficients: other kernels
    for (L, M) in ((8, 9), (33, 33), (16, 12)):
        K.transform_two_body(dev(rng.standard_normal((L,) * 4)), dev(rng.standard_normal((L, M))))
        assert "small4" not in K.last_dispatch()
    K.transform_two_body(dev(rng.standard_normal((8,) * 4)), dev(crand(rng, 8, 8)))
    assert "small4" not in K.last_dispatch()
    # odd item counts (the last item quad is incomplete) next to live memory: a slab of a larger tensor, rows after it poisoned
    L = 5
    big = torch.full((L + 1, L, L, L), float("nan"), dtype=torch.float64, device="cuda")
    uu = rng.standard_normal((L,) * 4)
    big[:L] = dev(uu)
    Cn = rng.standard_normal((L, L))
    got = K.transform_two_body(big[:L], dev(Cn))
    assert "small4" in K.last_dispatch() and torch.isfinite(got).all()
    assert relerr(host(got), orc.transform_two_body(uu, Cn)) <= RTOL


def test_streamed_fp64_kernel_is_bit_identical_to_the_16_wide_path(K):
    # qs_quad4s.hip: both fused passes of a REAL basis of 17 ... 64 orbitals, item quads streamed through a ring of row quads,
    # one wave per column group; the same k-ordered chains as the 16-wide kernels.  Automatic for 17 ... 32 orbitals.
    rng = np.random.default_rng(91)
    shapes = [(l, l) for l in (5, 8, 9, 11, 12, 13, 16, 17, 19, 20, 21, 24, 25, 27, 28, 29, 31, 32, 33, 36, 39, 41, 45, 48, 50, 53,
                               55, 56, 57, 60, 63, 64)]
    shapes += [(7, 5), (10, 12), (18, 20), (20, 17), (30, 32), (32, 29), (55, 53), (62, 64)]     # rectangular within one quad count
    shapes += [(65, 65), (66, 68), (71, 69), (77, 78), (84, 81), (91, 91), (96, 93)]             # two workgroups per item quad
    for (L, M) in shapes:
        u = dev(rng.standard_normal((L,) * 4))
        C = dev(rng.standard_normal((L, M)) / np.sqrt(L))
        Ct = dev(rng.standard_normal((M, L)) / np.sqrt(L))
        with K.tuning(quad4s=2, small4=0):
            got = K.transform_two_body(u, C, Ct)
            name = f"qs::quad4s_kernel<{-(-L // 4)}{', 2' if L > 64 else ''}>"
            assert K.last_dispatch() == f"{name} x2", (L, M, K.last_dispatch())
        with K.tuning(quad4s=0, small4=0, sandwich=0):
            wide = K.transform_two_body(u, C, Ct)
            assert "quad4s" not in K.last_dispatch() and "sandwich4" not in K.last_dispatch() and "small4" not in K.last_dispatch()
        assert torch.equal(got, wide), (L, M)
        K.transform_two_body(u, C, Ct)
        auto = (9 <= min(L, M) and max(L, M) <= 32) or (65 <= min(L, M) and (L, M) != (96, 96))
        assert ("quad4s" in K.last_dispatch()) == auto, (L, M, K.last_dispatch())
    # an item count that is not a multiple of four next to poisoned memory, and non-finite values staying in their slabs
    L = 21
    big = torch.full((L + 1, L, L, L), float("nan"), dtype=torch.float64, device="cuda")
    uu = rng.standard_normal((L,) * 4)
    big[:L] = dev(uu)
    Cn = rng.standard_normal((L, L))
    got = K.transform_two_body(big[:L], dev(Cn))
    assert "quad4s" in K.last_dispatch() and torch.isfinite(got).all()
    assert relerr(host(got), orc.transform_two_body(uu, Cn)) <= RTOL


def test_complex_fused_kernel_is_bit_identical_to_the_16_wide_path(K):
    # qs_pair4s.h: both fused passes of a COMPLEX128 basis of 5 ... 64 orbitals, two items per matrix instruction (blocks =
    # (item, re | im)), item pairs streamed through a ring of row quads; the same chains of fused multiply-adds as the four
    # 16-wide passes, element for element.  Automatic up to 56 orbitals.
    rng = np.random.default_rng(77)
    shapes = [(5, 5), (8, 6), (9, 11), (13, 16), (17, 17), (21, 24), (25, 25), (28, 27), (29, 32), (32, 32), (33, 33), (36, 34), (37, 40),
              (41, 44), (47, 45), (48, 48), (49, 52), (55, 55), (56, 53), (57, 60), (64, 61)]
    for (L, M) in shapes:
        u = dev(rng.standard_normal((L,) * 4) + 1j * rng.standard_normal((L,) * 4))
        C = dev((rng.standard_normal((L, M)) + 1j * rng.standard_normal((L, M))) / np.sqrt(L))
        Ct = dev((rng.standard_normal((M, L)) + 1j * rng.standard_normal((M, L))) / np.sqrt(L))
        with K.tuning(pair4c=2, small4=0):
            got = K.transform_two_body(u, C, Ct)
            assert K.last_dispatch() == f"qs::pair4s_kernel<{-(-L // 4)}, false> x2", (L, M, K.last_dispatch())
        K.transform_two_body(u, C, Ct)
        assert ("pair4s" in K.last_dispatch()) == (max(L, M) <= 56), (L, M, K.last_dispatch())
        with K.tuning(pair4c=0, small4=0):
            wide = K.transform_two_body(u, C, Ct)
            assert "pair4" not in K.last_dispatch() and "small4" not in K.last_dispatch()
        assert torch.equal(got, wide), (L, M)
    # up to 4 orbitals the kernel does not exist: qs_small4.hip
    u3 = dev(rng.standard_normal((3,) * 4) + 1j * rng.standard_normal((3,) * 4))
    with K.tuning(pair4c=2):
        K.transform_two_body(u3, dev(rng.standard_normal((3, 3)) + 0j))
        assert "small4" in K.last_dispatch()
    # an odd number of items (the last pair is half empty) next to poisoned memory
    L = 5
    big = torch.full((L + 1, L, L, L), float("nan"), dtype=torch.complex128, device="cuda")
    uu = rng.standard_normal((L,) * 4) + 1j * rng.standard_normal((L,) * 4)
    big[:L] = dev(uu)
    Cn = rng.standard_normal((L, L)) + 1j * rng.standard_normal((L, L))
    got = K.transform_two_body(big[:L], dev(Cn))
    assert "pair4s" in K.last_dispatch() and torch.isfinite(torch.view_as_real(got)).all()
    assert relerr(host(got), orc.transform_two_body(uu, Cn)) <= RTOL


def test_spf_golden(K, golden):
    g = golden("transform_spf")
    L = g["C"].shape[0]
    spf = g["spf"].reshape(L, -1)
    got = host(K.matmul(dev(g["C"].T.copy()), dev(spf))).reshape(g["spf_out"].shape)
    assert relerr(got, g["spf_out"]) <= RTOL
    bra = g["bra_spf"].reshape(L, -1)
    got = host(K.matmul(dev(g["C_tilde"]), dev(bra))).reshape(g["bra_out"].shape)
    assert relerr(got, g["bra_out"]) <= RTOL


@pytest.mark.parametrize("name", ["spin_statics_f64", "spin_statics_c128"])
def test_spin_statics_golden(K, golden, name):
    g = golden(name)
    u = dev(g["u"])
    assert np.array_equal(host(K.spin_expand_two_body(u)), g["u_spin"])
    assert np.array_equal(host(K.spin_expand_two_body(u, antisymmetrize=True)), g["u_spin_as"])
    assert np.array_equal(host(K.antisymmetrize(u)), g["u_as"])
    # unfused route: expand then anti-symmetrise in place
    us = K.spin_expand_two_body(u)
    K.antisymmetrize(us, out=us)
    assert np.array_equal(host(us), g["u_spin_as"])
    if "h" in g:
        assert np.array_equal(host(K.add_spin_one_body(dev(g["h"]))), g["h_spin"])
    # fused complex cast
    got = host(K.spin_expand_two_body(u, antisymmetrize=True, out_dtype=torch.complex128))
    assert got.dtype == np.complex128
    assert np.array_equal(got, g["u_spin_as"].astype(np.complex128))


def test_config1_u_golden(K, golden):
    g = golden("config1_l20_change_basis")
    np.random.seed(int(g["seed"]))
    st = orc.random_basis(20, 2)
    C = g["C"]
    got = host(K.transform_two_body(dev(st["u"]), dev(C)))
    assert relerr(got, g["u"]) <= RTOL
    np.testing.assert_allclose(got, g["u"], rtol=1e-10, atol=1e-12)
    for k in ("h", "s"):
        assert relerr(host(K.transform_one_body(dev(st[k]), dev(C))), g[k]) <= RTOL
    pos = host(K.transform_one_body(dev(st["position"]), dev(C)))
    assert pos.shape == g["position"].shape and relerr(pos, g["position"]) <= RTOL


def test_spin_squared_two_body_golden(K, golden):
    g = golden("gos_l5_default_spinors")
    S = np.stack([g["gos_spin_x"], g["gos_spin_y"], g["gos_spin_z"]])
    got = host(K.spin_squared_two_body(dev(S), antisymmetrize=True))
    np.testing.assert_allclose(got, g["gos_spin_2_tb"], rtol=1e-13, atol=1e-14)
    g = golden("gos_l4_custom_spinors_no_as")
    S = np.stack([g["gos_spin_x"], g["gos_spin_y"], g["gos_spin_z"]])
    got = host(K.spin_squared_two_body(dev(S), antisymmetrize=False))
    np.testing.assert_allclose(got, g["gos_spin_2_tb"], rtol=1e-13, atol=1e-14)
    # slab form
    part = host(K.spin_squared_two_body(dev(S), antisymmetrize=False, p_lo=2, p_hi=5))
    np.testing.assert_allclose(part, g["gos_spin_2_tb"][2:5], rtol=1e-13, atol=1e-14)


# ------------------------------------------------ oracle on seeded inputs


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize(
    "L,M",
    [(1, 1), (2, 3), (5, 5), (16, 16), (17, 13), (20, 20), (10, 18), (33, 32), (55, 55), (64, 64), (70, 40),
     # the automatic route above 70 orbitals, directly against the oracle (VERDICT r03 "next" 6): the streamed kernel with two
     # workgroups per item quad (fp64 65-95), the strip kernels (fp64 from 96, complex128 65-128), rectangular C
     (78, 78), (91, 91), (96, 93), (84, 81), (100, 100), (113, 97)],
)
def test_transform_vs_oracle(K, L, M, cplx):
    rng = np.random.default_rng(1000 * L + M + cplx)
    if cplx:
        u, C, Ct = crand(rng, L, L, L, L) - 0.5, crand(rng, L, M) - 0.5, crand(rng, M, L)
    else:
        u = rng.standard_normal((L, L, L, L))
        C = rng.standard_normal((L, M))
        Ct = rng.standard_normal((M, L))
    # default bra = C^dagger
    got = host(K.transform_two_body(dev(u), dev(C)))
    assert relerr(got, orc.transform_two_body(u, C)) <= RTOL
    # explicit, unrelated C_tilde (bi-orthogonal bases)
    got = host(K.transform_two_body(dev(u), dev(C), dev(Ct)))
    assert relerr(got, orc.transform_two_body(u, C, Ct)) <= RTOL
    h = u[0, 0]
    assert relerr(host(K.transform_one_body(dev(h), dev(C), dev(Ct))),
                  orc.transform_one_body(h, C, Ct)) <= RTOL


def test_transform_reference_test_case(K):
    # the reference's own check (tests/test_helper.py:38-69): l=10 complex,
    # non-unitary C, against the 5-operand einsum, atol 1e-10
    rng = np.random.default_rng(5)
    l = 10
    u, C = crand(rng, l, l, l, l), crand(rng, l, l)
    ref = np.einsum("ls,kr,jq,ip,ijkl->pqrs", C, C, C.conj(), C.conj(), u, optimize=True)
    got = host(K.transform_two_body(dev(u), dev(C)))
    np.testing.assert_allclose(got, ref, atol=1e-10)
    got2 = host(K.transform_two_body(dev(u), dev(C), dev(C.conj().T.copy())))
    np.testing.assert_allclose(got, got2)


def test_transform_does_not_touch_input_and_handles_views(K):
    rng = np.random.default_rng(6)
    l = 12
    u, C = crand(rng, l, l, l, l), crand(rng, l, l)
    du = dev(u)
    keep = du.clone()
    # lazy-conjugated, transposed view as C_tilde (torch's .conj() is a view)
    dC = dev(C)
    got = host(K.transform_two_body(du, dC, dC.conj().T))
    assert torch.equal(du, keep)
    assert relerr(got, orc.transform_two_body(u, C)) <= RTOL
    # non-contiguous u view
    big = dev(crand(rng, l, l, l, 2 * l))
    view = big[..., ::2]
    got = host(K.transform_two_body(view, dC))
    assert relerr(got, orc.transform_two_body(host(view), C)) <= RTOL


def test_partial_transform_matches_slab_of_full(K):
    rng = np.random.default_rng(7)
    L, M = 12, 9
    u, C, Ct = crand(rng, L, L, L, L), crand(rng, L, M), crand(rng, M, L)
    # v[a,q,r,s] = Ct[q,b] u[a,b,c,d] C[c,r] C[d,s]
    ref = np.einsum("qb,abcd,cr,ds->aqrs", Ct, u, C, C, optimize=True)
    got = host(K.transform_two_body_partial(dev(u[3:8]), dev(C), dev(Ct)))
    assert relerr(got, ref[3:8]) <= RTOL
    # closing the contraction over a with a plain product gives the full result
    v = K.transform_two_body_partial(dev(u), dev(C), dev(Ct))
    full = host(K.matmul(dev(Ct), v.reshape(L, -1))).reshape(M, M, M, M)
    assert relerr(full, orc.transform_two_body(u, C, Ct)) <= RTOL


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("l", [1, 2, 7, 31, 32, 33, 64, 70])
def test_antisymmetrize_vs_oracle(K, l, cplx):
    rng = np.random.default_rng(l)
    n = 3 if l > 32 else l
    u = rng.standard_normal((n, n, l, l))
    if cplx:
        u = u + 1j * rng.standard_normal((n, n, l, l))
    ref = u - u.transpose(0, 1, 3, 2)
    d = dev(u)
    assert np.array_equal(host(K.antisymmetrize(d)), ref)
    K.antisymmetrize(d, out=d)  # in place
    assert np.array_equal(host(d), ref)


@pytest.mark.parametrize("l", [1, 3, 8, 33])
@pytest.mark.parametrize("mode", ["f64", "c128", "f64->c128"])
def test_spin_expand_vs_oracle(K, l, mode):
    rng = np.random.default_rng(l)
    u = rng.standard_normal((l, l, l, l))
    odt = None
    if mode == "c128":
        u = u + 1j * rng.standard_normal((l, l, l, l))
    if mode == "f64->c128":
        odt = torch.complex128
    us = orc.add_spin_two_body(u)
    for anti in (False, True):
        ref = orc.anti_symmetrize_u(us) if anti else us
        if odt is not None:
            ref = ref.astype(np.complex128)
        got = host(K.spin_expand_two_body(dev(u), antisymmetrize=anti, out_dtype=odt))
        assert got.dtype == ref.dtype and np.array_equal(got, ref)
    # p-slab form writes rows [2 p_lo, 2 p_hi)
    if l >= 3:
        ref = orc.anti_symmetrize_u(us)
        got = host(K.spin_expand_two_body(dev(u), antisymmetrize=True, p_lo=1, p_hi=3))
        assert np.array_equal(got, ref[2:6])


def test_spin_index_law_like_reference(K):
    # reference tests/test_helper.py:87-135 (explicit spin_delta loops)
    rng = np.random.default_rng(8)
    u = rng.random((4, 4, 4, 4))
    u = u + u.transpose(1, 0, 3, 2)
    ref = orc.spin_two_body_index_law(u)
    got = host(K.spin_expand_two_body(dev(u), antisymmetrize=True))
    np.testing.assert_allclose(got, ref, atol=1e-10)
    assert np.array_equal(got, ref)
    # antisymmetry properties (tests/test_helper.py:138-147)
    np.testing.assert_allclose(got, -got.transpose(0, 1, 3, 2), atol=1e-10)
    np.testing.assert_allclose(got, -got.transpose(1, 0, 2, 3), atol=1e-10)
    np.testing.assert_allclose(got, got.transpose(1, 0, 3, 2), atol=1e-10)


def test_add_spin_one_body_vs_oracle(K):
    rng = np.random.default_rng(9)
    for shape in [(1, 1), (5, 5), (3, 6, 6)]:
        h = rng.standard_normal(shape)
        ref = (
            np.stack([orc.add_spin_one_body(m) for m in h]) if h.ndim == 3 else orc.add_spin_one_body(h)
        )
        assert np.array_equal(host(K.add_spin_one_body(dev(h))), ref)
        got = host(K.add_spin_one_body(dev(h), out_dtype=torch.complex128))
        assert np.array_equal(got, ref.astype(np.complex128))


# ----------------------------------------- larger sizes: properties only


@pytest.mark.parametrize("dt", [torch.float64, torch.complex128])
def test_randomised_identity_l96(K, dt):
    # sum out.x(x)y(x)z(x)w == sum u.(Ct^T x)(x)(Ct^T y)(x)(C z)(x)(C w)   
    l = 96
    g = torch.Generator(device="cuda:0").manual_seed(3)
    u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt)
    C = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt)
    if dt == torch.complex128:
        C = C + 1j * torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g)
    C = C / l**0.5
    Ct = C.conj().T.contiguous()
    out = K.transform_two_body(u, C, Ct)
    x, y, z, w = (torch.randn(l, dtype=torch.float64, device="cuda:0", generator=g).to(dt) for _ in range(4))
    lhs = torch.einsum("pqrs,p,q,r,s->", out, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()
    # linearity in u
    out2 = K.transform_two_body(2.5 * u, C, Ct)
    assert (out2 - 2.5 * out).abs().max().item() <= 1e-10 * out.abs().max().item()


def test_unitary_round_trip_l64(K):
    # transform with a unitary C then with C^dagger returns u
    l = 64
    g = torch.Generator(device="cuda:0").manual_seed(4)
    u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda:0", generator=g)
    Q, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g))
    back = K.transform_two_body(K.transform_two_body(u, Q), Q.T.contiguous())
    assert (back - u).abs().max().item() <= 1e-10


def test_bad_arguments_raise(K):
    u = torch.zeros(4, 4, 4, 4, dtype=torch.float64, device="cuda:0")
    C = torch.zeros(5, 4, dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError):
        K.transform_two_body(u, C)
    with pytest.raises(RuntimeError):
        K.transform_two_body(u.cpu(), torch.zeros(4, 4, dtype=torch.float64))
    with pytest.raises(TypeError):
        K.antisymmetrize(u.to(torch.float16))


# ------------------------------------------- exact-tiling fast path (qs_gemm_fast.hip)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,n,k,batch", [(128, 128, 16, 1), (256, 128, 48, 1), (128, 256, 64, 3),
                                          (64, 128, 24, 2), (64, 64, 8, 5), (384, 128, 40, 1)])
def test_fast_path_product_vs_oracle(K, m, n, k, batch, cplx):
    # products whose extents are whole tiles take the scalar-addressed kernel;
    # checked against NumPy's matmul (what np.tensordot lowers to)
    rng = np.random.default_rng(m + n + k + batch + cplx)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n))
    if cplx:
        A = A + 1j * rng.standard_normal((m, k))
        B = B + 1j * rng.standard_normal((batch, k, n))
    ref = np.matmul(A, B)
    for fast in (1, 0):
        K.tuning_set("gemm_fast", fast)
        got = host(K.matmul(dev(A), dev(B)))
        assert relerr(got, ref) <= 1e-13, f"fast={fast}"
        # accumulate form: out += A.B
        out = dev(ref.copy())
        K.matmul(dev(A), dev(B), out=out, accumulate=True)
        assert relerr(host(out), 2 * ref) <= 1e-13, f"fast={fast} accumulate"
    K.tuning_set("gemm_fast", 1)


@pytest.mark.parametrize("dt", [torch.float64, torch.complex128])
def test_fast_and_general_kernels_agree_l128(K, dt):
    l = 128
    g = torch.Generator(device="cuda:0").manual_seed(11)
    u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt)
    C = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt) / l**0.5
    if dt == torch.complex128:
        C = C + 1j * torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g) / l**0.5
    Ct = C.conj().T.contiguous()
    K.tuning_set("gemm_fast", 1)
    fast = K.transform_two_body(u, C, Ct)
    K.tuning_set("gemm_fast", 0)
    gen = K.transform_two_body(u, C, Ct)
    K.tuning_set("gemm_fast", 1)
    scale = gen.abs().max().item()
    assert (fast - gen).abs().max().item() <= 1e-12 * scale
    x, y, z, w = (torch.randn(l, dtype=torch.float64, device="cuda:0", generator=g).to(dt) for _ in range(4))
    lhs = torch.einsum("pqrs,p,q,r,s->", fast, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()


# ------------------------------------------- edge form of the fast kernel (any extent)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,n,k,batch", [(1, 1, 1, 1), (55, 55, 55, 3), (166, 55, 55, 1), (100, 100, 100, 2),
                                          (130, 258, 17, 1), (200, 136, 40, 2), (129, 65, 33, 1),
                                          (64, 300, 15, 1), (257, 102, 96, 1)])
def test_edge_form_product_vs_oracle(K, m, n, k, batch, cplx):
    # gemm_fast = 3 routes every product through the VALU-free kernel's edge form: odd extents
    # take its 8-byte accesses, even ones the 16-byte ones; K tails are zeroed in registers
    rng = np.random.default_rng(m * 7 + n * 3 + k + batch + cplx)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n))
    if cplx:
        A = A + 1j * rng.standard_normal((m, k))
        B = B + 1j * rng.standard_normal((batch, k, n))
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_fast", 0)
        general = host(K.matmul(dev(A), dev(B)))
        for shape in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
            K.tuning_set("gemm_fast", 3)
            K.tuning_set("gemm_fast_shape", shape)
            got = host(K.matmul(dev(A), dev(B)))
            assert relerr(got, ref) <= 1e-13, f"shape={shape}"
            # same k order as the general kernel: identical bits
            assert np.array_equal(got, general), f"shape={shape}"
            out = dev(ref.copy())
            K.matmul(dev(A), dev(B), out=out, accumulate=True)
            assert relerr(host(out), 2 * ref) <= 1e-13, f"shape={shape} accumulate"
    finally:
        K.tuning_set("gemm_fast", 1)
        K.tuning_set("gemm_fast_shape", 0)


@pytest.mark.parametrize("m,n,k,batch", [(66, 4356, 66, 2), (130, 1000, 130, 1), (255, 300, 255, 1), (81, 81, 81, 5), (1000, 130, 130, 1),
                                          (4356, 66, 66, 1), (500, 255, 77, 2), (97, 200, 97, 3), (144, 144, 144, 2)])
def test_fitted_tile_shapes_vs_oracle(K, m, n, k, batch):
    # the general kernel's FITTED shapes (round 3): the basis size covered by one tile, to the next multiple of 16 --
    # (16 t) x 64 when it is the m extent (c, b, a contractions), 64 x (16 t) when it is n (d, c); odd t on 8-byte staging
    rng = np.random.default_rng(m + n + k + batch)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n))
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_fast", 0)
        K.tuning_set("gemm_fit", 0)
        K.tuning_set("gemm_strip", 0)          # (round 4: the strip kernels would take these products first)
        general = host(K.matmul(dev(A), dev(B)))
        assert "gemm_kernel<2, 2" in K.last_dispatch()
        K.tuning_set("gemm_fit", 2)
        got = host(K.matmul(dev(A), dev(B)))
        ran = K.last_dispatch()
        assert "gemm_kernel<1, 4, " in ran or "gemm_kernel<4, 1, 1, " in ran, ran
        assert relerr(got, ref) <= 1e-13
        assert np.array_equal(got, general)                # the same k-ordered sums: identical bits
        out = dev(ref.copy())
        K.matmul(dev(A), dev(B), out=out, accumulate=True)
        assert relerr(host(out), 2 * ref) <= 1e-13
    finally:
        K.tuning_reset()


@pytest.mark.parametrize("m,n,k,batch", [
    (130, 130, 130, 40),       # contraction c of l = 130: a batch of narrow products tiles as one long row of virtual columns
    (100, 10000, 100, 1),      # contraction a-like: one product, a huge column extent
    (144, 144 * 144, 144, 3),  # contraction b-like, whole blocks of 16
    (129, 129, 129, 17), (97, 97 * 5, 97, 7),       # odd sizes: 8-byte items
    (253, 600, 253, 2), (256, 384, 256, 2),         # sixteen row blocks (one register set)
    (20, 300, 20, 5), (1, 7, 1, 3), (17, 33, 5, 2),
    (300, 700, 300, 2), (288, 288 * 3, 288, 1), (3000, 300, 300, 1), (700, 513, 40, 1), (520, 300, 77, 3),   # SEVERAL tiles along the small extent
    (70000, 130, 130, 1),      # contraction d: the rows of the tensor against the coefficient matrix (wide tiles)
    (4097, 100, 100, 1), (2000, 97, 97, 1), (1500, 253, 253, 1), (1000, 256, 256, 1), (300, 7, 3, 1), (129, 16, 4, 1),
])
def test_strip_kernels_vs_oracle_and_bit_identical_to_the_general_kernel(K, m, n, k, batch):
    # round 4: the small extent of a product covered by ONE tile to the next multiple of 16, eight waves per workgroup
    # (qs_gemm_strip.hip): tall tiles over virtual columns when the left operand is shared and small, wide tiles over the
    # rows when the right one is
    rng = np.random.default_rng(m + 3 * n + 5 * k + batch)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n)) if batch > 1 else rng.standard_normal((k, n))
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_fast", 0)
        K.tuning_set("gemm_fit", 0)
        K.tuning_set("gemm_strip", 0)
        K.tuning_set("gemm_stream", 0)
        K.tuning_set("gemm_skinny", 0)
        general = host(K.matmul(dev(A), dev(B)))
        assert "gemm_kernel<2, 2" in K.last_dispatch(), K.last_dispatch()
        K.tuning_set("gemm_strip", 2)
        got = host(K.matmul(dev(A), dev(B)))
        ran = K.last_dispatch()
        assert "gemm_strip_kernel<" in ran, ran
        assert relerr(got, ref) <= 1e-13
        assert np.array_equal(got, general)                # the same k-ordered sums: identical bits
    finally:
        K.tuning_reset()


@pytest.mark.parametrize("m,n,k,batch", [
    (66, 66, 66, 30), (80, 6400, 80, 3), (100, 100000, 100, 1), (127, 127, 127, 9), (128, 384, 128, 2), (112, 112 * 3, 112, 5),
    (130, 130, 130, 9), (153, 306, 153, 3), (200, 400, 200, 1), (2000, 153, 153, 1), (1000, 250, 250, 1), (300, 260, 31, 2),   # several tiles along the small extent
    (20, 300, 20, 5), (1, 7, 1, 3), (17, 33, 5, 2),
    (30000, 66, 66, 1), (4097, 100, 100, 1), (2000, 97, 97, 1), (1500, 127, 127, 1), (1000, 128, 128, 1), (300, 7, 3, 1),
])
def test_complex_strip_kernels_vs_oracle_and_bit_identical_to_the_general_kernel(K, m, n, k, batch):
    # the complex128 form of the strip kernels (re / im planes, four real matrix instructions per fragment pair in the order of
    # the other complex kernels): bases of up to 128 complex orbitals -- every RandomBasisSet and every spin-doubled tensor
    rng = np.random.default_rng(m + 3 * n + 5 * k + batch)
    A = crand(rng, m, k)
    B = crand(rng, batch, k, n) if batch > 1 else crand(rng, k, n)
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_fast", 0)
        K.tuning_set("gemm_strip", 0)
        K.tuning_set("gemm_stream", 0)
        K.tuning_set("gemm_skinny", 0)
        general = host(K.matmul(dev(A), dev(B)))
        assert "gemm_kernel<" in K.last_dispatch(), K.last_dispatch()
        K.tuning_set("gemm_strip", 2)
        got = host(K.matmul(dev(A), dev(B)))
        ran = K.last_dispatch()
        assert "gemm_strip_kernel<true, " in ran, ran
        assert relerr(got, ref) <= 1e-13
        assert np.array_equal(got, general)                # the same k-ordered sums: identical bits
    finally:
        K.tuning_reset()


@pytest.mark.parametrize("L,M", [(66, 66), (72, 70), (80, 80), (100, 100), (91, 112), (127, 127), (130, 130), (144, 150)])
def test_complex_transform_on_the_strip_kernels_equals_the_general_kernel(K, L, M):
    g = torch.Generator(device="cuda:0").manual_seed(L + M)
    u = torch.complex(torch.rand(L, L, L, L, dtype=torch.float64, device="cuda:0", generator=g),
                      torch.rand(L, L, L, L, dtype=torch.float64, device="cuda:0", generator=g))
    C = torch.complex(torch.randn(L, M, dtype=torch.float64, device="cuda:0", generator=g),
                      torch.randn(L, M, dtype=torch.float64, device="cuda:0", generator=g)) / L**0.5
    Ct = C.conj().T.contiguous()
    try:
        K.tuning_set("gemm_fast", 0)
        K.tuning_set("gemm_strip", 0)
        K.tuning_set("pair4c", 0)
        gen = K.transform_two_body(u, C, Ct)
        assert "gemm_strip" not in K.last_dispatch()
        K.tuning_set("gemm_strip", 2)
        got = K.transform_two_body(u, C, Ct)
        ran = K.last_dispatch()
        assert "gemm_strip_kernel<true, 1, " in ran and "gemm_strip_kernel<true, 0, " in ran and "gemm_kernel" not in ran, ran
        assert torch.equal(got, gen)
        K.tuning_reset()
        assert torch.equal(K.transform_two_body(u, C, Ct), gen)        # whatever the automatic choice is: the same bits
    finally:
        K.tuning_reset()
    x, y, z, w = (torch.randn(M, dtype=torch.float64, device="cuda:0", generator=g).to(torch.complex128) for _ in range(4))
    lhs = torch.einsum("pqrs,p,q,r,s->", got, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()


def test_strip_kernels_keep_non_finite_values_in_their_rows_and_columns(K):
    # rows / columns beyond the small extent and k >= K multiply whatever the loads returned: nothing of it may reach a
    # stored element (selects on the K tail, never-stored blocks elsewhere)
    rng = np.random.default_rng(78)
    try:
        K.tuning_set("gemm_strip", 2)
        K.tuning_set("gemm_stream", 0)
        K.tuning_set("gemm_skinny", 0)
        for (m, n, k, batch) in [(70, 58, 21, 3), (3000, 58, 21, 1)]:
            A = rng.standard_normal((m, k))               # lda == k: the tail of row i is the head of row i + 1
            B = rng.standard_normal((batch, k, n)) if batch > 1 else rng.standard_normal((k, n))
            A[31, 0] = np.nan
            A[40, 3] = np.inf
            B[..., 2, 57] = np.nan
            B[..., 20, 0] = np.inf
            with np.errstate(invalid="ignore"):
                ref = np.matmul(A, B)
            got = host(K.matmul(dev(A), dev(B)))
            assert "gemm_strip_kernel<" in K.last_dispatch()
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            ok = np.isfinite(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-12)
    finally:
        K.tuning_reset()


@pytest.mark.parametrize("L,M", [(100, 100), (129, 129), (130, 130), (97, 113), (144, 120), (150, 150), (84, 81)])
def test_transform_on_the_strip_kernels_equals_the_general_kernel_and_the_oracle(K, L, M):
    rng = np.random.default_rng(L * 7 + M)
    u = rng.standard_normal((L, L, L, L)) if L <= 100 else None
    g = torch.Generator(device="cuda:0").manual_seed(L)
    ud = dev(u) if u is not None else torch.rand(L, L, L, L, dtype=torch.float64, device="cuda:0", generator=g)
    C = torch.randn(L, M, dtype=torch.float64, device="cuda:0", generator=g) / L**0.5
    Ct = torch.randn(M, L, dtype=torch.float64, device="cuda:0", generator=g) / L**0.5
    try:
        K.tuning_set("gemm_fast", 0)
        K.tuning_set("gemm_fit", 0)
        K.tuning_set("gemm_strip", 0)
        K.tuning_set("quad4s", 0)
        gen = K.transform_two_body(ud, C, Ct)
        assert "gemm_strip" not in K.last_dispatch()
        K.tuning_set("gemm_strip", 2)
        got = K.transform_two_body(ud, C, Ct)
        ran = K.last_dispatch()
        assert "gemm_strip_kernel<false, 1, " in ran and "gemm_strip_kernel<false, 0, " in ran and "gemm_kernel" not in ran and "gemm_fast" not in ran, ran
        assert torch.equal(got, gen)
        K.tuning_reset()
        auto = K.transform_two_body(ud, C, Ct)              # whatever the automatic choice is: the same bits
        assert torch.equal(auto, gen)
    finally:
        K.tuning_reset()
    if u is not None:
        ref = orc.transform_two_body(u, host(C), host(Ct))
        assert relerr(host(got), ref) <= 1e-12
    x, y, z, w = (torch.randn(n_, dtype=torch.float64, device="cuda:0", generator=g) for n_ in (M, M, M, M))
    lhs = torch.einsum("pqrs,p,q,r,s->", got, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", ud, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()


@pytest.mark.parametrize("m,n,k,batch", [(253, 300, 253, 2), (129, 129, 129, 5), (1000, 97, 97, 1), (255, 255, 255, 1), (131, 77, 35, 3)])
def test_edge_form_with_16_byte_items_at_odd_strides_is_bit_identical(K, m, n, k, batch):
    # round 4: gfx950 carries out 16-byte buffer loads and stores at any 8-byte-aligned address, so the VALU-free kernel's edge
    # form stages odd sizes with 16-byte items too (tuning key gemm_fast_unaligned = 0: the 8-byte items of rounds 1-3)
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n))
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_strip", 0)
        K.tuning_set("gemm_fast", 3)
        K.tuning_set("gemm_fast_unaligned", 0)
        old = host(K.matmul(dev(A), dev(B)))
        assert ", false, true>" in K.last_dispatch(), K.last_dispatch()
        K.tuning_set("gemm_fast_unaligned", 1)
        got = host(K.matmul(dev(A), dev(B)))
        assert ", true, true>" in K.last_dispatch(), K.last_dispatch()
        assert relerr(got, ref) <= 1e-13
        assert np.array_equal(got, old)
        out = dev(ref.copy())
        K.matmul(dev(A), dev(B), out=out, accumulate=True)
        assert relerr(host(out), 2 * ref) <= 1e-13
    finally:
        K.tuning_reset()


def test_edge_form_keeps_non_finite_values_in_their_rows(K):
    # the K tail is removed with selects, not by multiplying with zero: a NaN/Inf in one row of A
    # (or one column of B) must not reach any other row (column) of the product
    rng = np.random.default_rng(77)
    m, n, k = 70, 58, 21          # lda == k: the tail of row i is the head of row i + 1
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    A[31, 0] = np.nan
    A[40, 3] = np.inf
    B[2, 57] = np.nan
    with np.errstate(invalid="ignore"):
        ref = A @ B
    try:
        K.tuning_set("gemm_fast", 3)
        got = host(K.matmul(dev(A), dev(B)))
    finally:
        K.tuning_set("gemm_fast", 1)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = np.isfinite(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dt,l", [(torch.float64, 55), (torch.float64, 100), (torch.complex128, 36)])
def test_edge_form_transform_equals_general_kernel(K, dt, l):
    g = torch.Generator(device="cuda:0").manual_seed(l)
    u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt)
    C = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g).to(dt) / l**0.5
    Ct = C.conj().T.contiguous()
    try:
        K.tuning_set("gemm_fast", 0)
        gen = K.transform_two_body(u, C, Ct)
        K.tuning_set("gemm_fast", 3)
        edge = K.transform_two_body(u, C, Ct)
    finally:
        K.tuning_set("gemm_fast", 1)
    assert torch.equal(edge, gen)
    x, y, z, w = (torch.randn(l, dtype=torch.float64, device="cuda:0", generator=g).to(dt) for _ in range(4))
    lhs = torch.einsum("pqrs,p,q,r,s->", edge, x, y, z, w)
    rhs = torch.einsum("abcd,a,b,c,d->", u, Ct.T @ x, Ct.T @ y, C @ z, C @ w)
    assert abs(lhs - rhs).item() <= 1e-10 * abs(rhs).item()


# ------------------------------------------- short-and-wide streaming product (qs_gemm_skinny.hip)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,k", [(16, 8), (32, 64), (48, 20), (64, 36)])
def test_skinny_product_vs_oracle(K, m, k, cplx):
    # the leading-index contraction of the sharded layout: few rows of Ct times the whole tensor
    rng = np.random.default_rng(m * 100 + k + cplx)
    n = 1 << 16
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    if cplx:
        A = A + 1j * rng.standard_normal((m, k))
        B = B + 1j * rng.standard_normal((k, n))
    ref = A @ B
    for skinny in (1, 0):
        K.tuning_set("gemm_skinny", skinny)
        got = host(K.matmul(dev(A), dev(B)))
        assert relerr(got, ref) <= 1e-13, f"skinny={skinny}"
    K.tuning_set("gemm_skinny", 1)


# ------------------------------------------- small-coefficient streaming product (qs_gemm_stream.hip)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,k,n,batch", [(55, 55, 55 * 55, 55), (55, 55, 55, 3025), (64, 64, 4096, 16),
                                          (1, 1, 70000, 1), (17, 3, 1000, 70), (40, 61, 33, 2100),
                                          (64, 5, 100001, 1), (33, 64, 130, 600)])
def test_stream_product_vs_oracle(K, m, k, n, batch, cplx):
    # the c, b, a contractions of a small-l transform: A (m x k) is Ct or C^T, shared by the batch
    rng = np.random.default_rng(m * 1000 + k * 10 + batch)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((batch, k, n))
    if cplx:
        A = A + 1j * rng.standard_normal((m, k))
        B = B + 1j * rng.standard_normal((batch, k, n))
    ref = np.matmul(A, B)
    try:
        K.tuning_set("gemm_stream", 0)
        general = host(K.matmul(dev(A), dev(B)))
        for knob in (1, 2):       # 2 = never split the rows of A over two waves
            K.tuning_set("gemm_stream", knob)
            got = host(K.matmul(dev(A), dev(B)))
            assert relerr(got, ref) <= 1e-13, f"gemm_stream={knob}"
            assert np.array_equal(got, general), f"gemm_stream={knob}: same k order as the tiled kernels"
    finally:
        K.tuning_set("gemm_stream", 1)


def test_stream_product_keeps_non_finite_values_in_their_columns(K):
    rng = np.random.default_rng(5)
    m, k, n = 30, 23, 70001            # k tail of 3, odd n: 8-byte accesses, clamped last block
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    B[22, 70000] = np.nan
    B[0, 17] = np.inf
    with np.errstate(invalid="ignore"):
        ref = A @ B
    got = host(K.matmul(dev(A), dev(B)))
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = np.isfinite(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-12)


# ------------------------------------------- fused (d, c) pass for small bases (qs_slab_pair.hip)


@pytest.mark.parametrize("L,M", [(16, 16), (20, 33), (32, 32), (33, 17), (48, 64), (55, 55), (64, 64), (64, 5), (41, 60)])
def test_fused_dc_pass_equals_two_products(K, L, M):
    # d and c in one pass (Y = X.C stays in the MFMA accumulators): same chains of fused multiply-adds
    # as the two separate products, so the whole transform is bit-identical with and without it
    rng = np.random.default_rng(L * 100 + M)
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, M)) / np.sqrt(L)
    Ct = rng.standard_normal((M, L)) / np.sqrt(L)
    ref = orc.transform_two_body(u, C, Ct)
    try:
        K.tuning_set("slab_pair", 0)
        plain = host(K.transform_two_body(dev(u), dev(C), dev(Ct)))
        K.tuning_set("slab_pair", 2)          # one wave per slab
        fused1 = host(K.transform_two_body(dev(u), dev(C), dev(Ct)))
        K.tuning_set("slab_pair", 1)          # automatic: two waves per slab when M spans an even number of tiles
        fused = host(K.transform_two_body(dev(u), dev(C), dev(Ct)))
        part = host(K.transform_two_body_partial(dev(u[3:9]), dev(C), dev(Ct)))
    finally:
        K.tuning_set("slab_pair", 1)
    assert relerr(fused, ref) <= 1e-13
    assert np.array_equal(fused, plain) and np.array_equal(fused1, plain)
    np.testing.assert_allclose(part, orc.transform_two_body_dcb(u[3:9], C, Ct), rtol=1e-12, atol=1e-13)


def test_fused_dc_pass_keeps_non_finite_values_in_their_slabs(K):
    # K tails of a slab alias its next row and the row tail of a slab aliases the NEXT slab: a NaN / Inf
    # at the very start of slab (4, 0) / (6, 0) must not reach slab (3, L-1) / (5, L-1).  (Checked per
    # leading index a: the b contraction that follows mixes the slabs of one a, as it does in NumPy.)
    rng = np.random.default_rng(6)
    L = 21
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, L))
    u[4, 0, 0, 0] = np.nan
    u[6, 0, 0, 1] = np.inf
    clean = np.delete(u, [4, 6], axis=0)
    ref = orc.transform_two_body_dcb(clean, C, C.T.copy())
    got = host(K.transform_two_body_partial(dev(u), dev(C), dev(C.T.copy())))
    assert not np.isfinite(got[4]).any() and not np.isfinite(got[6]).any()
    rest = np.delete(got, [4, 6], axis=0)
    assert np.isfinite(rest).all()
    np.testing.assert_allclose(rest, ref, rtol=1e-11, atol=1e-11)


# ------------------------- both halves of a small-basis transform on the 4-wide instruction (qs_sandwich4.hip)


def _sandwich_launches(dispatch, n4):
    """Number of launches in a dispatch record that are one of the two 4-wide kernels for ceil(l/4) = n4 (0 if anything
    else ran).  The balanced kernel's instantiation for an even N4 also takes ceil(l/4) = N4 - 1."""
    import re

    total = 0
    for item in dispatch.split(";"):
        m = re.fullmatch(rf"qs::sandwich4_kernel<{n4}>(?: x(\d+))?", item)
        if not m:
            m = re.fullmatch(rf"qs::sandwich4b_kernel<({n4}|{n4 + (n4 & 1)})>(?:\+tail)?(?: x(\d+))?", item)
            if not m:
                return 0
            total += int(m.group(2) or 1)
        else:
            total += int(m.group(1) or 1)
    return total


@pytest.mark.parametrize("L,M", [(32, 32), (33, 33), (36, 34), (40, 40), (41, 44), (46, 46), (47, 48), (48, 48), (50, 49),
                                 (53, 55), (55, 55), (55, 53), (56, 56), (57, 57), (58, 60), (61, 64), (64, 64)])
def test_sandwich_passes_bit_identical_to_the_16_wide_path(K, L, M):
    # (d, c) per slab and (b, a) per column as Out = Lm . In . R on v_mfma_f64_4x4x4_4b_f64: four items per
    # instruction, extents padded to 4 instead of 16, Y chained through the accumulators.  The sums are the same
    # k-ordered FMA chains, so the transform is bit-identical to the 16-wide kernels' -- with the kernel forced
    # wherever it is legal (sandwich = 4), in every work split, for each pass alone, and in the automatic choice
    rng = np.random.default_rng(L * 100 + M)
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, M)) / np.sqrt(L)
    Ct = rng.standard_normal((M, L)) / np.sqrt(L)
    ref = orc.transform_two_body(u, C, Ct)
    du, dC, dCt = dev(u), dev(C), dev(Ct)
    with K.tuning(sandwich=0):
        plain = host(K.transform_two_body(du, dC, dCt))
        assert "sandwich4" not in K.last_dispatch()
        plain_part = host(K.transform_two_body_partial(du[3:L - 2], dC, dCt))
    assert relerr(plain, ref) <= 1e-13
    auto = host(K.transform_two_body(du, dC, dCt))
    assert np.array_equal(auto, plain)
    # (sandwich_t2: the intermediate between the two passes in its natural layout / transposed)
    # sandwich_v2: the balanced kernel with a cooperative fetch (qs_sandwich4b.hip) never / wherever it exists)
    for knobs in (dict(sandwich=4), dict(sandwich=4, sandwich_mode=0), dict(sandwich=4, sandwich_mode=1),
                  dict(sandwich=4, sandwich_mode=3), dict(sandwich=4, sandwich_t2=0), dict(sandwich=4, sandwich_t2=1),
                  dict(sandwich=4, sandwich_v2=0), dict(sandwich=4, sandwich_v2=1, sandwich_t2=1),
                  dict(sandwich=4, sandwich_v2=1, sandwich_t2=0), dict(sandwich=1, sandwich_v2=1),
                  dict(sandwich=1, sandwich_t2=0), dict(sandwich=1, sandwich_t2=1), dict(sandwich=2), dict(sandwich=3)):
        with K.tuning(**knobs):
            got = host(K.transform_two_body(du, dC, dCt))
            disp = K.last_dispatch()
            part = host(K.transform_two_body_partial(du[3:L - 2], dC, dCt))
        assert np.array_equal(got, plain), knobs
        assert np.array_equal(part, plain_part), knobs
        # (ceil(l/4) = 15 has its second pass only on slabs through the balanced kernel: without either, one fused pass)
        if knobs.get("sandwich") == 4 and not (-(-L // 4) == 15 and 0 in (knobs.get("sandwich_t2"), knobs.get("sandwich_v2"))):
            assert _sandwich_launches(disp, -(-L // 4)) == 2, disp


@pytest.mark.parametrize("l", [46, 47, 48, 56, 57])
def test_sandwich_tail_splits_a_partly_filled_last_round(K, l):
    # 56 orbitals: 784 item quads = 3 rounds of the 256 workgroups + 16 quads; those are split by column groups over all
    # workgroups inside the launch (qs_sandwich4b.hip, "+tail" in the dispatch record) -- same bits with and without
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the sizes are chosen for 256 CUs")
    rng = np.random.default_rng(l)
    du = dev(rng.standard_normal((l,) * 4))
    dC = dev(np.linalg.qr(rng.standard_normal((l, l)))[0])
    with K.tuning(sandwich_tail=0):
        plain = K.transform_two_body(du, dC)
        assert "sandwich4" in K.last_dispatch() and "+tail" not in K.last_dispatch()
    got = K.transform_two_body(du, dC)
    n4 = -(-l // 4)      # (ceil(l/4) = 12 runs its second pass on columns, in the first kernel)
    assert K.last_dispatch().startswith(f"qs::sandwich4b_kernel<{n4 + (n4 & 1)}>+tail")
    assert torch.equal(got, plain)
    with K.tuning(sandwich=0):
        assert torch.equal(K.transform_two_body(du, dC), plain)


def test_sandwich_is_the_automatic_choice_for_config_2(K):
    # BASELINE.json configs[1]: l = 55 -> two launches of the 4-wide kernel
    g = torch.Generator(device="cuda:0").manual_seed(55)
    u = torch.rand(55, 55, 55, 55, dtype=torch.float64, device="cuda:0", generator=g)
    C, _ = torch.linalg.qr(torch.randn(55, 55, dtype=torch.float64, device="cuda:0", generator=g))
    out = K.transform_two_body(u, C.contiguous())
    assert K.last_dispatch() == "qs::sandwich4b_kernel<14> x2"
    ref = orc.transform_two_body(host(u), host(C))
    assert relerr(host(out), ref) <= 1e-13


def test_sandwich_pass_keeps_non_finite_values_in_their_slabs(K):
    # the padded rows / columns / items of a quad are parked lanes (hardware zeros), never neighbouring data:
    # a NaN at the start of slab (4, 0) and an Inf in slab (6, 0) stay in the slabs of a = 4 and a = 6
    rng = np.random.default_rng(61)
    L = 53
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, L)) / np.sqrt(L)
    u[4, 0, 0, 0] = np.nan
    u[6, 0, 0, 1] = np.inf
    with K.tuning(sandwich=4):
        got = host(K.transform_two_body_partial(dev(u[:21]), dev(C), dev(C.T.copy())))      # 21 x 53 slabs
        assert "sandwich4" in K.last_dispatch()
    clean = np.delete(u[:21], [4, 6], axis=0)
    ref = orc.transform_two_body_dcb(clean, C, C.T.copy())
    assert not np.isfinite(got[4]).any() and not np.isfinite(got[6]).any()
    rest = np.delete(got, [4, 6], axis=0)
    assert np.isfinite(rest).all()
    np.testing.assert_allclose(rest, ref, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("L", [33, 43, 53, 55, 63])
def test_sandwich_reads_nothing_outside_the_tensor(K, L):
    # The fetch takes 16 bytes per lane; a lane whose second element does not exist (the last k of an odd l in a slab,
    # the last item of an odd item count in a column) fetches 8 bytes earlier instead of 8 bytes past the end.  With
    # NaNs directly in front of and behind u (and behind the result buffer) the transform must not change by a bit.
    g = torch.Generator(device="cuda:0").manual_seed(L)
    n = L ** 4
    pad = 64
    big = torch.full((n + 2 * pad,), float("nan"), dtype=torch.float64, device="cuda:0")
    u = big[pad:pad + n].view(L, L, L, L)
    u.copy_(torch.rand(L, L, L, L, dtype=torch.float64, device="cuda:0", generator=g) - 0.5)
    C = torch.randn(L, L, dtype=torch.float64, device="cuda:0", generator=g) / np.sqrt(L)
    Ct = torch.randn(L, L, dtype=torch.float64, device="cuda:0", generator=g) / np.sqrt(L)
    alone = u.clone()
    with K.tuning(sandwich=0):
        ref = K.transform_two_body(alone, C, Ct)
    obig = torch.full((n + 2 * pad,), float("nan"), dtype=torch.float64, device="cuda:0")
    out = obig[pad:pad + n].view(L, L, L, L)
    with K.tuning(sandwich=4):
        K.transform_two_body(u, C, Ct, out=out)
        assert _sandwich_launches(K.last_dispatch(), -(-L // 4)) == 2
    assert torch.equal(out, ref)
    assert torch.isnan(obig[:pad]).all() and torch.isnan(obig[pad + n:]).all()      # nothing written outside either
    assert torch.isnan(big[:pad]).all() and torch.isnan(big[pad + n:]).all()


def test_replicated_layout_matches_full_transform(K):
    # sharded.transform_two_body_replicated on one GPU, every rank's slab (uses the skinny product)
    from quantum_systems_amd import sharded

    l = 64
    g = torch.Generator(device="cuda:0").manual_seed(21)
    u = torch.rand(l, l, l, l, dtype=torch.float64, device="cuda:0", generator=g)
    C = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g) / 8
    Ct = torch.randn(l, l, dtype=torch.float64, device="cuda:0", generator=g) / 8
    full = K.transform_two_body(u, C, Ct)
    for world in (2, 4):
        for rank in range(world):
            lo, hi = sharded.SlabPartition(l, world).bounds(rank)
            slab = sharded.transform_two_body_replicated(u, C, Ct, rank, world)
            assert (slab - full[lo:hi]).abs().max().item() <= 1e-12 * full.abs().max().item()


def test_dynamic_lds_limit_is_raised_when_a_later_call_needs_more(K):
    # kernels whose LDS size depends on run-time extents (spin2_tb: 96 n bytes; gemm_skinny:
    # 8 NP m (k + 2)) used to opt in to the FIRST call's size only, so a later, larger call failed with QS_ERR_HIP.
    g = torch.Generator(device="cuda:0").manual_seed(77)
    for n in (700, 1100):                                   # 67 KB, then 106 KB in the same process
        S = torch.complex(torch.randn(3, n, n, dtype=torch.float64, device="cuda:0", generator=g),
                          torch.randn(3, n, n, dtype=torch.float64, device="cuda:0", generator=g))
        rows = K.spin_squared_two_body(S, antisymmetrize=True, p_lo=n - 1, p_hi=n)
        assert K.last_dispatch() == "qs::spin2_tb_kernel"
        q = n // 3
        ref = (sum(torch.outer(S[k, n - 1], S[k, q]) for k in range(3))
               - sum(torch.outer(S[k, q], S[k, n - 1]) for k in range(3)))
        assert (rows[0, q] - ref).abs().max().item() <= 1e-13 * 3 * S.abs().max().item() ** 2
        del rows, S
    rng = np.random.default_rng(5)
    m, n = 16, 1 << 16
    for k in (260, 512):                                    # complex, 16 rows: 67 KB, then 132 KB
        A = rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k))
        B = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
        got = K.matmul(dev(A), dev(B))
        assert "gemm_skinny_kernel<true, 1>" in K.last_dispatch()
        assert relerr(host(got), A @ B) <= 1e-13


def test_spin_squared_two_body_with_1024_spin_orbitals(K):
    # n = 1024 (l = 512 spatial orbitals, BASELINE.json configs[4]): the rows of S_x, S_y, S_z staged per workgroup
    # need 96 KB of LDS -- the kernel opts in beyond 64 KB (per device) instead of refusing n > 682
    n = 1024
    g = torch.Generator(device="cuda:0").manual_seed(1024)
    S = torch.complex(torch.randn(3, n, n, dtype=torch.float64, device="cuda:0", generator=g),
                      torch.randn(3, n, n, dtype=torch.float64, device="cuda:0", generator=g))
    for anti in (False, True):
        rows = K.spin_squared_two_body(S, antisymmetrize=anti, p_lo=700, p_hi=701)       # one row: 17 GB
        assert K.last_dispatch() == "qs::spin2_tb_kernel" and tuple(rows.shape) == (1, n, n, n)
        scale = 3 * S.abs().max().item() ** 2
        for q in (0, 513, 700, 1023):
            ref = sum(torch.outer(S[k, 700], S[k, q]) for k in range(3))                  # [r, s] = S[p,r] S[q,s]
            if anti:
                ref = ref - sum(torch.outer(S[k, q], S[k, 700]) for k in range(3))        # - S[q,r] S[p,s]
            assert (rows[0, q] - ref).abs().max().item() <= 1e-13 * scale
        del rows
