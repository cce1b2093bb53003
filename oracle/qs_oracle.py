"""CPU oracle for the integral basis-transformation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``quantum-systems_amd/`` may import
this file; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the
timed CPU baseline -- never as the thing shipped.

This is a fresh NumPy restatement of the algorithms the reference
(HyQD/quantum-systems v0.2.6, pure NumPy) uses on the path.  Every function
cites the reference ``file:line`` it follows (paths relative to the reference
checkout).  The reference has no native code, so there is no ``oracle/_ref``
build: the reference *is* NumPy, and this restatement issues the same NumPy
calls in the same order (same contraction order d, c, b, a; same
``kron``-by-zero construction, hence the same signed zeros).

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference
itself in the build container (``tests/golden/make_golden.py``); the CPU test
``tests/test_oracle_golden.py`` checks this file against every one of them, and
against restatements of the reference's own analytic tests
(``tests/test_helper.py``, ``tests/test_custom_system.py``).

State is kept in plain dicts rather than classes on purpose: the oracle is a
checker of arithmetic, not a second implementation of the API.
"""

import numpy as np

# --------------------------------------------------------------------------
# a1 / a2 / a3  -- transforms
# --------------------------------------------------------------------------


def _bra(C, C_tilde):
    # quantum_systems/basis_set.py:331-332, 338-339: default bra coefficients
    # are the Hermitian adjoint of C.
    return C.conj().T if C_tilde is None else C_tilde


def transform_one_body(h, C, C_tilde=None):
    """``Ct @ (h @ C)`` -- quantum_systems/basis_set.py:329-334."""
    Ct = _bra(C, C_tilde)
    return np.dot(Ct, np.dot(h, C))


def transform_two_body(u, C, C_tilde=None):
    """Four single-index contractions in the order d, c, b, a.

    quantum_systems/basis_set.py:336-350.  The intermediate axis orders (and
    therefore the BLAS calls NumPy issues) are those of the reference:
    tensordot over the last axis, then axis 2 (result axis moved back with a
    (0,1,3,2) view), then axis 1 against axis 1 of the bra matrix (view
    (0,3,1,2)), then the leading axis.
    """
    Ct = _bra(C, C_tilde)
    t = transform_two_body_dcb(u, C, Ct)
    return np.tensordot(Ct, t, axes=(1, 0))  # pa,aqrs->pqrs    :348


def transform_two_body_dcb(u, C, C_tilde=None):
    """The first three contractions (d, c, b) of basis_set.py:342-346 on any
    block of leading-index rows: v[a,q,r,s] = Ct[q,b] u[a,b,c,d] C[c,r] C[d,s].
    Rows of the leading index are independent here, which is what the sharded
    layouts of the build rely on."""
    Ct = _bra(C, C_tilde)
    t = np.tensordot(u, C, axes=(3, 0))  # abcd,ds->abcs        :342
    t = np.tensordot(t, C, axes=(2, 0))  # abcs,cr->absr        :344
    t = t.transpose(0, 1, 3, 2)  # ->abrs
    t = np.tensordot(t, Ct, axes=(1, 1))  # abrs,qb->arsq        :346
    return t.transpose(0, 3, 1, 2)  # ->aqrs


def transform_two_body_pq_samples(u, C, C_tilde, pairs):
    """``out[p, q, :, :]`` of basis_set.py:336-350 for a few index pairs (p, q) only: O(l^4) per pair
    instead of the O(l^5) of the whole transform, so the headline size (l = 256, 34 GB) can be checked
    element-wise inside the default GPU suite from one download of ``u``.

    The same four sums of basis_set.py:341-348, with the two bra contractions taken first because
    their free indices are fixed:  ``W[c,d] = sum_ab Ct[p,a] Ct[q,b] u[a,b,c,d]`` (one GEMM for all the
    pairs at once), then ``out[p,q] = C^T W C`` (:342-344).  Returns an array (len(pairs), M, M)."""
    Ct = _bra(C, C_tilde)
    L = u.shape[0]
    weights = np.stack([np.multiply.outer(Ct[p], Ct[q]).reshape(L * L) for (p, q) in pairs])   # (k, (a,b))
    W = np.dot(weights, u.reshape(L * L, L * L)).reshape(len(pairs), L, L)                     # (k, c, d)
    return np.stack([np.dot(C.T, np.dot(w, C)) for w in W])


def transform_two_body_einsum(u, C, C_tilde=None):
    """Five-operand einsum form the reference's tests compare against
    (tests/test_helper.py:43-51, tests/test_custom_system.py:19-24)."""
    Ct = _bra(C, C_tilde)
    return np.einsum("pa,qb,abcd,cr,ds->pqrs", Ct, Ct, u, C, C, optimize=True)


def transform_spf(spf, C):
    """quantum_systems/basis_set.py:321-323."""
    return np.tensordot(C, spf, axes=((0), (0)))


def transform_bra_spf(bra_spf, C_tilde):
    """quantum_systems/basis_set.py:325-327."""
    return np.tensordot(C_tilde, bra_spf, axes=((1), (0)))


# --------------------------------------------------------------------------
# a5 / a6 / a7  -- anti-symmetrisation and spin doubling
# --------------------------------------------------------------------------


def anti_symmetrize_u(u):
    """``u[pqrs] - u[pqsr]`` -- quantum_systems/basis_set.py:776-778."""
    return u - u.transpose(0, 1, 3, 2)


def add_spin_one_body(h):
    """``kron(h, I2)`` -- quantum_systems/basis_set.py:768-770."""
    return np.kron(h, np.eye(2))


def add_spin_two_body(u):
    """``kron(u, d_pr d_qs)`` -- quantum_systems/basis_set.py:772-774.

    The multiplication by the 0.0 entries of the delta tensor is kept, so
    negative inputs leave ``-0.0`` in the structurally empty blocks exactly as
    the reference does.
    """
    eye = np.eye(2)
    return np.kron(u, np.einsum("pr, qs -> pqrs", eye, eye))


def add_spin_spf(spf):
    """Row interleave -- quantum_systems/basis_set.py:751-759."""
    out = np.zeros((2 * spf.shape[0],) + tuple(spf.shape[1:]), dtype=spf.dtype)
    out[0::2] = spf
    out[1::2] = spf
    return out


def spin_delta(p, q):
    """1 when p and q have equal spin parity -- system_helper.py:9-11."""
    return ((p & 1) ^ (q & 1)) ^ 1


def spin_two_body_index_law(u):
    """Explicit-loop statement of add_spin_two_body o anti_symmetrize_u as the
    reference's tests spell it (tests/test_helper.py:108-135).  Small l only."""
    l = 2 * u.shape[0]
    out = np.zeros((l, l, l, l), dtype=u.dtype)
    for p in range(l):
        for q in range(l):
            for r in range(l):
                for s in range(l):
                    v = (
                        spin_delta(p, r)
                        * spin_delta(q, s)
                        * u[p // 2, q // 2, r // 2, s // 2]
                    )
                    v = v - (
                        spin_delta(p, s)
                        * spin_delta(q, r)
                        * u[p // 2, q // 2, s // 2, r // 2]
                    )
                    out[p, q, r, s] = v
    return out


# --------------------------------------------------------------------------
# a8 helpers -- Pauli matrices and the spin-squared operator
# --------------------------------------------------------------------------

_PAULI = (
    np.array([[0, 1], [1, 0]], dtype=np.complex128),
    np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    np.array([[1, 0], [0, -1]], dtype=np.complex128),
)


def setup_pauli_matrices(a, b):
    """Pauli matrices in the spinor basis {a, b} (column vectors).

    quantum_systems/basis_set.py:638-697: element [i, j] = <s_i| sigma |s_j>
    with s_0 = a, s_1 = b.
    """
    basis = (a, b)
    out = []
    for sig in _PAULI:
        m = np.zeros((2, 2), dtype=np.complex128)
        for i, bra in enumerate(basis):
            for j, ket in enumerate(basis):
                m[i, j] = np.dot(bra.conj().T, np.dot(sig, ket))[0, 0]
        out.append(m)
    return tuple(out)


def setup_spin_squared_operator(spin_x, spin_y, spin_z, overlap):
    """One- and two-body parts of S^2 -- quantum_systems/basis_set.py:699-749."""
    l = len(spin_x)
    spin_2 = np.zeros_like(spin_x)
    spin_2_tb = np.zeros((l, l, l, l), dtype=spin_2.dtype)
    for s_i in (spin_x, spin_y, spin_z):
        spin_2 += s_i @ overlap @ s_i
        spin_2_tb += np.einsum("pr, qs -> pqrs", s_i, s_i)
    return spin_2, spin_2_tb


# --------------------------------------------------------------------------
# a12 -- RandomBasisSet input generator
# --------------------------------------------------------------------------


def _rand_complex(shape):
    # quantum_systems/random_basis.py:52-69: two draws from the legacy global
    # stream, real part first.
    return np.random.random(shape) + 1j * np.random.random(shape)


def random_basis(l, dim):
    """Synthetic matrix elements with the symmetries of second-quantised
    integrals, drawn from NumPy's *global* legacy RNG in the reference's order
    h, s, u, position, nuclear repulsion energy, charge
    (quantum_systems/random_basis.py:21-50)."""
    st = new_state(l, dim)
    h = _rand_complex((l, l))
    st["h"] = 0.5 * (h + h.conj().T)
    s = _rand_complex((l, l))
    st["s"] = 0.5 * (s + s.conj().T)
    u = _rand_complex((l, l, l, l))
    st["u"] = 0.5 * (u + u.transpose(1, 0, 3, 2))
    pos = _rand_complex((dim, l, l))
    for i in range(dim):
        pos[i] = 0.5 * (pos[i] + pos[i].conj().T)
    st["position"] = pos
    st["nuclear_repulsion_energy"] = np.random.random()
    st["charge"] = np.random.choice([-1, 1])
    return st


# --------------------------------------------------------------------------
# a4 / a8 -- orchestration on a plain dict
# --------------------------------------------------------------------------

_FIELDS = (
    "h", "s", "u", "position", "momentum", "spf", "bra_spf",
    "spin_x", "spin_y", "spin_z", "spin_2", "spin_2_tb",
)


def new_state(l, dim, includes_spin=False, anti_symmetrized_u=False):
    """Field inventory of a basis set -- quantum_systems/basis_set.py:32-69."""
    st = {k: None for k in _FIELDS}
    st.update(
        l=l, dim=dim, includes_spin=includes_spin,
        anti_symmetrized_u=anti_symmetrized_u,
        nuclear_repulsion_energy=0, particle_charge=-1,
    )
    return st


def change_basis(st, C, C_tilde=None):
    """In-place basis change of every stored operator.

    quantum_systems/basis_set.py:413-464 with the helpers at :358-411.  Note
    the reference computes the transformed spin one-body operators and drops
    them (:368-372), so spin_x/y/z/spin_2 are left untouched here too, while
    spin_2_tb *is* transformed (:379-382).
    """
    st["l"] = C.shape[1]
    Ct = _bra(C, C_tilde)
    st["h"] = transform_one_body(st["h"], C, Ct)
    if st["s"] is not None:
        st["s"] = transform_one_body(st["s"], C, Ct)
    st["u"] = transform_two_body(st["u"], C, Ct)
    if st["spin_2_tb"] is not None:
        st["spin_2_tb"] = transform_two_body(st["spin_2_tb"], C, Ct)
    for name in ("position", "momentum"):
        if st[name] is not None:
            st[name] = np.asarray(
                [transform_one_body(m, C, Ct) for m in st[name]]
            )
    if st["spf"] is not None:
        bra = st["bra_spf"] if st["bra_spf"] is not None else st["spf"].conj()
        st["bra_spf"] = transform_bra_spf(bra, Ct)
        st["spf"] = transform_spf(st["spf"], C)
    return st


def anti_symmetrize_two_body_elements(st):
    """quantum_systems/basis_set.py:511-528 (guarded by the flag)."""
    if not st["anti_symmetrized_u"]:
        st["u"] = anti_symmetrize_u(st["u"])
        if st["spin_2_tb"] is not None:
            st["spin_2_tb"] = anti_symmetrize_u(st["spin_2_tb"])
        st["anti_symmetrized_u"] = True
    return st


def change_to_general_orbital_basis(st, a=(1, 0), b=(0, 1), anti_symmetrize=True):
    """Spin doubling of a spatial-orbital basis, in place.

    quantum_systems/basis_set.py:530-636: l doubles, one-body operators get
    kron(., I2), u gets the delta-kron, spin operators are built from the
    spatial overlap, everything is anti-symmetrised (optionally) and finally
    cast to complex128 (:634).
    """
    assert not st["includes_spin"]
    st["includes_spin"] = True
    st["l"] = 2 * st["l"]
    overlap = st["s"].copy()
    st["h"] = add_spin_one_body(st["h"])
    st["s"] = add_spin_one_body(st["s"])
    st["u"] = add_spin_two_body(st["u"])

    av = np.array(a).astype(np.complex128).reshape(-1, 1)
    bv = np.array(b).astype(np.complex128).reshape(-1, 1)
    assert abs(np.dot(av.conj().T, av) - 1) < 1e-12
    assert abs(np.dot(bv.conj().T, bv) - 1) < 1e-12
    assert abs(np.dot(av.conj().T, bv)) < 1e-12
    sx, sy, sz = setup_pauli_matrices(av, bv)
    st["sigma_x"], st["sigma_y"], st["sigma_z"] = sx, sy, sz
    st["spin_x"] = 0.5 * np.kron(overlap, sx)
    st["spin_y"] = 0.5 * np.kron(overlap, sy)
    st["spin_z"] = 0.5 * np.kron(overlap, sz)
    st["spin_2"], st["spin_2_tb"] = setup_spin_squared_operator(
        st["spin_x"], st["spin_y"], st["spin_z"], st["s"]
    )
    if anti_symmetrize:
        anti_symmetrize_two_body_elements(st)
    for name in ("position", "momentum"):
        if st[name] is not None:
            st[name] = np.array([add_spin_one_body(m) for m in st[name]])
    if st["spf"] is not None:
        st["spf"] = add_spin_spf(st["spf"])
        if st["bra_spf"] is not None:
            st["bra_spf"] = add_spin_spf(st["bra_spf"])
    for name in _FIELDS:
        if st[name] is not None:
            st[name] = st[name].astype(np.complex128)
    return st


# --------------------------------------------------------------------------
# work figure shared with bench.py
# --------------------------------------------------------------------------


# --------------------------------------------------------------------------
# f2  -- first consumers of (transformed) h, u: reference energy, Fock matrix
# --------------------------------------------------------------------------


def reference_energy(h, u, n_occ, nuclear_repulsion_energy=0.0, spin_orbitals=False):
    """Energy of the reference determinant with ``n_occ`` occupied ORBITALS of the basis.

    closed shell over spatial orbitals (quantum_systems/spatial_orbital_system.py:140-150):
        2 h_ii + 2 u_ijij - u_ijji + E_nuc
    spin orbitals with anti-symmetrised u (quantum_systems/general_orbital_system.py:108-121):
        h_ii + 1/2 u_ijij + E_nuc
    written with the same nested traces as the reference."""
    o = slice(0, n_occ)
    if spin_orbitals:
        return (
            np.trace(h[o, o])
            + 0.5 * np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=3))
            + nuclear_repulsion_energy
        )
    return (
        2 * np.trace(h[o, o])
        + 2 * np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=3))
        - np.trace(np.trace(u[o, o, o, o], axis1=1, axis2=2))
        + nuclear_repulsion_energy
    )


def fock_matrix(h, u, n_occ, spin_orbitals=False):
    """closed shell (spatial_orbital_system.py:176-190): f_pq = h_pq + 2 u_piqi - u_piiq;
    spin orbitals (general_orbital_system.py:147-159): f_pq = h_pq + u_piqi."""
    o = slice(0, n_occ)
    f = np.zeros_like(h)
    f += h
    if spin_orbitals:
        f += np.einsum("piqi -> pq", u[:, o, :, o])
        return f
    f += 2 * np.einsum("piqi -> pq", u[:, o, :, o])
    f -= np.einsum("piiq -> pq", u[:, o, o, :])
    return f


def transform_flops(L, M, complex_=False):
    """2(L^4 M + L^3 M^2 + L^2 M^3 + L M^4) real flops, x4 for complex128."""
    k = 4 if complex_ else 1
    return k * 2 * (L**4 * M + L**3 * M**2 + L**2 * M**3 + L * M**4)


# --------------------------------------------------------------------------
# grid / DVR contractions of the same family
# --------------------------------------------------------------------------


def two_body_from_grid(K, C, C_tilde=None, antisymmetrize=False):
    """out[p,q,r,s] = sum_ab Ct[p,a] Ct[q,b] K[a,b] C[a,r] C[b,s]
    (sinc_dvr/one_dim/sinc_dvr.py:227-256: transform of a 2-d ``u``; the fused
    anti-symmetrisation subtracts the r <-> s exchanged contraction).  With
    ``C_tilde = C.T`` this is the quadrature of one_dim_qd.py:275-280."""
    if C_tilde is None:
        C_tilde = C.conj().T
    out = np.einsum("bs,ar,qb,pa,ab->pqrs", C, C, C_tilde, C_tilde, K, optimize=True)
    if antisymmetrize:
        out = out - np.einsum("br,as,qb,pa,ab->pqrs", C, C, C_tilde, C_tilde, K, optimize=True)
    return out


def odqd_setup(l, grid_length, num_grid_points, potential, a=0.25, alpha=1.0, beta=0.0):
    """Arrays of the 1-D quantum dot basis (one_dim_qd.py:232-289): the lowest
    ``l`` eigenpairs of the three-point finite-difference Hamiltonian on the
    interior grid points, the grid functions, and the quadratures for ``u`` and
    ``position``.  ``potential`` is a callable V(x).  Returns a dict."""
    import scipy.linalg

    grid = np.linspace(-grid_length, grid_length, num_grid_points)
    x = grid[1:-1]
    dx = grid[1] - grid[0]
    eps, C = scipy.linalg.eigh_tridiagonal(
        1.0 / dx**2 + potential(x), -np.ones(num_grid_points - 3) / (2 * dx**2),
        select="i", select_range=(0, l - 1),
    )
    spf = np.zeros((l, num_grid_points), dtype=np.complex128)
    spf[:, 1:-1] = C.T / np.sqrt(dx)
    K = alpha / np.sqrt((x[None, :] - x[:, None]) ** 2 + a**2)
    u = np.einsum("pa,qb,pc,qd,pq->abcd", C, C, C, C, K, optimize=True)
    position = np.zeros((1, l, l), dtype=np.complex128)
    position[0] = np.einsum("pa,p,pb->ab", C, x + beta * x**2, C, optimize=True)
    return {"grid": grid, "eigen_energies": eps, "C": C, "spf": spf,
            "h": np.diag(eps).astype(np.complex128), "s": np.eye(l), "u": u, "position": position}
