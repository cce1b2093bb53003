"""Small index helpers shared by the system classes
(reference: quantum_systems/system_helper.py)."""


def delta(p, q):
    """Kronecker delta (system_helper.py:4-6)."""
    return p == q


def spin_delta(p, q):
    """1 when spin orbitals p and q carry the same spin, i.e. equal parity of
    the index P = 2p + sigma (system_helper.py:9-11)."""
    return ((p & 0x1) ^ (q & 0x1)) ^ 0x1


def compute_particle_density(rho_qp, ket_spf, bra_spf, np):
    """rho(r) = sum_pq bra_p(r) rho_qp[q, p] ket_q(r) on the grid
    (system_helper.py:14-27), as one contraction over q followed by a
    pointwise product summed over p."""
    assert bra_spf.shape == ket_spf.shape
    assert bra_spf.dtype == ket_spf.dtype
    l = ket_spf.shape[0]
    grid = tuple(ket_spf.shape[1:])
    ket = ket_spf.reshape(l, -1)
    bra = bra_spf.reshape(l, -1)
    # tmp[p, g] = sum_q rho_qp[q, p] ket[q, g]
    tmp = np.tensordot(rho_qp, ket, axes=((0,), (0,)))
    rho = (bra * tmp).sum(0)
    return rho.reshape(grid)
