"""``BasisSet`` operations when the array module is a ``ShardedDeviceModule``: the same orchestration as
basis_set.py (reference: quantum_systems/basis_set.py:358-464, :511-636), with the rank-4 tensors held as
``ShardedTensor4`` slabs.  One-body work (O(l^2) data) is replicated -- every rank computes it on its own
GPU, no communication; the rank-4 work is slab-local except for ONE all-to-all inside each four-index
transform.  All arithmetic goes through ``np.engine`` (``sharded.HipEngine`` = libqs_amd.so).

``BasisSet`` dispatches here at the top of ``change_basis``, ``change_to_general_orbital_basis``,
``anti_symmetrize_two_body_elements``, the ``spin_2_tb`` property and ``change_module``."""

import warnings

import numpy
import torch

from . import sharded
from .array_module import to_host, wrap
from .sharded_module import ShardedTensor4, is_sharded

_C128 = torch.complex128


def _bra(C):
    return C.conj().transpose(0, 1).resolve_conj().contiguous()


def _plain(t):
    """torch.Tensor view of a (device) array for the engine."""
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) else t


def transform_two_body(t, C, Ct, np):
    """Four-index transform of a sharded tensor: rows of one leading index in, rows of the other one out (the
    sharded index flips), one all-to-all per streamed chunk of rows; per rank never more than the input rows, the
    result rows and O(l^3) of scratch (``sharded.transform_two_body_rows``; on RCCL the whole pipeline is ONE C-ABI
    call with the exchange overlapped on a second stream, ``ShardedDeviceModule.rccl``)."""
    M = C.shape[1]
    comm = np.rccl() if hasattr(np, "rccl") else None
    if comm is not None:
        rows = comm.transform_two_body_rows(t.rows, C, Ct, in_part=t.part)
    else:
        rows = sharded.transform_two_body_rows(t.rows, C, Ct, t.rank, t.world, t.group, engine=np.engine,
                                               in_part=t.part)
    return ShardedTensor4(None, M, 1 - t.axis, t.rank, t.world, t.group, rows=rows)   # balanced split of M


def change_basis(bs, C, C_tilde=None):
    """basis_set.py:413-464 on a sharded basis set."""
    np = bs.np
    eng = np.engine
    bs.l = C.shape[1]                                           # :448
    d_C = _plain(np.asarray(C)).contiguous()
    d_Ct = _bra(d_C) if C_tilde is None else _plain(np.asarray(C_tilde)).contiguous()

    def one_body(arr):
        return wrap(eng.transform_one_body(_plain(np.asarray(arr)), d_C, d_Ct))

    bs.h = one_body(bs.h)
    if bs.s is not None:
        bs.s = one_body(bs.s)
    # :368-372: spin_x/y/z/spin_2 are transformed into a loop local and dropped upstream -- untouched here
    u = bs._u if is_sharded(bs._u) else np.shard(bs._u)
    bs.u = transform_two_body(u, d_C, d_Ct, np)
    del u
    if bs._spin_2_tb_recipe_valid():
        # :379-382 on the recipe: spin_2_tb = sum_i S_i (x) S_i (- r <-> s), so its transform is the same expression in
        # S'_i = C~ S_i C -- three replicated one-body transforms, no second O(l^5) transform, no second slab
        stack, anti = bs._spin_2_tb_recipe
        S = _plain(np.asarray(stack)).to(_C128).contiguous()
        bs._spin_2_tb = None
        bs._spin_2_tb_recipe = (wrap(eng.transform_one_body(S, d_C.to(_C128), d_Ct.to(_C128))), anti)
    elif bs.spin_2_tb is not None:                              # :379-382
        bs.spin_2_tb = transform_two_body(bs._spin_2_tb, d_C, d_Ct, np)
    if bs.position is not None:
        bs.position = one_body(bs.position)
    if bs.momentum is not None:
        bs.momentum = one_body(bs.momentum)
    if bs.spf is not None:
        L = d_C.shape[0]
        bra_in, ket_in = _plain(bs.bra_spf), _plain(bs.spf)
        bra = eng.matmul(d_Ct, bra_in.reshape(L, -1).contiguous())
        ket = eng.matmul(d_C.transpose(0, 1).contiguous(), ket_in.reshape(L, -1).contiguous())
        bs.bra_spf = wrap(bra.reshape((d_Ct.shape[0],) + tuple(bra_in.shape[1:])))
        bs.spf = wrap(ket.reshape((d_C.shape[1],) + tuple(ket_in.shape[1:])))


def anti_symmetrize_two_body_elements(bs):
    """basis_set.py:511-528: slab-local, whichever index is sharded (the exchange is r <-> s)."""
    if bs._anti_symmetrized_u:
        return
    eng = bs.np.engine
    u = bs._u
    bs.u = u._like(eng.antisymmetrize(u.rows))
    if bs._spin_2_tb_recipe_valid():
        stack, _ = bs._spin_2_tb_recipe
        bs._spin_2_tb_recipe = (stack, True)
        bs._spin_2_tb = None                                     # rebuilt anti-symmetrised on the next access
    elif bs._spin_2_tb is not None:
        t = bs._spin_2_tb
        bs.spin_2_tb = t._like(eng.antisymmetrize(t.rows))
    bs._anti_symmetrized_u = True


def spin_2_tb_rows(bs):
    """This rank's rows of the two-body S^2 (basis_set.py:745-747), built from the (3, n, n) spin matrices."""
    np = bs.np
    stack, anti = bs._spin_2_tb_recipe            # (kept: change_basis transforms the recipe, not the rows)
    S = _plain(np.asarray(stack)).to(_C128).contiguous()
    n = S.shape[-1]
    lo, hi = sharded.SlabPartition(n, np.world).bounds(np.rank)
    if hi > lo:
        rows = np.engine.spin_squared_two_body(S, antisymmetrize=anti, p_lo=lo, p_hi=hi)
    else:
        rows = torch.empty((0, n, n, n), dtype=_C128, device=S.device)
    return ShardedTensor4(None, n, 0, np.rank, np.world, np.group, rows=rows)


def change_to_general_orbital_basis(bs, a=[1, 0], b=[0, 1], anti_symmetrize=True):
    """basis_set.py:530-636 on a sharded basis set: the rank's block of the spatial ``u`` is expanded into
    its block of the spin tensor (fused with the anti-symmetrisation and the complex cast, one read and one
    write), everything else is replicated O(l^2) work."""
    if bs._includes_spin:
        warnings.warn("The basis has already been spin-doubled. Avoiding a second doubling.")
        return None
    np = bs.np
    eng = np.engine
    bs._includes_spin = True
    bs.l = 2 * bs.l

    d_overlap = _plain(np.asarray(bs.s))
    d_h = eng.add_spin_one_body(_plain(np.asarray(bs.h)), out_dtype=_C128)
    d_s = eng.add_spin_one_body(d_overlap, out_dtype=_C128)

    anti_now = bool(anti_symmetrize) and not bs._anti_symmetrized_u
    old = bs._u if is_sharded(bs._u) else np.shard(bs._u)
    bs._u = None
    if old.axis == 1:
        # the doubling ties the FIRST index to the third and the second to the fourth: expand leading-index rows
        # (one all-to-all of the spatial tensor, 1/16 of what is built from it)
        old = old.reshard(0)
    if old.rows.numel():
        block = eng.spin_expand_block(old.rows, antisymmetrize=anti_now, out_dtype=_C128)
    else:
        shape = tuple(2 * x for x in old.rows.shape)
        block = torch.empty(shape, dtype=_C128, device=old.rows.device)
    # spin rows 2p, 2p+1 stay with the rank that holds spatial row p: the offsets double
    new_u = ShardedTensor4(None, bs.l, 0, old.rank, old.world, old.group, old.part.doubled(), rows=block)
    del old, block

    bs.h = wrap(d_h)
    bs.s = wrap(d_s)
    bs.u = new_u

    av = numpy.asarray(to_host(a)).astype(numpy.complex128).reshape(-1, 1)
    bv = numpy.asarray(to_host(b)).astype(numpy.complex128).reshape(-1, 1)
    assert abs(numpy.dot(av.conj().T, av) - 1) < 1e-12
    assert abs(numpy.dot(bv.conj().T, bv) - 1) < 1e-12
    assert abs(numpy.dot(av.conj().T, bv)) < 1e-12
    bs.a, bs.b = np.asarray(av), np.asarray(bv)
    sig = bs.setup_pauli_matrices(av, bv, numpy)
    bs.sigma_x, bs.sigma_y, bs.sigma_z = (np.asarray(m) for m in sig)
    d_ov = d_overlap.to(_C128)
    stack = torch.stack([0.5 * torch.kron(d_ov, torch.from_numpy(m).to(d_ov.device)) for m in sig])
    bs.spin_x, bs.spin_y, bs.spin_z = (wrap(stack[k]) for k in range(3))
    spin_2 = None
    for k in range(3):
        term = eng.matmul(stack[k].contiguous(), eng.matmul(d_s, stack[k].contiguous()))
        spin_2 = term if spin_2 is None else spin_2 + term
    bs.spin_2 = wrap(spin_2)
    bs._spin_2_tb = None
    bs._spin_2_tb_recipe = (wrap(stack), anti_now)           # rows built on first access (spin_2_tb_rows)

    if anti_symmetrize:
        bs._anti_symmetrized_u = True
    if bs.position is not None:
        bs.position = wrap(eng.add_spin_one_body(_plain(np.asarray(bs.position)), out_dtype=_C128))
    if bs.momentum is not None:
        bs.momentum = wrap(eng.add_spin_one_body(_plain(np.asarray(bs.momentum)), out_dtype=_C128))
    if bs.spf is not None:
        had_bra = bs._bra_spf is not None
        old_bra = bs._bra_spf
        bs.spf = wrap(torch.repeat_interleave(_plain(bs.spf), 2, dim=0))
        if had_bra:
            bs.bra_spf = wrap(torch.repeat_interleave(_plain(old_bra), 2, dim=0))
    bs.cast_to_complex()
    return bs


# ------------------------------------------------------------------------------------------------
# first consumers: Fock matrix and reference energy from a sharded u
# ------------------------------------------------------------------------------------------------


def construct_fock_matrix(h, u, n_occ, spin_orbitals, f=None):
    """Full Fock matrix on every rank from a sharded ``u`` (spatial_orbital_system.py:152-190,
    general_orbital_system.py:123-159): slab-local sums, then l*l numbers over the node (rows gathered
    when the leading index is sharded, partial sums added when the second index is)."""
    hp = _plain(h)
    if u.axis == 0:
        out = sharded.construct_fock_matrix_sharded(hp, u.local, n_occ, u.rank, u.world, spin_orbitals, u.group,
                                                    part=u.part)
    else:
        part = sharded.fock_partial_second_index(u.local, n_occ, u.lo, spin_orbitals)
        out = hp.to(part.dtype) + sharded.all_reduce_sum(part, u.world, u.group)
    if f is not None:
        f.fill(0)
        f += wrap(out)
        return f
    return wrap(out)


def compute_reference_energy(h, u, n_occ, spin_orbitals, nuclear_repulsion_energy):
    """Reference-determinant energy from a sharded ``u`` (spatial_orbital_system.py:106-150,
    general_orbital_system.py:75-121): one number per rank, one all-reduce."""
    hp = _plain(h)
    if u.axis == 0:
        part = sharded.reference_energy_partial(hp, u.local, n_occ, u.lo, spin_orbitals)
    else:
        part = sharded.reference_energy_partial_second_index(hp, u.local, n_occ, u.lo, spin_orbitals)
    total = sharded.all_reduce_sum(part.reshape(1), u.world, u.group)[0]
    return wrap(total + nuclear_repulsion_energy)
