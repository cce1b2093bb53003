"""Worker for tests/test_sharded_gloo.py: run under torch.distributed.run with
the gloo backend on CPU.  Exercises the partitioning and the exchange of
quantum_systems_amd.sharded with an oracle-backed engine standing in for the
HIP kernels (the product engine needs a GPU)."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qs_oracle as orc  # noqa: E402
from quantum_systems_amd import sharded  # noqa: E402


class OracleEngine:
    """CPU test double with the interface of sharded.HipEngine."""

    name = "oracle"

    @staticmethod
    def matmul(A, B, out=None, accumulate=False):
        res = torch.from_numpy(np.matmul(A.resolve_conj().numpy(), B.resolve_conj().numpy()))
        if out is None:
            return res
        view = out.reshape(res.shape)
        if accumulate:
            view += res
        else:
            view.copy_(res)
        return out

    @staticmethod
    def gemm_strided(dt, A, B, out, m, n, k, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0,
                     accumulate=False, a_off=0, b_off=0, c_off=0):
        # strided views over the flat storage, exactly the addressing qs_matmul uses
        def view(t, off, rows, cols, ld, stride):
            flat = t.reshape(-1)
            return torch.as_strided(flat, (batch, rows, cols), (stride, ld, 1), storage_offset=flat.storage_offset() + off)

        a = view(A, a_off, m, k, lda, sa).numpy()
        b = view(B, b_off, k, n, ldb, sb).numpy()
        res = torch.from_numpy(np.matmul(a, b))
        c = view(out, c_off, m, n, ldc, sc)
        if accumulate:
            c += res
        else:
            c.copy_(res)
        return out

    @staticmethod
    def partial(u_slab, C, Ct):
        v = orc.transform_two_body_dcb(
            u_slab.numpy(), C.resolve_conj().numpy(), Ct.resolve_conj().numpy()
        )
        return torch.from_numpy(np.ascontiguousarray(v))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(42)  # same stream on every rank
    for (L, M, cplx) in [(6, 6, True), (7, 5, False), (5, 9, True), (3, 3, False), (12, 12, False)]:
        if cplx:
            u = rng.random((L,) * 4) + 1j * rng.random((L,) * 4)
            C = rng.random((L, M)) + 1j * rng.random((L, M))
            Ct = rng.random((M, L)) + 1j * rng.random((M, L))
        else:
            u = rng.standard_normal((L,) * 4)
            C = rng.standard_normal((L, M))
            Ct = rng.standard_normal((M, L))
        ref = orc.transform_two_body(u, C, Ct)
        tu, tC, tCt = torch.from_numpy(u), torch.from_numpy(C), torch.from_numpy(Ct)
        p_lo, p_hi = sharded.SlabPartition(M, world).bounds(rank)

        # layout 1: replicated u, no collective on the data path
        slab = sharded.transform_two_body_replicated(tu, tC, tCt, rank, world, engine=OracleEngine)
        np.testing.assert_allclose(slab.numpy(), ref[p_lo:p_hi], rtol=1e-12, atol=1e-12)
        full = sharded.all_gather_slabs(slab, M, rank, world)
        np.testing.assert_allclose(full.numpy(), ref, rtol=1e-12, atol=1e-12)

        # layout 2: u sharded over its second index, one all-to-all
        b_lo, b_hi = sharded.SlabPartition(L, world).bounds(rank)
        ub = tu[:, b_lo:b_hi].contiguous()
        slab2 = sharded.transform_two_body_sharded(ub, tC, tCt, rank, world, engine=OracleEngine)
        np.testing.assert_allclose(slab2.numpy(), ref[p_lo:p_hi], rtol=1e-12, atol=1e-12)

        # layout 3: in-place, memory-lean form (square transforms, l divisible by world)
        if L == M and L % world == 0:
            for rows in (1, 2):
                ub2 = tu[:, b_lo:b_hi].contiguous().clone()
                slab4 = sharded.transform_two_body_sharded_inplace(
                    ub2, tC, tCt, rank, world, engine=OracleEngine, staging_rows=rows)
                np.testing.assert_allclose(slab4.numpy(), ref[p_lo:p_hi], rtol=1e-12, atol=1e-12)
                assert torch.equal(ub2, tu[:, b_lo:b_hi])           # the resident slab is untouched
            # the buffer reused across steps, as a time loop would
            keep = torch.empty((L // world + 1, L, L, L), dtype=slab4.dtype)
            for _ in range(2):
                slab5 = sharded.transform_two_body_sharded_inplace(
                    ub2, tC, tCt, rank, world, engine=OracleEngine, staging_rows=3, out=keep)
                np.testing.assert_allclose(slab5.numpy(), ref[p_lo:p_hi], rtol=1e-12, atol=1e-12)
                assert slab5.data_ptr() == keep.data_ptr()

        # layout 4 (what ShardedDeviceModule uses): rows of one leading index in, rows of the other one out,
        # streamed chunk by chunk -- both shardings, several chunk sizes, the buffer reused, the input untouched
        j_lo, j_hi = sharded.SlabPartition(M, world).bounds(rank)
        i_lo, i_hi = sharded.SlabPartition(L, world).bounds(rank)
        for second in (False, True):
            full_rows = tu.transpose(0, 1) if second else tu          # rows[i][j] = u[j, i] resp. u[i, j]
            want = ref if second else ref.transpose(1, 0, 2, 3)       # out_rows[j'][i']
            rows = full_rows[i_lo:i_hi].contiguous()
            keep = None
            for ni in (1, 2, None):
                got = sharded.transform_two_body_rows(rows, tC, tCt, rank, world, engine=OracleEngine, chunk_rows=ni,
                                                      out=keep)
                np.testing.assert_allclose(got.numpy(), want[j_lo:j_hi], rtol=1e-12, atol=1e-12)
                assert got.is_contiguous() and torch.equal(rows, full_rows[i_lo:i_hi])
                if keep is None:
                    keep = torch.empty(sharded.rows_buffer_elems(L, M, j_hi - j_lo), dtype=got.dtype)
                else:
                    assert got.data_ptr() == keep.data_ptr()
        if not cplx:
            # a real tensor against complex coefficients: cast chunk-wise, never as a whole
            Cc = tC * (1 + 0.5j)
            got = sharded.transform_two_body_rows(tu[i_lo:i_hi].contiguous(), Cc, None, rank, world, engine=OracleEngine,
                                                  chunk_rows=1)
            refc = orc.transform_two_body(u, Cc.numpy())
            np.testing.assert_allclose(got.numpy(), refc.transpose(1, 0, 2, 3)[j_lo:j_hi], rtol=1e-12, atol=1e-12)
        if L % 2 == 0 and L // 2 >= 1:
            # the partition spin doubling leaves behind (twice the offsets of L/2 spatial rows)
            part = sharded.SlabPartition(L // 2, world).doubled()
            d_lo, d_hi = part.bounds(rank)
            got = sharded.transform_two_body_rows(tu[d_lo:d_hi].contiguous(), tC, tCt, rank, world, engine=OracleEngine,
                                                  in_part=part, chunk_rows=2)
            np.testing.assert_allclose(got.numpy(), ref.transpose(1, 0, 2, 3)[j_lo:j_hi], rtol=1e-12, atol=1e-12)

        # default bra (C^dagger) path
        slab3 = sharded.transform_two_body_sharded(ub, tC, None, rank, world, engine=OracleEngine)
        np.testing.assert_allclose(
            slab3.numpy(), orc.transform_two_body(u, C)[p_lo:p_hi], rtol=1e-12, atol=1e-12
        )
    # first consumers of a p-sharded u: Fock matrix and reference energy,
    # against the formulas of spatial_orbital_system.py:106-190 / general_orbital_system.py:75-159
    for (l, n_occ, cplx) in [(7, 3, False), (6, 6, True), (5, 1, True)]:
        h = rng.standard_normal((l, l))
        u = rng.standard_normal((l,) * 4)
        if cplx:
            h = h + 1j * rng.standard_normal((l, l))
            u = u + 1j * rng.standard_normal((l,) * 4)
        o = slice(0, n_occ)
        lo, hi = sharded.SlabPartition(l, world).bounds(rank)
        th, tslab = torch.from_numpy(h), torch.from_numpy(np.ascontiguousarray(u[lo:hi]))
        f_s = h + 2 * np.einsum("piqi->pq", u[:, o, :, o]) - np.einsum("piiq->pq", u[:, o, o, :])
        f_g = h + np.einsum("piqi->pq", u[:, o, :, o])
        e_s = 2 * np.trace(h[o, o]) + 2 * np.einsum("ijij->", u[o, o, o, o]) - np.einsum("ijji->", u[o, o, o, o]) + 0.25
        e_g = np.trace(h[o, o]) + 0.5 * np.einsum("ijij->", u[o, o, o, o]) + 0.25
        for spin, f_ref, e_ref in ((False, f_s, e_s), (True, f_g, e_g)):
            f = sharded.construct_fock_matrix_sharded(th, tslab, n_occ, rank, world, spin_orbitals=spin)
            np.testing.assert_allclose(f.numpy(), f_ref, rtol=1e-12, atol=1e-12)
            e = sharded.reference_energy_sharded(th, tslab, n_occ, rank, world, spin_orbitals=spin,
                                                 nuclear_repulsion_energy=0.25)
            np.testing.assert_allclose(complex(e), e_ref, rtol=1e-12, atol=1e-12)
    # ... and pinned to the reference itself: values computed by its SpatialOrbitalSystem /
    # GeneralOrbitalSystem on a seeded RandomBasisSet (tests/golden/fock_energy_random_basis.npz)
    with np.load(os.path.join(ROOT, "tests", "golden", "fock_energy_random_basis.npz")) as z:
        g = {k: z[k] for k in z.files}
    n_occ = int(g["n"]) // 2
    st = dict(h=g["h"], u=g["u"])
    gos = orc.new_state(int(g["l"]), 2)
    gos.update(h=g["h"].copy(), u=g["u"].copy(), s=g["s"].copy())
    gos = orc.change_to_general_orbital_basis(gos)
    for (h, u, n, spin, f_ref, e_ref) in ((st["h"], st["u"], n_occ, False, g["spas_fock"], g["spas_energy"]),
                                          (gos["h"], gos["u"], 2 * n_occ, True, g["gos_fock"], g["gos_energy"])):
        lo, hi = sharded.SlabPartition(h.shape[0], world).bounds(rank)
        th, tslab = torch.from_numpy(h), torch.from_numpy(np.ascontiguousarray(u[lo:hi]))
        f = sharded.construct_fock_matrix_sharded(th, tslab, n, rank, world, spin_orbitals=spin)
        np.testing.assert_allclose(f.numpy(), f_ref, rtol=1e-12, atol=1e-12)
        e = sharded.reference_energy_sharded(th, tslab, n, rank, world, spin_orbitals=spin,
                                             nuclear_repulsion_energy=float(g["e_nuc"]))
        np.testing.assert_allclose(complex(e), e_ref, rtol=1e-12, atol=1e-12)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
