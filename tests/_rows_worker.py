"""Worker for tests/test_gpu_sharded_rows.py (torch.distributed.run, gloo carrying the collectives, the HIP engine,
every rank on cuda:0 -- the one-device rehearsal of a node): the streamed rows-in / rows-out transform behind the
sharded array module against the oracle, bit-for-bit against the out-of-place layouts of rounds 1-2, and within its
memory bound: input rows + result rows + 2 rows + c l^3 per rank, c = 7 chunk_rows."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qs_oracle as orc  # noqa: E402
import quantum_systems_amd as qsa  # noqa: E402
from quantum_systems_amd import kernels as K, sharded  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)                      # the same stream on every rank
    for (L, M, cplx) in ((12, 12, False), (18, 18, True), (14, 9, True), (9, 14, False), (66, 66, False)):
        u = rng.standard_normal((L,) * 4)
        C = rng.standard_normal((L, M)) / np.sqrt(L)
        Ct = rng.standard_normal((M, L)) / np.sqrt(L)
        if cplx:
            u = u + 1j * rng.standard_normal((L,) * 4)
            C = C + 1j * rng.standard_normal((L, M)) / np.sqrt(L)
            Ct = Ct + 1j * rng.standard_normal((M, L)) / np.sqrt(L)
        ref = orc.transform_two_body(u, C, Ct)
        es = 16 if cplx else 8
        dC, dCt = torch.from_numpy(C).to(dev), torch.from_numpy(Ct).to(dev)
        i_lo, i_hi = sharded.SlabPartition(L, world).bounds(rank)
        j_lo, j_hi = sharded.SlabPartition(M, world).bounds(rank)
        # the single-GPU transform of the same numbers: every element one chain of fused multiply-adds per contraction, in
        # index order -- the streamed layout keeps exactly those chains (each product has the WHOLE contracted index)
        whole = K.transform_two_body(torch.from_numpy(u).to(dev), dC, dCt)
        # the out-of-place layouts of rounds 1-2 on the same numbers: they close the sharded index peer block by peer
        # block (accumulating products), so beyond one rank they agree to rounding only
        old_a = sharded.transform_two_body_sharded_a(torch.from_numpy(np.ascontiguousarray(u[i_lo:i_hi])).to(dev), dC, dCt,
                                                     rank, world)                    # (M, ql, M, M) = out[:, q_lo:q_hi]
        old_b = sharded.transform_two_body_sharded(torch.from_numpy(np.ascontiguousarray(u[:, i_lo:i_hi])).to(dev), dC, dCt,
                                                   rank, world)                      # (pc, M, M, M) = out[p_lo:p_hi]
        for second in (False, True):
            full_rows = u.transpose(1, 0, 2, 3) if second else u
            want = ref if second else ref.transpose(1, 0, 2, 3)
            rows = torch.from_numpy(np.ascontiguousarray(full_rows[i_lo:i_hi])).to(dev)
            for ni in (1, 3, None):
                K.workspace.release()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
                base = torch.cuda.memory_allocated()
                got = sharded.transform_two_body_rows(rows, dC, dCt, rank, world, chunk_rows=ni)
                torch.cuda.synchronize()
                peak = torch.cuda.max_memory_allocated() - base
                n_eff = min(ni or 10**9, sharded.stream_chunk_rows(L, M, -(-L // world), es) if ni is None else ni,
                            -(-L // world))
                lmax = max(L, M)
                bound = ((j_hi - j_lo) * lmax * M * M + 2 * lmax**3 + 7 * n_eff * lmax**3) * es + (1 << 20)
                assert peak <= bound, (L, M, ni, peak, bound)
                g = got.cpu().numpy()
                assert np.abs(g - want[j_lo:j_hi]).max() <= 1e-10 * np.abs(ref).max(), (L, M, second, ni)
                mine = whole[j_lo:j_hi] if second else whole[:, j_lo:j_hi].transpose(0, 1)
                if not second:
                    assert torch.equal(got, mine), (L, M, second, ni)                # d, c, b, a: the same sums in the same order
                else:                                                                # d, c, a, b (the order of the b-sharded layouts)
                    assert (got - mine).abs().max().item() <= 1e-12 * mine.abs().max().item(), (L, M, second, ni)
                old = old_b if second else old_a.transpose(0, 1)
                if world == 1:
                    assert torch.equal(got, old), (L, M, second, ni)
                else:
                    assert (got - old).abs().max().item() <= 1e-12 * old.abs().max().item(), (L, M, second, ni)
                del got
        del old_a, old_b, whole
    # ---- through the array module: ShardedTensor4 in, ShardedTensor4 out, the sharded index flips and flips back
    mod = qsa.ShardedDeviceModule(rank, world, device="cuda:0")
    assert mod.rccl() is None                          # gloo: the exchange goes through torch.distributed
    L = 10
    u = rng.standard_normal((L,) * 4) + 1j * rng.standard_normal((L,) * 4)
    C = np.linalg.qr(rng.standard_normal((L, L)) + 1j * rng.standard_normal((L, L)))[0]
    t = mod.shard(u)
    one = qsa.BasisSet.transform_two_body_elements(t, mod.asarray(C), mod)
    ref = orc.transform_two_body(u, C)
    assert one.axis == 1 and tuple(one.rows.shape)[1:] == (L, L, L)
    assert np.abs(one.local.cpu().numpy() - ref[:, one.lo:one.hi]).max() <= 1e-10 * np.abs(ref).max()
    back = qsa.BasisSet.transform_two_body_elements(one, mod.asarray(C.conj().T.copy()), mod)
    assert back.axis == 0
    assert np.abs(back.local.cpu().numpy() - u[back.lo:back.hi]).max() <= 1e-9 * np.abs(u).max()      # unitary round trip
    assert np.abs(one.gather().cpu().numpy() - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(one.reshard(0).local.cpu().numpy() - ref[one.reshard(0).lo:one.reshard(0).hi]).max() <= 1e-10 * np.abs(ref).max()
    if os.environ.get("QS_ROWS_WORKER_RCCL") == "1":
        # the same through the module's own communicator: ONE C-ABI call per transform on every rank (the transport is the
        # test suite's stand-in for librccl, QS_AMD_RCCL_LIB: real RCCL wants one GPU per rank)
        cmod = qsa.ShardedDeviceModule(rank, world, device="cuda:0", exchange="rccl")
        tc = cmod.shard(u)
        onec = qsa.BasisSet.transform_two_body_elements(tc, cmod.asarray(C), cmod)
        assert cmod.rccl() is not None and "rccl grouped send/recv" in K.last_dispatch(), K.last_dispatch()
        assert onec.axis == 1 and torch.equal(onec.rows, one.rows)            # the same sums in the same order
        backc = qsa.BasisSet.transform_two_body_elements(onec, cmod.asarray(C.conj().T.copy()), cmod)
        assert backc.axis == 0 and np.abs(backc.local.cpu().numpy() - u[backc.lo:backc.hi]).max() <= 1e-9 * np.abs(u).max()
        # a real tensor against complex coefficients stays real on its way in
        ur = rng.standard_normal((L,) * 4)
        outr = qsa.BasisSet.transform_two_body_elements(cmod.shard(ur), cmod.asarray(C), cmod)
        refr = orc.transform_two_body(ur, C)
        assert np.abs(outr.local.cpu().numpy() - refr[:, outr.lo:outr.hi]).max() <= 1e-10 * np.abs(refr).max()
        cmod.rccl().close()
        print(f"rank {rank}/{world} module on the C entry")
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main()
