"""The multi-rank logic of the C ABI's sharded transform (qs_transform_two_body_sharded, csrc/qs_comm.hip) without
GPUs: the library hands out the exchange plan of every rank as numbers (qs_sharded_exchange_plan), the test replays
the plans of a whole world with NumPy -- the three local contractions, the grouped sends / receives matched pairwise
in issue order as RCCL matches them, the closing contraction on the received rows -- and compares every rank's
result slab with the oracle's full transform.  (With real RCCL the same code runs on the GPU box with one rank:
tests/test_gpu_comm_cabi.py; worlds of 2..8 need the 8-GPU node.)"""

import ctypes

import numpy as np
import pytest

from oracle import qs_oracle as orc


def plan_of(lib, L, M, world, rank, nchunks):
    header = (ctypes.c_int64 * 7)()
    ct_rows = (ctypes.c_int64 * M)()
    chunks = (ctypes.c_int64 * (4 * 16))()
    rows = 4 * M + 64
    table = (ctypes.c_int64 * (7 * rows))()
    n = lib.qs_sharded_exchange_plan(L, M, world, rank, nchunks, header, ct_rows, chunks, table, rows)
    assert n >= 0, n
    ops = np.array(table[: 7 * n], dtype=np.int64).reshape(n, 7)
    nch = int(header[6])
    return dict(b_lo=header[0], bl=header[1], p_lo=header[2], pc=header[3], row_x=header[4], row_r=header[5],
                nchunks=nch, ct_rows=np.array(ct_rows[:M]), chunks=np.array(chunks[: 4 * nch]).reshape(nch, 4), ops=ops)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("L,M,nchunks,cplx", [(8, 8, 4, False), (9, 7, 3, True), (16, 12, 1, False), (11, 11, 16, False)])
def test_replayed_exchange_gives_the_transform(world, L, M, nchunks, cplx):
    from quantum_systems_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(L * 1000 + M * 10 + world)
    u = rng.standard_normal((L,) * 4)
    C = rng.standard_normal((L, M))
    Ct = rng.standard_normal((M, L))
    if cplx:
        u = u + 1j * rng.standard_normal((L,) * 4)
        C = C + 1j * rng.standard_normal((L, M))
        Ct = Ct + 1j * rng.standard_normal((M, L))
    ref = orc.transform_two_body(u, C, Ct)
    MM = M * M
    plans = [plan_of(lib, L, M, world, r, nchunks) for r in range(world)]
    # the partition is the balanced one of sharded.SlabPartition
    from quantum_systems_amd.sharded import SlabPartition
    for r, pl in enumerate(plans):
        assert (pl["b_lo"], pl["b_lo"] + pl["bl"]) == SlabPartition(L, world).bounds(r)
        assert (pl["p_lo"], pl["p_lo"] + pl["pc"]) == SlabPartition(M, world).bounds(r)
        assert sorted(pl["ct_rows"].tolist()) == list(range(M))                   # every row of Ct exactly once
        assert pl["chunks"][:, 1].sum() == M and pl["chunks"][:, 3].sum() == pl["pc"]
    # local phase of every rank: X rows in exchange order, flat; R zero-filled
    X, R = [], []
    for r, pl in enumerate(plans):
        ub = u[:, pl["b_lo"]:pl["b_lo"] + pl["bl"]]
        t = np.tensordot(ub, C, axes=(3, 0))                                       # d
        t = np.tensordot(t, C, axes=(2, 0)).transpose(0, 1, 3, 2)                  # c
        x = np.tensordot(Ct, t, axes=(1, 0))                                       # a (local in this layout): [p, b, r, s]
        X.append(np.ascontiguousarray(x[pl["ct_rows"]]).reshape(-1))
        R.append(np.full(pl["pc"] * L * MM, np.nan, dtype=ref.dtype))
    # exchange, chunk by chunk: the k-th send of rank a to rank b pairs with the k-th receive of b from a
    for k in range(max(pl["nchunks"] for pl in plans)):
        sends, recvs = {}, {}
        for r, pl in enumerate(plans):
            for (chunk, peer, kind, x_off, r_off, count, rows) in pl["ops"]:
                if chunk != k:
                    continue
                if kind == 0:
                    sends.setdefault((r, peer), []).append((x_off, count))
                elif kind == 1:
                    recvs.setdefault((peer, r), []).append((r_off, count))
                else:                                                              # own rows, strided copy
                    assert peer == r
                    for i in range(rows):
                        src = X[r][x_off + i * pl["row_x"]: x_off + i * pl["row_x"] + count]
                        R[r][r_off + i * pl["row_r"]: r_off + i * pl["row_r"] + count] = src
        assert set(sends) == set(recvs)
        for (src, dst), msgs in sends.items():
            got = recvs[(src, dst)]
            assert [c for (_, c) in msgs] == [c for (_, c) in got], (src, dst)     # RCCL needs matching sizes in order
            for (x_off, count), (r_off, _) in zip(msgs, got):
                R[dst][r_off: r_off + count] = X[src][x_off: x_off + count]
    # closing contraction on the received rows
    for r, pl in enumerate(plans):
        assert not np.isnan(R[r]).any()                                            # every element of R was delivered
        rows = R[r].reshape(pl["pc"], L, MM)
        out = np.einsum("qb,pbn->pqn", Ct, rows).reshape(pl["pc"], M, M, M)
        np.testing.assert_allclose(out, ref[pl["p_lo"]:pl["p_lo"] + pl["pc"]], rtol=1e-11, atol=1e-11)
