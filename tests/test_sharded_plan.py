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


# ---------------------------------------------------------------------------------------------------------------
# rows in, rows out (qs_transform_two_body_sharded_rows): the streamed, memory-lean form
# ---------------------------------------------------------------------------------------------------------------


def rows_plan_of(lib, L, M, world, rank, starts, chunk_rows, coalesced=False):
    header = (ctypes.c_int64 * 8)()
    cap = 64 + 4 * (M + world) * (L + 1)
    table = (ctypes.c_int64 * (7 * cap))()
    p_starts = (ctypes.c_int64 * (world + 1))(*starts) if starts is not None else None
    fn = lib.qs_sharded_rows_exchange_plan_coalesced if coalesced else lib.qs_sharded_rows_exchange_plan
    n = fn(L, M, world, rank, p_starts, chunk_rows, header, table, cap)
    assert n >= 0, n
    keys = ("i_start", "il", "jl", "il_max", "r0", "out_elems", "chunk_rows", "nsteps")
    pl = dict(zip(keys, (int(x) for x in header)))
    pl["ops"] = np.array(table[: 7 * n], dtype=np.int64).reshape(n, 7)
    return pl


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("L,M,chunk_rows,cplx,doubled", [
    (8, 8, 1, False, False), (9, 7, 2, True, False), (6, 10, 1, True, False), (12, 12, 3, False, False),
    (10, 8, 2, True, True),         # the partition spin doubling leaves behind (twice the offsets of 5 spatial rows)
    (11, 11, 0, False, False),      # chunk_rows <= 0: the library's own choice
])
@pytest.mark.parametrize("second_index", [False, True])
@pytest.mark.parametrize("coalesced", [False, True])
def test_replayed_rows_exchange_gives_the_transform(world, L, M, chunk_rows, cplx, doubled, second_index, coalesced):
    # Replay of every rank's plan with NumPy: per step the three local products into the send block W[j'][i][(r,s)],
    # the grouped sends / receives matched pairwise in issue order, the own-rows copy; after the last step the closing
    # product row by row INSIDE the result buffer.  Checked: every rank's result rows against the oracle (both
    # shardings: the routine is symmetric in the two leading indices), nothing read before it arrived, and the
    # in-buffer product never overwriting a received row that is still to be read.  coalesced: ONE message per peer and
    # step (the handle's option "rows_coalesce"): the peer's block of W as it is, received into a staging area whose size
    # is the bound qs_comm_rows_workspace adds, put in place by strided copies behind the group.
    from quantum_systems_amd import _lib
    from quantum_systems_amd.sharded import SlabPartition

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
    starts = SlabPartition(L // 2, world).doubled().starts if doubled else None
    ipart = SlabPartition(L, world, starts)
    jpart = SlabPartition(M, world)
    plans = [rows_plan_of(lib, L, M, world, r, starts, chunk_rows, coalesced) for r in range(world)]
    ni, nsteps = plans[0]["chunk_rows"], plans[0]["nsteps"]
    for r, pl in enumerate(plans):
        assert (pl["i_start"], pl["i_start"] + pl["il"]) == ipart.bounds(r) and pl["jl"] == jpart.count(r)
        assert (pl["chunk_rows"], pl["nsteps"]) == (ni, nsteps) and nsteps == -(-pl["il_max"] // ni)
        assert pl["out_elems"] == pl["jl"] * max(L, M) * MM + L * MM
    # rows[i][j] = u[i_lo + i, j] (leading-index sharding) or u[j, i_lo + i] (second-index sharding)
    full_rows = u.transpose(1, 0, 2, 3) if second_index else u
    want = ref.transpose(1, 0, 2, 3) if not second_index else ref          # out_rows[j'][i'] = out[i', j'] resp. out[j', i']
    bufs = [np.full(pl["out_elems"], np.nan, dtype=ref.dtype) for pl in plans]
    for t in range(nsteps):
        Ws = []
        for r, pl in enumerate(plans):
            i0 = t * ni
            n = max(0, min(ni, pl["il"] - i0))
            rows = full_rows[pl["i_start"] + i0: pl["i_start"] + i0 + n]
            t2 = np.einsum("ijcd,cr,ds->ijrs", rows, C, C)                  # d, c
            W = np.einsum("kj,ijrs->kirs", Ct, t2)                          # J: W[j', i, r, s]
            Ws.append(np.ascontiguousarray(W).reshape(-1))
        sends, recvs, scatters = {}, {}, []
        stages = [np.full(pl["jl"] * (world - 1) * ni * MM, np.nan, dtype=ref.dtype) for pl in plans]
        for r, pl in enumerate(plans):
            n = max(0, min(ni, pl["il"] - t * ni))
            group_closed = False
            for (step, peer, kind, w_off, b_off, count, nrows) in pl["ops"]:
                if step != t:
                    continue
                assert kind in ((0, 2, 3, 4) if coalesced else (0, 1, 2))
                assert not (group_closed and kind != 4)                   # the copies out of the staging area come last
                if kind == 0:
                    sends.setdefault((r, peer), []).append((w_off, count))
                    assert not coalesced or len(sends[(r, peer)]) == 1     # one message per peer and step
                elif kind == 1:
                    recvs.setdefault((peer, r), []).append((b_off, count))
                elif kind == 3:
                    assert b_off + count <= stages[r].size                  # inside the staging area the workspace query adds
                    recvs.setdefault((peer, r), []).append((b_off, count))
                elif kind == 4:
                    group_closed = True
                    scatters.append((r, w_off, b_off, count, nrows))
                else:
                    assert peer == r and count == n * MM
                    for i in range(nrows):
                        bufs[r][b_off + i * L * MM: b_off + i * L * MM + count] = Ws[r][w_off + i * count: w_off + (i + 1) * count]
        assert set(sends) == set(recvs)
        for (src, dst), msgs in sends.items():
            got = recvs[(src, dst)]
            assert [c for (_, c) in msgs] == [c for (_, c) in got], (src, dst)
            for (w_off, count), (b_off, _) in zip(msgs, got):
                (stages if coalesced else bufs)[dst][b_off: b_off + count] = Ws[src][w_off: w_off + count]
        for (r, s_off, b_off, count, nrows) in scatters:
            for i in range(nrows):
                piece = stages[r][s_off + i * count: s_off + (i + 1) * count]
                assert not np.isnan(piece).any()
                bufs[r][b_off + i * L * MM: b_off + i * L * MM + count] = piece
    for r, pl in enumerate(plans):
        buf, r0, jl = bufs[r], pl["r0"], pl["jl"]
        assert not np.isnan(buf[r0: r0 + jl * L * MM]).any()                # every received element was delivered
        for p in range(jl):
            row = buf[r0 + p * L * MM: r0 + (p + 1) * L * MM].reshape(L, MM).copy()
            assert p * M * MM + M * MM <= r0 + p * L * MM                  # the product stops short of the row it reads ...
            buf[p * M * MM: (p + 1) * M * MM] = (Ct @ row).reshape(-1)      # ... and of every later one
        got = buf[: jl * M * MM].reshape(jl, M, M, M)
        j_lo, j_hi = jpart.bounds(r)
        np.testing.assert_allclose(got, want[j_lo:j_hi], rtol=1e-11, atol=1e-11)
