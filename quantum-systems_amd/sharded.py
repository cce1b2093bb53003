"""Sharding of the four-index transform over the GPUs of one node
(one process per GPU, ``torch.distributed`` with the ``nccl`` backend = RCCL
over xGMI; ``gloo`` in the CPU tests).

The reference has no parallelism at all; this layer exists so the
same ``transform_two_body_elements`` call scales past one device.  Two layouts,
both producing the result sharded over its LEADING index ``p``
(``out[p_lo:p_hi, :, :, :]`` on each rank, contiguous):

``transform_two_body_replicated``
    ``u`` is resident on every rank (l=256 fp64: 34 GB of 288 GB).  Rank g
    contracts ``a`` first with its rows of ``Ct`` and then d, c, b on its own
    slab.  No collective on the data path; 8 l^5 / G flops per rank.
    ``all_gather_slabs`` replicates the p-sharded result when a caller wants
    the whole tensor everywhere (the single all-gather of the north star).

``transform_two_body_sharded``
    ``u`` is sharded over its SECOND index (``u[:, b_lo:b_hi]``), so it never
    has to fit on one device (l=512 complex128 = 1.1 TB).  d, c and a are
    contracted locally -- ``a`` is fully local in this layout -- then ONE
    all-to-all re-shards ``[p, b_loc] -> [p_loc, b]`` (per-rank traffic
    (G-1)/G^2 * l^4 elements, G times less than an all-gather of the
    intermediate) and the contraction over b closes on the received slabs.

The contraction order differs from the reference's d, c, b, a only in where the
``a`` sum sits; every element is still the same four sums, and parity is
checked to the same 1e-10 bound.

All arithmetic goes through an *engine* with two methods (``matmul`` and
``partial``).  The only engine in the package is the HIP one below; the CPU
tests inject their own (oracle-backed) engine to exercise partitioning and the
exchange under ``gloo`` with world_size 2.
"""

import torch
import torch.distributed as dist

from . import kernels


class SlabPartition:
    """Contiguous split of ``n`` rows over ``world`` ranks: balanced by default, or the explicit
    ``starts`` (world + 1 non-decreasing offsets from 0 to n) -- spin doubling turns the balanced split of
    l spatial rows into twice those offsets, which is not the balanced split of 2l when l % world != 0."""

    def __init__(self, n, world, starts=None):
        if world < 1 or n < 1:
            raise ValueError("need n >= 1 and world >= 1")
        self.n, self.world = int(n), int(world)
        if starts is None:
            base, extra = divmod(self.n, self.world)
            self.starts = [r * base + min(r, extra) for r in range(self.world + 1)]
        else:
            self.starts = [int(x) for x in starts]
            if (len(self.starts) != self.world + 1 or self.starts[0] != 0 or self.starts[-1] != self.n
                    or any(b < a for a, b in zip(self.starts, self.starts[1:]))):
                raise ValueError(f"bad partition {self.starts} of {self.n} rows over {self.world} ranks")

    def bounds(self, rank):
        return self.starts[rank], self.starts[rank + 1]

    def count(self, rank):
        lo, hi = self.bounds(rank)
        return hi - lo

    def doubled(self):
        """The partition of the 2n spin rows P = 2p + sigma that keeps every rank's rows together."""
        return SlabPartition(2 * self.n, self.world, [2 * x for x in self.starts])

    def is_balanced(self):
        return self.starts == SlabPartition(self.n, self.world).starts


class HipEngine:
    """The product engine: every call lands in libqs_amd.so."""

    name = "hip"

    @staticmethod
    def matmul(A, B, out=None, accumulate=False):
        return kernels.matmul(A, B, out=out, accumulate=accumulate)

    @staticmethod
    def partial(u_slab, C, C_tilde):
        return kernels.transform_two_body_partial(u_slab, C, C_tilde)

    @staticmethod
    def gemm_strided(dt, A, B, out, m, n, k, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0,
                     accumulate=False, a_off=0, b_off=0, c_off=0):
        """Batched product on sub-blocks of larger buffers (offsets, leading
        dimensions and batch strides in elements) -- ``qs_matmul`` as is."""
        return kernels.gemm_raw(dt, A, B, out, m, n, k, lda, ldb, ldc, batch, sa, sb, sc,
                                accumulate, a_off, b_off, c_off)

    # the slab-local operations behind a sharded BasisSet (sharded_basis.py)
    @staticmethod
    def transform_one_body(h, C, C_tilde):
        return kernels.transform_one_body(h, C, C_tilde)

    @staticmethod
    def add_spin_one_body(h, out_dtype=None):
        return kernels.add_spin_one_body(h, out_dtype=out_dtype)

    @staticmethod
    def spin_expand_block(u_block, antisymmetrize=False, out_dtype=None):
        return kernels.spin_expand_two_body_block(u_block, antisymmetrize=antisymmetrize, out_dtype=out_dtype)

    @staticmethod
    def antisymmetrize(u_block, in_place=False):
        return kernels.antisymmetrize(u_block, out=u_block if in_place else None)

    @staticmethod
    def spin_squared_two_body(S, antisymmetrize=False, p_lo=0, p_hi=None):
        return kernels.spin_squared_two_body(S, antisymmetrize=antisymmetrize, p_lo=p_lo, p_hi=p_hi)


def _bra(C, C_tilde):
    return kernels.default_bra(C) if C_tilde is None else C_tilde


def transform_two_body_replicated(u, C, C_tilde=None, rank=0, world=1, engine=HipEngine):
    """Rows ``p_lo:p_hi`` of the transformed tensor from a replicated ``u``.

    out[p_loc, q, r, s] = sum_bcd Ct[q,b] (sum_a Ct[p_loc,a] u[a,b,c,d]) C[c,r] C[d,s]
    """
    Ct = _bra(C, C_tilde)
    L, M = C.shape
    lo, hi = SlabPartition(M, world).bounds(rank)
    if hi == lo:
        return u.new_empty((0, M, M, M))
    rows = Ct[lo:hi].contiguous()
    w = engine.matmul(rows, u.reshape(L, L * L * L))          # (p_loc, bcd)
    return engine.partial(w.reshape(hi - lo, L, L, L), C, Ct)  # (p_loc, q, r, s)


def _as_real_flat(t):
    """1-D float64 view for the collectives (complex -> interleaved pairs)."""
    t = t.contiguous()
    if t.is_complex():
        t = torch.view_as_real(t)
    return t.reshape(-1)


def all_gather_slabs(out_slab, M, rank, world, group=None, full=None, part=None):
    """Replicate the p-sharded result: returns the full (M,M,M,M) tensor (the single
    all-gather of the north star).  Even slabs are gathered straight into the result
    tensor (no staging copy); uneven slabs are broadcast one by one into their places
    (no staging either).  ``full`` may supply the result buffer."""
    part = part or SlabPartition(M, world)
    if world == 1:
        return out_slab
    per_row = M * M * M
    width = 2 if out_slab.is_complex() else 1
    if full is None:
        full = torch.empty((M, M, M, M), dtype=out_slab.dtype, device=out_slab.device)
    full_flat = _as_real_flat(full)  # view of `full`
    flat = _as_real_flat(out_slab)
    if M % world == 0 and part.is_balanced():
        dist.all_gather_into_tensor(full_flat, flat, group=group)
        return full
    # uneven slabs: every rank's slab is broadcast straight into its place in the result (G collectives on contiguous
    # views of `full`, no padded staging copy of the tensor on either side)
    for r in range(world):
        lo, hi = part.bounds(r)
        if hi == lo:
            continue
        view = full_flat[lo * per_row * width: hi * per_row * width]
        if r == rank:
            view.copy_(flat)
        dist.broadcast(view, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
    return full


def transform_two_body_sharded(u_bslab, C, C_tilde=None, rank=0, world=1, group=None,
                               engine=HipEngine, in_part=None):
    """p-slab of the transform from a ``u`` sharded over its second index.

    ``u_bslab = u[:, b_lo:b_hi, :, :]`` (contiguous) with the balanced split of
    ``SlabPartition(L, world)``.  Returns ``out[p_lo:p_hi]`` with the split of
    ``SlabPartition(M, world)``.  Exactly one collective (all-to-all).
    """
    Ct = _bra(C, C_tilde)
    L, M = C.shape
    bpart, ppart = (in_part or SlabPartition(L, world)), SlabPartition(M, world)   # in_part: a non-balanced input split
    b_lo, b_hi = bpart.bounds(rank)
    bl = b_hi - b_lo
    if tuple(u_bslab.shape) != (L, bl, L, L):
        raise ValueError(f"rank {rank}: slab shape {tuple(u_bslab.shape)}, expected {(L, bl, L, L)}")
    dt = kernels.result_dtype(u_bslab, C, Ct)
    u_bslab, C, Ct = u_bslab.to(dt), C.to(dt), Ct.to(dt)
    width = 2 if dt.is_complex else 1
    p_lo, p_hi = ppart.bounds(rank)
    pc = p_hi - p_lo

    if bl > 0:
        # d:  T1[(a,b,c), s] = u[(a,b,c), d] C[d, s]
        t1 = engine.matmul(u_bslab.reshape(L * bl * L, L), C)
        # c:  T2[(a,b)][r, s] = CT[r, c] T1[(a,b)][c, s]
        CT = C.transpose(0, 1).contiguous()
        t2 = engine.matmul(CT, t1.reshape(L * bl, L, M))
        del t1
        # a:  X[p, (b,r,s)] = Ct[p, a] T2[a, (b,r,s)]      (a is local in this layout)
        x = engine.matmul(Ct, t2.reshape(L, bl * M * M))
        del t2
    else:
        x = u_bslab.new_empty((M, 0), dtype=dt)

    if world == 1:
        recv_blocks = [x.reshape(M, bl, M * M)]
    else:
        # one all-to-all: rows p of X go to the owner of p; we receive, from every
        # source g, the block [p_loc, b in slab(g), (r,s)]
        send = _as_real_flat(x)
        in_splits = [ppart.count(g) * bl * M * M * width for g in range(world)]
        out_splits = [pc * bpart.count(g) * M * M * width for g in range(world)]
        recv = torch.empty(sum(out_splits), dtype=torch.float64, device=send.device)
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=group)
        del x, send
        recv_blocks, off = [], 0
        for g in range(world):
            blk = recv[off: off + out_splits[g]]
            off += out_splits[g]
            if width == 2:
                blk = torch.view_as_complex(blk.reshape(-1, 2))
            recv_blocks.append(blk.reshape(pc, bpart.count(g), M * M))

    # b:  out[p][q, (r,s)] = sum_g Ct[q, b in slab(g)] R_g[p][b, (r,s)]
    out = torch.empty((pc, M, M * M), dtype=dt, device=u_bslab.device)
    if pc == 0:
        return out.reshape(0, M, M, M)
    first = True
    for g in range(world):
        g_lo, g_hi = bpart.bounds(g)
        if g_hi == g_lo:
            continue
        engine.matmul(Ct[:, g_lo:g_hi].contiguous(), recv_blocks[g], out=out, accumulate=not first)
        first = False
    return out.reshape(pc, M, M, M)


def transform_two_body_sharded_a(u_aslab, C, C_tilde=None, rank=0, world=1, group=None,
                                 engine=HipEngine, in_part=None):
    """The mirror image of ``transform_two_body_sharded``: ``u`` sharded over its LEADING index
    (``u_aslab = u[a_lo:a_hi]``, the layout every slab-local producer leaves behind) -> the result sharded
    over its SECOND index, ``out[:, q_lo:q_hi]`` of shape (M, ql, M, M).

    d, c and b are contracted on the slab (``qs_transform_two_body_partial``: rows of the leading index
    are independent there), ONE all-to-all re-shards ``[a_loc, q] -> [a, q_loc]`` -- the same
    (G-1)/G^2 l^4 elements per rank as the other layout -- and the contraction over a closes on the
    received slabs.  A transform therefore flips which of the two leading indices is the sharded one;
    every consumer on the path (anti-symmetrisation, spin doubling, Fock matrix, reference energy, the
    next transform) works on either."""
    Ct = _bra(C, C_tilde)
    L, M = C.shape
    apart, qpart = (in_part or SlabPartition(L, world)), SlabPartition(M, world)
    a_lo, a_hi = apart.bounds(rank)
    al = a_hi - a_lo
    if tuple(u_aslab.shape) != (al, L, L, L):
        raise ValueError(f"rank {rank}: slab shape {tuple(u_aslab.shape)}, expected {(al, L, L, L)}")
    dt = kernels.result_dtype(u_aslab, C, Ct)
    u_aslab, C, Ct = u_aslab.to(dt), C.to(dt), Ct.to(dt)
    width = 2 if dt.is_complex else 1
    q_lo, q_hi = qpart.bounds(rank)
    ql = q_hi - q_lo
    MM = M * M

    # d, c, b on the slab:  v[a_loc, q, r, s]
    v = engine.partial(u_aslab.contiguous(), C, Ct) if al > 0 else u_aslab.new_empty((0, M, M, M), dtype=dt)

    if world == 1:
        recv_blocks = [v.reshape(al, M * MM)]
    else:
        # one all-to-all: columns q of v go to the owner of q.  Packed per destination g as
        # [a_loc][q in slab(g)][(r, s)]; from source g we receive [a in slab(g)][q_loc][(r, s)]
        send = torch.cat([v[:, qpart.bounds(g)[0]:qpart.bounds(g)[1]].reshape(-1) for g in range(world)])
        send = _as_real_flat(send)
        in_splits = [al * qpart.count(g) * MM * width for g in range(world)]
        out_splits = [apart.count(g) * ql * MM * width for g in range(world)]
        recv = torch.empty(sum(out_splits), dtype=torch.float64, device=send.device)
        del v
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=group)
        del send
        recv_blocks, off = [], 0
        for g in range(world):
            blk = recv[off: off + out_splits[g]]
            off += out_splits[g]
            if width == 2:
                blk = torch.view_as_complex(blk.reshape(-1, 2))
            recv_blocks.append(blk.reshape(apart.count(g), ql * MM))

    # a:  out[p, (q_loc, r, s)] = sum_g Ct[p, a in slab(g)] R_g[a, (q_loc, r, s)]
    out = torch.empty((M, ql * MM), dtype=dt, device=u_aslab.device)
    if ql == 0:
        return out.reshape(M, 0, M, M)
    first = True
    for g in range(world):
        g_lo, g_hi = apart.bounds(g)
        if g_hi == g_lo:
            continue
        engine.matmul(Ct[:, g_lo:g_hi].contiguous(), recv_blocks[g], out=out, accumulate=not first)
        first = False
    return out.reshape(M, ql, M, M)


def transform_two_body_sharded_inplace(u_bslab, C, C_tilde=None, rank=0, world=1, group=None,
                                       engine=HipEngine, staging_rows=1, out=None):
    """Memory-lean form of ``transform_two_body_sharded`` for tensors that only
    just fit the node (BASELINE.json configs[4]: l = 512 complex128, 128 GiB per
    GPU at G = 8).  Square transforms with ``l`` divisible by ``world`` only.

    ``u_bslab = u[:, b_lo:b_hi]`` is left untouched (the per-step caller of
    system.py:222-225 keeps ``u`` resident and transforms it again with the next
    ``C(t)``); everything else happens INSIDE THE OUTPUT BUFFER, which is the
    only slab-sized allocation: (pc + 1, l, l, l) elements for pc = l / world
    result rows.

    1. local phase, one ``b`` at a time (two l^3 temporaries): d, c, then the
       contraction over ``a`` -- local in this layout -- writes
       ``X[p, b, r, s] = Ct[p,a] u[a,b,c,d] C[c,r] C[d,s]`` into the buffer, the row
       of global index ``p = g pc + p'`` going to slot ``(p', g)`` (the rows of
       ``Ct`` are permuted once so that this is one plain GEMM per ``b``);
    2. ONE exchange step, chunked: slot ``(p', g)`` -- what peer g needs from us
       for its row p' -- is swapped for what peer g holds for our row p'
       (``staging_rows`` rows per all-to-all through a staging buffer).  After
       it slot ``(p', g)`` holds ``X[p_loc = p', b in slab(g)]``: row p' of the buffer
       is the whole ``(b, (r, s))`` matrix of that result row, contiguous;
    3. the contraction over b runs row by row in place: the product of row p'
       is stored one row EARLIER than its operand (the buffer has one spare row
       in front), so no copy and no second slab is ever needed.

    Returns the view ``buffer[:pc]`` = ``out[p_lo:p_hi]``.  Peak memory: input
    slab + output slab + one row + O(l^3) + staging.  Same arithmetic and the
    same single exchange as the out-of-place layout.  ``out`` may supply the
    buffer (contiguous, (pc + 1, l, l, l), result dtype) to reuse it across steps.
    """
    Ct = _bra(C, C_tilde)
    L, M = C.shape
    if L != M or L % world:
        raise ValueError("the in-place layout needs a square transform and l divisible by world")
    bl = L // world           # b (and p) rows per rank
    pc = bl
    if tuple(u_bslab.shape) != (L, bl, L, L) or not u_bslab.is_contiguous():
        raise ValueError(f"rank {rank}: expected a contiguous slab of shape {(L, bl, L, L)}")
    dt = kernels.result_dtype(u_bslab, C, Ct)
    if u_bslab.dtype != dt:
        raise ValueError("the slab must already have the result dtype (no slab-sized cast is made)")
    C, Ct = C.to(dt).contiguous(), Ct.to(dt).contiguous()
    CT = C.transpose(0, 1).contiguous()
    dev = u_bslab.device
    L2, L3 = L * L, L * L * L
    if out is None:
        buf = torch.empty((pc + 1, L, L, L), dtype=dt, device=dev)
    else:
        if (tuple(out.shape) != (pc + 1, L, L, L) or out.dtype != dt or not out.is_contiguous()
                or out.device != dev):
            raise ValueError(f"`out` must be a contiguous {dt} buffer of shape {(pc + 1, L, L, L)}")
        buf = out
    # rows of Ct in slot order: slot p' * world + g  <-  global row g * pc + p'
    order = torch.arange(L, device=Ct.device).reshape(world, pc).transpose(0, 1).reshape(-1)
    Ct_slots = Ct[order].contiguous()

    # ---- 1. local phase, one b at a time; X goes to rows 1 .. pc of the buffer
    t1 = torch.empty((L, L, L), dtype=dt, device=dev)
    t2 = torch.empty((L, L, L), dtype=dt, device=dev)
    for j in range(bl):
        # d:  t1[a][c, s] = u[a, j][c, d] C[d, s]            batch over a, A strided by bl*L^2
        engine.gemm_strided(dt, u_bslab, C, t1, L, L, L, L, L, L, batch=L, sa=bl * L2, sb=0, sc=L2,
                            a_off=j * L2)
        # c:  t2[a][r, s] = CT[r, c] t1[a][c, s]
        engine.gemm_strided(dt, CT, t1, t2, L, L, L, L, L, L, batch=L, sa=0, sb=L2, sc=L2)
        # a:  X[slot, j][(r, s)] = Ct_slots[slot, a] t2[a][(r, s)]      slots are bl*L^2 apart
        engine.gemm_strided(dt, Ct_slots, t2, buf, L, L2, L, L, L2, bl * L2, c_off=L3 + j * L2)
    del t1, t2

    # ---- 2. exchange in place: slot (p', g) <-> what peer g holds for our row p'
    x = buf[1:]                                   # (pc, L, L, L) = [p'][g][b][(r, s)]
    if world > 1:
        width = 2 if dt.is_complex else 1
        blk = bl * L2 * width                     # float64 words per slot
        slots = _as_real_flat(x).reshape(pc, world, blk)          # view
        rows = max(1, min(int(staging_rows), pc))
        send = torch.empty((world, rows, blk), dtype=torch.float64, device=dev)
        recv = torch.empty_like(send)
        for r0 in range(0, pc, rows):
            nr = min(rows, pc - r0)
            sv, rv = send[:, :nr], recv[:, :nr]
            if nr != rows:
                sv = torch.empty((world, nr, blk), dtype=torch.float64, device=dev)
                rv = torch.empty_like(sv)
            sv.copy_(slots[r0:r0 + nr].transpose(0, 1))
            dist.all_to_all_single(rv.reshape(-1), sv.reshape(-1), group=group)
            slots[r0:r0 + nr].copy_(rv.transpose(0, 1))
        del send, recv

    # ---- 3. b:  out[p'][q, (r,s)] = Ct[q, b] X[p'][b, (r,s)], stored one row earlier than X[p']
    for p in range(pc):
        engine.gemm_strided(dt, Ct, buf, buf, L, L2, L, L, L2, L2, b_off=(p + 1) * L3, c_off=p * L3)
    return buf[:pc]


# ---------------------------------------------------------------------------
# The layout behind the API (ShardedDeviceModule): rows of one leading index in, rows of the other one out,
# streamed so that a rank never holds more than its input rows, its output rows and O(l^3) of scratch.
# ---------------------------------------------------------------------------

# Scratch the streamed transform may use per rank (t1, t2, the send block and the receive staging, each
# chunk_rows * l^3 elements).  A constant, not a query of free memory: every rank must derive the same chunking.
STREAM_BUDGET_BYTES = 8 << 30


def rows_buffer_elems(L, M, jl):
    """Elements of the result buffer of ``transform_two_body_rows`` for ``jl`` result rows: the rows themselves,
    room for the received rows while they are still (L, M, M) each, and one spare received row (see there)."""
    return jl * max(L, M) * M * M + L * M * M


def stream_chunk_rows(L, M, il_max, elem_bytes, budget_bytes=None):
    """Input rows handled per exchange step: as many as the scratch budget allows (all of them when the tensor is
    small: one exchange, today's out-of-place behaviour), at least one."""
    budget = STREAM_BUDGET_BYTES if budget_bytes is None else budget_bytes
    unit = max(L * L * M, L * M * M, M * M * M) * elem_bytes
    ni = int(max(1, min(il_max, budget // (6 * unit))))
    if il_max >= 4:
        ni = min(ni, -(-il_max // 4))                      # at least four steps: the exchange has products to hide under
    return ni


def transform_two_body_rows(rows, C, C_tilde=None, rank=0, world=1, group=None, engine=HipEngine, in_part=None,
                            chunk_rows=None, out=None):
    """The four-index transform of a tensor sharded over ONE of its two leading indices, result sharded over the
    OTHER one, with everything but the input rows and the result rows O(l^3) -- the layout behind
    ``ShardedDeviceModule`` (``change_basis`` of a sharded basis set, basis_set.py:374-382, and the per-step
    functional call of a solver, system.py:222-225, which keeps ``u`` resident).

    ``rows[i, j, c, d]``: this rank's rows ``i`` of the sharded index I (split ``in_part``, balanced by default),
    the other leading index J whole -- ``u[a_lo + i, j]`` for a leading-index sharding, ``u[j, b_lo + i]`` for a
    second-index sharding (``ShardedTensor4.rows``).  The transform is symmetric under swapping its two leading
    index pairs (both are contracted with ``C_tilde``), so the same code serves both.  Returns
    ``out_rows[j', i', r, s]`` for this rank's rows ``j'`` of the transformed index J' (balanced split of M), I'
    whole: the sharded index has flipped.

    Per step of ``chunk_rows`` input rows (three l^3-sized temporaries per row):
      d, c   ``t2[i][j][r, s] = C[c, r] rows[i][j][c, d] C[d, s]``                        (slab-local)
      J      ``W[j', i, (r,s)] = Ct[j', j] t2[i][j][(r,s)]``                               (J is whole on this rank)
      send   the rows ``j'`` of W that rank g owns go to g -- W is stored ``[j'][i][(r,s)]``, so a peer's share is one
             contiguous block, no packing -- and what the peers computed for OUR ``j'`` lands in
             ``R[j'_loc][i_global][(r,s)]``                                                 (ONE all-to-all per step)
    After the last step row ``j'_loc`` of R is the whole (L, M*M) matrix of that result row, and
      I      ``out[j'_loc][i', (r,s)] = Ct[i', i] R[j'_loc][i, (r,s)]``
    runs row by row INSIDE the buffer: result rows are packed from its start, received rows sit behind a gap of
    one row (and of the growth ``(M - L) M^2`` per row when M > L), so the product of row p never reaches a
    received row that is still to be read.  Every sum runs over the same index in the same order as in the
    out-of-place layouts (d, c, the whole leading index, the sharded one): bit-identical results.

    Peak memory: input rows + ``rows_buffer_elems`` (= result rows + one row when L = M) + 6 chunk_rows l^3 (two send
    blocks and two receive stagings: the exchange of a step overlaps the products of the next).
    ``out`` may supply the (flat) buffer to reuse it across steps of a time loop; the result is a view of it."""
    Ct = _bra(C, C_tilde)
    L, M = C.shape
    ipart, jpart = (in_part or SlabPartition(L, world)), SlabPartition(M, world)
    il, jl = ipart.count(rank), jpart.count(rank)
    if tuple(rows.shape) != (il, L, L, L) or not rows.is_contiguous():
        raise ValueError(f"rank {rank}: rows of shape {tuple(rows.shape)}, expected contiguous {(il, L, L, L)}")
    dt = kernels.result_dtype(rows, C, Ct)
    C, Ct = C.to(dt).contiguous(), Ct.to(dt).contiguous()
    CT = C.transpose(0, 1).contiguous()
    dev = rows.device
    MM = M * M
    width = 2 if dt.is_complex else 1
    es = 16 if dt.is_complex else 8
    nbuf = rows_buffer_elems(L, M, jl)
    if out is None:
        buf = torch.empty(nbuf, dtype=dt, device=dev)
    else:
        if out.dtype != dt or out.numel() < nbuf or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"`out` must be a contiguous {dt} buffer of at least {nbuf} elements")
        buf = out.reshape(-1)
    r0 = jl * max(M - L, 0) * MM + L * MM                  # element offset of received row 0
    R = buf[r0: r0 + jl * L * MM].view(jl, L, MM)
    il_max = max(ipart.count(g) for g in range(world))
    ni = stream_chunk_rows(L, M, il_max, es) if chunk_rows is None else max(1, min(int(chunk_rows), il_max))
    nsteps = -(-il_max // ni)
    t1 = torch.empty(ni * L * L * M, dtype=dt, device=dev)
    t2 = torch.empty(ni * L * MM, dtype=dt, device=dev)
    # The exchange of step t runs (asynchronously, on the backend's own stream) under the products of step t + 1: two send
    # blocks and two receive stagings, the received rows are put in place one step late.  (On RCCL the C-ABI form does
    # the same without staging: qs_transform_two_body_sharded_rows.)
    nbuf_x = 2 if (world > 1 and nsteps > 1) else 1
    Ws = [torch.empty(M * ni * MM, dtype=dt, device=dev) for _ in range(nbuf_x)]
    stages = []
    if world > 1:
        n_stage = jl * sum(min(ni, ipart.count(g)) for g in range(world)) * MM * width
        stages = [torch.empty(n_stage, dtype=torch.float64, device=dev) for _ in range(nbuf_x)]

    def settle(pending):
        """The received blocks of an earlier step into place: R[j'_loc][i_global][(r,s)]."""
        work, recv, out_splits, counts, i0 = pending
        work.wait()
        off = 0
        for g in range(world):
            if out_splits[g]:
                blk = recv[off: off + out_splits[g]]
                if width == 2:
                    blk = torch.view_as_complex(blk.reshape(-1, 2))
                g0 = ipart.starts[g] + i0
                R[:, g0:g0 + counts[g]].copy_(blk.reshape(jl, counts[g], MM))
            off += out_splits[g]

    pending = None
    for t in range(nsteps):
        i0 = t * ni
        n = max(0, min(ni, il - i0))
        W = Ws[t % nbuf_x]
        if n > 0:
            src = rows[i0:i0 + n]
            if src.dtype != dt:
                src = src.to(dt)                           # (real u, complex C: cast chunk-wise, never the slab)
            # d:  t1[(i,j,c), s] = rows[(i,j,c), d] C[d, s]
            engine.gemm_strided(dt, src, C, t1, n * L * L, M, L, L, M, M)
            # c:  t2[(i,j)][r, s] = CT[r, c] t1[(i,j)][c, s]
            engine.gemm_strided(dt, CT, t1, t2, M, M, L, L, M, M, batch=n * L, sa=0, sb=L * M, sc=MM)
            # J:  W[j', i, (r,s)] = Ct[j', j] t2[i][j, (r,s)]      one product per row i, rows of W n*MM apart
            engine.gemm_strided(dt, Ct, t2, W, M, MM, L, L, MM, n * MM, batch=n, sa=0, sb=L * MM, sc=MM)
        counts = [max(0, min(ni, ipart.count(g) - i0)) for g in range(world)]   # rows every rank brings to this step
        if world == 1:
            if n > 0:
                R[:, i0:i0 + n].copy_(W[:M * n * MM].view(M, n, MM))
            continue
        in_splits = [jpart.count(g) * n * MM * width for g in range(world)]
        out_splits = [jl * counts[g] * MM * width for g in range(world)]
        recv = stages[t % nbuf_x][:sum(out_splits)]
        work = dist.all_to_all_single(recv, _as_real_flat(W[:M * n * MM]), out_splits, in_splits, group=group, async_op=True)
        if pending is not None:
            settle(pending)                                # step t - 1: done by now, or waited for here
        pending = (work, recv, out_splits, counts, i0)
    if pending is not None:
        settle(pending)
    del t1, t2, Ws, stages
    # I:  out[p][i', (r,s)] = Ct[i', i] R[p][i, (r,s)], packed from the start of the buffer
    for p in range(jl):
        engine.gemm_strided(dt, Ct, buf, buf, M, MM, L, L, MM, MM, b_off=r0 + p * L * MM, c_off=p * M * MM)
    return buf[:jl * M * MM].view(jl, M, M, M)


# ---------------------------------------------------------------------------
# First consumers of a p-sharded u: Fock matrix and reference energy
# (SURVEY 8f #2).  Every term needs u[p, ...] for ONE leading index p, so each
# rank works on its slab and the node exchanges l*l (Fock rows) or one number
# (energy) -- u stays sharded end to end.
# ---------------------------------------------------------------------------


def fock_rows(h, u_slab, n_occ, p_lo, spin_orbitals=False):
    """Rows ``p_lo : p_lo + u_slab.shape[0]`` of the Fock matrix.

    spatial orbitals (closed shell, spatial_orbital_system.py:152-190):
        f_pq = h_pq + 2 u_piqi - u_piiq
    spin orbitals with anti-symmetrised u (general_orbital_system.py:123-159):
        f_pq = h_pq + u_piqi
    summed over the ``n_occ`` occupied orbitals i.  ``u_slab = u[p_lo:p_hi]``.
    """
    pc = u_slab.shape[0]
    f = h[p_lo:p_lo + pc].clone()
    o = slice(0, n_occ)
    direct = torch.einsum("piqi->pq", u_slab[:, o, :, o])
    if spin_orbitals:
        return f + direct
    return f + 2 * direct - torch.einsum("piiq->pq", u_slab[:, o, o, :])


def construct_fock_matrix_sharded(h, u_slab, n_occ, rank=0, world=1, spin_orbitals=False, group=None,
                                  part=None):
    """Full Fock matrix on every rank from a p-sharded ``u``: slab-local rows,
    then one all-gather of l*l numbers."""
    l = h.shape[0]
    part = part or SlabPartition(l, world)
    lo, hi = part.bounds(rank)
    if tuple(u_slab.shape[:1]) != (hi - lo,):
        raise ValueError(f"rank {rank}: slab has {u_slab.shape[0]} rows, expected {hi - lo}")
    rows = fock_rows(h, u_slab, n_occ, lo, spin_orbitals)
    if world == 1:
        return rows
    width = 2 if rows.is_complex() else 1
    biggest = max(part.count(r) for r in range(world)) * l * width
    send = torch.zeros(biggest, dtype=torch.float64, device=rows.device)
    flat = _as_real_flat(rows)
    send[: flat.numel()] = flat
    recv = torch.empty(world * biggest, dtype=torch.float64, device=rows.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    f = torch.empty((l, l), dtype=rows.dtype, device=rows.device)
    ff = _as_real_flat(f)
    for r in range(world):
        r_lo, r_hi = part.bounds(r)
        n = (r_hi - r_lo) * l * width
        ff[r_lo * l * width: r_lo * l * width + n] = recv[r * biggest: r * biggest + n]
    return f


def reference_energy_partial(h, u_slab, n_occ, p_lo, spin_orbitals=False):
    """This slab's share of the reference-determinant energy (no nuclear term): the sum over the occupied
    rows i in [p_lo, p_lo + rows) of  2 h_ii + 2 u_ijij - u_ijji  (spatial orbitals) or  h_ii + 1/2 u_ijij
    (spin orbitals, anti-symmetrised u); the shares of all slabs add up to the energy."""
    lo, hi = p_lo, p_lo + u_slab.shape[0]
    i_lo, i_hi = min(lo, n_occ), min(hi, n_occ)          # occupied rows of this slab
    o = slice(0, n_occ)
    part = torch.zeros((), dtype=u_slab.dtype, device=u_slab.device)
    if i_hi > i_lo:
        blk = u_slab[i_lo - lo:i_hi - lo, o, :, :][:, :, i_lo:i_hi, o]    # u[i, j, i', j'], i' in the same rows
        coul = torch.einsum("ijij->", blk)
        hd = torch.diagonal(h)[i_lo:i_hi].sum().to(u_slab.dtype)
        if spin_orbitals:
            part = hd + 0.5 * coul
        else:
            part = 2 * hd + 2 * coul - torch.einsum("ijji->", u_slab[i_lo - lo:i_hi - lo, o, o, :][:, :, :, i_lo:i_hi])
    return part


def reference_energy_sharded(h, u_slab, n_occ, rank=0, world=1, spin_orbitals=False,
                             nuclear_repulsion_energy=0.0, group=None):
    """Reference-determinant energy from a p-sharded ``u``.

    spatial (spatial_orbital_system.py:106-150): 2 h_ii + 2 u_ijij - u_ijji + E_nuc
    spin orbitals (general_orbital_system.py:75-121): h_ii + 1/2 u_ijij + E_nuc
    The leading index i of ``u`` is the sharded one: each rank sums its occupied
    rows, one all-reduce of a single number closes the sum.
    """
    l = h.shape[0]
    lo, hi = SlabPartition(l, world).bounds(rank)
    if u_slab.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: slab has {u_slab.shape[0]} rows, expected {hi - lo}")
    part = reference_energy_partial(h, u_slab, n_occ, lo, spin_orbitals)
    if world > 1:
        buf = torch.view_as_real(part.to(torch.complex128).reshape(1)).reshape(-1).contiguous()
        dist.all_reduce(buf, group=group)
        total = torch.view_as_complex(buf.reshape(1, 2))[0]
        part = total if u_slab.is_complex() else total.real
    return part + nuclear_repulsion_energy


def fock_partial_second_index(u_bslab, n_occ, b_lo, spin_orbitals=False):
    """This rank's share of the two-body part of the Fock matrix when ``u`` is sharded over its SECOND
    index (``u_bslab = u[:, b_lo:b_hi]``): the sums over the occupied i run over the second index,
    so each rank adds the terms of its own occupied i and the shares add up (one all-reduce of l*l)."""
    L = u_bslab.shape[0]
    i_lo, i_hi = min(b_lo, n_occ), min(b_lo + u_bslab.shape[1], n_occ)
    f = torch.zeros((L, L), dtype=u_bslab.dtype, device=u_bslab.device)
    if i_hi > i_lo:
        loc = slice(i_lo - b_lo, i_hi - b_lo)
        direct = torch.einsum("piqi->pq", u_bslab[:, loc, :, i_lo:i_hi])
        if spin_orbitals:
            return f + direct
        return f + 2 * direct - torch.einsum("piiq->pq", u_bslab[:, loc, i_lo:i_hi, :])
    return f


def reference_energy_partial_second_index(h, u_bslab, n_occ, b_lo, spin_orbitals=False):
    """Share of the reference energy held by a second-index slab ``u[:, b_lo:b_hi]``: the terms
    u_ijij / u_ijji whose j lies in the slab, plus the one-body terms h_jj of those j."""
    j_lo, j_hi = min(b_lo, n_occ), min(b_lo + u_bslab.shape[1], n_occ)
    o = slice(0, n_occ)
    part = torch.zeros((), dtype=u_bslab.dtype, device=u_bslab.device)
    if j_hi > j_lo:
        loc = slice(j_lo - b_lo, j_hi - b_lo)
        coul = torch.einsum("ijij->", u_bslab[o, loc, o, j_lo:j_hi])
        hd = torch.diagonal(h)[j_lo:j_hi].sum().to(u_bslab.dtype)
        if spin_orbitals:
            part = hd + 0.5 * coul
        else:
            part = 2 * hd + 2 * coul - torch.einsum("ijji->", u_bslab[o, loc, j_lo:j_hi, o])
    return part


def all_reduce_sum(t, world, group=None):
    """Sum of a (small) tensor over the ranks; complex values travel as interleaved pairs."""
    if world == 1:
        return t
    if t.is_complex():
        buf = torch.view_as_real(t.contiguous()).contiguous()
        dist.all_reduce(buf, group=group)
        return torch.view_as_complex(buf)
    buf = t.contiguous()
    dist.all_reduce(buf, group=group)
    return buf
