"""The sharded array module: pass it wherever the reference takes ``np=`` and the rank-4 tensors
of a ``BasisSet`` live as one slab per GPU instead of one array.

The reference's only seam is array-module injection (quantum_systems/basis_set.py:32-38, :268-296),
and it has no notion of more than one device.  With

    mod = ShardedDeviceModule(rank, world)            # one process per GPU, torch.distributed set up
    system.change_module(mod)                         # or BasisSet(l, dim, np=mod)

every small array (``h``, ``s``, ``position``, spin matrices, ``spf``: O(l^2)) is replicated on each
rank's GPU exactly as with the single-GPU module ``hip``, while ``u`` (and ``spin_2_tb``) become
``ShardedTensor4`` objects: this rank's contiguous slab of the leading index (``u[p_lo:p_hi]``) or of
the second index (``u[:, q_lo:q_hi]``).  ``BasisSet.change_basis``, ``change_to_general_orbital_basis``,
``anti_symmetrize_two_body_elements`` and the systems' Fock matrix / reference energy dispatch on that
type (sharded_basis.py): slab-local kernels plus ONE all-to-all per four-index transform, l*l numbers
for a Fock matrix, one number for an energy.  BASELINE.json configs[3] (SpatialOrbitalSystem l = 256 ->
GeneralOrbitalSystem with 512 spin orbitals, 1.1 TB) runs as named on 8 GPUs this way: no rank ever
holds, copies or gathers the whole tensor.
"""

import numpy as _np
import torch
import torch.distributed as dist

from . import sharded
from .array_module import DeviceModule, as_torch_dtype, wrap


class ShardedTensor4:
    """One rank's share of an (l, l, l, l) tensor ``t``: the rows ``lo:hi`` of ONE of its two leading indices
    (``axis``), with the other three indices whole.  The rows follow ``part`` (the balanced
    ``sharded.SlabPartition(l, world)`` unless a producer says otherwise: spin doubling keeps every rank's rows
    together, which doubles the offsets).  ``shape`` / ``dtype`` report the WHOLE tensor, so the shape assertions of
    the ``BasisSet`` setters (basis_set.py:93, :103, :113) read the same.

    Storage: ``rows`` is contiguous with the SHARDED index leading, whichever it is --
    ``rows[i, j] = t[lo + i, j]`` for ``axis = 0`` and ``rows[i, j] = t[j, lo + i]`` for ``axis = 1`` -- so both
    shardings are a stack of whole (l, l, l) row blocks.  The four-index transform is symmetric under swapping its
    two leading index pairs, which makes ONE streamed algorithm (``sharded.transform_two_body_rows``) serve both:
    rows of one leading index in, rows of the other one out.  ``local`` is the logical view
    (``t[lo:hi]`` or ``t[:, lo:hi]``; the latter is a transposed, non-contiguous view of ``rows``)."""

    ndim = 4

    def __init__(self, local, l, axis, rank, world, group=None, part=None, rows=None):
        if axis not in (0, 1):
            raise ValueError("a rank-4 tensor is sharded over its first or its second index")
        self.l, self.axis, self.rank, self.world, self.group = int(l), axis, int(rank), int(world), group
        self.part = part or sharded.SlabPartition(self.l, self.world)
        if (self.part.n, self.part.world) != (self.l, self.world):
            raise ValueError("the partition does not describe this tensor")
        self.lo, self.hi = self.part.bounds(self.rank)
        cnt = self.hi - self.lo
        if rows is None:
            want = (cnt, self.l, self.l, self.l) if axis == 0 else (self.l, cnt, self.l, self.l)
            if tuple(local.shape) != want:
                raise ValueError(f"rank {rank}: local block {tuple(local.shape)}, expected {want}")
            rows = local if axis == 0 else local.transpose(0, 1)
        elif tuple(rows.shape) != (cnt, self.l, self.l, self.l):
            raise ValueError(f"rank {rank}: rows {tuple(rows.shape)}, expected {(cnt, self.l, self.l, self.l)}")
        self.rows = rows.contiguous()

    @property
    def local(self):
        """``t[lo:hi]`` (axis 0) or ``t[:, lo:hi]`` (axis 1, a transposed view of ``rows``)."""
        return self.rows if self.axis == 0 else self.rows.transpose(0, 1)

    # -- what the BasisSet / system layers ask of an array
    @property
    def shape(self):
        return (self.l,) * 4

    @property
    def dtype(self):
        return self.rows.dtype

    @property
    def device(self):
        return self.rows.device

    def dim(self):
        return 4

    def is_complex(self):
        return self.rows.is_complex()

    def _like(self, rows, axis=None):
        """Same sharding, other numbers: ``rows`` in the storage order (sharded index leading)."""
        return ShardedTensor4(None, self.l, self.axis if axis is None else axis,
                              self.rank, self.world, self.group, self.part, rows=rows)

    def astype(self, dtype):
        return self._like(self.rows.to(as_torch_dtype(dtype)))

    def copy(self):
        return self._like(self.rows.clone())

    def conj(self):
        return self._like(self.rows.conj().resolve_conj())

    def __deepcopy__(self, memo):
        out = self.copy()
        memo[id(self)] = out
        return out

    # element-wise algebra of the time-dependent Hamiltonian (system.py:206-215: u_0 + sum of operator terms;
    # operator.py:193-196 scales u)
    def _binary(self, other, op):
        if isinstance(other, ShardedTensor4):
            if (other.l, other.axis, other.world) != (self.l, self.axis, self.world):
                raise ValueError("operands are sharded differently (reshard one of them first)")
            return self._like(op(self.rows, other.rows))
        if isinstance(other, torch.Tensor) and other.dim() > 0:
            raise TypeError("a sharded tensor combines with scalars and equally sharded tensors only")
        return self._like(op(self.rows, other))

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b)

    def __radd__(self, other):
        return self._binary(other, lambda a, b: b + a)

    def __sub__(self, other):
        return self._binary(other, lambda a, b: a - b)

    def __mul__(self, other):
        return self._binary(other, lambda a, b: a * b)

    def __rmul__(self, other):
        return self._binary(other, lambda a, b: b * a)

    def __neg__(self):
        return self._like(-self.rows)

    # -- data movement (each is ONE collective)
    def gather(self):
        """The whole tensor on every rank (all-gather).  For checks and small tensors: at the sizes the
        sharding exists for it does not fit one device."""
        if self.world == 1:
            full = self.rows
        else:
            full = sharded.all_gather_slabs(self.rows, self.l, self.rank, self.world, self.group, part=self.part)
        # the gathered rows have the sharded index leading: put the axes back for a second-index sharding
        return wrap(full if self.axis == 0 else full.transpose(0, 1).contiguous())

    def reshard(self, axis):
        """The same tensor sharded over the other leading index (one all-to-all of (G-1)/G^2 of the
        tensor per rank): a distributed transposition of the two leading indices, the same code both ways."""
        if axis == self.axis:
            return self
        l, part = self.l, self.part
        if self.world == 1:
            return self._like(self.rows.transpose(0, 1).contiguous(), axis=axis)
        mine = self.hi - self.lo
        width = 2 if self.rows.is_complex() else 1
        blk = l * l
        # the block (my rows, rank g's rows of the other index) goes to rank g, other index leading
        pieces = [self.rows[:, part.bounds(g)[0]:part.bounds(g)[1]].transpose(0, 1).reshape(-1)
                  for g in range(self.world)]
        send = sharded._as_real_flat(torch.cat(pieces))
        splits = [mine * part.count(g) * blk * width for g in range(self.world)]
        recv = torch.empty(sum(splits), dtype=torch.float64, device=send.device)
        dist.all_to_all_single(recv, send, splits, splits, group=self.group)
        out, off = [], 0
        for g in range(self.world):
            piece = recv[off: off + splits[g]]
            off += splits[g]
            if width == 2:
                piece = torch.view_as_complex(piece.reshape(-1, 2))
            out.append(piece.reshape(mine, part.count(g), l, l))    # [my new rows][sender's rows of the old index]
        return self._like(torch.cat(out, dim=1).contiguous(), axis=axis)

    def __repr__(self):
        return (f"ShardedTensor4(l={self.l}, axis={self.axis}, rank {self.rank}/{self.world}, "
                f"rows [{self.lo}, {self.hi}), {self.dtype})")


class ShardedDeviceModule(DeviceModule):
    """``DeviceModule`` for one rank of a node-wide job: small arrays replicated on this rank's
    GPU, rank-4 tensors sharded (``shard``).  ``engine`` is the slab-local compute back end -- the HIP
    library in the product; the CPU tests of the partitioning / exchange logic pass an oracle-backed
    stand-in together with ``device="cpu"`` (gloo)."""

    name = "quantum_systems_amd.sharded_hip"

    def __init__(self, rank=None, world=None, group=None, device=None, engine=None, exchange="auto"):
        super().__init__(device)
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("pass rank and world, or initialise torch.distributed first")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.engine = sharded.HipEngine if engine is None else engine
        self.exchange = exchange
        self._rccl = None

    def rccl(self):
        """The C ABI's own RCCL communicator for this module's ranks (``kernels.RcclComm``), or ``None`` when the
        exchange goes through ``torch.distributed`` collectives.

        ``exchange="auto"`` (default): RCCL when the process group's backend is ``nccl`` and the engine is the HIP
        library -- the sharded transform is then ONE C-ABI call per tensor (``qs_transform_two_body_sharded_rows``:
        local products, grouped send / receive on a second stream overlapped with them, closing products) instead of
        Python-driven products around ``all_to_all_single``; ``"torch"`` forces the collectives (``gloo`` runs, CPU
        tests), ``"rccl"`` insists on the communicator.  Created on first use: rank 0 draws the unique id, the group
        broadcasts it."""
        if self.exchange == "torch":
            return None
        if self._rccl is None:
            on_nccl = dist.is_initialized() and dist.get_backend(self.group) == "nccl"
            if self.exchange == "auto" and not (on_nccl and self.engine is sharded.HipEngine):
                return None
            from .kernels import RcclComm

            box = [RcclComm.unique_id() if self.rank == 0 else None]
            if self.world > 1:
                if dist.get_backend(self.group) == "nccl":
                    with torch.cuda.device(self.device):
                        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0,
                                                   group=self.group)
                else:
                    dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0,
                                               group=self.group, device=torch.device("cpu"))
            with torch.cuda.device(self.device):
                self._rccl = RcclComm(self.rank, self.world, box[0])
        return self._rccl

    def shard(self, arr, axis=0):
        """This rank's slab of a whole (l,l,l,l) array (NumPy or torch, host or device): only the slab
        is uploaded / kept.  A ``ShardedTensor4`` passes through (resharded if asked)."""
        if isinstance(arr, ShardedTensor4):
            return arr.reshard(axis)
        l = arr.shape[0]
        if tuple(arr.shape) != (l, l, l, l):
            raise ValueError("only (l,l,l,l) tensors are sharded")
        lo, hi = sharded.SlabPartition(l, self.world).bounds(self.rank)
        if isinstance(arr, torch.Tensor):
            part = arr[lo:hi] if axis == 0 else arr[:, lo:hi].transpose(0, 1)
            rows = part.to(self.device).contiguous()
        else:
            part = arr[lo:hi] if axis == 0 else arr[:, lo:hi].transpose(1, 0, 2, 3)
            rows = torch.from_numpy(_np.ascontiguousarray(part)).to(self.device)
        return ShardedTensor4(None, l, axis, self.rank, self.world, self.group, rows=rows)

    def from_local(self, local, l, axis=0):
        """Wrap a slab this rank produced itself (a generator that never builds the whole tensor): ``t[lo:hi]`` for
        ``axis = 0``, ``t[:, lo:hi]`` for ``axis = 1``."""
        return ShardedTensor4(local.to(self.device), l, axis, self.rank, self.world, self.group)

    def from_rows(self, rows, l, axis=0):
        """The same from the storage order (the sharded index leading, see ``ShardedTensor4``): no copy."""
        return ShardedTensor4(None, l, axis, self.rank, self.world, self.group, rows=rows.to(self.device))

    def zeros_like(self, a, dtype=None):
        if isinstance(a, ShardedTensor4):
            return a._like(torch.zeros_like(a.rows, dtype=as_torch_dtype(dtype)))
        return super().zeros_like(a, dtype=dtype)


def is_sharded_module(mod):
    return isinstance(mod, ShardedDeviceModule)


def is_sharded(arr):
    return isinstance(arr, ShardedTensor4)
