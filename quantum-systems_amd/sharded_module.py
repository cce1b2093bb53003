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
    """One rank's share of an (l, l, l, l) tensor: ``local = t[lo:hi]`` (``axis = 0``) or
    ``local = t[:, lo:hi]`` (``axis = 1``), contiguous; the rows follow ``part`` (the balanced
    ``sharded.SlabPartition(l, world)`` unless a producer says otherwise: spin doubling keeps every
    rank's rows together, which doubles the offsets).  ``shape`` / ``dtype`` report the WHOLE tensor, so the
    shape assertions of the ``BasisSet`` setters (basis_set.py:93, :103, :113) read the same."""

    ndim = 4

    def __init__(self, local, l, axis, rank, world, group=None, part=None):
        if axis not in (0, 1):
            raise ValueError("a rank-4 tensor is sharded over its first or its second index")
        self.l, self.axis, self.rank, self.world, self.group = int(l), axis, int(rank), int(world), group
        self.part = part or sharded.SlabPartition(self.l, self.world)
        if (self.part.n, self.part.world) != (self.l, self.world):
            raise ValueError("the partition does not describe this tensor")
        self.lo, self.hi = self.part.bounds(self.rank)
        want = (self.hi - self.lo, self.l, self.l, self.l) if axis == 0 else (self.l, self.hi - self.lo, self.l, self.l)
        if tuple(local.shape) != want:
            raise ValueError(f"rank {rank}: local block {tuple(local.shape)}, expected {want}")
        self.local = local.contiguous()

    # -- what the BasisSet / system layers ask of an array
    @property
    def shape(self):
        return (self.l,) * 4

    @property
    def dtype(self):
        return self.local.dtype

    @property
    def device(self):
        return self.local.device

    def dim(self):
        return 4

    def is_complex(self):
        return self.local.is_complex()

    def _like(self, local, axis=None):
        return ShardedTensor4(local, self.l, self.axis if axis is None else axis,
                              self.rank, self.world, self.group, self.part)

    def astype(self, dtype):
        return self._like(self.local.to(as_torch_dtype(dtype)))

    def copy(self):
        return self._like(self.local.clone())

    def conj(self):
        return self._like(self.local.conj().resolve_conj())

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
            return self._like(op(self.local, other.local))
        if isinstance(other, torch.Tensor) and other.dim() > 0:
            raise TypeError("a sharded tensor combines with scalars and equally sharded tensors only")
        return self._like(op(self.local, other))

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
        return self._like(-self.local)

    # -- data movement (each is ONE collective)
    def gather(self):
        """The whole tensor on every rank (all-gather).  For checks and small tensors: at the sizes the
        sharding exists for it does not fit one device."""
        if self.world == 1:
            return wrap(self.local)
        if self.axis == 0:
            return wrap(sharded.all_gather_slabs(self.local, self.l, self.rank, self.world, self.group,
                                                 part=self.part))
        # second-index slabs: gather the (q-major) transposes, then put the axes back
        qmajor = self.local.transpose(0, 1).contiguous()
        full = sharded.all_gather_slabs(qmajor, self.l, self.rank, self.world, self.group, part=self.part)
        return wrap(full.transpose(0, 1).contiguous())

    def reshard(self, axis):
        """The same tensor sharded over the other leading index (one all-to-all of (G-1)/G^2 of the
        tensor per rank)."""
        if axis == self.axis:
            return self
        if self.world == 1:
            return self._like(self.local, axis=axis)
        l, part = self.l, self.part
        mine = self.hi - self.lo
        width = 2 if self.local.is_complex() else 1
        blk = l * l
        # the block (my rows of the sharded index, rank g's rows of the other index) goes to rank g
        if self.axis == 0:
            pieces = [self.local[:, part.bounds(g)[0]:part.bounds(g)[1]] for g in range(self.world)]
        else:
            pieces = [self.local[part.bounds(g)[0]:part.bounds(g)[1]] for g in range(self.world)]
        send = sharded._as_real_flat(torch.cat([p.reshape(-1) for p in pieces]))
        in_splits = [mine * part.count(g) * blk * width for g in range(self.world)]
        recv = torch.empty(sum(in_splits), dtype=torch.float64, device=send.device)
        dist.all_to_all_single(recv, send, in_splits, in_splits, group=self.group)
        out, off = [], 0
        for g in range(self.world):
            piece = recv[off: off + in_splits[g]]
            off += in_splits[g]
            if width == 2:
                piece = torch.view_as_complex(piece.reshape(-1, 2))
            cnt = part.count(g)
            # the block arrives as [sender's index][my index][r][s] going 0 -> 1, [my index][sender's index] going 1 -> 0
            if self.axis == 0:
                out.append(piece.reshape(cnt, mine, l, l))          # [a in slab(g)][q mine]
            else:
                out.append(piece.reshape(mine, cnt, l, l))          # [p mine][b in slab(g)]
        local = torch.cat(out, dim=0 if self.axis == 0 else 1)
        return self._like(local.contiguous(), axis=axis)

    def __repr__(self):
        return (f"ShardedTensor4(l={self.l}, axis={self.axis}, rank {self.rank}/{self.world}, "
                f"rows [{self.lo}, {self.hi}), {self.dtype})")


class ShardedDeviceModule(DeviceModule):
    """``DeviceModule`` for one rank of a node-wide job: small arrays replicated on this rank's
    GPU, rank-4 tensors sharded (``shard``).  ``engine`` is the slab-local compute back end -- the HIP
    library in the product; the CPU tests of the partitioning / exchange logic pass an oracle-backed
    stand-in together with ``device="cpu"`` (gloo)."""

    name = "quantum_systems_amd.sharded_hip"

    def __init__(self, rank=None, world=None, group=None, device=None, engine=None):
        super().__init__(device)
        if rank is None or world is None:
            if not dist.is_initialized():
                raise RuntimeError("pass rank and world, or initialise torch.distributed first")
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.engine = sharded.HipEngine if engine is None else engine

    def shard(self, arr, axis=0):
        """This rank's slab of a whole (l,l,l,l) array (NumPy or torch, host or device): only the slab
        is uploaded / kept.  A ``ShardedTensor4`` passes through (resharded if asked)."""
        if isinstance(arr, ShardedTensor4):
            return arr.reshard(axis)
        l = arr.shape[0]
        if tuple(arr.shape) != (l, l, l, l):
            raise ValueError("only (l,l,l,l) tensors are sharded")
        lo, hi = sharded.SlabPartition(l, self.world).bounds(self.rank)
        part = arr[lo:hi] if axis == 0 else arr[:, lo:hi]
        if isinstance(part, torch.Tensor):
            local = part.to(self.device).contiguous()
        else:
            local = torch.from_numpy(_np.ascontiguousarray(part)).to(self.device)
        return ShardedTensor4(local, l, axis, self.rank, self.world, self.group)

    def from_local(self, local, l, axis=0):
        """Wrap a slab this rank produced itself (a generator that never builds the whole tensor)."""
        return ShardedTensor4(local.to(self.device), l, axis, self.rank, self.world, self.group)

    def zeros_like(self, a, dtype=None):
        if isinstance(a, ShardedTensor4):
            return a._like(torch.zeros_like(a.local, dtype=as_torch_dtype(dtype)))
        return super().zeros_like(a, dtype=dtype)


def is_sharded_module(mod):
    return isinstance(mod, ShardedDeviceModule)


def is_sharded(arr):
    return isinstance(arr, ShardedTensor4)
