"""The device array module: the object to pass wherever the reference takes
``np=`` (``BasisSet(l, dim, np=...)``, ``system.change_module(np)``).

The reference's seam is array-module injection (basis_set.py:32-38, :268-296).
A raw ``torch`` module cannot be injected there: callers use
``np.asarray``, ``np.tensordot(..., axes=)``, ``arr.transpose(0, 1, 3, 2)``,
``arr.astype``, ``arr.copy()``.  This module offers that NumPy-flavoured surface
over tensors resident in HBM:

* arrays are ``DeviceArray`` -- a ``torch.Tensor`` subclass adding ``copy``,
  ``astype`` and NumPy's permutation form of ``transpose``;
* ``dot`` / ``tensordot`` / ``matmul`` of fp64 / complex128 operands run on the
  HIP GEMM of this package (``qs_matmul``);
* the remaining entries (``zeros``, ``eye``, ``kron``, ``einsum``, ``trace``,
  ``random``...) are thin torch calls: allocation and O(l^2) bookkeeping, not
  the transform path.

``hip`` is the ready-made instance for ``cuda:current``.
"""

import numpy as _np
import torch

from . import kernels

_TORCH_DTYPES = {
    _np.dtype("float64"): torch.float64,
    _np.dtype("complex128"): torch.complex128,
    _np.dtype("float32"): torch.float32,
    _np.dtype("complex64"): torch.complex64,
    _np.dtype("int64"): torch.int64,
    _np.dtype("int32"): torch.int32,
    _np.dtype("bool"): torch.bool,
}


def as_torch_dtype(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    if dtype is complex:
        return torch.complex128
    if dtype is float:
        return torch.float64
    if dtype is int:
        return torch.int64
    return _TORCH_DTYPES[_np.dtype(dtype)]


class DeviceArray(torch.Tensor):
    """torch.Tensor with the handful of ndarray methods the reference's
    callers rely on.  Everything else is inherited."""

    def copy(self):
        return self.clone()

    def astype(self, dtype):
        return self.to(as_torch_dtype(dtype))

    def fill(self, value):
        self.fill_(value)

    def __deepcopy__(self, memo):
        # copy.deepcopy of a basis set / system (copy_basis, copy_system):
        # a device-side clone, still a DeviceArray
        out = self.detach().clone().as_subclass(DeviceArray)
        memo[id(self)] = out
        return out

    def transpose(self, *axes):
        # ndarray.transpose(*perm) for full permutations; torch's two-axis swap
        # otherwise (for 2-D arrays both readings of (1, 0) coincide).
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if len(axes) == 0:
            return self.permute(*reversed(range(self.dim())))
        if len(axes) == self.dim() and self.dim() != 2:
            return self.permute(*axes)
        return super().transpose(*axes)

    def get(self):
        """Host copy as a NumPy array (CuPy's spelling)."""
        return self.detach().cpu().resolve_conj().numpy()

    def __array__(self, dtype=None, copy=None):
        raise TypeError(
            "implicit conversion of a device array to NumPy is not allowed; "
            "use change_module(numpy) or .get()"
        )


def wrap(t):
    return t if isinstance(t, DeviceArray) or not isinstance(t, torch.Tensor) else t.as_subclass(DeviceArray)


class _Random:
    """``np.random`` look-alike on the device generator (Philox).  The stream
    is NOT NumPy's: seeded parity with the reference's RandomBasisSet needs the
    NumPy module (then ``change_module`` to the device)."""

    def __init__(self, owner):
        self._owner = owner

    def seed(self, seed):
        torch.manual_seed(int(seed))
        torch.cuda.manual_seed_all(int(seed))

    def random(self, size=None):
        shape = () if size is None else (size if isinstance(size, (tuple, list)) else (size,))
        out = torch.rand(tuple(shape), dtype=torch.float64, device=self._owner.device)
        return out.item() if size is None else wrap(out)

    def choice(self, seq):
        idx = int(torch.randint(len(seq), (1,)).item())
        return seq[idx]


class DeviceModule:
    """NumPy-shaped namespace whose arrays live on one GPU."""

    name = "quantum_systems_amd.hip"
    float64 = torch.float64
    complex128 = torch.complex128
    ndarray = DeviceArray
    pi = _np.pi

    def __init__(self, device=None):
        self._device = device
        self.random = _Random(self)

    @property
    def device(self):
        if self._device is not None:
            return torch.device(self._device)
        if not torch.cuda.is_available():
            raise RuntimeError("the device array module needs a GPU")
        return torch.device("cuda", torch.cuda.current_device())

    # -- construction --------------------------------------------------
    def asarray(self, a, dtype=None):
        dt = as_torch_dtype(dtype)
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dt) if dt is not None else a.to(self.device)
        elif isinstance(a, (list, tuple)) and len(a) and isinstance(a[0], torch.Tensor):
            t = torch.stack([x.to(self.device) for x in a])
            t = t.to(dt) if dt is not None else t
        else:
            h = _np.asarray(a)
            t = torch.from_numpy(_np.ascontiguousarray(h)).to(self.device)
            t = t.to(dt) if dt is not None else t
        return wrap(t)

    def array(self, a, dtype=None):
        out = self.asarray(a, dtype=dtype)
        return wrap(out.clone()) if isinstance(a, torch.Tensor) else out

    def zeros(self, shape, dtype=None):
        return wrap(torch.zeros(shape, dtype=as_torch_dtype(dtype) or torch.float64, device=self.device))

    def empty(self, shape, dtype=None):
        return wrap(torch.empty(shape, dtype=as_torch_dtype(dtype) or torch.float64, device=self.device))

    def zeros_like(self, a, dtype=None):
        return wrap(torch.zeros_like(a, dtype=as_torch_dtype(dtype)))

    def eye(self, n, dtype=None):
        return wrap(torch.eye(n, dtype=as_torch_dtype(dtype) or torch.float64, device=self.device))

    def arange(self, *args, dtype=None):
        return wrap(torch.arange(*args, dtype=as_torch_dtype(dtype), device=self.device))

    # -- products (HIP GEMM for 2-D fp64 / complex128) -------------------
    @staticmethod
    def _gemm_ok(a, b):
        ok = (torch.float64, torch.complex128)
        return a.is_cuda and b.is_cuda and a.dtype in ok and b.dtype in ok

    def matmul(self, a, b):
        if a.dim() == 2 and b.dim() == 2 and self._gemm_ok(a, b):
            return wrap(kernels.matmul(a, b))
        return wrap(torch.matmul(a, b))

    def dot(self, a, b):
        if a.dim() == 2 and b.dim() == 2:
            return self.matmul(a, b)
        if a.dim() <= 1 or b.dim() <= 1:
            return wrap(torch.tensordot(a, b, dims=([a.dim() - 1], [0]))) if a.dim() and b.dim() else a * b
        return self.tensordot(a, b, axes=([a.dim() - 1], [b.dim() - 2]))

    def tensordot(self, a, b, axes=2):
        """NumPy semantics: free axes of ``a`` then free axes of ``b``."""
        if isinstance(axes, int):
            ax_a = list(range(a.dim() - axes, a.dim()))
            ax_b = list(range(axes))
        else:
            ax_a, ax_b = axes
            ax_a = [ax_a] if isinstance(ax_a, int) else list(ax_a)
            ax_b = [ax_b] if isinstance(ax_b, int) else list(ax_b)
        ax_a = [x % a.dim() for x in ax_a]
        ax_b = [x % b.dim() for x in ax_b]
        free_a = [i for i in range(a.dim()) if i not in ax_a]
        free_b = [i for i in range(b.dim()) if i not in ax_b]
        k = 1
        for i in ax_a:
            k *= a.shape[i]
        am = a.permute(*free_a, *ax_a).reshape(-1, k)
        bm = b.permute(*ax_b, *free_b).reshape(k, -1)
        shape = [a.shape[i] for i in free_a] + [b.shape[i] for i in free_b]
        return wrap(self.matmul(am, bm).reshape(shape))

    # -- O(l^2) bookkeeping ------------------------------------------------
    def kron(self, a, b):
        if not isinstance(b, torch.Tensor):
            b = self.asarray(b)
        dt = torch.promote_types(a.dtype, b.dtype)
        return wrap(torch.kron(a.to(dt).contiguous(), b.to(dt).contiguous()))

    def einsum(self, subscripts, *operands, optimize=None):
        spec = subscripts.replace(" ", "")
        dt = operands[0].dtype
        for o in operands[1:]:
            dt = torch.promote_types(dt, o.dtype)
        return wrap(torch.einsum(spec, *[o.to(dt) for o in operands]))

    def trace(self, a, axis1=0, axis2=1):
        return wrap(torch.diagonal(a, dim1=axis1, dim2=axis2).sum(-1))

    def sum(self, a, axis=None):
        return wrap(a.sum() if axis is None else a.sum(dim=axis))

    def abs(self, a):
        return wrap(torch.abs(a))

    def conj(self, a):
        return wrap(torch.conj(a))

    def allclose(self, a, b, rtol=1e-5, atol=1e-8):
        return bool(torch.allclose(a, b, rtol=rtol, atol=atol))


hip = DeviceModule()


def is_device_module(mod):
    return isinstance(mod, DeviceModule)


def to_host(arr):
    """NumPy copy of a device (or host) array; ``None`` passes through."""
    if arr is None:
        return None
    if isinstance(arr, torch.Tensor):
        return arr.detach().cpu().resolve_conj().numpy()
    return _np.asarray(arr)


def convert(arr, mod):
    """``mod.asarray(arr)`` that also works device -> NumPy
    (basis_set.py:268-270 assumes a plain ``np.asarray`` suffices)."""
    if arr is None:
        return None
    if is_device_module(mod):
        return mod.asarray(arr)
    return mod.asarray(to_host(arr))
