"""Typed Python front of the C ABI: torch tensors in, torch tensors out.

PyTorch is used here for what it is good at on ROCm -- owning device memory
and the current HIP stream.  Every function hands raw device pointers to
``libqs_amd.so``; none of them computes anything with torch ops, and none has a
CPU path: a tensor that is not on a ``cuda`` device is an error.
"""

import torch

from . import _lib
from ._lib import QS_C128, QS_F64, check

_F64 = torch.float64
_C128 = torch.complex128


def dtype_code(dtype):
    if dtype == _F64:
        return QS_F64
    if dtype == _C128:
        return QS_C128
    raise TypeError(f"only float64 and complex128 are supported, got {dtype}")


def _stream():
    """Raw handle of the current device's current stream (``torch.cuda.current_stream().cuda_stream`` without the
    Python object around it: the wrappers below are on the path of every small-basis transform)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _plain(fn):
    """The wrapped function sees device arrays as plain tensors: every tensor method called on a ``torch.Tensor``
    subclass (``DeviceArray``) otherwise goes through ``__torch_function__`` and re-wraps its result -- ~100 of those
    per ``change_basis`` cost more than the kernels of a 55-orbital transform.  Results come back as plain tensors
    (the callers wrap what they hand out)."""
    import functools

    @functools.wraps(fn)
    def inner(*args, **kwargs):
        with torch._C.DisableTorchFunctionSubclass():
            return fn(*args, **kwargs)

    return inner


# Measurement hook (bench.py): when this is a list, every compute call appends the kernels it
# launched (qs_last_dispatch), so a bench line can name what actually ran.  None = off.
dispatch_log = None


def _ran(code, what):
    check(code, what)
    if dispatch_log is not None:
        dispatch_log.append(_lib.load().qs_last_dispatch().decode())


def _dev(t, dtype=None):
    """Contiguous, conjugation-resolved device tensor of the wanted dtype."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            "the transform path runs on the GPU only: move the array to the "
            "device module first (change_module)"
        )
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.is_conj():
        t = t.resolve_conj()
    return t if t.is_contiguous() else t.contiguous()


class _on_device_of:
    """Launch context: makes the device that owns the operands current (the C ABI
    launches on the current device, and ``_stream()`` is that device's current
    stream) and refuses operands that live on different devices."""

    def __init__(self, *tensors):
        dev = None
        for t in tensors:
            if t is None:
                continue
            if dev is None:
                dev = t.device
            elif t.device != dev:
                devs = {x.device for x in tensors if x is not None}
                raise ValueError(f"operands live on different devices: {sorted(str(d) for d in devs)}")
        if dev is None:
            raise ValueError("operands live on different devices: []")
        self.device = dev
        self._guard = None

    def __enter__(self):
        if self.device.index != torch._C._cuda_getDevice():
            self._guard = torch.cuda.device(self.device)
            self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        if self._guard is not None:
            self._guard.__exit__(*exc)
        return False


def _check_out(out, shape, dt, what):
    """A caller-supplied output buffer goes to the kernels as a raw pointer: it
    must be exactly what the kernel will write (dtype, shape, contiguous, on the GPU)."""
    if not isinstance(out, torch.Tensor) or not out.is_cuda:
        raise ValueError(f"{what}: `out` must be a device tensor")
    if out.dtype != dt or tuple(out.shape) != tuple(shape) or not out.is_contiguous():
        raise ValueError(
            f"{what}: `out` must be a contiguous {dt} tensor of shape {tuple(shape)}, got "
            f"{out.dtype} {tuple(out.shape)}{'' if out.is_contiguous() else ' (non-contiguous)'}"
        )
    return out


def result_dtype(*tensors):
    """NumPy-style promotion restricted to {float64, complex128}."""
    out = _F64
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t).__name__}")
        if t.dtype in (torch.complex64, _C128):
            out = _C128
        elif t.dtype not in (_F64, torch.float32, torch.int64, torch.int32):
            raise TypeError(f"unsupported dtype {t.dtype}")
    return out


@_plain
def default_bra(C):
    """``C.conj().T`` materialised (basis_set.py:331-332, 338-339)."""
    if not isinstance(C, torch.Tensor):
        raise TypeError(f"expected a torch.Tensor, got {type(C).__name__}")
    return C.conj().transpose(0, 1).resolve_conj().contiguous()


class Workspace:
    """Grow-only device scratch shared by the transforms of one process, ONE BUFFER PER DEVICE (a process that drives
    several GPUs -- the one-process-eight-devices style -- would otherwise free and re-allocate the scratch of the
    other device on every call).

    The C ABI never allocates; this keeps the buffers alive between calls so a time loop calling the transform every step
    does not hit the allocator.  Calls that share a device's buffer are meant to be issued on ONE stream per device (stream
    order keeps them apart); concurrent streams of one device should call the C ABI with their own workspaces, or use one
    ``TransformPlan`` each."""

    def __init__(self):
        self._bufs = {}

    def get(self, nbytes, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            self._bufs[key] = None  # release before growing
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf

    def release(self, device=None):
        """Drop the scratch of one device (or of all of them)."""
        if device is None:
            self._bufs.clear()
        else:
            device = torch.device(device)
            self._bufs.pop((device.type, device.index if device.index is not None else torch.cuda.current_device()), None)


workspace = Workspace()

# A/B switch for measurements (bench.py --dtype mixed, tests): False restores the route of rounds 1-2 for a real u against
# complex coefficients -- a complex copy of the whole tensor, then the complex transform.
mixed_real_u = True


@_plain
def gemm_raw(dt, A, B, out, m, n, k, lda, ldb, ldc, batch=1, sa=0, sb=0, sc=0,
             accumulate=False, a_off=0, b_off=0, c_off=0):
    """Thin call of ``qs_matmul`` on tensors already prepared by the caller
    (contiguous storage, matching dtype); offsets and strides in elements."""
    lib = _lib.load()
    es = 16 if dt == _C128 else 8
    with _on_device_of(A, B, out):
        _ran(
            lib.qs_matmul(
                dtype_code(dt), A.data_ptr() + a_off * es, B.data_ptr() + b_off * es,
                out.data_ptr() + c_off * es, m, n, k, lda, ldb, ldc, batch, sa, sb, sc,
                1 if accumulate else 0, _stream(),
            ),
            "qs_matmul",
        )
    return out


@_plain
def matmul(A, B, out=None, accumulate=False):
    """Row-major ``A @ B`` for 2-D operands (or a shared 2-D ``A`` against a
    batch ``B`` of shape (batch, k, n)); ``accumulate`` adds into ``out``."""
    dt = result_dtype(A, B)
    A = _dev(A, dt)
    B = _dev(B, dt)
    if A.dim() != 2:
        raise ValueError("A must be 2-D")
    m, k = A.shape
    if B.dim() == 2:
        batch, (kb, n) = 1, B.shape
        oshape = (m, n)
    elif B.dim() == 3:
        batch, kb, n = B.shape
        oshape = (batch, m, n)
    else:
        raise ValueError("B must be 2-D or 3-D")
    if kb != k:
        raise ValueError(f"inner dimensions differ: {k} vs {kb}")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output buffer")
        out = torch.empty(oshape, dtype=dt, device=A.device)
    elif (not isinstance(out, torch.Tensor) or not out.is_cuda or out.dtype != dt
          or out.numel() != batch * m * n or not out.is_contiguous()):
        raise ValueError("bad output buffer")
    return gemm_raw(dt, A, B, out, m, n, k, k, n, n, batch, 0, k * n, m * n, accumulate)


@_plain
def transform_two_body(u, C, C_tilde=None, out=None):
    """out[pqrs] = Ct[pa] Ct[qb] u[abcd] C[cr] C[ds]  (basis_set.py:336-350)."""
    lib = _lib.load()
    if C_tilde is None:
        C_tilde = default_bra(C)
    dt = result_dtype(u, C, C_tilde)
    # a real fp64 tensor against complex coefficients (NumPy's promotion, basis_set.py:341-342; the time-propagation
    # call on a real quantum-dot u) stays real: the d contraction reads it as it is (qs_transform_two_body_mixed)
    mixed = dt == _C128 and isinstance(u, torch.Tensor) and u.dtype == _F64 and mixed_real_u
    u = _dev(u, _F64 if mixed else dt)
    C = _dev(C, dt)
    Ct = _dev(C_tilde, dt)
    L, M = C.shape
    if tuple(u.shape) != (L, L, L, L):
        raise ValueError(f"u has shape {tuple(u.shape)}, C is {tuple(C.shape)}")
    if tuple(Ct.shape) != (M, L):
        raise ValueError(f"C_tilde has shape {tuple(Ct.shape)}, expected {(M, L)}")
    code = dtype_code(dt)
    nbytes = check(lib.qs_transform_two_body_workspace(code, L, M), "workspace query")
    if out is None:
        out = torch.empty((M, M, M, M), dtype=dt, device=u.device)
    else:
        _check_out(out, (M, M, M, M), dt, "transform_two_body")
    with _on_device_of(u, C, Ct, out):
        work = workspace.get(nbytes, u.device)
        if mixed:
            _ran(
                lib.qs_transform_two_body_mixed(
                    u.data_ptr(), C.data_ptr(), Ct.data_ptr(), out.data_ptr(),
                    work.data_ptr(), work.numel(), L, M, _stream(),
                ),
                "qs_transform_two_body_mixed",
            )
        else:
            _ran(
                lib.qs_transform_two_body(
                    code, u.data_ptr(), C.data_ptr(), Ct.data_ptr(), out.data_ptr(),
                    work.data_ptr(), work.numel(), L, M, _stream(),
                ),
                "qs_transform_two_body",
            )
    return out


@_plain
def transform_two_body_(u, C, C_tilde=None):
    """The transform IN PLACE: ``u`` (L,L,L,L; float64 / complex128, contiguous, owning its storage) is overwritten
    and the result (M,M,M,M), M <= L, is returned as a view of the start of its storage.  One L^3 M spare buffer
    instead of workspace + result (qs_transform_two_body_inplace): for callers that drop the old tensor."""
    lib = _lib.load()
    if C_tilde is None:
        C_tilde = default_bra(C)
    dt = result_dtype(u, C, C_tilde)
    if not isinstance(u, torch.Tensor) or not u.is_cuda or u.dtype != dt or not u.is_contiguous():
        raise ValueError("the in-place transform needs a contiguous device tensor that already has the result dtype")
    C, Ct = _dev(C, dt), _dev(C_tilde, dt)
    L, M = C.shape
    if tuple(u.shape) != (L, L, L, L) or tuple(Ct.shape) != (M, L) or M > L:
        raise ValueError(f"u {tuple(u.shape)}, C {tuple(C.shape)}, C_tilde {tuple(Ct.shape)}: need u (L,L,L,L) and M <= L")
    code = dtype_code(dt)
    nbytes = check(lib.qs_transform_two_body_inplace_workspace(code, L, M), "workspace query")
    with _on_device_of(u, C, Ct):
        work = workspace.get(nbytes, u.device)
        _ran(
            lib.qs_transform_two_body_inplace(code, u.data_ptr(), C.data_ptr(), Ct.data_ptr(), work.data_ptr(),
                                              work.numel(), L, M, _stream()),
            "qs_transform_two_body_inplace",
        )
    # (the tensor itself when the size does not change: no view object keeps a second handle on the storage, so
    # the next change_basis can reuse it again)
    return u if M == L else u.reshape(-1)[: M**4].reshape(M, M, M, M)


class TransformPlan:
    """The four-index transform of a RESIDENT ``u`` captured once as a HIP graph,
    for loops that transform every step with new coefficients (the
    time-propagation pattern of BASELINE.json configs[4]: ``C(t)`` changes, ``u``
    stays).  At small l a transform is a handful of 10 us kernels and the host
    side of each call (argument checks, ctypes, allocator) costs as much as the
    GPU work; a graph replay is one launch.

        plan = TransformPlan(u, C)          # u (L,L,L,L), C (L,M), C_tilde optional
        for step in ...:
            plan.C.copy_(C_t)               # update the static coefficient buffers in place
            plan.C_tilde.copy_(Ct_t)
            out = plan.replay()             # valid until the next replay

    The captured launches are exactly those of ``transform_two_body``
    (``qs_transform_two_body`` on the capture stream); buffers, workspace and the
    output are owned by the plan."""

    def __init__(self, u, C, C_tilde=None):
        lib = _lib.load()
        if C_tilde is None:
            C_tilde = default_bra(C)
        dt = result_dtype(u, C, C_tilde)
        self.u = _dev(u, dt)
        self.C = _dev(C, dt).clone()
        self.C_tilde = _dev(C_tilde, dt).clone()
        L, M = self.C.shape
        if tuple(self.u.shape) != (L, L, L, L) or tuple(self.C_tilde.shape) != (M, L):
            raise ValueError("operand shapes do not match C")
        code = dtype_code(dt)
        nbytes = check(lib.qs_transform_two_body_workspace(code, L, M), "workspace query")
        self.out = torch.empty((M, M, M, M), dtype=dt, device=self.u.device)
        self._work = torch.empty(int(nbytes), dtype=torch.uint8, device=self.u.device)

        def launch():
            _ran(
                lib.qs_transform_two_body(
                    code, self.u.data_ptr(), self.C.data_ptr(), self.C_tilde.data_ptr(),
                    self.out.data_ptr(), self._work.data_ptr(), self._work.numel(), L, M, _stream(),
                ),
                "qs_transform_two_body",
            )

        if self.u.device.index != torch.cuda.current_device():
            raise ValueError("TransformPlan: make the device that owns `u` current first (torch.cuda.device)")

        # one eager run on a side stream (first-launch set-up: function attributes, device
        # properties), then the capture
        side = torch.cuda.Stream(device=self.u.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            launch()

    def replay(self):
        self.graph.replay()
        return self.out


@_plain
def transform_two_body_partial(u_slab, C, C_tilde=None, out=None):
    """Contractions over d, c, b of a leading-index slab:
    v[a,q,r,s] = Ct[qb] u[a,b,c,d] C[cr] C[ds] for the rows of the slab."""
    lib = _lib.load()
    if C_tilde is None:
        C_tilde = default_bra(C)
    dt = result_dtype(u_slab, C, C_tilde)
    u_slab = _dev(u_slab, dt)
    C = _dev(C, dt)
    Ct = _dev(C_tilde, dt)
    L, M = C.shape
    rows = u_slab.shape[0]
    if tuple(u_slab.shape[1:]) != (L, L, L):
        raise ValueError("slab shape does not match C")
    code = dtype_code(dt)
    nbytes = check(
        lib.qs_transform_two_body_partial_workspace(code, L, M, rows), "workspace query"
    )
    if out is None:
        out = torch.empty((rows, M, M, M), dtype=dt, device=u_slab.device)
    else:
        _check_out(out, (rows, M, M, M), dt, "transform_two_body_partial")
    with _on_device_of(u_slab, C, Ct, out):
        work = workspace.get(nbytes, u_slab.device)
        _ran(
            lib.qs_transform_two_body_partial(
                code, u_slab.data_ptr(), C.data_ptr(), Ct.data_ptr(), out.data_ptr(),
                work.data_ptr(), work.numel(), L, M, rows, _stream(),
            ),
            "qs_transform_two_body_partial",
        )
    return out


@_plain
def transform_one_body(h, C, C_tilde=None):
    """``Ct @ (h @ C)`` for one (L,L) matrix or a stack (n,L,L)
    (basis_set.py:329-334)."""
    lib = _lib.load()
    if C_tilde is None:
        C_tilde = default_bra(C)
    dt = result_dtype(h, C, C_tilde)
    h = _dev(h, dt)
    C = _dev(C, dt)
    Ct = _dev(C_tilde, dt)
    L, M = C.shape
    single = h.dim() == 2
    hs = h.reshape(-1, L, L) if not single else h.reshape(1, L, L)
    if tuple(hs.shape[1:]) != (L, L) or tuple(Ct.shape) != (M, L):
        raise ValueError("operand shapes do not match C")
    nmat = hs.shape[0]
    out = torch.empty((nmat, M, M), dtype=dt, device=h.device)
    es = 16 if dt == _C128 else 8
    with _on_device_of(hs, C, Ct):
        work = workspace.get(nmat * L * M * es, h.device)
        _ran(
            lib.qs_transform_one_body(
                dtype_code(dt), hs.data_ptr(), C.data_ptr(), Ct.data_ptr(), out.data_ptr(),
                work.data_ptr(), work.numel(), nmat, L, M, _stream(),
            ),
            "qs_transform_one_body",
        )
    return out[0] if single else out.reshape(*h.shape[:-2], M, M)


@_plain
def antisymmetrize(u, out=None):
    """``u - u.transpose(0,1,3,2)`` (basis_set.py:776-778); pass ``out=u`` for
    the in-place form."""
    lib = _lib.load()
    dt = result_dtype(u)
    src = _dev(u, dt)
    l = src.shape[-1]
    if src.dim() < 2 or src.shape[-2] != l:
        raise ValueError("last two axes must be square")
    npq = src.numel() // (l * l)
    in_place = out is u
    if out is None:
        out = torch.empty_like(src)
    elif in_place:
        out = src          # the kernel's tile-pair scheme is safe in place
    else:
        _check_out(out, src.shape, dt, "antisymmetrize")
    with _on_device_of(src, out):
        _ran(
            lib.qs_antisymmetrize(dtype_code(dt), src.data_ptr(), out.data_ptr(), npq, l, _stream()),
            "qs_antisymmetrize",
        )
    if in_place and src is not u:
        u.copy_(src)       # `u` was a view / other dtype: write the result back into it
        return u
    return out


@_plain
def spin_expand_two_body(u, antisymmetrize=False, out_dtype=None, p_lo=0, p_hi=None, out=None):
    """Spin doubling of (l,l,l,l) -> rows [2 p_lo, 2 p_hi) of (2l,2l,2l,2l)
    (basis_set.py:772-774), optionally fused with the anti-symmetrisation
    (:776-778) and the complex cast (:634).

    ``u`` may also be a leading-index slab ``u[a:b]`` of shape (rows,l,l,l) -- the
    p-sharded layout, where no rank holds the whole tensor; ``p_lo``/``p_hi`` then
    count rows of the slab (the expansion is slab-local: output row 2p+s needs
    input row p only)."""
    lib = _lib.load()
    dt = result_dtype(u)
    u = _dev(u, dt)
    if u.dim() != 4:
        raise ValueError("u must be (l,l,l,l)")
    rows, l = u.shape[0], u.shape[-1]
    if tuple(u.shape[1:]) != (l, l, l) or rows > l or rows < 1:
        raise ValueError("u must be (l,l,l,l) or a leading-index slab (rows,l,l,l)")
    p_hi = rows if p_hi is None else p_hi
    if not 0 <= p_lo < p_hi <= rows:
        raise ValueError(f"rows [{p_lo}, {p_hi}) are not inside the {rows} rows of u")
    odt = dt if out_dtype is None else out_dtype
    shape = (2 * (p_hi - p_lo), 2 * l, 2 * l, 2 * l)
    if out is None:
        out = torch.empty(shape, dtype=odt, device=u.device)
    else:
        _check_out(out, shape, odt, "spin_expand_two_body")
    with _on_device_of(u, out):
        _ran(
            lib.qs_spin_expand_two_body(
                dtype_code(dt), dtype_code(odt), u.data_ptr(), out.data_ptr(), l, p_lo, p_hi,
                1 if antisymmetrize else 0, _stream(),
            ),
            "qs_spin_expand_two_body",
        )
    return out


@_plain
def spin_expand_two_body_block(u_block, antisymmetrize=False, out_dtype=None, out=None):
    """Spin doubling of a block ``u[p0:p0+np, q0:q0+nq, :, :]`` of shape (np, nq, l, l) ->
    (2 np, 2 nq, 2l, 2l): what a rank of a sharded tensor holds, whichever of the two leading
    indices is the sharded one (basis_set.py:772-778, :634 on the rank's share)."""
    lib = _lib.load()
    dt = result_dtype(u_block)
    u_block = _dev(u_block, dt)
    if u_block.dim() != 4 or u_block.shape[2] != u_block.shape[3]:
        raise ValueError("u_block must be (np, nq, l, l)")
    np_, nq, l = u_block.shape[0], u_block.shape[1], u_block.shape[3]
    if np_ > l or nq > l or np_ < 1 or nq < 1:
        raise ValueError("block extents must be within 1..l")
    odt = dt if out_dtype is None else out_dtype
    shape = (2 * np_, 2 * nq, 2 * l, 2 * l)
    if out is None:
        out = torch.empty(shape, dtype=odt, device=u_block.device)
    else:
        _check_out(out, shape, odt, "spin_expand_two_body_block")
    with _on_device_of(u_block, out):
        _ran(
            lib.qs_spin_expand_two_body_block(
                dtype_code(dt), dtype_code(odt), u_block.data_ptr(), out.data_ptr(), l, np_, nq,
                1 if antisymmetrize else 0, _stream(),
            ),
            "qs_spin_expand_two_body_block",
        )
    return out


@_plain
def add_spin_one_body(h, out_dtype=None):
    """``kron(h, I2)`` for (l,l) or a stack (n,l,l) (basis_set.py:768-770)."""
    lib = _lib.load()
    dt = result_dtype(h)
    h = _dev(h, dt)
    l = h.shape[-1]
    if h.shape[-2] != l:
        raise ValueError("last two axes must be square")
    nmat = h.numel() // (l * l)
    odt = dt if out_dtype is None else out_dtype
    out = torch.empty(tuple(h.shape[:-2]) + (2 * l, 2 * l), dtype=odt, device=h.device)
    with _on_device_of(h):
        _ran(
            lib.qs_add_spin_one_body(
                dtype_code(dt), dtype_code(odt), h.data_ptr(), out.data_ptr(), nmat, l, _stream()
            ),
            "qs_add_spin_one_body",
        )
    return out


@_plain
def spin_squared_two_body(S, antisymmetrize=False, p_lo=0, p_hi=None):
    """Two-body S^2 from the stacked (3,n,n) spin matrices
    (basis_set.py:745-747)."""
    lib = _lib.load()
    S = _dev(S, _C128)
    n = S.shape[-1]
    if tuple(S.shape) != (3, n, n):
        raise ValueError("S must be (3,n,n)")
    p_hi = n if p_hi is None else p_hi
    out = torch.empty((p_hi - p_lo, n, n, n), dtype=_C128, device=S.device)
    with _on_device_of(S):
        _ran(
            lib.qs_spin_squared_two_body(
                S.data_ptr(), out.data_ptr(), n, p_lo, p_hi, 1 if antisymmetrize else 0, _stream()
            ),
            "qs_spin_squared_two_body",
        )
    return out


@_plain
def two_body_from_grid(K, C, C_tilde=None, antisymmetrize=False):
    """Two-body elements of an interaction that is DIAGONAL on a grid / DVR basis,

        out[p,q,r,s] = sum_ab Ct[p,a] Ct[q,b] K[a,b] C[a,r] C[b,s]

    -- the transform of a sinc-DVR ``u`` kept in its 2-d form
    (sinc_dvr/one_dim/sinc_dvr.py:217-246, optional fused anti-symmetrisation
    :247-256) and, with ``C_tilde = C.T``, the grid quadrature that builds the
    ``u`` of a 1-D quantum dot (quantum_dots/one_dim/one_dim_qd.py:275-280).
    Two GEMMs on the MFMA kernels instead of a five-operand einsum:
    ``rho[a,(p,r)] = Ct[p,a] C[a,r]``, ``W = K rho``, ``out[(p,r),(q,s)] = rho^T W``;
    O(N^2 M^2 + N M^4) for N grid points and M orbitals.  ``C_tilde`` defaults to
    ``C.conj().T`` as in the reference."""
    Ct = default_bra(C) if C_tilde is None else C_tilde
    dt = result_dtype(K, C, Ct)
    K, C, Ct = _dev(K, dt), _dev(C, dt), _dev(Ct, dt)
    N, M = C.shape
    if tuple(K.shape) != (N, N) or tuple(Ct.shape) != (M, N):
        raise ValueError(f"K {tuple(K.shape)}, C {tuple(C.shape)}, C_tilde {tuple(Ct.shape)} do not fit")
    rho = (Ct.transpose(0, 1).unsqueeze(2) * C.unsqueeze(1)).reshape(N, M * M)
    W = matmul(K, rho)                                            # (N, (q,s))
    out = matmul(rho.transpose(0, 1).contiguous(), W)             # ((p,r), (q,s))
    out = out.reshape(M, M, M, M).permute(0, 2, 1, 3).contiguous()
    if antisymmetrize:
        antisymmetrize_(out)
    return out


@_plain
def antisymmetrize_(u):
    """In-place form of ``antisymmetrize``."""
    return antisymmetrize(u, out=u)


class RcclComm:
    """The C ABI's communicator (``qs_comm_init``: RCCL over xGMI, one process per GPU) for hosts that drive the
    library from Python without ``torch.distributed``.  ``unique_id()`` on one rank, the 128 bytes to the others by
    any means, then ``RcclComm(rank, world, id_bytes)`` on every rank (collective)."""

    @staticmethod
    def unique_id():
        import ctypes

        buf = (ctypes.c_char * 128)()
        check(_lib.load().qs_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "qs_comm_unique_id")
        return bytes(buf)

    def __init__(self, rank, world, unique_id, rows_coalesce=False):
        import ctypes

        if len(unique_id) != 128:
            raise ValueError("the unique id is 128 bytes")
        self._handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(_lib.load().qs_comm_init(ctypes.byref(self._handle), int(rank), int(world),
                                       ctypes.cast(buf, ctypes.c_void_p)), "qs_comm_init")
        self.rank, self.world = int(rank), int(world)
        if rows_coalesce:
            self.set_option("rows_coalesce", 1)

    def set_option(self, key, value):
        """Per-handle option (``qs_comm_set_option``; every rank must choose the same).  ``rows_coalesce`` = 1: the rows
        exchange as ONE message per peer and step through a staging area instead of one per peer and result row."""
        check(_lib.load().qs_comm_set_option(self._handle, key.encode(), int(value)), "qs_comm_set_option")

    def close(self):
        if self._handle:
            check(_lib.load().qs_comm_destroy(self._handle), "qs_comm_destroy")
            self._handle = None

    def abort(self):
        """Tear down without waiting for outstanding operations: after a call failed in the middle of its exchange."""
        if self._handle:
            handle, self._handle = self._handle, None
            check(_lib.load().qs_comm_abort(handle), "qs_comm_abort")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @_plain
    def transform_two_body_rows(self, rows, C, C_tilde=None, in_part=None, chunk_rows=0, out=None):
        """Rows of one leading index of ``u`` in, rows of the other transformed leading index out, everything else
        O(chunk_rows l^3): ``qs_transform_two_body_sharded_rows`` (see include/qs_amd.h and
        ``sharded.transform_two_body_rows``, the same algorithm on torch.distributed).  ``rows`` (il, L, L, L) follows
        ``in_part`` (a ``sharded.SlabPartition``; balanced by default); a real ``rows`` against complex coefficients
        is not copied to complex.  Returns the (jl, M, M, M) view of the result buffer (``out`` may supply the flat
        buffer to reuse it across the steps of a time loop)."""
        import ctypes

        lib = _lib.load()
        if C_tilde is None:
            C_tilde = default_bra(C)
        dt = result_dtype(rows, C, C_tilde)
        in_dt = _F64 if (dt == _C128 and rows.dtype == _F64 and mixed_real_u) else dt
        rows, C, Ct = _dev(rows, in_dt), _dev(C, dt), _dev(C_tilde, dt)
        L, M = C.shape
        code = dtype_code(dt)
        starts = None
        if in_part is not None:
            if (in_part.n, in_part.world) != (L, self.world):
                raise ValueError("the input partition does not describe L rows over this communicator's ranks")
            starts = (ctypes.c_int64 * (self.world + 1))(*in_part.starts)
            il = in_part.count(self.rank)
        else:
            base, extra = divmod(L, self.world)
            il = base + (1 if self.rank < extra else 0)
        base, extra = divmod(M, self.world)
        jl = base + (1 if self.rank < extra else 0)
        if tuple(rows.shape) != (il, L, L, L) or tuple(Ct.shape) != (M, L):
            raise ValueError(f"rank {self.rank}: expected rows of shape {(il, L, L, L)} and C_tilde {(M, L)}")
        p_starts = ctypes.cast(starts, ctypes.c_void_p) if starts is not None else None
        ni = int(chunk_rows)
        if ni < 1:
            ni = check(lib.qs_sharded_rows_default_chunk(code, L, M, self.world, p_starts), "chunk query")
        es = 16 if dt == _C128 else 8
        out_bytes = check(lib.qs_transform_two_body_sharded_rows_out_bytes(code, L, M, self.world, self.rank), "size query")
        if out is None:
            out = torch.empty(out_bytes // es, dtype=dt, device=rows.device)
        elif (not isinstance(out, torch.Tensor) or out.dtype != dt or out.numel() * es < out_bytes
              or not out.is_contiguous() or out.device != rows.device):
            raise ValueError(f"`out` must be a contiguous {dt} buffer of at least {out_bytes // es} elements")
        nbytes = check(lib.qs_comm_rows_workspace(self._handle, code, L, M, ni), "workspace query")
        with _on_device_of(rows, C, Ct, out):
            work = workspace.get(nbytes, rows.device)
            _ran(
                lib.qs_transform_two_body_sharded_rows(
                    self._handle, dtype_code(in_dt), code, rows.data_ptr(), p_starts, C.data_ptr(), Ct.data_ptr(),
                    out.data_ptr(), out.numel() * es, work.data_ptr(), work.numel(), L, M, ni, _stream(),
                ),
                "qs_transform_two_body_sharded_rows",
            )
        return out.reshape(-1)[: jl * M * M * M].view(jl, M, M, M)

    def transform_two_body(self, u_bslab, C, C_tilde=None, out=None, nchunks=4):
        """``out[p_lo:p_hi]`` of the transform from ``u[:, b_lo:b_hi]`` (balanced splits): the whole sharded
        transform in ONE C-ABI call -- local contractions, chunked grouped send / receive on the communicator's
        stream overlapped with them, closing contraction (``qs_transform_two_body_sharded``)."""
        lib = _lib.load()
        if C_tilde is None:
            C_tilde = default_bra(C)
        dt = result_dtype(u_bslab, C, C_tilde)
        u_bslab, C, Ct = _dev(u_bslab, dt), _dev(C, dt), _dev(C_tilde, dt)
        L, M = C.shape
        code = dtype_code(dt)
        base, extra = divmod(M, self.world)
        pc = base + (1 if self.rank < extra else 0)
        base, extra = divmod(L, self.world)
        bl = base + (1 if self.rank < extra else 0)
        if tuple(u_bslab.shape) != (L, bl, L, L) or tuple(Ct.shape) != (M, L):
            raise ValueError(f"rank {self.rank}: expected a slab of shape {(L, bl, L, L)} and C_tilde {(M, L)}")
        if out is None:
            out = torch.empty((pc, M, M, M), dtype=dt, device=u_bslab.device)
        else:
            _check_out(out, (pc, M, M, M), dt, "RcclComm.transform_two_body")
        nbytes = check(lib.qs_transform_two_body_sharded_workspace(code, L, M, self.world, self.rank), "workspace query")
        with _on_device_of(u_bslab, C, Ct, out):
            work = workspace.get(nbytes, u_bslab.device)
            _ran(
                lib.qs_transform_two_body_sharded(
                    self._handle, code, u_bslab.data_ptr(), C.data_ptr(), Ct.data_ptr(), out.data_ptr(),
                    work.data_ptr(), work.numel(), L, M, int(nchunks), _stream(),
                ),
                "qs_transform_two_body_sharded",
            )
        return out


def tuning_set(key, value):
    """Tuning / test hook: kernel-choice override for the CALLING THREAD only
    (the library keeps no process-global mutable state)."""
    check(_lib.load().qs_tuning_set(key.encode(), int(value)), "qs_tuning_set")


def tuning_reset():
    """Back to the automatic kernel choice on the calling thread."""
    check(_lib.load().qs_tuning_reset(), "qs_tuning_reset")


class tuning:
    """``with tuning(gemm_fast=0): ...`` -- overrides for the block, automatic policy after it."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        for key, value in self.knobs.items():
            tuning_set(key, value)
        return self

    def __exit__(self, *exc):
        tuning_reset()
        return False


def last_dispatch():
    """Names (as rocprofv3 prints them) of the kernels the calling thread's most
    recent library call launched, e.g. ``qs::gemm_fast_kernel<false, 4, 4, true, false> x4``."""
    return _lib.load().qs_last_dispatch().decode()
