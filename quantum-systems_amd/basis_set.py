"""``BasisSet``: container of second-quantised matrix elements whose basis
changes, anti-symmetrisation and spin doubling run on the MI355X.

API mirror of the reference class (quantum_systems/basis_set.py): same
constructor, attribute names, static-method signatures, mutation semantics and
assertions, so code written against the reference runs unchanged.  What is
different is where the arithmetic happens: every method that touches the
rank-4 tensor (and the one-body transforms) hands device pointers to
``libqs_amd.so``.  There is no NumPy implementation in this file:

* with the device array module (``array_module.hip``) the arrays stay resident
  in HBM between calls;
* with plain ``numpy`` as the module (the reference's default) inputs are
  staged to the GPU, computed there and copied back -- the same kernels, plus
  PCIe.  Without a GPU these methods raise.

Reference line numbers in the docstrings are those of
quantum_systems/basis_set.py unless another file is named.
"""

import copy
import sys
import warnings

import numpy
import torch

from . import kernels, sharded_basis
from .array_module import convert, is_device_module, to_host, wrap
from .sharded_module import is_sharded, is_sharded_module
from .system_helper import compute_particle_density

_ARRAY_FIELDS = (
    "_h", "_s", "_u", "_spf", "_bra_spf", "_position", "_momentum",
    "_spin_x", "_spin_y", "_spin_z", "_spin_2", "_spin_2_tb",
)


def _stage(arr):
    """Device tensor for a NumPy array, torch tensor or nested list."""
    if isinstance(arr, torch.Tensor):
        if arr.is_cuda:
            return arr
        return arr.cuda()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "quantum_systems_amd computes on the GPU only and no GPU is visible "
            "(there is no CPU fallback for the transform path)"
        )
    return torch.from_numpy(numpy.ascontiguousarray(numpy.asarray(arr))).cuda()


def _deliver(t, np):
    """Hand a device result back in the caller's array module."""
    if is_device_module(np):
        return wrap(t)
    return np.asarray(to_host(t))


def _ownership_signature(t):
    """Every count through which somebody else could reach the storage of ``t``: the storage's own use count (tensors
    made with ``set_`` on it, views, the alias ``wrap`` makes), the tensor implementation's, and -- a device array made
    by ``wrap`` is an alias of the tensor it was made from, ``t._base``, and views of it attach to that base -- the
    base's Python references and implementation count.  The numbers themselves are CPython / PyTorch internals;
    they are only ever compared with what the same function returns for a freshly made array nobody else holds
    (``_UNSHARED`` below), so an interpreter that counts differently cannot turn sharing into "unshared"."""
    base = t._base
    return (
        torch._C._storage_Use_Count(t.untyped_storage()._cdata),
        t._use_count(),
        base is not None,
        sys.getrefcount(base) if base is not None else 0,
        base._use_count() if base is not None else 0,
        base is None or base._base is None,
    )


def _refs_to_attr(obj, slot):
    """Python references to the array stored in ``obj.<slot>`` as counted from inside this helper (compared with the
    count of an attribute nobody else holds, never with a literal)."""
    arr = getattr(obj, slot)
    return sys.getrefcount(arr)


class _Holder:
    pass


def _calibrate_unshared():
    """Signatures of arrays that are provably unshared, one per way a device array comes into being: wrapped by
    ``wrap`` (an alias with a base) and returned by a tensor operation."""
    sigs = set()
    for make in (lambda: wrap(torch.empty(4)), lambda: wrap(torch.empty(4)) * 1):
        h = _Holder()
        h._x = make()
        refs = _refs_to_attr(h, "_x")
        sigs.add((refs,) + _ownership_signature(h._x))
    return sigs


_UNSHARED = _calibrate_unshared()


def _sole_owner(obj, slot):
    """True when nothing but ``obj.<slot>`` can reach the storage of the array stored there: no second Python
    reference, no view, no storage-level alias.  Anything unexpected reads as "shared" (no donation)."""
    refs = _refs_to_attr(obj, slot)
    t = getattr(obj, slot)
    if not isinstance(t, torch.Tensor):
        return False
    if t.storage_offset() != 0 or t.untyped_storage().nbytes() != t.numel() * t.element_size():
        return False
    return (refs,) + _ownership_signature(t) in _UNSHARED


def _module_of(np):
    return numpy if np is None else np


class _Checked:
    """Attribute whose every axis must equal ``owner.l`` (:87-115, :780-782)."""

    def __init__(self, needs_spin=False, stacked=False, check=True):
        self.needs_spin, self.stacked, self.check = needs_spin, stacked, check

    def __set_name__(self, owner, name):
        self.slot = "_" + name

    def __get__(self, obj, objtype=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        if self.needs_spin:
            assert obj.includes_spin
        if self.check:
            if self.stacked:
                assert len(value) == obj.dim
                for block in value:
                    assert all(obj.check_axis_lengths(block, obj.l))
            else:
                assert all(obj.check_axis_lengths(value, obj.l))
        if type(value) is torch.Tensor and is_device_module(obj.np):
            value = wrap(value)         # (a plain tensor made while subclass dispatch was off)
        setattr(obj, self.slot, value)


class BasisSet:
    """Matrix elements ``h``, ``s``, ``u``, ``position``... of ``l`` basis
    functions in ``dim`` dimensions, held by the array module ``np``.

    Parameters follow the reference constructor (:32-34): ``l``, ``dim``,
    ``np=None`` (NumPy), ``includes_spin=False``, ``anti_symmetrized_u=False``.
    """

    h = _Checked()
    u = _Checked()
    s = _Checked()
    position = _Checked(stacked=True)
    momentum = _Checked(stacked=True)
    spin_x = _Checked(needs_spin=True)
    spin_y = _Checked(needs_spin=True)
    spin_z = _Checked(needs_spin=True)
    spin_2 = _Checked(needs_spin=True)
    sigma_x = _Checked(needs_spin=True, check=False)
    sigma_y = _Checked(needs_spin=True, check=False)
    sigma_z = _Checked(needs_spin=True, check=False)

    # change_basis reuses the storage of the tensor it drops from this many orbitals up (below, the fused
    # small-basis passes are faster and memory is no concern; None switches the reuse off).  None of the reference's
    # semantics change: the storage is only reused when no other reference, view or storage alias of the old array
    # exists (_sole_owner)
    donate_u_from = 96

    def __init__(self, l, dim, np=None, includes_spin=False, anti_symmetrized_u=False):
        self.np = _module_of(np)
        self.l = l
        self.dim = dim
        self._grid = None
        for slot in _ARRAY_FIELDS + ("_sigma_x", "_sigma_y", "_sigma_z"):
            setattr(self, slot, None)
        self._spin_2_tb_recipe = None
        self._spin_2_tb_version = -1
        self._nuclear_repulsion_energy = 0
        self.particle_charge = -1  # electrons
        self._includes_spin = includes_spin
        self._anti_symmetrized_u = anti_symmetrized_u

    # ------------------------------------------------------------ plain props
    @property
    def includes_spin(self):
        return self._includes_spin

    @property
    def anti_symmetrized_u(self):
        return self._anti_symmetrized_u

    @property
    def grid(self):
        return self._grid

    @grid.setter
    def grid(self, grid):
        self._grid = grid

    @property
    def dipole_moment(self):
        return self.particle_charge * self.position

    @property
    def nuclear_repulsion_energy(self):
        return self._nuclear_repulsion_energy

    @nuclear_repulsion_energy.setter
    def nuclear_repulsion_energy(self, value):
        self._nuclear_repulsion_energy = value

    @property
    def spf(self):
        return self._spf

    @spf.setter
    def spf(self, spf):
        if spf is not None:
            assert spf.shape[0] == self.l
            assert len(tuple(spf.shape[1:])) == self.dim
        self._spf = spf

    @property
    def bra_spf(self):
        # Hermitian basis: the dual functions are the conjugates (:244-250)
        if self._bra_spf is None and self._spf is not None:
            self._bra_spf = self._spf.conj()
        return self._bra_spf

    @bra_spf.setter
    def bra_spf(self, bra_spf):
        if bra_spf is not None:
            assert bra_spf.shape[0] == self.l
            assert len(tuple(bra_spf.shape[1:])) == self.dim
        self._bra_spf = bra_spf

    @property
    def spin_2_tb(self):
        """Two-body S^2.  At (2l)^4 complex elements it is as large as ``u``; the spin doubling records how to build
        it -- ``sum_i S_i[p,r] S_i[q,s]`` (:745-747), minus the r <-> s term once anti-symmetrised (:525-526), from
        three (2l, 2l) matrices -- and the tensor is produced on first access (SURVEY 7 "memory capacity").  The
        recipe outlives the tensor it built for as long as nobody writes into that tensor: ``change_basis`` then
        transforms the three matrices (O(l^3)) instead of the tensor (O(l^5), :379-382) and the tensor is rebuilt on the
        next access."""
        if self._spin_2_tb is None and self._spin_2_tb_recipe is not None:
            if is_sharded_module(self.np):
                self._spin_2_tb = sharded_basis.spin_2_tb_rows(self)      # this rank's rows only
                self._spin_2_tb_version = self._spin_2_tb.rows._version
            else:
                stack, anti = self._spin_2_tb_recipe
                self._spin_2_tb = _deliver(
                    kernels.spin_squared_two_body(_stage(stack), antisymmetrize=anti), self.np
                )
                if isinstance(self._spin_2_tb, torch.Tensor):
                    self._spin_2_tb_version = self._spin_2_tb._version
                else:
                    self._spin_2_tb_recipe = None      # a host array: writes into it cannot be noticed
        return self._spin_2_tb

    @spin_2_tb.setter
    def spin_2_tb(self, value):
        assert self.includes_spin
        assert all(self.check_axis_lengths(value, self.l))
        self._spin_2_tb_recipe = None
        self._spin_2_tb = value

    def _spin_2_tb_recipe_valid(self):
        """True while the recipe (three spin matrices + the anti-symmetry flag) still IS ``spin_2_tb``: the tensor has
        not been built, or the copy handed out has not been written to since (the version counter of the device
        tensor).  A recipe that no longer holds is dropped."""
        if self._spin_2_tb_recipe is None:
            return False
        t = self._spin_2_tb
        if t is None:
            return True
        loc = t.rows if is_sharded(t) else t
        if isinstance(loc, torch.Tensor) and loc._version == self._spin_2_tb_version:
            return True
        self._spin_2_tb_recipe = None
        return False

    # --------------------------------------------------------- module plumbing
    @staticmethod
    def change_arr_module(arr, np):
        return convert(arr, np)

    def change_module(self, np):
        """Re-home every stored array in ``np`` (:272-296).  NumPy -> device
        uploads, device -> NumPy downloads."""
        self.np = np
        self.bra_spf  # materialise the lazy dual before converting, as :287 does
        if self._spin_2_tb_recipe_valid():
            self._spin_2_tb = None        # rebuilt in the new module on the next access: no (2l)^4 transfer
        for slot in _ARRAY_FIELDS:
            arr = getattr(self, slot)
            if slot in ("_u", "_spin_2_tb") and arr is not None and len(arr.shape) == 4:
                if is_sharded_module(np):
                    # rank-4 tensors become one slab per rank: only this rank's rows are kept / uploaded
                    setattr(self, slot, np.shard(arr, axis=arr.axis if is_sharded(arr) else 0))
                    continue
                if is_sharded(arr):
                    arr = arr.gather()      # leaving the sharded module: the whole tensor on every rank
            setattr(self, slot, convert(arr, np))
        if self._spin_2_tb_recipe is not None:
            stack, anti = self._spin_2_tb_recipe
            self._spin_2_tb_recipe = (convert(stack, np), anti)

    def cast_to_complex(self):
        """Every stored array -> complex128 (:298-319)."""
        np = self.np
        self.bra_spf
        for slot in _ARRAY_FIELDS:
            arr = getattr(self, slot)
            if arr is not None:
                setattr(self, slot, arr.astype(np.complex128))

    def copy_basis(self):
        """Deep copy (:784-805); device arrays are cloned on the device."""
        np = self.np
        self.np = None
        try:
            new = copy.deepcopy(self)
        finally:
            self.np = np
        new.np = np
        return new

    # ------------------------------------------------------------- transforms
    @staticmethod
    def transform_spf(spf, C, np):
        """spf'[p, g] = sum_a C[a, p] spf[a, g] (:321-323)."""
        d_spf, d_C = _stage(spf), _stage(C)
        L = d_C.shape[0]
        out = kernels.matmul(d_C.transpose(0, 1), d_spf.reshape(L, -1))
        return _deliver(out.reshape((d_C.shape[1],) + tuple(d_spf.shape[1:])), np)

    @staticmethod
    def transform_bra_spf(bra_spf, C_tilde, np):
        """bra'[p, g] = sum_a Ct[p, a] bra[a, g] (:325-327)."""
        d_bra, d_Ct = _stage(bra_spf), _stage(C_tilde)
        L = d_Ct.shape[1]
        out = kernels.matmul(d_Ct, d_bra.reshape(L, -1))
        return _deliver(out.reshape((d_Ct.shape[0],) + tuple(d_bra.shape[1:])), np)

    @staticmethod
    def transform_one_body_elements(h, C, np, C_tilde=None):
        """``Ct (h C)`` with ``Ct = C^dagger`` unless given (:329-334)."""
        Ct = None if C_tilde is None else _stage(C_tilde)
        return _deliver(kernels.transform_one_body(_stage(h), _stage(C), Ct), np)

    @staticmethod
    def transform_two_body_elements(u, C, np, C_tilde=None):
        """out[pqrs] = Ct[pa] Ct[qb] u[abcd] C[cr] C[ds] (:336-350), four
        single-index contractions in the reference's order d, c, b, a.  Returns
        a new array; ``u`` is left untouched.  A sharded ``u`` (``np`` = the sharded module) returns a
        sharded result: one all-to-all, see sharded_basis.transform_two_body."""
        if is_sharded(u):
            d_C = np.asarray(C).as_subclass(torch.Tensor).contiguous()
            d_Ct = (kernels.default_bra(d_C) if C_tilde is None
                    else np.asarray(C_tilde).as_subclass(torch.Tensor).contiguous())
            return sharded_basis.transform_two_body(u, d_C, d_Ct, np)
        Ct = None if C_tilde is None else _stage(C_tilde)
        return _deliver(kernels.transform_two_body(_stage(u), _stage(C), Ct), np)

    def get_transformed_h(self, C):
        return self.transform_one_body_elements(self.h, C, np=self.np)

    def get_transformed_u(self, C):
        return self.transform_two_body_elements(self.u, C, np=self.np)

    def change_basis(self, C, C_tilde=None):
        """In-place change of basis with ket coefficients ``C`` (l_old, l_new)
        and bra coefficients ``C_tilde`` (l_new, l_old), default ``C^dagger``
        (:413-464).  Rectangular ``C`` changes ``l``."""
        if is_sharded_module(self.np):
            return sharded_basis.change_basis(self, C, C_tilde)
        # (the host side of a small basis costs more than its kernels: inside, device arrays are handled as plain
        # tensors -- no __torch_function__ round trip per tensor method -- and re-wrapped when they are stored)
        with torch._C.DisableTorchFunctionSubclass():
            return self._change_basis_on_device(C, C_tilde)

    def _change_basis_on_device(self, C, C_tilde):
        np = self.np
        self.l = C.shape[1]                                     # :448
        d_C = _stage(C)
        d_Ct = kernels.default_bra(d_C) if C_tilde is None else _stage(C_tilde)

        # real coefficients against complex matrices (the 2-D oscillator's spf, a complex h): the complex copies of
        # C and C~ are made once per call, not once per transformed array
        cast = {}

        def coeffs(arr):
            if d_C.is_complex() or not (isinstance(arr, torch.Tensor) and arr.is_complex()):
                return d_C, d_Ct
            if not cast:
                cast["C"], cast["Ct"] = d_C.to(torch.complex128), d_Ct.to(torch.complex128)
            return cast["C"], cast["Ct"]

        def one_body(arr):
            arr = _stage(arr)
            c, ct = coeffs(arr)
            return _deliver(kernels.transform_one_body(arr, c, ct), np)

        # h, s, position[dim], momentum[dim] (:358-366, :384-406) as ONE stacked call -- two launches for all of them
        # instead of two per array, one trip through the wrapper (the host side of a small basis costs more than its
        # kernels) -- when they share a dtype; the results are handed out as the slices of one (n, M, M) array
        names = [k for k in ("h", "s", "position", "momentum") if getattr(self, k) is not None]
        mats = [_stage(getattr(self, k)) for k in names]
        stacked = len(mats) > 1 and all(isinstance(m, torch.Tensor) and m.dtype == mats[0].dtype
                                        and tuple(m.shape[-2:]) == tuple(mats[0].shape[-2:]) for m in mats)
        if stacked:
            L_old = mats[0].shape[-1]
            counts = [m.numel() // (L_old * L_old) for m in mats]
            pile = torch.cat([m.reshape(-1, L_old, L_old) for m in mats])
            c, ct = coeffs(pile)
            res = kernels.transform_one_body(pile, c, ct)
            done, at = {}, 0
            for k, m, cnt in zip(names, mats, counts):
                part = res[at] if m.dim() == 2 else res[at:at + cnt]
                done[k] = _deliver(part, np)
                at += cnt
            self.h = done["h"]
            if "s" in done:
                self.s = done["s"]
        else:
            done = {}
            self.h = one_body(self.h)
            if self.s is not None:
                self.s = one_body(self.s)
        # :368-372 transforms spin_x/y/z/spin_2 into a loop local and drops
        # the result; they keep their old values (and shapes) here as well.

        # through the (overridable) method, as :374-382 does -- ODSincDVR replaces it and looks
        # at the stored u while doing so; the old tensor is released when the attribute is rebound.
        # The reference DROPS the old tensor here, so when nobody else can see it (no second Python
        # reference, no view) its storage is reused: the transform runs in place with one spare buffer
        # instead of workspace + result -- 69 GB instead of 103 GB at l = 256.
        for slot in ("_u", "_spin_2_tb"):
            if slot == "_spin_2_tb" and self._spin_2_tb_recipe_valid():
                # :379-382 transforms the (2l)^4 tensor; it is sum_i S_i (x) S_i (minus the r <-> s term), so its
                # transform is the same expression in S'_i = C~ S_i C: three one-body transforms, and the tensor is
                # rebuilt from them when somebody asks for it.  (spin_x/y/z themselves keep their old values, 0.6.)
                stack, anti = self._spin_2_tb_recipe
                d_stack = _stage(stack)
                c, ct = coeffs(d_stack)
                self._spin_2_tb = None
                self._spin_2_tb_recipe = (_deliver(kernels.transform_one_body(d_stack, c, ct), np), anti)
                continue
            if slot == "_spin_2_tb" and self.spin_2_tb is None:  # :379-382
                continue
            unshared = _sole_owner(self, slot)                    # (asked before `old` adds a reference of its own)
            old = getattr(self, slot)
            donate = (
                unshared and self.donate_u_from is not None
                and type(self).transform_two_body_elements is BasisSet.transform_two_body_elements
                and is_device_module(np) and isinstance(old, torch.Tensor) and old.is_cuda
                and len(old.shape) == 4 and old.is_contiguous()
                and C.shape[1] <= C.shape[0] and C.shape[0] >= self.donate_u_from
                and old.dtype == kernels.result_dtype(old, d_C, d_Ct)
            )
            if donate:
                setattr(self, slot, None)
                plain = old.as_subclass(torch.Tensor)
                del old
                try:
                    res = wrap(kernels.transform_two_body_(plain, d_C, d_Ct))
                except BaseException:
                    # refused before anything was launched (workspace allocation, argument checks): the tensor is
                    # intact and stays the basis set's
                    setattr(self, slot, wrap(plain))
                    raise
                del plain
            else:
                res = self.transform_two_body_elements(old, d_C, np, C_tilde=d_Ct)
                del old
            if slot == "_u":
                self.u = res
            else:
                self.spin_2_tb = res
            del res

        if self.position is not None:
            self.position = done["position"] if "position" in done else one_body(self.position)    # (dim, l, l)
        if self.momentum is not None:
            self.momentum = done["momentum"] if "momentum" in done else one_body(self.momentum)
        if self.spf is not None:
            bra = self.transform_bra_spf(self.bra_spf, coeffs(_stage(self.bra_spf))[1], np)
            ket = self.transform_spf(self.spf, coeffs(_stage(self.spf))[0], np)
            self.bra_spf = bra
            self.spf = ket

    def change_basis_plan(self, C_tilde_given=False):
        """``change_basis`` with a SQUARE coefficient matrix, captured once as a HIP graph (``ChangeBasisPlan``): for loops
        that rotate a small basis again and again (orbital optimisation; :413-464 every iteration), where the Python and
        launch side of a call costs several times its kernels."""
        return ChangeBasisPlan(self, C_tilde_given)

    def compute_particle_density(self, rho_qp, C=None, C_tilde=None):
        """rho(r) = bra_q(r) rho_qp ket_p(r), optionally in a rotated basis
        (:466-509)."""
        assert (
            self._spf is not None
        ), "Set up single-particle functions prior to calling this function"
        ket, bra = self.spf, self.bra_spf
        if C is not None:
            ket = self.transform_spf(ket, C, self.np)
            C_tilde = C_tilde if C_tilde is not None else C.conj().T
            bra = self.transform_bra_spf(bra, C_tilde, self.np)
        return compute_particle_density(rho_qp, ket, bra, self.np)

    # ------------------------------------------- anti-symmetry and spin doubling
    @staticmethod
    def anti_symmetrize_u(_u):
        """``u[pqrs] - u[pqsr]`` as a new array (:776-778)."""
        if is_sharded(_u):
            return _u._like(kernels.antisymmetrize(_u.rows))
        out = kernels.antisymmetrize(_stage(_u))
        if isinstance(_u, torch.Tensor):
            return wrap(out)
        return to_host(out)

    def _statics_overridden(self):
        """True when a subclass replaces the spin / anti-symmetry statics (ODSincDVR does, for its 2-d
        ``u``): the drivers then go through ``self.add_spin_two_body`` / ``self.anti_symmetrize_u`` exactly
        as :523 and :576 do, instead of the fused kernels written for the rank-4 tensor."""
        cls = type(self)
        return (cls.add_spin_two_body is not BasisSet.add_spin_two_body
                or cls.anti_symmetrize_u is not BasisSet.anti_symmetrize_u)

    def anti_symmetrize_two_body_elements(self):
        """Anti-symmetrise ``u`` and ``spin_2_tb`` once (:511-528)."""
        if self._anti_symmetrized_u:
            return
        if is_sharded(self._u):
            return sharded_basis.anti_symmetrize_two_body_elements(self)
        if self._statics_overridden() or len(self.u.shape) != 4:
            self.u = self.anti_symmetrize_u(self.u)                       # :523
            if self.spin_2_tb is not None:
                self.spin_2_tb = self.anti_symmetrize_u(self.spin_2_tb)   # :525-526
            self._anti_symmetrized_u = True
            return
        d = _stage(self.u)
        on_device = d is self.u or isinstance(self.u, torch.Tensor)
        res = kernels.antisymmetrize(d, out=d if not on_device else None)
        self.u = _deliver(res, self.np)
        if self._spin_2_tb_recipe_valid():
            stack, _ = self._spin_2_tb_recipe
            self._spin_2_tb_recipe = (stack, True)
            self._spin_2_tb = None                         # rebuilt anti-symmetrised on the next access
        elif self._spin_2_tb is not None:
            self.spin_2_tb = _deliver(kernels.antisymmetrize(_stage(self._spin_2_tb)), self.np)
        self._anti_symmetrized_u = True

    @staticmethod
    def add_spin_one_body(h, np):
        """``kron(h, I2)`` (:768-770)."""
        return _deliver(kernels.add_spin_one_body(_stage(h)), np)

    @staticmethod
    def add_spin_two_body(_u, np):
        """``kron(u, delta_pr delta_qs)`` (:772-774): 16x the elements, 4 of 16
        spin blocks non-zero."""
        return _deliver(kernels.spin_expand_two_body(_stage(_u)), np)

    @staticmethod
    def add_spin_spf(spf, np):
        """Each spatial function appears once per spin direction: rows
        interleaved (:751-759)."""
        d = _stage(spf)
        out = torch.repeat_interleave(d, 2, dim=0)
        return _deliver(out, np)

    @staticmethod
    def add_spin_bra_spf(bra_spf, np):
        if bra_spf is None:
            return None
        return BasisSet.add_spin_spf(bra_spf, np)

    @staticmethod
    def setup_pauli_matrices(a, b, np):
        """Pauli matrices in the orthonormal spinor basis {a, b} (column
        vectors): element [i, j] = <s_i| sigma |s_j> (:638-697).

        >>> import numpy as np
        >>> a = np.array([1, 0]).reshape(-1, 1)
        >>> b = np.array([0, 1]).reshape(-1, 1)
        >>> sx, sy, sz = BasisSet.setup_pauli_matrices(a, b, np)
        >>> print(sz)
        [[ 1.+0.j  0.+0.j]
         [ 0.+0.j -1.+0.j]]
        """
        sv = numpy.concatenate(
            [to_host(a).reshape(2, 1), to_host(b).reshape(2, 1)], axis=1
        ).astype(numpy.complex128)                 # columns a, b
        paulis = (
            numpy.array([[0, 1], [1, 0]], dtype=numpy.complex128),
            numpy.array([[0, -1j], [1j, 0]], dtype=numpy.complex128),
            numpy.array([[1, 0], [0, -1]], dtype=numpy.complex128),
        )
        # 2x2 bookkeeping, done on the host
        out = tuple(sv.conj().T @ (sig @ sv) for sig in paulis)
        return tuple(convert(m, np) for m in out)

    @staticmethod
    def setup_spin_squared_operator(spin_x, spin_y, spin_z, overlap, np):
        """One-body ``sum_i S_i s S_i`` and two-body
        ``sum_i S_i[pr] S_i[qs]`` parts of S^2 (:699-749)."""
        stack = torch.stack([_stage(spin_x), _stage(spin_y), _stage(spin_z)]).to(torch.complex128)
        d_s = _stage(overlap).to(torch.complex128)
        spin_2 = None
        for k in range(3):  # same accumulation order as :745-746
            term = kernels.matmul(stack[k], kernels.matmul(d_s, stack[k]))
            spin_2 = term if spin_2 is None else spin_2 + term
        tb = kernels.spin_squared_two_body(stack, antisymmetrize=False)
        return _deliver(spin_2, np), _deliver(tb, np)

    def change_to_general_orbital_basis(self, a=[1, 0], b=[0, 1], anti_symmetrize=True):
        """Spin doubling in place (:530-636): every spatial orbital becomes an
        alpha and a beta spin orbital (index P = 2p + sigma), ``u`` gets the
        delta-Kronecker expansion (fused here with the anti-symmetrisation and
        the complex cast: one read of ``u``, one write of the 16x tensor), spin
        operators are set up in the spinor basis {a, b} and all arrays end up
        complex128.  Returns ``self``; warns and returns ``None`` when the basis
        already carries spin."""
        if is_sharded_module(self.np):
            return sharded_basis.change_to_general_orbital_basis(self, a=a, b=b, anti_symmetrize=anti_symmetrize)
        if self._includes_spin:
            warnings.warn(
                "The basis has already been spin-doubled. Avoiding a second doubling."
            )
            return None
        np = self.np
        c128 = torch.complex128
        self._includes_spin = True
        self.l = 2 * self.l

        d_overlap = _stage(self.s)
        d_h = kernels.add_spin_one_body(_stage(self.h), out_dtype=c128)
        d_s = kernels.add_spin_one_body(d_overlap, out_dtype=c128)

        old_u = self._u
        self._u = None
        fused = not self._statics_overridden() and len(old_u.shape) == 4
        if fused:
            # one read of the spatial tensor, one write of the 16x spin tensor: expansion, anti-symmetrisation
            # (:605-606) and the complex cast (:634) in one kernel
            anti_now = bool(anti_symmetrize) and not self._anti_symmetrized_u
            d_u = kernels.spin_expand_two_body(_stage(old_u), antisymmetrize=anti_now, out_dtype=c128)
            new_u = _deliver(d_u, np)
            del d_u
        else:
            # a subclass's own expansion (ODSincDVR: kron(K, ones(2, 2)) of its 2-d u), as :576 does; the
            # anti-symmetrisation follows below through self.anti_symmetrize_u
            anti_now = False
            new_u = self.add_spin_two_body(old_u, np=np)
        del old_u

        self.h = _deliver(d_h, np)
        self.s = _deliver(d_s, np)
        self.u = new_u
        del new_u

        if getattr(self, "u_repr", "4d") != "2d":
            av = numpy.asarray(to_host(a)).astype(numpy.complex128).reshape(-1, 1)
            bv = numpy.asarray(to_host(b)).astype(numpy.complex128).reshape(-1, 1)
            assert abs(numpy.dot(av.conj().T, av) - 1) < 1e-12
            assert abs(numpy.dot(bv.conj().T, bv) - 1) < 1e-12
            assert abs(numpy.dot(av.conj().T, bv)) < 1e-12
            self.a, self.b = convert(av, np), convert(bv, np)
            sig = self.setup_pauli_matrices(av, bv, numpy)
            self.sigma_x, self.sigma_y, self.sigma_z = (convert(m, np) for m in sig)
            d_ov = d_overlap.to(c128)
            stack = torch.stack(
                [0.5 * torch.kron(d_ov, torch.from_numpy(m).to(d_ov.device)) for m in sig]
            )
            self.spin_x, self.spin_y, self.spin_z = (_deliver(stack[k], np) for k in range(3))
            spin_2 = None
            for k in range(3):
                term = kernels.matmul(stack[k], kernels.matmul(d_s, stack[k]))
                spin_2 = term if spin_2 is None else spin_2 + term
            self.spin_2 = _deliver(spin_2, np)
            # (2l)^4 complex: built on first access, see the property
            self._spin_2_tb = None
            self._spin_2_tb_recipe = (_deliver(stack, np), anti_now)

        if anti_symmetrize:
            if fused:
                self._anti_symmetrized_u = True       # done inside the expansion kernel (and the lazy spin_2_tb)
            else:
                self.anti_symmetrize_two_body_elements()                  # :605-606

        if self.position is not None:
            self.position = _deliver(
                kernels.add_spin_one_body(_stage(self.position), out_dtype=c128), np)
        if self.momentum is not None:
            self.momentum = _deliver(
                kernels.add_spin_one_body(_stage(self.momentum), out_dtype=c128), np)
        if self.spf is not None:
            had_bra = self._bra_spf is not None
            old_bra = self._bra_spf
            self.spf = self.add_spin_spf(self.spf, np)
            if had_bra:
                self.bra_spf = self.add_spin_bra_spf(old_bra, np)
        self.cast_to_complex()
        return self

    @staticmethod
    def check_axis_lengths(arr, length):
        return [length == axis for axis in arr.shape]


class ChangeBasisPlan:
    """``BasisSet.change_basis(C)`` for a SQUARE ``C`` on the device array module, captured once as a HIP graph.

    Below ~32 orbitals a change of basis is a handful of 5-10 us kernels while the Python side of ``change_basis`` (staging,
    wrapping, ctypes, allocator, setters) costs ~55 us (profiles/r03_api_overhead.txt).  The plan does that work ONCE: the
    stacked one-body call (h, s, position, momentum), the two-body call (u) and the transform of the spin_2_tb recipe are
    captured on buffers the plan owns -- two sets of them, so that the arrays of call n are the input of call n + 1 --
    and a call is: copy ``C`` into the plan's coefficient buffer, replay, rebind the attributes.

        plan = basis.change_basis_plan()
        for it in range(n):
            plan(C_it)                      # == basis.change_basis(C_it): the same kernels, the same bits

    What differs from ``change_basis``: the arrays handed out by call n are overwritten by call n + 2 (keep a ``.copy()`` of
    what must live longer); ``C`` must be (l, l); single-particle functions are not carried (``spf`` must be None); the
    dtype of the arrays is fixed at planning time (complex ``C`` needs complex arrays: promote before planning).
    ``C_tilde_given=True`` captures the form with explicit bra coefficients: ``plan(C, C_tilde)``."""

    def __init__(self, basis, C_tilde_given=False):
        from . import _lib

        if not is_device_module(basis.np) or is_sharded_module(basis.np):
            raise NotImplementedError("ChangeBasisPlan: device array module only")
        if basis.spf is not None:
            raise NotImplementedError("ChangeBasisPlan does not carry single-particle functions: use change_basis")
        if basis._spin_2_tb is not None and not basis._spin_2_tb_recipe_valid():
            raise NotImplementedError("ChangeBasisPlan: spin_2_tb is held as a tensor somebody wrote into: use change_basis")
        lib = _lib.load()
        self.basis, self.l, self.explicit_bra = basis, basis.l, bool(C_tilde_given)
        l = basis.l
        u = _stage(basis.u).as_subclass(torch.Tensor)
        dt = u.dtype
        self.dtype = dt
        names = [k for k in ("h", "s", "position", "momentum") if getattr(basis, k) is not None]
        mats = [_stage(getattr(basis, k)).as_subclass(torch.Tensor) for k in names]
        if any(m.dtype != dt or tuple(m.shape[-2:]) != (l, l) for m in mats) or tuple(u.shape) != (l, l, l, l):
            raise NotImplementedError("ChangeBasisPlan: h, s, position, momentum and u of one dtype and one basis size")
        recipe = basis._spin_2_tb_recipe if basis._spin_2_tb_recipe_valid() else None
        if recipe is not None:
            rstack = _stage(recipe[0]).as_subclass(torch.Tensor)
            if rstack.dtype != dt:
                raise NotImplementedError("ChangeBasisPlan: the spin matrices of the spin_2_tb recipe in the basis set's dtype")
            mats.append(rstack)
            names.append("__recipe")
        counts = [m.numel() // (l * l) for m in mats]
        nmat = sum(counts)
        dev = u.device
        code = kernels.dtype_code(dt)
        es = 16 if dt.is_complex else 8
        self.C = torch.empty((l, l), dtype=dt, device=dev)
        self.Ct = torch.empty((l, l), dtype=dt, device=dev)
        self._pile = [torch.empty((nmat, l, l), dtype=dt, device=dev) for _ in range(2)]
        self._u = [torch.empty((l, l, l, l), dtype=dt, device=dev) for _ in range(2)]
        self._pile[0].copy_(torch.cat([m.reshape(-1, l, l) for m in mats]))
        self._u[0].copy_(u)
        w2 = kernels.check(lib.qs_transform_two_body_workspace(code, l, l), "workspace query")
        self._work = torch.empty(int(max(w2, nmat * l * l * es)) + 64, dtype=torch.uint8, device=dev)
        # the arrays the basis set is handed, per buffer set (wrapped once)
        self._views = []
        for k in range(2):
            v, at = {}, 0
            for name, m, cnt in zip(names, mats, counts):
                part = self._pile[k][at] if m.dim() == 2 else self._pile[k][at:at + cnt].reshape(m.shape)
                v[name] = wrap(part)
                at += cnt
            v["u"] = wrap(self._u[k])
            self._views.append(v)
        self._anti = recipe[1] if recipe is not None else None
        stream = kernels._stream

        def launch(src, dst):
            if not self.explicit_bra:
                torch.conj_physical(self.C.transpose(0, 1), out=self.Ct) if dt.is_complex else self.Ct.copy_(self.C.transpose(0, 1))
            kernels._ran(lib.qs_transform_one_body(code, self._pile[src].data_ptr(), self.C.data_ptr(), self.Ct.data_ptr(),
                                                   self._pile[dst].data_ptr(), self._work.data_ptr(), self._work.numel(),
                                                   nmat, l, l, stream()), "qs_transform_one_body")
            kernels._ran(lib.qs_transform_two_body(code, self._u[src].data_ptr(), self.C.data_ptr(), self.Ct.data_ptr(),
                                                   self._u[dst].data_ptr(), self._work.data_ptr(), self._work.numel(), l, l,
                                                   stream()), "qs_transform_two_body")

        if dev.index != torch.cuda.current_device():
            raise ValueError("ChangeBasisPlan: make the device that owns the basis set current first (torch.cuda.device)")
        self.C.copy_(torch.eye(l, dtype=dt, device=dev))
        self.Ct.copy_(self.C)
        keep = (self._pile[0].clone(), self._u[0].clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch(0, 1)                                  # first-launch set-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        self._graphs = []
        for src in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launch(src, src ^ 1)
            self._graphs.append(g)
        self._pile[0].copy_(keep[0])
        self._u[0].copy_(keep[1])
        self._cur = 0
        self._bind(0)

    def _bind(self, k):
        b, v = self.basis, self._views[k]
        with torch._C.DisableTorchFunctionSubclass():
            b._h = v["h"]
            if "s" in v:
                b._s = v["s"]
            if "position" in v:
                b._position = v["position"]
            if "momentum" in v:
                b._momentum = v["momentum"]
            b._u = v["u"]
            if "__recipe" in v:
                b._spin_2_tb = None
                b._spin_2_tb_recipe = (v["__recipe"], self._anti)

    def __call__(self, C, C_tilde=None):
        if (C_tilde is not None) != self.explicit_bra:
            raise ValueError("this plan was captured " + ("with" if self.explicit_bra else "without") + " explicit bra coefficients")
        with torch._C.DisableTorchFunctionSubclass():
            # (copy_ would broadcast a vector and drop an imaginary part without a word: checked here)
            for name, src, dst in (("C", C, self.C), ("C_tilde", C_tilde, self.Ct)):
                if src is None:
                    continue
                src = torch.as_tensor(src)
                if tuple(src.shape) != (self.l, self.l):
                    raise ValueError(f"ChangeBasisPlan: {name} must be ({self.l}, {self.l}), got {tuple(src.shape)}")
                if src.is_complex() and not self.dtype.is_complex:
                    raise TypeError(f"ChangeBasisPlan: complex {name} for a plan captured on real arrays (promote the basis set before planning)")
                dst.copy_(src)
            self._graphs[self._cur].replay()
        self._cur ^= 1
        self._bind(self._cur)
        return self.basis
