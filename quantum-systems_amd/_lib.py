"""ctypes binding of ``libqs_amd.so`` (C ABI: ``include/qs_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc,
``--offload-arch=gfx950``).  There is no fallback: if the shared object is
missing or does not export a symbol the header declares, importing the compute
layer raises -- the transforms never silently run anywhere else.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QS_AMD_LIB: development hook -- another build of the SAME library (kernel A/B runs inside one gpurun call)
LIB_PATH = os.environ.get("QS_AMD_LIB") or os.path.join(_HERE, "libqs_amd.so")

QS_F64 = 0
QS_C128 = 1

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/qs_amd.h line by line
SIGNATURES = {
    "qs_abi_version": (c_int, []),
    "qs_error_string": (ctypes.c_char_p, [c_int]),
    "qs_last_hip_error": (ctypes.c_char_p, []),
    "qs_matmul": (c_int, [c_int, c_ptr, c_ptr, c_ptr] + [c_i64] * 10 + [c_int, c_ptr]),
    "qs_transform_two_body_workspace": (c_i64, [c_int, c_i64, c_i64]),
    "qs_transform_two_body": (
        c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "qs_transform_two_body_mixed": (
        c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "qs_transform_two_body_inplace_workspace": (c_i64, [c_int, c_i64, c_i64]),
    "qs_transform_two_body_inplace": (
        c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "qs_transform_two_body_partial_workspace": (c_i64, [c_int, c_i64, c_i64, c_i64]),
    "qs_transform_two_body_partial": (
        c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "qs_transform_one_body": (
        c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "qs_antisymmetrize": (c_int, [c_int, c_ptr, c_ptr, c_i64, c_i64, c_ptr]),
    "qs_spin_expand_two_body": (
        c_int, [c_int, c_int, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr]),
    "qs_spin_expand_two_body_block": (
        c_int, [c_int, c_int, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr]),
    "qs_add_spin_one_body": (c_int, [c_int, c_int, c_ptr, c_ptr, c_i64, c_i64, c_ptr]),
    "qs_spin_squared_two_body": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr]),
    "qs_tdho_coulomb_elements": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "qs_tdho_coulomb_elements_nm": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "qs_comm_unique_id": (c_int, [c_ptr]),
    "qs_comm_init": (c_int, [ctypes.POINTER(c_ptr), c_int, c_int, c_ptr]),
    "qs_comm_destroy": (c_int, [c_ptr]),
    "qs_comm_abort": (c_int, [c_ptr]),
    "qs_comm_rank": (c_int, [c_ptr]),
    "qs_comm_world": (c_int, [c_ptr]),
    "qs_last_comm_error": (ctypes.c_char_p, []),
    "qs_comm_set_option": (c_int, [c_ptr, ctypes.c_char_p, c_i64]),
    "qs_transform_two_body_sharded_workspace": (c_i64, [c_int, c_i64, c_i64, c_int, c_int]),
    "qs_transform_two_body_sharded": (
        c_int, [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr]),
    "qs_sharded_exchange_plan": (c_int, [c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64]),
    "qs_sharded_rows_default_chunk": (c_i64, [c_int, c_i64, c_i64, c_int, c_ptr]),
    "qs_transform_two_body_sharded_rows_out_bytes": (c_i64, [c_int, c_i64, c_i64, c_int, c_int]),
    "qs_transform_two_body_sharded_rows_workspace": (c_i64, [c_int, c_i64, c_i64, c_i64]),
    "qs_transform_two_body_sharded_rows": (
        c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "qs_comm_rows_workspace": (c_i64, [c_ptr, c_int, c_i64, c_i64, c_i64]),
    "qs_sharded_rows_exchange_plan": (c_int, [c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_i64]),
    "qs_sharded_rows_exchange_plan_coalesced": (c_int, [c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_i64]),
    "qs_last_dispatch": (ctypes.c_char_p, []),
    "qs_tuning_set": (c_int, [ctypes.c_char_p, c_i64]),
    "qs_tuning_reset": (c_int, []),
    "qs_probe_mfma_f64": (c_int, [c_ptr, c_i64, c_i64, c_ptr]),
    "qs_probe_stream_copy": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
}

ABI_VERSION = 4


class QsLibraryError(RuntimeError):
    """The HIP extension is missing, stale, or a call into it failed."""


_lib = None


def load():
    """Load (once) and return the ctypes handle with every prototype set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QsLibraryError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the transform path."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime, wrong arch, ...
        raise QsLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise QsLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.qs_abi_version() != ABI_VERSION:
        raise QsLibraryError(
            f"ABI mismatch: library {lib.qs_abi_version()}, binding {ABI_VERSION}"
        )
    _lib = lib
    return lib


def check(code, what):
    """Raise on a negative status from the C ABI."""
    if code >= 0:
        return code
    lib = load()
    msg = lib.qs_error_string(int(code)).decode()
    if code == -5:
        msg += " (" + lib.qs_last_hip_error().decode() + ")"
    if code == -8:
        msg += " (" + lib.qs_last_comm_error().decode() + ")"
    if code in (-1, -3, -6, -7):
        # the same user mistakes raise AssertionError in the reference's
        # setters (basis_set.py:93,103,113); a wrong extent is a ValueError
        raise ValueError(f"{what}: {msg}")
    raise QsLibraryError(f"{what}: {msg}")
