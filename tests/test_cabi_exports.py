"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol ``include/qs_amd.h`` declares; argument validation that
does not touch the GPU behaves; the product path refuses to run without a GPU
(no CPU fallback)."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry

    entry.build()
    from quantum_systems_amd import _lib

    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "qs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in (
        "qs_transform_two_body", "qs_transform_one_body", "qs_antisymmetrize",
        "qs_spin_expand_two_body", "qs_add_spin_one_body", "qs_matmul",
        "qs_transform_two_body_workspace", "qs_transform_two_body_partial",
    ):
        assert must in names


def test_every_declared_symbol_is_exported_and_bound(lib):
    from quantum_systems_amd import _lib

    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} declared in qs_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == set(declared_functions())


def test_abi_version_and_error_strings(lib):
    assert lib.qs_abi_version() == 4
    assert lib.qs_error_string(0) == b"ok"
    for code in range(-7, 0):
        assert len(lib.qs_error_string(code)) > 0


def test_workspace_queries(lib):
    # square: CT + one l^4 buffer (T2 lives in `out`)
    l = 256
    assert lib.qs_transform_two_body_workspace(0, l, l) == 8 * (l * l + l**4)
    assert lib.qs_transform_two_body_workspace(1, l, l) == 16 * (l * l + l**4)
    # shrinking basis needs the extra L^2 M^2 buffer
    L, M = 10, 4
    assert lib.qs_transform_two_body_workspace(0, L, M) == 8 * (L * M + L**3 * M + L * L * M * M)
    assert lib.qs_transform_two_body_workspace(0, 0, 4) < 0
    assert lib.qs_transform_two_body_workspace(7, 4, 4) < 0
    assert lib.qs_transform_two_body_partial_workspace(0, 8, 8, 2) == 8 * (64 + 2 * 8**3 + 2 * 8**3)


def test_argument_validation_without_gpu(lib):
    # these are rejected before any HIP call is made
    assert lib.qs_transform_two_body(0, None, None, None, None, None, 0, 4, 4, None) == -2
    assert lib.qs_transform_two_body(5, 8, 8, 8, 8, 8, 0, 4, 4, None) == -6
    assert lib.qs_transform_two_body(0, 8, 8, 8, 16, 32, 0, -1, 4, None) == -1
    assert lib.qs_transform_two_body(0, 8, 8, 8, 16, 32, 1, 4, 4, None) == -4   # workspace
    assert lib.qs_transform_two_body(0, 8, 8, 8, 8, 32, 1 << 40, 4, 4, None) == -7  # out == u
    assert lib.qs_transform_two_body(0, 12, 8, 8, 16, 32, 1 << 40, 4, 4, None) == -3  # misaligned u
    assert lib.qs_antisymmetrize(0, None, None, 4, 4, None) == -2
    assert lib.qs_spin_expand_two_body(1, 0, 16, 32, 4, 0, 4, 1, None) == -6  # complex -> real
    assert lib.qs_spin_expand_two_body(0, 1, 16, 32, 4, 2, 2, 1, None) == -1  # empty slab


def test_product_path_has_no_cpu_fallback():
    import torch

    from quantum_systems_amd import kernels

    u = torch.zeros(3, 3, 3, 3, dtype=torch.float64)
    C = torch.zeros(3, 3, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="GPU only"):
        kernels.transform_two_body(u, C)
    with pytest.raises(TypeError):
        kernels.transform_two_body(np.zeros((3, 3, 3, 3)), np.zeros((3, 3)))


def test_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "quantum-systems_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "qs_oracle" not in text, f
