"""ctypes front of oracle/coulomb_ho.c -- TEST INFRASTRUCTURE ONLY.

CPU oracle of the 2-D harmonic-oscillator Coulomb elements (reference:
quantum_systems/quantum_dots/two_dim/coulomb_elements.py:6-152,
two_dim_helper.py:132-182, :250-268).  Built by ``__graft_entry__.build()``
(or on first use) with gcc into ``oracle/_build/``; pinned by
``tests/test_oracle_golden.py`` against the reference's own table and against
elements computed by the reference code.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "coulomb_ho.c")
_OUT = os.path.join(_HERE, "_build", "libtdho_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", _OUT, _SRC, "-lm"], check=True)
    return _OUT


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.tdho_coulomb_ho.restype = ctypes.c_double
        lib.tdho_coulomb_ho.argtypes = [ctypes.c_int] * 8
        lib.tdho_indices_nm.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.tdho_coulomb_elements.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib = lib
    return _lib


def indices_nm(p):
    """(n, m) of orbital p -- two_dim_helper.py:132-166."""
    n, m = ctypes.c_int(), ctypes.c_int()
    _load().tdho_indices_nm(int(p), ctypes.byref(n), ctypes.byref(m))
    return n.value, m.value


def shell_energy(n, m):
    """two_dim_helper.py:169-171."""
    return 2 * n + abs(m) + 1


def one_body_elements(l):
    """diag of shell energies -- two_dim_helper.py:174-182."""
    h = np.zeros((l, l))
    for p in range(l):
        h[p, p] = shell_energy(*indices_nm(p))
    return h


def coulomb_ho(n_i, m_i, n_j, m_j, n_l, m_l, n_k, m_k):
    return _load().tdho_coulomb_ho(n_i, m_i, n_j, m_j, n_l, m_l, n_k, m_k)


def coulomb_elements(l, p_lo=0, p_hi=None):
    """u[p,q,r,s] for p in [p_lo, p_hi) -- two_dim_helper.py:250-268."""
    p_hi = l if p_hi is None else p_hi
    out = np.empty((p_hi - p_lo, l, l, l))
    _load().tdho_coulomb_elements(l, p_lo, p_hi, out.ctypes.data)
    return out
