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


# --------------------------------------------------------------------------
# One-body side of the 2-D dots: orbitals on the polar grid, dipole elements and
# the double-well Hamiltonian.  Radial integrals by symbolic integration, as the
# reference does them (two_dim_helper.py:50-67) -- the product evaluates the
# same integrals in closed form, so the two are independent.
# --------------------------------------------------------------------------


def spf_norm(n, m, mass=1, omega=1):
    """two_dim_helper.py:26-33."""
    from scipy.special import factorial

    return np.sqrt(mass * omega) * np.sqrt(factorial(n) / (np.pi * factorial(n + abs(m))))


def radial_integral(n_p, m_p, n_q, m_q, mass=1, omega=1, order=1):
    """int_0^inf r^(1+order) R_p R_q dr, symbolically (two_dim_helper.py:50-67)."""
    import sympy

    a = sympy.Float(np.sqrt(mass * omega))
    r = sympy.Symbol("r", positive=True)

    def radial(n, m):
        return (a * r) ** abs(m) * sympy.assoc_laguerre(n, abs(m), a**2 * r**2) * sympy.exp(-(a**2) * r**2 / 2)

    return float(sympy.integrate(r * r**order * radial(n_p, m_p) * radial(n_q, m_q), (r, 0, sympy.oo)))


def spf_table(l, radius, theta, mass=1, omega=1):
    """Orbitals on meshgrid(radius, theta) (two_dim_ho.py:96-108, two_dim_helper.py:16-50)."""
    from scipy.special import assoc_laguerre

    R, T = np.meshgrid(radius, theta)
    a = np.sqrt(mass * omega)
    out = np.zeros((l,) + R.shape, dtype=np.complex128)
    for p in range(l):
        n, m = indices_nm(p)
        out[p] = (spf_norm(n, m, mass, omega) * np.exp(1j * m * T) * (a * R) ** abs(m)
                  * assoc_laguerre(a**2 * R**2, n, abs(m)) * np.exp(-(a**2) * R**2 / 2.0))
    return out


def position_integrals(l, mass=1, omega=1):
    """<p|x|q>, <p|y|q> (two_dim_ho.py:113-139; angular factors two_dim_helper.py:78-89)."""
    pos = np.zeros((2, l, l), dtype=np.complex128)
    for p in range(l):
        n_p, m_p = indices_nm(p)
        for q in range(l):
            n_q, m_q = indices_nm(q)
            if abs(m_p - m_q) != 1:
                continue
            amp = spf_norm(n_p, m_p, mass, omega) * spf_norm(n_q, m_q, mass, omega) * radial_integral(
                n_p, m_p, n_q, m_q, mass, omega)
            pos[0, p, q] = amp * np.pi
            pos[1, p, q] = amp * (-(m_p - m_q) * 1j * np.pi)
    return pos


def double_well_one_body(l, omega, mass, barrier_strength, axis=0):
    """two_dim_helper.py:304-339 (angular factors :92-105)."""
    h = np.zeros((l, l), dtype=np.complex128)
    for p in range(l):
        n_p, m_p = indices_nm(p)
        h[p, p] += omega * shell_energy(n_p, m_p) + omega**2 * barrier_strength**2 / 8.0
        for q in range(l):
            n_q, m_q = indices_nm(q)
            d = m_p - m_q
            if abs(d) % 2 == 1:
                continue
            ang = 4 / (1 - d**2)
            if axis == 0 and (abs(d) // 2) % 2 == 1:
                ang = -ang
            h[p, q] -= (0.5 * omega**2 * barrier_strength * spf_norm(n_p, m_p, mass, omega)
                        * spf_norm(n_q, m_q, mass, omega) * radial_integral(n_p, m_p, n_q, m_q, mass, omega) * ang)
    return h


def coulomb_element_nm(nm, p, q, r, s):
    """One element for an explicit orbital table (two_dim_helper.py:284-301)."""
    (n_p, m_p), (n_q, m_q), (n_r, m_r), (n_s, m_s) = nm[p], nm[q], nm[r], nm[s]
    return coulomb_ho(int(n_p), int(m_p), int(n_q), int(m_q), int(n_r), int(m_r), int(n_s), int(m_s))


def level_table(n_array, m_array, omega_c=0.0, omega=1.0):
    """Rows (n, m, E) of the reference's frame: sorted by E = omega (2n+|m|+1) - omega_c m / 2,
    ties by m (two_dim_helper.py:271-272, :380-392)."""
    rows = [(float(n), float(m), omega * (2 * n + abs(m) + 1) - (omega_c * m) / 2) for n in n_array for m in m_array]
    rows.sort(key=lambda t: (t[2], t[1]))
    return np.array([(int(n), int(m)) for n, m, _ in rows]), np.array([e for _, _, e in rows])
