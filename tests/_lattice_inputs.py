"""Inputs given by a closed integer formula, for golden vectors at sizes whose tensors are too large to commit.

`tests/golden/make_golden.py` (case_mid_size_sampled) feeds these to the REFERENCE's own
`BasisSet.transform_two_body_elements` and stores sampled outputs; the tests rebuild the same inputs -- with numpy for the
oracle, with torch on the device for the HIP path -- from nothing but (l, salt).  Every value is k / 32768 with an integer
-32768 <= k < 32768 out of 64-bit integer arithmetic that never overflows, so numpy, torch-CPU and torch-on-the-GPU agree
bit for bit on every platform (no random generator, no libm call).
"""

import numpy as np

_A, _B, _C, _D, _S = 73856093, 19349663, 83492791, 2654435761, 40503
_M1, _M2 = 1540483477, 668265263          # < 2^31: products with a 32-bit value stay below 2^63
_MASK = 0xFFFFFFFF


def _mix(h):
    # h: non-negative int64 (array or tensor) -> 32 well-mixed bits; the same operators exist for numpy and torch
    h = h & _MASK
    h = h ^ (h >> 15)
    h = (h * _M1) & _MASK
    h = h ^ (h >> 13)
    h = (h * _M2) & _MASK
    h = h ^ (h >> 16)
    return h


def _values(h, cplx, to_float, as_complex):
    re = to_float((h & 0xFFFF) - 32768) / 32768.0
    if not cplx:
        return re
    im = to_float(((h >> 16) & 0xFFFF) - 32768) / 32768.0
    return as_complex(re, im)


def tensor_np(l, salt, cplx=False):
    """u[p, q, r, s] of the formula as a numpy array (built one leading index at a time: temporaries stay at l^3)."""
    out = np.empty((l,) * 4, dtype=np.complex128 if cplx else np.float64)
    q = np.arange(l, dtype=np.int64).reshape(l, 1, 1)
    r = np.arange(l, dtype=np.int64).reshape(1, l, 1)
    s = np.arange(l, dtype=np.int64).reshape(1, 1, l)
    rest = q * _B + r * _C + s * _D + salt * _S
    for p in range(l):
        h = _mix(rest + p * _A)
        out[p] = _values(h, cplx, lambda a: a.astype(np.float64), lambda re, im: re + 1j * im)
    return out


def tensor_torch(l, salt, cplx=False, device="cpu"):
    """The same tensor built by torch on `device`."""
    import torch

    out = torch.empty((l,) * 4, dtype=torch.complex128 if cplx else torch.float64, device=device)
    ar = torch.arange(l, dtype=torch.int64, device=device)
    rest = ar.view(l, 1, 1) * _B + ar.view(1, l, 1) * _C + ar.view(1, 1, l) * _D + salt * _S
    for p in range(l):
        h = _mix(rest + p * _A)
        out[p] = _values(h, cplx, lambda a: a.to(torch.float64), torch.complex)
    return out


def matrix_np(rows, cols, salt, cplx=False):
    """A rows x cols coefficient matrix of the formula, scaled by 1 / sqrt(rows) (numpy; the tests upload it)."""
    i = np.arange(rows, dtype=np.int64).reshape(rows, 1)
    j = np.arange(cols, dtype=np.int64).reshape(1, cols)
    h = _mix(i * _C + j * _A + salt * _S + 977)
    m = _values(h, cplx, lambda a: a.astype(np.float64), lambda re, im: re + 1j * im)
    return m / np.sqrt(float(rows))


def sample_positions(m, count, salt):
    """`count` index quadruples into an m^4 result: the corners and tile edges first, then positions of the formula."""
    edge = sorted({0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, m // 2, m - 2, m - 1} & set(range(m)))
    pos = [(a, a, a, a) for a in edge] + [(0, m - 1, a, m - 1 - a) for a in edge] + [(m - 1, a, 0, a) for a in edge]
    k = np.arange(count, dtype=np.int64)
    cols = [(_mix(k * _A + c * _D + salt * _S + 31) % m) for c in range(4)]
    pos += list(zip(*(c.tolist() for c in cols)))
    return np.array(pos[:count], dtype=np.int64)


# (name, L, M, tensor complex, coefficients complex, salt): the sizes of tests/golden/mid_size_sampled.npz -- one per kernel
# family that serves bases beyond the small fixtures (streamed quads, strip tiles of one and several blocks, 256-wide tiles,
# the complex strip form, real tensor x complex coefficients)
CASES = [
    ("f64_78", 78, 78, False, False, 1),
    ("f64_100", 100, 100, False, False, 2),
    ("f64_130", 130, 130, False, False, 3),
    ("f64_150_to_120", 150, 120, False, False, 4),
    ("f64_180", 180, 180, False, False, 5),
    ("c128_72", 72, 72, True, True, 6),
    ("c128_100", 100, 100, True, True, 7),
    ("c128_140", 140, 140, True, True, 8),
    ("mixed_90", 90, 90, False, True, 9),
]
N_SAMPLES = 1536


def case_inputs_np(L, M, ucplx, ccplx, salt):
    """(C, C_tilde) of a case as numpy arrays (C_tilde is NOT C's adjoint: both matrices are general)."""
    return matrix_np(L, M, 2 * salt, ccplx), matrix_np(L, M, 2 * salt + 1, ccplx).T.copy()
