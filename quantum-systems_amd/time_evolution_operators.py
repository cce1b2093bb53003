"""Time-dependent terms of the Hamiltonian (reference:
quantum_systems/time_evolution_operators/operator.py:1-221).

A ``QuantumSystem`` sums ``op.h_t(t)`` / ``op.u_t(t)`` of its operators on top
of the static ``h`` / ``u`` (system.py:189-215).  The operators are
array-module generic: with the device module every term is an elementwise
operation on tensors already resident in HBM, so a propagation loop can
evaluate ``h_t(t)``, ``u_t(t)`` and ``transform_two_body_elements(u_t, C(t))``
each step without leaving the GPU.
"""

import abc


def _as_function_of_time(value):
    """Constants are promoted to constant functions of t."""
    return value if callable(value) else (lambda t, _v=value: _v)


class TimeEvolutionOperator(metaclass=abc.ABCMeta):
    """Base class: contributes nothing to either part of the Hamiltonian
    (operator.py:4-85)."""

    is_one_body_operator = False
    is_two_body_operator = False

    def set_system(self, system):
        """Bind to the system whose elements the operator reads; returns
        ``self`` so ``set_time_evolution_operator`` can collect the result."""
        self._system = system
        return self

    def h_t(self, current_time):
        return 0

    def u_t(self, current_time):
        return 0


class DipoleFieldInteraction(TimeEvolutionOperator):
    """Semi-classical laser field in the dipole approximation
    (operator.py:87-172).

    length gauge:   h(t) = -E(t) * eps(t) . d          (d = dipole moment)
    velocity gauge: h(t) =  A(t) * eps(t) . p  [+ A(t)^2 / 2 * 1]

    ``field_strength`` and ``polarization_vector`` may be constants or functions
    of time; the polarisation defaults to the x axis."""

    is_one_body_operator = True

    def __init__(self, field_strength, polarization_vector=None, gauge="length", quadratic_term=True):
        assert gauge in ["length", "velocity"], "gauge must be either length or velocity."
        self._length_gauge = gauge == "length"
        self._quadratic_term = quadratic_term
        self._field_strength = field_strength
        self._polarization = polarization_vector

    def h_t(self, current_time):
        np = self._system.np
        self._field_strength = _as_function_of_time(self._field_strength)
        if self._polarization is None:
            e_x = np.zeros(self._system.dipole_moment.shape[0])
            e_x[0] = 1
            self._polarization = e_x
        self._polarization = _as_function_of_time(self._polarization)
        strength = self._field_strength(current_time)
        direction = self._polarization(current_time)
        if self._length_gauge:
            return -strength * np.tensordot(direction, self._system.dipole_moment, axes=(0, 0))
        H_t = strength * np.tensordot(direction, self._system.momentum, axes=(0, 0))
        if self._quadratic_term:
            H_t += 0.5 * strength**2 * np.eye(self._system.l)
        return H_t


class AdiabaticSwitching(TimeEvolutionOperator):
    """Two-body interaction scaled by a switching function:
    u(t) = f(t) * u (operator.py:175-196)."""

    is_two_body_operator = True

    def __init__(self, switching_function):
        self._switching_function = _as_function_of_time(switching_function)

    def u_t(self, current_time):
        return self._switching_function(current_time) * self._system.u


class CustomOneBodyOperator(TimeEvolutionOperator):
    """h(t) = w(t) * O for a user-supplied matrix O (operator.py:199-221)."""

    is_one_body_operator = True

    def __init__(self, weight, operator):
        self._weight = weight
        self._operator = operator

    def h_t(self, current_time):
        self._weight = _as_function_of_time(self._weight)
        return self._weight(current_time) * self._operator
