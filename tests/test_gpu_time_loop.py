"""The solver-side call pattern of BASELINE.json configs[4]: a
propagation loop that, every time step, assembles h(t), u(t) from the resident
tensors and transforms them with a fresh complex C(t) through
``system.transform_two_body_elements`` -- everything on the device, checked
against the oracle on host copies.  Scaled down to l = 24 (the l = 512 case
needs 8 GPUs)."""

import numpy as np
import pytest
import torch

from oracle import qs_oracle as orc

pytestmark = pytest.mark.gpu


def test_time_loop_stays_on_device_and_matches_oracle():
    import quantum_systems_amd as qsa
    from quantum_systems_amd.array_module import to_host
    from quantum_systems_amd.time_evolution_operators import AdiabaticSwitching, DipoleFieldInteraction

    np.random.seed(3)
    l = 24
    spas = qsa.SpatialOrbitalSystem(4, qsa.RandomBasisSet(l, 2))
    h0, u0, dip = spas.h.copy(), spas.u.copy(), spas.dipole_moment.copy()
    spas.change_module(qsa.hip)
    field = lambda t: 0.1 * np.sin(3 * t)      # noqa: E731
    ramp = lambda t: 1 - np.exp(-2 * t)        # noqa: E731
    spas.set_time_evolution_operator([DipoleFieldInteraction(field), AdiabaticSwitching(ramp)], add_u_0=False)
    rng = np.random.default_rng(1)
    gen = rng.standard_normal((l, l)) + 1j * rng.standard_normal((l, l))
    gen = gen + gen.conj().T                    # Hermitian generator
    w, v = np.linalg.eigh(gen)
    u_ptr = spas.u.data_ptr()
    for t in (0.1, 0.35, 0.8):
        C = (v * np.exp(-1j * w * t)) @ v.conj().T          # unitary C(t)
        dC = qsa.hip.asarray(C)
        h_t, u_t = spas.h_t(t), spas.u_t(t)
        assert isinstance(u_t, torch.Tensor) and u_t.is_cuda
        h_new = spas.transform_one_body_elements(h_t, dC)
        u_new = spas.transform_two_body_elements(u_t, dC)
        assert u_new.is_cuda and u_new.dtype == torch.complex128
        h_ref = orc.transform_one_body(h0 - field(t) * dip[0], C)
        u_ref = orc.transform_two_body(ramp(t) * u0, C)
        assert np.abs(to_host(h_new) - h_ref).max() <= 1e-10 * np.abs(h_ref).max()
        assert np.abs(to_host(u_new) - u_ref).max() <= 1e-10 * np.abs(u_ref).max()
    # the resident tensor was never re-uploaded or modified
    assert spas.u.data_ptr() == u_ptr
    assert np.array_equal(to_host(spas.u), u0)


@pytest.mark.parametrize("l,cplx", [(12, True), (32, False)])
def test_graph_captured_transform_plan(l, cplx):
    # the same launches as transform_two_body, replayed as one HIP graph with coefficients
    # updated in place between replays
    from quantum_systems_amd import kernels as K

    rng = np.random.default_rng(l)
    u = rng.standard_normal((l,) * 4)
    if cplx:
        u = u + 1j * rng.standard_normal((l,) * 4)
    du = torch.from_numpy(u).cuda()

    def coeffs(t):
        a = rng.standard_normal((l, l)) + (1j * rng.standard_normal((l, l)) if cplx else 0)
        q, _ = np.linalg.qr(a)
        return q

    C0 = coeffs(0)
    plan = K.TransformPlan(du, torch.from_numpy(C0).cuda())
    for step in range(3):
        C = coeffs(step)
        plan.C.copy_(torch.from_numpy(C).cuda())
        plan.C_tilde.copy_(torch.from_numpy(np.ascontiguousarray(C.conj().T)).cuda())
        got = plan.replay()
        direct = K.transform_two_body(du, torch.from_numpy(C).cuda())
        assert torch.equal(got, direct)
        ref = orc.transform_two_body(u, C)
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-10 * np.abs(ref).max()
