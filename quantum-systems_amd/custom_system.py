"""User-supplied matrix elements -> ``BasisSet`` / system
(reference: quantum_systems/custom_system.py:11-95; the PySCF factories of
that file need pyscf and are outside the transform path)."""

from .basis_set import BasisSet
from .general_orbital_system import GeneralOrbitalSystem
from .spatial_orbital_system import SpatialOrbitalSystem


def setup_basis_set(n, l, s, h, u, dim=3, particle_charge=-1, np=None, includes_spin=False,
                    anti_symmetrized_u=False, **kwargs):
    """Fill a ``BasisSet`` with the given overlap ``s``, one-body ``h`` and
    two-body ``u`` elements; ``position``, ``momentum`` and
    ``nuclear_repulsion_energy`` are picked up from ``kwargs``
    (custom_system.py:11-50)."""
    bs = BasisSet(l, dim=dim, np=np, includes_spin=includes_spin,
                  anti_symmetrized_u=anti_symmetrized_u)
    bs.h, bs.u, bs.s = h, u, s
    bs.particle_charge = particle_charge
    for name in ("position", "momentum", "nuclear_repulsion_energy"):
        if name in kwargs:
            setattr(bs, name, kwargs[name])
    return bs


def construct_custom_system(n, l, s, h, u, dim=3, particle_charge=-1, np=None,
                            includes_spin=False, anti_symmetrized_u=False,
                            system_type="general", **kwargs):
    """``GeneralOrbitalSystem`` (``system_type="general"``) or
    ``SpatialOrbitalSystem`` (``"spatial"``) from raw elements; a coefficient
    matrix passed as ``C=`` is applied with ``change_basis``
    (custom_system.py:53-95)."""
    bs = setup_basis_set(n, l, s, h, u, dim, particle_charge, np, includes_spin,
                         anti_symmetrized_u, **kwargs)
    kind = system_type.lower()
    if kind == "general":
        system = GeneralOrbitalSystem(n, bs)
    elif kind == "spatial":
        system = SpatialOrbitalSystem(n, bs)
    else:
        raise NotImplementedError(f"System type: {system_type} is not supported!")
    if "C" in kwargs:
        system.change_basis(kwargs["C"])
    return system
