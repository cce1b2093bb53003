"""MI355X-native integral basis transformation behind the quantum-systems API.

Drop-in for the hot path of HyQD/quantum-systems: the same ``BasisSet`` /
``QuantumSystem`` classes, with the four-index transform, one-body transforms,
anti-symmetrisation and spin doubling executed by hand-written HIP kernels
(``libqs_amd.so``, C ABI in ``include/qs_amd.h``).  Pass ``array_module.hip``
(exported here as ``hip``) wherever the reference takes ``np=`` to keep the
tensors resident in HBM.

The directory is named ``quantum-systems_amd`` after the upstream project;
import it as ``quantum_systems_amd`` (the sibling shim package aliases it).
"""

from . import _lib, kernels, sharded  # noqa: F401
from .array_module import DeviceArray, DeviceModule, hip
from .sharded_module import ShardedDeviceModule, ShardedTensor4
from .basis_set import BasisSet, ChangeBasisPlan
from .custom_system import construct_custom_system, setup_basis_set
from .general_orbital_system import GeneralOrbitalSystem
from .one_dim_qd import ODQD
from .random_basis import RandomBasisSet
from .sinc_dvr import ODSincDVR
from .spatial_orbital_system import SpatialOrbitalSystem
from .system import QuantumSystem
from .two_dim_ho import TwoDimensionalDoubleWell, TwoDimensionalHarmonicOscillator, TwoDimHarmonicOscB

__all__ = [
    "BasisSet", "RandomBasisSet", "QuantumSystem", "SpatialOrbitalSystem",
    "GeneralOrbitalSystem", "setup_basis_set", "construct_custom_system",
    "TwoDimensionalHarmonicOscillator", "TwoDimensionalDoubleWell", "TwoDimHarmonicOscB", "ODQD", "ODSincDVR",
    "ChangeBasisPlan", "hip", "DeviceModule", "DeviceArray", "ShardedDeviceModule", "ShardedTensor4", "kernels", "sharded",
]
