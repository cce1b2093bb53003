"""MI355X-native integral basis transformation behind the quantum-systems API.

The directory is named ``quantum-systems_amd`` (with the hyphen of the upstream
project); import it as ``quantum_systems_amd`` -- the sibling shim package of
that name aliases this one -- or via ``importlib.import_module``.
"""

from . import _lib, kernels  # noqa: F401
