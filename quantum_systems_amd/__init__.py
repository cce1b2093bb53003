"""Import alias: ``import quantum_systems_amd`` -> the ``quantum-systems_amd``
package directory (whose name is not a Python identifier)."""

import importlib
import sys

_real = importlib.import_module("quantum-systems_amd")
_prefix = "quantum-systems_amd"
for _name, _mod in list(sys.modules.items()):
    if _name == _prefix or _name.startswith(_prefix + "."):
        sys.modules[__name__ + _name[len(_prefix):]] = _mod
