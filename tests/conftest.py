"""Shared pytest configuration.

* registers the ``gpu`` marker (tests that need a real MI355X);
* puts the repo root on ``sys.path`` so ``oracle`` (checker only) and the
  package are importable without installation;
* ``golden(name)`` loads a committed fixture from ``tests/golden``.
"""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)"
    )


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _automatic_kernel_choice():
    """Tuning knobs are thread-local library state; whatever a test sets is
    undone after it (also when it fails), so one test cannot change the kernel
    dispatch of the next."""
    yield
    mod = sys.modules.get("quantum_systems_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod._lib.qs_tuning_reset()
