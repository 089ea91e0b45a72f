import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): compile the HIP library for gfx950, as
    `__graft_entry__.build()` does - also when a source is newer than the library, so that the suite never runs
    against a binary of code that has since changed (`_build.build_library()` checks the time stamps).  The product
    itself never builds or falls back at run time."""
    from acmpc_amd import _build
    _build.build_library()


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by importing the reference (tests/golden/gen_golden.py)."""
    return np.load(os.path.join(GOLDEN_DIR, "reference_ingredients.npz"))


@pytest.fixture(scope="session")
def golden_cases(golden):
    return [str(c) for c in golden["cases"]]
