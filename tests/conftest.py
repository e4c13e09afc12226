import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def rel_state_error(x, ref):
    """max_i |x_i - ref_i|_inf / max_i |ref_i|_inf over the xyz columns."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)[:, :3]
    ref = np.asarray(ref, dtype=np.float64)[:, :3]
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-300))
