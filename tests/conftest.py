import os
import sys

import pytest

# torch bundles its own HIP runtime; when it is going to be used in the same process as
# liblongbow_gpu.so it has to be loaded FIRST, otherwise its later initialisation finds the
# device already claimed by the system runtime ("No HIP GPUs are available").
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)["cases"]
