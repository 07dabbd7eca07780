import os
import sys
from pathlib import Path

import pytest

# torch bundles its own HIP runtime.  If libcattus_hip.so (linked against /opt/rocm) initialises the GPU first,
# a later `import torch` in the same process finds "No HIP GPUs": load torch's runtime first, once, for every
# test session, whatever subset of the tests runs (bench.py and the scripts import torch first as well).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before the HIP runtime initialises (cattus_amd/__init__.py)
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover - the CPU tests do not need it
    pass

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
