import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    import os

    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")
    # a kernel that fits fewer blocks on a CU than its launch geometry assumes FAILS under test (outside it is a
    # warning line in fe_kernel_resources and a slower launch): a compiler regression must not pass at half the speed
    os.environ.setdefault("FEINSUM_STRICT_RESIDENCY", "1")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """A fresh checkout has no libfeinsum_hip.so (built artefacts are not in the history):
    compile it once (hipcc cross-compiles gfx950 without a GPU).  Stale sources rebuild too."""
    import __graft_entry__

    return __graft_entry__.build_library()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
