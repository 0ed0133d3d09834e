import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """A fresh checkout has no libfeinsum_hip.so (built artefacts are not in the history):
    compile it once (hipcc cross-compiles gfx950 without a GPU).  Stale sources rebuild too."""
    import __graft_entry__

    return __graft_entry__.build_library()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
