import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/libpds_oracle.so if needed."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    import json

    return json.loads((ROOT / "tests" / "golden" / "reference_notebook.json").read_text())
