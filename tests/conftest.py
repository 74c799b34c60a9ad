"""pytest configuration: the `gpu` marker and shared paths/fixtures."""
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_search():
    return json.loads((GOLDEN / "golden_search.json").read_text())["cases"]


@pytest.fixture(scope="session")
def golden_host():
    return json.loads((GOLDEN / "golden_host.json").read_text())
