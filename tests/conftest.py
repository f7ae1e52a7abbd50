import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


from suite_budget import (pytest_collection_modifyitems, pytest_runtest_call, pytest_runtest_setup,  # noqa: E402,F401  (hooks)
                          pytest_sessionfinish, pytest_terminal_summary)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "host_heavy(nominal_seconds): mostly HOST time (full-size CPU oracle); skipped when the suite's "
                                       "wall-time budget would not hold on a slow host (tests/suite_budget.py)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
