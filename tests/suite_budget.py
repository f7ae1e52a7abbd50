"""A wall-time valve for the GPU suite (pytest hooks; tests/conftest.py imports them).

Several GPU parity tests spend most of their time on the HOST: they run the CPU oracle at the benchmark's full size next to
the HIP path (30 to 60 s of host matmuls each on a pool box with a fast host).  The hosts of the pool differ by 2.5 x, the
round-end run of `pytest -m gpu` has a fixed limit, and a run killed at the limit reports nothing at all.  So those tests
carry `@pytest.mark.host_heavy(nominal_seconds)` and, before each, the time the REST of the suite is going to take is
projected from how slow this host has turned out to be so far:

    projected end = elapsed + slow x (this test + the host-heavy tests after it) + the light tests' reserve x min(slow, 1.5)
    slow          = the largest (measured / nominal) duration among the host-heavy tests already run (at least 1)

A test whose projection passes U2_GPU_SUITE_BUDGET_S (default 1080 s; 0 switches the valve off) is SKIPPED with a reason that
says so -- on a host of normal speed the projection stays near 500 s and nothing is ever skipped; every skipped test runs
on its own in well under the budget (`pytest tests/test_gpu_configs.py -k <name>` with U2_GPU_SUITE_BUDGET_S=0)."""
import os
import time

import pytest

LIGHT_RESERVE_S = 150.0       # everything that is not marked: ~400 tests, GPU-bound (profiles/r05_pytest_gpu.log)
DEFAULT_BUDGET_S = 1080.0
_state = {"t0": time.monotonic(), "slow": 1.0}


def projected_end(elapsed, slow, nominal, later_heavy, reserve=LIGHT_RESERVE_S):
    return elapsed + slow * (nominal + later_heavy) + reserve * min(slow, 1.5)


def budget_s():
    try:
        return float(os.environ.get("U2_GPU_SUITE_BUDGET_S", DEFAULT_BUDGET_S))
    except ValueError:
        return DEFAULT_BUDGET_S


def _nominal(item):
    mk = item.get_closest_marker("host_heavy")
    if mk is None or not mk.args:
        return None
    try:
        return max(float(mk.args[0]), 1.0)
    except (TypeError, ValueError):
        return None


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    later = 0.0
    for it in reversed(items):
        n = _nominal(it)
        if n is not None:
            it._u2_later_heavy = later
            later += n


def pytest_runtest_setup(item):
    n = _nominal(item)
    b = budget_s()
    if n is None or b <= 0:
        return
    elapsed, slow = time.monotonic() - _state["t0"], _state["slow"]
    end = projected_end(elapsed, slow, n, getattr(item, "_u2_later_heavy", 0.0))
    if end > b:
        pytest.skip(f"host too slow for this full-size oracle run inside the suite's wall-time budget: {elapsed:.0f} s elapsed, host-heavy "
                    f"tests have run {slow:.1f} x their nominal time, projected end {end:.0f} s > U2_GPU_SUITE_BUDGET_S = {b:.0f} s "
                    "(tests/suite_budget.py; run it on its own with U2_GPU_SUITE_BUDGET_S=0)")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    t = time.monotonic()
    yield
    n = _nominal(item)
    if n is not None:
        _state["slow"] = max(_state["slow"], (time.monotonic() - t) / n)
