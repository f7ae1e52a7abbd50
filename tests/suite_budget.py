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
on its own in well under the budget (`pytest tests/test_gpu_configs.py -k <name>` with U2_GPU_SUITE_BUDGET_S=0).

A skipped gate must not look like a green run: the names of the tests the valve skipped are written
to gpurun_out/r06_parity.json (key "budget_skipped"; an empty list when the valve never fired), a banner with the names is printed in
the terminal summary, and the SESSION FAILS (exit status 1) unless U2_ALLOW_BUDGET_SKIPS=1 says the caller has seen it."""
import json
import os
import time
from pathlib import Path

import pytest

LIGHT_RESERVE_S = 150.0       # everything that is not marked: ~400 tests, GPU-bound (profiles/r05_pytest_gpu.log)
DEFAULT_BUDGET_S = 1080.0
PARITY_JSON = Path(__file__).resolve().parents[1] / "gpurun_out" / "r06_parity.json"
_state = {"t0": time.monotonic(), "slow": 1.0, "skipped": [], "heavy_seen": 0}


def record(key, value, path=None):
    """Merge {key: value} into gpurun_out/r06_parity.json (best effort: what is recorded is also asserted by the tests)."""
    p = Path(path or os.environ.get("U2_PARITY_JSON") or PARITY_JSON)
    try:
        p.parent.mkdir(exist_ok=True)
        data = json.loads(p.read_text()) if p.exists() else {}
        data[key] = value
        p.write_text(json.dumps(data, indent=1, sort_keys=True))
    except (OSError, ValueError):
        pass


def skips_allowed():
    return os.environ.get("U2_ALLOW_BUDGET_SKIPS", "0") not in ("", "0")


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
    _state["heavy_seen"] += 1
    elapsed, slow = time.monotonic() - _state["t0"], _state["slow"]
    end = projected_end(elapsed, slow, n, getattr(item, "_u2_later_heavy", 0.0))
    if end > b:
        _state["skipped"].append(item.nodeid)
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


def pytest_sessionfinish(session, exitstatus):
    """The valve's skips are part of the parity record, and a run that skipped a gate is not a green run."""
    if _state["heavy_seen"]:
        record("budget_skipped", sorted(_state["skipped"]))
    if _state["skipped"] and not skips_allowed() and session.exitstatus == 0:
        session.exitstatus = 1


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _state["skipped"]:
        return
    tr = terminalreporter
    tr.section("PARITY GATES SKIPPED BY THE WALL-TIME VALVE (tests/suite_budget.py)", sep="!", red=True, bold=True)
    for name in _state["skipped"]:
        tr.write_line("  SKIPPED GATE: " + name)
    tr.write_line("these full-size oracle comparisons did NOT run: host-heavy tests took %.1f x their nominal time." % _state["slow"])
    tr.write_line("the session FAILS for it (exit status 1)" if not skips_allowed() else
                  "U2_ALLOW_BUDGET_SKIPS=1: the session's exit status is left alone")
    tr.write_line("run them on their own: U2_GPU_SUITE_BUDGET_S=0 python -m pytest -m gpu " + " ".join(_state["skipped"]))
