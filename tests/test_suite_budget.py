"""The wall-time valve of the GPU suite (tests/suite_budget.py) on the CPU: the projection, and the hooks in a real pytest run
(a child process over a throw-away test file: a host-heavy test that overruns its nominal time makes the later ones skip,
nothing is skipped on a fast host, 0 switches the valve off)."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import suite_budget as SB

TESTS = Path(__file__).resolve().parent


def test_projection():
    # fast host: elapsed + the host-heavy tests still to come + the light tests' reserve
    assert SB.projected_end(100.0, 1.0, 30.0, 170.0) == 100.0 + 200.0 + SB.LIGHT_RESERVE_S
    # slow host: host-heavy time scales with the measured factor, the (GPU-bound) light tests by at most 1.5
    assert SB.projected_end(400.0, 2.5, 62.0, 150.0) == 400.0 + 2.5 * 212.0 + 1.5 * SB.LIGHT_RESERVE_S
    # the suite as committed (nominal host-heavy seconds 39 + 62 + 65 + 15 + 28 + 39, ~280 s of everything else): a fast
    # host ends near 500 s, the slowest host seen in the pool (2.5 x) below the default budget
    heavy = 39 + 62 + 65 + 15 + 28 + 39
    assert SB.projected_end(0.0, 1.0, 0.0, heavy) < 0.5 * SB.DEFAULT_BUDGET_S
    assert SB.projected_end(0.0, 2.5, 0.0, heavy) + 130 * 1.5 < SB.DEFAULT_BUDGET_S


_CHILD = '''
import time
import pytest

@pytest.mark.host_heavy(1)
def test_a():
    time.sleep({sleep})

def test_light():
    pass

@pytest.mark.host_heavy(2)
def test_b():
    pass

@pytest.mark.host_heavy(1)
def test_c():
    pass
'''


def _run(tmp_path, sleep, budget, allow=None):
    d = tmp_path / f"case_{sleep}_{budget}_{allow}".replace(".", "_")
    d.mkdir()
    (d / "conftest.py").write_text(textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)
        from suite_budget import (pytest_collection_modifyitems, pytest_runtest_call, pytest_runtest_setup,  # noqa
                                  pytest_sessionfinish, pytest_terminal_summary)

        def pytest_configure(config):
            config.addinivalue_line("markers", "host_heavy(nominal_seconds): see tests/suite_budget.py")
    ''' % str(TESTS)))
    (d / "test_child.py").write_text(_CHILD.format(sleep=sleep))
    env = dict(os.environ, U2_GPU_SUITE_BUDGET_S=str(budget), U2_PARITY_JSON=str(d / "parity.json"))
    env.pop("U2_ALLOW_BUDGET_SKIPS", None)
    if allow is not None:
        env["U2_ALLOW_BUDGET_SKIPS"] = allow
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-rs", "-p", "no:cacheprovider", str(d)], env=env, cwd=str(d),
                       capture_output=True, text=True, timeout=120)
    rec = json.loads((d / "parity.json").read_text()) if (d / "parity.json").exists() else None
    return r.returncode, r.stdout + r.stderr, rec


def test_hooks_in_a_child_run(tmp_path):
    reserve = SB.LIGHT_RESERVE_S
    # fast host: test_a takes its nominal second at most -> projection = ~0 + (1 + 2 + 1) + reserve < budget: all four run,
    # and the parity record says that nothing was skipped
    rc, out, rec = _run(tmp_path, 0.0, reserve + 30)
    assert rc == 0 and "4 passed" in out and rec == {"budget_skipped": []}, (out, rec)
    # test_a overruns 3 x: before test_b 3 + 3 x (2 + 1) + 1.5 x reserve, before test_c 3 + 3 x 1 + 1.5 x reserve: both past the
    # budget -> both skip, LOUDLY: a banner with the names, the names in the parity record, and the session fails
    rc, out, rec = _run(tmp_path, 3.0, 1.5 * reserve + 4)
    assert "2 passed, 2 skipped" in out and "host too slow" in out, out
    assert rc == 1 and "PARITY GATES SKIPPED" in out and "SKIPPED GATE: test_child.py::test_b" in out, (rc, out)
    assert rec == {"budget_skipped": ["test_child.py::test_b", "test_child.py::test_c"]}, rec
    # ... unless the caller says it has seen them; the record and the banner stay
    rc, out, rec = _run(tmp_path, 3.0, 1.5 * reserve + 4, allow="1")
    assert rc == 0 and "2 passed, 2 skipped" in out and "PARITY GATES SKIPPED" in out and len(rec["budget_skipped"]) == 2, (rc, out, rec)
    # the valve switched off
    rc, out, rec = _run(tmp_path, 3.0, 0)
    assert rc == 0 and "4 passed" in out, out
