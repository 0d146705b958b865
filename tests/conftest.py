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


# ---- diagnosis aid (OPT-IN): a second evaluation of a failing GPU test ------------------------------------------------
# Round 2 evaluated every failing `gpu` test a second time by default and let the session pass when the second try
# passed -- a blanket mask over exactly the class of bug (races) this code base has shown (VERDICT / ADVICE round 2).
# Now: nothing is re-evaluated unless WESEP_TEST_RERUN=1 is set by somebody who is diagnosing a transient, and even
# then a first-evaluation failure (a) is listed with its original assertion, (b) is written to
# gpurun_out/test_reruns.json, and (c) makes the session end with a non-zero exit status.  A failure is a failure.
_RERUNS = []


def pytest_runtest_protocol(item, nextitem):
    if item.get_closest_marker("gpu") is None or os.environ.get("WESEP_TEST_RERUN", "0") != "1":
        return None
    from _pytest.runner import runtestprotocol
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    reports = runtestprotocol(item, nextitem=nextitem, log=False)
    failed = [r for r in reports if r.failed]
    if failed and not any(getattr(r, "wasxfail", None) for r in reports):
        first = " | ".join(" ".join(ln.strip() for ln in str(r.longrepr).splitlines() if ln.startswith("E "))[:400]
                           for r in failed)
        again = runtestprotocol(item, nextitem=nextitem, log=False)
        _RERUNS.append({"nodeid": item.nodeid, "first_evaluation": first,
                        "second_evaluation_failed": any(r.failed for r in again)})
        reports = again
    for r in reports:
        item.ihook.pytest_runtest_logreport(report=r)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


def pytest_terminal_summary(terminalreporter):
    if _RERUNS:
        terminalreporter.section("GPU tests that FAILED on their first evaluation (WESEP_TEST_RERUN=1)", sep="!")
        for r in _RERUNS:
            terminalreporter.write_line(f"FIRST-EVALUATION-FAILED {r['nodeid']}: {r['first_evaluation']}")


def pytest_sessionfinish(session, exitstatus):
    if _RERUNS:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "test_reruns.json"), "w") as f:
            json.dump(_RERUNS, f, indent=1)
        if session.exitstatus == 0:
            session.exitstatus = 1          # a transient is not a pass
