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


@pytest.fixture
def lib_option():
    """set(name, value): override one of the library's VQA_* knobs (vqa_set_option) for this test; every knob touched
    goes back to what the environment says afterwards.  (The library reads the environment once per knob, so
    monkeypatch.setenv would not reach it.)"""
    from vqa_playground_pytorch_amd import _lib
    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.set_option(name, value)

    yield set_
    for name in touched:
        _lib.set_option(name, os.environ.get(name))


# ---- measured errors in the test output --------------------------------------------------------------------------------------
# Parity tests that hold an error under a bar also SAY what they measured: `measured(name, value, bar)` collects it and the
# terminal summary prints one line each (also under -q, where a passing test's stdout is not shown), so a regression that
# stays inside the bar is visible in the driver's log.  The same lines go to gpurun_out/measured.jsonl.
_MEASURED = []


@pytest.fixture
def measured(request):
    def record(name, value, bar=None, note=""):
        _MEASURED.append((request.node.nodeid.split("::")[-1], str(name), float(value), None if bar is None else float(bar), note))
    return record


def pytest_terminal_summary(terminalreporter):
    if not _MEASURED:
        return
    import json
    terminalreporter.section("measured errors (value / bar)")
    rows = []
    for test, name, value, bar, note in _MEASURED:
        terminalreporter.write_line("%-78s %-28s %.3e%s %s" % (test[:78], name[:28], value, "" if bar is None else " / %.1e" % bar, note))
        rows.append({"test": test, "name": name, "value": value, "bar": bar, "note": note})
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "measured.jsonl"), "a") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")
    except OSError:
        pass
