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
