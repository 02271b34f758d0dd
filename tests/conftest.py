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


# FEMO_CALL_AUDIT=<file>: count the calls of every C-ABI entry point during the session (tests only: the product's ctypes table is
# wrapped, not changed) and write {symbol: calls} -- the coverage audit of include/femo_hip.h behind DESIGN.md section 2.
def pytest_sessionstart(session):
    out = os.environ.get("FEMO_CALL_AUDIT")
    if not out:
        return
    from femo_alpha_amd import _lib
    lib = _lib.load()
    counts = {name: 0 for name in _lib.SIGNATURES}

    def wrap(name, fn):
        def call(*a):
            counts[name] += 1
            return fn(*a)
        return call
    for name in _lib.SIGNATURES:
        setattr(lib, name, wrap(name, getattr(lib, name)))
    session.config._femo_call_audit = (out, counts)


def pytest_sessionfinish(session, exitstatus):
    audit = getattr(session.config, "_femo_call_audit", None)
    if audit:
        import json
        with open(audit[0], "w") as f:
            json.dump(audit[1], f, indent=1, sort_keys=True)
