import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


class _Tuning:
    """monkeypatch-like handle on the library's testing switches (dfx_debug_set_tuning): the
    library reads its environment once, so tests flip switches through the C ABI instead."""

    def __init__(self):
        import importlib
        self._capi = importlib.import_module("deep-fusion_amd.capi")
        self._set = []

    def setenv(self, key, value):
        self._capi.set_tuning(key, value)
        self._set.append(key)

    def undo(self):
        for k in self._set:
            self._capi.set_tuning(k, None)
        self._set = []


@pytest.fixture
def tuning():
    t = _Tuning()
    yield t
    t.undo()
