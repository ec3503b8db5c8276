import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """The oracle library must exist before any test touches it (the GPU box
    receives the prebuilt .so files with the snapshot; building is a no-op)."""
    import _oracle
    if not os.path.exists(os.path.join(_oracle.ORACLE_DIR, "liboracle_fsk.so")):
        _oracle.build_oracle()
    yield
