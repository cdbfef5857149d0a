import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def port():
    import oracle
    return oracle.port()


@pytest.fixture(scope="session")
def ref():
    import oracle
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref/libcsdr_ref.so not built (needs /root/reference)")
    return oracle.ref()
