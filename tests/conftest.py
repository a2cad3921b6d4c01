import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def reference():
    import oracle_lib
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return oracle_lib.Reference()


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI. No fallback: a missing library or device is an error, not a skip."""
    import kmercamel_amd
    c = kmercamel_amd.Context(0)
    yield c
    c.close()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
