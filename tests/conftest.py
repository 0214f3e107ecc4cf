import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle, build
    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from pyoracle import RefElas
    if not RefElas.available():
        pytest.skip("oracle/_ref/libelas_ref.so not built (needs /root/reference)")
    return RefElas()
