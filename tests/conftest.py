import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _blas_threads():
    """The oracle's GEMMs are small (16..64 rows): OpenBLAS's default 8 busy-waiting threads are slower on them than 4 (3.0 s vs 2.2 s
    per 20,000 products here) and, on a host shared with other jobs, have been seen to stretch a 75 s oracle test to tens of minutes."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits={"openblas": 4}):      # numpy's BLAS only; torch keeps its own pool
        yield
