import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def oracle_threads(n=4):
    """Context manager: BLAS threads for the oracle's NumPy work.  The GPU boxes have 256 host cores and OpenBLAS starts 64
    threads by default, which makes the oracle's many mid-sized products THREE times slower than four threads do (12 site
    updates of the headline evolve: 19.4 s against ~6.7 s, profiles/r06_oracle_threads.txt) - the reference's own advice
    (README.md:57-70: RENO_NUM_THREADS = 4)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except ImportError:             # pragma: no cover - the suite still runs, only slower
        import contextlib
        return contextlib.nullcontext()
