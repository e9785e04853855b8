import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_b
    oracle_b.build()
    return oracle_b


@pytest.fixture(autouse=True)
def _library_options_as_found():
    """The per-call options (MgsOptions: bin_mode, tight_bins, ...) are process-wide: whatever a test changes is put back, so
    that the tests behind it run under the library's defaults (a sweep that left bin_mode = 1 behind once hid the default
    binning from every oracle case after it)."""
    try:
        from manigaussian_amd import _lib
        before = dict(_lib.DEFAULT_OPTIONS)
    except Exception:  # (the library is not built: the tests that need it say so themselves)
        yield
        return
    yield
    for k, v in before.items():
        if _lib.DEFAULT_OPTIONS.get(k) != v:
            _lib.set_option(k, v)
