import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own kappa sources compiled by oracle/Makefile (oracle/_ref)."""
    from oracle.binding import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libemme_ref.so not built (needs /root/reference at build time)")
    return Reference()


@pytest.fixture(scope="session")
def emme():
    import emme_amd
    if not os.path.exists(emme_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    emme_amd.load()
    # the product builds its node cache only for calls with >= 8 omegas; the parity tests use a
    # handful and must still go through the cached kernels (the policy itself is tested separately)
    # (a Python-side default merged into every Context the tests create -- emme_options_t, not the environment)
    emme_amd.set_default_options(cache_min_batch=1)
    return emme_amd
