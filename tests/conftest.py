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
    from oracle import bindings
    if not bindings.oracle_available():
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return bindings.OracleLib()


@pytest.fixture(scope="session")
def ref_lib():
    from oracle import bindings
    if not bindings.ref_available():
        pytest.skip("oracle/_ref/libshasta_ref.so not built (needs /root/reference)")
    return bindings.RefLib()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU.  Fails loudly (no skip, no fallback)."""
    import shasta_amd
    lib = shasta_amd.load()
    assert lib.device_count() >= 1, "no gfx950 device visible: the HIP path cannot run"
    return lib
