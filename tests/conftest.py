import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch before anything of this suite loads the product library (test modules do at import, during collection), as in
    # bench.py: the torch wheel carries a HIP runtime of its own, and whichever libamdhip64 a process loads first is the one
    # both use -- with ROCm's loaded first, torch.cuda finds no device (test_gpu_beyond_4g_markers.py generates its reads on the
    # device with torch).  Nothing happens on a machine without a GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


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


EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "_build", "libshasta_mi355x_emu.so")


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel SOURCES of shasta_amd/csrc compiled by clang++ (host only) against the wave64 emulator of
    tests/emu (test infrastructure: kernels run on CPU fibers; nothing in the product loads it)."""
    import subprocess
    from shasta_amd import lib as libmod
    if os.environ.get("SHASTA_EMU_LIBRARY"):
        # An emulated build made with other compile-time options (e.g. make -C tests/emu OUT=... with
        # -DSHASTA_CELLS_GRID=1 appended to FLAGS): experiments are pre-flighted like everything else.
        return libmod.Library(os.environ["SHASTA_EMU_LIBRARY"])
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return libmod.Library(EMU_SO)


@pytest.fixture(scope="session")
def gpu_lib(request):
    """The product library on a real GPU.  Fails loudly (no skip, no fallback).
    Developer switch: SHASTA_EMU=1 pytest -m gpu runs the same tests on the emulated build
    (kernel sources on CPU fibers) -- a pre-flight for a machine without a GPU, never a result."""
    if os.environ.get("SHASTA_EMU") == "1":
        return request.getfixturevalue("emu_lib")
    import shasta_amd
    lib = shasta_amd.load()
    assert lib.device_count() >= 1, "no gfx950 device visible: the HIP path cannot run"
    return lib
