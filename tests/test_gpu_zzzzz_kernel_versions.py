"""Kernels that exist in two versions.  The two forward kernels of the banded DP (K10b) on the MI355X: each against the oracle over a
sweep of band geometries, and the library's own choice -- it compares the two on the device when
the first DP runs and must settle on the second; a fallback to the first version is a failure here,
not a silent slowdown.  Each case runs in a process of its own (the version is fixed per process).
Sorted after the files that were green on the GPU before this kernel was written."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("version", [1, 2, 0])
def test_forward_dp_version_against_oracle(gpu_lib, version):
    env = dict(os.environ)
    env.pop("SHASTA_MI355X_DP_FORWARD", None)
    if version:
        env["SHASTA_MI355X_DP_FORWARD"] = str(version)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_versions_check.py"), gpu_lib.path, str(version or 2), "11"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_this_process_runs_the_second_version(gpu_lib):
    if os.environ.get("SHASTA_MI355X_DP_FORWARD"):
        pytest.skip("version forced by the environment")
    assert gpu_lib.dp_forward_version() == 2


@pytest.mark.parametrize("version", [1, 2])
def test_window_hash_kernel_versions_for_every_m(gpu_lib, version):
    """K1 (LowHash0's window-hash kernel) with and without shared block transforms, m = 1 .. 13, against the oracle."""
    env = dict(os.environ)
    env.pop("SHASTA_MI355X_HASH", None)
    if version == 1:
        env["SHASTA_MI355X_HASH"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hash_versions_check.py"), gpu_lib.path],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
