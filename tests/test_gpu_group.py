"""The in-process multi-GPU seam (include/shasta_mi355x.h: shasta_mi355x_group, *_multi) on the MI355X.  The box has one
GPU: the device list names it two and three times, so the sharded LowHash0 (both exchanges as device-to-device copies,
host-side reductions) and the split aligners run for real; a node with more GPUs runs the same code with distinct ids."""
import pytest

pytestmark = pytest.mark.gpu


def test_several_devices_behind_one_call(gpu_lib, oracle_lib):
    from tests import group_checks
    assert group_checks.lowhash0_and_aligners(gpu_lib, oracle_lib) == 4


def test_group_errors_do_not_hang(gpu_lib):
    from tests import group_checks
    group_checks.errors_do_not_hang(gpu_lib)


def test_adversarial_lowhash0_through_the_group(gpu_lib, oracle_lib):
    from tests import group_checks
    assert group_checks.adversarial_lowhash0(gpu_lib, oracle_lib) > 15


def test_a_job_whose_iterations_do_not_fit_one_pass_falls_back_on_every_device(gpu_lib, oracle_lib):
    from tests import group_checks
    group_checks.one_pass_that_does_not_fit(gpu_lib, oracle_lib)


def test_the_staged_job_of_one_device_over_peer_copies_and_over_rccl(gpu_lib, oracle_lib):
    """The C++ seam's two transports with the one rank a one-GPU box has: device-to-device copies, and RCCL (librccl.so opened at
    run time, ncclCommInitAll, both exchanges as grouped ncclSend / ncclRecv)."""
    from tests import group_checks
    assert group_checks.staged_job_of_one_device(gpu_lib, oracle_lib, "peer") == 2
    assert group_checks.staged_job_of_one_device(gpu_lib, oracle_lib, "rccl") == 2
