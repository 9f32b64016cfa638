"""Two ranks (gloo, both on cuda:0 -- the GPU box has one GPU) run the sharded job with the real
HIP stages: same driver as the RCCL path, only the transport differs.  Bit-exact against the
single-process oracle, for LowHash0 and for Align4 on the re-split candidate list."""
import os

import pytest

from tests import dist_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,kw", dist_checks.CASES)
def test_two_ranks_equal_single_process_oracle(gpu_lib, oracle_lib, seed, kw):
    emulated = gpu_lib.path if os.environ.get("SHASTA_EMU") == "1" else None
    dist_checks.two_ranks_equal_single_process_oracle(oracle_lib, seed, kw, library_path=emulated)


@pytest.mark.parametrize("seed,kw", dist_checks.CASES[:2])
def test_one_rank_over_rccl(gpu_lib, oracle_lib, seed, kw):
    """The RCCL transport itself -- backend "nccl", device tensors, the zero-copy views of the library's stage outputs handed to
    all_to_all_single / all_gather / all_reduce -- with the one rank a one-GPU box allows: every collective of the driver executes
    on the device (a rank exchanging with itself); what only more GPUs can show is the links."""
    if os.environ.get("SHASTA_EMU") == "1":
        pytest.skip("RCCL needs the GPU")
    dist_checks.two_ranks_equal_single_process_oracle(oracle_lib, seed, kw, port_base=29900, world=1, transport="nccl")
