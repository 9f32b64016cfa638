"""The DP tie policy switch on the MI355X: the kernels compiled for the alternative policy against the oracle under the same
policy (tests/tie_policy_checks.py)."""
import pytest

from tests import tie_policy_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alternative", tie_policy_checks.ALTERNATIVES)
def test_dp_tasks_under_the_alternative_tie_policy(gpu_lib, oracle_lib, alternative):
    tasks, bad_default, bad_alternative, differ = tie_policy_checks.dp_tasks_under_the_alternative_policy(gpu_lib, oracle_lib, alternative=alternative)
    assert tasks >= 50 and bad_default == 0 and bad_alternative == 0
    assert differ >= 10              # the inputs do tell the two policies apart


@pytest.mark.parametrize("alternative", tie_policy_checks.ALTERNATIVES)
def test_aligner_under_the_alternative_tie_policy(gpu_lib, oracle_lib, alternative):
    candidates, differ = tie_policy_checks.aligner_under_the_alternative_policy(gpu_lib, oracle_lib, alternative=alternative)
    assert candidates >= 200


def test_a_tie_policy_that_is_not_compiled_is_refused(gpu_lib):
    assert tie_policy_checks.unknown_policy_is_refused(gpu_lib)
