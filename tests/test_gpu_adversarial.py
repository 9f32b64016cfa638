"""Inputs the synthetic read sets never produce (tests/adversarial.py), one named case per test, on the MI355X
against the oracle: degenerate read lengths, tandem repeats, tiny alphabets, identical / contained reads, long
reads; degenerate LowHash0 parameters (hashFraction >= 1, forced bucket counts, every m, palindromic flags)."""
import pytest

pytestmark = pytest.mark.gpu

from tests import adversarial


@pytest.mark.parametrize("name", adversarial.READ_SET_NAMES)
def test_adversarial_read_sets_through_both_aligners(gpu_lib, oracle_lib, ref_lib, name):
    adversarial.aligner_case(gpu_lib, oracle_lib, name, ref_lib=ref_lib)


@pytest.mark.parametrize("name", adversarial.LOWHASH_CASE_NAMES)
def test_adversarial_lowhash0_parameters(gpu_lib, oracle_lib, name):
    adversarial.lowhash_case(gpu_lib, oracle_lib, name)


@pytest.mark.parametrize("name", adversarial.LOWHASH_READ_SET_NAMES)
def test_adversarial_lowhash0_read_sets(gpu_lib, oracle_lib, name):
    adversarial.lowhash_read_set(gpu_lib, oracle_lib, name)


def test_lowhash0_rejects_a_bucket_count_below_the_minimum(gpu_lib):
    adversarial.lowhash_rejects_small_bucket_count(gpu_lib)


def test_task_list_overflow(gpu_lib, oracle_lib, monkeypatch):
    adversarial.task_list_overflow(gpu_lib, oracle_lib, monkeypatch)
