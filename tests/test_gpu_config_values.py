"""Parity at the parameter values of BASELINE.json configs[0] / [3] / [4] (tests/config_value_checks.py): k = 14 kmer ids over
0 .. 2^28 and ids at the top of the 32-bit range, Nanopore-Dec2019's minAlignedFraction = 0.4, the ultra-long shape with
MinHash 10/50/5 -- LowHash0, Align4, align method 3 and their device-list forms on the MI355X against the oracle and the
reference's own code (oracle/_ref travels to the GPU box)."""
import pytest

from tests import config_value_checks

pytestmark = pytest.mark.gpu


def test_kmer_ids_of_k_14_through_both_stages_and_both_aligners(gpu_lib, oracle_lib, ref_lib):
    assert config_value_checks.wide_id_range(gpu_lib, oracle_lib, ref_lib, k=14) > 200


def test_kmer_ids_of_k_16(gpu_lib, oracle_lib):
    assert config_value_checks.wide_id_range(gpu_lib, oracle_lib, None, k=16, n_reads=160, genome_markers=9000, limit=400) > 100


def test_kmer_ids_at_the_top_of_the_32_bit_range(gpu_lib, oracle_lib):
    assert config_value_checks.top_of_the_id_range(gpu_lib, oracle_lib) >= 8


def test_nanopore_dec2019_values(gpu_lib, oracle_lib, ref_lib):
    kept, kept_without = config_value_checks.dec2019_values(gpu_lib, oracle_lib, ref_lib)
    assert kept < kept_without


def test_ultra_long_shape_with_minhash_10_50_5(gpu_lib, oracle_lib):
    assert config_value_checks.ultra_long_shape(gpu_lib, oracle_lib) >= 20
