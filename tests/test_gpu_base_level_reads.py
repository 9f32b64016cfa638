"""BASELINE.json configs[0] (the plumbing configuration: ~4 k base-level reads, conf/Nanopore-Dec2019.conf) on the MI355X against
the reference's own code running live on the box's host cores -- tests/base_level_checks.py."""
import pytest

from tests import base_level_checks

pytestmark = pytest.mark.gpu


def test_four_thousand_base_level_reads_against_the_reference_running_beside_the_device(gpu_lib, ref_lib):
    reads, markers, candidates, stored = base_level_checks.plumbing(gpu_lib, ref_lib)
    assert reads >= 3600 and candidates > 50000 and stored > 30000
