"""The aligner with a batch's first round prepared by the HOST loop (SHASTA_MI355X_DEVICE_BATCH_PREP=0) instead of the kernels of
align4_prepare.hpp (the default since round 3, which every other test runs): the two must make the same lists.  Against the
oracle on the adversarial read sets, mixed-length reads (all classes, the overflow ladder, the HBM-scratch list) and many small
batches."""
import os

import numpy as np
import pytest

from shasta_amd import abi, synthetic
from tests import adversarial, support

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", adversarial.READ_SET_NAMES)
def test_adversarial_read_sets_with_the_first_chunk_lists_made_on_the_host(gpu_lib, oracle_lib, ref_lib, monkeypatch, name):
    monkeypatch.setenv("SHASTA_MI355X_DEVICE_BATCH_PREP", "0")
    adversarial.aligner_case(gpu_lib, oracle_lib, name, long_reads=True, ref_lib=ref_lib)


@pytest.mark.parametrize("seed,mean,sigma", [(51, 3000.0, 0.8), (52, 6000.0, 0.6)])
def test_mixed_length_reads_with_the_first_chunk_lists_made_on_the_host(gpu_lib, oracle_lib, monkeypatch, seed, mean, sigma):
    monkeypatch.setenv("SHASTA_MI355X_DEVICE_BATCH_PREP", "0")
    toc, kmer = synthetic.marker_reads(160, 60000, mean_markers=mean, sigma=sigma, min_markers=300, seed=seed)
    data7 = synthetic.pack_markers(toc, kmer)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=40, minFrequency=1)
    cand = oracle_lib.lowhash0(toc, data7, None, p).candidates[:1200]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    a = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    b = gpu_lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    support.same_align(a, b)


def test_many_small_batches_with_the_first_chunk_lists_made_on_the_host(gpu_lib):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (and the borrowed results through the workers' staging buffers -- what happens where the result array cannot be page-locked)
    env = dict(os.environ, SHASTA_MI355X_ALIGN_BATCH_LOG2="10", SHASTA_MI355X_RESULTS_NOT_PAGE_LOCKED="1", SHASTA_MI355X_SLICE_COPY_MIN_BYTES="1")
    out = subprocess.run([sys.executable, "-m", "tests.borrowed_checks", gpu_lib.path, "oracle", "both-preparations"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "equal owned results" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
