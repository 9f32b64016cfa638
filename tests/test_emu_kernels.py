"""Pre-flight of the HIP kernel SOURCES without a GPU: shasta_amd/csrc/*.hip compiled unmodified by
clang++ (host only) against the wave64 emulator of tests/emu (work-items are fibers that meet at every cross-lane
operation and barrier) and run through the same C ABI and the same checks as the -m gpu tests, at
sizes a CPU finishes in seconds.  This is test infrastructure: it proves the kernel logic against
the oracle, not the MI355X run (no LDS limits, no timing, no inter-workgroup memory model), and
nothing in the product can load the emulated library.  The whole -m gpu suite runs on it with
SHASTA_EMU=1 (see tests/conftest.py)."""
import os

import numpy as np
import pytest

from shasta_amd import abi
from tests import align3_checks, support

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_library_is_the_same_abi(emu_lib):
    assert emu_lib.device_count() == 1 and "gfx950" in emu_lib.version()


@pytest.mark.parametrize("m", [3, 4, 5])
def test_hash_windows(emu_lib, oracle_lib, m):
    rng = np.random.default_rng(m)
    k = rng.integers(0, 1 << 20, size=3000, dtype=np.uint32)
    for iteration in (0, 7):
        assert np.array_equal(emu_lib.hash_windows(k, m, iteration), oracle_lib.hash_windows(k, m, iteration))


@pytest.mark.parametrize("width", [20, 65, 300, 1000])
def test_banded_dp(emu_lib, oracle_lib, width):
    from tests.test_gpu_align4 import noisy_copy
    rng = np.random.default_rng(width)
    for trial in range(3):
        alphabet = (1 << 20) if trial % 2 == 0 else 12          # small alphabet: many score ties
        n = int(rng.integers(150, 500))
        genome = rng.integers(0, alphabet, size=n + 400, dtype=np.uint32)
        a = noisy_copy(rng, genome[:n], alphabet=alphabet)
        off = int(rng.integers(0, 300))
        b = noisy_copy(rng, genome[off:off + n], alphabet=alphabet)
        lo = off + int(rng.integers(-30, 30)) - width // 2
        x, sx = oracle_lib.banded_dp(a, b, lo, lo + width - 1)
        y, sy = emu_lib.banded_dp(a, b, lo, lo + width - 1)
        assert sx == sy and np.array_equal(x, y)


def test_dp_tie_policy_switch(emu_lib, oracle_lib):
    from tests import tie_policy_checks
    for alternative in tie_policy_checks.ALTERNATIVES:
        tasks, bad_default, bad_alternative, differ = tie_policy_checks.dp_tasks_under_the_alternative_policy(emu_lib, oracle_lib, alternative=alternative)
        assert tasks >= 50 and bad_default == 0 and bad_alternative == 0 and differ >= 10
    candidates, differ = tie_policy_checks.aligner_under_the_alternative_policy(emu_lib, oracle_lib, reads=80, candidates=120, alternative=2)
    assert candidates >= 100
    assert tie_policy_checks.unknown_policy_is_refused(emu_lib)


def test_bands_of_more_than_1024_diagonals(emu_lib, oracle_lib):
    from tests import wide_band_checks
    tasks, bad = wide_band_checks.dp_tasks(emu_lib, oracle_lib, widths=(1100, 40, 2500, 64), n_range=(300, 700))
    assert tasks == 4 and bad == 0
    assert wide_band_checks.aligner(emu_lib, oracle_lib, max_band=2000, length=4400, every=3) >= 1


def test_lowhash0_and_align4(emu_lib, oracle_lib):
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=12000, seed=5)
    flags = np.zeros(150, np.uint8)
    flags[[5, 17]] = 1
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    a, b = emu_lib.lowhash0(toc, data7, flags, p), oracle_lib.lowhash0(toc, data7, flags, p)
    support.same_lowhash(a, b)
    cand = b.candidates[:400]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    x = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    y = emu_lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    assert y.dp_cell_count == x.dp_cell_count and y.kmer_id_bytes == x.kmer_id_bytes     # dpSizeKernel's sums
    # The kernel table books the dense kernels' cells: all of the reference's DP cells with the sparse path switched off, a fraction
    # of them with it (the tasks whose alignment is the unique optimal chain of their matches never reach the dense kernels).
    for sparse, monkey in (("1", None), ("0", None)):
        os.environ["SHASTA_MI355X_SPARSE_DP"] = sparse
        try:
            with emu_lib.context(0) as ctx:
                ctx.set_kmer_ids(toc, kmer)
                z = ctx.align4(cand, o, want_ordinals=True)
                t = ctx.kernel_table()
        finally:
            del os.environ["SHASTA_MI355X_SPARSE_DP"]
        forward = [v for k, v in t.items() if k.startswith("bandedDpForwardKernel")]
        assert z.dp_cell_count == x.dp_cell_count
        if sparse == "0":
            assert sum(v["work"] for v in forward) == z.dp_cell_count and "sparseChainKernel" not in t and "sparseChainWaveKernel" not in t and sum(v["bytes"] for v in forward) > 0
        else:
            # (possibly none at all: what the chain kernel does not certify, the anchor kernel mostly does)
            assert sum(v["work"] for v in forward) < z.dp_cell_count // 2 and t["sparseChainWaveKernel"]["launches"] >= 1 and t["sparseAnchorKernel"]["launches"] >= 1
        if not (x.status & 0x80).any():
            support.same_align(x, z)
    if not (x.status & 0x80).any():
        support.same_align(x, y)
    else:
        keep = (x.status & 0x80) == 0
        assert np.array_equal(x.status[keep], y.status[keep])


def test_lowhash0_iteration_after_iteration(emu_lib, oracle_lib, monkeypatch):
    # SHASTA_MI355X_LOWHASH_ONE_PASS=0 against the default (all iterations in one pass) and the oracle.
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=12000, seed=41)
    for kw in (dict(m=3, minHashIterationCount=16, minBucketSize=2, maxBucketSize=30, minFrequency=2), dict(m=7, minHashIterationCount=5, hashFraction=0.02, minBucketSize=2, maxBucketSize=30, minFrequency=2)):
        p = abi.default_lowhash0_params(**kw)
        a = emu_lib.lowhash0(toc, data7, None, p)
        monkeypatch.setenv("SHASTA_MI355X_LOWHASH_ONE_PASS", "0")
        b = emu_lib.lowhash0(toc, data7, None, p)
        monkeypatch.delenv("SHASTA_MI355X_LOWHASH_ONE_PASS")
        support.same_lowhash(a, b)
        support.same_lowhash(a, oracle_lib.lowhash0(toc, data7, None, p))


def test_lowhash0_golden_fixture(emu_lib):
    g = support.Golden("tiny.npz")
    out = emu_lib.lowhash0(g.toc, g.data7, None, abi.default_lowhash0_params())
    support.check_lowhash(out, g.z, 0)


@pytest.mark.parametrize("i", [0])        # (option sets 1 and 2 of the fixture: on the GPU only, 80 s of emulation)
def test_align3_reference_fixture(emu_lib, i):
    align3_checks.golden_fixture(emu_lib, "tiny", i)


@pytest.mark.parametrize("seed,kw", [
    (21, dict()),
    (23, dict(downsamplingFactor=0.25, bandExtend=2, maxBand=30, minAlignedMarkerCount=20, suppressContainments=1)),
    (24, dict(downsamplingFactor=0.002, minAlignedMarkerCount=40)),
    (25, dict(matchScore=3, mismatchScore=-2, gapScore=-3, minAlignedMarkerCount=40)),        # (scores of the caller's choice)
])
def test_align3_against_oracle(emu_lib, oracle_lib, seed, kw):
    align3_checks.against_oracle(emu_lib, oracle_lib, seed, kw)


def test_align3_context_paths(emu_lib, oracle_lib):
    align3_checks.context_paths(emu_lib, oracle_lib)


def test_borrowed_results_and_calls_of_several_batches(emu_lib, oracle_lib):
    # (a process of its own: the batch size is read once per process)
    import subprocess, sys
    env = dict(os.environ, SHASTA_MI355X_ALIGN_BATCH_LOG2="10", SHASTA_MI355X_SLICE_COPY_MIN_BYTES="1")     # (the tail's copies in slices at this size too)
    out = subprocess.run([sys.executable, "-m", "tests.borrowed_checks", emu_lib.path, "oracle", "both-preparations"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "equal owned results" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_candidate_and_alignment_tables_and_read_graph_selection(emu_lib, oracle_lib):
    from tests import table_checks
    table_checks.check(emu_lib)
    assert table_checks.table_of_the_last_aligner_call(emu_lib, oracle_lib, n_reads=90, limit=250) > 100


def test_align3_long_reads(emu_lib, oracle_lib):
    # (smaller reads than the GPU test's, and only the case beyond 8192 diagonals: the emulator's time; pairs with 1025 .. 8192
    # diagonals occur in test_align3_against_oracle's small factors)
    align3_checks.long_reads(emu_lib, oracle_lib, mean_markers=7100.0, factors=(0.95, 0.25), cases=(0,))


def test_align3_rejected_options(emu_lib):
    align3_checks.rejected_options(emu_lib)


@pytest.mark.parametrize("align_method", [3, 4])
def test_stages_on_a_data_directory(emu_lib, oracle_lib, tmp_path, monkeypatch, align_method):
    # The C++ host layer (Data/ files in, Data/ files out) linked against the emulated library.
    import os
    from tests import mirror_checks
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    mirror_checks.stages_on_a_data_directory(oracle_lib, tmp_path, monkeypatch, host, align_method)


def test_marker_finding(emu_lib, oracle_lib):
    from tests import marker_checks
    marker_checks.golden_fixture(emu_lib.find_markers)
    for seed, k in ((1, 10), (2, 7), (5, 12)):
        marker_checks.against_oracle(emu_lib, oracle_lib, seed, k)
    marker_checks.resident_markers_feed_lowhash0(emu_lib)


def test_find_markers_on_a_data_directory(emu_lib, tmp_path):
    import os
    from tests import mirror_checks
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    mirror_checks.find_markers_on_a_data_directory(tmp_path, host)


from tests import adversarial


@pytest.mark.parametrize("name", adversarial.READ_SET_NAMES[:-1])      # the long reads are too slow on CPU fibers
def test_adversarial_read_sets_through_both_aligners(emu_lib, oracle_lib, ref_lib, name):
    adversarial.aligner_case(emu_lib, oracle_lib, name, long_reads=False, ref_lib=ref_lib)


@pytest.mark.parametrize("name", ["degenerate lengths", "tandem repeats", "duplicated segments"])      # (retries after host-made lists in the first two)
def test_adversarial_read_sets_with_the_first_chunk_lists_made_on_the_host(emu_lib, oracle_lib, ref_lib, monkeypatch, name):
    # SHASTA_MI355X_DEVICE_BATCH_PREP=0 (read for every batch): classes, grouping sort and chunk lists by the host loop instead of
    # the kernels of align4_prepare.hpp (the default, which every other test runs)
    monkeypatch.setenv("SHASTA_MI355X_DEVICE_BATCH_PREP", "0")
    adversarial.aligner_case(emu_lib, oracle_lib, name, long_reads=False, ref_lib=ref_lib)


def test_mixed_length_reads_with_the_first_chunk_lists_made_on_the_host(emu_lib, oracle_lib, monkeypatch):
    # Every table class, swapped chunks, the overflow ladder after host-made lists, the HBM-scratch list (the test of
    # tests/test_gpu_align4.py, its second read set cut to 450 candidates for the emulator's time).
    from shasta_amd import synthetic
    monkeypatch.setenv("SHASTA_MI355X_DEVICE_BATCH_PREP", "0")
    toc, kmer = synthetic.marker_reads(160, 60000, mean_markers=6000.0, sigma=0.6, min_markers=300, seed=52)
    data7 = synthetic.pack_markers(toc, kmer)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=40, minFrequency=1)
    cand = oracle_lib.lowhash0(toc, data7, None, p).candidates[:450]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    a = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    b = emu_lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    support.same_align(a, b)
    assert (a.status == abi.SHASTA_ALIGN_STORED).sum() > 10


def test_adversarial_lowhash0(emu_lib, oracle_lib):
    adversarial.lowhash0(emu_lib, oracle_lib)


def test_task_list_overflow(emu_lib, oracle_lib, monkeypatch):
    adversarial.task_list_overflow(emu_lib, oracle_lib, monkeypatch)


def test_smoke_body(emu_lib):
    import __graft_entry__
    __graft_entry__.smoke_on(emu_lib)


def test_two_ranks_equal_single_process_oracle(emu_lib, oracle_lib):
    # The sharded job (staged lh_* entry points, both exchanges, candidate re-split) on two ranks.
    from tests import dist_checks
    seed, kw = dist_checks.CASES[0]
    dist_checks.two_ranks_equal_single_process_oracle(oracle_lib, seed, kw, library_path=emu_lib.path, port_base=29800)



def test_banded_dp_geometries_against_oracle(emu_lib, oracle_lib):
    from tests import dp_geometry_checks
    dp_geometry_checks.check(emu_lib, oracle_lib, seed=5, tasks=48, trials=3)


def test_banded_dp_wavefront_with_tasks_of_both_runs_of_its_class(emu_lib, oracle_lib):
    # (written after the round's last GPU call: on the emulated build, whose device checks watch the trace bounds; joins the
    # -m gpu suite with the next round's first call)
    from tests import dp_geometry_checks
    for seed in (3, 4):
        cases, bad = dp_geometry_checks.straddling_bundles(emu_lib, oracle_lib, seed)
        assert cases == 13 and bad == 0


def test_window_hash_kernel_for_every_m(emu_lib, oracle_lib):
    from tests import hash_every_m_checks
    assert hash_every_m_checks.sweep(emu_lib, oracle_lib, reads=60) > 8 * 60


def test_randomized_campaign(emu_lib, oracle_lib):
    from tests import campaign
    assert campaign.align4(emu_lib, oracle_lib, range(900, 904)) > 300
    assert campaign.lowhash0(emu_lib, oracle_lib, range(950, 960)) >= 4


def test_stage_scripts_in_a_run_directory(emu_lib, oracle_lib, tmp_path):
    from tests import mirror_checks
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    mirror_checks.stage_scripts_in_a_run_directory(oracle_lib, tmp_path, host)


def test_several_devices_behind_one_call(emu_lib, oracle_lib):
    from tests import group_checks
    assert group_checks.lowhash0_and_aligners(emu_lib, oracle_lib, device_lists=((0, 0), (0, 0, 0)), n_reads=160, limit=300) == 4
    group_checks.errors_do_not_hang(emu_lib)
    group_checks.one_pass_that_does_not_fit(emu_lib, oracle_lib)
    assert group_checks.staged_job_of_one_device(emu_lib, oracle_lib, "peer", n_reads=160) == 2        # (a world of one through the staged job)


def test_lowhash0_calls_of_one_context_share_their_allocations(emu_lib, oracle_lib):
    # A context keeps the device allocations of its last LowHash0 job for the next one: different parameters in turn
    # (more and fewer records, more and fewer iterations, a dynamic iteration count) must each equal the oracle.
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=8000, seed=61)
    cases = [abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30),
             abi.default_lowhash0_params(m=3, hashFraction=0.05, minHashIterationCount=4, minBucketSize=2, maxBucketSize=60, minFrequency=1),
             abi.default_lowhash0_params(minHashIterationCount=0, alignmentCandidatesPerRead=6.0, minBucketSize=2, maxBucketSize=30),
             abi.default_lowhash0_params(m=5, hashFraction=0.002, minHashIterationCount=25, minBucketSize=2, maxBucketSize=30, minFrequency=1),
             abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)]
    with emu_lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        for p in cases:
            support.same_lowhash(ctx.lowhash0(p), oracle_lib.lowhash0(toc, data7, None, p))


def test_kmer_ids_of_k_14_and_the_top_of_the_32_bit_range(emu_lib, oracle_lib):
    # BASELINE configs[3] / [4]: Kmers.k = 14 (ids up to 2^28); and raw ids just below 2^32 (tests/config_value_checks.py).
    from tests import config_value_checks
    assert config_value_checks.wide_id_range(emu_lib, oracle_lib, None, k=14, n_reads=110, genome_markers=6000, limit=150) > 100
    assert config_value_checks.top_of_the_id_range(emu_lib, oracle_lib) >= 8


def test_base_level_reads_against_the_reference_running_live(emu_lib, ref_lib):
    # BASELINE configs[0] at a tenth of its size (tests/base_level_checks.py; the -m gpu test runs 4 000 reads).
    from tests import base_level_checks
    reads, markers, candidates, stored = base_level_checks.plumbing(emu_lib, ref_lib, n_reads=400, genome_length=150000, limit=400)
    assert reads >= 360 and stored > 100


def test_sparse_form_of_the_banded_alignment(emu_lib, oracle_lib):
    # align4_sparse.hpp: on and off, under every compiled tie policy, and through the aligner (tests/sparse_checks.py).
    from tests import sparse_checks
    tasks, clean_share, tie_heavy_share = sparse_checks.dp_tasks(emu_lib, oracle_lib, clean=45, tie_heavy=30, alternatives=(2,), long_every=44)
    assert tasks >= 25 and clean_share > 0.5 and tie_heavy_share < 0.3          # (two of the clean tasks straddle the 8192-marker limit and hold half of the cells)
    assert sparse_checks.aligner(emu_lib, oracle_lib, n_reads=90, limit=160) > 0.6


def test_locally_ambiguous_tasks_through_the_anchor_kernel(emu_lib, oracle_lib):
    # align4_anchor.hpp (tests/sparse_checks.py): rectangles between anchors, under every compiled tie policy, on and off.
    from tests import sparse_checks
    runs, cells_all, cells_sparse, cells_anchored = sparse_checks.anchored_tasks(emu_lib, oracle_lib, seeds=(3, 4), tasks=24)
    assert runs >= 120 and cells_sparse > 0.8 * cells_all and cells_anchored < 0.4 * cells_all
    assert sparse_checks.tiny_tasks(emu_lib, oracle_lib, tasks=200, alternatives=(3,)) >= 400


def test_wave_kernel_forms_and_the_anchor_kernel_second_launch(emu_lib, oracle_lib):
    """align4_chainwave.hpp against the oracle and the forms it can be switched to (lane-per-task kernel, its own ordering of the hits,
    the side stream), tasks of every capacity class; align4_anchor.hpp's second launch on rectangles beyond the first one's LDS."""
    from tests import sparse_checks
    assert sparse_checks.wave_kernel_forms(emu_lib, oracle_lib) >= 200
    with_second, without = sparse_checks.anchor_kernel_second_launch(emu_lib, oracle_lib)
    assert 0 < with_second < without


def test_read_statistics_over_several_partitions_and_spans(emu_lib, oracle_lib):
    # readStatisticsKernel (lowhash0.hip): 4 200 table entries = 3 partitions, > 65 536 records = several spans; and the atomics form.
    from tests import statistics_checks
    assert statistics_checks.several_partitions_and_spans(emu_lib, oracle_lib, cases=((0.05, 6),)) >= 1400


def test_pairs_of_two_long_reads(emu_lib, oracle_lib, monkeypatch):
    """The windowed class of the cells stage (align4CellsLongKernel) and the sort kernel's classes for reads beyond 8 192 markers."""
    from tests import long_read_checks
    monkeypatch.setenv("SHASTA_MI355X_MATCH_SHIFT", "20")
    monkeypatch.setenv("SHASTA_MI355X_ALIGN_WORKERS", "1")
    r = long_read_checks.both_long(emu_lib, oracle_lib, lengths=(9000, 12500, 9500, 8300, 4000), genome_markers=16000)
    assert r["both_long"] == 12 and r["in_the_windowed_class"] == 12 and r["in_the_hbm_scratch_kernel"] == 0 and not r["dense_because"]


def test_a_full_cell_table_ends_the_counting_and_the_climb_ends_in_the_reference_answer(emu_lib, oracle_lib, monkeypatch):
    """Repeat-rich pairs of two long reads fill the windowed class's cell table and the first tables of the HBM-scratch kernel
    (tests/long_read_checks.py: before the early exit this took five minutes here, four seconds since)."""
    from tests import long_read_checks
    monkeypatch.setenv("SHASTA_MI355X_ALIGN_WORKERS", "1")
    r = long_read_checks.full_tables(emu_lib, oracle_lib, lengths=(9000, 12000, 8800), alphabet_size=150)
    assert r["candidates"] == 6 and r["windowed_launches"] >= 1 and r["hbm_scratch_launches"] >= 2 and r["hbm_scratch_candidates"] > r["candidates"]


@pytest.mark.parametrize("force", ["long", "big"])
def test_every_candidate_through_the_windowed_kernels(emu_lib, oracle_lib, force, monkeypatch):
    from tests import long_read_checks
    monkeypatch.setenv("SHASTA_MI355X_ALIGN_WORKERS", "1")
    assert long_read_checks.forced(emu_lib, oracle_lib, None, force, n_reads=120, limit=400, adversarial_sets=False) >= 700


def test_a_call_without_ordinals(emu_lib, oracle_lib):
    from tests import sparse_checks
    assert sparse_checks.without_ordinals(emu_lib, oracle_lib, n_reads=120, limit=400) >= 100


def test_long_dense_paths(emu_lib, oracle_lib):
    from tests import sparse_checks
    assert sparse_checks.long_dense_paths(emu_lib, oracle_lib) == 20
