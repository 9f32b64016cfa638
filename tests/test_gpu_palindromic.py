"""Palindromic-read flagging on the MI355X: the device screen against its definition,
the stage (device screen + host method 0) through the Python mirror and through the stage executable
against what the reference answered (tests/golden/palindromic.npz).  Written after the round's GPU
access closed: sorted after the files that were green on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from tests import host_support, palindromic_checks as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def host_library_of(gpu_lib):
    if os.environ.get("SHASTA_EMU") == "1":
        return os.path.join(os.path.dirname(gpu_lib.path), "libshasta_mi355x_host_emu.so")
    return os.path.join(ROOT, "shasta_amd", "_build", "libshasta_mi355x_host.so")


def test_screen_is_the_bound_and_is_sound(gpu_lib):
    z, inputs = pc.golden()
    for name, (toc, data7) in inputs.items():
        kmer = np.ascontiguousarray(data7.reshape(-1, 7)[:, :4]).view("<u4").reshape(-1)
        with gpu_lib.context(0) as ctx:
            ctx.set_markers(toc, data7)
            for i, kw in enumerate(pc.PARAMETER_SETS):
                bound = ctx.palindromic_screen(kw["delta_threshold"])
                assert np.array_equal(bound, pc.numpy_bound(toc, kmer, kw["delta_threshold"]))
                screened = pc.screen_is_sound(bound, toc, z["%s_%d_near" % (name, i)], kw["near_diagonal_fraction_threshold"])
                assert not np.any(screened & (z["%s_%d_flags" % (name, i)] != 0))
            for delta in (1, 2, 63, 64, 65, 1000, 4096):
                assert np.array_equal(ctx.palindromic_screen(delta), pc.numpy_bound(toc, kmer, delta)), delta


def test_stage_equals_reference_fixture(gpu_lib, tmp_path):
    z, inputs = pc.golden()
    for name, (toc, data7) in inputs.items():
        for i, kw in enumerate(pc.PARAMETER_SETS):
            read_count = (len(toc) - 1) // 2
            before = (np.arange(read_count) % 4).astype(np.uint8)
            flags, counts = pc.flag_through_stage(toc, data7, tmp_path / ("%s%d" % (name, i)), host_library_of(gpu_lib),
                                                  initial_flags=before, **kw)
            assert np.array_equal(flags & 1, z["%s_%d_flags" % (name, i)])
            assert np.array_equal(flags & 0xfe, before & 0xfe)
            assert counts[0] == read_count and counts[2] == int(z["%s_%d_flags" % (name, i)].sum())


def test_stage_equals_oracle_on_other_reads(gpu_lib, oracle_lib, tmp_path):
    toc, kmer, data7, kinds = pc.read_set(n_reads=96, seed=77)
    expected = oracle_lib.flag_palindromic_reads(toc, data7, **pc.DEFAULTS)[0]
    flags, counts = pc.flag_through_stage(toc, data7, tmp_path, host_library_of(gpu_lib), **pc.DEFAULTS)
    assert np.array_equal(flags & 1, expected) and counts[2] == int(expected.sum()) > 20


def test_stage_executable_prints_the_reference_lines(gpu_lib, tmp_path):
    if os.environ.get("SHASTA_EMU") == "1":
        pytest.skip("the stage executable is linked against the product library")
    z, inputs = pc.golden()
    toc, data7 = inputs["hairpins"]
    d = str(tmp_path / "Data")
    os.makedirs(d)
    host_support.HostShim().write_data_dir(d, toc, data7, np.zeros((len(toc) - 1) // 2, np.uint8))
    out = subprocess.run([host_support.STAGE, "palindromic", d], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    flagged, total = int(z["hairpins_0_flags"].sum()), len(z["hairpins_0_flags"])
    # src/AssemblerAlign.cpp:694-697
    assert "Flagged %d reads as palindromic out of %d total." % (flagged, total) in out.stdout
    assert "Palindromic fraction is " in out.stdout
    stored, _ = host_support.HostShim().open_vector(os.path.join(d, "ReadFlags"), 1)
    assert np.array_equal(stored.reshape(-1) & 1, z["hairpins_0_flags"])


def test_whole_chain_on_the_tiny_reads(gpu_lib, oracle_lib, tmp_path, monkeypatch):
    # reads -> markers -> palindromic flags -> LowHash0 -> candidate table -> Align4 -> read graph on one Data/ directory
    from tests import mirror_checks
    mirror_checks.whole_chain_on_the_tiny_reads(oracle_lib, tmp_path, monkeypatch, host_library_of(gpu_lib))


def test_randomized_campaign(gpu_lib, oracle_lib):
    # seeded random read sets, candidate lists and parameter draws through both stages, against the oracle
    from tests import campaign
    emulated = os.environ.get("SHASTA_EMU") == "1"
    assert campaign.align4(gpu_lib, oracle_lib, range(1000, 1010 if emulated else 1060)) > 1000
    assert campaign.lowhash0(gpu_lib, oracle_lib, range(1100, 1125 if emulated else 1250)) >= 15


def test_stage_scripts_in_a_run_directory(gpu_lib, oracle_lib, tmp_path):
    # scripts/FindMarkers.py ... CreateReadGraph.py as processes in a run directory
    from tests import mirror_checks
    mirror_checks.stage_scripts_in_a_run_directory(oracle_lib, tmp_path, host_library_of(gpu_lib))
