"""bench.py's reporting path with the GPU library mocked out: the one JSON line carries every key of
the driver's contract (plus roofline / cpu_baseline) and the per-kernel bookkeeping is consistent.
The numbers are fake; only the plumbing is under test."""
import json
import sys
import types

import numpy as np
import pytest

import bench
from shasta_amd import abi


FAKE_TABLE = {
    "hashWindowsKernel<4, true, unsigned int>": {"seconds": 0.02, "launches": 20, "bytes": 20 * 1_250_000_000, "work": 20 * 300_000_000},
    "radix sort of low-hash records": {"seconds": 0.004, "launches": 20, "bytes": 20 * 300_000_000, "work": 20 * 3_000_000},
    "align4CellsChunkKernel<2, false>": {"seconds": 0.22, "launches": 64, "bytes": 64 * 500_000_000, "work": 3_000_000},
    "bandedDpForwardKernel<16, 2>": {"seconds": 0.08, "launches": 32, "bytes": 32 * 296_000_000, "work": int(3e10)},
    "bandedDpForwardKernel<16, 4, 0, false>": {"seconds": 0.28, "launches": 32, "bytes": 32 * 1_186_000_000, "work": int(2.2e11)},
    "bandedDpForwardKernel<32, 4>": {"seconds": 0.20, "launches": 32, "bytes": 32 * 353_000_000, "work": int(1.1e11)},
    "dpTracebackKernel": {"seconds": 0.16, "launches": 32, "bytes": 32 * 1_800_000_000, "work": 32 * 150_000},
}


def strict_loads(text):
    """json.loads that refuses NaN / Infinity (the driver's parser may)."""
    def refuse(name):
        raise ValueError("non-finite constant in the line: " + name)
    return json.loads(text, parse_constant=refuse)


class FakeResult:
    def __init__(self, n):
        self.candidates = abi.make_pairs(np.arange(n), np.arange(n) + 1, np.ones(n))
        self.alignment_data = np.zeros(n - 1, dtype=abi.ALIGNMENT_DATA_DTYPE)
        self.status = np.zeros(n, np.uint8)
        self.status[-1] = abi.SHASTA_ALIGN_REJECTED
        self.device_seconds, self.seconds = 0.4, 0.5
        self.dp_cell_count = int(4e11)


class FakeContext:
    def set_kmer_ids(self, toc, kmer):
        pass

    def lowhash0(self, p):
        return FakeResult(1000)

    def align4(self, candidates, o, want_ordinals=False, borrow=False):
        return FakeResult(len(candidates))

    def alignment_table(self, copy=True):
        return np.zeros(21, np.uint64), np.zeros(0, np.uint32)

    def kernel_table(self):
        return FAKE_TABLE

    def kernel_table_reset(self):
        pass

    def close(self):
        pass


class FakeLibrary:
    def device_count(self):
        return 1

    def context(self, device):
        return FakeContext()


def test_bench_prints_one_contract_line(monkeypatch, capsys, tmp_path):
    import torch
    import shasta_amd
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(shasta_amd, "load", lambda: FakeLibrary())
    monkeypatch.setattr(bench, "make_workload", lambda reads, seed: (np.zeros(2 * 10 + 1, np.uint64), np.zeros(0, np.uint32)))
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: ({"value": 18000.0, "unit": "candidate read-pairs aligned/s", "cores": 64,
                                                           "host_cores": 256, "kind": "reference", "sample": "fake"},
                                                          {"lowhash0_equal": True, "aligner_mismatches": 0}))
    # Counters as scripts/pmc_summary.py writes them, for the workload of this run.
    pmc = tmp_path / "pmc.json"
    pmc.write_text(json.dumps({"workload_reads": 100000, "kernel_source_hash": shasta_amd.kernel_source_hash(), "kernels": {
        "bandedDpForwardKernel<16, 4, 0, false>": {"hbm_bytes_per_launch": 2.6e9, "valu_wave_instructions_per_launch": 2.3e9},
        "hashWindowsKernel<4, true, unsigned int>": {"hbm_bytes_per_launch": 1.5e9, "valu_wave_instructions_per_launch": 3.6e8}}}))
    monkeypatch.setattr(bench, "PMC_FILE", str(pmc))
    monkeypatch.setattr(bench, "DETAILS_FILE", str(tmp_path / "details.json"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--reads", "100000"])
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    head = strict_loads(lines[0])
    # The driver's line: short (round 4's 20 KB line was cut and could not be parsed), strict JSON, the contract keys with the
    # roofline of one kernel and the CPU baseline as numbers; the tables are in the details file.
    assert len(lines[0]) < 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_at_bench_size", "details"):
        assert key in head, key
    assert "kernels" not in head and "kernels_one_worker" not in head and "hbm_budget_per_gpu" not in head
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"]) <= set(head["roofline"])
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(head["cpu_baseline"]) and len(head["cpu_baseline"]["sample"]) <= 160
    d = strict_loads(open(str(tmp_path / "details.json")).read())
    assert d["value"] == pytest.approx(head["value"], rel=1e-5) and d["roofline"]["kernel"] == head["roofline"]["kernel"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_at_bench_size", "aligner_status"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "computeAlignmentTable" in d["config"]["step"] and d["stage_seconds_per_step"]["alignment_table_call"] >= 0.0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # The dominant kernel of the fake table is the <= 64-diagonal DP class: integer VALU work, with the measured
    # instruction count of the PMC file over the live launch time as its VALU fraction.
    assert r["kernel"] == "bandedDpForwardKernel<16, 4, 0, false>" and r["bound"] == "valu" and r["traffic"] == 2.6e9
    assert r["valu"]["frac"] == pytest.approx(2.3e9 / (0.28 / 32) / bench.VALU_PEAK_WAVE_INSTRUCTIONS_PER_S)
    k = d["kernels"]
    assert k["hashWindowsKernel<4, true, unsigned int>"]["launches_per_step"] == 10 and k["hashWindowsKernel<4, true, unsigned int>"]["avg_ms"] == pytest.approx(1.0)
    assert k["hashWindowsKernel<4, true, unsigned int>"]["achieved_GBps"] == pytest.approx(1250.0)
    assert sum(v["share_of_kernel_time"] for v in k.values()) == pytest.approx(1.0)
    assert d["hbm_natured_kernel"]["kernel"] == "hashWindowsKernel<4, true, unsigned int>" and d["hbm_natured_kernel"]["traffic"] == 1.5e9
    assert d["aligner_status"]["stored"] == 999 and d["aligner_status"]["rejected_by_filters"] == 1
    assert d["cpu_baseline"]["cores"] == 64 and d["speedup_vs_cpu_baseline"] == pytest.approx(d["value"] / 18000.0)


def test_counters_of_another_build_are_refused(monkeypatch, tmp_path):
    """A PMC summary collected on other kernel sources prices nothing (round 4 divided round 3's instruction counts by round 4's
    launch times and printed VALU fractions above 1)."""
    import shasta_amd
    pmc = tmp_path / "pmc.json"
    rows = {"k": {"hbm_bytes_per_launch": 1.0, "valu_wave_instructions_per_launch": 2.0}}
    monkeypatch.setattr(bench, "PMC_FILE", str(pmc))
    pmc.write_text(json.dumps({"workload_reads": 100000, "kernel_source_hash": "0123456789abcdef", "kernels": rows}))
    assert bench.load_pmc(100000) == {}
    pmc.write_text(json.dumps({"workload_reads": 100000, "kernels": rows}))
    assert bench.load_pmc(100000) == {}
    pmc.write_text(json.dumps({"workload_reads": 100000, "kernel_source_hash": shasta_amd.kernel_source_hash(), "kernels": rows}))
    assert bench.load_pmc(100000) == rows and bench.load_pmc(20000) == {}


def test_headline_stays_short_whatever_the_details_hold():
    out = {"metric": "m", "value": float("nan"), "unit": "pairs/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "w" * 300, "junk": "x" * 5000}, "kernels": {"k%d" % i: {"a": 1.0} for i in range(500)},
           "roofline": {"bound": "valu", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1 / 8000.0, "traffic": None, "kernel": "k",
                        "note": "n" * 2000, "one_worker": {"frac": 0.1, "note": "n" * 2000}},
           "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "reference (long prose)", "sample": "s" * 2000},
           "stage_seconds_per_step": {"x%d" % i: 0.1 for i in range(400)},
           "dp_tie_sensitive": {"candidates": 1, "per_policy": ["p" * 100] * 11}}
    line = bench.headline(out, "gpurun_out/bench_details.json")
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.FINAL_LINE_LIMIT and strict_loads(text)["value"] is None
    assert line["cpu_baseline"]["kind"] == "reference" and "stage_seconds_per_step" not in line and "junk" not in line["config"]


def test_bench_script_end_to_end_on_the_emulated_build(emu_lib):
    """The real bench.py, real library calls (emulated build), real CPU baseline: every key of the
    driver's contract, the roofline and cpu_baseline objects, and the dry-run marker."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, SHASTA_BENCH_DETAILS=details)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "100", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    head = strict_loads(last)
    assert len(last) < 4096 and head["details"] and head["roofline"]["kernel"] and head["cpu_baseline"]["value"] > 0
    assert head["parity_at_bench_size"]["aligner_mismatches"] == 0
    line = strict_loads(open(details).read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "NOT A MEASUREMENT" in line["data"] and line["n_gpus"] == 1 and line["value"] > 0
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic"]) <= set(line["roofline"])
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["value"] > 0
    parity = line["parity_at_bench_size"]
    assert parity["lowhash0_equal"] is True and parity["aligner_mismatches"] == 0 and parity["aligner_tie_flags_equal"] is True
    # The step is computeAlignments end to end: the alignment table is in it, equal to the reference container's; the CPU leg
    # times computeSortedMarkers and computeAlignmentTable as well.
    assert parity["alignment_table_equal"] is True and "computeAlignmentTable" in line["config"]["step"]
    assert line["cpu_baseline"]["sorted_markers_seconds"] > 0 and line["cpu_baseline"]["alignment_table_seconds"] > 0
    assert any(k.startswith("alignment table") for k in line["kernels"])
    assert line["banded_dp"]["sparse_path"] is True and 0.3 < line["banded_dp"]["share_from_the_matches"] <= 1.0       # (1.0: what the chain kernel does not answer, the anchor kernel does)
    assert line["banded_dp"]["matches_in_the_bands_per_step"] > 0 and line["banded_dp"]["matches_walked_by_the_anchor_kernel_per_step"] is not None
    assert any(k.startswith("align4CellsChunkKernel") for k in line["kernels"]) and any(k.startswith("sparseChainWaveKernel") for k in line["kernels"])
    assert line["config"]["candidates"] > 0 and "workload" in line["config"]
    census = line["dp_tie_sensitive"]             # the checker under the 11 other DP tie policies
    assert census["candidates"] > 0 and len(census["per_policy"]) == 11
    assert census["candidates_changed"] >= census["markerCount_changed"] and census["candidates_changed"] >= census["stored_set_changed"]


def test_bench_script_may2022_workload_on_the_emulated_build(emu_lib):
    """--workload may2022: reads over the k = 14 marker alphabet, conf/Nanopore-May2022.conf's MinHash and Align sections, and
    Assembler::suppressAlignmentCandidates between the two stages (the host layer's emulated twin)."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, SHASTA_BENCH_HOST_LIBRARY=host, SHASTA_BENCH_DETAILS=details)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "100", "--steps", "1", "--warmup", "0", "--workload", "may2022", "--tie-census", "0"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    head = strict_loads(out.stdout.strip().splitlines()[-1])
    assert "NOT A MEASUREMENT" in head["data"] and "Nanopore-May2022.conf" in head["config"]["workload"] and "k = 14" in head["config"]["workload"]
    line = strict_loads(open(details).read())
    parity = line["parity_at_bench_size"]
    assert parity["lowhash0_equal"] is True and parity["aligner_mismatches"] == 0 and parity["alignment_table_equal"] is True
    assert parity["candidates_after_suppression"] <= parity["lowhash0_candidates"] and line["config"]["candidates"] == parity["candidates_after_suppression"] > 0
    assert "candidates_suppressed_between_the_stages" in line["config"]


def test_bench_script_two_ranks_on_the_emulated_build(emu_lib):
    """The N > 1 path of the real bench.py as the driver launches it (torch.distributed.run, one process
    per rank): sharded generation, all-gather of the kmer ids, staged LowHash0 with both exchanges,
    candidate re-split, Align4 -- on the emulated build over gloo.  Control flow only."""
    import os
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, HIPEMU_THREADS="4", SHASTA_BENCH_DETAILS=details)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "100"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    assert len(lines[0]) < 4096 and strict_loads(lines[0])["n_gpus"] == 2 and strict_loads(lines[0])["in_process_group"]["value"] > 0
    line = strict_loads(open(details).read())
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and "cpu_baseline" not in line
    assert line["config"]["candidates"] > 0 and line["config"]["alignments_stored"] > 0
    # Rank 0 measured the same job through the in-process group as well (both drivers on one line), and the line says what a
    # GPU must hold for this run and for BASELINE configs[3] / [4].
    g = line["in_process_group"]
    assert "error" not in g, g
    assert g["candidates"] == line["config"]["candidates"] and g["alignments_stored"] == line["config"]["alignments_stored"] and g["value"] > 0
    budget = line["hbm_budget_per_gpu"]
    assert budget["this_run"]["fits_288_GB"] and budget["configs[4] human 50x, 8 GPUs"]["fits_288_GB"]
    assert budget["configs[4] human 50x, 8 GPUs"]["bytes_per_gpu"]["kmer_ids_of_all_reads"] == 88_000_000_000


def test_bench_script_in_process_group_mode_on_the_emulated_build(emu_lib):
    """bench.py --group: ONE process, --gpus devices behind shasta_mi355x_group (here device 0 of the emulated build twice)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, HIPEMU_THREADS="4")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--group", "--gpus", "2", "--reads", "100", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "NOT A MEASUREMENT" in line["data"] and "in-process group" in line["metric"]
    assert line["in_process_group"]["devices"] == [0, 0] and line["config"]["candidates"] > 0 and line["config"]["alignments_stored"] > 0



def test_bench_script_one_rank_through_the_sharded_branch_on_the_emulated_build(emu_lib):
    """SHASTA_BENCH_FORCE_SHARDED=1: bench.py's N-rank branch with the one rank a one-GPU box allows (on the MI355X over RCCL;
    here over gloo on the emulated build) -- and the JSON line is the last thing on stdout."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, HIPEMU_THREADS="4", SHASTA_BENCH_FORCE_SHARDED="1", SHASTA_BENCH_NO_GROUP_LINE="1",
               MASTER_PORT="29641")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "100", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL all-to-all" in line["config"]["parallelism"]
    assert line["config"]["candidates"] > 0 and line["config"]["alignments_stored"] > 0 and "cpu_baseline" not in line


def test_bench_script_falls_back_on_the_library_switches_after_a_parity_failure(emu_lib):
    """bench.py's safety net: a failed parity check at bench size (forced here) makes the script run again with the library's own
    switch back to the earlier form of its kernels, and the line says so -- never silently."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, SHASTA_BENCH_FORCE_PARITY_FAILURE="1", SHASTA_BENCH_DETAILS=details)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "100", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    head = strict_loads(out.stdout.strip().splitlines()[-1])
    assert "SHASTA_MI355X_ANCHORED_DP=0" in head["path"] and head["earlier_attempts"][0]["switches"] == {}
    line = strict_loads(open(details).read())
    assert "SHASTA_MI355X_ANCHORED_DP=0" in line["path"] and len(line["earlier_attempts"]) == 1 and line["earlier_attempts"][0]["switches"] == {}
    assert line["banded_dp"]["matches_walked_by_the_anchor_kernel_per_step"] is None and line["value"] > 0
    assert "FAILED on this path" in out.stderr


def test_fallback_switches_one_step_at_a_time(monkeypatch):
    import bench
    for k in bench.FALLBACK_SWITCHES + ("SHASTA_BENCH_FORCE_PARITY_FAILURE", "SHASTA_BENCH_EARLIER_ATTEMPTS"):
        monkeypatch.delenv(k, raising=False)
    good = {"lowhash0_equal": True, "aligner_mismatches": 0, "alignment_table_equal": True, "aligner_tie_flags_equal": True}
    assert bench._fallback_environment(good, None) is None and bench._fallback_environment(None, None) is None
    assert bench._fallback_environment(dict(good, aligner_mismatches=3), None) == {"SHASTA_MI355X_ANCHORED_DP": "0"}
    assert bench._fallback_environment(dict(good, lowhash0_equal=False), None) == {"SHASTA_MI355X_STATISTICS_ATOMICS": "1"}
    monkeypatch.setenv("SHASTA_MI355X_ANCHORED_DP", "0")
    assert bench._fallback_environment(dict(good, alignment_table_equal=False), None) == {"SHASTA_MI355X_SPARSE_DP": "0"}
    monkeypatch.setenv("SHASTA_MI355X_SPARSE_DP", "0")
    assert bench._fallback_environment(dict(good, aligner_mismatches=1), None) is None               # nothing left to switch: the line stands as it is
    monkeypatch.delenv("SHASTA_MI355X_ANCHORED_DP"); monkeypatch.delenv("SHASTA_MI355X_SPARSE_DP")
    assert bench._fallback_environment(None, RuntimeError("HIP error")) == {"SHASTA_MI355X_STATISTICS_ATOMICS": "1", "SHASTA_MI355X_SPARSE_DP": "0"}
