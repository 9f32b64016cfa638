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


class FakeTimes:
    def __init__(self):
        self.lowhashHashSeconds, self.lowhashHashLaunches, self.lowhashHashBytes = 0.01, 10, 10 * 1_250_000_000
        self.alignDpSeconds, self.alignDpLaunches, self.alignDpCells, self.alignBytes = 0.4, 80, 190_000_000_000, 41_000_000_000
        self.dpForwardSeconds = [0.04, 0.14, 0.10, 0.05, 0.003, 0.0]
        self.dpForwardLaunches = [16, 16, 16, 16, 1, 0]
        self.dpForwardCells = [int(1.5e10), int(1.1e11), int(5.5e10), int(1e9), int(1e6), 0]
        self.dpForwardBytes = [16 * 296_000_000, 16 * 1_186_000_000, 16 * 353_000_000, 16 * 4_600_000, 33_000, 0]
        self.dpTracebackSeconds, self.dpTracebackLaunches = 0.08, 16


class FakeResult:
    def __init__(self, n):
        self.candidates = abi.make_pairs(np.arange(n), np.arange(n) + 1, np.ones(n))
        self.alignment_data = np.zeros(n - 1, dtype=abi.ALIGNMENT_DATA_DTYPE)
        self.device_seconds, self.seconds = 0.4, 0.5


class FakeContext:
    def set_kmer_ids(self, toc, kmer):
        pass

    def lowhash0(self, p):
        return FakeResult(1000)

    def align4(self, candidates, o, want_ordinals=False, borrow=False):
        return FakeResult(len(candidates))

    def kernel_times(self):
        return FakeTimes()

    def close(self):
        pass


class FakeLibrary:
    version = 1

    def dp_forward_version(self):
        return self.version

    def device_count(self):
        return 1

    def context(self, device):
        return FakeContext()


@pytest.mark.parametrize("dp_version", [1, 2])
def test_bench_prints_one_contract_line(monkeypatch, capsys, dp_version):
    import torch
    import shasta_amd
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(FakeLibrary, "version", dp_version)
    monkeypatch.setattr(shasta_amd, "load", lambda: FakeLibrary())
    monkeypatch.setattr(bench, "make_workload", lambda reads, seed: (np.zeros(2 * 10 + 1, np.uint64), np.zeros(0, np.uint32)))
    monkeypatch.setattr(bench, "cpu_baseline", lambda reads, seed, method=4: {"value": 18000.0, "unit": "candidate read-pairs aligned/s",
                                                                     "cores": 64, "kind": "reference", "sample": "fake"})
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--reads", "100000"])
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # The dominant kernel of the fake timings is the <=64-diagonal DP class; the traffic in the committed PMC file was
    # measured on the first version of the kernel and is not reported for the second.
    if dp_version == 1:
        assert r["kernel"] == "bandedDpForwardKernel<32, 2>" and r["traffic"] and r["traffic"] > 1e9
    else:
        assert r["kernel"] == "bandedDpForwardKernel2<32, 2>" and r["traffic"] is None and r["dp_forward_version"] == 2
    assert d["kernels"]["hashWindowsKernel<4>"]["launches_per_step"] == 10
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] == 64
    assert d["speedup_vs_cpu_baseline"] == pytest.approx(d["value"] / 18000.0)


def test_bench_script_end_to_end_on_the_emulated_build(emu_lib):
    """The real bench.py, real library calls (emulated build), real CPU baseline: every key of the
    driver's contract, the roofline and cpu_baseline objects, and the dry-run marker."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "150", "--steps", "1", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "NOT A MEASUREMENT" in line["data"] and line["n_gpus"] == 1 and line["value"] > 0
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic"]) <= set(line["roofline"])
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["value"] > 0
    assert line["config"]["candidates"] > 0 and "workload" in line["config"]


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
    env = dict(os.environ, SHASTA_BENCH_LIBRARY=emu_lib.path, HIPEMU_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "150"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and "cpu_baseline" not in line
    assert line["config"]["candidates"] > 0 and line["config"]["alignments_stored"] > 0

